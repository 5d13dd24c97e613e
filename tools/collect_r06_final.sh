#!/bin/bash
# Re-creates the round-6 headline evidence on a GPU box:  bash tools/collect_r06_final.sh   (-> gpurun_out/r06late/, copied by hand into profiles/r06_*)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06late; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.json 2>/dev/null
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$tag -o bench -- python $GRAFT_REPO_ROOT/bench.py --settle-ms 0 --steps 10 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/pmc_$tag k_cost_pairs >> $OUT/cost_kernel_pmc.txt 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/setup_bench.py 384 2>&1 | grep "granule\|timeline" > $OUT/setup_bench.txt
python tools/run_configs.py 2>/dev/null | grep config > $OUT/configs.txt
grep -h "k_cost_pairs\|k_pairs_gn" $OUT/stats/*kernel_stats.csv | head -5 | cut -c1-200
python - <<'PY'
import json,os
d=json.loads(open(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r06late/bench_driver.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ['value','ms_per_step','frame_pairs_per_sec','frame_pairs_per_sec_ragged_masks','frame_pairs_per_sec_sam_masks','frame_pairs_per_sec_sam_masks_sustained','frame_pairs_per_sec_from_raw_frames']}, d['roofline']['frac'], d['roofline'].get('kernel_ms'))
PY
cat $OUT/smoke.txt | tail -3
