#!/bin/bash
# Round-4 evidence on a GPU box:  tools/collect_r04.sh   (outputs under gpurun_out/r04/; tools/refresh_r04.py copies the summaries
# into profiles/).  HBM traffic of the dominant kernel is measured by bench.py itself (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one
# counter per pass, --kernel-trace only).
export TMPDIR=/tmp
OUT=gpurun_out/r04
mkdir -p $OUT
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6) > $OUT/pytest.txt
(timeout 400 python -m pytest tests/test_gpu_sigma05.py tests/test_gpu_window_gn.py tests/test_gpu_sequence.py tests/test_gpu_rccl.py tests/test_gpu_drivers.py -m gpu -q -s 2>&1 \
   | grep -v "^make\|amdgpu.ids\|^$\|^   per-frame\|^   frame" | cut -c1-1500) > $OUT/parity.txt
(timeout 200 python -m pytest tests/test_gpu_pairs.py tests/test_gpu_fullsize.py -m gpu -q -s -k "slot_level or config5_as_a_batch" 2>&1 | grep "pairs\|config 5\|passed" | cut -c1-600) >> $OUT/parity.txt
timeout 900 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python bench.py --no-cpu-baseline --no-extras --no-pmc > $OUT/bench_under_rocprof.json 2>/dev/null
timeout 300 python bench.py --no-depth-table --no-cpu-baseline --no-extras > $OUT/bench_n1_log_depth_tables.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_logtab -o bench -- python bench.py --no-depth-table --no-cpu-baseline --no-extras --no-pmc > /dev/null 2>&1
timeout 300 python bench.py --mode adam --no-cpu-baseline --no-extras > $OUT/bench_n1_adam.json 2>/dev/null
timeout 500 python bench.py --segments 128 --no-cpu-baseline > $OUT/bench_n1_seg128.json 2>/dev/null
# instruction counts of the dominant kernel, depth tables against log-depth tables (SQ counters, their own passes, --kernel-trace only)
for tab in "" "--no-depth-table"; do
  tag=dt; [ -n "$tab" ] && tag=logtab
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pmc_${tag}_A -o x -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-pmc $tab > /dev/null 2>&1)
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d /tmp/pmc_${tag}_B -o x -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-pmc $tab > /dev/null 2>&1)
  (echo "== $tag"; python tools/pmc_summary.py /tmp/pmc_${tag}_A k_cost_pairs; python tools/pmc_summary.py /tmp/pmc_${tag}_B k_cost_pairs) >> $OUT/cost_kernel_pmc.txt 2>&1
done
timeout 200 python tools/kbench.py --pairs 384 --tile-points 8192 --granule 64 --ab-depth-table --reps 60 2>/dev/null | grep level > $OUT/kbench_depth_table_ab.txt
(timeout 300 python tools/phase_sweep.py 2>&1 | grep "pairs/s"; echo "--- from the reference start"; timeout 300 python tools/phase_sweep.py --reference-start 2>&1 | grep "pairs/s") > $OUT/phase_sweep.txt
(timeout 300 python tools/reference_start_sweep.py 2>&1 | grep "pose-only"; timeout 200 python tools/hard_starts.py 2>&1 | grep hard) > $OUT/reference_start.txt
timeout 300 python tools/window_bench.py 1 2 3 4 2>&1 | grep -v "^make\|amdgpu.ids" > $OUT/window_bench.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_window -o wb -- python tools/window_bench.py 2 > /dev/null 2>&1
timeout 400 python tools/run_configs.py 2>/dev/null | grep config > $OUT/configs.txt
timeout 300 python tools/stream_bench.py 384 3 2>&1 | grep batches > $OUT/stream_bench.txt
SP_GRANULE=64 timeout 200 python tools/setup_bench.py 2>/dev/null | grep "set-up\|timeline" > $OUT/setup.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_setup -o setup -- python tools/setup_profile.py 128 > /dev/null 2>&1
# package power and shader clock across a 12 s run of the bench step
(python bench.py --steps 12000 --warmup 10 --no-cpu-baseline --no-extras --no-pmc > $OUT/bench_long.json 2>/dev/null &)
for i in $(seq 1 18); do sleep 1; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Package Power" | sed "s/.*: //" | tr "\n" " "; echo; done > $OUT/power_clock_trace.txt
wait
tail -3 $OUT/pytest.txt; cat $OUT/configs.txt | cut -c1-220; cat $OUT/setup.txt; cat $OUT/cost_kernel_pmc.txt | cut -c1-400
