#!/bin/bash
# Round-5 evidence on a GPU box:  tools/collect_r05.sh   (outputs under gpurun_out/r05/; tools/refresh_r05.py copies the summaries into
# profiles/).  HBM traffic of the dominant kernel is measured by bench.py itself (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter per
# pass, --kernel-trace only).
export TMPDIR=/tmp
OUT=gpurun_out/r05
mkdir -p $OUT
(timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -6) > $OUT/pytest.txt
(timeout 600 python -m pytest tests/test_gpu_sigma05.py tests/test_gpu_window_gn.py tests/test_gpu_sequence.py tests/test_gpu_rccl.py tests/test_gpu_drivers.py -m gpu -q -s 2>&1 \
   | grep -v "^make\|amdgpu.ids\|^$\|^   per-frame\|^   frame\|^   keyframe\|^   mapping" | cut -c1-1500) > $OUT/parity.txt
(timeout 300 python -m pytest tests/test_gpu_pairs.py tests/test_gpu_fullsize.py -m gpu -q -s -k "slot_level or config5_as_a_batch or verdict or box" 2>&1 | grep "pairs\|config 5\|passed" | cut -c1-600) >> $OUT/parity.txt
timeout 900 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python bench.py --no-cpu-baseline --no-extras --no-pmc > $OUT/bench_under_rocprof.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-pmc > $OUT/bench_n1_driver_flags.json 2>/dev/null
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2>/dev/null      # the driver's own command (38 s)
timeout 300 python bench.py --mode adam --no-cpu-baseline --no-extras > $OUT/bench_n1_adam.json 2>/dev/null
timeout 500 python bench.py --segments 128 --no-cpu-baseline > $OUT/bench_n1_seg128.json 2>/dev/null
for S in 64 300 1200; do
  timeout 900 python bench.py --shape blobs --segments $S --no-cpu-baseline --no-pmc > $OUT/bench_blobs_${S}.json 2> /dev/null
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/stats_blobs64 -o bench -- python $GRAFT_REPO_ROOT/bench.py --shape blobs --segments 64 --no-cpu-baseline --no-extras --no-pmc > /dev/null 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/stats_blobs300 -o bench -- python $GRAFT_REPO_ROOT/bench.py --shape blobs --segments 300 --no-cpu-baseline --no-extras --no-pmc > /dev/null 2>&1)
# instruction counts of the dominant kernel (SQ counters, their own passes, --kernel-trace only)
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pmc_A -o x -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --min-timed-ms 0 --no-cpu-baseline --no-extras --no-pmc > /dev/null 2>&1)
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d /tmp/pmc_B -o x -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --min-timed-ms 0 --no-cpu-baseline --no-extras --no-pmc > /dev/null 2>&1)
(python tools/pmc_summary.py /tmp/pmc_A k_cost_pairs; python tools/pmc_summary.py /tmp/pmc_B k_cost_pairs) > $OUT/cost_kernel_pmc.txt 2>&1
# the verdict over many reference starts: grid on 384 and 768 slots, ragged masks
(echo "### grid tiling, 9216 starts, 384 slots"; timeout 600 python tools/verdict_sweep.py --variants shipped,no_retry,undamped 2>&1 | grep -v "^make\|amdgpu.ids\|^batch\|^rendered" | cut -c1-700
 echo "### grid tiling, 9216 starts, 768 slots"; timeout 600 python tools/verdict_sweep.py --slots 768 --alone "" --variants shipped,undamped 2>&1 | grep -v "^make\|amdgpu.ids\|^batch\|^rendered" | cut -c1-700
 echo "### ragged masks (64 overlapping ellipses, rho 1.2), 12288 starts, 384 slots"; timeout 900 python tools/verdict_sweep.py --shape blobs --starts 12288 --alone "" --variants shipped,no_retry,undamped 2>&1 | grep -v "^make\|amdgpu.ids\|^batch\|^rendered" | cut -c1-700) > $OUT/reference_start.txt
timeout 400 python tools/run_configs.py 2>/dev/null | grep config > $OUT/configs.txt
timeout 300 python tools/window_bench.py 1 2 3 4 2>&1 | grep -v "^make\|amdgpu.ids" > $OUT/window_bench.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/stats_window -o w -- python $GRAFT_REPO_ROOT/tools/window_bench.py 2 > /dev/null 2>&1)
timeout 300 python tools/chain_profile.py 40 2>&1 | grep -v "^make\|amdgpu.ids" | head -64 | cut -c1-200 > $OUT/chain_profile.txt
timeout 300 python tools/stream_bench.py 384 3 2>&1 | grep batches > $OUT/stream_bench.txt
SP_GRANULE=64 timeout 200 python tools/setup_bench.py 2>/dev/null | grep "set-up\|timeline" > $OUT/setup.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_setup -o setup -- python tools/setup_profile.py 128 > /dev/null 2>&1
# package power and shader clock across a 12 s run of the bench step
(python bench.py --steps 12000 --warmup 10 --min-timed-ms 0 --no-cpu-baseline --no-extras --no-pmc > $OUT/bench_long.json 2>/dev/null &)
for i in $(seq 1 18); do sleep 1; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Package Power" | sed "s/.*: //" | tr "\n" " "; echo; done > $OUT/power_clock_trace.txt
wait
tail -3 $OUT/pytest.txt; cat $OUT/configs.txt | cut -c1-220; cat $OUT/cost_kernel_pmc.txt | cut -c1-400
python - <<'PY'
import json
for f in ("bench_n1", "bench_under_rocprof", "bench_n1_driver_flags", "bench_n1_seg128", "bench_blobs_64", "bench_blobs_300", "bench_blobs_1200"):
    try:
        d = json.load(open(f"gpurun_out/r05/{f}.json"))
        print(f, round(d["value"]), round(d["roofline"]["frac"], 4), d.get("frame_pairs_per_sec"), d.get("frame_pairs_per_sec_ragged_masks"), d.get("timed_regions"))
    except Exception as e:
        print(f, "no line", e)
PY
