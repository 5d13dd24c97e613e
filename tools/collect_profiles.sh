#!/bin/bash
# Re-creates the evidence under profiles/ on a GPU box:  tools/collect_profiles.sh r01
# (bench line, rocprofv3 kernel stats of the same command, PMC passes for HBM traffic -- one counter group per pass,
#  --kernel-trace only, as MI355X_MICROARCH.md prescribes)
set -e
R=${1:-r01}
export TMPDIR=/tmp
OUT=gpurun_out/$R
mkdir -p $OUT
python bench.py > $OUT/bench_n1.json 2>/dev/null
python bench.py --mode adam --no-cpu-baseline --no-extras > $OUT/bench_n1_adam.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python bench.py --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.json 2>/dev/null
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$tag -o bench -- python bench.py --settle-ms 0 --steps 10 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
done
python tools/kbench.py --pairs 384 --tile-points 8192 --modes 1,0,1,0,10,11,12,13 --reps 40 2>/dev/null | grep level > $OUT/kbench_ablation.txt
python tools/run_configs.py 2>/dev/null | grep config > $OUT/configs.txt
# package power and shader clock, sampled once a second across a 16 s run (16000 steps of 384 pairs) of the bench step
(python bench.py --steps 16000 --warmup 10 --no-cpu-baseline --no-extras > $OUT/bench_long.json 2>/dev/null &)
for i in $(seq 1 24); do sleep 1; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Package Power" | sed "s/.*: //" | tr "\n" " "; echo; done > $OUT/power_clock_trace.txt
wait
ls $OUT
# BASELINE configs[4]'s pair shape (128 segments per keyframe): bench line + kernel stats of the same command
python bench.py --segments 128 --no-cpu-baseline > $OUT/bench_n1_seg128.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats128 -o bench -- python bench.py --segments 128 --no-cpu-baseline --no-extras > /dev/null 2>&1
python tools/parity_report.py > $OUT/parity.txt 2>/dev/null
python tools/power_by_mode.py --modes 1,14,11,13,0 --seconds 4 2>/dev/null | grep mode > $OUT/power_by_mode.txt
# set-up of frame pairs from raw frames (optim/batch_prepare.py): timings and the kernel stats of the same command
python tools/setup_profile.py 128 2>/dev/null | grep "PairBatch of\|run_scheduled\|build:" > $OUT/setup.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_setup -o setup -- python tools/setup_profile.py 128 > /dev/null 2>&1
ls $OUT
