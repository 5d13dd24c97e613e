#!/bin/bash
# Round-3 evidence on a GPU box:  tools/collect_r03.sh   (outputs under gpurun_out/r03/; tools/refresh_r03.py copies the summaries
# into profiles/).  HBM traffic of the dominant kernel is measured by bench.py itself (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one
# counter per pass, --kernel-trace only).
export TMPDIR=/tmp
OUT=gpurun_out/r03
mkdir -p $OUT
(timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -6) > $OUT/pytest.txt
(timeout 300 python -m pytest tests/test_gpu_sigma05.py tests/test_gpu_window_gn.py tests/test_gpu_sequence.py -m gpu -q -s 2>&1 | grep -v "^make\|amdgpu.ids\|^$") > $OUT/parity.txt
timeout 600 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python bench.py --no-cpu-baseline --no-extras --no-pmc > $OUT/bench_under_rocprof.json 2>/dev/null
timeout 300 python bench.py --mode adam --no-cpu-baseline --no-extras > $OUT/bench_n1_adam.json 2>/dev/null
timeout 300 python bench.py --granule 64 --no-cpu-baseline --sigma05-scenes 0 > $OUT/bench_n1_g64.json 2>/dev/null
timeout 400 python bench.py --segments 128 --no-cpu-baseline --sigma05-scenes 0 > $OUT/bench_n1_seg128.json 2>/dev/null
for N in 64 300 1200; do
  for G in 256 64; do
    timeout 400 python bench.py --shape blobs --segments $N --granule $G --no-cpu-baseline --sigma05-scenes 0 > $OUT/bench_blobs_${N}_g$G.json 2>/dev/null
  done
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_blobs_1200_g64 -o bench -- python bench.py --shape blobs --segments 1200 --granule 64 --no-cpu-baseline --no-extras --no-pmc > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_blobs_1200_g256 -o bench -- python bench.py --shape blobs --segments 1200 --no-cpu-baseline --no-extras --no-pmc > /dev/null 2>&1
timeout 300 python tools/sigma05_sweep.py --size 240x320x8 --scenes 12 --variants r02,pose1st,pose1st_10,pose1st_25,pose1st_allpts,L4,it40,tol5e-4,lam1e-2,eps1e-2 > $OUT/sigma05_sweep_small.txt 2>/dev/null
timeout 400 python tools/sigma05_sweep.py --size 480x640x64 --scenes 64 --seed0 1000 --variants r02,pose1st,pose1st_10,pose1st_25,pose1st_allpts,pose1st_L4,L4,it40,tol5e-4,lam1e-2,eps1e-2 > $OUT/sigma05_sweep_full.txt 2>/dev/null
timeout 400 python tools/run_configs.py 2>/dev/null | grep config > $OUT/configs.txt
timeout 300 python tools/stream_bench.py 2>&1 | grep batches > $OUT/stream_bench.txt
SP_GRANULE=64 timeout 200 python tools/setup_bench.py 2>/dev/null | grep "set-up\|timeline" > $OUT/setup.txt
timeout 200 python tools/setup_profile.py 128 2>/dev/null | grep "PairBatch of\|run_scheduled\|build:" >> $OUT/setup.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_setup -o setup -- python tools/setup_profile.py 128 > /dev/null 2>&1
# where the set-up kernels' wave cycles go (SQ counters, their own passes, --kernel-trace only)
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pmcA -o x -- python $GRAFT_REPO_ROOT/tools/setup_profile.py 128 > /dev/null 2>&1)
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d /tmp/pmcB -o x -- python $GRAFT_REPO_ROOT/tools/setup_profile.py 128 > /dev/null 2>&1)
(python tools/pmc_summary.py /tmp/pmcA k_prep; python tools/pmc_summary.py /tmp/pmcB k_prep) > $OUT/setup_pmc.txt 2>&1
timeout 200 python tools/kbench.py --pairs 384 --tile-points 8192 --modes 1,16,1,16,1,16,0 --reps 40 2>/dev/null | grep level > $OUT/kbench_pixless.txt
# package power and shader clock across a 12 s run of the bench step
(python bench.py --steps 12000 --warmup 10 --no-cpu-baseline --no-extras --no-pmc > $OUT/bench_long.json 2>/dev/null &)
for i in $(seq 1 18); do sleep 1; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Package Power" | sed "s/.*: //" | tr "\n" " "; echo; done > $OUT/power_clock_trace.txt
wait
tail -3 $OUT/pytest.txt; cat $OUT/configs.txt | cut -c1-220; cat $OUT/setup.txt
