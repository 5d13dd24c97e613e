#!/bin/bash
# Round-4 evidence, the dominant kernel after its last change: bench line, its kernel statistics, SQ counters (own passes).
export TMPDIR=/tmp
OUT=gpurun_out/r04
mkdir -p $OUT
timeout 900 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python bench.py --no-cpu-baseline --no-extras --no-pmc > $OUT/bench_under_rocprof.json 2>/dev/null
rm -f $OUT/cost_kernel_pmc.txt
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pmc_dt_A -o x -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-pmc > /dev/null 2>&1)
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d /tmp/pmc_dt_B -o x -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-pmc > /dev/null 2>&1)
(echo "== dt"; python tools/pmc_summary.py /tmp/pmc_dt_A k_cost_pairs; python tools/pmc_summary.py /tmp/pmc_dt_B k_cost_pairs) >> $OUT/cost_kernel_pmc.txt 2>&1
timeout 300 python bench.py --mode adam --no-cpu-baseline --no-extras > $OUT/bench_n1_adam.json 2>/dev/null
python -m pytest tests/test_gpu_pairs.py tests/test_gpu_window_gn.py tests/test_gpu_sequence.py tests/test_gpu_drivers.py -m gpu -q 2>&1 | tail -1
cat $OUT/cost_kernel_pmc.txt | cut -c1-300
