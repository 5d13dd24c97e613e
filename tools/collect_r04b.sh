#!/bin/bash
# Round-4 evidence, final pass (after the VALU diet of k_cost_pairs, the wave-private fill pass and the host-side work of the set-up):
# refreshes what those changed under gpurun_out/r04/; tools/refresh_r04.py copies the summaries into profiles/.
export TMPDIR=/tmp
OUT=gpurun_out/r04
mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -s > /tmp/pytest_full.txt 2>&1
tail -4 /tmp/pytest_full.txt > $OUT/pytest.txt
grep -v "^make\|amdgpu.ids\|^$\|^   per-frame\|^   frame\|^hipcc" /tmp/pytest_full.txt | cut -c1-1500 > $OUT/parity.txt
timeout 900 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python bench.py --no-cpu-baseline --no-extras --no-pmc > $OUT/bench_under_rocprof.json 2>/dev/null
timeout 300 python bench.py --no-depth-table --no-cpu-baseline --no-extras > $OUT/bench_n1_log_depth_tables.json 2>/dev/null
timeout 300 python bench.py --mode adam --no-cpu-baseline --no-extras > $OUT/bench_n1_adam.json 2>/dev/null
rm -f $OUT/cost_kernel_pmc.txt
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pmc_dt_A -o x -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-pmc > /dev/null 2>&1)
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d /tmp/pmc_dt_B -o x -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-pmc > /dev/null 2>&1)
(echo "== dt"; python tools/pmc_summary.py /tmp/pmc_dt_A k_cost_pairs; python tools/pmc_summary.py /tmp/pmc_dt_B k_cost_pairs) >> $OUT/cost_kernel_pmc.txt 2>&1
timeout 200 python tools/kbench.py --pairs 384 --tile-points 8192 --granule 64 --ab-depth-table --reps 60 2>/dev/null | grep level > $OUT/kbench_depth_table_ab.txt
timeout 400 python tools/run_configs.py 2>/dev/null | grep config > $OUT/configs.txt
timeout 300 python tools/stream_bench.py 384 3 2>&1 | grep batches > $OUT/stream_bench.txt
SP_GRANULE=64 timeout 200 python tools/setup_bench.py 2>/dev/null | grep "set-up\|timeline" > $OUT/setup.txt
timeout 200 python tools/host_profile.py 384 2>/dev/null | grep -v "^$" | cut -c1-170 | head -34 > $OUT/host_profile.txt
bash tools/fill_check.sh r04_fill pmc > /dev/null 2>&1
cp gpurun_out/r04_fill/kernels.txt $OUT/setup_kernels_128.txt; cp gpurun_out/r04_fill/pmc.txt $OUT/fill_pmc.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_setup -o setup -- python tools/setup_profile.py 128 > /dev/null 2>&1
(python bench.py --steps 12000 --warmup 10 --no-cpu-baseline --no-extras --no-pmc > $OUT/bench_long.json 2>/dev/null &)
for i in $(seq 1 18); do sleep 1; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Package Power" | sed "s/.*: //" | tr "\n" " "; echo; done > $OUT/power_clock_trace.txt
wait
cat $OUT/pytest.txt; cat $OUT/configs.txt | cut -c1-220; cat $OUT/setup.txt; cat $OUT/cost_kernel_pmc.txt | cut -c1-400; cat $OUT/stream_bench.txt
