export TMPDIR=/tmp
O=gpurun_out/s2; mkdir -p $O
(timeout 300 python -m pytest tests/test_gpu_pairs.py tests/test_gpu_edge_cases.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -4) > $O/pytest0.txt
SP_GRANULE=64 timeout 200 python tools/setup_bench.py 2>/dev/null | grep "set-up" > $O/setup.txt
cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pmcA -o x -- python $GRAFT_REPO_ROOT/tools/setup_profile.py 128 > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d /tmp/pmcB -o x -- python $GRAFT_REPO_ROOT/tools/setup_profile.py 128 > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -o x -- python $GRAFT_REPO_ROOT/tools/setup_profile.py 128 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py /tmp/pmcA k_prep > $O/pmcA.txt 2>&1
python tools/pmc_summary.py /tmp/pmcB k_prep > $O/pmcB.txt 2>&1
cp /tmp/st/*/*kernel_stats.csv $O/ 2>/dev/null || find /tmp/st -name "*kernel_stats.csv" -exec cp {} $O/ \;
cat $O/pytest0.txt $O/setup.txt $O/pmcA.txt $O/pmcB.txt; head -12 $O/*kernel_stats.csv | cut -c1-150
