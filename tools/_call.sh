export TMPDIR=/tmp
O=gpurun_out/s10; mkdir -p $O
(timeout 400 python -m pytest tests/test_gpu_pairs.py tests/test_gpu_edge_cases.py -m gpu -q -x 2>&1 | tail -3) > $O/pytest0.txt
SP_GRANULE=64 timeout 200 python tools/setup_bench.py 2>/dev/null | grep "set-up\|timeline" > $O/setup.txt
cat $O/pytest0.txt $O/setup.txt
