for v in 0 1 0 1; do echo "side fill $v"; SP_SETUP_SIDE_FILL=$v python tools/setup_bench.py 384 2>&1 | grep "granule\|timeline" | cut -c1-420; done
SP_SETUP_SIDE_FILL=1 python -m pytest tests/test_gpu_pairs.py -m gpu -q -x -k "prepar or raw or stream or table" 2>&1 | tail -3
