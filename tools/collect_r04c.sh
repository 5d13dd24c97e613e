#!/bin/bash
# Round-4 evidence, last refresh (after the count pass's non-temporal loads and the wave-level row scan): the test suite, the bench line, its
# kernel statistics and the set-up timings under gpurun_out/r04/; tools/refresh_r04.py copies the summaries into profiles/.
export TMPDIR=/tmp
OUT=gpurun_out/r04
mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -s > /tmp/pytest_full.txt 2>&1
tail -4 /tmp/pytest_full.txt > $OUT/pytest.txt
grep -v "^make\|amdgpu.ids\|^$\|^   per-frame\|^   frame\|^hipcc" /tmp/pytest_full.txt | cut -c1-1500 > $OUT/parity.txt
timeout 900 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python bench.py --no-cpu-baseline --no-extras --no-pmc > $OUT/bench_under_rocprof.json 2>/dev/null
SP_GRANULE=64 timeout 200 python tools/setup_bench.py 2>/dev/null | grep "set-up\|timeline" > $OUT/setup.txt
timeout 200 python tools/host_profile.py 384 2>/dev/null | grep -v "^$" | cut -c1-170 | head -34 > $OUT/host_profile.txt
bash tools/fill_check.sh r04_fill pmc > /dev/null 2>&1
cp gpurun_out/r04_fill/kernels.txt $OUT/setup_kernels_128.txt; cp gpurun_out/r04_fill/pmc.txt $OUT/fill_pmc.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_setup -o setup -- python tools/setup_profile.py 128 > /dev/null 2>&1
cat $OUT/pytest.txt; cat $OUT/setup.txt | cut -c1-300; cat $OUT/setup_kernels_128.txt
