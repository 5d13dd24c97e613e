mkdir -p gpurun_out/r06ae
for tp in 8192 2048 1024 512 256 0 8192; do echo "== SP_WINDOW_TILE_POINTS=$tp"; SP_WINDOW_TILE_POINTS=$tp python tools/chain_profile.py 64 native 2>&1 | grep "native chain" | tail -2; done > gpurun_out/r06ae/tile_points.txt
cat gpurun_out/r06ae/tile_points.txt
