#!/usr/bin/env python
"""Copies the evidence tools/collect_r03.sh (+ the final bench runs) left under gpurun_out/r03/ into profiles/ (tracked)."""
import os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R, P = os.path.join(ROOT, "gpurun_out", "r03"), os.path.join(ROOT, "profiles")
pairs = [("bench_n1.json", "bench_n1.json"), ("bench_n1_g256.json", "bench_n1_granule256.json"), ("bench_n1_adam.json", "bench_n1_adam.json"),
         ("bench_n1_seg128.json", "bench_n1_seg128.json"), ("bench_under_rocprof.json", "bench_n1_under_rocprof.json"),
         ("bench_long.json", "bench_n1_long_run.json"), ("stats/bench_kernel_stats.csv", "bench_n1_kernel_stats.csv"),
         ("stats_g256/bench_kernel_stats.csv", "bench_n1_granule256_kernel_stats.csv"),
         ("stats_blobs_1200_g64/bench_kernel_stats.csv", "bench_blobs_1200_g64_kernel_stats.csv"),
         ("stats_blobs_1200_g256/bench_kernel_stats.csv", "bench_blobs_1200_g256_kernel_stats.csv"),
         ("stats_setup/setup_kernel_stats.csv", "setup_kernel_stats.csv"), ("configs.txt", "configs.txt"), ("parity.txt", "parity.txt"),
         ("sigma05_sweep_small.txt", "sigma05_sweep_320x240x8.txt"), ("sigma05_sweep_full.txt", "sigma05_sweep_640x480x64.txt"),
         ("stream_bench.txt", "stream_bench.txt"), ("setup.txt", "setup.txt"), ("setup_pmc.txt", "setup_pmc.txt"), ("kbench_pixless.txt", "kbench_pixless.txt"),
         ("kbench_granule_ab.txt", "kbench_granule_ab.txt"), ("power_clock_trace.txt", "power_clock_trace.txt")]
for N in (64, 300, 1200):
    for G in (256, 64):
        pairs.append((f"bench_blobs_{N}_g{G}.json", f"bench_blobs_{N}_g{G}.json"))
for a, b in pairs:
    src = os.path.join(R, a)
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, "r03_" + b))
        print("copied", a)
    else:
        print("MISSING", a)
