#!/usr/bin/env python
"""Copies the evidence tools/collect_r04.sh left under gpurun_out/r04/ into profiles/ (tracked)."""
import os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R, P = os.path.join(ROOT, "gpurun_out", "r04"), os.path.join(ROOT, "profiles")
pairs = [("bench_n1.json", "bench_n1.json"), ("bench_n1_log_depth_tables.json", "bench_n1_log_depth_tables.json"), ("bench_n1_adam.json", "bench_n1_adam.json"),
         ("bench_n1_seg128.json", "bench_n1_seg128.json"), ("bench_under_rocprof.json", "bench_n1_under_rocprof.json"), ("bench_long.json", "bench_n1_long_run.json"),
         ("stats/bench_kernel_stats.csv", "bench_n1_kernel_stats.csv"), ("stats_logtab/bench_kernel_stats.csv", "bench_n1_log_depth_tables_kernel_stats.csv"),
         ("stats_window/wb_kernel_stats.csv", "window_bench_kernel_stats.csv"), ("stats_setup/setup_kernel_stats.csv", "setup_kernel_stats.csv"),
         ("configs.txt", "configs.txt"), ("parity.txt", "parity.txt"), ("pytest.txt", "pytest.txt"), ("cost_kernel_pmc.txt", "cost_kernel_pmc.txt"),
         ("kbench_depth_table_ab.txt", "kbench_depth_table_ab.txt"), ("window_bench.txt", "window_bench.txt"), ("phase_sweep.txt", "phase_sweep.txt"),
         ("reference_start.txt", "reference_start.txt"), ("stream_bench.txt", "stream_bench.txt"), ("setup.txt", "setup.txt"), ("power_clock_trace.txt", "power_clock_trace.txt"),
         ("host_profile.txt", "setup_host_profile.txt"), ("setup_kernels_128.txt", "setup_kernels_128_keyframes.txt"), ("fill_pmc.txt", "fill_pmc.txt")]
for a, b in pairs:
    src = os.path.join(R, a)
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, "r04_" + b))
        print("copied", a)
    else:
        print("MISSING", a)
