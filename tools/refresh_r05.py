#!/usr/bin/env python
"""Copies the evidence tools/collect_r05.sh (and the earlier round-5 calls it names) left under gpurun_out/ into profiles/ (tracked)."""
import os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
pairs = [("r05/bench_n1.json", "bench_n1.json"), ("r05/bench_under_rocprof.json", "bench_n1_under_rocprof.json"), ("r05/bench_n1_driver_flags.json", "bench_n1_driver_flags.json"), ("r05/bench_driver_command.json", "bench_n1_driver_command.json"),
         ("r05/bench_n1_adam.json", "bench_n1_adam.json"), ("r05/bench_n1_seg128.json", "bench_n1_seg128.json"), ("r05/bench_long.json", "bench_n1_long_run.json"),
         ("r05/bench_blobs_64.json", "bench_blobs_64.json"), ("r05/bench_blobs_300.json", "bench_blobs_300.json"), ("r05/bench_blobs_1200.json", "bench_blobs_1200.json"),
         ("r05/stats/bench_kernel_stats.csv", "bench_n1_kernel_stats.csv"), ("r05/stats_blobs64/bench_kernel_stats.csv", "bench_blobs_64_kernel_stats.csv"),
         ("r05/stats_blobs300/bench_kernel_stats.csv", "bench_blobs_300_kernel_stats.csv"), ("r05/stats_setup/setup_kernel_stats.csv", "setup_kernel_stats.csv"),
         ("r05/configs.txt", "configs.txt"), ("r05/parity.txt", "parity.txt"), ("r05/pytest.txt", "pytest.txt"), ("r05/cost_kernel_pmc.txt", "cost_kernel_pmc.txt"),
         ("r05/reference_start.txt", "reference_start.txt"), ("r05/window_bench.txt", "window_bench.txt"), ("r05/stream_bench.txt", "stream_bench.txt"),
         ("r05/setup.txt", "setup.txt"), ("r05/power_clock_trace.txt", "power_clock_trace.txt"),
         ("r05/stats_window/w_kernel_stats.csv", "window_bench_kernel_stats.csv"), ("r05/chain_profile.txt", "config3_chain_host_profile.txt"),
         # the sweeps the round's decisions were made on (earlier calls of the round)
         ("r05a/verdict_sweep.txt", "reference_start_sweep_1_second_attempts.txt"), ("r05b/verdict_sweep_blobs.txt", "reference_start_sweep_2_ragged_undamped.txt"),
         ("r05e/verdict_sweep.txt", "reference_start_sweep_3_damping_grid.txt"), ("r05e/verdict_sweep_blobs.txt", "reference_start_sweep_3_damping_ragged.txt"),
         ("r05e/verdict_sweep_768.txt", "reference_start_sweep_3_damping_768_slots.txt"), ("r05d/ab_onercp.txt", "kernel_experiments_one_rcp_ab.txt"), ("r05k/ab_occupancy.txt", "kernel_experiments_occupancy_ab.txt"),
         ("r05k/alone_probe.txt", "reference_start_g20y_pairs_alone.txt"), ("r05k/verdict_sweep_blobs.txt", "reference_start_sweep_4_damping_ragged_12288.txt"),
         ("r05v/reference_start_49152.txt", "reference_start_sweep_6_49152_starts.txt"), ("r05m/hard_ragged_probe.txt", "reference_start_hard_ragged_starts.txt"), ("r05m/hard_ragged_probe2.txt", "reference_start_hard_ragged_starts_2.txt")]
for a, b in pairs:
    src = os.path.join(G, a)
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, "r05_" + b))
        print("copied", a)
    else:
        print("MISSING", a)
