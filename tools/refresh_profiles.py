#!/usr/bin/env python
"""Copies the evidence tools/collect_profiles.sh left under gpurun_out/<round>/ into profiles/ (tracked) and
re-derives profiles/pmc_traffic.json (HBM bytes per launch of the dominant kernel from the PMC passes).

    python tools/refresh_profiles.py r01
"""
import collections, csv, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
R = os.path.join(ROOT, "gpurun_out", rnd)
P = os.path.join(ROOT, "profiles")
KERN = "k_cost_pairs<1, 0, 0>"


def per_dispatch(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    with open(path) as f:
        for row in csv.DictReader(f):
            if KERN in row["Kernel_Name"]:
                agg[row["Dispatch_Id"]][row["Counter_Name"]] += float(row["Counter_Value"])
    return agg


def mean(agg, name):
    v = [d[name] for d in agg.values()]
    return sum(v) / len(v)


fetch = mean(per_dispatch(f"{R}/pmc_FETCH_SIZE/bench_counter_collection.csv"), "FETCH_SIZE")
write = mean(per_dispatch(f"{R}/pmc_WRITE_SIZE/bench_counter_collection.csv"), "WRITE_SIZE")
rdreq = mean(per_dispatch(f"{R}/pmc_TCC_EA0_RDREQ_sum/bench_counter_collection.csv"), "TCC_EA0_RDREQ_sum")
bench = json.load(open(f"{R}/bench_n1.json"))
alg = bench["roofline"]["algorithmic_bytes_per_launch"]
hbm = (2 * fetch + write) * 1024
json.dump({
    "round": int(rnd[1:]),
    "command": "python bench.py --settle-ms 0 --steps 10 --warmup 2 --no-cpu-baseline --no-extras (one rocprofv3 --pmc pass per "
               "counter group, --kernel-trace only; tools/collect_profiles.sh)",
    "kernel": "k_cost_pairs<1,0,0>", "pairs_per_gpu": bench["config"]["pairs_per_gpu"], "mode": "gn",
    "tile_points": bench["config"]["tile_points"],
    "FETCH_SIZE_KB_mean": fetch, "WRITE_SIZE_KB_mean": write,
    "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B while wide coalesced reads are "
                  "128-B requests -> read bytes = 2 x FETCH_SIZE; KB -> x1024. Cross-check: TCC_EA0_RDREQ_sum x 128 B = "
                  f"{rdreq * 128:.0f} B; expected from the layout (20 B/pt source stream + 12 B/px HWC3 target) = {alg} B.",
    "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "ratio": hbm / alg,
}, open(f"{P}/pmc_traffic.json", "w"), indent=1)

sq = per_dispatch(f"{R}/pmc_SQ_WAVES/bench_counter_collection.csv")
names = sorted({k for v in sq.values() for k in v})
with open(f"{P}/{rnd}_bench_n1_pmc_cost_kernel.csv", "w") as f:
    f.write(f"# per-dispatch PMC values of {KERN} (GN), one rocprofv3 --pmc pass per counter group (tools/collect_profiles.sh)\n")
    f.write("dispatch," + ",".join(names) + "\n")
    for k, v in list(sq.items())[:12]:
        f.write(k + "," + ",".join(f"{v[n]:.0f}" for n in names) + "\n")
    f.write(f"# FETCH_SIZE_KB mean {fetch:.1f}  WRITE_SIZE_KB mean {write:.1f}  TCC_EA0_RDREQ_sum mean {rdreq:.0f}\n")
    f.write("# means: " + "  ".join(f"{n} {mean(sq, n):.0f}" for n in names) + "\n")

for a, b in [("bench_n1.json", "bench_n1.json"), ("bench_n1_adam.json", "bench_n1_adam.json"),
             ("bench_under_rocprof.json", "bench_n1_under_rocprof.json"), ("bench_long.json", "bench_n1_long_run.json"),
             ("stats/bench_kernel_stats.csv", "bench_n1_kernel_stats.csv"), ("kbench_ablation.txt", "kbench_ablation.txt"),
             ("configs.txt", "configs.txt"), ("bench_n1_seg128.json", "bench_n1_seg128.json"),
             ("stats128/bench_kernel_stats.csv", "bench_n1_seg128_kernel_stats.csv"), ("parity.txt", "parity.txt"),
             ("power_by_mode.txt", "power_by_mode.txt"), ("setup.txt", "setup.txt"),
             ("stats_setup/setup_kernel_stats.csv", "setup_kernel_stats.csv")]:
    if os.path.exists(f"{R}/{a}"):
        shutil.copy(f"{R}/{a}", f"{P}/{rnd}_{b}")
with open(f"{P}/{rnd}_power_clock_trace.txt", "w") as f:
    f.write("# sclk and socket package power (W), sampled once a second with rocm-smi across `python bench.py --steps 16000`\n"
            "# (16 s of GN steps, idle before and after)\n")
    f.write(open(f"{R}/power_clock_trace.txt").read())
print(f"hbm/alg = {hbm / alg:.4f}; " + "  ".join(f"{n} {mean(sq, n):.3g}" for n in names))
for f in ("bench_n1.json", "bench_n1_adam.json", "bench_under_rocprof.json", "bench_long.json"):
    d = json.load(open(f"{R}/{f}"))
    print(f, round(d["value"]), round(d["ms_per_step"], 4), round(d["roofline"]["kernel_ms"], 4), round(d["roofline"]["frac"], 4),
          d.get("single_pair_gn_iters_per_sec"), d.get("single_pair_gn_iters_per_sec_hipgraph"), d.get("frame_pairs_per_sec"))
