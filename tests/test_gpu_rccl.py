"""-m gpu: RCCL initialised on hardware with the one GPU there is: the N > 1 launch path of ``bench.py`` --
``python -m torch.distributed.run`` -> ``init_process_group("nccl")`` -> barrier / all_reduce(MAX) / all_gather -- and the package's two
collectives (``dist.gather_results``, ``dist.complete_depth_sharded``) run through the ``nccl`` backend with world size 1, so that the
driver's 8-GPU run is not the first time any of it executes.  (The world-2 control flow is covered on CPU by tests/test_dist_gloo.py.)"""
import json
import os
import re
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torchrun(script_args, timeout=600, **extra_env):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + script_args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines, r.stdout[-2000:] + r.stderr[-2000:]
    return json.loads(lines[-1])


def test_bench_under_torch_distributed_run_goes_through_rccl():
    line = _torchrun(["bench.py", "--gpus", "1", "--steps", "5", "--warmup", "2", "--pairs", "64", "--settle-ms", "20", "--no-extras",
                      "--no-cpu-baseline", "--no-pmc"])
    print("\nbench.py under torch.distributed.run, 1 rank:", {k: line[k] for k in ("n_gpus", "rccl_world", "ranks", "value")})
    assert line["n_gpus"] == 1 and line["rccl_world"] == 1 and len(line["ranks"]) == 1
    r0 = line["ranks"][0]
    assert r0["rank"] == 0 and r0["device_index"] == 0 and r0["pairs"] == 64 and r0["kernel_ms"] > 0
    assert re.fullmatch(r"[0-9a-f]{4}:[0-9a-f]{2}:[0-9a-f]{2}", r0["pci_bus_id"]), r0
    assert line["value"] > 0 and line["scaling"] == "weak"


def test_package_collectives_run_under_the_nccl_backend():
    rec = _torchrun([os.path.join("tools", "rccl_check.py")])
    print("\ntools/rccl_check.py under torch.distributed.run, 1 rank:", rec)
    assert rec["backend"] == "nccl" and rec["world"] == 1 and rec["all_reduce_of_ones"] == 1.0
    assert rec["gather_results_ok"] and rec["complete_depth_sharded_equals_single_process"]
    assert rec["gathered_shapes"] == [[6, 4, 4], [6, 7]]


def test_the_legs_of_an_n_gpu_bench_line_run_on_the_one_gpu_there_is():
    """what ``bench.py --gpus N`` does AFTER its timed steps when N > 1 -- the near-start schedule and the
    reference-start leg with slot-level continuous batching on every rank at once, timed between barriers with a MAX all_reduce over the
    group -- rehearsed under torch.distributed.run with the one rank there is (``SP_BENCH_REHEARSE_MULTI=1``): hooks, collectives and the
    verdict counts of the line have executed on hardware before an 8-GPU node sees them; the one-GPU side measurements are skipped."""
    line = _torchrun(["bench.py", "--gpus", "1", "--steps", "5", "--warmup", "2", "--pairs", "32", "--settle-ms", "20", "--sigma05-scenes", "2",
                      "--no-cpu-baseline", "--no-pmc"], SP_BENCH_REHEARSE_MULTI="1")
    print("\nN > 1 legs, 1 rank:", {k: line.get(k) for k in ("frame_pairs_per_sec", "frame_pairs_per_sec_near_start", "frame_pairs_status")})
    assert line["rccl_world"] == 1 and line["frame_pairs_per_sec"] > 0 and line["frame_pairs_per_sec_near_start"] > 0
    st = line["frame_pairs_status"]
    assert st["pairs"] == 128 and st["silent_failures"] == 0 and st["converged_first_attempt"] + st["converged_second_attempt"] + st["flagged_failed"] == 128
    assert "single_pair_gn_iters_per_sec" not in line and "from_raw_frames" not in line          # (the one-GPU side measurements)
    assert "slot_level_continuous_batching" in line["reference_start"]
