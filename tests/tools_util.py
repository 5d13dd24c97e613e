"""Helpers shared by the -m gpu tests that rebuild bench.py's workloads (picklable: used from multiprocessing pools)."""


def render_reference_start_scene(seed, H=480, W=640, N=64):
    """One scene of bench.py's reference-start leg (bench.py:_render_sigma05)."""
    from super_primitive_amd import synth
    return synth.make_pair(H, W, N, seed=seed, overlap=4, init_sigma=0.05, texture="octaves", init_mode="reference")
