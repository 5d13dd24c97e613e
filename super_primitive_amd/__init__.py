"""super_primitive_amd -- MI355X (gfx950) implementation of SuperPrimitive's per-segment photometric
pose-and-depth optimisation hot path, behind the reference's own Python function API.

Sub-packages mirror the reference's module paths (``core.dense_optim``, ``core.dense_optim_batch``, ``core.ops``,
``core.depth_render``, ``image.keyframe``, ``image.gaussian_pyramid``, ``lie.lie_algebra``, ``lie.lietorch_utils``,
``tool.point_utils``, ``odometery.depth_init``, ...).  ``install_as_reference_modules()`` additionally registers
them under the reference's top-level names so an unmodified reference driver (``import core.dense_optim as
dense_optim``) picks up the HIP path.  See DESIGN.md / INTEGRATION.md.
"""
import importlib
import sys

__version__ = "0.1.0"

_REFERENCE_PACKAGES = ("core", "image", "lie", "tool", "odometery", "depth_completion")


def install_as_reference_modules():
    """Alias ``super_primitive_amd.<pkg>[.<module>]`` as ``<pkg>[.<module>]`` in ``sys.modules``."""
    import pkgutil
    for pkg in _REFERENCE_PACKAGES:
        mod = importlib.import_module(f"{__name__}.{pkg}")
        sys.modules[pkg] = mod
        for info in pkgutil.iter_modules(mod.__path__):
            sub = importlib.import_module(f"{__name__}.{pkg}.{info.name}")
            sys.modules[f"{pkg}.{info.name}"] = sub
