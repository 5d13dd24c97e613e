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
# single modules of packages whose other modules stay the reference's own (SAM / normals frontend: out of scope)
_REFERENCE_LEAF_MODULES = ("frontend.segment.post_processer",)


def install_as_reference_modules():
    """Alias ``super_primitive_amd.<pkg>[.<module>]`` as ``<pkg>[.<module>]`` in ``sys.modules``.  ``frontend.segment.post_processer``
    (the keyframe post-processing) is aliased as a LEAF: when the reference's ``frontend`` package is importable
    it keeps its other modules (``sam_tools``, ``mask_generation``, the normals code) and only ``post_processer`` is replaced; otherwise
    the (otherwise empty) parent packages of this tree stand in."""
    import importlib.util
    import pkgutil
    for pkg in _REFERENCE_PACKAGES:
        mod = importlib.import_module(f"{__name__}.{pkg}")
        sys.modules[pkg] = mod
        for info in pkgutil.iter_modules(mod.__path__):
            sub = importlib.import_module(f"{__name__}.{pkg}.{info.name}")
            sys.modules[f"{pkg}.{info.name}"] = sub
    for leaf in _REFERENCE_LEAF_MODULES:
        ours = importlib.import_module(f"{__name__}.{leaf}")
        parent_name, _, name = leaf.rpartition(".")
        try:
            found = importlib.util.find_spec(parent_name) is not None
        except (ImportError, ValueError):
            found = False
        if found and not sys.modules.get(parent_name, None) is importlib.import_module(f"{__name__}.{parent_name}"):
            parent = importlib.import_module(parent_name)                     # the reference's own package
        else:
            parts = parent_name.split(".")
            for i in range(1, len(parts) + 1):
                sys.modules[".".join(parts[:i])] = importlib.import_module(f"{__name__}." + ".".join(parts[:i]))
            parent = sys.modules[parent_name]
        sys.modules[leaf] = ours
        setattr(parent, name, ours)
