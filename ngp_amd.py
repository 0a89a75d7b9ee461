"""Import alias: the product package lives in `instant-ngp-pp_amd/` (a directory name Python
cannot spell in an `import` statement), so `import ngp_amd` loads it through importlib and
registers the package and every sub-module under the `ngp_amd.*` names as well (the SAME module
objects, so there is exactly one copy of each module and of the loaded libngp_hip.so)."""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

_REAL = "instant-ngp-pp_amd"
_pkg = importlib.import_module(_REAL)
sys.modules[__name__] = _pkg
for _sub in ("_lib", "build", "vren", "tinycudann", "torch_scatter", "custom_functions", "rendering", "networks",
             "losses", "metrics", "synthetic", "trainer", "ckpt", "datasets", "datasets.base", "datasets.ray_utils",
             "datasets.color_utils", "datasets.colmap_utils", "datasets.nerf", "datasets.colmap", "datasets.tnt",
             "datasets.nsvf", "datasets.nerfpp", "datasets.export"):
    sys.modules[f"{__name__}.{_sub}"] = importlib.import_module(f"{_REAL}.{_sub}")
