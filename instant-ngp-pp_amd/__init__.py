"""MI355X-native instant-NGP hot path behind the reference's operator surface.

Sub-modules (same names / call signatures as zhihao-lin/instant-ngp-pp):
  vren              — the 15 native entry points of models/csrc/binding.cpp:323-342
  tinycudann        — Encoding / Network / NetworkWithInputEncoding (models/networks.py:5)
  torch_scatter     — segment_csr (models/custom_functions.py:4)
  custom_functions  — RayAABBIntersector, RayMarcher, VolumeRenderer, RefLoss, ...
  rendering         — render(model, rays_o, rays_d, **kwargs)
  networks          — NGP
  losses            — NeRFLoss, DistortionLoss
  trainer           — the training schedule of train.py (no Lightning)

Every compute call goes through libngp_hip.so (include/ngp_hip.h); there is no CPU fallback.
"""
from . import _lib  # noqa: F401  (parses the header; the .so is loaded on first use)

__all__ = ["vren", "tinycudann", "torch_scatter", "custom_functions", "rendering", "networks", "losses",
           "metrics", "trainer", "synthetic", "ckpt", "install_as_reference_modules"]


def __getattr__(name):
    if name in __all__ and name != "install_as_reference_modules":
        import importlib
        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)


def install_as_reference_modules():
    """Registers this package's modules under the names the reference imports
    (`import vren`, `import tinycudann as tcnn`, `from torch_scatter import segment_csr`),
    so an unmodified reference train.py / render.py picks up the MI355X path."""
    import importlib
    import sys
    for ref_name in ("vren", "tinycudann", "torch_scatter"):
        sys.modules[ref_name] = importlib.import_module(f"{__name__}.{ref_name}")
