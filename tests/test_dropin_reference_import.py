"""Drop-in check in the container that has the reference checkout (skipped elsewhere, e.g. on the
GPU box): with this package registered as `vren` / `tinycudann` / `torch_scatter`
(ngp_amd.install_as_reference_modules), the reference's OWN model and operator modules import, its
NGP class constructs on this tinycudann surface, and the resulting parameters / buffers have exactly
the names and shapes of this package's NGP — so reference checkpoints and training scripts see the
layout they expect.  Nothing is computed (no GPU here); the reference is only imported."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

SCRIPT = r'''
import sys
sys.path.insert(0, %(root)r)
import ngp_amd
ngp_amd.install_as_reference_modules()
sys.path.insert(0, %(ref)r)
import models.custom_functions as ref_cf      # imports vren, torch_scatter
import models.rendering as ref_render         # imports vren and the operator classes
import models.networks as ref_net             # imports tinycudann, vren
from ngp_amd.networks import NGP
for kw in ({"scale": 0.5}, {"scale": 8.0, "embed_a": True, "embed_a_len": 8}, {"scale": 0.5, "use_skybox": True}):
    theirs, ours = ref_net.NGP(**kw), NGP(**kw)
    a = {k: tuple(v.shape) for k, v in theirs.state_dict().items()}
    b = {k: tuple(v.shape) for k, v in ours.state_dict().items()}
    assert a == b, (kw, sorted(set(a.items()) ^ set(b.items())))
    assert theirs.cascades == ours.cascades and theirs.grid_size == ours.grid_size
    assert theirs.rgb_net.n_input_dims == ours.rgb_net.n_input_dims
# the operator classes the reference's render() uses exist with the same names here
import ngp_amd.custom_functions as cf
for name in ("RayAABBIntersector", "RaySphereIntersector", "RayMarcher", "VolumeRenderer", "RefLoss", "TruncExp",
             "TruncTanh"):
    assert hasattr(ref_cf, name) and hasattr(cf, name), name
import vren
for fn in ("ray_aabb_intersect", "ray_sphere_intersect", "morton3D", "morton3D_invert", "packbits", "raymarching_train",
           "raymarching_test", "composite_alpha_fw", "composite_train_fw", "composite_train_bw", "composite_test_fw",
           "composite_refloss_fw", "composite_refloss_bw", "distortion_loss_fw", "distortion_loss_bw"):
    assert callable(getattr(vren, fn)), fn
# the reference's loss module imports on this vren and carries the same weights as ours
import importlib.util
spec = importlib.util.spec_from_file_location("ref_losses", %(ref)r + "/losses.py")
ref_losses = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_losses)
from ngp_amd.losses import NeRFLoss
rl, ol = ref_losses.NeRFLoss(), NeRFLoss()
for k in ("lambda_opa", "lambda_distortion", "lambda_depth_mono", "lambda_normal_mono", "lambda_normal_ref_rp",
          "lambda_normal_ref_ro", "lambda_sky", "lambda_semantic"):
    assert getattr(rl, k) == getattr(ol, k), k
print("DROPIN_OK")
'''


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "models")), reason="reference checkout not present")
def test_reference_modules_import_and_build_on_this_package():
    out = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT, "ref": REF}], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0 and "DROPIN_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
