#!/usr/bin/env python3
"""Captures the inputs of the two hash-grid scatter launches (ngp_grid_bwd_param) of one steady-state
training step and writes them to gpurun_out/scatter_capture.npz: normalised sample positions, the
per-(sample, level) "gradient is non-zero" mask, and rays_a.  tools/scatter_model.py replays the
kernel's flush logic on them on the CPU to count memory-side atomic requests per level (GPU only)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import ngp_amd  # noqa: F401
from ngp_amd import _lib
from ngp_amd.networks import NGP
from ngp_amd.synthetic import LegoProxy
from ngp_amd.trainer import NGPTrainer

dev = torch.device("cuda", 0)
torch.manual_seed(20220806)
steps = int(os.environ.get("CAP_STEPS", "600"))
model = NGP(scale=0.5).to(dev)
G = model.grid_size
model.register_buffer("density_grid", torch.zeros(model.cascades, G ** 3, device=dev))
coords = torch.stack(torch.meshgrid(*[torch.arange(G, dtype=torch.int32, device=dev)] * 3, indexing="ij"), -1)
model.register_buffer("grid_coords", coords.reshape(-1, 3).contiguous())
scene = LegoProxy(n_images=100, img_wh=(800, 800), device=dev, seed=20220806)
tr = NGPTrainer(model)
gen = torch.Generator(device=dev).manual_seed(20220806)

captured = {}
orig_call = _lib.call


def spy(name, *args):
    if name == "grid_bwd_param" and captured.get("on"):
        desc, x, dy, lddy, n, buf = args
        L, F = desc.n_levels, desc.n_features
        nz = (dy[:, :L * F].reshape(n, L, F) != 0).any(-1)
        captured.setdefault("launches", []).append((x.clone(), nz.clone(), int(desc.offsets[L])))
    return orig_call(name, *args)


for mod in ("tinycudann", "networks"):
    setattr(sys.modules[f"ngp_amd.{mod}"], "call", spy)

for i in range(steps):
    img, pix = scene.sample_batch(8192, generator=gen)
    o, d = scene.rays(img, pix)
    gt, _ = scene.ground_truth(o, d, n_quad=256)
    captured["on"] = i == steps - 1
    loss, res = tr.step(o, d, gt)
torch.cuda.synchronize()
print("samples/ray", int(res["total_samples"]) / 8192, "loss", float(loss))
out = {"rays_a": res["rays_a"].cpu().numpy()}
for k, (x, nz, rows) in enumerate(captured["launches"]):
    out[f"x{k}"] = x.cpu().numpy()
    out[f"nz{k}"] = np.packbits(nz.cpu().numpy(), axis=1)
    out[f"rows{k}"] = np.int64(rows)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "scatter_capture.npz"), **out)
print("wrote gpurun_out/scatter_capture.npz", {k: np.shape(v) for k, v in out.items()})
