#!/usr/bin/env python3
"""How far below the clip threshold (50) is the gradient norm, and its cheap upper bounds, during training?
norm ||g|| <= sum over samples of ||dL_dy[s]||_2 (triangle inequality; trilinear weights have sum of squares <= 1)
         <= sum of |dL_dy| entries.  Prints the three quantities per encoder for a few steps (GPU only)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ngp_amd  # noqa: F401
from ngp_amd import _lib
from ngp_amd.networks import NGP
from ngp_amd.synthetic import LegoProxy
from ngp_amd.trainer import NGPTrainer

dev = torch.device("cuda", 0)
torch.manual_seed(20220806)
model = NGP(scale=0.5).to(dev)
G = model.grid_size
model.register_buffer("density_grid", torch.zeros(model.cascades, G ** 3, device=dev))
coords = torch.stack(torch.meshgrid(*[torch.arange(G, dtype=torch.int32, device=dev)] * 3, indexing="ij"), -1)
model.register_buffer("grid_coords", coords.reshape(-1, 3).contiguous())
scene = LegoProxy(n_images=100, img_wh=(800, 800), device=dev, seed=20220806)
tr = NGPTrainer(model)
gen = torch.Generator(device=dev).manual_seed(20220806)
probe = {}
orig_call = _lib.call


def spy(name, *args):
    if name in ("grid_bwd_param", "grid_bwd_param_scaled") and probe.get("on"):
        if name == "grid_bwd_param_scaled":
            desc, x, dy, lddy, rs, n, buf = args
            rows = dy[:, :128] * rs[:, None]
        else:
            desc, x, dy, lddy, n, buf = args
            rows = dy[:, :128]
        probe.setdefault("b", []).append((float(rows.norm(dim=1).sum()), float(rows.abs().sum())))
    if name == "clip_coef" and probe.get("on"):
        torch.cuda.synchronize()
        probe["norm"] = float(args[0].sqrt())
    return orig_call(name, *args)


for mod in ("tinycudann", "networks", "trainer"):
    setattr(sys.modules[f"ngp_amd.{mod}"], "call", spy)
for i in range(1200):
    img, pix = scene.sample_batch(8192, generator=gen)
    o, d = scene.rays(img, pix)
    gt, _ = scene.ground_truth(o, d, n_quad=256)
    probe.clear()
    probe["on"] = i in (0, 1, 2, 5, 10, 20, 50, 100, 200, 255, 256, 300, 400, 600, 800, 1000, 1199)
    loss, res = tr.step(o, d, gt)
    if probe.get("on"):
        tr.wait()
        torch.cuda.synchronize()
        print(f"step {i:5d} loss {float(loss):.2e} samples {int(res['total_samples']):8d}  ||g|| {probe.get('norm', float('nan')):.3e}   "
              + "   ".join(f"sum_s||row||2 {a:.3e}  sum|entries| {b:.3e}" for a, b in probe.get("b", [])))
