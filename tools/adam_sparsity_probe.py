#!/usr/bin/env python3
"""How much of the optimizer state is still untouched (g = m = v = 0 for a whole 16-byte quad / a whole 1 KiB wave piece) after N
training steps on the proxy scene?  Such quads stay exactly as they are under torch.optim.Adam without weight decay."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ngp_amd
from ngp_amd.networks import NGP
from ngp_amd.synthetic import LegoProxy
from ngp_amd.trainer import NGPTrainer

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, nargs="+", default=[100, 600, 2000, 6000])
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(20220806)
model = NGP(scale=0.5).to(dev)
G = model.grid_size
model.register_buffer("density_grid", torch.zeros(model.cascades, G ** 3, device=dev))
coords = torch.stack(torch.meshgrid(*[torch.arange(G, dtype=torch.int32, device=dev)] * 3, indexing="ij"), -1)
model.register_buffer("grid_coords", coords.reshape(-1, 3).contiguous())
scene = LegoProxy(img_wh=(800, 800), device=dev)
tr = NGPTrainer(model, lr=1e-2)
gen = torch.Generator(device=dev).manual_seed(1)
done = 0
for target in args.steps:
    while done < target:
        img, pix = scene.sample_batch(8192, generator=gen)
        o, d = scene.rays(img, pix)
        gt, _ = scene.ground_truth(o, d, n_quad=128)
        tr.step(o, d, gt)
        done += 1
    tr.wait()
    torch.cuda.synchronize()
    v = tr.exp_avg_sq
    n = v.numel() // 256 * 256
    nzq = (v[:n].view(-1, 4) != 0).any(1)
    nzw = nzq.view(-1, 64).any(1)
    print(f"after {done} steps: {v.numel() / 1e6:.1f} M parameters; quads with v != 0: {nzq.float().mean().item():.3f}; "
          f"1 KiB pieces with any v != 0: {nzw.float().mean().item():.3f}", flush=True)
    for name, p in model.named_parameters():
        if p.numel() > 1e6:
            off = (p.data_ptr() - tr.flat_param.data_ptr()) // 4
            vv = v[off:off + p.numel()].view(-1, 4)
            q = (vv != 0).any(1)
            per = q.view(16, -1).float().mean(1) if q.numel() % 16 == 0 else None
            print(f"  {name}: {q.float().mean().item():.3f}", "" if per is None else
                  "by sixteenth: " + " ".join(f"{x:.2f}" for x in per.tolist()), flush=True)
