#!/usr/bin/env python3
"""Trains the lego-proxy scene with the reference's schedule and logs throughput / PSNR (GPU only)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ngp_amd
from ngp_amd.networks import NGP
from ngp_amd.synthetic import LegoProxy
from ngp_amd.trainer import NGPTrainer
from ngp_amd.metrics import psnr
from ngp_amd.rendering import render

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=2000)
ap.add_argument("--rays", type=int, default=8192)
ap.add_argument("--log-every", type=int, default=100)
ap.add_argument("--lr", type=float, default=1e-2)
ap.add_argument("--analytic-init", action="store_true")
ap.add_argument("--eval-rays", type=int, default=65536)
args = ap.parse_args()

dev = torch.device("cuda", 0)
torch.manual_seed(20220806)
model = NGP(scale=0.5).to(dev)
G = model.grid_size
model.register_buffer("density_grid", torch.zeros(model.cascades, G ** 3, device=dev))
coords = torch.stack(torch.meshgrid(*[torch.arange(G, dtype=torch.int32, device=dev)] * 3, indexing="ij"), -1)
model.register_buffer("grid_coords", coords.reshape(-1, 3).contiguous())
scene = LegoProxy(device=dev)
tr = NGPTrainer(model, lr=args.lr)
if args.analytic_init:
    model.density_grid.copy_(scene.occupancy_from_analytic(model))
    ngp_amd.vren.packbits(model.density_grid.view(-1), 0.5, model.density_bitfield)
    tr.global_step, tr.warmup_steps = 1024, 0
gen = torch.Generator(device=dev).manual_seed(1)
egen = torch.Generator(device=dev).manual_seed(99)
eimg, epix = scene.sample_batch(args.eval_rays, generator=egen)
eo, ed = scene.rays(eimg, epix)
egt, _ = scene.ground_truth(eo, ed, n_quad=512)

t_last = time.perf_counter()
samples = 0
for i in range(args.steps):
    img, pix = scene.sample_batch(args.rays, generator=gen)
    o, d = scene.rays(img, pix)
    gt, _ = scene.ground_truth(o, d, n_quad=256)
    loss, res = tr.step(o, d, gt)
    samples += int(res["total_samples"])
    if (i + 1) % args.log_every == 0:
        torch.cuda.synchronize()
        now = time.perf_counter()
        dt = now - t_last
        with torch.no_grad():
            ev = render(model, eo, ed, test_time=True, exp_step_factor=0.0, T_threshold=1e-4)
            ep = float(psnr(ev["rgb"], egt))
        occ = float((model.density_bitfield != 0).float().mean())
        print(json.dumps({"step": i + 1, "ms_per_step": dt / args.log_every * 1e3,
                          "rays_per_s": args.rays * args.log_every / dt, "samples_per_ray": samples / args.log_every / args.rays,
                          "train_psnr": float(psnr(res["rgb"].detach(), gt)), "eval_psnr": ep, "loss": float(loss),
                          "occupied_bytes_frac": occ}), flush=True)
        samples = 0
        torch.cuda.synchronize()
        t_last = time.perf_counter()
