#!/usr/bin/env python3
"""How many padded sample slots does the test-time loop allocate against the valid ones? (GPU only)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ngp_amd
from ngp_amd import vren
from ngp_amd.networks import NGP
from ngp_amd.synthetic import LegoProxy
from ngp_amd.trainer import NGPTrainer
from ngp_amd.rendering import intersect_scene, MAX_SAMPLES

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
for i in range(600):
    img, pix = scene.sample_batch(8192, generator=gen)
    o, d = scene.rays(img, pix)
    gt, _ = scene.ground_truth(o, d, n_quad=128)
    tr.step(o, d, gt)
tr.wait()
n = 640000
o, d = scene.rays(torch.zeros(n, dtype=torch.long, device=dev), torch.arange(n, device=dev))
with torch.no_grad():
    hits_t = intersect_scene(model, o.contiguous(), d.contiguous())[:, 0, :].contiguous()
    opacity = torch.zeros(n, device=dev); depth = torch.zeros(n, device=dev); rgb = torch.zeros(n, 3, device=dev)
    nrm = torch.zeros(n, 3, device=dev); nrm2 = torch.zeros(n, 3, device=dev); sem = torch.zeros(n, 7, device=dev)
    alive = torch.arange(n, device=dev)
    samples = 0; it = 0; slots = 0; valid = 0
    while samples < MAX_SAMPLES:
        na = len(alive)
        if na == 0:
            break
        ns = max(min(n // na, 64), 1)
        samples += ns
        xyzs, dirs, deltas, ts, neff = vren.raymarching_test(o, d, hits_t, alive, model.density_bitfield, model.cascades,
                                                            model.scale, 0.0, model.grid_size, MAX_SAMPLES, ns)
        v = int(neff.sum()); slots += na * ns; valid += v; it += 1
        if v == 0:
            break
        x = xyzs.reshape(-1, 3); dd = dirs.reshape(-1, 3)
        s, r, npd, nr, sm = model.forward_test(x, dd)
        vren.composite_test_fw(s.view(na, ns).contiguous(), r.view(na, ns, 3).contiguous(), npd.view(na, ns, 3).contiguous(),
                               nr.view(na, ns, 3).contiguous(), sm.view(na, ns, 7).contiguous(), deltas, ts, hits_t, alive,
                               1e-2, 7, neff, opacity, depth, rgb, nrm, nrm2, sem)
        alive = alive[alive >= 0]
        if it % 5 == 0 or na < 2000:
            print(f"iter {it}: alive {na} n_samples {ns} slots {na*ns} valid {v}")
print(f"iterations {it}, padded slots {slots}, valid samples {valid}, waste {slots/valid:.2f}x")
