#!/usr/bin/env python3
"""In-process A/B of one trainer switch on the bench workload (GPU only): one model, one trainer, alternating windows of
training steps with the switch off / on (box-to-box and run-to-run spread of bench.py is +-3 %, more than most single
changes are worth; alternating inside one process removes it).

    python tools/ab_step.py fused_tail            # NGPTrainer attribute toggled False / True
    python tools/ab_step.py env:NGP_X             # os.environ switch toggled unset / "1" (only for switches read per call)
    AB_WINDOWS=12 AB_STEPS=40 python tools/ab_step.py fused_tail"""
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

what = sys.argv[1]
windows = int(os.environ.get("AB_WINDOWS", "10"))
steps = int(os.environ.get("AB_STEPS", "48"))
dev = torch.device("cuda", 0)
torch.manual_seed(20220806)
model = NGP(scale=0.5).to(dev)
G = model.grid_size
model.register_buffer("density_grid", torch.zeros(model.cascades, G ** 3, device=dev))
coords = torch.stack(torch.meshgrid(*[torch.arange(G, dtype=torch.int32, device=dev)] * 3, indexing="ij"), -1)
model.register_buffer("grid_coords", coords.reshape(-1, 3).contiguous())
model.grid_rng = torch.Generator(device=dev).manual_seed(20220806)
scene = LegoProxy(device=dev)
tr = NGPTrainer(model)
gen = torch.Generator(device=dev).manual_seed(1)


def batch():
    img, pix = scene.sample_batch(8192, generator=gen)
    o, d = scene.rays(img, pix)
    gt, _ = scene.ground_truth(o, d, n_quad=256)
    return o, d, gt


batches = [batch() for _ in range(64)]
for _ in range(int(os.environ.get("AB_PRETRAIN", "600"))):
    tr.step(*batch())


def set_switch(on):
    if what.startswith("env:"):
        if on:
            os.environ[what[4:]] = "1"
        else:
            os.environ.pop(what[4:], None)
    else:
        setattr(tr, what, on)


def window(i0):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        o, d, gt = batches[(i0 + i) % 64]
        nxt = batches[(i0 + i + 1) % 64][:2] if i + 1 < steps else None
        tr.step(o, d, gt, next_rays=nxt)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


import gc
gc.collect()
gc.freeze()
window(0)
res = {False: [], True: []}
pos = 0
for w in range(windows):
    for on in ((False, True) if w % 2 == 0 else (True, False)):
        set_switch(on)
        res[on].append(window(pos))
        pos += steps
for on in (False, True):
    v = sorted(res[on])
    print(f"{what} {'on ' if on else 'off'}: median {v[len(v) // 2]:.3f} ms/step  mean {sum(v) / len(v):.3f}  min {v[0]:.3f}  max {v[-1]:.3f}  "
          + " ".join(f"{x:.3f}" for x in res[on]))
