#!/usr/bin/env python3
"""Replays the two hash-grid scatter launches of one steady-state training step REPS times each
(after CAP_STEPS training steps), for timing and for rocprofv3 --pmc passes:

  rocprofv3 --pmc TCC_EA0_ATOMIC_sum WRITE_SIZE --output-format csv -d /tmp/p -- python3 tools/scatter_replay.py
  python3 tools/scatter_replay.py --summarise /tmp/p 2*REPS

The summary averages the counters over the LAST 2*REPS dispatches of the scatter kernel (the replay
phase), so the warm-up launches with their millions of samples do not enter the per-launch figures."""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def summarise(root, last):
    rows = []
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if "grid_bwd_param" in r["Kernel_Name"]]
    by = {}
    for r in rows:
        by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(by)[-last:]
    half = len(ids) // 2
    for tag, sel in (("launch A (first table replayed)", ids[:half]), ("launch B (second table replayed)", ids[half:])):
        keys = sorted({k for i in sel for k in by[i]})
        print(tag, {k: round(sum(by[i].get(k, 0.0) for i in sel) / max(len(sel), 1), 1) for k in keys}, "dispatches", len(sel))


if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    summarise(sys.argv[2], int(sys.argv[3]))
    raise SystemExit(0)

import torch
import ngp_amd  # noqa: F401
from ngp_amd import _lib
from ngp_amd.networks import NGP
from ngp_amd.synthetic import LegoProxy
from ngp_amd.trainer import NGPTrainer

dev = torch.device("cuda", 0)
torch.manual_seed(20220806)
steps = int(os.environ.get("CAP_STEPS", "400"))
reps = int(os.environ.get("REPS", "5"))
cap_file = os.environ.get("CAP_FILE")   # captured launches are kept here: later runs (other variants) only replay


def replay(launches):
    import ctypes
    for desc, x, dy, lddy, rs, n, buf in launches:
        if isinstance(desc, bytes):
            d = _lib.GridDesc()
            ctypes.memmove(ctypes.addressof(d), desc, len(desc))
            desc = d
        tbl = torch.zeros(desc.offsets[desc.n_levels] * desc.n_features, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        _lib.call("grid_bwd_param_scaled", desc, x, dy, lddy, rs, n, tbl)   # warm
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            _lib.call("grid_bwd_param_scaled", desc, x, dy, lddy, rs, n, tbl)
        e1.record()
        torch.cuda.synchronize()
        print(f"rows={desc.offsets[desc.n_levels]} n={n}: {e0.elapsed_time(e1) / reps:.3f} ms per launch, "
              f"table sum {float(tbl.double().sum()):.6e} abs {float(tbl.double().abs().sum()):.6e}")


if cap_file and os.path.exists(cap_file):
    launches = torch.load(cap_file, map_location=dev, weights_only=False)
    replay(launches)
    raise SystemExit(0)
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
    if name in ("grid_bwd_param", "grid_bwd_param_scaled") and captured.get("on"):
        if name == "grid_bwd_param_scaled":
            desc, x, dy, lddy, rs, n, buf = args
        else:
            (desc, x, dy, lddy, n, buf), rs = args, None
        cap = (desc, x, dy, lddy, rs, n, buf)
        captured.setdefault("launches", []).append(tuple(a.clone() if isinstance(a, torch.Tensor) and a.numel() < 2e8 else a
                                                         for a in cap))
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
if cap_file:
    import ctypes
    torch.save([(bytes(ctypes.string_at(ctypes.addressof(l[0]), ctypes.sizeof(l[0]))),) + tuple(l[1:]) for l in captured["launches"]],
               cap_file)
replay(captured["launches"])
