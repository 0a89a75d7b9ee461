#!/usr/bin/env python3
"""A/B of the hash-grid gathers on the launches of a real training step (GPU only, A/B build):
the ray-coherent tile kernels (product) against the item-per-(sample, level) kernels they replaced.

    NGP_AB_VARIANTS=1 python -m instant-ngp-pp_amd.build && NGP_AB_VARIANTS=1 python tools/gather_replay.py

Every captured `grid_fwd` / `grid_bwd_input` call is replayed with both kernels (NGP_GRID_GATHER_OLD is looked up per
call in the A/B build), outputs compared, and timed two ways: WARM (20 repetitions back to back: the 174 MB density
table stays in the 256 MiB Infinity Cache) and COLD (a 1 GiB buffer is rewritten between repetitions, so every
repetition starts with the table in HBM, as it does in the step behind the Adam sweep).
GR_ONLY=old|run|tile (comma list) restricts the run to those kernels (for rocprofv3 --pmc passes)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ngp_amd
from ngp_amd import _lib
from ngp_amd.networks import NGP
from ngp_amd.synthetic import LegoProxy
from ngp_amd.trainer import NGPTrainer

assert os.environ.get("NGP_AB_VARIANTS"), "needs the A/B build (NGP_AB_VARIANTS=1)"
dev = torch.device("cuda", 0)
torch.manual_seed(20220806)
steps = int(os.environ.get("GR_STEPS", "300"))
model = NGP(scale=0.5).to(dev)
G = model.grid_size
model.register_buffer("density_grid", torch.zeros(model.cascades, G ** 3, device=dev))
coords = torch.stack(torch.meshgrid(*[torch.arange(G, dtype=torch.int32, device=dev)] * 3, indexing="ij"), -1)
model.register_buffer("grid_coords", coords.reshape(-1, 3).contiguous())
scene = LegoProxy(device=dev)
tr = NGPTrainer(model)
gen = torch.Generator(device=dev).manual_seed(1)

captured = {}
orig_call = _lib.call


def keep(a):
    if not isinstance(a, torch.Tensor) or a.numel() >= 2e8:
        return a
    if a.dim() == 2 and not a.is_contiguous() and a.stride(1) == 1:
        b = torch.zeros(a.shape[0], a.stride(0), dtype=a.dtype, device=a.device)
        b[:, :a.shape[1]] = a
        return b[:, :a.shape[1]]
    return a.clone()


def spy(name, *args):
    if name in ("grid_fwd", "grid_bwd_input") and captured.get("on"):
        captured.setdefault(name, []).append(tuple(keep(a) for a in args))
    return orig_call(name, *args)


for mod in ("tinycudann", "networks"):
    setattr(sys.modules[f"ngp_amd.{mod}"], "call", spy)

os.environ["NGP_GRID_GATHER_OLD"] = "2"    # the training steps run on the third variant: PMC summaries of `old` and `run`
for i in range(steps):                      # then hold the replays only
    img, pix = scene.sample_batch(8192, generator=gen)
    o, d = scene.rays(img, pix)
    gt, _ = scene.ground_truth(o, d, n_quad=256)
    captured["on"] = i == steps - 1 and (i % 16) != 0
    loss, res = tr.step(o, d, gt)
torch.cuda.synchronize()
print("samples/ray", int(res["total_samples"]) / 8192, "loss", float(loss), flush=True)

flush_buf = torch.empty(1 << 28, dtype=torch.float32, device=dev)   # 1 GiB


def select(kind):   # old: item-per-(sample, level); run: run leaders + bpermute (product); tile: staged unique cells
    os.environ["NGP_GRID_GATHER_OLD"] = {"old": "1", "tile": "2", "run": "0"}[kind]


def warm(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def cold(fn, reps=6):
    ts = []
    for _ in range(reps):
        flush_buf.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


kinds = os.environ["GR_ONLY"].split(",") if os.environ.get("GR_ONLY") else ["old", "run", "tile"]
for name in ("grid_fwd", "grid_bwd_input"):
    for idx, args in enumerate(captured.get(name, [])):
        ints = [a for a in args if isinstance(a, int)]
        n = ints[0] if name == "grid_fwd" else ints[1]
        out_pos = 4 if name == "grid_fwd" else 6
        outs = {}
        line = f"{name}[{idx}] rows={args[0].offsets[args[0].n_levels]} n={n}:"
        for kind in kinds:
            select(kind)
            a = list(args)
            a[out_pos] = torch.zeros_like(args[out_pos]) if args[out_pos].is_contiguous() else args[out_pos]
            if not args[out_pos].is_contiguous():
                a[out_pos].zero_()
            orig_call(name, *a)
            torch.cuda.synchronize()
            outs[kind] = a[out_pos].clone()
            w = warm(lambda: orig_call(name, *a))
            c = cold(lambda: orig_call(name, *a))
            line += f"  {kind}: warm {w:.3f} ms ({n * 4608 / w / 1e6:.0f} GB/s alg)  cold {c:.3f} ms ({n * 4608 / c / 1e6:.0f} GB/s alg)"
        for k in kinds[1:]:
            d = (outs[kinds[0]] - outs[k]).abs().max().item()
            ref = outs[kinds[0]].abs().max().item()
            line += f"  | max|{kinds[0]}-{k}| {d:.3e} of {ref:.3e}" + ("  BIT-IDENTICAL" if torch.equal(outs[kinds[0]], outs[k]) else "")
        print(line, flush=True)
