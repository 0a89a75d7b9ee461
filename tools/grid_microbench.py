#!/usr/bin/env python3
"""Times the hash-grid kernels on tensors captured from a real training step (GPU only).
The superseded scatter variants (MB_VARIANTS=slide,pair,merge,simple) exist only in the A/B build:
    NGP_AB_VARIANTS=1 python -m instant-ngp-pp_amd.build && NGP_AB_VARIANTS=1 python tools/grid_microbench.py
The library reads its variant switches once per process: run one variant per invocation (MB_VARIANTS=slide ...).
(`line` = the product kernel; tools/scatter_replay.py times it on a steady-state batch and summarises PMC passes)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ngp_amd
from ngp_amd import _lib
from ngp_amd.networks import NGP
from ngp_amd.synthetic import LegoProxy
from ngp_amd.trainer import NGPTrainer

dev = torch.device("cuda", 0)
torch.manual_seed(20220806)
steps = int(os.environ.get("MB_STEPS", "30"))
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
    """copy of a tensor argument WITH its row stride: the encoders write into / read from column windows of wider
    matrices (rgb_in[:, 16:], row stride 144), and the replay passes the captured leading dimension"""
    if not isinstance(a, torch.Tensor) or a.numel() >= 2e8:
        return a
    if a.dim() == 2 and not a.is_contiguous() and a.stride(1) == 1:
        b = torch.zeros(a.shape[0], a.stride(0), dtype=a.dtype, device=a.device)
        b[:, :a.shape[1]] = a
        return b[:, :a.shape[1]]
    return a.clone()


def spy(name, *args):
    if name in ("grid_bwd_param", "grid_fwd", "grid_bwd_input") and captured.get("on"):
        captured.setdefault(name, []).append(tuple(keep(a) for a in args))
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


def timeit(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for idx, args in enumerate(captured["grid_bwd_param"]):
    desc, x, dy, lddy, n, buf = args
    nz_rows = (dy.abs().sum(1) > 0).float().mean().item()
    nz_el = (dy != 0).float().mean().item()
    tbl = torch.zeros(desc.offsets[desc.n_levels] * desc.n_features, device=dev)
    for variant in os.environ.get("MB_VARIANTS", "line,slide,pair,merge,simple").split(","):
        os.environ.pop("NGP_GRID_BWD_SIMPLE", None)
        os.environ.pop("NGP_GRID_BWD_NOPAIR", None)
        os.environ.pop("NGP_GRID_BWD_NOSLIDE", None)
        os.environ.pop("NGP_GRID_BWD_NOLINE", None)
        if variant == "slide":
            os.environ["NGP_GRID_BWD_NOLINE"] = "1"
        if variant == "pair":
            os.environ["NGP_GRID_BWD_NOSLIDE"] = "1"
        if variant == "simple":
            os.environ["NGP_GRID_BWD_SIMPLE"] = "1"
        if variant == "merge":
            os.environ["NGP_GRID_BWD_NOPAIR"] = "1"
        ms = timeit(lambda: orig_call("grid_bwd_param", desc, x, dy, lddy, n, tbl))
        print(f"bwd_param[{idx}] table_rows={desc.offsets[desc.n_levels]} n={n} nonzero_rows={nz_rows:.3f} nonzero_el={nz_el:.3f} "
              f"{variant}: {ms:.3f} ms  alg {n*4608/ms/1e6:.0f} GB/s  (nonzero-only {n*nz_rows*4608/ms/1e6:.0f} GB/s)")
    os.environ.pop("NGP_GRID_BWD_SIMPLE", None)
    os.environ.pop("NGP_GRID_BWD_NOPAIR", None)
    os.environ.pop("NGP_GRID_BWD_NOSLIDE", None)
    # all-nonzero gradient for the same positions: the kernel's ceiling without sparsity
    dyr = torch.randn_like(dy)
    ms = timeit(lambda: orig_call("grid_bwd_param", desc, x, dyr, lddy, n, tbl))
    print(f"   dense random dy: {ms:.3f} ms  alg {n*4608/ms/1e6:.0f} GB/s")
for name in ("grid_fwd", "grid_bwd_input"):
    for idx, args in enumerate(captured.get(name, [])):
        n = [a for a in args if isinstance(a, int)][0 if name == "grid_fwd" else 1]
        ms = timeit(lambda: orig_call(name, *args))
        print(f"{name}[{idx}] n={n}: {ms:.3f} ms  alg {n*4608/ms/1e6:.0f} GB/s")
