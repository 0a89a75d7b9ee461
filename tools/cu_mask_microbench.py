#!/usr/bin/env python3
"""Does partitioning the CUs between the atomic-bound colour scatter and the density head's MFMA products
(hipExtStreamCreateWithCUMask) buy real overlap?  Captures both from a training step, then times them
alone on k CUs and side by side on complementary masks.  GPU only."""
import ctypes
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

dev = torch.device("cuda", 0)
hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(bits):
    """bits: iterable of CU indices (0..255) to enable"""
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)


torch.manual_seed(20220806)
steps = int(os.environ.get("MB_STEPS", "320"))
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


def spy(name, *args):
    if captured.get("on") and name in ("grid_bwd_param", "mlp_bwd_weight", "mlp_bwd_input"):
        captured.setdefault(name, []).append(tuple(a.clone() if isinstance(a, torch.Tensor) and a.numel() < 2e8 else a for a in args))
    return orig_call(name, *args)


for mod in ("tinycudann", "networks"):
    setattr(sys.modules[f"ngp_amd.{mod}"], "call", spy)
for i in range(steps):
    img, pix = scene.sample_batch(8192, generator=gen)
    o, d = scene.rays(img, pix)
    gt, _ = scene.ground_truth(o, d, n_quad=256)
    captured["on"] = i == steps - 1
    loss, res = tr.step(o, d, gt)
tr.wait()
torch.cuda.synchronize()
print("samples/ray", int(res["total_samples"]) / 8192, flush=True)
sc = captured["grid_bwd_param"][0]                  # colour scatter (first one launched in backward)
desc, x, dy, lddy, n, buf = sc
tbl = torch.zeros(desc.offsets[desc.n_levels] * desc.n_features, device=dev)
wg = captured["mlp_bwd_weight"][-1]                 # density head
dg = captured["mlp_bwd_input"][-1]
print("scatter n", n, "table rows", desc.offsets[desc.n_levels], flush=True)


def scatter():
    orig_call("grid_bwd_param", desc, x, dy, lddy, n, tbl)


def gemms():
    orig_call("mlp_bwd_weight", *wg)
    orig_call("mlp_bwd_input", *dg)


def time_on(stream, fn, reps=10):
    with torch.cuda.stream(stream):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def time_both(sa, fa, sb, fb, reps=10):
    torch.cuda.synchronize()
    main = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)
    for _ in range(reps):
        sa.wait_stream(main); sb.wait_stream(main)
        with torch.cuda.stream(sa):
            fa()
        with torch.cuda.stream(sb):
            fb()
        main.wait_stream(sa); main.wait_stream(sb)
    e1.record(main)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


full = masked_stream(range(256))
print(f"all 256 CUs: scatter {time_on(full, scatter):.3f} ms, gemms {time_on(full, gemms):.3f} ms, "
      f"side by side (two unmasked streams) {time_both(full, scatter, masked_stream(range(256)), gemms):.3f} ms", flush=True)
for layout in ("low", "strided"):
    for k in (32, 64, 96, 128):
        if layout == "low":
            a = list(range(k))
        else:
            step = 256 // k
            a = [i for i in range(256) if i % step == 0][:k] if 256 % k == 0 else [i for i in range(256) if (i * k) // 256 != ((i - 1) * k) // 256]
        b = [i for i in range(256) if i not in set(a)]
        sa, sb = masked_stream(a), masked_stream(b)
        ts, tg = time_on(sa, scatter), time_on(sb, gemms)
        tb = time_both(sa, scatter, sb, gemms)
        print(f"{layout:8s} scatter on {len(a):3d} CUs {ts:.3f} ms | gemms on {len(b):3d} CUs {tg:.3f} ms | side by side {tb:.3f} ms", flush=True)
