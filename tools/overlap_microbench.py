#!/usr/bin/env python3
"""Does the atomic-bound grid scatter overlap with MFMA GEMMs on another stream? (GPU only)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ngp_amd
from ngp_amd._lib import call, GridDesc, call_host

dev = torch.device("cuda", 0)
n = 430000
torch.manual_seed(0)
# realistic positions: rays through the unit cube, 50 consecutive samples per ray
o = torch.rand(n // 50, 1, 3, device=dev) * 0.6 + 0.2
d = torch.nn.functional.normalize(torch.randn(n // 50, 1, 3, device=dev), dim=-1)
t = torch.arange(50, device=dev).view(1, 50, 1) * (3 ** 0.5 / 1024)
x = (o + d * t).reshape(-1, 3).clamp(0, 1).contiguous()
n = x.shape[0]
desc = GridDesc()
npar = call_host("grid_layout", 16, 8, 21, 16, 1.3195079107728942, desc)
tbl = torch.zeros(npar, device=dev)
dy = torch.randn(n, 128, device=dev)
a = torch.randn(n, 128, device=dev); W = torch.randn(128, 128, device=dev) * 0.05
out = torch.empty(n, 128, device=dev); dW = torch.zeros(128, 128, device=dev)
side = torch.cuda.Stream()


def scatter():
    call("grid_bwd_param", desc, x, dy, 128, n, tbl)


def gemms():
    call("linear_bwd_weight", a, 128, dy, 128, n, 128, 128, dW, 128, None)
    call("linear_bwd_input", a, 128, W, 128, n, 128, 128, out, 128, 0)
    call("linear_fwd", a, 128, W, 128, None, n, 128, 128, 1, out, 128, None)


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def both():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        scatter()
    gemms()
    main.wait_stream(side)


for cap in ("0", "20000", "40000", "60000", "80000", "120000"):
    os.environ["NGP_SCATTER_LDS"] = cap
    ts, tg, tb = timeit(scatter), timeit(gemms), timeit(both)
    print(f"lds_cap={cap:>6s}: scatter {ts:.3f} ms, 3 gemms {tg:.3f} ms, sequential {ts+tg:.3f} ms, overlapped {tb:.3f} ms")
