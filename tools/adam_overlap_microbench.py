#!/usr/bin/env python3
"""How well does the streaming Adam pass (colour table, 153 M parameters) overlap with the
atomic-bound grid scatter and with the MFMA products on another stream? (GPU only)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ngp_amd
from ngp_amd._lib import call, GridDesc, call_host

dev = torch.device("cuda", 0)
n = 430000
torch.manual_seed(0)
o = torch.rand(n // 50, 1, 3, device=dev) * 0.6 + 0.2
d = torch.nn.functional.normalize(torch.randn(n // 50, 1, 3, device=dev), dim=-1)
t = torch.arange(50, device=dev).view(1, 50, 1) * (3 ** 0.5 / 1024)
x = (o + d * t).reshape(-1, 3).clamp(0, 1).contiguous()
n = x.shape[0]
desc = GridDesc()
npar = call_host("grid_layout", 16, 8, 19, 16, 1.3195079107728942, desc)
tbl = torch.zeros(npar, device=dev)
dy = torch.randn(n, 128, device=dev)
a = torch.randn(n, 128, device=dev); W = torch.randn(128, 128, device=dev) * 0.05
out = torch.empty(n, 128, device=dev); dW = torch.zeros(128, 128, device=dev)
na = 153_000_000
p, g, m, v = (torch.zeros(na, device=dev) for _ in range(4))
coef = torch.ones(1, device=dev)
side = torch.cuda.Stream()


def scatter():
    call("grid_bwd_param", desc, x, dy, 128, n, tbl)


def gemms():
    call("linear_bwd_weight", a, 128, dy, 128, n, 128, 128, dW, 128, None)
    call("linear_bwd_input", a, 128, W, 128, n, 128, 128, out, 128, 0)


def adam():
    call("adam_step", p, g, m, v, na, 1e-2, 0.9, 0.999, 1e-8, 0.0, 10, coef, 1)


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def par(f1, f2):
    def run():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            f2()
        f1()
        main.wait_stream(side)
    return run


def seq(*fs):
    def run():
        for f in fs:
            f()
    return run


ts, tg, ta = timeit(scatter), timeit(gemms), timeit(adam)
print(f"NGP_ADAM_BLOCKS={os.environ.get('NGP_ADAM_BLOCKS', '2048')}: scatter {ts:.3f}  wgrad+dgrad {tg:.3f}  adam {ta:.3f} ms")
print(f"  scatter || adam          : {timeit(par(scatter, adam)):.3f} ms  (sequential {ts+ta:.3f})")
print(f"  gemms   || adam          : {timeit(par(gemms, adam)):.3f} ms  (sequential {tg+ta:.3f})")
print(f"  gemms+scatter || adam    : {timeit(par(seq(gemms, scatter), adam)):.3f} ms  (sequential {tg+ts+ta:.3f})")
