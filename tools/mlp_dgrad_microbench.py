#!/usr/bin/env python3
"""ngp_mlp_bwd_input (fused hidden-layer data gradient) against fp64, with timing.
NGP_MLP_NO_STREAM=1 selects the tiled kernel instead of the streaming one.  GPU only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ngp_amd
from ngp_amd._lib import call

dev = torch.device("cuda", 0)
torch.manual_seed(0)
n = 433001
for n_out, act1 in ((1, 3), (3, 1), (2, 1), (4, 3)):
    H = n_in = 128
    hidden = torch.rand(n, H, device=dev) * 2 - (0.5 if act1 == 1 else 0.0)      # relu outputs can be 0 / "negative" = off
    if act1 == 1:
        hidden.clamp_(min=0)
    W1 = torch.randn(H, n_in, device=dev) * 0.1
    W2 = torch.randn(n_out, H, device=dev) * 0.1
    dz2 = torch.randn(n, n_out, device=dev)
    dx = torch.full((n, n_in), float("nan"), device=dev)

    def run():
        call("mlp_bwd_input", dz2, n_out, W2, H, hidden, H, act1, W1, n_in, n, n_in, H, n_out, dx, n_in, 0)

    run(); torch.cuda.synchronize()
    sel = torch.cat([torch.arange(0, 4096, device=dev), torch.arange(n - 4096, n, device=dev)])
    h = hidden[sel].double()
    g = (h > 0).double() if act1 == 1 else -torch.expm1(-h)
    ref = ((dz2[sel].double() @ W2.double()) * g) @ W1.double()
    err = float((dx[sel].double() - ref).abs().max()); scale = float(ref.abs().max())
    nan = bool(torch.isnan(dx).any())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20
    print(f"n_out={n_out} act1={act1}: {t:.3f} ms ({2.0 * n * H * n_in / t / 1e9:.1f} TF)   max|err| {err:.2e} of {scale:.2e} nan={nan}", flush=True)
