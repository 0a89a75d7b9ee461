#!/usr/bin/env python3
"""ngp_mlp_bwd_weight (fused hidden-layer weight gradient + dW2 / db2 / db1) against fp64, with timing.
NGP_MLP_NO_STREAM=1 selects the tiled kernels instead of the streaming one.  GPU only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ngp_amd
from ngp_amd._lib import call

dev = torch.device("cuda", 0)
torch.manual_seed(0)
for n, n_in, n_out, act1, bias in ((433001, 128, 1, 3, True), (433001, 144, 3, 1, False), (433001, 160, 3, 1, False), (70003, 128, 2, 1, True), (37, 144, 4, 3, True)):
    H = 128
    hidden = torch.rand(n, H, device=dev) * 2 - (0.5 if act1 == 1 else 0.0)
    if act1 == 1:
        hidden.clamp_(min=0)
    x = torch.randn(n, n_in, device=dev)
    W2 = torch.randn(n_out, H, device=dev) * 0.1
    dz2 = torch.randn(n, n_out, device=dev)
    dW1 = torch.zeros(H, n_in, device=dev); db1 = torch.zeros(H, device=dev)
    dW2 = torch.zeros(n_out, H, device=dev); db2 = torch.zeros(n_out, device=dev)

    def run():
        call("mlp_bwd_weight", dz2, n_out, W2, H, hidden, H, act1, x, n_in, n, n_in, H, n_out, dW1, n_in,
             db1 if bias else None, dW2, H, db2 if bias else None)

    run(); torch.cuda.synchronize()
    h = hidden.double()
    g = (h > 0).double() if act1 == 1 else -torch.expm1(-h)
    dz1 = (dz2.double() @ W2.double()) * g
    refs = {"dW1": (dW1, dz1.T @ x.double()), "dW2": (dW2, dz2.double().T @ h)}
    if bias:
        refs["db1"] = (db1, dz1.sum(0)); refs["db2"] = (db2, dz2.double().sum(0))
    errs = {k: f"{float((a.double() - r).abs().max() / r.abs().max()):.1e}" for k, (a, r) in refs.items()}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20
    print(f"n={n} n_in={n_in} n_out={n_out} act1={act1}: {t:.3f} ms ({2.0 * n * H * n_in / t / 1e9:.1f} TF)   rel err {errs}", flush=True)
