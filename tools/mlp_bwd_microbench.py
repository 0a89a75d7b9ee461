#!/usr/bin/env python3
"""Backward of the first layer of a 2-layer MLP: mlp_hidden_bwd + linear_bwd_* (dz1 through HBM)
against the operand-transform products ngp_mlp_bwd_* (dz1 formed on the fly).  GPU only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ngp_amd
from ngp_amd._lib import call

dev = torch.device("cuda", 0)
torch.manual_seed(0)
n = 433000
for H, n_in, n_out, act1 in ((128, 128, 1, 3), (128, 160, 3, 1), (32, 128, 3, 1)):
    hidden = torch.rand(n, H, device=dev)
    d_out = torch.randn(n, n_out, device=dev)
    out = torch.rand(n, n_out, device=dev)
    W2 = torch.randn(n_out, H, device=dev)
    W1 = torch.randn(H, n_in, device=dev)
    x = torch.randn(n, n_in, device=dev)
    dW1 = torch.zeros(H, n_in, device=dev); db1 = torch.zeros(H, device=dev)
    dx = torch.empty(n, n_in, device=dev)
    dz2p = torch.empty(n, 4, device=dev); dz1 = torch.empty(n, H, device=dev)
    dz2 = torch.empty(n, n_out, device=dev)

    def plain():
        call("mlp_hidden_bwd", d_out, n_out, out, n_out, 2, W2, H, hidden, H, act1, n, H, n_out, dz2p, 4, dz1, H, None, 0, None)
        call("linear_bwd_weight", dz1, H, x, n_in, n, n_in, H, dW1, n_in, db1)
        call("linear_bwd_input", dz1, H, W1, n_in, n, n_in, H, dx, n_in, 0)

    def fused():
        call("act_bwd", d_out, out, n * n_out, 2, dz2)
        call("mlp_bwd_weight", dz2, n_out, W2, H, hidden, H, act1, x, n_in, n, n_in, H, n_out, dW1, n_in, db1, None, 0, None)
        call("mlp_bwd_input", dz2, n_out, W2, H, hidden, H, act1, W1, n_in, n, n_in, H, n_out, dx, n_in, 0)

    def timeit(fn, reps=20):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    print(f"H={H} n_in={n_in} n_out={n_out}: plain {timeit(plain):.3f} ms   fused {timeit(fused):.3f} ms")
