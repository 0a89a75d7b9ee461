#!/usr/bin/env python3
"""Backward of a 2-layer MLP as the field runs it: plain route (mlp_hidden_bwd materialises dz1, then
linear_bwd_*) against the operand-transform route (dz1 formed inside ngp_mlp_bwd_*).  GPU only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ngp_amd
from ngp_amd._lib import call

dev = torch.device("cuda", 0)
torch.manual_seed(0)
n = 433000
for H, n_in, n_out, act1 in ((128, 128, 1, 3), (128, 144, 3, 1), (32, 128, 3, 1)):
    hidden = torch.rand(n, H, device=dev)
    d_out = torch.randn(n, n_out, device=dev)
    out = torch.rand(n, n_out, device=dev)
    W2 = torch.randn(n_out, H, device=dev)
    W1 = torch.randn(H, n_in, device=dev)
    x = torch.randn(n, n_in, device=dev)
    dW1 = torch.zeros(H, n_in, device=dev); db1 = torch.zeros(H, device=dev)
    dW2 = torch.zeros(n_out, H, device=dev); db2 = torch.zeros(n_out, device=dev)
    dx = torch.empty(n, 128, device=dev)
    dz1 = torch.empty(n, H, device=dev)
    dz2 = torch.empty(n, n_out, device=dev)
    wide = n_in > 128
    rem = n_in - 128

    def plain():
        call("mlp_hidden_bwd", d_out, n_out, out, n_out, 2, W2, H, hidden, H, act1, n, H, n_out, None, 0, dz1, H, dW2, H, db2)
        if wide:
            call("linear_bwd_weight", dz1, H, x, n_in, n, 128, H, dW1, n_in, db1)
            call("linear_bwd_weight", dz1, H, x[:, 128:], n_in, n, rem, H, dW1[:, 128:], n_in, None)
        else:
            call("linear_bwd_weight", dz1, H, x, n_in, n, n_in, H, dW1, n_in, db1)
        call("linear_bwd_input", dz1, H, W1, n_in, n, 128, H, dx, 128, 0)

    def fused():
        call("act_bwd", d_out, out, n * n_out, 2, dz2)
        if wide:
            call("mlp_bwd_weight", dz2, n_out, W2, H, hidden, H, act1, x, n_in, n, 128, H, n_out, dW1, n_in, db1, None, 0, None)
            call("mlp_bwd_weight", dz2, n_out, W2, H, hidden, H, act1, x[:, 128:], n_in, n, rem, H, n_out, dW1[:, 128:], n_in,
                 None, dW2, H, db2)
        else:
            call("mlp_bwd_weight", dz2, n_out, W2, H, hidden, H, act1, x, n_in, n, n_in, H, n_out, dW1, n_in, db1, dW2, H, db2)
        call("mlp_bwd_input", dz2, n_out, W2, H, hidden, H, act1, W1, n_in, n, 128, H, n_out, dx, 128, 0)

    def timeit(fn, reps=20):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    print(f"H={H} n_in={n_in} n_out={n_out}: plain {timeit(plain):.3f} ms   fused {timeit(fused):.3f} ms")
