#!/usr/bin/env python3
"""2-layer MLP forward: linear_fwd (MFMA) + linear_fwd (skinny) against ngp_mlp2_fwd.  GPU only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ngp_amd
from ngp_amd._lib import call

dev = torch.device("cuda", 0)
torch.manual_seed(0)
n = 433000
for n_in, H, n_out, act1, act2 in ((128, 128, 1, 3, 3), (160, 128, 3, 1, 2), (128, 32, 3, 1, 0)):
    x = torch.randn(n, n_in, device=dev)
    W1 = torch.randn(H, n_in, device=dev) * 0.1
    W2 = torch.randn(n_out, H, device=dev) * 0.1
    hidden = torch.empty(n, H, device=dev)
    out = torch.empty(n, n_out, device=dev)

    def plain():
        call("linear_fwd", x, n_in, W1, n_in, None, n, n_in, H, act1, hidden, H, None)
        call("linear_fwd", hidden, H, W2, H, None, n, H, n_out, act2, out, n_out, None)

    def fused():
        call("mlp2_fwd", x, n_in, W1, n_in, None, act1, W2, H, None, act2, n, n_in, H, n_out, hidden, H, out, n_out)

    def timeit(fn, reps=20):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    print(f"n_in={n_in} H={H} n_out={n_out}: plain {timeit(plain):.3f} ms   fused {timeit(fused):.3f} ms")
