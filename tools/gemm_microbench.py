#!/usr/bin/env python3
"""Times the MLP layer kernels (GPU only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ngp_amd
from ngp_amd._lib import call

dev = torch.device("cuda", 0)
n = int(os.environ.get("MB_N", "1000000"))


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for (ni, no) in ((144, 128), (128, 128), (128, 32), (32, 16)):
    x = torch.randn(n, ni, device=dev); W = torch.randn(no, ni, device=dev) * 0.05; b = torch.randn(no, device=dev)
    y = torch.empty(n, no, device=dev); dz = torch.randn(n, no, device=dev); dx = torch.empty(n, ni, device=dev)
    dW = torch.zeros(no, ni, device=dev); db = torch.zeros(no, device=dev)
    fl = 2.0 * n * ni * no
    for act, name in ((0, "none"), (1, "relu"), (3, "softplus"), (2, "sigmoid")):
        ms = timeit(lambda: call("linear_fwd", x, ni, W, ni, b, n, ni, no, act, y, no, None))
        print(f"fwd  {ni:4d}->{no:4d} {name:9s} {ms:7.3f} ms {fl/ms/1e9:7.1f} TF  io {(n*(ni+no)*4)/ms/1e6:6.0f} GB/s")
    ms = timeit(lambda: call("linear_bwd_input", dz, no, W, ni, n, ni, no, dx, ni, 0))
    print(f"dgrad {ni:4d}<-{no:4d}          {ms:7.3f} ms {fl/ms/1e9:7.1f} TF  io {(n*(ni+no)*4)/ms/1e6:6.0f} GB/s")
    ms = timeit(lambda: call("linear_bwd_weight", dz, no, x, ni, n, ni, no, dW, ni, db))
    print(f"wgrad {no:4d}x{ni:4d}          {ms:7.3f} ms {fl/ms/1e9:7.1f} TF  io {(n*(ni+no)*4)/ms/1e6:6.0f} GB/s")
# skinny + hidden bwd
a1 = torch.randn(n, 128, device=dev); W2 = torch.randn(3, 128, device=dev); o3 = torch.empty(n, 3, device=dev)
ms = timeit(lambda: call("linear_fwd", a1, 128, W2, 128, None, n, 128, 3, 2, o3, 3, None))
print(f"skinny fwd 128->3 {ms:.3f} ms  {n*512/ms/1e6:.0f} GB/s")
d3 = torch.randn(n, 3, device=dev); dz2 = torch.empty(n, 4, device=dev); dz1 = torch.empty(n, 128, device=dev)
ms = timeit(lambda: call("mlp_hidden_bwd", d3, 3, o3, 3, 2, W2, 128, a1, 128, 1, n, 128, 3, dz2, 4, dz1, 128, None, 0, None))
print(f"hidden_bwd H=128 O=3 {ms:.3f} ms  {n*1024/ms/1e6:.0f} GB/s")
dW2 = torch.zeros(3, 128, device=dev)
ms = timeit(lambda: call("linear_bwd_weight", dz2, 4, a1, 128, n, 128, 3, dW2, 128, None))
print(f"skinny wgrad 3x128 {ms:.3f} ms  {n*512/ms/1e6:.0f} GB/s")
