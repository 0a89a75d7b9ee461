#!/usr/bin/env python3
"""2-layer MLP forward: linear_fwd (MFMA) + linear_fwd (skinny) against ngp_mlp2_fwd, with a check
against fp64.  NGP_MLP_NO_STREAM=1 selects the tiled kernel instead of the streaming one.  GPU only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ngp_amd
from ngp_amd._lib import call

dev = torch.device("cuda", 0)
torch.manual_seed(0)
n = 433001
ACT = {0: lambda v: v, 1: torch.relu, 2: torch.sigmoid, 3: torch.nn.functional.softplus}
for n_in, H, n_out, act1, act2 in ((128, 128, 1, 3, 3), (144, 128, 3, 1, 2), (160, 128, 3, 1, 2), (144, 128, 7, 1, 0), (128, 32, 3, 1, 0)):
    x = torch.randn(n, n_in, device=dev)
    W1 = torch.randn(H, n_in, device=dev) * 0.1
    b1 = torch.randn(H, device=dev) * 0.1
    W2 = torch.randn(n_out, H, device=dev) * 0.1
    b2 = torch.randn(n_out, device=dev) * 0.1
    hidden = torch.empty(n, H, device=dev)
    out = torch.empty(n, n_out, device=dev)

    def plain():
        call("linear_fwd", x, n_in, W1, n_in, b1, n, n_in, H, act1, hidden, H, None)
        call("linear_fwd", hidden, H, W2, H, b2, n, H, n_out, act2, out, n_out, None)

    def fused():
        call("mlp2_fwd", x, n_in, W1, n_in, b1, act1, W2, H, b2, act2, n, n_in, H, n_out, hidden, H, out, n_out)

    def timeit(fn, reps=20):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    hidden.fill_(float("nan")); out.fill_(float("nan"))
    fused(); torch.cuda.synchronize()
    sel = torch.cat([torch.arange(0, 4096, device=dev), torch.arange(n - 4096, n, device=dev)])
    h64 = ACT[act1](x[sel].double() @ W1.double().T + b1.double())
    o64 = ACT[act2](h64 @ W2.double().T + b2.double())
    eh = float((hidden[sel].double() - h64).abs().max()); eo = float((out[sel].double() - o64).abs().max())
    nan = bool(torch.isnan(hidden).any() or torch.isnan(out).any())
    flops = 2.0 * n * (n_in * H + H * n_out)
    tf = timeit(fused)
    print(f"n_in={n_in} H={H} n_out={n_out}: plain {timeit(plain):.3f} ms   fused {tf:.3f} ms ({flops / tf / 1e9:.1f} TF)"
          f"   max|err| hidden {eh:.2e} out {eo:.2e} nan={nan}", flush=True)
