import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, ngp_amd
from ngp_amd._lib import call
dev = torch.device("cuda", 0); n = 1000000
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for ni in (32, 64, 128, 256, 512):
    no = 128
    x = torch.randn(n, ni, device=dev); W = torch.randn(no, ni, device=dev) * 0.05
    y = torch.empty(n, no, device=dev)
    ms = timeit(lambda: call("linear_fwd", x, ni, W, ni, None, n, ni, no, 1, y, no, None))
    print(f"fwd K={ni:4d} N=128: {ms:.3f} ms  {2.0*n*ni*no/ms/1e9:.1f} TF")
