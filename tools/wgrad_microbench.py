import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, ngp_amd
from ngp_amd._lib import call
dev = torch.device("cuda", 0); n = 430000
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
x = torch.randn(n, 128, device=dev); dz = torch.randn(n, 128, device=dev); dW = torch.zeros(128, 128, device=dev); db = torch.zeros(128, device=dev)
ms = timeit(lambda: call("linear_bwd_weight", dz, 128, x, 128, n, 128, 128, dW, 128, db))
print(os.environ.get("NGP_WGRAD_BLOCKS"), f"wgrad 128x128 n={n}: {ms:.3f} ms {2.0*n*128*128/ms/1e9:.1f} TF")
