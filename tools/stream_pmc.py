#!/usr/bin/env python3
"""mlp2_fwd alone (n = 433 k, 144 -> 128 -> 3, ReLU) for a rocprofv3 --pmc pass on the streaming kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ngp_amd
from ngp_amd._lib import call
dev = torch.device("cuda", 0)
torch.manual_seed(0)
n = 433000
x = torch.randn(n, 144, device=dev); W1 = torch.randn(128, 144, device=dev) * 0.1; W2 = torch.randn(3, 128, device=dev) * 0.1
hid = torch.empty(n, 128, device=dev); out = torch.empty(n, 3, device=dev)
for _ in range(3):
    call("mlp2_fwd", x, 144, W1, 144, None, 1, W2, 128, None, 2, n, 144, 128, 3, hid, 128, out, 3)
torch.cuda.synchronize()
print("done")
