#!/usr/bin/env python3
"""Launches each MLP kernel a few times in isolation (n = 433 k) so that a rocprofv3 --pmc pass can
attribute SQ counters to them (tools/pmc_summary.py condenses the csv).  GPU only.

  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
            SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d /tmp/gp --output-format csv -- python3 tools/gemm_pmc.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ngp_amd
from ngp_amd._lib import call

dev = torch.device("cuda", 0)
torch.manual_seed(0)
n = 433000
x = torch.randn(n, 128, device=dev); W1 = torch.randn(128, 128, device=dev) * 0.1; W2 = torch.randn(1, 128, device=dev) * 0.1
hid = torch.empty(n, 128, device=dev); out = torch.empty(n, 1, device=dev)
dz2 = torch.randn(n, 1, device=dev); dx = torch.empty(n, 128, device=dev)
dW1 = torch.zeros(128, 128, device=dev); dW2 = torch.zeros(1, 128, device=dev); db = torch.zeros(128, device=dev); db2 = torch.zeros(1, device=dev)
for _ in range(3):
    call("mlp2_fwd", x, 128, W1, 128, None, 3, W2, 128, None, 3, n, 128, 128, 1, hid, 128, out, 1)
    call("linear_fwd", x, 128, W1, 128, None, n, 128, 128, 1, hid, 128, None)
    call("linear_bwd_input", x, 128, W1, 128, n, 128, 128, dx, 128, 0)
    call("linear_bwd_weight", x, 128, hid, 128, n, 128, 128, dW1, 128, None)
    call("mlp_bwd_input", dz2, 1, W2, 128, hid, 128, 3, W1, 128, n, 128, 128, 1, dx, 128, 0)
    call("mlp_bwd_weight", dz2, 1, W2, 128, hid, 128, 3, x, 128, n, 128, 128, 1, dW1, 128, db, dW2, 128, db2)
torch.cuda.synchronize()
print("done")
