#!/usr/bin/env python3
"""What kind of box is this?  Streaming rates of plain copies / fills and of the clip + Adam sweep on the flat buffers'
sizes (the sweep's per-launch time differs by up to 25 % between boxes of the pool: 0.64 ... 0.82 ms).  GPU only."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ngp_amd
from ngp_amd._lib import call

dev = torch.device("cuda", 0)
n = 154_500_000          # the colour table's parameter count
p, g, m, v = (torch.randn(n, device=dev) * 1e-2 for _ in range(4))
v.abs_()
coef = torch.ones(1, device=dev)


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


t = timeit(lambda: m.copy_(p))
print(f"copy  {n * 8 / t / 1e6:8.0f} GB/s  ({t:.3f} ms for {n * 8 / 1e9:.2f} GB read + written)")
t = timeit(lambda: m.zero_())
print(f"fill  {n * 4 / t / 1e6:8.0f} GB/s  ({t:.3f} ms)")
t = timeit(lambda: torch.add(p, g, out=m))
print(f"add   {n * 12 / t / 1e6:8.0f} GB/s  ({t:.3f} ms, 2 reads + 1 write)")
m.normal_(); m.mul_(1e-3)
t = timeit(lambda: call("adam_step", p, g, m, v, n, 1e-2, 0.9, 0.999, 1e-8, 0.0, 7, coef, 1))
print(f"adam  {n * 28 / t / 1e6:8.0f} GB/s  ({t:.3f} ms, 28 B per parameter)")
for cmd in (["rocm-smi", "--showclocks", "--showpower", "--showtemp"],):
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=20).stdout
        print("\n".join(l for l in out.splitlines() if any(k in l.lower() for k in ("sclk", "mclk", "fclk", "power", "temperature (sensor junction)", "temperature (sensor memory)")))[:1500])
    except Exception as e:
        print("rocm-smi:", e)
