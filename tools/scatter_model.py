#!/usr/bin/env python3
"""Counts the memory-side 64-byte atomic requests of the hash-grid scatter per level, for several
accumulation policies, by replaying the flush logic on the CPU (tools/scatter_model.c) over sample
positions captured from a training step (tools/capture_scatter_inputs.py -> gpurun_out/scatter_capture.npz).

usage: python tools/scatter_model.py [capture.npz]      (without a capture: synthetic rays through a shell)
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = "/tmp/scatter_model.so"


class Level(C.Structure):
    _fields_ = [("size", C.c_uint32), ("res", C.c_uint32), ("hashed", C.c_uint32), ("pow2", C.c_uint32),
                ("scale", C.c_float)]


def lib():
    src = os.path.join(HERE, "scatter_model.c")
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", SO, src, "-lm"])
    return C.CDLL(SO)


def synthetic(n_rays=2048, seed=0):
    """rays from a sphere of radius 1.5 through a thin occupied shell of radius 0.3 in [-0.5,0.5]^3, step sqrt(3)/1024"""
    g = np.random.default_rng(seed)
    o = g.standard_normal((n_rays, 3)); o = 1.5 * o / np.linalg.norm(o, axis=1, keepdims=True)
    tgt = (g.random((n_rays, 3)) - 0.5) * 0.5
    d = tgt - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
    dt = 3 ** 0.5 / 1024
    xs, nzs = [], []
    for i in range(n_rays):
        t = np.arange(0.9, 2.1, dt) + g.random() * dt
        p = o[i] + t[:, None] * d[i]
        r = np.linalg.norm(p, axis=1)
        m = (np.abs(r - 0.3) < 0.02) & (np.abs(p).max(1) < 0.5)
        p = p[m][:60]
        xs.append(p + 0.5)
        nz = np.ones(len(p), bool); nz[int(len(p) * 0.8):] = False    # ~20 % behind the termination point
        nzs.append(nz)
    x = np.concatenate(xs).astype(np.float32)
    nz = np.repeat(np.concatenate(nzs)[:, None], 16, 1)
    return [(x, nz, 19), (x, nz, 21)]


def main():
    L = 16
    if len(sys.argv) > 1:
        cap = np.load(sys.argv[1])
        launches = []
        k = 0
        while f"x{k}" in cap:
            x = cap[f"x{k}"]
            nz = np.unpackbits(cap[f"nz{k}"], axis=1)[:, :L].astype(bool)
            rows = int(cap[f"rows{k}"])
            launches.append((x, nz, 21 if rows > 10_000_000 else 19))
            k += 1
    else:
        launches = synthetic()
    m = lib()
    b = float(np.exp(np.log(2048 * 0.5 / 16) / 15))
    for x, nz, log2_T in launches:
        n = x.shape[0]
        lv = (Level * L)()
        m.model_layout(L, log2_T, 16, C.c_double(b), lv)
        x = np.ascontiguousarray(x, np.float32)
        nzb = np.ascontiguousarray(nz, np.uint8)
        print(f"\n=== table T=2^{log2_T}: n = {n} samples, non-zero (sample,level) fraction {nz.mean():.3f}")
        res = {}
        for name, chunk, policy in (("shipped slide c32", 32, 0), ("slide c64", 64, 0), ("slide c128", 128, 0),
                                    ("line window c32", 32, 1), ("line window c64", 64, 1), ("line window c128", 128, 1),
                                    ("floor c32", 32, 2), ("floor c128", 128, 2), ("floor c4096", 4096, 2),
                                    ("floor batch", 1 << 30, 2)):
            counts = np.zeros(2 * L, np.int64)
            m.model_run(x.ctypes.data_as(C.c_void_p), nzb.ctypes.data_as(C.c_void_p), C.c_int64(n), L, lv, chunk, policy,
                        counts.ctypes.data_as(C.c_void_p))
            res[name] = counts.reshape(L, 2)
        names = list(res)
        print("requests per sample, per level (res, H = hashed):")
        print("level   res  " + "".join(f"{nm:>18s}" for nm in names))
        for l in range(L):
            tag = "H" if lv[l].hashed else " "
            print(f"{l:5d} {lv[l].res:5d}{tag} " + "".join(f"{res[nm][l, 0] / n:18.3f}" for nm in names))
        print("total        " + "".join(f"{res[nm][:, 0].sum() / n:18.2f}" for nm in names))
        print("rows/request " + "".join(f"{(res[nm][:, 1].sum() / max(res[nm][:, 0].sum(), 1)) if nm.find('floor') < 0 else float('nan'):18.2f}" for nm in names))


if __name__ == "__main__":
    main()
