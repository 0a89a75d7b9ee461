"""GPU parity tests: the HIP path (through the C ABI, via the vren/tinycudann-shaped modules)
against the CPU oracle on the same seeded inputs.  Run with `-m gpu` on an MI355X.

Bars: bit-exact for integer/index work and for the fp32 marcher/intersector (same operation
order, no FMA contraction on either side); stated tolerances for compositing (parallel scans,
__expf), the hash grid (atomic summation order) and the MLP (MFMA accumulation order).
"""
import os
import sys

import numpy as np
import pytest
import torch

import oracle
from helpers import borderline_rays, make_bitfield, make_rays, make_segments, rng

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def N(t):
    return t.detach().cpu().numpy()


def close(a, b, rtol, atol):
    np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=rtol, atol=atol)


# ---------------------------------------------------------------------------- R1 / O1
@pytest.mark.parametrize("n_rays,n_vox,max_hits", [(0, 1, 1), (1, 1, 1), (1000, 1, 1), (513, 7, 3), (300, 20, 4)])
def test_ray_aabb_intersect_bit_exact(ngp, n_rays, n_vox, max_hits):
    g = rng(10)
    o, d = make_rays(max(n_rays, 1), scale=1.0, seed=11)
    o, d = o[:n_rays], d[:n_rays]
    if n_rays > 10:
        d[3, 0] = 0.0  # axis-parallel ray: 1/d = inf
        d[4] = [0.0, 0.0, 1.0]
    centers = ((g.random((n_vox, 3)) - 0.5) * (0.0 if n_vox == 1 else 1.5)).astype(np.float32)
    half = (0.2 + 0.5 * g.random((n_vox, 3))).astype(np.float32)
    cnt, t, idx = ngp.vren.ray_aabb_intersect(T(o), T(d), T(centers), T(half), max_hits)
    ocnt, ot, oidx = oracle.ray_aabb_intersect(o, d, centers, half, max_hits)
    assert np.array_equal(N(cnt), ocnt)
    assert np.array_equal(N(t), ot, equal_nan=True)
    assert np.array_equal(N(idx), oidx)


def test_ray_sphere_intersect_bit_exact(ngp):
    g = rng(12)
    o, d = make_rays(777, scale=1.0, seed=13)
    centers = ((g.random((9, 3)) - 0.5) * 1.5).astype(np.float32)
    radii = (0.1 + 0.4 * g.random(9)).astype(np.float32)
    cnt, t, idx = ngp.vren.ray_sphere_intersect(T(o), T(d), T(centers), T(radii), 4)
    ocnt, ot, oidx = oracle.ray_sphere_intersect(o, d, centers, radii, 4)
    assert np.array_equal(N(cnt), ocnt)
    assert np.array_equal(N(t), ot, equal_nan=True)
    assert np.array_equal(N(idx), oidx)


def test_morton_roundtrip_and_oracle(ngp):
    g = rng(14)
    coords = g.integers(0, 128, (5000, 3)).astype(np.int32)
    idx = ngp.vren.morton3D(T(coords))
    assert np.array_equal(N(idx), oracle.morton3D(coords))
    back = ngp.vren.morton3D_invert(idx)
    assert np.array_equal(N(back), coords)
    assert np.array_equal(N(back), oracle.morton3D_invert(N(idx)))
    # full 128^3 grid is a permutation
    gx = np.stack(np.meshgrid(*[np.arange(128)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.int32)
    full = N(ngp.vren.morton3D(T(gx)))
    assert np.array_equal(np.sort(full), np.arange(128 ** 3))


def test_packbits_bit_exact(ngp):
    g = rng(15)
    grid = (g.random(2 * 64 ** 3) * 10 - 1).astype(np.float32)
    grid[::7] = -1.0
    out = torch.zeros(grid.size // 8, dtype=torch.uint8, device=DEV)
    ngp.vren.packbits(T(grid), 4.5, out)
    assert np.array_equal(N(out), oracle.packbits(grid, 4.5))


# ---------------------------------------------------------------------------- R3 / T1
MARCH_CASES = [
    # cascades, scale, exp_step_factor, fill, n_rays, max_samples
    (1, 0.5, 0.0, 0.05, 2048, 1024),
    (1, 0.5, 0.0, 1.00, 257, 1024),     # warm-up: everything occupied, rays hit the 1024 cap region
    (5, 8.0, 1 / 256, 0.03, 1500, 1024),
    (6, 16.0, 1 / 256, 0.02, 700, 256),
    (1, 0.5, 0.0, 0.0, 64, 1024),       # empty grid: no samples at all
    (6, 16.0, 0.0, 0.02, 300, 1024),    # constant fine step in a 32-wide box: chains of >10^4 elements, i.e. many
                                        # 1024-element segments and skips that end beyond a segment (wave marcher)
    (3, 2.0, 0.0, 0.3, 400, 128),       # dense occupancy with a small sample cap reached mid-segment
]


def _march_inputs(cascades, scale, fill, n_rays, seed):
    o, d = make_rays(n_rays, scale=min(scale, 2.0), seed=seed)
    center = np.zeros((1, 3), np.float32)
    half = np.full((1, 3), scale, np.float32)
    _, hits_t, _ = oracle.ray_aabb_intersect(o, d, center, half, 1)
    hits_t = hits_t[:, 0]
    m = (hits_t[:, 0] >= 0) & (hits_t[:, 0] < 0.01)
    hits_t[m, 0] = 0.01
    bits = make_bitfield(cascades, 128, fill=fill, seed=seed + 1) if fill > 0 else np.zeros(cascades * 128 ** 3 // 8, np.uint8)
    noise = rng(seed + 2).random(n_rays).astype(np.float32)
    return o, d, np.ascontiguousarray(hits_t), bits, noise


@pytest.mark.parametrize("case", MARCH_CASES)
def test_raymarching_train_bit_exact(ngp, case):
    cascades, scale, esf, fill, n_rays, max_samples = case
    o, d, hits_t, bits, noise = _march_inputs(cascades, scale, fill, n_rays, seed=20)
    ra, xyz, dirs, dl, ts, cnt = ngp.vren.raymarching_train(T(o), T(d), T(hits_t), T(bits), cascades, scale, esf,
                                                            T(noise), 128, max_samples)
    ora, oxyz, odirs, odl, ots, ocnt = oracle.raymarching_train(o, d, hits_t, bits, cascades, scale, esf, noise, 128,
                                                               max_samples)
    assert np.array_equal(N(cnt), ocnt)
    n = int(ocnt[0])
    if fill > 0 and n_rays > 100:
        assert n > 0
    assert np.array_equal(N(ra), ora)
    assert np.array_equal(N(ts)[:n], ots[:n])
    assert np.array_equal(N(dl)[:n], odl[:n])
    assert np.array_equal(N(xyz)[:n], oxyz[:n])
    assert np.array_equal(N(dirs)[:n], odirs[:n])
    # drop-in contract: rows behind counter[0] are zero (reference allocates torch::zeros)
    assert not N(ts)[n:].any() and not N(xyz)[n:].any()


def test_raymarcher_function_trims_and_orders(ngp):
    cascades, scale, esf = 1, 0.5, 0.0
    o, d, hits_t, bits, _ = _march_inputs(cascades, scale, 0.05, 1000, seed=30)
    torch.manual_seed(1)
    ra, xyz, dirs, dl, ts, total = ngp.custom_functions.RayMarcher.apply(
        T(o), T(d), T(hits_t), T(bits), cascades, scale, esf, 128, 1024)
    n = int(total)
    assert xyz.shape == (n, 3) and dirs.shape == (n, 3) and dl.shape == (n,) and ts.shape == (n,)
    ra = N(ra)
    assert np.array_equal(ra[:, 0], np.arange(1000))
    assert np.array_equal(ra[:, 1], np.cumsum(ra[:, 2]) - ra[:, 2])
    assert ra[:, 2].sum() == n
    # samples of a ray are strictly increasing in t and spaced by delta
    tsn, dln = N(ts), N(dl)
    for r in range(0, 1000, 37):
        s, c = ra[r, 1], ra[r, 2]
        if c > 1:
            assert np.all(np.diff(tsn[s:s + c]) >= dln[s:s + c - 1] * 0.999)


@pytest.mark.parametrize("case", [(1, 0.5, 0.0, 0.05, 1500), (5, 8.0, 1 / 256, 0.03, 900)])
def test_raymarching_test_bit_exact(ngp, case):
    cascades, scale, esf, fill, n_rays = case
    o, d, hits_t, bits, _ = _march_inputs(cascades, scale, fill, n_rays, seed=40)
    alive = np.arange(n_rays, dtype=np.int64)[::2].copy()
    hits_gpu = T(hits_t.copy())
    hits_cpu = hits_t.copy()
    for n_samples in (1, 4, 64):  # successive rounds resume from the mutated hits_t
        out = ngp.vren.raymarching_test(T(o), T(d), hits_gpu, T(alive), T(bits), cascades, scale, esf, 128, 1024,
                                        n_samples)
        ref = oracle.raymarching_test(o, d, hits_cpu, alive, bits, cascades, scale, esf, 128, 1024, n_samples)
        for a, b in zip(out, ref):
            assert np.array_equal(N(a), b)
        assert np.array_equal(N(hits_gpu), hits_cpu)


# ---------------------------------------------------------------------------- V1 / V2 / V3 / D1
def _composite_inputs(n_rays, max_len, classes, seed, sigma_scale=30.0):
    g = rng(seed)
    rays_a, n = make_segments(n_rays, max_len, seed=seed + 1)
    # shuffle rows: the kernels must honour ray_idx, not the row number
    perm = g.permutation(n_rays)
    rays_a = rays_a[perm]
    sig = (g.random(n) ** 3 * sigma_scale).astype(np.float32)
    sig[g.random(n) < 0.3] = 0.0
    deltas = (0.002 + 0.02 * g.random(n)).astype(np.float32)
    ts = np.zeros(n, np.float32)
    for _, s, c in rays_a:
        ts[s:s + c] = 0.5 + np.cumsum(deltas[s:s + c])
    rgbs = g.random((n, 3)).astype(np.float32)
    nrm = g.normal(size=(n, 3)).astype(np.float32)
    sems = g.random((n, classes)).astype(np.float32)
    return rays_a, n, sig, deltas, ts, rgbs, nrm, sems


@pytest.mark.parametrize("n_rays,max_len,classes,T_thr", [
    (1, 1, 7, 1e-4), (700, 40, 7, 1e-4), (300, 700, 7, 1e-4), (200, 90, 0, 1e-2), (150, 64, 10, 0.0),
    (64, 33, 40, 1e-4)])
def test_composite_train_fw_bw(ngp, n_rays, max_len, classes, T_thr):
    rays_a, n, sig, deltas, ts, rgbs, nrm, sems = _composite_inputs(n_rays, max_len, classes, seed=50 + max_len)
    ok = ~borderline_rays(sig, deltas, rays_a, T_thr)
    out = ngp.vren.composite_train_fw(T(sig), T(rgbs), T(nrm), T(sems), T(deltas), T(ts), T(rays_a), T_thr, classes)
    ref = oracle.composite_train_fw(sig, rgbs, nrm, sems, deltas, ts, rays_a, T_thr, classes)
    total, opacity, depth, rgb, normal, sem, ws = [N(x) for x in out]
    ray_ok = np.zeros(n_rays, bool)
    ray_ok[rays_a[ok, 0]] = True
    assert np.array_equal(total[ray_ok], ref[0][ray_ok])
    for a, b in zip((opacity, depth, rgb, normal, sem), ref[1:6]):
        close(a[ray_ok], b[ray_ok], rtol=2e-5, atol=2e-6)
    smask = np.zeros(n, bool)
    for i, (_, s, c) in enumerate(rays_a):
        smask[s:s + c] = ok[i]
    close(ws[smask], ref[6][smask], rtol=2e-5, atol=1e-7)

    # backward with random upstream gradients, fed the ORACLE forward outputs on both sides
    g = rng(60)
    dO, dD = g.normal(size=n_rays).astype(np.float32), g.normal(size=n_rays).astype(np.float32)
    dRGB, dN = g.normal(size=(n_rays, 3)).astype(np.float32), g.normal(size=(n_rays, 3)).astype(np.float32)
    dS, dws = g.normal(size=(n_rays, classes)).astype(np.float32), g.normal(size=n).astype(np.float32)
    bw = ngp.vren.composite_train_bw(T(dO), T(dD), T(dRGB), T(dN), T(dS), T(dws), T(sig), T(rgbs), T(nrm),
                                     T(ref[6]), T(deltas), T(ts), T(rays_a), T(ref[1]), T(ref[2]), T(ref[3]),
                                     T(ref[4]), T_thr, classes)
    rbw = oracle.composite_train_bw(dO, dD, dRGB, dN, dS, dws, sig, rgbs, nrm, ref[6], deltas, ts, rays_a, ref[1],
                                    ref[2], ref[3], ref[4], T_thr, classes)
    for a, b in zip(bw, rbw):
        a = N(a)
        # dL_dsigmas is a difference of O(1) running sums: absolute tolerance scaled by delta
        close(a[smask], b[smask], rtol=2e-4, atol=2e-5)


def test_composite_alpha_fw(ngp):
    rays_a, n, sig, deltas, *_ = _composite_inputs(400, 50, 0, seed=70)
    ok = ~borderline_rays(sig, deltas, rays_a, 1e-4)
    a, w = ngp.vren.composite_alpha_fw(T(sig), T(deltas), T(rays_a), 1e-4)
    ra, rw = oracle.composite_alpha_fw(sig, deltas, rays_a, 1e-4)
    smask = np.zeros(n, bool)
    for i, (_, s, c) in enumerate(rays_a):
        smask[s:s + c] = ok[i]
    close(N(a)[smask], ra[smask], 2e-5, 1e-7)
    close(N(w)[smask], rw[smask], 2e-5, 1e-7)


def test_refloss_fw_bw(ngp):
    rays_a, n, sig, deltas, ts, rgbs, nrm, _ = _composite_inputs(500, 60, 0, seed=80)
    ok = ~borderline_rays(sig, deltas, rays_a, 1e-4)
    ndiff = (rgbs ** 2).astype(np.float32)
    nori = np.abs(nrm[:, 0]).astype(np.float32)
    lo, lp = ngp.vren.composite_refloss_fw(T(sig), T(ndiff), T(nori), T(deltas), T(ts), T(rays_a), 1e-4)
    rlo, rlp = oracle.composite_refloss_fw(sig, ndiff, nori, deltas, ts, rays_a, 1e-4)
    ray_ok = np.zeros(len(rays_a), bool)
    ray_ok[rays_a[ok, 0]] = True
    close(N(lo)[ray_ok], rlo[ray_ok], 2e-5, 2e-6)
    close(N(lp)[ray_ok], rlp[ray_ok], 2e-5, 2e-6)
    g = rng(81)
    dlo, dlp = g.normal(size=len(rays_a)).astype(np.float32), g.normal(size=(len(rays_a), 3)).astype(np.float32)
    bw = ngp.vren.composite_refloss_bw(T(dlo), T(dlp), T(sig), T(ndiff), T(nori), T(deltas), T(ts), T(rays_a),
                                       T(rlo), T(rlp), 1e-4)
    rbw = oracle.composite_refloss_bw(dlo, dlp, sig, ndiff, nori, deltas, ts, rays_a, rlo, rlp, 1e-4)
    smask = np.zeros(n, bool)
    for i, (_, s, c) in enumerate(rays_a):
        smask[s:s + c] = ok[i]
    for a, b in zip(bw, rbw):
        close(N(a)[smask], b[smask], 2e-4, 2e-5)


@pytest.mark.parametrize("n_rays,max_len", [(600, 40), (100, 900), (5, 1)])
def test_distortion_loss_fw_bw(ngp, n_rays, max_len):
    rays_a, n, sig, deltas, ts, *_ = _composite_inputs(n_rays, max_len, 0, seed=90 + max_len)
    ws = oracle.composite_train_fw(sig, np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32),
                                   np.zeros((n, 0), np.float32), deltas, ts, rays_a, 1e-4, 0)[6]
    loss, wi, wti = ngp.vren.distortion_loss_fw(T(ws), T(deltas), T(ts), T(rays_a))
    rl, rwi, rwti = oracle.distortion_loss_fw(ws, deltas, ts, rays_a)
    close(N(wi), rwi, 1e-5, 1e-7)
    close(N(wti), rwti, 1e-5, 1e-7)
    # the loss is a cancellation of O(w*wt) products: absolute tolerance on that scale
    close(N(loss), rl, 1e-3, 2e-6)
    g = rng(91)
    dl = g.normal(size=n_rays).astype(np.float32)
    d = ngp.vren.distortion_loss_bw(T(dl), T(rwi), T(rwti), T(ws), T(deltas), T(ts), T(rays_a))
    rd = oracle.distortion_loss_bw(dl, rwi, rwti, ws, deltas, ts, rays_a)
    close(N(d), rd, 1e-5, 1e-6)


def test_composite_test_fw(ngp):
    g = rng(100)
    n_rays, na, ns, C = 900, 500, 16, 7
    alive = np.sort(g.choice(n_rays, na, replace=False)).astype(np.int64)
    sig = (g.random((na, ns)) ** 2 * 40).astype(np.float32)
    deltas = (0.002 + 0.01 * g.random((na, ns))).astype(np.float32)
    ts = np.cumsum(deltas, 1).astype(np.float32)
    rgbs, nrm, nraw = (g.random((na, ns, 3)).astype(np.float32) for _ in range(3))
    sems = g.random((na, ns, C)).astype(np.float32)
    n_eff = g.integers(0, ns + 1, na).astype(np.int32)
    hits_t = np.zeros((n_rays, 2), np.float32)
    state = [g.random(n_rays).astype(np.float32) * 0.5, g.random(n_rays).astype(np.float32),
             g.random((n_rays, 3)).astype(np.float32), g.random((n_rays, 3)).astype(np.float32),
             g.random((n_rays, 3)).astype(np.float32), g.random((n_rays, C)).astype(np.float32)]
    gpu_state = [T(s.copy()) for s in state]
    gpu_alive = T(alive.copy())
    ngp.vren.composite_test_fw(T(sig), T(rgbs), T(nrm), T(nraw), T(sems), T(deltas), T(ts), T(hits_t), gpu_alive,
                               1e-2, C, T(n_eff), *gpu_state)
    cpu_alive = alive.copy()
    oracle.composite_test_fw(sig, rgbs, nrm, nraw, sems, deltas, ts, hits_t, cpu_alive, 1e-2, C, n_eff, *state)
    # alive flags may differ only for rays whose T lands within rounding of the threshold
    agree = N(gpu_alive) == cpu_alive
    assert agree.mean() > 0.995
    for a, b in zip(gpu_state, state):
        close(N(a), b, 2e-5, 2e-6)


def test_segment_csr(ngp):
    g = rng(110)
    rays_a, n = make_segments(300, 50, seed=111)
    src = g.normal(size=(n, 3)).astype(np.float32)
    indptr = np.concatenate([rays_a[:, 1], rays_a[-1:, 1] + rays_a[-1:, 2]])
    out = ngp.torch_scatter.segment_csr(T(src), T(indptr))
    close(N(out), oracle.segment_csr_sum(src, indptr), 1e-5, 1e-6)


# ---------------------------------------------------------------------------- H1-H5
GRID_CASES = [
    # L, F, log2_T, base, per_level_scale, n
    (16, 8, 19, 16, 1.3195079, 3000),      # the reference's xyz_encoder at scale 0.5
    (16, 8, 14, 16, 1.3195079, 2000),      # small table: heavy hash collisions
    (8, 2, 16, 16, 2.0, 2500),             # implicit_mask.py configuration
    (16, 2, 19, 16, 1.3819, 1500),
    (4, 4, 12, 8, 1.5, 1000),
    (3, 1, 10, 4, 2.0, 777),               # group of 3 lanes: serial input-grad fallback
]


@pytest.mark.parametrize("case", GRID_CASES)
def test_grid_fwd_bwd(ngp, case):
    L, Fd, log2T, base, pls, n = case
    tcnn = ngp.tinycudann
    enc = tcnn.Encoding(3, {"otype": "HashGrid", "n_levels": L, "n_features_per_level": Fd,
                            "log2_hashmap_size": log2T, "base_resolution": base, "per_level_scale": pls}).to(DEV)
    desc, n_params = oracle.grid_layout(L, Fd, log2T, base, pls)
    assert n_params == enc.params.numel()
    assert list(desc.offsets)[:L + 1] == list(enc.desc.offsets)[:L + 1]
    g = rng(120 + L)
    table = g.uniform(-1, 1, n_params).astype(np.float32)
    x = g.random((n, 3)).astype(np.float32)
    x[0] = [0.0, 0.0, 0.0]
    x[1] = [1.0, 1.0, 1.0]      # the far corner (index wrap path)
    x[2] = [0.5, 0.25, 0.75]    # lands exactly on cell boundaries at power-of-two resolutions
    with torch.no_grad():
        enc.params.copy_(T(table))
    xt = T(x).requires_grad_(True)
    y = enc(xt)
    ry = oracle.grid_fwd(desc, table, x)
    close(N(y), ry, 1e-5, 1e-6)

    dy = g.normal(size=ry.shape).astype(np.float32)
    dy[5:50] = 0.0              # zero-gradient samples (skipped by the scatter kernel)
    gx, gp = torch.autograd.grad(y, [xt, enc.params], T(dy))
    rgx = oracle.grid_bwd_input(desc, table, x, dy)
    # sums of ~L*F*8 terms of size scale*|dy|*|table|: tolerance relative to that magnitude
    close(N(gx), rgx, 2e-4, 3e-6 * np.abs(rgx).max())
    rgp = oracle.grid_bwd_param(desc, x, dy, n_params)
    # atomic accumulation order differs from the sequential oracle
    close(N(gp), rgp, 1e-4, 1e-4 * max(1.0, np.abs(rgp).max() * 0.01))


@pytest.mark.parametrize("log2T", [14, 19])
def test_grid_bwd_param_ray_ordered(ngp, log2T):
    """The scatter keeps per-line running sums while consecutive samples stay in (or step between) the same
    64-byte lines.  Random points almost never do that, so this case feeds what training feeds: samples marched
    along rays (step sqrt(3)/1024), in both directions along every axis, axis-aligned rays (pure face moves),
    diagonal rays, a coarse-step ray (jumps of several cells: every sum leaves), repeated positions,
    zero-gradient tails, ray boundaries inside a chunk."""
    from ngp_amd._lib import call
    L, Fd, base = 16, 8, 16
    pls = 1.3195079107728942
    desc, n_params = oracle.grid_layout(L, Fd, log2T, base, pls)
    g = rng(150 + log2T)
    step = np.float32(3 ** 0.5 / 1024)
    xs = []
    dirs = [g.normal(size=3) for _ in range(300)]
    dirs += [np.array(v, float) for v in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1),
                                          (1, 1, 1), (-1, -1, -1), (1, -1, 0), (0, 1, -1))]
    for i, d in enumerate(dirs):
        d = d / np.linalg.norm(d)
        o = g.random(3) * 0.5 + 0.25
        cnt = int(g.integers(1, 90))                       # not a multiple of the chunk length
        st = step * (5.0 if i % 17 == 0 else 1.0)          # some rays jump several fine cells per step
        t = np.arange(cnt, dtype=np.float32) * st
        pts = o[None, :] + d[None, :] * t[:, None]
        if i % 11 == 0:
            pts[cnt // 2:] = pts[cnt // 2]                  # stuck samples (identical positions)
        xs.append(pts)
    x = np.clip(np.concatenate(xs), 0.0, 1.0).astype(np.float32)
    n = x.shape[0]
    dy = g.normal(size=(n, L * Fd)).astype(np.float32)
    dead = g.random(n) < 0.2
    dy[dead] = 0.0                                          # samples behind the early-termination point
    dy[:, 8:16][g.random(n) < 0.1] = 0.0                    # a level with zero gradient on some samples
    ref = oracle.grid_bwd_param(desc, x, dy, n_params)
    out = torch.zeros(n_params, device=DEV)
    gd = ngp._lib.GridDesc()
    assert ngp._lib.call_host("grid_layout", L, Fd, log2T, base, pls, gd) == n_params
    call("grid_bwd_param", gd, T(x), T(dy), L * Fd, n, out)
    scale = np.abs(ref).max()
    close(N(out), ref, 1e-4, 2e-5 * scale)
    # and the same through a wider gradient matrix (the field passes a column window)
    wide = np.zeros((n, L * Fd + 24), np.float32)
    wide[:, :L * Fd] = dy
    out2 = torch.zeros(n_params, device=DEV)
    call("grid_bwd_param", gd, T(x), T(wide), L * Fd + 24, n, out2)
    close(N(out2), ref, 1e-4, 2e-5 * scale)
    # per-sample row scale (the density head's d_sigma[s] * d(sigma)/d(features)[s]), incl. zero rows
    rs = g.normal(size=n).astype(np.float32)
    rs[g.random(n) < 0.15] = 0.0
    ref_s = oracle.grid_bwd_param(desc, x, dy * rs[:, None], n_params)
    out3 = torch.zeros(n_params, device=DEV)
    call("grid_bwd_param_scaled", gd, T(x), T(dy), L * Fd, T(rs), n, out3)
    close(N(out3), ref_s, 1e-4, 2e-5 * np.abs(ref_s).max())


@pytest.mark.parametrize("L,n", [(1, 1), (5, 15), (6, 17), (16, 16), (16, 63), (7, 64), (16, 65), (10, 1000), (3, 129)])
def test_grid_bwd_param_f8_edge_shapes(ngp, L, n):
    """The F = 8 scatter works in rounds of 16 samples x 4 levels per wave and chunks of 64 samples: level counts that
    are not multiples of four (idle level lanes), batches shorter than a round, ending inside a round or one sample
    behind a chunk, repeated positions (one line hit by consecutive samples), and hash collisions of a tiny table."""
    from ngp_amd._lib import call
    Fd, base, pls, log2T = 8, 4, 1.7, 9
    desc, n_params = oracle.grid_layout(L, Fd, log2T, base, pls)
    g = rng(900 + 31 * L + n)
    x = g.random((n, 3)).astype(np.float32)
    if n > 4:
        x[n // 2:n // 2 + 3] = x[n // 2]          # consecutive samples in one cell
        x[-1] = [1.0, 1.0, 1.0]                    # the far corner (index wrap path)
    dy = g.normal(size=(n, L * Fd)).astype(np.float32)
    if n > 8:
        dy[3:6] = 0.0
    ref = oracle.grid_bwd_param(desc, x, dy, n_params)
    gd = ngp._lib.GridDesc()
    assert ngp._lib.call_host("grid_layout", L, Fd, log2T, base, pls, gd) == n_params
    out = torch.zeros(n_params, device=DEV)
    call("grid_bwd_param", gd, T(x), T(dy), L * Fd, n, out)
    close(N(out), ref, 1e-4, 2e-5 * max(np.abs(ref).max(), 1e-6))
    # accumulate: a second launch into the same buffer doubles it
    call("grid_bwd_param", gd, T(x), T(dy), L * Fd, n, out)
    close(N(out), 2 * ref, 1e-4, 4e-5 * max(np.abs(ref).max(), 1e-6))
    # a gradient matrix whose row stride is not a multiple of four floats (no 16-byte loads: the narrower-row kernel)
    odd = np.zeros((n, L * Fd + 5), np.float32)
    odd[:, :L * Fd] = dy
    out2 = torch.zeros(n_params, device=DEV)
    call("grid_bwd_param", gd, T(x), T(odd), L * Fd + 5, n, out2)
    close(N(out2), ref, 1e-4, 2e-5 * max(np.abs(ref).max(), 1e-6))


def _ray_ordered_points(g, n_rays=320):
    """samples marched along rays as training feeds them (step sqrt(3)/1024): both directions along every axis,
    diagonals, coarse-step rays, stuck samples; segment lengths that are no multiple of the tile length"""
    step = np.float32(3 ** 0.5 / 1024)
    dirs = [g.normal(size=3) for _ in range(n_rays - 10)]
    dirs += [np.array(v, float) for v in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1),
                                          (1, 1, 1), (-1, -1, -1), (1, -1, 0), (0, 1, -1))]
    xs = []
    for i, d in enumerate(dirs):
        d = d / np.linalg.norm(d)
        o = g.random(3) * 0.5 + 0.25
        cnt = int(g.integers(1, 90))
        st = step * (5.0 if i % 17 == 0 else 1.0)
        t = np.arange(cnt, dtype=np.float32) * st
        pts = o[None, :] + d[None, :] * t[:, None]
        if i % 11 == 0:
            pts[cnt // 2:] = pts[cnt // 2]
        xs.append(pts)
    return np.clip(np.concatenate(xs), 0.0, 1.0).astype(np.float32)


@pytest.mark.parametrize("log2T", [14, 19])
def test_grid_gathers_ray_ordered(ngp, log2T):
    """The F = 8 gathers load each UNIQUE cell of a 32-sample tile once (runs of consecutive samples in one cell share
    the staged corners).  Random points never form runs; this feeds ray-ordered samples, through a plain and a
    column-window output / gradient matrix (the field's rgb_in[:, 16:])."""
    from ngp_amd._lib import call
    L, Fd, base, pls = 16, 8, 16, 1.3195079107728942
    desc, n_params = oracle.grid_layout(L, Fd, log2T, base, pls)
    g = rng(160 + log2T)
    x = _ray_ordered_points(g)
    n = x.shape[0]
    table = g.uniform(-1, 1, n_params).astype(np.float32)
    gd = ngp._lib.GridDesc()
    assert ngp._lib.call_host("grid_layout", L, Fd, log2T, base, pls, gd) == n_params
    ref = oracle.grid_fwd(desc, table, x)
    tt, xt = T(table), T(x)
    y = torch.full((n, L * Fd), 7.0, device=DEV)
    call("grid_fwd", gd, tt, xt, n, y, L * Fd)
    close(N(y), ref, 1e-5, 1e-6)
    wide = torch.full((n, 16 + L * Fd + 8), 7.0, device=DEV)       # column window: 64-byte offset, 608-byte rows
    call("grid_fwd", gd, tt, xt, n, wide[:, 16:], wide.shape[1])
    assert torch.equal(wide[:, 16:16 + L * Fd], y)
    assert bool((wide[:, :16] == 7.0).all()) and bool((wide[:, 16 + L * Fd:] == 7.0).all())
    dy = g.normal(size=(n, L * Fd)).astype(np.float32)
    dy[g.random(n) < 0.2] = 0.0
    rgx = oracle.grid_bwd_input(desc, table, x, dy)
    gx = torch.full((n, 3), 7.0, device=DEV)
    call("grid_bwd_input", gd, tt, xt, T(dy), L * Fd, n, gx)
    close(N(gx), rgx, 2e-4, 3e-6 * np.abs(rgx).max())
    dyw = torch.zeros(n, L * Fd + 24, device=DEV)
    dyw[:, 16:16 + L * Fd] = T(dy)
    gx2 = torch.zeros(n, 3, device=DEV)
    call("grid_bwd_input", gd, tt, xt, dyw[:, 16:], dyw.shape[1], n, gx2)
    assert torch.equal(gx2, gx)


@pytest.mark.parametrize("L,n", [(1, 1), (4, 31), (5, 15), (6, 33), (8, 32), (12, 70), (16, 1), (16, 31), (16, 32),
                                 (16, 33), (16, 65), (10, 1000), (3, 129)])
def test_grid_gathers_f8_edge_shapes(ngp, L, n):
    """Tiles of 32 samples x 4 levels per wave: level counts that are no multiple of four (idle level slots, and for
    the input gradient 3 waves per tile -> the item-per-(sample, level) kernel), batches shorter than a tile, ending
    one sample before / at / behind a tile boundary, repeated positions, the far corner, a tiny table (collisions)."""
    from ngp_amd._lib import call
    Fd, base, pls, log2T = 8, 4, 1.7, 9
    desc, n_params = oracle.grid_layout(L, Fd, log2T, base, pls)
    g = rng(950 + 31 * L + n)
    x = g.random((n, 3)).astype(np.float32)
    if n > 4:
        x[n // 2:n // 2 + 3] = x[n // 2]
        x[-1] = [1.0, 1.0, 1.0]
        x[0] = [0.0, 0.0, 0.0]
    table = g.uniform(-1, 1, n_params).astype(np.float32)
    gd = ngp._lib.GridDesc()
    assert ngp._lib.call_host("grid_layout", L, Fd, log2T, base, pls, gd) == n_params
    guard = 64
    ybuf = torch.full((n * L * Fd + 2 * guard,), 7.0, device=DEV)
    y = ybuf[guard:guard + n * L * Fd].view(n, L * Fd)
    call("grid_fwd", gd, T(table), T(x), n, y, L * Fd)
    close(N(y), oracle.grid_fwd(desc, table, x), 1e-5, 1e-6)
    assert bool((ybuf[:guard] == 7.0).all()) and bool((ybuf[-guard:] == 7.0).all())
    dy = g.normal(size=(n, L * Fd)).astype(np.float32)
    rgx = oracle.grid_bwd_input(desc, table, x, dy)
    gbuf = torch.full((n * 3 + 2 * guard,), 7.0, device=DEV)
    gx = gbuf[guard:guard + n * 3].view(n, 3)
    call("grid_bwd_input", gd, T(table), T(x), T(dy), L * Fd, n, gx)
    close(N(gx), rgx, 2e-4, 3e-6 * max(np.abs(rgx).max(), 1e-6))
    assert bool((gbuf[:guard] == 7.0).all()) and bool((gbuf[-guard:] == 7.0).all())


def test_grid_gathers_positions_outside_the_unit_cube(ngp):
    """The C ABI takes any position: outside [0, 1] the dense levels' index wraps (tcnn: idx % size) instead of leaving
    the level — the run-leader kernels' one-subtraction modulo falls back to a real one there."""
    from ngp_amd._lib import call
    L, Fd, base, pls, log2T = 16, 8, 16, 1.3195079107728942, 15
    desc, n_params = oracle.grid_layout(L, Fd, log2T, base, pls)
    g = rng(975)
    n = 257
    x = (g.random((n, 3)) * 8.0 - 3.7).astype(np.float32)          # [-3.7, 4.3]
    x[::5] = g.random((len(x[::5]), 3)).astype(np.float32)         # some inside, so that runs of both kinds meet in a tile
    table = g.uniform(-1, 1, n_params).astype(np.float32)
    gd = ngp._lib.GridDesc()
    assert ngp._lib.call_host("grid_layout", L, Fd, log2T, base, pls, gd) == n_params
    y = torch.empty(n, L * Fd, device=DEV)
    call("grid_fwd", gd, T(table), T(x), n, y, L * Fd)
    close(N(y), oracle.grid_fwd(desc, table, x), 1e-5, 1e-6)
    dy = g.normal(size=(n, L * Fd)).astype(np.float32)
    gx = torch.empty(n, 3, device=DEV)
    call("grid_bwd_input", gd, T(table), T(x), T(dy), L * Fd, n, gx)
    rgx = oracle.grid_bwd_input(desc, table, x, dy)
    close(N(gx), rgx, 2e-4, 3e-6 * np.abs(rgx).max())


def test_grid_bwd_param_nonfinite_gradient_stays_in_table(ngp):
    """A non-finite upstream gradient must reach only the rows its sample touches: the F = 8 scatter keeps a second
    running sum per corner slot whose line tag is INVALID when both x-corners share a 64-byte line; 0 * inf = NaN in
    that sum must not be flushed (the INVALID tag is not an address of the table)."""
    from ngp_amd._lib import call
    L, Fd, base, pls, log2T = 16, 8, 16, 1.3195079107728942, 14
    desc, n_params = oracle.grid_layout(L, Fd, log2T, base, pls)
    g = rng(970)
    x = _ray_ordered_points(g, n_rays=40)
    n = x.shape[0]
    dy = g.normal(size=(n, L * Fd)).astype(np.float32)
    bad_inf, bad_nan = n // 3, 2 * n // 3
    x[bad_inf] = [0.3137, 0.4242, 0.5151]                    # strictly inside cells: no exactly-zero corner weights
    x[bad_nan] = [0.6161, 0.2727, 0.7373]
    dy[bad_inf] = np.inf
    dy[bad_nan] = np.nan
    gd = ngp._lib.GridDesc()
    assert ngp._lib.call_host("grid_layout", L, Fd, log2T, base, pls, gd) == n_params
    guard = 1 << 16
    buf = torch.zeros(n_params + 2 * guard, device=DEV)
    out = buf[guard:guard + n_params]
    call("grid_bwd_param", gd, T(x), T(dy), L * Fd, n, out)
    torch.cuda.synchronize()
    assert bool((buf[:guard] == 0).all()) and bool((buf[-guard:] == 0).all())
    # rows touched by finite samples only are finite and equal the oracle's on the finite part of the batch
    fin = np.ones(n, bool)
    fin[[bad_inf, bad_nan]] = False
    touched_bad = oracle.grid_bwd_param(desc, x[~fin], np.ones((2, L * Fd), np.float32), n_params) != 0
    # the kernel accumulates per 64-byte line (two rows): 0 * NaN also reaches the other row of a touched line
    touched_bad = np.repeat(touched_bad.reshape(-1, 2 * Fd).any(1), 2 * Fd)
    ref = oracle.grid_bwd_param(desc, x[fin], dy[fin], n_params)
    got = N(out)
    assert np.isfinite(got[~touched_bad]).all()
    close(got[~touched_bad], ref[~touched_bad], 1e-4, 2e-5 * np.abs(ref).max())
    assert not np.isfinite(got[touched_bad]).all()          # the bad samples did poison their own rows


def test_grid_double_backward(ngp):
    L, Fd, log2T, base, pls, n = 8, 8, 15, 16, 1.5, 600
    tcnn = ngp.tinycudann
    enc = tcnn.Encoding(3, {"otype": "HashGrid", "n_levels": L, "n_features_per_level": Fd,
                            "log2_hashmap_size": log2T, "base_resolution": base, "per_level_scale": pls}).to(DEV)
    desc, n_params = oracle.grid_layout(L, Fd, log2T, base, pls)
    g = rng(130)
    table = g.uniform(-1, 1, n_params).astype(np.float32)
    x = g.random((n, 3)).astype(np.float32)
    dy = g.normal(size=(n, L * Fd)).astype(np.float32)
    v = g.normal(size=(n, 3)).astype(np.float32)
    with torch.no_grad():
        enc.params.copy_(T(table))
    xt = T(x).requires_grad_(True)
    dyt = T(dy).requires_grad_(True)
    y = enc(xt)
    (gx,) = torch.autograd.grad(y, xt, dyt, create_graph=True)
    d_dy, d_p = torch.autograd.grad(gx, [dyt, enc.params], T(v))
    rp, rdy = oracle.grid_bwd_bwd_input(desc, table, x, dy, v)
    close(N(d_dy), rdy, 2e-4, 3e-6 * np.abs(rdy).max())
    close(N(d_p), rp, 1e-4, 3e-6 * np.abs(rp).max() + 2e-4)


@pytest.mark.parametrize("degree", [1, 2, 3, 4])
def test_sh(ngp, degree):
    g = rng(140)
    d = g.normal(size=(1000, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True) + 1) / 2
    d = d.astype(np.float32)
    enc = ngp.tinycudann.Encoding(3, {"otype": "SphericalHarmonics", "degree": degree}).to(DEV)
    xt = T(d).requires_grad_(True)
    y = enc(xt)
    close(N(y), oracle.sh_fwd(d, degree), 1e-6, 1e-6)
    # input gradient against central differences of the oracle (fp64 step on fp32 function)
    dy = g.normal(size=(1000, degree * degree)).astype(np.float32)
    (gx,) = torch.autograd.grad(y, xt, T(dy))
    eps = 1e-3
    fd = np.zeros((1000, 3))
    for k in range(3):
        dp, dm = d.copy(), d.copy()
        dp[:, k] += eps
        dm[:, k] -= eps
        fd[:, k] = ((oracle.sh_fwd(dp, degree).astype(np.float64) - oracle.sh_fwd(dm, degree)) * dy).sum(1) / (2 * eps)
    close(N(gx), fd, 2e-2, 2e-2)


# ---------------------------------------------------------------------------- M1-M4
@pytest.mark.parametrize("n,n_in,n_out,act,bias", [
    (1000, 128, 128, "Softplus", True), (777, 144, 128, "ReLU", False), (513, 128, 32, "ReLU", False),
    (300, 32, 16, "None", False), (1000, 128, 1, "Softplus", True), (129, 128, 16, "Sigmoid", False),
    (64, 16, 64, "ReLU", False), (2000, 160, 128, "ReLU", False)])
def test_linear_layers(ngp, n, n_in, n_out, act, bias):
    g = rng(150 + n_in + n_out)
    x = g.normal(size=(n, n_in)).astype(np.float32)
    W = (g.normal(size=(n_out, n_in)) / np.sqrt(n_in)).astype(np.float32)
    b = g.normal(size=n_out).astype(np.float32) if bias else None
    code = oracle.ACT[act]
    call = ngp._lib.call
    y = torch.empty(n, n_out, device=DEV)
    z = torch.empty(n, n_out, device=DEV)
    call("linear_fwd", T(x), n_in, T(W), n_in, T(b) if bias else None, n, n_in, n_out, code, y, n_out, z)
    ref = oracle.linear_fwd(x, W, b, act)
    close(N(y), ref, 2e-5, 2e-5)
    x64, W64 = x.astype(np.float64), W.astype(np.float64)
    zref = x64 @ W64.T + (b if bias else 0)
    close(N(z), zref, 2e-5, 2e-5)
    dz = g.normal(size=(n, n_out)).astype(np.float32)
    dx = torch.empty(n, n_in, device=DEV)
    call("linear_bwd_input", T(dz), n_out, T(W), n_in, n, n_in, n_out, dx, n_in, 0)
    close(N(dx), dz.astype(np.float64) @ W64, 3e-5, 3e-5)
    dW = torch.zeros(n_out, n_in, device=DEV)
    db = torch.zeros(n_out, device=DEV)
    call("linear_bwd_weight", T(dz), n_out, T(x), n_in, n, n_in, n_out, dW, n_in, db)
    close(N(dW), dz.astype(np.float64).T @ x64, 1e-4, 1e-3)
    close(N(db), dz.astype(np.float64).sum(0), 1e-4, 1e-3)


def test_tcnn_network_matches_torch(ngp):
    tcnn = ngp.tinycudann
    net = tcnn.Network(144, 3, {"otype": "CutlassMLP", "activation": "ReLU", "output_activation": "Sigmoid",
                                "n_neurons": 128, "n_hidden_layers": 1}).to(DEV)
    assert net.params.numel() == 128 * 144 + 16 * 128   # 20,480
    x = torch.randn(1500, 144, device=DEV, requires_grad=True)
    y = net(x)
    W1 = net.layer_weight(0).double()
    W2 = net.layer_weight(1).double()
    ref = torch.sigmoid(torch.relu(x.double() @ W1.T) @ W2.T)[:, :3]
    close(N(y), N(ref), 2e-5, 2e-5)
    g = torch.randn_like(y)
    gx, gp = torch.autograd.grad(y, [x, net.params], g)
    rx, rp = torch.autograd.grad(ref, [x, net.params], g.double())
    close(N(gx), N(rx), 1e-4, 1e-4)
    close(N(gp), N(rp), 1e-4, 1e-3)
    # skybox-style network: 9 inputs padded to 16 with ones
    sky = tcnn.Network(9, 3, {"otype": "CutlassMLP", "activation": "ReLU", "output_activation": "Sigmoid",
                              "n_neurons": 32, "n_hidden_layers": 1}).to(DEV)
    xs = torch.rand(100, 9, device=DEV)
    xp = torch.cat([xs, torch.ones(100, 7, device=DEV)], 1).double()
    refs = torch.sigmoid(torch.relu(xp @ sky.layer_weight(0).double().T) @ sky.layer_weight(1).double().T)[:, :3]
    close(N(sky(xs)), N(refs), 2e-5, 2e-5)


def test_adam_matches_torch(ngp):
    n = 100003
    torch.manual_seed(0)
    p0 = torch.randn(n, device=DEV)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=1e-2, eps=1e-8)
    p = p0.clone()
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    for step in range(1, 6):
        gr = torch.randn(n, device=DEV)
        p_ref.grad = gr.clone()
        opt.step()
        gbuf = gr.clone()
        ngp._lib.call("adam_step", p, gbuf, m, v, n, 1e-2, 0.9, 0.999, 1e-8, 0.0, step, None, 1)
        assert not gbuf.any()
    close(N(p), N(p_ref), 1e-5, 1e-6)


@pytest.mark.parametrize("case", [
    # n, n_in, H, n_out, act1, act2, biases
    (1000, 128, 128, 1, 3, 3, True),     # xyz_net: softplus / softplus with biases
    (1531, 160, 128, 3, 1, 2, False),    # rgb_net with appearance codes: ReLU / sigmoid
    (777, 128, 32, 3, 1, 0, False),      # norm_pred_header
    (130, 144, 128, 4, 1, 0, False),     # 4 outputs, two row tiles, second one ragged
    (5, 16, 64, 2, 1, 4, True),          # H = 64 (half-empty column tile), exp output
    (900, 128, 32, 7, 1, 0, False),      # semantic_header: 7 classes (two passes of the epilogue)
    (333, 128, 128, 8, 1, 0, True),      # 8 outputs on the wide tile
])
def test_mlp2_fwd_fused(ngp, case):
    """ngp_mlp2_fwd (second layer applied in the MFMA kernel's epilogue) against fp64"""
    from ngp_amd._lib import call
    n, n_in, H, n_out, act1, act2, biases = case
    g = rng(320 + n)
    x = g.normal(size=(n, n_in)).astype(np.float32)
    W1 = (g.normal(size=(H, n_in)) / np.sqrt(n_in)).astype(np.float32)
    W2 = (g.normal(size=(n_out, H)) / np.sqrt(H)).astype(np.float32)
    b1 = g.normal(size=H).astype(np.float32) if biases else None
    b2 = g.normal(size=n_out).astype(np.float32) if biases else None

    def act(v, a):
        return {0: v, 1: np.maximum(v, 0), 2: 1 / (1 + np.exp(-v)), 3: np.logaddexp(0, v), 4: np.exp(v)}[a]
    z1 = x.astype(np.float64) @ W1.astype(np.float64).T + (0 if b1 is None else b1)
    h_ref = act(z1, act1)
    o_ref = act(h_ref @ W2.astype(np.float64).T + (0 if b2 is None else b2), act2)
    hidden = torch.full((n, H), 9.0, device=DEV)
    out = torch.full((n, n_out), 9.0, device=DEV)
    call("mlp2_fwd", T(x), n_in, T(W1), n_in, None if b1 is None else T(b1), act1, T(W2), H,
         None if b2 is None else T(b2), act2, n, n_in, H, n_out, hidden, H, out, n_out)
    close(N(hidden), h_ref, 2e-5, 2e-5)
    close(N(out), o_ref, 3e-5, 3e-5)
    # strided operands: input taken from a wider matrix, hidden written into a wider one
    wide = torch.zeros(n, n_in + 16, device=DEV)
    wide[:, 16:] = T(x)
    hid_w = torch.zeros(n, H + 8, device=DEV)
    call("mlp2_fwd", wide[:, 16:], n_in + 16, T(W1), n_in, None if b1 is None else T(b1), act1, T(W2), H,
         None if b2 is None else T(b2), act2, n, n_in, H, n_out, hid_w, H + 8, out, n_out)
    close(N(hid_w[:, :H]), h_ref, 2e-5, 2e-5)
    assert not hid_w[:, H:].any()
    close(N(out), o_ref, 3e-5, 3e-5)
    # ngp_mlp2_fwd_dact: the same launch also leaves act2'(z2) through the output — bitwise ngp_act_bwd(NULL, out)
    out2 = torch.full((n, n_out), 9.0, device=DEV)
    hid2 = torch.empty(n, H, device=DEV)
    dact = torch.full((n, n_out), 9.0, device=DEV)
    call("mlp2_fwd_dact", T(x), n_in, T(W1), n_in, None if b1 is None else T(b1), act1, T(W2), H,
         None if b2 is None else T(b2), act2, n, n_in, H, n_out, hid2, H, out2, n_out, dact)
    assert torch.equal(out2, out) and torch.equal(hid2, hid_w[:, :H].contiguous())
    want = torch.empty_like(out2)
    call("act_bwd", None, out2, out2.numel(), act2, want)
    assert torch.equal(dact, want)


def test_adam_width_entry_and_trainer_measurement(ngp):
    """ngp_adam_step_width: the result does not depend on the launch width; NGPTrainer's measurement of the two
    candidates ends with one of them and both window medians recorded"""
    n = 300007
    torch.manual_seed(3)
    p0, g0 = torch.randn(n, device=DEV), torch.randn(n, device=DEV)
    outs = []
    for width in (0, 256, 512, 7):
        p, g, m, v = p0.clone(), g0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        ngp._lib.call("adam_step_width", p, g, m, v, n, 1e-2, 0.9, 0.999, 1e-8, 0.0, 1, None, 1, width)
        assert not g.any()
        outs.append((p, m, v))
    for o in outs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(o, outs[0]))
    from ngp_amd.networks import NGP
    from ngp_amd.synthetic import LegoProxy
    from ngp_amd.trainer import NGPTrainer
    torch.manual_seed(20220806)
    model = NGP(scale=0.5).to(DEV)
    G = model.grid_size
    model.register_buffer("density_grid", torch.zeros(model.cascades, G ** 3, device=DEV))
    coords = torch.stack(torch.meshgrid(*[torch.arange(G, dtype=torch.int32, device=DEV)] * 3, indexing="ij"), -1)
    model.register_buffer("grid_coords", coords.reshape(-1, 3).contiguous())
    scene = LegoProxy(img_wh=(200, 200), device=DEV)
    tr = NGPTrainer(model, lr=1e-2)
    if tr.adam_width is not None:
        pytest.skip("NGP_ADAM_WIDTH fixes the width")
    tr.adam_tune = (16, 2)
    gen = torch.Generator(device=DEV).manual_seed(1)
    for i in range(16 + 2 * 2 * 16 + 8):
        img, pix = scene.sample_batch(1024, generator=gen)
        o, d = scene.rays(img, pix)
        gt, _ = scene.ground_truth(o, d, n_quad=64)
        loss, _ = tr.step(o, d, gt)
        if i % 16 == 15:
            torch.cuda.synchronize()
    tr.wait()
    torch.cuda.synchronize()
    assert tr.adam_width in tr.adam_candidates
    assert set(tr.adam_tune_ms) == set(tr.adam_candidates) and all(v > 0 for v in tr.adam_tune_ms.values())
    assert torch.isfinite(loss)


@pytest.mark.parametrize("n", [1, 3, 4, 1023, 100003, 4_000_001])
def test_sumsq_and_clip_coef(ngp, n):
    """global gradient norm + clip coefficient (torch.nn.utils.clip_grad_norm_ semantics), aligned and
    unaligned buffers, ragged tails"""
    torch.manual_seed(n)
    buf = torch.randn(n + 1, device=DEV)
    for x in (buf[:n], buf[1:]):                       # 16-byte aligned / misaligned start
        out = torch.zeros(2, device=DEV)
        ngp._lib.call("sumsq", x, n, out[0:1])
        ref = float((x.double() ** 2).sum())
        assert abs(float(out[0]) - ref) <= 1e-5 * ref + 1e-12
        for max_norm, extra in ((50.0, 1.0), (0.5 * ref ** 0.5, 1.0), (0.5 * ref ** 0.5, 0.125)):
            ngp._lib.call("clip_coef", out[0:1], float(max_norm), float(extra), out[1:2])
            norm = ref ** 0.5 * extra
            want = extra * min(1.0, max_norm / (norm + 1e-6))
            assert abs(float(out[1]) - want) <= 1e-5 * want


def test_streaming_mlp_kernels_ragged_sizes_and_strides(ngp):
    """Same kernels over a spread of batch sizes (one row, tile size +-1, fewer tiles than waves, a few tiles
    per wave) with padded leading dimensions, against fp64; and a row's result must not depend on the
    batch it sits in (bit-identical rows between a batch and a prefix of it)."""
    from ngp_amd._lib import call
    H = 128
    gen = torch.Generator(device=DEV).manual_seed(811)
    for n, n_in, n_out, act1 in ((1, 128, 1, 3), (31, 144, 3, 1), (33, 160, 2, 1), (2049, 128, 4, 3), (65537, 144, 3, 1),
                                 (70001, 128, 1, 3)):
        ldx, ldh, ldo = n_in + 16, H + 4, n_out + 1
        xbuf = torch.randn(n, ldx, device=DEV, generator=gen)
        x = xbuf[:, 16:]                                   # 64-byte offset, padded rows (the field's colour input looks like this)
        W1 = torch.randn(H, n_in, device=DEV, generator=gen) * 0.1
        W2 = torch.randn(n_out, H, device=DEV, generator=gen) * 0.1
        b1 = torch.randn(H, device=DEV, generator=gen) * 0.1
        hbuf = torch.full((n, ldh), float("nan"), device=DEV)
        obuf = torch.full((n, ldo), float("nan"), device=DEV)
        call("mlp2_fwd", x, ldx, W1, n_in, b1, act1, W2, H, None, 0, n, n_in, H, n_out, hbuf, ldh, obuf, ldo)
        f1 = torch.relu if act1 == 1 else torch.nn.functional.softplus
        h64 = f1(x.double() @ W1.double().T + b1.double())
        assert float((hbuf[:, :H].double() - h64).abs().max()) < 2e-5, n
        assert float((obuf[:, :n_out].double() - h64 @ W2.double().T).abs().max()) < 2e-5, n
        assert torch.isnan(hbuf[:, H:]).all() and torch.isnan(obuf[:, n_out:]).all()          # padding untouched
        if n > 40:                                          # a prefix of the batch gives the same bits for its rows
            m = n // 2 + 3
            h2 = torch.empty(m, ldh, device=DEV)
            o2 = torch.empty(m, ldo, device=DEV)
            call("mlp2_fwd", x[:m], ldx, W1, n_in, b1, act1, W2, H, None, 0, m, n_in, H, n_out, h2, ldh, o2, ldo)
            assert torch.equal(h2[:, :H], hbuf[:m, :H]) and torch.equal(o2[:, :n_out], obuf[:m, :n_out])
        hidden = hbuf[:, :H]
        dz2 = torch.randn(n, n_out, device=DEV, generator=gen)
        g64 = (hidden.double() > 0).double() if act1 == 1 else -torch.expm1(-hidden.double())
        dz1 = (dz2.double() @ W2.double()) * g64
        dxb = torch.full((n, 132), float("nan"), device=DEV)
        W1c = W1[:, n_in - 128:]
        call("mlp_bwd_input", dz2, n_out, W2, H, hidden, ldh, act1, W1c, n_in, n, 128, H, n_out, dxb, 132, 0)
        ref = dz1 @ W1c.double()
        assert float((dxb[:, :128].double() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max())), n
        assert torch.isnan(dxb[:, 128:]).all()
        dW1 = torch.zeros(H, n_in, device=DEV); dW2 = torch.zeros(n_out, H, device=DEV)
        db1 = torch.zeros(H, device=DEV); db2 = torch.zeros(n_out, device=DEV)
        call("mlp_bwd_weight", dz2, n_out, W2, H, hidden, ldh, act1, x, ldx, n, n_in, H, n_out, dW1, n_in, db1, dW2, H, db2)
        for got, want in ((dW1, dz1.T @ x.double()), (dW2, dz2.double().T @ hidden.double()), (db1, dz1.sum(0)),
                          (db2, dz2.double().sum(0))):
            assert float((got.double() - want).abs().max()) < 5e-6 * float(want.abs().max()) + 2e-6, (n, got.shape)


@pytest.mark.parametrize("case", [
    # n_in, n_out, act1 (hidden), biases
    (128, 1, 3, True),      # xyz_net
    (144, 3, 1, False),     # rgb_net
    (160, 7, 1, False),     # rgb_net with appearance codes, two epilogue passes
])
def test_streaming_mlp_kernels_many_tiles(ngp, case):
    """The streaming kernels of the 128-wide layers (mlp_stream_fwd / dgrad / wgrad) at a size where every
    wave walks several tiles — the rolling row prefetch with its hand-placed vmcnt waits, the register
    ring of the weight gradient, the clamped / zeroed tail of the last chunk — against fp64 on EVERY row."""
    from ngp_amd._lib import call
    n_in, n_out, act1, biases = case
    n, H = 150001, 128
    gen = torch.Generator(device=DEV).manual_seed(700 + n_in)
    x = torch.randn(n, n_in, device=DEV, generator=gen)
    W1 = torch.randn(H, n_in, device=DEV, generator=gen) * 0.1
    W2 = torch.randn(n_out, H, device=DEV, generator=gen) * 0.1
    b1 = torch.randn(H, device=DEV, generator=gen) * 0.1 if biases else None
    b2 = torch.randn(n_out, device=DEV, generator=gen) * 0.1 if biases else None
    f1 = (lambda v: torch.relu(v)) if act1 == 1 else (lambda v: torch.nn.functional.softplus(v))
    hidden = torch.full((n, H), float("nan"), device=DEV)
    out = torch.full((n, n_out), float("nan"), device=DEV)
    call("mlp2_fwd", x, n_in, W1, n_in, b1, act1, W2, H, b2, 0, n, n_in, H, n_out, hidden, H, out, n_out)
    z1 = x.double() @ W1.double().T + (0 if b1 is None else b1.double())
    h64 = f1(z1)
    o64 = h64 @ W2.double().T + (0 if b2 is None else b2.double())
    assert float((hidden.double() - h64).abs().max()) < 2e-5
    assert float((out.double() - o64).abs().max()) < 2e-5
    # backward products from the stored hidden activations
    n_o = min(n_out, 4)                               # the operand-transform kernels take up to 4 outputs
    dz2 = torch.randn(n, n_o, device=DEV, generator=gen)
    W2b = W2[:n_o].contiguous()
    g64 = (hidden.double() > 0).double() if act1 == 1 else -torch.expm1(-hidden.double())
    dz1 = (dz2.double() @ W2b.double()) * g64
    dx = torch.full((n, 128), float("nan"), device=DEV)
    W1c = W1[:, n_in - 128:]                          # gradient w.r.t. the last 128 input columns (ld = n_in), as the field asks
    call("mlp_bwd_input", dz2, n_o, W2b, H, hidden, H, act1, W1c, n_in, n, 128, H, n_o, dx, 128, 0)
    ref_dx = dz1 @ W1c.double()
    assert float((dx.double() - ref_dx).abs().max()) < 2e-5 * max(1.0, float(ref_dx.abs().max()))
    dW1 = torch.zeros(H, n_in, device=DEV)
    dW2 = torch.zeros(n_o, H, device=DEV)
    db1 = torch.zeros(H, device=DEV) if biases else None
    db2 = torch.zeros(n_o, device=DEV) if biases else None
    call("mlp_bwd_weight", dz2, n_o, W2b, H, hidden, H, act1, x, n_in, n, n_in, H, n_o, dW1, n_in, db1, dW2, H, db2)
    for got, ref in ((dW1, dz1.T @ x.double()), (dW2, dz2.double().T @ hidden.double())) + \
            (((db1, dz1.sum(0)), (db2, dz2.double().sum(0))) if biases else ()):
        assert float((got.double() - ref).abs().max()) < 5e-6 * float(ref.abs().max()) + 1e-6, (got.shape,)


@pytest.mark.parametrize("case", [
    # n, H, n_in, n_out, act1
    (1000, 128, 128, 1, 3),     # density head: softplus hidden, one output
    (1531, 128, 144, 3, 1),     # rgb_net: ReLU hidden, 144 (not a multiple of the tile) inputs
    (777, 32, 128, 3, 1),       # norm_pred_header
    (64, 128, 16, 4, 1),        # narrow input, fewer rows than one tile
])
def test_mlp_fused_first_layer_backward(ngp, case):
    """ngp_mlp_bwd_input / ngp_mlp_bwd_weight (dz1 formed inside the MFMA product) against fp64:
    dz1 = act1'(hidden) * (dz2 . W2), dx = dz1 . W1, dW1 = dz1^T . x, db1 = colsum(dz1)."""
    from ngp_amd._lib import call
    n, H, n_in, n_out, act1 = case
    g = rng(300 + n)
    z = g.normal(size=(n, H)).astype(np.float32)
    hidden = np.log1p(np.exp(z)).astype(np.float32) if act1 == 3 else np.maximum(z, 0).astype(np.float32)
    dz2 = g.normal(size=(n, n_out)).astype(np.float32)
    W2 = g.normal(size=(n_out, H)).astype(np.float32)
    W1 = g.normal(size=(H, n_in)).astype(np.float32)
    x = g.normal(size=(n, n_in)).astype(np.float32)
    d = (1 - np.exp(-hidden.astype(np.float64))) if act1 == 3 else (hidden > 0).astype(np.float64)
    dz1 = d * (dz2.astype(np.float64) @ W2.astype(np.float64))
    ref_dx, ref_dW, ref_db = dz1 @ W1.astype(np.float64), dz1.T @ x.astype(np.float64), dz1.sum(0)

    dx = torch.full((n, n_in), 7.0, device=DEV)
    call("mlp_bwd_input", T(dz2), n_out, T(W2), H, T(hidden), H, act1, T(W1), n_in, n, n_in, H, n_out, dx, n_in, 0)
    close(N(dx), ref_dx, 2e-5, 2e-5 * np.abs(ref_dx).max())
    call("mlp_bwd_input", T(dz2), n_out, T(W2), H, T(hidden), H, act1, T(W1), n_in, n, n_in, H, n_out, dx, n_in, 1)
    close(N(dx), 2 * ref_dx, 2e-5, 4e-5 * np.abs(ref_dx).max())      # accumulate
    dW = torch.zeros(H, n_in, device=DEV)
    db = torch.zeros(H, device=DEV)
    dW2 = torch.zeros(n_out, H, device=DEV)
    db2 = torch.zeros(n_out, device=DEV)
    call("mlp_bwd_weight", T(dz2), n_out, T(W2), H, T(hidden), H, act1, T(x), n_in, n, n_in, H, n_out, dW, n_in, db,
         dW2, H, db2)
    close(N(dW), ref_dW, 2e-5, 2e-5 * np.abs(ref_dW).max())
    close(N(db), ref_db, 2e-5, 2e-5 * np.abs(ref_db).max())
    # second-layer gradients from the same pass: dW2 = dz2^T . hidden, db2 = colsum(dz2)
    ref_dW2 = dz2.astype(np.float64).T @ hidden.astype(np.float64)
    close(N(dW2), ref_dW2, 2e-5, 2e-5 * np.abs(ref_dW2).max())
    close(N(db2), dz2.astype(np.float64).sum(0), 2e-5, 2e-5 * np.abs(dz2).sum(0).max())
    # plain route: ngp_mlp_hidden_bwd materialises dz1 and can leave dW2 / db2 as well
    d_out = g.normal(size=(n, n_out)).astype(np.float32)
    out = (1 / (1 + np.exp(-g.normal(size=(n, n_out))))).astype(np.float32)     # sigmoid outputs
    dz2p = d_out.astype(np.float64) * out * (1 - out)
    dz1_ref = d * (dz2p @ W2.astype(np.float64))
    dz1_t = torch.empty(n, H, device=DEV)
    dW2p = torch.zeros(n_out, H, device=DEV)
    db2p = torch.zeros(n_out, device=DEV)
    call("mlp_hidden_bwd", T(d_out), n_out, T(out), n_out, 2, T(W2), H, T(hidden), H, act1, n, H, n_out, None, 0,
         dz1_t, H, dW2p, H, db2p)
    close(N(dz1_t), dz1_ref, 2e-5, 2e-5 * np.abs(dz1_ref).max())
    ref = dz2p.T @ hidden.astype(np.float64)
    close(N(dW2p), ref, 3e-5, 3e-5 * np.abs(ref).max())
    close(N(db2p), dz2p.sum(0), 3e-5, 3e-5 * np.abs(dz2p).sum(0).max())
    # a column window of W1 / x (the trainer passes rgb_net's grid-feature columns only)
    if n_in >= 64:
        dxw = torch.empty(n, 32, device=DEV)
        call("mlp_bwd_input", T(dz2), n_out, T(W2), H, T(hidden), H, act1, T(W1)[:, 16:], n_in, n, 32, H, n_out,
             dxw, 32, 0)
        close(N(dxw), ref_dx[:, 16:48], 2e-5, 2e-5 * np.abs(ref_dx).max())


# ---------------------------------------------------------------------------- NGP field (fused node)
def _make_model(ngp, embed_a=False, table_scale=0.3, scale=0.5):
    torch.manual_seed(5)
    model = ngp.networks.NGP(scale=scale, embed_a=embed_a, embed_a_len=8).to(DEV)
    with torch.no_grad():  # tcnn's 1e-4 init makes every feature ~0: use O(1) tables for a real test
        model.xyz_encoder.params.uniform_(-table_scale, table_scale)
        model.rgb_encoder.params.uniform_(-table_scale, table_scale)
    return model


def test_field_forward_matches_oracle(ngp):
    from oracle.field import CpuNGP
    model = _make_model(ngp)
    g = rng(200)
    n = 1500
    x = ((g.random((n, 3)) - 0.5) * 0.98).astype(np.float32)
    d = g.normal(size=(n, 3)).astype(np.float32)
    with torch.no_grad():
        sig, rgb, n_raw, n_pred, sem = model(T(x), T(d))
        sig_t, rgb_t, n_pred_t, n_raw_t, sem_t = model.forward_test(T(x), T(d))
        dens = model.density(T(x))
    state = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    ref = CpuNGP(state, scale=0.5)
    rs, rrgb, rn_raw, rn_pred, rsem, _ = ref(x, d)
    close(N(sig), rs, 1e-4, 1e-5)
    close(N(dens), rs, 1e-4, 1e-5)
    close(N(rgb), rrgb, 1e-4, 1e-5)
    close(N(n_pred), rn_pred, 1e-3, 1e-4)
    close(N(sem), rsem, 1e-4, 1e-5)
    # unit normals from d(sigma)/dx: compare direction
    cos = (N(n_raw) * rn_raw).sum(-1)
    assert np.percentile(cos, 1) > 0.9999
    # test path returns the same values with the two normals swapped
    assert torch.equal(sig, sig_t) and torch.equal(rgb, rgb_t)
    assert torch.equal(n_pred, n_pred_t) and torch.equal(n_raw, n_raw_t) and torch.equal(sem, sem_t)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_field_matches_reference_ngp_golden(ngp, golden, tag):
    """The HIP field against the outputs of the reference's OWN models/networks.py::NGP (G6 fixture:
    forward, forward_test, density recorded on the CPU with a pure-torch tinycudann stand-in and
    the hash tables filled by helpers.table_rule).  a: scale 0.5; b: scale 8 + appearance codes."""
    from helpers import g6_state
    g = golden("g6_ngp_field.npz")
    scale = float(g[f"{tag}_scale"])
    embed = f"{tag}_embedding_a" in g.files
    model = ngp.networks.NGP(scale=scale, embed_a=embed, embed_a_len=8).to(DEV)
    state = g6_state(g, tag, model.xyz_encoder.params.numel(), model.rgb_encoder.params.numel())
    with torch.no_grad():
        for k, v in state.items():
            dict(model.named_parameters())[k].copy_(T(v))
    kw = {"embedding_a": T(g[f"{tag}_embedding_a"])} if embed else {}
    x, d = T(g[f"{tag}_x"]), T(g[f"{tag}_d"])
    with torch.no_grad():
        sig, rgb, n_raw, n_pred, sem = model(x, d, **kw)
        sig_t, rgb_t, n_pred_t, n_raw_t, sem_t = model.forward_test(x, d, **kw)
        dens = model.density(x)
    close(N(sig), g[f"{tag}_fwd_sigmas"], 2e-4, 1e-5)
    close(N(dens), g[f"{tag}_density"], 2e-4, 1e-5)
    close(N(rgb), g[f"{tag}_fwd_rgbs"], 2e-4, 1e-5)
    close(N(n_pred), g[f"{tag}_fwd_normals_pred"], 1e-3, 1e-4)
    close(N(sem), g[f"{tag}_fwd_semantic"], 2e-4, 1e-5)
    cos = (N(n_raw) * g[f"{tag}_fwd_normals_raw"]).sum(-1)
    assert np.percentile(cos, 2) > 0.9995
    close(N(sig_t), g[f"{tag}_test_sigmas"], 2e-4, 1e-5)
    close(N(rgb_t), g[f"{tag}_test_rgbs"], 2e-4, 1e-5)
    close(N(n_pred_t), g[f"{tag}_test_normals_pred"], 1e-3, 1e-4)
    close(N(sem_t), g[f"{tag}_test_semantic"], 2e-4, 1e-5)
    assert np.percentile((N(n_raw_t) * g[f"{tag}_test_normals_raw"]).sum(-1), 2) > 0.9995


def test_skybox_matches_reference_ngp_golden(ngp, golden):
    """forward_skybox (M4) against the reference's own NGP(use_skybox=True).forward_skybox (G6)"""
    g = golden("g6_ngp_field.npz")
    model = ngp.networks.NGP(scale=0.5, use_skybox=True).to(DEV)
    with torch.no_grad():
        model.skybox_rgb_net.params.copy_(T(g["sky_params"]))
        out = model.forward_skybox(T(g["sky_d"]))
    close(N(out), g["sky_rgb"], 2e-4, 1e-5)


@pytest.mark.parametrize("scale", [8.0, 16.0])
def test_field_forward_matches_oracle_unbounded(ngp, scale):
    """BASELINE configs 2/3 (Playground-like scale 8 with appearance codes, bicycle-like scale 16):
    per-level growth b = exp(ln(2048*scale/16)/15), positions normalised by the larger box."""
    from oracle.field import CpuNGP
    model = _make_model(ngp, embed_a=True, scale=scale)
    g = rng(201)
    n = 1200
    x = ((g.random((n, 3)) - 0.5) * 2 * scale * 0.98).astype(np.float32)
    x[: n // 2] *= 0.05          # half of the points near the origin, where the scene is
    d = g.normal(size=(n, 3)).astype(np.float32)
    emb = g.normal(size=(n, 8)).astype(np.float32)
    with torch.no_grad():
        sig, rgb, n_raw, n_pred, sem = model(T(x), T(d), embedding_a=T(emb))
    state = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    ref = CpuNGP(state, scale=scale)
    rs, rrgb, rn_raw, rn_pred, rsem, _ = ref(x, d, emb)
    close(N(sig), rs, 2e-4, 1e-5)
    close(N(rgb), rrgb, 2e-4, 1e-5)
    close(N(n_pred), rn_pred, 1e-3, 1e-4)
    close(N(sem), rsem, 2e-4, 1e-5)
    cos = (N(n_raw) * rn_raw).sum(-1)
    assert np.percentile(cos, 2) > 0.999


@pytest.mark.parametrize("cfg", [
    # scale, embed_a, exp_step_factor, rays, random_bg       (BASELINE configs 2 and 3, shapes only)
    (8.0, True, 1 / 256, 8192, True),
    (16.0, False, 1 / 256, 16384, False),
])
def test_trainer_unbounded_configs(ngp, cfg):
    """Playground-like (K=5 cascades, exponential stepping, appearance codes, random background) and
    bicycle-like (K=6, 16384 rays) configurations run the full schedule: finite, and the loss drops."""
    from ngp_amd.synthetic import LegoProxy
    from ngp_amd.trainer import NGPTrainer
    scale, embed_a, esf, n_rays, random_bg = cfg
    torch.manual_seed(31)
    model = ngp.networks.NGP(scale=scale, embed_a=embed_a, embed_a_len=8).to(DEV)
    assert model.cascades == (5 if scale == 8.0 else 6)
    G = model.grid_size
    model.register_buffer("density_grid", torch.zeros(model.cascades, G ** 3, device=DEV))
    coords = torch.stack(torch.meshgrid(*[torch.arange(G, dtype=torch.int32, device=DEV)] * 3, indexing="ij"), -1)
    model.register_buffer("grid_coords", coords.reshape(-1, 3).contiguous())
    scene = LegoProxy(n_images=20, img_wh=(200, 200), device=DEV)
    codes = torch.nn.Parameter(torch.zeros(20, 8, device=DEV)) if embed_a else None
    tr = NGPTrainer(model, lr=1e-2, exp_step_factor=esf)
    gen = torch.Generator(device=DEV).manual_seed(32)
    losses = []
    for i in range(24):
        img, pix = scene.sample_batch(n_rays, generator=gen)
        o, d = scene.rays(img, pix)
        gt, _ = scene.ground_truth(o, d, n_quad=64)
        tr.render_kwargs = {"random_bg": random_bg}
        if embed_a:
            tr.render_kwargs["embedding_a"] = codes[img]
        loss, res = tr.step(o, d, gt)
        losses.append(float(loss))
        assert res["rgb"].shape == (n_rays, 3) and int(res["total_samples"]) == res["xyzs"].shape[0]
    tr.wait()
    assert np.isfinite(losses).all()
    assert np.mean(losses[-4:]) < np.mean(losses[:4])
    assert torch.isfinite(tr.flat_param).all()
    if embed_a:
        assert codes.grad is not None and torch.isfinite(codes.grad).all() and codes.grad.abs().sum() > 0


def test_fused_tail_with_random_background_and_codes(ngp):
    """The one-launch render + loss tail on a Playground-like configuration (scale 8, exponential stepping, appearance
    codes, random background colour): every result, the loss terms and every gradient (field parameters, appearance
    codes) against the launch-per-operation route on the same samples and the same background draw."""
    from ngp_amd.losses import nerf_loss_and_grads, NeRFLoss
    from ngp_amd.rendering import render
    from ngp_amd.synthetic import LegoProxy
    torch.manual_seed(33)
    model = ngp.networks.NGP(scale=8.0, embed_a=True, embed_a_len=8).to(DEV)
    G = model.grid_size
    model.register_buffer("density_grid", torch.zeros(model.cascades, G ** 3, device=DEV))
    coords = torch.stack(torch.meshgrid(*[torch.arange(G, dtype=torch.int32, device=DEV)] * 3, indexing="ij"), -1)
    model.register_buffer("grid_coords", coords.reshape(-1, 3).contiguous())
    with torch.no_grad():
        model.xyz_net[2].bias.fill_(1.5)
    model.update_density_grid(0.01 * 1024 / 3 ** 0.5, warmup=True)
    scene = LegoProxy(n_images=6, img_wh=(100, 100), device=DEV)
    gen = torch.Generator(device=DEV).manual_seed(34)
    img, pix = scene.sample_batch(1500, generator=gen)
    o, d = scene.rays(img, pix)
    gt = torch.rand(1500, 3, device=DEV, generator=gen)
    codes = torch.nn.Parameter(torch.randn(6, 8, device=DEV) * 0.1)
    lam_o, lam_d = NeRFLoss().lambda_opa, NeRFLoss().lambda_distortion
    params = [p for p in model.parameters() if p.numel() > 0] + [codes]
    out = {}
    for fused in (False, True):
        for p in params:
            p.grad = None
        torch.manual_seed(35)                      # same marcher noise, same background colour
        kw = dict(exp_step_factor=1 / 256, num_classes=7, random_bg=True, embedding_a=codes[img])
        if fused:
            res = render(model, o, d, _fused_loss=(gt, lam_o, lam_d), **kw)
            assert "_loss_terms" in res
            terms = res.pop("_loss_terms")
            torch.autograd.backward([terms], [torch.tensor([1.0, 0, 0, 0], device=DEV)])
        else:
            res = render(model, o, d, **kw)
            terms, (d_rgb, d_op, d_ws) = nerf_loss_and_grads(res["rgb"], res["opacity"], res["ws"], res["deltas"], res["ts"],
                                                            res["rays_a"], gt, lam_o, lam_d)
            torch.autograd.backward([res["rgb"], res["opacity"], res["ws"]], [d_rgb, d_op, d_ws])
        out[fused] = (res, N(terms), [None if p.grad is None else N(p.grad).copy() for p in params])
    ra, ta, ga = out[False]
    rb, tb, gb = out[True]
    assert int(ra["total_samples"]) == int(rb["total_samples"]) > 0
    for k in ("opacity", "depth", "rgb", "normal_pred", "semantic", "ws", "Ro", "Rp"):
        close(N(rb[k]), N(ra[k]), 2e-5, 2e-6)
    close(tb, ta, 1e-4, 1e-9)
    for p, a, b in zip(params, ga, gb):
        if a is None:
            assert b is None or not b.any()
            continue
        scale = np.abs(a).max()
        assert np.abs(a - b).max() <= 3e-4 * scale + 1e-12, (tuple(p.shape), np.abs(a - b).max(), scale)
    assert np.abs(ga[-1]).sum() > 0                # the appearance codes did receive a gradient


@pytest.mark.parametrize("embed_a", [False, True])
def test_field_backward_matches_torch_fp64(ngp, embed_a):
    """The fused node's parameter gradients against an fp64 torch re-implementation of the same
    field built on the (separately verified) grid encoder."""
    model = _make_model(ngp, embed_a=embed_a)
    g = rng(210)
    n = 700
    x = T(((g.random((n, 3)) - 0.5) * 0.98).astype(np.float32))
    d = T(g.normal(size=(n, 3)).astype(np.float32))
    kw = {"embedding_a": T(g.normal(size=(n, 8)).astype(np.float32)).requires_grad_(True)} if embed_a else {}
    ws = [T(g.normal(size=s).astype(np.float32)) for s in ((n,), (n, 3), (n, 3), (n, 7))]
    params = [p for p in model.parameters() if p.numel() > 0] + ([kw["embedding_a"]] if embed_a else [])

    sig, rgb, _, n_pred, sem = model(x, d, **kw)
    loss = (sig * ws[0]).sum() + (rgb * ws[1]).sum() + (n_pred * ws[2]).sum() + (sem * ws[3]).sum()
    fused = torch.autograd.grad(loss, params, allow_unused=True)

    F_ = torch.nn.functional
    D = torch.float64
    xn = ((x - model.xyz_min) / (model.xyz_max - model.xyz_min)).contiguous()
    feat = model.xyz_encoder(xn).to(D)
    l1, l2 = model.xyz_net[0], model.xyz_net[2]
    a1 = F_.softplus(feat @ l1.weight.to(D).T + l1.bias.to(D))
    rsig = F_.softplus(a1 @ l2.weight.to(D).T + l2.bias.to(D))[:, 0]
    frgb = model.rgb_encoder(xn).to(D)
    sh = model.dir_encoder((F_.normalize(d, dim=-1) + 1) / 2).to(D)
    cols = [sh, frgb] + ([kw["embedding_a"].to(D)] if embed_a else [])
    inp = torch.cat(cols, 1)
    Kp = model.rgb_net.padded_in
    if inp.shape[1] < Kp:
        inp = torch.cat([inp, torch.ones(n, Kp - inp.shape[1], dtype=D, device=DEV)], 1)
    rrgb = torch.sigmoid(torch.relu(inp @ model.rgb_net.layer_weight(0).to(D).T) @ model.rgb_net.layer_weight(1).to(D).T)[:, :3]
    hn = torch.relu(frgb @ model.norm_pred_header.layer_weight(0).to(D).T) @ model.norm_pred_header.layer_weight(1).to(D).T
    rnp = -F_.normalize(hn[:, :3], dim=-1, eps=1e-6)
    hs = torch.relu(frgb @ model.semantic_header.layer_weight(0).to(D).T) @ model.semantic_header.layer_weight(1).to(D).T
    rsem = torch.softmax(hs[:, :7], -1)
    rloss = (rsig * ws[0]).sum() + (rrgb * ws[1]).sum() + (rnp * ws[2]).sum() + (rsem * ws[3]).sum()
    ref = torch.autograd.grad(rloss, params, allow_unused=True)
    close(N(sig), N(rsig), 1e-4, 1e-5)
    close(N(rgb), N(rrgb), 1e-4, 1e-5)
    names = [n_ for n_, p in model.named_parameters() if p.numel() > 0] + (["embedding_a"] if embed_a else [])
    for name, a, b in zip(names, fused, ref):
        assert (a is None) == (b is None), name
        if a is None:
            continue
        a, b = N(a).astype(np.float64), N(b).astype(np.float64)
        scale = np.abs(b).max() + 1e-12
        assert np.abs(a - b).max() <= 2e-4 * scale, (name, np.abs(a - b).max(), scale)


def test_field_skips_unused_heads(ngp):
    """No gradient on the normal / semantic outputs -> their headers receive no gradient at all
    (the reference computes all-zero gradients for them)."""
    model = _make_model(ngp)
    g = rng(220)
    x = T(((g.random((300, 3)) - 0.5) * 0.9).astype(np.float32))
    d = T(g.normal(size=(300, 3)).astype(np.float32))
    sig, rgb, *_ = model(x, d)
    (sig.sum() + rgb.sum()).backward()
    assert model.norm_pred_header.params.grad is None and model.semantic_header.params.grad is None
    assert model.rgb_net.params.grad.abs().sum() > 0 and model.xyz_encoder.params.grad.abs().sum() > 0


# ---------------------------------------------------------------------------- trainer
def test_trainer_steps_reduce_loss(ngp):
    """A few steps of the full schedule (grid update, render, NeRFLoss, backward, clip + fused Adam
    on the flat parameter buffer) on the synthetic scene lower the loss and keep parameters finite."""
    from ngp_amd.synthetic import LegoProxy
    from ngp_amd.trainer import NGPTrainer
    torch.manual_seed(3)
    model = ngp.networks.NGP(scale=0.5).to(DEV)
    G = model.grid_size
    model.register_buffer("density_grid", torch.zeros(model.cascades, G ** 3, device=DEV))
    coords = torch.stack(torch.meshgrid(*[torch.arange(G, dtype=torch.int32, device=DEV)] * 3, indexing="ij"), -1)
    model.register_buffer("grid_coords", coords.reshape(-1, 3).contiguous())
    scene = LegoProxy(n_images=20, img_wh=(200, 200), device=DEV)
    tr = NGPTrainer(model, lr=1e-2)
    gen = torch.Generator(device=DEV).manual_seed(4)
    losses = []
    for i in range(40):
        img, pix = scene.sample_batch(1024, generator=gen)
        o, d = scene.rays(img, pix)
        gt, _ = scene.ground_truth(o, d, n_quad=128)
        loss, res = tr.step(o, d, gt)
        losses.append(float(loss))
    tr.wait()
    assert np.isfinite(losses).all()
    assert np.mean(losses[-5:]) < 0.5 * np.mean(losses[:5])
    assert torch.isfinite(tr.flat_param).all()
    assert not tr.flat_grad.any()              # the fused Adam zeroes the gradient buffer
    assert tr.global_step == 40
    # parameters are views of the flat buffer (state-dict keys stay the reference's)
    assert model.rgb_encoder.params.data_ptr() == tr.flat_param.data_ptr()


def test_trainer_march_ahead_matches_inline(ngp):
    """Marching batch k+1 on the side stream under step k (MarchAhead) gives the same training run
    as marching every batch inside its own step: identical sample counts (the marcher is
    deterministic and sees the same bitfield and noise), losses equal up to atomic summation order.
    Crosses two density-grid updates (steps 0 and 16), where nothing is marched ahead."""
    from ngp_amd.synthetic import LegoProxy
    from ngp_amd.trainer import NGPTrainer
    scene = LegoProxy(n_images=20, img_wh=(200, 200), device=DEV)
    gen = torch.Generator(device=DEV).manual_seed(11)
    batches = []
    for i in range(20):
        img, pix = scene.sample_batch(1024, generator=gen)
        o, d = scene.rays(img, pix)
        gt, _ = scene.ground_truth(o, d, n_quad=64)
        batches.append((o, d, gt))
    runs = []
    for ahead in (False, True):
        torch.manual_seed(5)
        model = ngp.networks.NGP(scale=0.5).to(DEV)
        G = model.grid_size
        model.register_buffer("density_grid", torch.zeros(model.cascades, G ** 3, device=DEV))
        coords = torch.stack(torch.meshgrid(*[torch.arange(G, dtype=torch.int32, device=DEV)] * 3, indexing="ij"), -1)
        model.register_buffer("grid_coords", coords.reshape(-1, 3).contiguous())
        tr = NGPTrainer(model, lr=1e-2)
        torch.manual_seed(6)
        losses, counts = [], []
        for i, (o, d, gt) in enumerate(batches):
            nxt = batches[i + 1][:2] if (ahead and i + 1 < len(batches)) else None
            loss, res = tr.step(o, d, gt, next_rays=nxt)
            losses.append(float(loss))
            counts.append(int(res["total_samples"]))
        tr.wait()
        runs.append((losses, counts))
    (l0, c0), (l1, c1) = runs
    assert c0[:16] == c1[:16]           # until the first re-thresholded grid update the bitfields agree exactly
    close(np.array(l1), np.array(l0), 2e-2, 1e-6)
    assert abs(c0[-1] - c1[-1]) <= 0.02 * c0[-1]


def test_dense_render_matches_nocuda_path_psnr(ngp):
    """north-star acceptance: on identical rays (and identical sample depths) the HIP field +
    compositor reproduce the reference's rendering_noCUDA path (oracle restatement: CPU hash grid,
    CPU MLPs, raw2outputs) — PSNR against the scene's ground truth differs by < 0.05 dB, and the two
    images agree to > 60 dB.  The model is trained briefly first so that the images are not noise."""
    from ngp_amd.synthetic import LegoProxy
    from ngp_amd.trainer import NGPTrainer
    from ngp_amd.rendering import render_dense
    from oracle import nocuda
    from oracle.field import CpuNGP
    torch.manual_seed(21)
    model = ngp.networks.NGP(scale=0.5).to(DEV)
    G = model.grid_size
    model.register_buffer("density_grid", torch.zeros(model.cascades, G ** 3, device=DEV))
    coords = torch.stack(torch.meshgrid(*[torch.arange(G, dtype=torch.int32, device=DEV)] * 3, indexing="ij"), -1)
    model.register_buffer("grid_coords", coords.reshape(-1, 3).contiguous())
    scene = LegoProxy(n_images=30, img_wh=(200, 200), device=DEV)
    tr = NGPTrainer(model, lr=1e-2)
    gen = torch.Generator(device=DEV).manual_seed(22)
    for i in range(150):
        img, pix = scene.sample_batch(2048, generator=gen)
        o, d = scene.rays(img, pix)
        gt, _ = scene.ground_truth(o, d, n_quad=128)
        tr.step(o, d, gt)
    tr.wait()
    img, pix = scene.sample_batch(1024, generator=gen)     # config 0: 200x200 crop, 1024 rays
    o, d = scene.rays(img, pix)
    gt, _ = scene.ground_truth(o, d, n_quad=512)
    state = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()
             if k.endswith("params") or k.startswith("xyz_net")}
    field = CpuNGP(state, scale=0.5)
    for S in (64, 128):
        cpu = nocuda.render([field, field], N(o), N(d), [S])
        gpu = render_dense(model, o, d, T(cpu["z_vals0"]))
        # rays that miss the scene box have near == far and hit 0/0 in the depth warp of
        # rendering_noCUDA.py:146 (in the reference too): compare the rays that enter the box
        hit = np.isfinite(cpu["rgb0"]).all(-1)
        assert hit.sum() > 300
        rgb_cpu, rgb_gpu, gt_h = cpu["rgb0"][hit], N(gpu["rgb"])[hit], N(gt)[hit]

        def psnr(a, b):
            return -10 * np.log10(np.mean((a - b) ** 2))
        p_cpu, p_gpu = psnr(rgb_cpu, gt_h), psnr(rgb_gpu, gt_h)
        assert p_cpu > 15, p_cpu                               # a real image, not noise
        assert abs(p_cpu - p_gpu) < 0.05, (S, p_cpu, p_gpu)
        assert psnr(rgb_gpu, rgb_cpu) > 60, (S, psnr(rgb_gpu, rgb_cpu))
        close(N(gpu["opacity"])[hit], cpu["opacity0"][hit], 1e-3, 1e-4)
        close(N(gpu["depth"])[hit], cpu["depth0"][hit], 1e-3, 1e-3)


def test_train_from_dataset_directory(ngp, tmp_path):
    """NeRF-Synthetic directory -> loader -> the reference's schedule -> test
    split PSNR, end to end (the scene is the analytic proxy exported in Blender format; the
    renderer adds a black background for synthetic scenes, so the export is RGB on black)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import train_dataset as td
    from ngp_amd.datasets import NeRFDataset, write_synthetic_dataset
    from ngp_amd.synthetic import LegoProxy
    scene = LegoProxy(n_images=26, img_wh=(80, 80), device=DEV)              # int(800 * 0.1)
    root = write_synthetic_dataset(str(tmp_path), scene, n_train=24, n_test=2, rgba=False, n_quad=128)
    train_set = NeRFDataset(root, "train", 0.1, device=DEV)
    test_set = NeRFDataset(root, "test", 0.1, device=DEV)
    assert train_set.rays.is_cuda and train_set.rays.shape == (24, 6400, 3)
    torch.manual_seed(41)
    model = td.build_model(0.5, DEV)
    tr = td.train(model, train_set, num_epochs=2, steps_per_epoch=200, batch_size=2048, lr=1e-2)
    assert tr.global_step == 400
    psnrs = td.evaluate(model, test_set)
    assert len(psnrs) == 2 and min(psnrs) > 22.0, psnrs


@pytest.mark.parametrize("fmt", ["colmap", "tnt"])
def test_train_from_other_dataset_formats(ngp, tmp_path, fmt):
    """SURVEY §8(f) rank 4 loaders feeding the hot path: the analytic proxy exported as a COLMAP model
    (binary sparse/0 + images, every 8th frame held out) or as a Tanks-and-Temples directory
    (pose/*.txt, intrinsics.txt, 0_/1_ prefixes), loaded onto the GPU, trained and evaluated.  Both
    loaders rescale the cameras (by the largest camera distance), so the scene is learnt at a
    different size than in the Blender-format test — the pipeline must not care."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import train_dataset as td
    from ngp_amd.datasets import dataset_dict, export
    from ngp_amd.synthetic import LegoProxy
    scene = LegoProxy(n_images=34, img_wh=(80, 80), device=DEV)
    images = export.render_scene_views(scene, range(34), rgba=False, n_quad=128)
    c2w = scene.poses.cpu().numpy().astype(np.float64)
    K = scene.K.cpu().numpy().astype(np.float64)
    root = str(tmp_path / "scene")
    if fmt == "colmap":
        export.export_colmap(root, images, c2w, K, shuffle_seed=3)
    else:
        export.export_tnt(root, images, c2w, K, [1 if i % 8 == 0 else 0 for i in range(34)])
    n_train, n_test = 29, 5             # frames 0, 8, 16, 24, 32 are held out in both layouts
    train_set = dataset_dict[fmt](root, "train", 1.0, device=DEV)
    test_set = dataset_dict[fmt](root, "test", 1.0, device=DEV)
    assert train_set.rays.is_cuda and train_set.rays.shape == (n_train, 80 * 80, 3) and len(test_set) == n_test
    assert float(train_set.poses[:, :, 3].norm(dim=-1).max()) <= 1.0 + 1e-5      # cameras were rescaled
    torch.manual_seed(43)
    model = td.build_model(0.5, DEV)
    tr = td.train(model, train_set, num_epochs=3, steps_per_epoch=200, batch_size=2048, lr=1e-2)
    assert tr.global_step == 600
    psnrs = td.evaluate(model, test_set)
    # held-out views of a 29-view, 80 x 80 capture: a wrong pose / intrinsics convention gives ~10 dB
    assert len(psnrs) == n_test and sum(psnrs) / n_test > 20.0 and min(psnrs) > 16.0, psnrs


def test_trainer_fused_loss_path_matches_module_path(ngp):
    """NGPTrainer with the fused loss kernels (directly seeded backward) follows the same
    trajectory as with the reference's NeRFLoss module + loss.backward()."""
    from ngp_amd.synthetic import LegoProxy
    from ngp_amd.trainer import NGPTrainer
    scene = LegoProxy(n_images=10, img_wh=(100, 100), device=DEV)
    gen = torch.Generator(device=DEV).manual_seed(51)
    batches = []
    for i in range(6):
        img, pix = scene.sample_batch(1024, generator=gen)
        o, d = scene.rays(img, pix)
        gt, _ = scene.ground_truth(o, d, n_quad=64)
        batches.append((o, d, gt))
    out = []
    for fused in (True, False):
        torch.manual_seed(52)
        model = ngp.networks.NGP(scale=0.5).to(DEV)
        G = model.grid_size
        model.register_buffer("density_grid", torch.zeros(model.cascades, G ** 3, device=DEV))
        coords = torch.stack(torch.meshgrid(*[torch.arange(G, dtype=torch.int32, device=DEV)] * 3, indexing="ij"), -1)
        model.register_buffer("grid_coords", coords.reshape(-1, 3).contiguous())
        tr = NGPTrainer(model, lr=1e-2)
        tr.fused_loss = fused
        torch.manual_seed(53)
        losses = [float(tr.step(o, d, gt)[0]) for o, d, gt in batches]
        tr.wait()
        out.append((losses, N(model.xyz_net[0].weight).copy(), N(model.rgb_net.params).copy()))
    close(np.array(out[0][0]), np.array(out[1][0]), 1e-3, 1e-7)
    close(out[0][1], out[1][1], 5e-3, 5e-5)
    close(out[0][2], out[1][2], 5e-3, 5e-5)


def test_test_time_render_fast_loop_identical(ngp):
    """render(test_time=True) runs the field on the marcher's padded blocks instead of compacting
    with a mask and scattering back; every per-ray output must be bit-identical to the literal
    reference loop (same rounds, same samples, row-independent field)."""
    from ngp_amd.synthetic import LegoProxy
    from ngp_amd.trainer import NGPTrainer
    from ngp_amd.rendering import render
    torch.manual_seed(61)
    model = ngp.networks.NGP(scale=0.5).to(DEV)
    G = model.grid_size
    model.register_buffer("density_grid", torch.zeros(model.cascades, G ** 3, device=DEV))
    coords = torch.stack(torch.meshgrid(*[torch.arange(G, dtype=torch.int32, device=DEV)] * 3, indexing="ij"), -1)
    model.register_buffer("grid_coords", coords.reshape(-1, 3).contiguous())
    scene = LegoProxy(n_images=12, img_wh=(120, 120), device=DEV)
    tr = NGPTrainer(model, lr=1e-2)
    gen = torch.Generator(device=DEV).manual_seed(62)
    for i in range(300):
        img, pix = scene.sample_batch(2048, generator=gen)
        o, d = scene.rays(img, pix)
        gt, _ = scene.ground_truth(o, d, n_quad=64)
        tr.step(o, d, gt)
    tr.wait()
    n = 120 * 120
    o, d = scene.rays(torch.full((n,), 3, dtype=torch.long, device=DEV), torch.arange(n, device=DEV))
    # device_rounds = loop head, alive compaction and sample count on the device, no host round trip per round; default =
    # the host-driven fast loop (one sync per round); reference_test_loop = the literal loop.
    # max_samples = 24 ends the loop on the `samples < max_samples` head with rays still alive; exp_step_factor > 0
    # takes the 4-samples-per-round minimum.
    for kw in (dict(T_threshold=1e-2), dict(T_threshold=1e-4), dict(T_threshold=1e-4, max_samples=24),
               dict(T_threshold=1e-3, exp_step_factor=1 / 256)):
        fast = render(model, o, d, test_time=True, device_rounds=True, **kw)
        host = render(model, o, d, test_time=True, **kw)
        ref = render(model, o, d, test_time=True, reference_test_loop=True, **kw)
        assert int(fast["total_samples"]) == int(ref["total_samples"]) == int(host["total_samples"]) > 0, kw
        for k in ("opacity", "depth", "rgb", "normal_pred", "normal_raw", "semantic", "points"):
            assert torch.equal(fast[k], ref[k]), (k, kw)
            assert torch.equal(host[k], ref[k]), (k, kw)
        if "max_samples" not in kw:
            assert float(fast["opacity"].max()) > 0.9


def test_render_with_no_samples(ngp):
    """Edge case of the path: every cell empty -> the marcher returns zero samples; the train and
    test renders still return well-formed (all-background) results and backward is a no-op."""
    from ngp_amd.rendering import render
    model = _make_model(ngp)
    model.density_bitfield.zero_()
    o, d = make_rays(300, scale=1.0, seed=71)
    o, d = T(o), T(d)
    res = render(model, o, d, exp_step_factor=0.0)
    assert int(res["total_samples"]) == 0 and res["xyzs"].shape == (0, 3)
    assert res["rgb"].shape == (300, 3) and not res["rgb"].any() and not res["opacity"].any()
    loss = ((res["rgb"] - 0.5) ** 2).mean() + res["opacity"].mean()
    loss.backward()                       # nothing to propagate into: must not fault
    for p in model.parameters():
        assert p.grad is None or not p.grad.any()
    test = render(model, o, d, test_time=True, exp_step_factor=0.0)
    assert int(test["total_samples"]) == 0 and not test["rgb"].any()
    ref = render(model, o, d, test_time=True, exp_step_factor=0.0, reference_test_loop=True)
    assert torch.equal(test["rgb"], ref["rgb"]) and torch.equal(test["opacity"], ref["opacity"])
    # a whole trainer step on such a batch (fused loss kernels, clip + Adam): finite loss, no fault
    from ngp_amd.trainer import NGPTrainer
    tr = NGPTrainer(model, lr=1e-2)
    tr.global_step = 1                    # no occupancy update: the grid stays empty
    loss, res = tr.step(o, d, torch.rand(300, 3, device=DEV))
    tr.wait()
    torch.cuda.synchronize()
    assert torch.isfinite(loss) and int(res["total_samples"]) == 0


def test_render_matches_reference_render_golden(ngp, golden, monkeypatch):
    """render() — train path and test path — against the G7 fixture: the reference's OWN
    models/rendering.py::render run on the CPU (field = the reference's NGP class on the pure-torch
    tinycudann stand-in, vren = the C oracle).  Same rays, same occupancy bitfield, same marcher noise."""
    from helpers import table_rule
    from ngp_amd.rendering import render
    g = golden("g7_render_paths.npz")
    model = ngp.networks.NGP(scale=0.5).to(DEV)
    with torch.no_grad():
        model.xyz_encoder.params.copy_(T(table_rule(model.xyz_encoder.params.numel())))
        model.rgb_encoder.params.copy_(T(table_rule(model.rgb_encoder.params.numel())))
        named = dict(model.named_parameters())
        for k in ("xyz_net.0.weight", "xyz_net.0.bias", "xyz_net.2.weight", "xyz_net.2.bias", "rgb_net.params",
                  "norm_pred_header.params", "semantic_header.params"):
            named[k].copy_(T(g[k]))
        model.density_bitfield.copy_(T(g["density_bitfield"]))
    o, d = T(g["rays_o"]), T(g["rays_d"])
    noise = T(g["noise"])
    monkeypatch.setattr(torch, "rand_like", lambda t, *a, **k: noise.clone())   # the marcher's jitter draw
    with torch.no_grad():
        res = render(model, o, d, exp_step_factor=0.0, num_classes=7)
    monkeypatch.undo()
    # marching: exact
    assert int(res["total_samples"]) == int(g["train_total_samples"])
    assert np.array_equal(N(res["rays_a"]), g["train_rays_a"])
    assert np.array_equal(N(res["ts"]), g["train_ts"]) and np.array_equal(N(res["deltas"]), g["train_deltas"])
    assert np.array_equal(N(res["xyzs"]), g["train_xyzs"])
    close(N(res["sigma"]), g["train_sigma"], 2e-4, 1e-5)
    # compositing: rays whose transmittance passes within 1e-4 (relative) of the threshold may stop one
    # sample earlier / later under the parallel scan; everything else to fp32 tolerance
    bad = borderline_rays(g["train_sigma"], g["train_deltas"], g["train_rays_a"], 1e-4)
    keep = ~bad
    assert keep.sum() >= 0.9 * len(keep)
    for k, tol in (("opacity", 2e-5), ("depth", 1e-4), ("rgb", 1e-4), ("normal_pred", 1e-3), ("semantic", 1e-4),
                   ("Ro", 2e-3), ("Rp", 2e-3)):
        close(N(res[k])[keep], g["train_" + k][keep], 2e-3 if k in ("Ro", "Rp") else 5e-4, tol)
    smask = np.zeros(len(g["train_ws"]), bool)
    for r, st, c in g["train_rays_a"]:
        if keep[r]:
            smask[st:st + c] = True
    close(N(res["ws"])[smask], g["train_ws"][smask], 5e-4, 1e-6)
    assert abs(int(res["vr_samples"]) - int(g["train_vr_samples"])) <= int(bad.sum()) + 1
    # test-time path
    with torch.no_grad():
        for ref_loop in (False, True):
            tst = render(model, o, d, test_time=True, exp_step_factor=0.0, num_classes=7, T_threshold=1e-2,
                         reference_test_loop=ref_loop)
            assert abs(int(tst["total_samples"]) - int(g["test_total_samples"])) <= 0.01 * int(g["test_total_samples"])
            close(N(tst["opacity"]), g["test_opacity"], 2e-3, 2e-4)
            hit = g["test_opacity"] > 0.5
            close(N(tst["depth"])[hit], g["test_depth"][hit], 2e-3, 2e-4)
            close(N(tst["rgb"]), g["test_rgb"], 2e-3, 3e-4)
            close(N(tst["points"])[hit], g["test_points"][hit], 2e-3, 3e-4)
            assert ((N(tst["normal_pred"]) * g["test_normal_pred"]).sum(-1)[hit] > 0.999).all()
            assert np.percentile((N(tst["normal_raw"]) * g["test_normal_raw"]).sum(-1)[hit], 5) > 0.999
            assert (N(tst["semantic"])[hit] == g["test_semantic"][hit]).mean() > 0.97


# Whole-step gradient bars, as max |mine - ref| / max |ref| per tensor.  Measured on MI355X (round 2): G8 <= 5.5e-5
# (rgb_net; every other tensor <= 7.5e-6), G10 <= 3.5e-6 — fp32 summation order over ~3,000 samples (MFMA tiles,
# atomics, parallel scans vs the CPU fixture's sequential sums) and __expf in the compositor; the bars leave a
# factor ~5 (round 1 had 3e-3 / 5e-3 without having measured them).
G8_BAR, G10_BAR = 3e-4, 5e-5


@pytest.mark.parametrize("compact", [False, True, "fused_tail"])
def test_training_step_matches_reference_golden(ngp, golden, monkeypatch, compact):
    """(compact: the same step with the colour branch on the live samples only, model.compact_dead_samples;
    "fused_tail": normals, softmax, compositor, RefLoss, distortion, loss and the compositor's backward as ONE launch,
    ngp_render_loss_fused — the trainer's default route — with every per-ray result compared against the
    launch-per-operation route on the same samples.)
    One whole training step against the G8 fixture — the reference's OWN render() -> NeRFLoss
    (losses.py) -> sum of term means -> backward through its autograd Functions -> clip_grad_norm_(50)
    -> torch.optim.Adam(lr=1e-2, eps=1e-8).step(), run on the CPU (tinycudann = pure-torch stand-in,
    vren = C oracle).  Here: render() -> fused loss kernels -> backward -> NGPTrainer.optimizer_step()."""
    from helpers import table_rule
    from ngp_amd.losses import nerf_loss_and_grads
    from ngp_amd.rendering import render
    from ngp_amd.trainer import NGPTrainer
    g = golden("g8_train_step.npz")
    small = ("xyz_net.0.weight", "xyz_net.0.bias", "xyz_net.2.weight", "xyz_net.2.bias", "rgb_net.params",
             "norm_pred_header.params", "semantic_header.params")
    tables = ("xyz_encoder.params", "rgb_encoder.params")
    model = ngp.networks.NGP(scale=0.5).to(DEV)
    with torch.no_grad():
        model.xyz_encoder.params.copy_(T(table_rule(model.xyz_encoder.params.numel())))
        model.rgb_encoder.params.copy_(T(table_rule(model.rgb_encoder.params.numel())))
        named = dict(model.named_parameters())
        for k in small:
            named[k].copy_(T(g[k]))
        model.density_bitfield.copy_(T(g["density_bitfield"]))
    tr = NGPTrainer(model, lr=1e-2)                 # flat parameter / gradient / Adam-state buffers
    fused_tail = compact == "fused_tail"
    model.compact_dead_samples = bool(compact) and not fused_tail
    named = dict(model.named_parameters())
    o, d, gt = T(g["rays_o"]), T(g["rays_d"]), T(g["rgb_gt"])
    noise = T(g["noise"])
    monkeypatch.setattr(torch, "rand_like", lambda t, *a, **k: noise.clone())
    if fused_tail:
        with torch.no_grad():
            ref_res = render(model, o, d, exp_step_factor=0.0, num_classes=7)
        res = render(model, o, d, exp_step_factor=0.0, num_classes=7,
                     _fused_loss=(gt, tr.loss_fn.lambda_opa, tr.loss_fn.lambda_distortion))
    else:
        res = render(model, o, d, exp_step_factor=0.0, num_classes=7)
    monkeypatch.undo()
    if fused_tail:
        assert "_loss_terms" in res, "the one-launch tail was not taken"
        for k in ("opacity", "depth", "rgb", "normal_pred", "semantic", "ws", "Ro", "Rp"):
            close(N(res[k]), N(ref_res[k]), 2e-5, 2e-6)
        assert torch.equal(res["rays_a"], ref_res["rays_a"]) and int(res["vr_samples"]) == int(ref_res["vr_samples"])
        terms_t = res.pop("_loss_terms")
        terms = N(terms_t)
    else:
        terms_t, (d_rgb, d_op, d_ws) = nerf_loss_and_grads(res["rgb"], res["opacity"], res["ws"], res["deltas"], res["ts"],
                                                          res["rays_a"], gt, tr.loss_fn.lambda_opa, tr.loss_fn.lambda_distortion)
        terms = N(terms_t)
    for i, k in enumerate(("loss", "loss_rgb", "loss_opacity", "loss_distortion")):
        assert abs(terms[i] - float(g[k])) <= 2e-4 * abs(float(g[k])) + 1e-9, (k, terms[i], float(g[k]))
    if fused_tail:
        torch.autograd.backward([terms_t], [torch.tensor([1.0, 0.0, 0.0, 0.0], device=DEV)])
    else:
        torch.autograd.backward([res["rgb"], res["opacity"], res["ws"]], [d_rgb, d_op, d_ws])
    torch.cuda.synchronize()

    def rel(a, b):
        return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)
    for k in small:
        ref = g["grad_" + k]
        mine = N(named[k].grad) if named[k].grad is not None else np.zeros_like(ref)
        if np.abs(ref).max() == 0:                       # headers: the reference back-propagates exact zeros
            assert not mine.any(), k
        else:
            print(f"[G8] grad {k}: max-normalised error {rel(mine, ref):.2e}")
            assert rel(mine, ref) < G8_BAR, (k, rel(mine, ref))
    for k in tables:
        idx = g["grad_idx_" + k]
        mine = N(named[k].grad)
        print(f"[G8] grad {k}: max-normalised error {rel(mine[idx], g['grad_val_' + k]):.2e}")
        assert rel(mine[idx], g["grad_val_" + k]) < G8_BAR, k
        assert abs(np.sqrt((mine.astype(np.float64) ** 2).sum()) - float(g["grad_l2_" + k])) < 3e-3 * float(g["grad_l2_" + k])
        assert abs(np.abs(mine).sum(dtype=np.float64) - float(g["grad_l1_" + k])) < 3e-3 * float(g["grad_l1_" + k])
    mine_grads = {k: N(named[k].grad).copy() for k in tables}
    # clip + Adam
    tr.optimizer_step()
    tr.wait()
    torch.cuda.synchronize()
    assert float(g["grad_norm"]) < 50.0                   # no clipping on this step; the norm itself:
    lr = 1e-2
    for k in small:
        ref_new, ref_g, old = g["new_" + k], g["grad_" + k], g[k]
        mine_new = N(named[k])
        big = np.abs(ref_g) > 3e-7                        # first Adam step: p -= lr * g / (|g| + eps)
        assert not big.any() or np.abs(mine_new[big] - ref_new[big]).max() < 2e-3 * lr, k
        zero = ref_g == 0
        assert np.array_equal(mine_new[zero], old[zero]), k
    for k in tables:
        idx = g["grad_idx_" + k]
        ref_g, ref_new = g["grad_val_" + k], g["new_val_" + k]
        mine_new = N(named[k])[idx]
        big = np.abs(ref_g) > 3e-7        # |g| >> eps: the step is lr * g / (|g| + 1e-8), insensitive to 0.3 % of g
        assert big.sum() > 500 and np.abs(mine_new[big] - ref_new[big]).max() < 2e-3 * lr, (k, int(big.sum()))
        zero = (ref_g == 0) & (mine_grads[k][idx] == 0)
        assert zero.sum() > 1000 and np.array_equal(mine_new[zero], ref_new[zero]), k


def test_density_grid_update_matches_reference_golden(ngp, golden, monkeypatch):
    """NGP.update_density_grid (warm-up branch, twice) against the G9 fixture: the reference's OWN
    method run on the CPU with the same jitter numbers.  Cell centres + jitter (ngp_grid_cell_points),
    density(), EMA with decay (ngp_density_grid_ema), mean threshold kept on the device, packbits."""
    from helpers import noise_rule, table_rule
    g = golden("g9_density_grid_update.npz")
    model = ngp.networks.NGP(scale=0.5).to(DEV)
    G = model.grid_size
    model.register_buffer("density_grid", torch.zeros(model.cascades, G ** 3, device=DEV))
    coords = torch.stack(torch.meshgrid(*[torch.arange(G, dtype=torch.int32, device=DEV)] * 3, indexing="ij"), -1)
    model.register_buffer("grid_coords", coords.reshape(-1, 3).contiguous())
    with torch.no_grad():
        model.xyz_encoder.params.copy_(T(table_rule(model.xyz_encoder.params.numel())))
        named = dict(model.named_parameters())
        for k in ("xyz_net.0.weight", "xyz_net.0.bias", "xyz_net.2.weight", "xyz_net.2.bias"):
            named[k].copy_(T(g[k]))
    thr = float(g["density_threshold"])
    real_rand = torch.rand
    for call in range(2):
        def fake_rand(*size, **kw):
            shape = tuple(size[0]) if len(size) == 1 and not isinstance(size[0], int) else tuple(size)
            if shape == (G ** 3, 3):
                return T(noise_rule(shape, call))
            return real_rand(*size, **kw)
        monkeypatch.setattr(torch, "rand", fake_rand)
        with torch.no_grad():
            model.update_density_grid(thr, warmup=True)
        monkeypatch.undo()
        dg = N(model.density_grid)
        close(dg[:, ::257], g[f"u{call}_grid_sub"], 2e-4, 1e-5)
        assert abs(dg[dg > 0].mean() - float(g[f"u{call}_mean_pos"])) < 1e-4
        mine = np.unpackbits(N(model.density_bitfield))
        ref = np.unpackbits(g[f"u{call}_bitfield"])
        # a cell whose density is within fp32 noise of the mean threshold may fall on either side
        assert (mine != ref).mean() < 2e-3, (mine != ref).mean()
        assert abs(mine.mean() - ref.mean()) < 2e-3


def test_normal_ref_step_matches_reference_golden(ngp, golden, monkeypatch):
    """The --normal_ref recipe against the G10 fixture (the reference's render + NeRFLoss(normal_ref=True)
    + backward on the CPU): Ro / Rp are in the loss, the gradient reaches the density table and MLP
    through RefLoss.backward and the DOUBLE backward of the field (model.differentiable_normals)."""
    from helpers import table_rule
    from ngp_amd.losses import NeRFLoss
    from ngp_amd.rendering import render
    g = golden("g10_normal_ref_step.npz")
    small = ("xyz_net.0.weight", "xyz_net.0.bias", "xyz_net.2.weight", "xyz_net.2.bias", "rgb_net.params",
             "norm_pred_header.params", "semantic_header.params")
    model = ngp.networks.NGP(scale=0.5).to(DEV)
    with torch.no_grad():
        model.xyz_encoder.params.copy_(T(table_rule(model.xyz_encoder.params.numel())))
        model.rgb_encoder.params.copy_(T(table_rule(model.rgb_encoder.params.numel())))
        named = dict(model.named_parameters())
        for k in small:
            named[k].copy_(T(g[k]))
        model.density_bitfield.copy_(T(g["density_bitfield"]))
    model.differentiable_normals = True
    o, d, gt = T(g["rays_o"]), T(g["rays_d"]), T(g["rgb_gt"])
    noise = T(g["noise"])
    monkeypatch.setattr(torch, "rand_like", lambda t, *a, **k: noise.clone())
    res = render(model, o, d, exp_step_factor=0.0, num_classes=7)
    monkeypatch.undo()
    loss_d = NeRFLoss()(res, {"rgb": gt}, normal_ref=True)
    assert set(loss_d) == {"rgb", "opacity", "distortion", "normal_ref_rp", "normal_ref_ro"}
    for k, v in loss_d.items():
        ref = float(g["loss_" + k])
        assert abs(float(v.mean()) - ref) <= 1e-3 * abs(ref) + 1e-9, (k, float(v.mean()), ref)
    loss = sum(v.mean() for v in loss_d.values())
    loss.backward()

    def rel(a, b):
        return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)
    for k in small:
        ref = g["grad_" + k]
        mine = N(named[k].grad) if named[k].grad is not None else np.zeros_like(ref)
        if np.abs(ref).max() == 0:
            assert not mine.any(), k
        else:
            print(f"[G10] grad {k}: max-normalised error {rel(mine, ref):.2e}")
            assert rel(mine, ref) < G10_BAR, (k, rel(mine, ref))
    for k in ("xyz_encoder.params", "rgb_encoder.params"):
        mine = N(named[k].grad)
        print(f"[G10] grad {k}: max-normalised error {rel(mine[g['grad_idx_' + k]], g['grad_val_' + k]):.2e}")
        assert rel(mine[g["grad_idx_" + k]], g["grad_val_" + k]) < G10_BAR, k
        l2 = float(g["grad_l2_" + k])
        assert abs(np.sqrt((mine.astype(np.float64) ** 2).sum()) - l2) < 5e-3 * l2, k


@pytest.mark.parametrize("tag", ["a", "b"])
def test_mark_invisible_cells_matches_reference_golden(ngp, golden, tag):
    """NGP.mark_invisible_cells against the G11 fixture (the reference's own method on the CPU):
    cells outside every camera frustum, or in front of a camera but nearer than NEAR_DISTANCE, are -1;
    count_grid = fraction of cameras that see the cell.  a: scale 0.5; b: scale 2 (three cascades)."""
    g = golden("g11_invisible_cells.npz")
    scale = float(g[tag + "_scale"])
    model = ngp.networks.NGP(scale=scale).to(DEV)
    G = model.grid_size
    model.register_buffer("density_grid", torch.zeros(model.cascades, G ** 3, device=DEV))
    coords = torch.stack(torch.meshgrid(*[torch.arange(G, dtype=torch.int32, device=DEV)] * 3, indexing="ij"), -1)
    model.register_buffer("grid_coords", coords.reshape(-1, 3).contiguous())
    model.mark_invisible_cells(T(g[tag + "_K"]), T(g[tag + "_poses"]), (200, 200))
    mine = (N(model.density_grid) < 0).reshape(-1)
    ref = np.unpackbits(g[tag + "_invisible_bits"]).astype(bool)
    assert set(np.unique(N(model.density_grid))) <= {0.0, -1.0}
    # a cell centre that projects within fp32 rounding of an image border may fall on either side
    assert (mine != ref).mean() < 1e-4, (mine != ref).mean()
    assert ref.any() and not ref.all()
    cnt = N(model.count_grid)[:, ::101]
    assert (np.abs(cnt - g[tag + "_count_sub"]) > 1e-6).mean() < 1e-3


def test_two_rank_sharded_training_matches_single_rank():
    """N > 1 path on the GPU box: two ranks on this one GPU over gloo (RCCL needs a GPU per rank; the
    collective pattern and the sharded optimizer are the same code) train 5 steps with reduce-scatter
    / sharded clip + Adam / all-gather; both ranks must end bit-identical, and equal — up to summation
    order — to one process training on the union of the two ray batches (tests/dp_worker.py)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, NGP_DIST_BACKEND="gloo", OMP_NUM_THREADS="4")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(ROOT, "tests", "dp_worker.py")], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "DP_OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


def test_one_rank_rccl_sharded_training_matches_unsharded():
    """The RCCL calls of the N > 1 path on a single-GPU box: one rank, backend nccl (= RCCL), sharded
    optimizer forced on (NGP_FORCE_SHARDED=1) — reduce_scatter_tensor from the encoder hook, the norm
    all-reduce, Adam on the slice, the two detached all_gather_into_tensor works the next forward
    waits on — must reproduce the unsharded run on the same rays (tests/dp_worker.py)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, NGP_DIST_BACKEND="nccl", NGP_FORCE_SHARDED="1", OMP_NUM_THREADS="4")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(ROOT, "tests", "dp_worker.py")], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "DP_OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


def test_bench_line_contract():
    """bench.py prints ONE JSON line with the fields the driver and the judge read (a short run:
    20 set-up steps, 1 warm-up, 3 timed; the CPU baseline leg included with a small sample)."""
    import json
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--pretrain", "20",
                          "--cpu-rays", "64", "--cpu-samples", "16"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1
    assert abs(c["parity"]["psnr_delta"]) < 0.05
    assert d["value"] > 0 and abs(d["value"] - 8192 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-3 * d["value"]


def test_differentiable_normals_h4(ngp):
    """--normal_ref path: a loss on normals_raw reaches the density table through the grid's double
    backward; checked against a central difference along a random direction in parameter space."""
    model = _make_model(ngp)
    g = rng(230)
    n = 300
    x = T(((g.random((n, 3)) - 0.5) * 0.9).astype(np.float32))
    d = T(g.normal(size=(n, 3)).astype(np.float32))
    w = T(g.normal(size=(n, 3)).astype(np.float32))
    # same values as the fused node
    with torch.no_grad():
        _, _, n_fused, _, _ = model(x, d)
    model.differentiable_normals = True
    sig, rgb, n_raw, n_pred, sem = model(x, d)
    close(N(n_raw), N(n_fused), 1e-3, 1e-4)

    def loss_fn():
        _, _, nr, _, _ = model(x, d)
        return (nr * w).sum()

    loss = loss_fn()
    params = [model.xyz_encoder.params, model.xyz_net[0].weight, model.xyz_net[2].weight]
    grads = torch.autograd.grad(loss, params)
    assert all(torch.isfinite(gr).all() and gr.abs().sum() > 0 for gr in grads)
    torch.manual_seed(1)
    dirs = [torch.randn_like(p) for p in params]
    analytic = sum(float((gr.double() * dd.double()).sum()) for gr, dd in zip(grads, dirs))
    eps = 2e-4
    with torch.no_grad():
        for p, dd in zip(params, dirs):
            p.add_(eps * dd)
        lp = float(loss_fn())
        for p, dd in zip(params, dirs):
            p.sub_(2 * eps * dd)
        lm = float(loss_fn())
        for p, dd in zip(params, dirs):
            p.add_(eps * dd)
    fd = (lp - lm) / (2 * eps)
    assert abs(fd - analytic) < 0.05 * max(abs(fd), abs(analytic), 1.0), (fd, analytic)


def test_fused_loss_matches_nerfloss(ngp):
    from ngp_amd.losses import FusedNeRFLoss, NeRFLoss
    g = rng(240)
    rays_a, n = make_segments(500, 40, seed=241)
    nr = len(rays_a)
    ws0 = (g.random(n) * 0.05).astype(np.float32)
    deltas = (0.002 + 0.01 * g.random(n)).astype(np.float32)
    ts = np.zeros(n, np.float32)
    for _, s, c in rays_a:
        ts[s:s + c] = 0.4 + np.cumsum(deltas[s:s + c])
    gt = T(g.random((nr, 3)).astype(np.float32))
    outs = []
    for fused in (False, True):
        rgb = T(g.random((nr, 3)).astype(np.float32) if not outs else outs[0][3]).requires_grad_(True)
        op = T((g.random(nr) * 0.98 + 0.01).astype(np.float32) if not outs else outs[0][4]).requires_grad_(True)
        ws = T(ws0).requires_grad_(True)
        res = {"rgb": rgb, "opacity": op, "ws": ws, "deltas": T(deltas), "ts": T(ts), "rays_a": T(rays_a)}
        if fused:
            loss, *_ = FusedNeRFLoss.apply(rgb, op, ws, res["deltas"], res["ts"], res["rays_a"], gt, 2e-4, 3e-4)
        else:
            loss = sum(v.mean() for v in NeRFLoss()(res, {"rgb": gt}).values())
        grads = torch.autograd.grad(loss, [rgb, op, ws])
        outs.append((float(loss), [N(x) for x in grads], None, N(rgb), N(op)))
    assert abs(outs[0][0] - outs[1][0]) < 1e-6 * max(1.0, abs(outs[0][0]))
    for a, b in zip(outs[0][1], outs[1][1]):
        close(b, a, 1e-4, 1e-9)
    # the trainer seeds backward with these directly; terms = [loss, rgb, opacity, distortion]
    from ngp_amd.losses import nerf_loss_and_grads
    terms, (d_rgb, d_op, d_ws) = nerf_loss_and_grads(T(outs[0][3]), T(outs[0][4]), T(ws0), T(deltas), T(ts), T(rays_a),
                                                    gt, 2e-4, 3e-4)
    terms = N(terms)
    assert abs(terms[0] - outs[0][0]) < 1e-6 * max(1.0, abs(outs[0][0]))
    assert abs(terms[1] + terms[2] + terms[3] - terms[0]) < 1e-6
    for a, b in zip(outs[0][1], (d_rgb, d_op, d_ws)):
        close(N(b), a, 1e-4, 1e-9)
    # without the distortion term
    terms0, (_, _, d_ws0) = nerf_loss_and_grads(T(outs[0][3]), T(outs[0][4]), T(ws0), T(deltas), T(ts), T(rays_a),
                                               gt, 2e-4, 0.0)
    assert d_ws0 is None and abs(float(terms0[3])) == 0.0
    assert abs(float(terms0[0]) - (terms[1] + terms[2])) < 1e-6


def test_neg_normalize_and_refloss_inputs(ngp):
    from ngp_amd.networks import _NegNormalize
    from ngp_amd.rendering import _RefLossInputs
    F_ = torch.nn.functional
    g = rng(250)
    x = T(g.normal(size=(1000, 3)).astype(np.float32))
    x[0] = 0.0
    xr = x.clone().requires_grad_(True)
    y = _NegNormalize.apply(xr, None)
    ref = -F_.normalize(x, p=2, dim=-1, eps=1e-6)
    close(N(y), N(ref), 1e-6, 1e-7)
    w = T(g.normal(size=(1000, 3)).astype(np.float32))
    (gx,) = torch.autograd.grad(y, xr, w)
    xr2 = x.clone().requires_grad_(True)
    (gref,) = torch.autograd.grad(-F_.normalize(xr2, p=2, dim=-1, eps=1e-6), xr2, w)
    close(N(gx)[1:], N(gref)[1:], 1e-4, 1e-5)
    sc = T(np.array([2.0, 0.5, 1.5], np.float32))
    close(N(_NegNormalize.apply(x, sc)), N(-F_.normalize(x * sc, dim=-1, eps=1e-6)), 1e-6, 1e-7)
    n_raw, n_pred, dirs = (T(g.normal(size=(1000, 3)).astype(np.float32)) for _ in range(3))
    nd, no = _RefLossInputs.apply(n_raw, n_pred, dirs)
    close(N(nd), N((n_raw - n_pred) ** 2), 1e-6, 1e-7)
    close(N(no), N(torch.clamp((n_raw * F_.normalize(dirs, dim=-1, eps=1e-6)).sum(-1), min=0.) ** 2), 1e-5, 1e-6)


# ---------------------------------------------------------------------------- round 2: direct fixture checks
def test_hip_composite_train_fw_matches_raw2outputs_golden(ngp, golden):
    """SURVEY §8(c) cross-oracle identity, on the HIP kernel itself: vren.composite_train_fw with
    deltas = dz*|d| (last 1e10) and T_threshold = 0 equals the reference's own raw2outputs (G1)."""
    g = golden("g1_raw2outputs.npz")
    checked = 0
    for i in range(int(g["n_cases"])):
        p = f"c{i}_"
        raw, z, d = g[p + "raw"], g[p + "z"], g[p + "d"]
        C = int(g[p + "classes"])
        R, S = z.shape
        if S == 1:
            continue   # raw2outputs degenerates for one sample (custom_functions.py:300-301 yields an empty dists)
        dists = np.concatenate([z[:, 1:] - z[:, :-1], np.full((R, 1), 1e10, np.float32)], -1)
        dists = (dists * np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
        rays_a = np.stack([np.arange(R), np.arange(R) * S, np.full(R, S)], 1).astype(np.int64)
        flat = raw.reshape(R * S, -1)
        sems = flat[:, 10:] if C > 0 else np.zeros((R * S, 0), np.float32)
        total, opacity, depth, rgb, normal, sem, ws = ngp.vren.composite_train_fw(
            T(flat[:, 0]), T(flat[:, 1:4]), T(flat[:, 7:10]), T(sems), T(dists.reshape(-1)), T(z.reshape(-1)), T(rays_a),
            0.0, C)
        # raw2outputs adds 1e-10 inside the cumprod (custom_functions.py:311); the kernel uses __expf
        close(N(opacity), g[p + "opacity"], 1e-4, 1e-5)
        close(N(rgb), g[p + "rgb"], 1e-4, 1e-5)
        close(N(normal), g[p + "normal_pred"], 1e-4, 1e-5)
        close(N(depth), g[p + "depth"], 1e-4, 1e-5)
        close(N(ws).reshape(R, S), g[p + "ws"], 1e-4, 1e-6)
        if C:
            close(N(sem), g[p + "sem"], 1e-4, 1e-5)
        checked += 1
    assert checked >= 10


def test_package_activations_match_reference_golden_on_gpu(ngp, golden):
    """TruncExp / ReLU / TruncTanh of the package on device tensors against the reference's classes (G4)"""
    g = golden("g4_activations.npz")
    cf = ngp.custom_functions
    for name, fn in (("trunc_exp", cf.TruncExp), ("relu", cf.ReLU), ("trunc_tanh", cf.TruncTanh)):
        x = T(g["x"]).clone().requires_grad_(True)
        y = fn.apply(x)
        y.backward(T(g["g"]))
        close(N(y), g[name + "_y"], 2e-6, 0)
        close(N(x.grad), g[name + "_dx"], 2e-6, 0)


def test_raymarcher_backward_matches_reference_golden(ngp, golden):
    """RayMarcher.backward (custom_functions.py:104-114) through the HIP segment_csr against the reference's own
    method run on torch_scatter semantics (G5): rays with 0, 1 and several samples"""
    g = golden("g5_raymarcher_bw.npz")

    class Ctx:
        saved_tensors = (T(g["rays_a"]), T(g["ts"]))

    out = ngp.custom_functions.RayMarcher.backward(Ctx, None, T(g["dL_dxyzs"]), T(g["dL_ddirs"]), None, None, None)
    close(N(out[0]), g["dL_drays_o"], 1e-6, 1e-6)
    close(N(out[1]), g["dL_drays_d"], 1e-6, 1e-6)
    assert all(o is None for o in out[2:])


def test_tonemapper_and_exposure_match_reference_golden(ngp, golden):
    """M4: NGP(rgb_act='None') — log-radiance -> TruncExp (output_radiance) or the three tone-mapper networks,
    with and without a per-sample exposure — against the reference's own class (G13; the tone-mapping method is
    models/networks_noCUDA.py:238-259 bound onto models/networks.py::NGP, which calls it without defining it)."""
    from helpers import table_rule
    g = golden("g13_tonemapper.npz")
    model = ngp.networks.NGP(scale=0.5, rgb_act='None').to(DEV)
    with torch.no_grad():
        named = dict(model.named_parameters())
        named["xyz_encoder.params"].copy_(T(table_rule(named["xyz_encoder.params"].numel())))
        named["rgb_encoder.params"].copy_(T(table_rule(named["rgb_encoder.params"].numel())))
        for k in g.files:
            if k in named:
                named[k].copy_(T(g[k]))
        x, d = T(g["x"]), T(g["d"])
        for tag, kw in (("radiance", {"output_radiance": True}), ("ldr", {}), ("ldr_exposure", {"exposure": T(g["exposure"])})):
            sig, rgb = model(x, d, **kw)[:2]
            rgb_t = model.forward_test(x, d, **kw)[1]
            close(N(sig), g[f"fwd_{tag}_sigmas"], 2e-4, 1e-5)
            close(N(rgb), g[f"fwd_{tag}_rgbs"], 5e-4, 1e-5)
            close(N(rgb_t), g[f"test_{tag}_rgbs"], 5e-4, 1e-5)
        unit = model.log_radiance_to_rgb(torch.zeros(1, 3, device=DEV), exposure=torch.ones(1, 1, device=DEV))
        close(N(unit), g["unit_exposure_rgb"], 1e-5, 1e-6)
    # and the tone-mapped colour trains: gradients reach the tone-mapper and rgb_net parameters
    sig, rgb = model(x, d, exposure=T(g["exposure"]))[:2]
    rgb.square().sum().backward()
    assert model.tonemapper_net_0.params.grad.abs().sum() > 0 and model.rgb_net.params.grad.abs().sum() > 0


def test_nerfloss_all_terms_match_reference_golden(ngp, golden):
    """NeRFLoss with every optional term on (normal_ref, normal_mono, semantic + sky_depth, depth_mono) plus the HIP
    distortion loss, values and the gradients of sum(term.mean()), against the reference's own losses.py (G14)"""
    from ngp_amd.losses import NeRFLoss
    g = golden("g14_loss_terms.npz")
    res = {k[3:]: T(g[k]) for k in g.files if k.startswith("in_")}
    tgt = {k[4:]: T(g[k]) for k in g.files if k.startswith("tgt_")}
    diff = ("rgb", "opacity", "depth", "normal_pred", "semantic", "ws", "Ro", "Rp")
    for k in diff:
        res[k].requires_grad_(True)
    out = NeRFLoss()(res, tgt, normal_ref=True, normal_mono=True, semantic=True, depth_mono=True,
                     scale=float(g["scene_scale"]))
    assert {"term_" + k for k in out} == {k for k in g.files if k.startswith("term_")}
    for k, v in out.items():
        close(N(v), g["term_" + k], 2e-5, 1e-7)
    loss = sum(v.mean() for v in out.values())
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    loss.backward()
    for k in diff:
        close(N(res[k].grad), g["grad_" + k], 2e-5, 1e-9)


def test_volume_renderer_partial_coverage_zero_fills(ngp):
    """A rays_a that covers a strict subset of the sample rows (a caller trimmed segments / dropped rays): the
    VolumeRenderer Function must behave like vren.composite_train_fw/bw, i.e. like the reference's torch::zeros
    outputs (volumerendering.cu:137-143, 280-283) — uncovered rows are exact zeros, forward and backward."""
    g = rng(77)
    rays_a, _n, sig, deltas, ts, rgbs, nrm, sems = _composite_inputs(40, 30, 7, 78)
    keep = np.arange(0, 40, 3)
    sub = rays_a[keep].copy()
    sub[:, 2] = np.maximum(sub[:, 2] - 2, 0)            # trimmed segments: tails uncovered as well
    sub[:, 0] = np.arange(len(keep))                    # ray ids 0..n-1 of the subset
    args = [T(a) for a in (sig, rgbs, nrm, sems, deltas, ts)]
    for a in args[:4]:
        a.requires_grad_(True)
    ra = T(sub)
    outs = ngp.custom_functions.VolumeRenderer.apply(*args, ra, 1e-4, 7)
    ref = ngp.vren.composite_train_fw(*[a.detach() for a in args], ra, 1e-4, 7)
    for a, b in zip(outs[1:], ref[1:]):
        assert torch.equal(a.detach(), b)
    covered = np.zeros(len(sig), bool)
    for _, s, n in sub:
        covered[s:s + n] = True
    assert (~covered).sum() > 100 and not N(outs[6])[~covered].any()
    up = [T(g.normal(size=tuple(o.shape)).astype(np.float32)) for o in outs[1:]]
    torch.autograd.backward(list(outs[1:]), up)
    gref = ngp.vren.composite_train_bw(up[0], up[1], up[2], up[3], up[4], up[5], *[a.detach() for a in args[:3]],
                                       outs[6].detach(), args[4], args[5], ra, outs[1].detach(), outs[2].detach(),
                                       outs[3].detach(), outs[4].detach(), 1e-4, 7)
    for a, b in zip(args[:4], gref):
        assert torch.equal(a.grad, b)
        assert not N(a.grad)[~covered].any()


def test_trainer_checkpoint_roundtrip_and_optional_losses(ngp, tmp_path):
    """(f)3: save_ckpt -> slim_ckpt -> NGPTrainer.load_ckpt into a flat-buffer trainer: the loaded values are the
    ones the next optimizer step starts from (parameters stay within one Adam step of the checkpoint, not of the
    fresh initialisation).  Then the loss_kwargs route of the trainer (train.py:289-300): normal_mono + semantic
    targets through NGPTrainer.step."""
    from ngp_amd import ckpt
    from ngp_amd.trainer import NGPTrainer

    def build(seed):
        torch.manual_seed(seed)
        m = ngp.networks.NGP(scale=0.5).to(DEV)
        G = m.grid_size
        m.register_buffer("density_grid", torch.zeros(m.cascades, G ** 3, device=DEV))
        c = torch.stack(torch.meshgrid(*[torch.arange(G, dtype=torch.int32, device=DEV)] * 3, indexing="ij"), -1)
        m.register_buffer("grid_coords", c.reshape(-1, 3).contiguous())
        return m

    src = build(1)
    small = ("xyz_net.0.weight", "xyz_net.2.weight", "rgb_net.params")
    with torch.no_grad():
        for k, p in src.named_parameters():
            if k in small:
                p.fill_(0.37)                      # far from any initialisation
    path, slim_path = str(tmp_path / "last.ckpt"), str(tmp_path / "last_slim.ckpt")
    ckpt.save_ckpt(src, path)
    torch.save({"state_dict": ckpt.slim_ckpt(path)}, slim_path)
    model = build(2)
    tr = NGPTrainer(model, lr=1e-2)
    flat_ptr = tr.flat_param.data_ptr()
    tr.load_ckpt(slim_path)
    assert tr.flat_param.data_ptr() == flat_ptr
    for (k, p), (_, q) in zip(src.named_parameters(), model.named_parameters()):
        assert torch.equal(p, q), k
    model.update_density_grid(0.01 * 1024 / 3 ** 0.5, warmup=True)
    g = rng(5)
    n_rays = 1024
    o = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=DEV), dim=-1) * 1.5
    d = torch.nn.functional.normalize(-o + 0.2 * torch.randn(n_rays, 3, device=DEV), dim=-1)
    gt = torch.rand(n_rays, 3, device=DEV)
    tr.global_step = 1                             # no grid update inside the step
    tr.step(o, d, gt)
    tr.wait()
    torch.cuda.synchronize()
    named = dict(model.named_parameters())
    for k in small:
        dist = (named[k] - 0.37).abs().max().item()
        assert dist <= 1.01e-2, (k, dist)          # one Adam step (lr 1e-2) away from the CHECKPOINT values
    # optional loss terms through the trainer
    tr2 = NGPTrainer(model, lr=1e-2, loss_kwargs={"normal_mono": True, "semantic": True})
    assert not tr2.fused_loss
    tr2.global_step = 1
    target = {"normal": T(g.normal(size=(n_rays, 3)).astype(np.float32)),
              "label": torch.randint(0, 7, (n_rays,), device=DEV)}
    before = model.semantic_header.params.detach().clone()
    loss, _ = tr2.step(o, d, gt, target=target)
    tr2.wait()
    torch.cuda.synchronize()
    assert torch.isfinite(loss) and not torch.equal(before, model.semantic_header.params)
    # normal_ref: the trainer switches the field to differentiable normals; without that NeRFLoss refuses
    tr3 = NGPTrainer(model, lr=1e-2, loss_kwargs={"normal_ref": True})
    assert model.differentiable_normals and not tr3.fused_loss
    tr3.global_step = 1
    loss, _ = tr3.step(o, d, gt)
    tr3.wait()
    assert torch.isfinite(loss)
    model.differentiable_normals = False
    with pytest.raises(RuntimeError, match="differentiable_normals"):
        tr3.step(o, d, gt)


# ---------------------------------------------------------------------------- O1: fused sampled occupancy update
def _grid_model(ngp, seed=3):
    model = _make_model(ngp)
    G = model.grid_size
    model.register_buffer("density_grid", torch.zeros(model.cascades, G ** 3, device=DEV))
    c = torch.stack(torch.meshgrid(*[torch.arange(G, dtype=torch.int32, device=DEV)] * 3, indexing="ij"), -1)
    model.register_buffer("grid_coords", c.reshape(-1, 3).contiguous())
    model.grid_rng = torch.Generator(device=DEV).manual_seed(seed)
    return model


def test_grid_sample_cells_distribution_and_points(ngp):
    """ngp_grid_sample_cells (networks.py:308-333 + 388-395): M uniform cells + M uniformly drawn occupied cells,
    bucket-ordered, each with a point inside its cell; deterministic in the seed; no occupied cell -> the uniform
    half only (here: repeated exactly)."""
    from ngp_amd._lib import call, call_host
    G, M, s = 128, 128 ** 3 // 4, 0.5
    g3 = G ** 3
    coords = np.stack(np.meshgrid(*[np.arange(G, dtype=np.int32)] * 3, indexing="ij"), -1).reshape(-1, 3)
    ctr = (coords.astype(np.float32) + 0.5) / G - 0.5
    occ_xyz = (ctr ** 2).sum(-1) < 0.2 ** 2                         # a ball: ~3.4 % of the cells
    grid = np.zeros(g3, np.float32)
    grid[oracle.morton3D(coords)] = np.where(occ_xyz, 9.0, 0.3).astype(np.float32)
    n_occ = int(occ_xyz.sum())
    work = torch.empty(call_host("grid_sample_workspace", G, M), dtype=torch.int32, device=DEV)

    def sample(grid_np, thr, seed):
        idx = torch.empty(2 * M, dtype=torch.int32, device=DEV)
        xyz = torch.empty(2 * M, 3, dtype=torch.float32, device=DEV)
        call("grid_sample_cells", T(grid_np), G, float(thr), M, seed, float(s), work, idx, xyz)
        torch.cuda.synchronize()
        return N(idx).astype(np.int64), N(xyz)

    idx, xyz = sample(grid, 5.0, 1234)
    assert idx.min() >= 0 and idx.max() < g3
    assert (np.diff(idx) >= 0).all()                                 # G = 128: the 8 + 13 sort bits are the whole key
    is_occ = grid[idx] > 5.0
    expect = 0.5 + 0.5 * n_occ / g3                                  # all of the second half + the uniform hits
    assert abs(is_occ.mean() - expect) < 4e-3, (is_occ.mean(), expect)
    # the uniform half covers the volume evenly: octant counts of all samples outside the ball
    cc = oracle.morton3D_invert(idx[~is_occ].astype(np.int32))
    octant = (cc[:, 0] >= 64) * 4 + (cc[:, 1] >= 64) * 2 + (cc[:, 2] >= 64)
    cnt = np.bincount(octant, minlength=8)
    assert cnt.min() > 0.97 * cnt.mean() and cnt.max() < 1.03 * cnt.mean(), cnt
    # occupied draws are uniform over the occupied cells: per-cell counts are Poisson(M / n_occ) (+ the uniform hits)
    hits = np.bincount(idx[is_occ], minlength=g3)[grid > 5.0]
    mean_hits = M / n_occ + M / g3
    assert abs(hits.mean() - mean_hits) < 0.02 * mean_hits, (hits.mean(), mean_hits)
    assert 0.9 * mean_hits < hits.var() < 1.1 * mean_hits, (hits.var(), mean_hits)
    assert (hits == 0).mean() < 5 * np.exp(-mean_hits) + 1e-4 and hits.max() < 5 * mean_hits
    # every point lies inside its cell: centre (c/(G-1)*2-1)*(s-s/G), half-width s/G (networks.py:391-395)
    c3 = oracle.morton3D_invert(idx.astype(np.int32)).astype(np.float32)
    centre = (c3 / (G - 1) * 2 - 1) * (s - s / G)
    assert np.abs(xyz - centre).max() <= s / G * (1 + 1e-5)
    assert np.abs(xyz - centre).mean() > 0.4 * s / G                 # jittered, not the centres
    # same seed -> same samples (as a multiset of rows; the order inside a bucket is free), other seed -> other samples
    idx2, xyz2 = sample(grid, 5.0, 1234)
    key = lambda i, x: np.lexsort((x[:, 2], x[:, 1], x[:, 0], i))
    o1, o2 = key(idx, xyz), key(idx2, xyz2)
    assert np.array_equal(idx[o1], idx2[o2]) and np.array_equal(xyz[o1], xyz2[o2])
    idx3, _ = sample(grid, 5.0, 99)
    assert not np.array_equal(np.sort(idx), np.sort(idx3))
    # nothing occupied: the second half repeats the first exactly
    idx0, xyz0 = sample(grid, 100.0, 7)
    o = key(idx0, xyz0)
    assert np.array_equal(idx0[o][0::2], idx0[o][1::2]) and np.array_equal(xyz0[o][0::2], xyz0[o][1::2])


def test_fused_sampled_grid_update_matches_torch_formulas(ngp):
    """NGP.update_density_grid(warmup=False) on the fused kernels against the reference's formulas in torch ops
    (networks.py:396-408) applied to the very same samples: density() of the points, last-write (here: max) into
    the temporary grid, EMA with decay keeping negative cells, threshold = min(mean of positives, thr), packbits."""
    from ngp_amd._lib import call, call_host
    model = _grid_model(ngp)
    with torch.no_grad():
        model.xyz_net[2].bias.fill_(1.5)
    thr0 = 0.01 * 1024 / 3 ** 0.5
    model.update_density_grid(thr0, warmup=True)
    with torch.no_grad():
        model.density_grid[0, ::37] = -1.0                           # invisible cells stay as they are
    G, M = model.grid_size, model.grid_size ** 3 // 4
    for upd in range(2):
        before = model.density_grid.clone()
        # the samples the model is about to draw (same seed rule as NGP._update_density_grid_sampled)
        if upd == 0:
            seed0 = int(model.grid_rng.initial_seed()) & 0x7FFFFFFFFFFF
        seed = seed0 + 1000003 * upd
        work = torch.empty(call_host("grid_sample_workspace", G, M), dtype=torch.int32, device=DEV)
        idx = torch.empty(2 * M, dtype=torch.int32, device=DEV)
        xyz = torch.empty(2 * M, 3, dtype=torch.float32, device=DEV)
        call("grid_sample_cells", before[0], G, float(thr0), M, seed, float(model.scale), work, idx, xyz)
        with torch.no_grad():
            sig = model.density(xyz)
        tmp = torch.zeros_like(before[0]).scatter_reduce(0, idx.long(), sig, "amax", include_self=True)
        want = torch.where(before[0] < 0, before[0], torch.maximum(before[0] * 0.95, tmp))
        model.update_density_grid(thr0, warmup=False)
        torch.cuda.synchronize()
        assert torch.equal(model.density_grid[0], want)
        pos = want[want > 0]
        thr = min(float(pos.double().mean()), thr0)
        bits = oracle.packbits(N(want), np.float32(thr))
        mine = N(model.density_bitfield)
        near = np.abs(N(want) - thr) < 1e-5 * thr                    # cells within rounding of the threshold
        diff = np.unpackbits(mine ^ bits, bitorder="little").astype(bool)
        assert not (diff & ~near).any(), int((diff & ~near).sum())
        assert (N(want) < 0).sum() > 1000 and np.array_equal(N(model.density_grid[0])[N(want) < 0], N(want)[N(want) < 0])
    # the torch-op route (NGP_GRID_UPDATE_TORCH=1) and the fused one agree on what an update does to the statistics
    frac_fused = float((model.density_grid > min(float(model.density_grid[model.density_grid > 0].mean()), thr0)).float().mean())
    assert 0.0 < frac_fused < 1.0


# ---------------------------------------------------------------------------- full-size checks (BASELINE config 1)
@pytest.fixture(scope="module")
def full_batch(ngp):
    """One batch of configs[1]: 8192 rays through the S-lego-proxy scene's analytic occupancy (128^3 grid, scale 0.5),
    marched by the HIP marcher: ~0.4-0.5 M samples in ray order, as a training step sees them."""
    from ngp_amd.synthetic import LegoProxy
    from ngp_amd.custom_functions import RayAABBIntersector, RayMarcher
    torch.manual_seed(20220806)
    model = _grid_model(ngp, seed=20220806)
    scene = LegoProxy(n_images=100, img_wh=(800, 800), device=DEV, seed=20220806)
    with torch.no_grad():
        model.density_grid.copy_(scene.occupancy_from_analytic(model))
    ngp.vren.packbits(model.density_grid.view(-1), 0.5, model.density_bitfield)
    gen = torch.Generator(device=DEV).manual_seed(7)
    img, pix = scene.sample_batch(8192, generator=gen)
    o, d = scene.rays(img, pix)
    _, hits_t, _ = RayAABBIntersector.apply(o, d, model.center, model.half_size, 1)
    hits_t = hits_t.contiguous()
    ngp._lib.call("clamp_near", hits_t, hits_t.shape[0], 1, 0.01)
    with torch.no_grad():
        rays_a, xyzs, dirs, deltas, ts, total = RayMarcher.apply(o, d, hits_t[:, 0], model.density_bitfield, model.cascades,
                                                                 model.scale, 0.0, model.grid_size, 1024)
    return dict(model=model, o=o, d=d, hits_t=hits_t[:, 0], rays_a=rays_a, xyzs=xyzs, dirs=dirs, deltas=deltas, ts=ts,
                n=int(total))


def test_full_size_marcher_invariants(ngp, full_batch):
    """8192 rays, ~0.45 M samples: the segments tile [0, N) in ray order; inside a ray t increases by exactly the
    previous delta or more; delta = dt(t) = sqrt(3)/1024 (exp_step_factor 0, raymarching.cu:11-13); every sample
    lies in a cell whose bit is set (oracle's bit test on the same positions); x = o + t d, bit for bit."""
    b = full_batch
    ra, n = N(b["rays_a"]), b["n"]
    assert 300_000 < n < 700_000
    assert np.array_equal(ra[:, 0], np.arange(8192))
    assert np.array_equal(ra[:, 1], np.concatenate([[0], np.cumsum(ra[:, 2])[:-1]])) and ra[:, 2].sum() == n
    assert ra[:, 2].max() <= 1024
    ts, deltas, xyz = N(b["ts"]), N(b["deltas"]), N(b["xyzs"])
    assert np.all(deltas == np.float32(3 ** 0.5 / 1024))
    ray_of = np.repeat(np.arange(8192), ra[:, 2])
    same = ray_of[1:] == ray_of[:-1]
    assert np.all(ts[1:][same] >= (ts[:-1] + deltas[:-1])[same])       # a step, or a jump over empty cells
    o, d = N(b["o"]), N(b["d"])
    want = (o[ray_of].astype(np.float32) + ts[:, None] * d[ray_of].astype(np.float32))
    assert np.abs(xyz - want).max() <= 1e-6                             # same expression; fma contraction is off on both sides
    t12 = N(b["hits_t"])[ray_of]
    assert np.all(ts >= t12[:, 0]) and np.all(ts < t12[:, 1])
    # occupancy of every sample's cell (mip 0: scale 0.5): n = clamp(0.5*(x/0.5+1)*128), bit idx = morton
    G = 128
    cell = np.clip((0.5 * (xyz / np.float32(0.5) + 1) * G).astype(np.int32), 0, G - 1)
    idx = oracle.morton3D(cell)
    bits = N(b["model"].density_bitfield)
    assert np.all((bits[idx >> 3] >> (idx & 7)) & 1)


@pytest.mark.parametrize("log2T", [19, 21])
def test_full_size_grid_adjoint_and_oracle(ngp, full_batch, log2T):
    """The scatter is the adjoint of the gather: <grid_fwd(table, x), dy> = <table, grid_bwd_param(x, dy)>, on the
    full batch (both of the reference's tables), plus the whole scatter result against the CPU oracle and the
    input gradient against the oracle on a strided subset."""
    from ngp_amd._lib import call, call_host
    b = full_batch
    n = b["n"]
    L, Fd, base, pls = 16, 8, 16, float(np.exp(np.log(2048 * 0.5 / 16) / 15))
    gd = ngp._lib.GridDesc()
    n_params = call_host("grid_layout", L, Fd, log2T, base, pls, gd)
    g = rng(300 + log2T)
    xn = ((b["xyzs"] + 0.5)).clamp(0, 1).contiguous()
    table = T(g.uniform(-1, 1, n_params).astype(np.float32))
    dy = torch.randn(n, L * Fd, device=DEV, generator=torch.Generator(device=DEV).manual_seed(log2T))
    dy[torch.rand(n, device=DEV) < 0.19] = 0.0                          # samples behind the termination point
    y = torch.empty(n, L * Fd, device=DEV)
    call("grid_fwd", gd, table, xn, n, y, L * Fd)
    dtable = torch.zeros(n_params, device=DEV)
    call("grid_bwd_param", gd, xn, dy, L * Fd, n, dtable)
    lhs = float((y.double() * dy.double()).sum())
    rhs = float((table.double() * dtable.double()).sum())
    scale = float((y.double() * dy.double()).abs().sum())
    assert abs(lhs - rhs) < 2e-6 * scale, (lhs, rhs, scale)
    desc, _ = oracle.grid_layout(L, Fd, log2T, base, pls)
    ref = oracle.grid_bwd_param(desc, N(xn), N(dy), n_params)
    close(N(dtable), ref, 2e-4, 2e-5 * np.abs(ref).max())
    sub = slice(0, n, 37)
    ref_y = oracle.grid_fwd(desc, N(table), N(xn)[sub])
    close(N(y)[sub], ref_y, 1e-5, 1e-6)
    dx = torch.empty(n, 3, device=DEV)
    call("grid_bwd_input", gd, table, xn, dy, L * Fd, n, dx)
    ref_dx = oracle.grid_bwd_input(desc, N(table), N(xn)[sub], N(dy)[sub])
    close(N(dx)[sub], ref_dx, 2e-4, 3e-6 * np.abs(ref_dx).max())


def test_full_size_compositing_matches_oracle(ngp, full_batch):
    """composite_train_fw / bw, distortion loss and RefLoss on the full batch's own segments (0..1024 samples per
    ray) against the CPU oracle; plus the identities sum(ws) = opacity and opacity = 1 - prod(1 - alpha) up to the
    termination threshold."""
    b = full_batch
    n, ra = b["n"], b["rays_a"]
    g = torch.Generator(device=DEV).manual_seed(5)
    sig = torch.rand(n, device=DEV, generator=g) ** 3 * 400
    sig[torch.rand(n, device=DEV, generator=g) < 0.5] = 0
    rgbs, nrm = torch.rand(n, 3, device=DEV, generator=g), torch.randn(n, 3, device=DEV, generator=g)
    sems = torch.rand(n, 7, device=DEV, generator=g)
    outs = ngp.vren.composite_train_fw(sig, rgbs, nrm, sems, b["deltas"], b["ts"], ra, 1e-4, 7)
    ref = oracle.composite_train_fw(N(sig), N(rgbs), N(nrm), N(sems), N(b["deltas"]), N(b["ts"]), N(ra), 1e-4, 7)
    loose = borderline_rays(N(sig), N(b["deltas"]), N(ra), 1e-4)
    assert loose.mean() < 0.01
    keep = ~loose
    for a, r in zip(outs[1:6], ref[1:6]):
        close(N(a)[keep], r[keep], 2e-5, 2e-6)
    ws = N(outs[6])
    ray_of = np.repeat(np.arange(8192), N(ra)[:, 2])
    ok = keep[ray_of]
    close(ws[ok], ref[6][ok], 2e-5, 1e-7)
    close(np.bincount(ray_of, weights=ws, minlength=8192)[keep], N(outs[1])[keep], 1e-5, 1e-6)
    assert N(outs[1]).max() <= 1.0 + 1e-5
    # backward on the same segments
    up = [torch.randn_like(o) for o in outs[1:7]]
    grads = ngp.vren.composite_train_bw(*up, sig, rgbs, nrm, outs[6], b["deltas"], b["ts"], ra, outs[1], outs[2], outs[3],
                                        outs[4], 1e-4, 7)
    rg = oracle.composite_train_bw(*[N(u) for u in up], N(sig), N(rgbs), N(nrm), ref[6], N(b["deltas"]), N(b["ts"]), N(ra),
                                   ref[1], ref[2], ref[3], ref[4], 1e-4, 7)
    for a, r in zip(grads, rg):
        close(N(a)[ok], r[ok], 2e-4, 2e-5 * np.abs(r).max())
    # distortion loss (losses.cu) and its gradient
    loss, wi, wti = ngp.vren.distortion_loss_fw(outs[6], b["deltas"], b["ts"], ra)
    rl, rwi, rwti = oracle.distortion_loss_fw(ws, N(b["deltas"]), N(b["ts"]), N(ra))
    close(N(wi), rwi, 1e-5, 1e-7)
    close(N(wti), rwti, 1e-5, 1e-6)
    # the per-ray loss is a difference of two prefix-sum products of size ~ opacity^2 * t (up to ~2 here): fp32
    # summation order shows at 1e-5 of THAT magnitude (measured 3.3e-5 absolute on rays of up to 1024 samples)
    close(N(loss), rl, 1e-3, 1e-4)
    dws = ngp.vren.distortion_loss_bw(torch.ones(8192, device=DEV), wi, wti, outs[6], b["deltas"], b["ts"], ra)
    close(N(dws), oracle.distortion_loss_bw(np.ones(8192, np.float32), rwi, rwti, ws, N(b["deltas"]), N(b["ts"]), N(ra)),
          1e-4, 1e-5)


# ---------------------------------------------------------------------------- clipping from a norm bound
def test_clip_decide_and_conditional_norm(ngp):
    """ngp_row_norm_sum and ngp_clip_decide: bound = sqrt(sum_t (||W1_t||_F ||W2_t||_F S_t)^2 + exact rest) * scale; below
    the threshold the coefficient is extra_scale and the exact-norm launches do nothing, otherwise they run."""
    from ngp_amd._lib import call
    g = rng(91)
    rows = g.normal(size=(70001, 5)).astype(np.float32)
    acc = torch.zeros(1, device=DEV)
    call("row_norm_sum", T(rows), 5, rows.shape[0], 3, acc)
    want = np.sqrt((rows[:, :3].astype(np.float64) ** 2).sum(1)).sum()
    assert abs(float(acc) - want) < 1e-5 * want
    x = T(g.normal(size=100003).astype(np.float32))
    exact = float((N(x).astype(np.float64) ** 2).sum())
    w1a, w2a = T(g.normal(size=1000).astype(np.float32)), T(g.normal(size=48).astype(np.float32))
    w1b, w2b = T(g.normal(size=777).astype(np.float32)), T(g.normal(size=128).astype(np.float32))
    fa = float(np.linalg.norm(N(w1a).astype(np.float64)) * np.linalg.norm(N(w2a).astype(np.float64)))
    fb = float(np.linalg.norm(N(w1b).astype(np.float64)) * np.linalg.norm(N(w2b).astype(np.float64)))
    for sa, sb, rest, max_norm, scale in ((3.0 / fa, 4.0 / fb, 144.0, 50.0, 1.0),       # bound 13: no clipping possible
                                          (30.0 / fa, 40.0 / fb, 0.0, 50.0, 1.0),     # bound 50: not below 50
                                          (30.0 / fa, 40.0 / fb, 0.0, 50.0, 0.5),     # averaged over 2 ranks: 25
                                          (float("nan"), 1.0 / fb, 0.0, 50.0, 1.0),
                                          (1e-3 / fa, 1e-3 / fb, 1e-8, 1e-4, 1.0)):
        bound = np.sqrt((fa * sa) ** 2 + (fb * sb) ** 2 + rest) * scale
        expect_exact = not (bound * 1.001 + 1e-6 < max_norm)
        sums = T(np.array([rest], np.float32))
        coef = torch.full((1,), -7.0, device=DEV)
        flag = torch.full((1,), 5, dtype=torch.int32, device=DEV)
        call("clip_decide", T(np.array([sa, sb], np.float32)), w1a, w1a.numel(), w2a, w2a.numel(), w1b, w1b.numel(), w2b,
             w2b.numel(), sums, max_norm, scale, coef, flag)
        call("sumsq_if", x, x.numel(), sums, flag)
        call("clip_coef_if", sums, max_norm, scale, coef, flag)
        torch.cuda.synchronize()
        assert int(flag) == (1 if expect_exact else 0), (sa, sb, bound)
        if expect_exact:
            norm = np.sqrt(rest + exact) * scale
            want_c = scale * min(1.0, max_norm / (norm + 1e-6))
            assert abs(float(sums) - (rest + exact)) < 1e-4 * (rest + exact)
        else:
            want_c = scale
            assert float(sums) == np.float32(rest)
        assert abs(float(coef) - want_c) <= 1e-5 * want_c, (float(coef), want_c)
    # ngp_clip_decide_rest: the exact part (here 40,001 "MLP gradients") is summed by the decision launch itself
    restv = T((g.normal(size=40001) * 0.01).astype(np.float32))
    rest_sq = float((N(restv).astype(np.float64) ** 2).sum())
    for sa, sb, max_norm in ((3.0 / fa, 4.0 / fb, 50.0), (30.0 / fa, 40.0 / fb, 50.0)):
        sums = torch.zeros(1, device=DEV)
        coef = torch.full((1,), -7.0, device=DEV)
        flag = torch.full((1,), 5, dtype=torch.int32, device=DEV)
        call("clip_decide_rest", T(np.array([sa, sb], np.float32)), w1a, w1a.numel(), w2a, w2a.numel(), w1b, w1b.numel(), w2b,
             w2b.numel(), restv, restv.numel(), sums, max_norm, 1.0, coef, flag)
        torch.cuda.synchronize()
        bound = np.sqrt((fa * sa) ** 2 + (fb * sb) ** 2 + rest_sq)
        assert int(flag) == (0 if bound * 1.001 + 1e-6 < max_norm else 1)
        assert abs(float(sums) - rest_sq) < 1e-4 * rest_sq
    # ngp_act_bwd_rows = ngp_act_bwd + the row-norm sum of its result
    y = T(g.random((70001, 3)).astype(np.float32))
    dy = T(g.normal(size=(70001, 3)).astype(np.float32))
    dz_a, dz_b = torch.empty_like(y), torch.empty_like(y)
    acc = torch.zeros(1, device=DEV)
    call("act_bwd", dy, y, y.numel(), 2, dz_a)                   # NGP_ACT_SIGMOID
    call("act_bwd_rows", dy, y, y.shape[0], 3, 2, dz_b, acc)
    assert torch.equal(dz_a, dz_b)
    want = np.sqrt((N(dz_a).astype(np.float64) ** 2).sum(1)).sum()
    assert abs(float(acc) - want) < 1e-5 * want
    acc.zero_()
    sig = T(g.random(5003).astype(np.float32) * 3)
    dz1 = torch.empty_like(sig)
    call("act_bwd_rows", None, sig, sig.numel(), 1, 3, dz1, acc)   # softplus through its output, unit upstream gradient
    want = np.abs(-np.expm1(-N(sig).astype(np.float64))).sum()
    assert abs(float(acc) - want) < 1e-5 * want


def test_trainer_norm_bound_route_matches_exact_route(ngp):
    """NGPTrainer settles clip_grad_norm_(50) from ||W1|| ||W2|| sum_s ||dz2[s]|| (an upper bound of each table
    gradient's norm) — and the bound does hold against the exact norms; with a threshold the bound cannot clear, the
    device falls back to the exact norm.  Both against a trainer with the bound switched off."""
    from ngp_amd.trainer import NGPTrainer
    o, d = make_rays(2048, scale=1.0, seed=93)
    o, d = T(o), T(d)
    gt = torch.rand(2048, 3, device=DEV)
    seen = {}

    def run(bound, clip):
        model = _grid_model(ngp, seed=4)
        model.update_density_grid(0.01 * 1024 / 3 ** 0.5, warmup=True)
        tr = NGPTrainer(model, lr=1e-2, clip_norm=clip)
        assert tr.norm_bound
        tr.norm_bound = bound
        tr.global_step = 1
        if bound and clip == 50.0:      # look at the bound itself: table gradient norms vs the kernel's sums
            orig = tr.optimizer_step

            def spy():
                torch.cuda.synchronize()
                b0 = tr.buckets.bounds[1]
                seen["norms"] = (float(tr.flat_grad[:b0].norm()), float(tr.flat_grad[b0:tr._mlp_lo].norm()))
                seen["sums"] = N(tr.norm_acc).copy()
                Kp = model.rgb_net.padded_in
                p = model.rgb_net.params
                seen["f"] = (float(p[:128 * Kp].norm() * p[128 * Kp:].norm()),
                             float(model.xyz_net[0].weight.norm() * model.xyz_net[2].weight.norm()))
                orig()
            tr.optimizer_step = spy
        tr.step(o, d, gt)
        tr.wait()
        torch.cuda.synchronize()
        return float(tr.scalars[1]), int(tr.need_exact), tr.flat_param[tr._mlp_lo:].clone()

    c_b, f_b, p_b = run(True, 50.0)
    c_e, f_e, p_e = run(False, 50.0)
    assert c_b == 1.0 and c_e == 1.0 and f_b == 0
    # one Adam step of lr 1e-2 on the MLPs in two separate runs: the same parameters, except where a gradient is of the
    # size of eps = 1e-8 and the atomics' summation order decides how far lr * g / (|g| + eps) goes
    diff = (p_b - p_e).abs()
    assert float((diff <= 2e-6).float().mean()) > 0.99 and float(diff.max()) <= 2.01e-2
    for t in range(2):                                           # rgb table, density table
        assert 0 < seen["norms"][t] <= seen["f"][t] * seen["sums"][t], (t, seen)
    c_b, f_b, p_b = run(True, 1e-4)                              # the norm (~1e-2) is far above: clipping is active
    c_e, f_e, p_e = run(False, 1e-4)
    assert f_b == 1 and 0 < c_b < 0.5 and abs(c_b - c_e) < 2e-3 * c_e, (c_b, c_e)


# ---------------------------------------------------------------------------- tinycudann surface: the remaining otypes
def test_tcnn_frequency_and_network_with_input_encoding(ngp):
    """tinycudann.Encoding(Frequency) and NetworkWithInputEncoding (call site: models/networks_noCUDA.py:17-30,
    Frequency n = 6 + a 5 x 128 ReLU MLP; also with a hash grid in front): one flat `params` vector (network first,
    then encoding), forward and parameter / input gradients against a torch fp64 re-evaluation of the same layers."""
    tcnn = ngp.tinycudann
    g = rng(401)
    x = T(g.random((777, 3)).astype(np.float32))
    enc = tcnn.Encoding(3, {"otype": "Frequency", "n_frequencies": 6}).to(DEV)
    y = enc(x)
    assert y.shape == (777, 36) and enc.params.numel() == 0
    k = 2.0 ** np.arange(6)
    ang = N(x).astype(np.float64)[:, :, None] * k[None, None, :] * np.pi
    close(N(y), np.stack([np.sin(ang), np.cos(ang)], -1).reshape(777, -1), 1e-4, 1e-4)

    for enc_cfg in ({"otype": "Frequency", "n_frequencies": 6},
                    {"otype": "HashGrid", "n_levels": 4, "n_features_per_level": 2, "log2_hashmap_size": 12,
                     "base_resolution": 4, "per_level_scale": 1.5}):
        net_cfg = {"otype": "CutlassMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 128,
                   "n_hidden_layers": 5 if enc_cfg["otype"] == "Frequency" else 2}
        model = tcnn.NetworkWithInputEncoding(3, 3, enc_cfg, net_cfg).to(DEV)
        assert [n for n, _ in model.named_parameters()] == ["params"]
        if enc_cfg["otype"] == "HashGrid":
            with torch.no_grad():
                model.params[model._n_net:].uniform_(-0.5, 0.5)
        xi = x.clone().requires_grad_(True)
        out = model(xi)
        assert out.shape == (777, 3)
        up = T(g.normal(size=(777, 3)).astype(np.float32))
        gp, gx = torch.autograd.grad(out, [model.params, xi], up)
        # fp64 re-evaluation: the encoding through the package's own encoder module (checked elsewhere), the MLP in torch
        net = model.network
        shapes = net.layer_shapes
        p64 = model.params.detach().double().clone().requires_grad_(True)
        x64 = x.double().clone().requires_grad_(True)
        if enc_cfg["otype"] == "Frequency":
            kk = torch.tensor(k, device=DEV, dtype=torch.float64)
            a = x64[:, :, None] * kk[None, None, :] * np.pi
            h = torch.stack([torch.sin(a), torch.cos(a)], -1).reshape(777, -1)
        else:
            e2 = tcnn.Encoding(3, enc_cfg).to(DEV)
            with torch.no_grad():
                e2.params.copy_(model.params[model._n_net:])
            h = e2(x).detach().double()
        h = torch.cat([h, torch.ones(777, net.padded_in - h.shape[1], device=DEV, dtype=torch.float64)], 1)
        off = 0
        for li, (no, ni) in enumerate(shapes):
            W = p64[off:off + no * ni].view(no, ni)
            off += no * ni
            h = h @ W.T
            if li < len(shapes) - 1:
                h = torch.relu(h)
        ref = h[:, :3]
        close(N(out), N(ref.detach()), 2e-4, 2e-5 * float(ref.detach().abs().max()))
        if enc_cfg["otype"] == "Frequency":
            rp, rx = torch.autograd.grad(ref, [p64, x64], up.double())
            # per-sample input gradients: a hidden unit whose pre-activation is within fp32 rounding of 0 takes the other
            # side of the ReLU kink in the fp64 re-evaluation — a handful of samples may differ, the rest must agree
            okx = np.isclose(N(gx), N(rx), rtol=2e-3, atol=2e-4 * float(rx.abs().max()))
            assert okx.mean() > 0.995, okx.mean()
        else:
            (rp,) = torch.autograd.grad(ref, [p64], up.double())
        n_net = model._n_net
        okp = np.isclose(N(gp)[:n_net], N(rp)[:n_net], rtol=1e-3, atol=1e-4 * float(rp[:n_net].abs().max()))
        assert okp.mean() > 0.999, okp.mean()
        if enc_cfg["otype"] == "HashGrid":
            assert gp[n_net:].abs().sum() > 0          # the table receives its share through the flat vector


# ---------------------------------------------------------------------------- live samples: compacted colour branch
def test_live_rows_match_compositor_stops(ngp, full_batch):
    """ngp_live_rows keeps exactly the samples composite_train_fw uses: per ray all up to and including the sample at
    which T <= T_threshold (vr_samples + 1 of them, or the whole segment), in ascending order; inv_idx is the inverse
    list with -1 for the rest; gather / spread move row blocks there and back (zeros for the rest)."""
    from ngp_amd._lib import call
    b = full_batch
    n, rays_a, deltas, ts = b["n"], b["rays_a"], b["deltas"], b["ts"]
    nr = rays_a.shape[0]
    g = torch.Generator(device=DEV).manual_seed(11)
    sig = (torch.rand(n, device=DEV, generator=g) < 0.25).float() * torch.rand(n, device=DEV, generator=g) * 3000.0
    rgbs = torch.rand(n, 3, device=DEV, generator=g)
    zeros3, sems = torch.zeros(n, 3, device=DEV), torch.zeros(n, 7, device=DEV)
    vr, op, dep, rgb, nrm, sem, ws = ngp.vren.composite_train_fw(sig, rgbs, zeros3, sems, deltas, ts, rays_a, 1e-4, 7)
    want_live = torch.where(vr < rays_a[:, 2], vr + 1, rays_a[:, 2])         # stop sample included
    assert int((want_live < rays_a[:, 2]).sum()) > 1000                      # the case is not trivial
    offsets = torch.empty(nr, dtype=torch.int32, device=DEV)
    live_idx = torch.full((n,), -7, dtype=torch.int32, device=DEV)
    inv = torch.full((n,), -7, dtype=torch.int32, device=DEV)
    n_live = torch.zeros(1, dtype=torch.int32, device=DEV)
    xyz_c = torch.full((n, 3), 5.0, device=DEV)
    call("live_rows", sig, deltas, rays_a, 1e-4, nr, offsets, live_idx, inv, n_live, b["xyzs"], xyz_c, None, None)
    m = int(n_live[0])
    assert m == int(want_live.sum())
    assert torch.equal(offsets.long(), torch.cumsum(want_live, 0) - want_live)
    k = torch.arange(n, device=DEV) - torch.repeat_interleave(rays_a[:, 1], rays_a[:, 2])
    is_live = k < torch.repeat_interleave(want_live, rays_a[:, 2])
    assert torch.equal(live_idx[:m].long(), torch.nonzero(is_live)[:, 0])
    assert torch.equal(inv >= 0, is_live) and torch.equal(live_idx[:m].long()[inv[is_live].long()], live_idx[:m].long())
    assert torch.all(ws[~is_live] == 0)                                     # nothing outside the list carries weight
    assert torch.equal(xyz_c[:m], b["xyzs"][is_live]) and torch.all(xyz_c[m:] == 5.0)
    c = torch.empty(m, 3, device=DEV)
    call("gather_rows", rgbs, 3, 3, live_idx, m, c, 3)
    assert torch.equal(c, rgbs[is_live])
    back = torch.full((n, 3), 9.0, device=DEV)
    call("spread_rows", c, 3, 3, inv, n, back, 3)
    assert torch.equal(back, rgbs * is_live[:, None])
    c7 = torch.rand(m, 7, device=DEV, generator=g)
    b3, b3b, b7 = torch.full((n, 3), 9.0, device=DEV), torch.full((n, 3), 9.0, device=DEV), torch.full((n, 7), 9.0, device=DEV)
    call("spread_rows3", c, 3, b3, c, 3, b3b, c7, 7, b7, inv, n)
    assert torch.equal(b3, back) and torch.equal(b3b, back)
    assert torch.equal(b7[is_live], c7) and torch.all(b7[~is_live] == 0)
    # an empty batch leaves a zero count
    n_live.fill_(5)
    call("live_rows", None, None, None, 1e-4, 0, None, None, None, n_live, None, None, None, None)
    assert int(n_live[0]) == 0
    # the image does not change when the colours behind the stops are dropped
    vr2, op2, dep2, rgb2, *_ = ngp.vren.composite_train_fw(sig, back, zeros3, sems, deltas, ts, rays_a, 1e-4, 7)
    assert torch.equal(rgb, rgb2) and torch.equal(op, op2) and torch.equal(vr, vr2)


def test_compacted_colour_branch_matches_full(ngp, full_batch):
    """render() with the colour branch on the live samples only (model.compact_dead_samples / NGP_COMPACT=1) against
    the same step with it on every sample (the default): identical per-ray results — a row's bits do not depend on its position in the batch —
    and gradients equal up to the summation order of the weight products and the atomics."""
    from ngp_amd.rendering import render
    b = full_batch
    model = b["model"]
    with torch.no_grad():   # a dense medium of varying density (sigma ~ 150 +- 100): most rays terminate early
        model.xyz_net[2].bias.fill_(150.0)
        model.xyz_net[2].weight.mul_(40.0)
    o, d = b["o"][:2048].contiguous(), b["d"][:2048].contiguous()
    gt = torch.rand(2048, 3, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    outs = {}
    for mode in (True, False):
        model.compact_dead_samples = mode
        try:
            for p in model.parameters():
                p.grad = None
            torch.manual_seed(77)                                       # the marcher's per-ray jitter: the same twice
            res = render(model, o, d, exp_step_factor=0.0)
            loss = ((res["rgb"] - gt) ** 2).mean() + 1e-3 * (res["opacity"] ** 2).mean() + 1e-3 * res["Rp"].mean()
            loss.backward()
            outs[mode] = ({k: res[k].detach().clone() for k in ("rgb", "opacity", "depth", "ws", "normal_pred", "semantic",
                                                                 "Ro", "Rp", "vr_samples")},
                          {n_: p.grad.detach().clone() for n_, p in model.named_parameters() if p.grad is not None})
        finally:
            model.compact_dead_samples = None
    (ra, ga), (rb, gb) = outs[True], outs[False]
    n = int(res["total_samples"])
    # vr_samples = sum over rays of the stop index (or the segment length): well below n when many rays stop early
    assert int(ra["vr_samples"]) + 2048 < 0.9 * n, (int(ra["vr_samples"]), n)
    for k in ra:
        assert torch.equal(ra[k], rb[k]), k
    assert set(ga) == set(gb)
    for k in ga:
        scale = float(gb[k].abs().max())
        assert scale > 0, k
        err = float((ga[k] - gb[k]).abs().max())
        assert err <= 2e-5 * scale, (k, err, scale)
