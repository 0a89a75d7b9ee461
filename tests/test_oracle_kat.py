"""Known-answer and consistency tests that pin the CPU oracle where the reference offers no golden
vectors (the `vren` kernels and tiny-cuda-nn cannot run here).  CPU only."""
import numpy as np
import pytest

import oracle
from helpers import make_bitfield, make_rays, make_segments, rng


def close(a, b, rtol, atol):
    np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=rtol, atol=atol)


# ---------------------------------------------------------------- intersections / morton / packbits
def test_aabb_hand_cases():
    o = np.array([[0, 0, -2], [0, 0, -2], [0, 0, 0], [2, 2, 2], [0, 0, -2]], np.float32)
    d = np.array([[0, 0, 1], [0, 0, -1], [0, 0, 1], [1, 0, 0], [0.25, 0, 1]], np.float32)
    cnt, t, idx = oracle.ray_aabb_intersect(o, d, np.zeros((1, 3), np.float32), np.full((1, 3), 0.5, np.float32), 1)
    assert list(cnt) == [1, 0, 1, 0, 1]
    close(t[0, 0], [1.5, 2.5], 0, 0)          # enters at z=-0.5, leaves at z=+0.5
    close(t[1, 0], [-1, -1], 0, 0)            # pointing away: miss
    close(t[2, 0], [0.0, 0.5], 0, 0)          # origin inside: t1 clamped to 0
    close(t[3, 0], [-1, -1], 0, 0)
    assert idx[0, 0] == 0 and idx[1, 0] == -1
    assert 1.5 <= t[4, 0, 0] < t[4, 0, 1]


def test_aabb_multi_voxel_sorted_with_unused_first():
    o = np.array([[-3, 0, 0]], np.float32)
    d = np.array([[1, 0, 0]], np.float32)
    centers = np.array([[2, 0, 0], [0, 0, 0], [0, 5, 0]], np.float32)
    half = np.full((3, 3), 0.5, np.float32)
    cnt, t, idx = oracle.ray_aabb_intersect(o, d, centers, half, 3)
    assert cnt[0] == 2
    # ascending on t1, the unused (-1) slot first — what torch::sort + gather produce (intersection.cu:94-97)
    assert list(idx[0]) == [-1, 1, 0]
    close(t[0], [[-1, -1], [2.5, 3.5], [4.5, 5.5]], 0, 0)


def test_morton_known_values_and_roundtrip():
    c = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 1], [2, 0, 0], [127, 127, 127], [5, 3, 6]], np.int32)
    m = oracle.morton3D(c)
    assert list(m[:7]) == [0, 1, 2, 4, 7, 8, 128 ** 3 - 1]
    # 5=101b 3=011b 6=110b -> bit k of the code = x_k | y_k<<1 | z_k<<2 per triple
    assert m[7] == sum((((5 >> k) & 1) | (((3 >> k) & 1) << 1) | (((6 >> k) & 1) << 2)) << (3 * k) for k in range(3))
    g = rng(1)
    c = g.integers(0, 1024, (10000, 3)).astype(np.int32)
    assert np.array_equal(oracle.morton3D_invert(oracle.morton3D(c)), c)


def test_packbits_bit_order():
    grid = np.zeros(16, np.float32)
    grid[[0, 3, 9]] = 1.0
    grid[5] = 0.5  # not strictly greater than the threshold
    out = oracle.packbits(grid, 0.5)
    assert list(out) == [0b00001001, 0b00000010]


# ---------------------------------------------------------------- marcher
def _setup(cascades, scale, fill, n_rays, seed):
    o, d = make_rays(n_rays, scale=min(scale, 2.0), seed=seed)
    _, hits_t, _ = oracle.ray_aabb_intersect(o, d, np.zeros((1, 3), np.float32), np.full((1, 3), scale, np.float32), 1)
    hits_t = np.ascontiguousarray(hits_t[:, 0])
    m = (hits_t[:, 0] >= 0) & (hits_t[:, 0] < 0.01)
    hits_t[m, 0] = 0.01
    bits = make_bitfield(cascades, 128, fill=fill, seed=seed + 1)
    noise = rng(seed + 2).random(n_rays).astype(np.float32)
    return o, d, hits_t, bits, noise


@pytest.mark.parametrize("cascades,scale,esf", [(1, 0.5, 0.0), (5, 8.0, 1 / 256)])
def test_marcher_invariants(cascades, scale, esf):
    n_rays = 400
    o, d, hits_t, bits, noise = _setup(cascades, scale, 0.05, n_rays, seed=5)
    rays_a, xyz, dirs, deltas, ts, counter = oracle.raymarching_train(o, d, hits_t, bits, cascades, scale, esf, noise,
                                                                      128, 1024)
    n = int(counter[0])
    assert counter[1] == n_rays and n > 0
    assert np.array_equal(rays_a[:, 0], np.arange(n_rays))
    assert np.array_equal(rays_a[:, 1], np.cumsum(rays_a[:, 2]) - rays_a[:, 2])
    assert rays_a[:, 2].max() <= 1024 and rays_a[:, 2].sum() == n
    assert np.all(rays_a[hits_t[:, 0] < 0, 2] == 0)          # rays that miss the box get no samples
    dt_min, dt_max = np.float32(3 ** 0.5 / 1024), np.float32(3 ** 0.5 * 2 * scale / 128)
    assert deltas[:n].min() >= dt_min * 0.999 and deltas[:n].max() <= dt_max * 1.001
    for r in range(0, n_rays, 7):
        s, c = rays_a[r, 1], rays_a[r, 2]
        if c == 0:
            continue
        t = ts[s:s + c]
        assert t[0] >= hits_t[r, 0] and t[-1] < hits_t[r, 1]
        assert np.all(np.diff(t) > 0)
        close(xyz[s:s + c], o[r] + t[:, None] * d[r], 1e-6, 1e-6)
        assert np.array_equal(dirs[s:s + c], np.repeat(d[r][None], c, 0))
        # every sample sits in an occupied cell of the cascade the marcher selected
        p = xyz[s:s + c]
        mx = np.abs(p).max(1)
        mip_pos = np.clip(np.frexp(mx)[1] + 1, 0, cascades - 1)
        mip_dt = np.clip(np.frexp(deltas[s:s + c] * 128)[1], 0, cascades - 1)
        mip = np.maximum(mip_pos, mip_dt)
        bound = np.minimum(np.ldexp(1.0, mip - 1), scale).astype(np.float32)
        cell = np.clip(0.5 * (p / bound[:, None] + 1) * 128, 0, 127).astype(np.int32)
        idx = mip.astype(np.int64) * 128 ** 3 + oracle.morton3D(cell).astype(np.int64)
        assert np.all((bits[idx // 8] >> (idx % 8)) & 1)


def test_marcher_full_and_empty_grids():
    o, d, hits_t, _, noise = _setup(1, 0.5, 0.05, 128, seed=9)
    full = np.full(128 ** 3 // 8, 255, np.uint8)
    rays_a, _, _, deltas, ts, counter = oracle.raymarching_train(o, d, hits_t, full, 1, 0.5, 0.0, noise, 128, 1024)
    hit = hits_t[:, 0] >= 0
    # all-occupied: uniform steps of sqrt(3)/1024 from t1+noise*dt to t2
    expect = np.ceil((hits_t[hit, 1] - (hits_t[hit, 0] + noise[hit] * deltas[0])) / deltas[0])
    assert np.all(np.abs(rays_a[hit, 2] - np.minimum(expect, 1024)) <= 1)
    empty = np.zeros(128 ** 3 // 8, np.uint8)
    _, _, _, _, _, counter = oracle.raymarching_train(o, d, hits_t, empty, 1, 0.5, 0.0, noise, 128, 1024)
    assert counter[0] == 0


def test_test_marcher_resumes_where_it_stopped():
    """raymarching_test in rounds of 4 reproduces one round of 16 when scale == cascades
    (its calc_dt receives `cascades` as scale, raymarching.cu:370,399)."""
    cascades, scale = 1, 1.0
    o, d, hits_t, bits, _ = _setup(cascades, scale, 0.2, 200, seed=11)
    alive = np.arange(200, dtype=np.int64)
    h1 = hits_t.copy()
    x16, _, d16, t16, n16 = oracle.raymarching_test(o, d, h1, alive, bits, cascades, scale, 0.0, 128, 1024, 16)
    h2 = hits_t.copy()
    got_t = [[] for _ in range(200)]
    for _ in range(4):
        _, _, _, t4, n4 = oracle.raymarching_test(o, d, h2, alive, bits, cascades, scale, 0.0, 128, 1024, 4)
        for r in range(200):
            got_t[r] += list(t4[r, :n4[r]])
    for r in range(200):
        assert np.array_equal(np.asarray(got_t[r], np.float32), t16[r, :n16[r]])
    assert np.array_equal(h1, h2)


# ---------------------------------------------------------------- compositing / losses
def _segments(n_rays=60, max_len=40, seed=3, classes=7):
    g = rng(seed)
    rays_a, n = make_segments(n_rays, max_len, seed=seed)
    sig = (g.random(n) * 20).astype(np.float32)
    deltas = (0.005 + 0.02 * g.random(n)).astype(np.float32)
    ts = np.zeros(n, np.float32)
    for _, s, c in rays_a:
        ts[s:s + c] = 0.3 + np.cumsum(deltas[s:s + c])
    return g, rays_a, n, sig, deltas, ts


def test_composite_train_fw_closed_form_and_early_stop():
    # one ray, constant sigma: w_k = (1-e^{-s d}) e^{-s d k}
    n = 50
    sig = np.full(n, 5.0, np.float32)
    dl = np.full(n, 0.1, np.float32)
    ts = (np.arange(n) * 0.1).astype(np.float32)
    rays_a = np.array([[0, 0, n]], np.int64)
    rgbs = np.ones((n, 3), np.float32)
    z3, z0 = np.zeros((n, 3), np.float32), np.zeros((n, 0), np.float32)
    total, op, depth, rgb, _, _, ws = oracle.composite_train_fw(sig, rgbs, z3, z0, dl, ts, rays_a, 0.0, 0)
    a = 1 - np.exp(-0.5)
    close(ws, a * np.exp(-0.5 * np.arange(n)), 1e-5, 1e-8)
    close(op[0], 1 - np.exp(-0.5 * n), 1e-5, 0)
    assert total[0] == n
    # T after k+1 samples is e^{-0.5 (k+1)}; first k with T <= 1e-4 is k = 18
    total, op, *_rest, ws = oracle.composite_train_fw(sig, rgbs, z3, z0, dl, ts, rays_a, 1e-4, 0)
    assert total[0] == 18 and np.all(ws[19:] == 0) and ws[18] > 0


def test_composite_train_bw_matches_finite_differences():
    g, rays_a, n, sig, deltas, ts = _segments()
    C = 3
    rgbs, nrm = g.random((n, 3)).astype(np.float32), g.random((n, 3)).astype(np.float32)
    sems = g.random((n, C)).astype(np.float32)
    nr = len(rays_a)
    wO, wD = g.normal(size=nr), g.normal(size=nr)
    wR, wW = g.normal(size=(nr, 3)), g.normal(size=n)

    def loss(sig_, rgbs_):
        _, op, dp, rgb, _, _, ws = oracle.composite_train_fw(sig_, rgbs_, nrm, sems, deltas, ts, rays_a, 0.0, C)
        return (op * wO).sum() + (dp * wD).sum() + (rgb * wR).sum() + (ws * wW).sum()

    _, op, dp, rgb, nm, sm, ws = oracle.composite_train_fw(sig, rgbs, nrm, sems, deltas, ts, rays_a, 0.0, C)
    dsig, drgbs, _, _ = oracle.composite_train_bw(wO, wD, wR, np.zeros((nr, 3)), np.zeros((nr, C)), wW, sig, rgbs, nrm,
                                                  ws, deltas, ts, rays_a, op, dp, rgb, nm, 0.0, C)
    eps = 1e-2
    for k in rng(4).choice(n, 12, replace=False):
        sp, sm_ = sig.copy(), sig.copy()
        sp[k] += eps
        sm_[k] -= eps
        fd = (loss(sp, rgbs) - loss(sm_, rgbs)) / (2 * eps)
        assert abs(fd - dsig[k]) < 2e-2 * max(1.0, abs(fd)), (k, fd, dsig[k])
        rp, rm = rgbs.copy(), rgbs.copy()
        rp[k, 1] += eps
        rm[k, 1] -= eps
        fd = (loss(sig, rp) - loss(sig, rm)) / (2 * eps)
        assert abs(fd - drgbs[k, 1]) < 2e-3 * max(1.0, abs(fd))


def test_distortion_loss_matches_definition_and_gradient():
    """Against the O(n^2) definition: sum_ij w_i w_j |t_i - t_j| + 1/3 sum_i w_i^2 delta_i (Mip-NeRF 360)."""
    g, rays_a, n, sig, deltas, ts = _segments(n_rays=30, max_len=25, seed=6)
    ws = oracle.composite_train_fw(sig, np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32),
                                   np.zeros((n, 0), np.float32), deltas, ts, rays_a, 0.0, 0)[6]
    loss, wi, wti = oracle.distortion_loss_fw(ws, deltas, ts, rays_a)
    for i, (ray, s, c) in enumerate(rays_a):
        w, t, dl = ws[s:s + c].astype(np.float64), ts[s:s + c].astype(np.float64), deltas[s:s + c].astype(np.float64)
        ref = (w[:, None] * w[None, :] * np.abs(t[:, None] - t[None, :])).sum() + (w * w * dl).sum() / 3
        close(loss[ray], ref, 2e-4, 1e-6)
    dl_dloss = g.normal(size=len(rays_a)).astype(np.float32)
    grad = oracle.distortion_loss_bw(dl_dloss, wi, wti, ws, deltas, ts, rays_a)
    eps = 1e-3
    for k in rng(7).choice(n, 10, replace=False):
        wp, wm = ws.copy(), ws.copy()
        wp[k] += eps
        wm[k] -= eps
        fd = ((oracle.distortion_loss_fw(wp, deltas, ts, rays_a)[0] - oracle.distortion_loss_fw(wm, deltas, ts, rays_a)[0])
              * dl_dloss).sum() / (2 * eps)
        assert abs(fd - grad[k]) < 2e-2 * max(0.05, abs(fd)), (k, fd, grad[k])


def test_refloss_is_compositing_with_other_payload():
    g, rays_a, n, sig, deltas, ts = _segments(seed=8)
    nd = g.random((n, 3)).astype(np.float32)
    no = g.random(n).astype(np.float32)
    lo, lp = oracle.composite_refloss_fw(sig, nd, no, deltas, ts, rays_a, 1e-4)
    _, _, depth, rgb, *_ = oracle.composite_train_fw(sig, nd, np.zeros((n, 3), np.float32), np.zeros((n, 0), np.float32),
                                                     deltas, no, rays_a, 1e-4, 0)
    close(lp, rgb, 1e-6, 1e-7)
    close(lo, depth, 1e-6, 1e-7)


# ---------------------------------------------------------------- tiny-cuda-nn semantics
def test_hash_index_known_answers():
    """Coherent-prime hash and dense indexing, computed by hand."""
    d, n = oracle.grid_layout(2, 1, 4, 2, 4.0)   # level 0: res 2 (dense, 8 rows); level 1: res 8 hashed into 16 rows
    assert list(d.resolution)[:2] == [2, 8] and list(d.offsets)[:3] == [0, 8, 24] and n == 24
    table = np.arange(24, dtype=np.float32)
    # x exactly on lattice points: the encoding is the table entry of that point
    # level 0: scale 1 -> pos = x+0.5; x=(0.5,0.5,0.5) -> cell (1,1,1), w=0 -> dense row 1+2+4 = 7
    # level 1: scale 7 -> x=(0.5,0.5,0.5) -> pos 4.0 -> cell (4,4,4), w=0 -> hash
    y = oracle.grid_fwd(d, table, np.array([[0.5, 0.5, 0.5]], np.float32))
    h = (4 * 1) ^ ((4 * 2654435761) & 0xFFFFFFFF) ^ ((4 * 805459861) & 0xFFFFFFFF)
    close(y[0], [7.0, 8 + h % 16], 0, 0)
    # another point: cell (1,2,3) at level 1: x = (k-0.5)/7
    x = np.array([[(1 - 0.5) / 7, (2 - 0.5) / 7, (3 - 0.5) / 7]], np.float32)
    h = (1 * 1) ^ ((2 * 2654435761) & 0xFFFFFFFF) ^ ((3 * 805459861) & 0xFFFFFFFF)
    y = oracle.grid_fwd(d, table, x)
    close(y[0, 1], 8 + h % 16, 1e-5, 1e-4)


def test_dense_level_reproduces_affine_functions():
    """A dense level interpolating f(cell) = a.cell + b reproduces a.(x*scale+0.5) + b exactly
    (SURVEY.md §8(c) KAT 1): pins the +0.5 offset, corner weights and dense index order."""
    d, n = oracle.grid_layout(1, 2, 19, 16, 2.0)   # res 16, 4096 dense rows
    assert d.resolution[0] == 16 and n == 4096 * 2
    a = np.array([0.3, -1.1, 0.7])
    table = np.zeros((4096, 2), np.float32)
    ix = np.arange(4096)
    cx, cy, cz = ix % 16, (ix // 16) % 16, ix // 256
    table[:, 0] = a[0] * cx + a[1] * cy + a[2] * cz + 0.25
    table[:, 1] = cx
    x = (rng(12).random((500, 3)) * 0.9 + 0.02).astype(np.float32)
    y = oracle.grid_fwd(d, table.reshape(-1), x)
    pos = x.astype(np.float64) * 15 + 0.5
    close(y[:, 0], pos @ a + 0.25, 1e-5, 1e-5)
    close(y[:, 1], pos[:, 0], 1e-5, 1e-5)
    # input gradient of an affine field is constant: a * scale
    gx = oracle.grid_bwd_input(d, table.reshape(-1), x, np.tile(np.array([[1.0, 0.0]], np.float32), (500, 1)))
    close(gx, np.tile(a * 15, (500, 1)), 1e-4, 1e-4)


def test_grid_gradients_match_finite_differences():
    d, n = oracle.grid_layout(4, 2, 8, 4, 1.7)
    g = rng(13)
    table = g.normal(size=n).astype(np.float32)
    x = (g.random((40, 3)) * 0.96 + 0.02).astype(np.float32)
    dy = g.normal(size=(40, 8)).astype(np.float32)

    def L(tab, xx):
        return float((oracle.grid_fwd(d, tab, xx).astype(np.float64) * dy).sum())

    gp = oracle.grid_bwd_param(d, x, dy, n)
    for k in g.choice(n, 15, replace=False):
        tp, tm = table.copy(), table.copy()
        tp[k] += 0.5
        tm[k] -= 0.5
        close(gp[k], L(tp, x) - L(tm, x), 1e-3, 1e-3)   # L is linear in the table
    gx = oracle.grid_bwd_input(d, table, x, dy)
    eps = 1e-4                                           # piecewise trilinear: stay inside the cell
    for i in range(0, 40, 5):
        for k in range(3):
            xp, xm = x.copy(), x.copy()
            xp[i, k] += eps
            xm[i, k] -= eps
            fd = (L(table, xp) - L(table, xm)) / (2 * eps)
            assert abs(fd - gx[i, k]) < 5e-2 * max(1.0, abs(fd)), (i, k, fd, gx[i, k])
    # double backward: <v, dL_dx> is bilinear in (table, dy)
    v = g.normal(size=(40, 3)).astype(np.float32)
    d_tab, d_dy = oracle.grid_bwd_bwd_input(d, table, x, dy, v)
    close((d_dy.astype(np.float64) * dy).sum(), (gx.astype(np.float64) * v).sum(), 1e-4, 1e-4)
    close((d_tab.astype(np.float64) * table).sum(), (gx.astype(np.float64) * v).sum(), 1e-4, 1e-4)


def test_sh_basis_is_orthonormal():
    g = rng(14)
    v = g.normal(size=(200000, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    y = oracle.sh_fwd(((v + 1) / 2).astype(np.float32), 4).astype(np.float64)
    gram = (y.T @ y) / len(v) * 4 * np.pi
    close(gram, np.eye(16), 0, 3e-2)
    close(y[:, 0], 0.28209479177387814, 1e-6, 0)
    assert oracle.sh_fwd(((v[:5] + 1) / 2).astype(np.float32), 3).shape == (5, 9)


def test_linear_layer_matches_numpy():
    g = rng(15)
    x = g.normal(size=(33, 20)).astype(np.float32)
    W = g.normal(size=(7, 20)).astype(np.float32)
    b = g.normal(size=7).astype(np.float32)
    z = x.astype(np.float64) @ W.T + b
    close(oracle.linear_fwd(x, W, b, "None"), z, 1e-5, 1e-5)
    close(oracle.linear_fwd(x, W, b, "ReLU"), np.maximum(z, 0), 1e-5, 1e-5)
    close(oracle.linear_fwd(x, W, b, "Sigmoid"), 1 / (1 + np.exp(-z)), 1e-5, 1e-5)
    close(oracle.linear_fwd(x, W, b, "Softplus"), np.log1p(np.exp(z)), 1e-5, 1e-5)
    close(oracle.linear_fwd(x, W, None, "Exponential"), np.exp(z - b), 1e-5, 1e-5)


def test_fused_cpu_field_matches_layered_field():
    """oracle.field_forward (the whole field for a batch of points in one C call, OpenMP over the points: what bench.py's
    cpu_baseline times) against oracle/field.py::CpuNGP (pinned to the reference's NGP by fixture G6)."""
    from oracle.field import CpuNGP
    L, F = 16, 8
    b = float(np.exp(np.log(2048 * 0.5 / 16) / 15))
    _, nx = oracle.grid_layout(L, F, 19, 16, b)
    _, nr = oracle.grid_layout(L, F, 21, 16, b)
    g = np.random.default_rng(5)
    st = {"xyz_encoder.params": g.uniform(-0.1, 0.1, nx).astype(np.float32),
          "rgb_encoder.params": g.uniform(-0.1, 0.1, nr).astype(np.float32),
          "xyz_net.0.weight": (g.normal(size=(128, 128)) * 0.1).astype(np.float32),
          "xyz_net.0.bias": (g.normal(size=128) * 0.1).astype(np.float32),
          "xyz_net.2.weight": (g.normal(size=(1, 128)) * 0.1).astype(np.float32),
          "xyz_net.2.bias": np.full(1, 0.3, np.float32),
          "rgb_net.params": (g.normal(size=128 * 144 + 16 * 128) * 0.1).astype(np.float32),
          "norm_pred_header.params": (g.normal(size=32 * 128 + 16 * 32) * 0.1).astype(np.float32),
          "semantic_header.params": (g.normal(size=32 * 128 + 16 * 32) * 0.1).astype(np.float32)}
    f = CpuNGP(st, scale=0.5)
    n = 700
    x = (g.random((n, 3)) - 0.5).astype(np.float32)
    x[0] = [-0.5, -0.5, -0.5]
    x[1] = [0.5, 0.5, 0.5]
    d = g.normal(size=(n, 3)).astype(np.float32)
    ref = f(x, d)
    got = oracle.field_forward(f, x, d)
    for a, c, name in zip(ref[:5], got, ("sigmas", "rgbs", "normals_raw", "normals_pred", "sems")):
        assert a.shape == c.shape, name
        np.testing.assert_allclose(c, a, rtol=2e-5, atol=2e-6, err_msg=name)
