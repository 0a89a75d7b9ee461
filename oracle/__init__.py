"""CPU oracle for the instant-NGP hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  It wraps oracle/ngp_oracle.c (a scalar C restatement of the reference's
kernels, see that file's header for the file:line each function follows and for the
pinning status) with numpy-in / numpy-out functions named after the reference's
`vren` entry points.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libngp_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "ngp_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        # NGP_ORACLE_SO: a sanitizer build of ngp_oracle.c (oracle/Makefile `asan`), CPU tests only
        _lib = C.CDLL(os.environ.get("NGP_ORACLE_SO") or build())
        _lib.ngp_cpu_grid_layout.restype = C.c_int64
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


f32 = C.c_float
i32 = C.c_int


# ---- intersections -------------------------------------------------------------------
def ray_aabb_intersect(rays_o, rays_d, centers, half_sizes, max_hits):
    rays_o, rays_d, centers, half_sizes = map(_f, (rays_o, rays_d, centers, half_sizes))
    n, v = len(rays_o), len(centers)
    cnt = np.zeros(n, np.int32)
    t = np.zeros((n, max_hits, 2), np.float32)
    idx = np.zeros((n, max_hits), np.int64)
    lib().ngp_cpu_ray_aabb_intersect(_p(rays_o), _p(rays_d), _p(centers), _p(half_sizes), n, v, max_hits,
                                     _p(cnt), _p(t), _p(idx))
    return cnt, t, idx


def ray_sphere_intersect(rays_o, rays_d, centers, radii, max_hits):
    rays_o, rays_d, centers, radii = map(_f, (rays_o, rays_d, centers, radii))
    n, v = len(rays_o), len(centers)
    cnt = np.zeros(n, np.int32)
    t = np.zeros((n, max_hits, 2), np.float32)
    idx = np.zeros((n, max_hits), np.int64)
    lib().ngp_cpu_ray_sphere_intersect(_p(rays_o), _p(rays_d), _p(centers), _p(radii), n, v, max_hits,
                                       _p(cnt), _p(t), _p(idx))
    return cnt, t, idx


# ---- occupancy grid helpers ----------------------------------------------------------
def morton3D(coords):
    coords = np.ascontiguousarray(coords, np.int32)
    out = np.zeros(len(coords), np.int32)
    lib().ngp_cpu_morton3D(_p(coords), len(coords), _p(out))
    return out


def morton3D_invert(indices):
    indices = np.ascontiguousarray(indices, np.int32)
    out = np.zeros((len(indices), 3), np.int32)
    lib().ngp_cpu_morton3D_invert(_p(indices), len(indices), _p(out))
    return out


def packbits(density_grid, thr, out=None):
    g = _f(density_grid).reshape(-1)
    nb = g.size // 8
    if out is None:
        out = np.zeros(nb, np.uint8)
    lib().ngp_cpu_packbits(_p(g), nb, f32(thr), _p(out))
    return out


def grid_cell_points(coords, noise, grid_size, s):
    coords = np.ascontiguousarray(coords, np.int32)
    noise = _f(noise)
    out = np.zeros((len(coords), 3), np.float32)
    lib().ngp_cpu_grid_cell_points(_p(coords), _p(noise), len(coords), grid_size, f32(s), _p(out))
    return out


def density_grid_ema(grid, tmp, decay):
    grid = _f(grid).copy()
    tmp = _f(tmp)
    lib().ngp_cpu_density_grid_ema(_p(grid), _p(tmp), grid.size, f32(decay))
    return grid


# ---- marchers ------------------------------------------------------------------------
def raymarching_train(rays_o, rays_d, hits_t, bitfield, cascades, scale, exp_step_factor, noise,
                      grid_size, max_samples):
    rays_o, rays_d, hits_t, noise = map(_f, (rays_o, rays_d, hits_t, noise))
    bitfield = np.ascontiguousarray(bitfield, np.uint8)
    n = len(rays_o)
    cap = n * max_samples
    rays_a = np.zeros((n, 3), np.int64)
    xyzs = np.zeros((cap, 3), np.float32)
    dirs = np.zeros((cap, 3), np.float32)
    deltas = np.zeros(cap, np.float32)
    ts = np.zeros(cap, np.float32)
    counter = np.zeros(2, np.int32)
    rc = lib().ngp_cpu_raymarching_train(_p(rays_o), _p(rays_d), _p(hits_t), _p(bitfield), cascades, f32(scale),
                                         f32(exp_step_factor), _p(noise), grid_size, max_samples, n,
                                         _p(rays_a), _p(xyzs), _p(dirs), _p(deltas), _p(ts), _p(counter))
    assert rc == 0
    return rays_a, xyzs, dirs, deltas, ts, counter


def raymarching_test(rays_o, rays_d, hits_t, alive_indices, bitfield, cascades, scale, exp_step_factor,
                     grid_size, max_samples, n_samples):
    """hits_t is updated IN PLACE (must be a contiguous float32 array)."""
    rays_o, rays_d = map(_f, (rays_o, rays_d))
    assert hits_t.dtype == np.float32 and hits_t.flags.c_contiguous
    alive = np.ascontiguousarray(alive_indices, np.int64)
    bitfield = np.ascontiguousarray(bitfield, np.uint8)
    na = len(alive)
    xyzs = np.zeros((na, n_samples, 3), np.float32)
    dirs = np.zeros((na, n_samples, 3), np.float32)
    deltas = np.zeros((na, n_samples), np.float32)
    ts = np.zeros((na, n_samples), np.float32)
    n_eff = np.zeros(na, np.int32)
    lib().ngp_cpu_raymarching_test(_p(rays_o), _p(rays_d), _p(hits_t), _p(alive), _p(bitfield), cascades,
                                   f32(scale), f32(exp_step_factor), grid_size, max_samples, n_samples, na,
                                   _p(xyzs), _p(dirs), _p(deltas), _p(ts), _p(n_eff))
    return xyzs, dirs, deltas, ts, n_eff


# ---- compositing ---------------------------------------------------------------------
def composite_alpha_fw(sigmas, deltas, rays_a, T_thr):
    sigmas, deltas = map(_f, (sigmas, deltas))
    rays_a = np.ascontiguousarray(rays_a, np.int64)
    alphas = np.zeros_like(sigmas)
    ws = np.zeros_like(sigmas)
    lib().ngp_cpu_composite_alpha_fw(_p(sigmas), _p(deltas), _p(rays_a), f32(T_thr), len(rays_a), _p(alphas), _p(ws))
    return alphas, ws


def composite_train_fw(sigmas, rgbs, normals_pred, sems, deltas, ts, rays_a, T_thr, classes):
    sigmas, rgbs, normals_pred, sems, deltas, ts = map(_f, (sigmas, rgbs, normals_pred, sems, deltas, ts))
    rays_a = np.ascontiguousarray(rays_a, np.int64)
    nr, N = len(rays_a), len(sigmas)
    total = np.zeros(nr, np.int64)
    opacity = np.zeros(nr, np.float32)
    depth = np.zeros(nr, np.float32)
    rgb = np.zeros((nr, 3), np.float32)
    normal = np.zeros((nr, 3), np.float32)
    sem = np.zeros((nr, classes), np.float32)
    ws = np.zeros(N, np.float32)
    lib().ngp_cpu_composite_train_fw(_p(sigmas), _p(rgbs), _p(normals_pred), _p(sems), _p(deltas), _p(ts),
                                     _p(rays_a), f32(T_thr), classes, nr, _p(total), _p(opacity), _p(depth),
                                     _p(rgb), _p(normal), _p(sem), _p(ws))
    return total, opacity, depth, rgb, normal, sem, ws


def composite_train_bw(dL_dopacity, dL_ddepth, dL_drgb, dL_dnormal_pred, dL_dsem, dL_dws, sigmas, rgbs,
                       normals_pred, ws, deltas, ts, rays_a, opacity, depth, rgb, normal_pred, T_thr, classes):
    (dL_dopacity, dL_ddepth, dL_drgb, dL_dnormal_pred, dL_dsem, dL_dws, sigmas, rgbs, normals_pred, ws, deltas, ts,
     opacity, depth, rgb, normal_pred) = map(_f, (dL_dopacity, dL_ddepth, dL_drgb, dL_dnormal_pred, dL_dsem, dL_dws,
                                                  sigmas, rgbs, normals_pred, ws, deltas, ts, opacity, depth, rgb,
                                                  normal_pred))
    rays_a = np.ascontiguousarray(rays_a, np.int64)
    N = len(sigmas)
    dsig = np.zeros(N, np.float32)
    drgbs = np.zeros((N, 3), np.float32)
    dnrm = np.zeros((N, 3), np.float32)
    dsem = np.zeros((N, classes), np.float32)
    lib().ngp_cpu_composite_train_bw(_p(dL_dopacity), _p(dL_ddepth), _p(dL_drgb), _p(dL_dnormal_pred), _p(dL_dsem),
                                     _p(dL_dws), _p(sigmas), _p(rgbs), _p(normals_pred), _p(ws), _p(deltas), _p(ts),
                                     _p(rays_a), _p(opacity), _p(depth), _p(rgb), _p(normal_pred), f32(T_thr),
                                     classes, len(rays_a), _p(dsig), _p(drgbs), _p(dnrm), _p(dsem))
    return dsig, drgbs, dnrm, dsem


def composite_test_fw(sigmas, rgbs, normals, normals_raw, sems, deltas, ts, hits_t, alive_indices, T_thr, classes,
                      n_eff, opacity, depth, rgb, normal, normal_raw, sem):
    """opacity/depth/rgb/normal/normal_raw/sem/alive_indices are updated IN PLACE."""
    sigmas, rgbs, normals, normals_raw, sems, deltas, ts, hits_t = map(
        _f, (sigmas, rgbs, normals, normals_raw, sems, deltas, ts, hits_t))
    n_eff = np.ascontiguousarray(n_eff, np.int32)
    for a in (opacity, depth, rgb, normal, normal_raw, sem):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    assert alive_indices.dtype == np.int64
    na, ns = sigmas.shape
    lib().ngp_cpu_composite_test_fw(_p(sigmas), _p(rgbs), _p(normals), _p(normals_raw), _p(sems), _p(deltas), _p(ts),
                                    _p(hits_t), _p(alive_indices), f32(T_thr), classes, _p(n_eff), na, ns,
                                    _p(opacity), _p(depth), _p(rgb), _p(normal), _p(normal_raw), _p(sem))


def composite_refloss_fw(sigmas, normals_diff, normals_ori, deltas, ts, rays_a, T_thr):
    sigmas, normals_diff, normals_ori, deltas, ts = map(_f, (sigmas, normals_diff, normals_ori, deltas, ts))
    rays_a = np.ascontiguousarray(rays_a, np.int64)
    nr = len(rays_a)
    lo = np.zeros(nr, np.float32)
    lp = np.zeros((nr, 3), np.float32)
    lib().ngp_cpu_composite_refloss_fw(_p(sigmas), _p(normals_diff), _p(normals_ori), _p(deltas), _p(ts), _p(rays_a),
                                       f32(T_thr), nr, _p(lo), _p(lp))
    return lo, lp


def composite_refloss_bw(dL_dloss_o, dL_dloss_p, sigmas, normals_diff, normals_ori, deltas, ts, rays_a, loss_o,
                         loss_p, T_thr):
    dL_dloss_o, dL_dloss_p, sigmas, normals_diff, normals_ori, deltas, ts, loss_o, loss_p = map(
        _f, (dL_dloss_o, dL_dloss_p, sigmas, normals_diff, normals_ori, deltas, ts, loss_o, loss_p))
    rays_a = np.ascontiguousarray(rays_a, np.int64)
    N = len(sigmas)
    ds = np.zeros(N, np.float32)
    dd = np.zeros((N, 3), np.float32)
    do = np.zeros(N, np.float32)
    lib().ngp_cpu_composite_refloss_bw(_p(dL_dloss_o), _p(dL_dloss_p), _p(sigmas), _p(normals_diff), _p(normals_ori),
                                       _p(deltas), _p(ts), _p(rays_a), _p(loss_o), _p(loss_p), f32(T_thr),
                                       len(rays_a), _p(ds), _p(dd), _p(do))
    return ds, dd, do


def distortion_loss_fw(ws, deltas, ts, rays_a):
    ws, deltas, ts = map(_f, (ws, deltas, ts))
    rays_a = np.ascontiguousarray(rays_a, np.int64)
    nr, N = len(rays_a), len(ws)
    loss = np.zeros(nr, np.float32)
    wi = np.zeros(N, np.float32)
    wti = np.zeros(N, np.float32)
    lib().ngp_cpu_distortion_loss_fw(_p(ws), _p(deltas), _p(ts), _p(rays_a), nr, _p(loss), _p(wi), _p(wti))
    return loss, wi, wti


def distortion_loss_bw(dL_dloss, ws_inc, wts_inc, ws, deltas, ts, rays_a):
    dL_dloss, ws_inc, wts_inc, ws, deltas, ts = map(_f, (dL_dloss, ws_inc, wts_inc, ws, deltas, ts))
    rays_a = np.ascontiguousarray(rays_a, np.int64)
    out = np.zeros(len(ws), np.float32)
    lib().ngp_cpu_distortion_loss_bw(_p(dL_dloss), _p(ws_inc), _p(wts_inc), _p(ws), _p(deltas), _p(ts), _p(rays_a),
                                     len(rays_a), _p(out))
    return out


def segment_csr_sum(src, indptr):
    src = _f(src)
    indptr = np.ascontiguousarray(indptr, np.int64)
    w = src.shape[1] if src.ndim > 1 else 1
    out = np.zeros((len(indptr) - 1, w), np.float32)
    lib().ngp_cpu_segment_csr_sum(_p(src), _p(indptr), len(indptr) - 1, w, _p(out))
    return out if src.ndim > 1 else out[:, 0]


# ---- tiny-cuda-nn pieces -------------------------------------------------------------
class GridDesc(C.Structure):
    _fields_ = [("n_levels", C.c_uint32), ("n_features", C.c_uint32),
                ("offsets", C.c_uint32 * 33), ("resolution", C.c_uint32 * 32), ("scale", C.c_float * 32)]


def grid_layout(n_levels, n_features, log2_hashmap_size, base_resolution, per_level_scale):
    d = GridDesc()
    n = lib().ngp_cpu_grid_layout(n_levels, n_features, log2_hashmap_size, base_resolution,
                                  C.c_double(per_level_scale), C.byref(d))
    assert n > 0
    return d, int(n)


def grid_fwd(desc, table, x):
    table, x = _f(table), _f(x)
    y = np.zeros((len(x), desc.n_levels * desc.n_features), np.float32)
    lib().ngp_cpu_grid_fwd(C.byref(desc), _p(table), _p(x), C.c_int64(len(x)), _p(y))
    return y


def grid_bwd_param(desc, x, dL_dy, n_params):
    x, dL_dy = _f(x), _f(dL_dy)
    g = np.zeros(n_params, np.float32)
    lib().ngp_cpu_grid_bwd_param(C.byref(desc), _p(x), _p(dL_dy), C.c_int64(len(x)), _p(g))
    return g


def grid_bwd_input(desc, table, x, dL_dy):
    table, x, dL_dy = _f(table), _f(x), _f(dL_dy)
    out = np.zeros((len(x), 3), np.float32)
    lib().ngp_cpu_grid_bwd_input(C.byref(desc), _p(table), _p(x), _p(dL_dy), C.c_int64(len(x)), _p(out))
    return out


def grid_bwd_bwd_input(desc, table, x, dL_dy, v):
    table, x, dL_dy, v = _f(table), _f(x), _f(dL_dy), _f(v)
    g = np.zeros(table.size, np.float32)
    ddy = np.zeros_like(dL_dy)
    lib().ngp_cpu_grid_bwd_bwd_input(C.byref(desc), _p(table), _p(x), _p(dL_dy), _p(v), C.c_int64(len(x)), _p(g), _p(ddy))
    return g, ddy


def sh_fwd(x, degree):
    x = _f(x)
    y = np.zeros((len(x), degree * degree), np.float32)
    lib().ngp_cpu_sh_fwd(_p(x), C.c_int64(len(x)), degree, _p(y))
    return y


ACT = {"None": 0, "ReLU": 1, "Sigmoid": 2, "Softplus": 3, "Exponential": 4}


def linear_fwd(x, W, b, act):
    x, W = _f(x), _f(W)
    b = _f(b) if b is not None else None
    n, n_in = x.shape
    n_out = W.shape[0]
    y = np.zeros((n, n_out), np.float32)
    lib().ngp_cpu_linear_fwd(_p(x), C.c_int64(n_in), _p(W), _p(b), C.c_int64(n), n_in, n_out,
                             ACT[act] if isinstance(act, str) else act, _p(y), C.c_int64(n_out))
    return y


def field_forward(field, x, d):
    """the whole NGP field of `field` (an oracle.field.CpuNGP without appearance codes) on points x (N,3) and
    directions d (N,3) in ONE C call, OpenMP over the points -> (sigmas, rgbs, normals_raw, normals_pred, sems)"""
    x, d = _f(x), _f(d)
    n = len(x)
    C7 = int(field.classes)
    sig = np.zeros(n, np.float32)
    rgb = np.zeros((n, 3), np.float32)
    nr = np.zeros((n, 3), np.float32)
    npd = np.zeros((n, 3), np.float32)
    sem = np.zeros((n, C7), np.float32)
    w = [_f(a) for a in (field.W1, field.b1, field.W2, field.b2, field.Wr1, field.Wr2, field.Wn1, field.Wn2, field.Ws1,
                         field.Ws2)]
    rc = lib().ngp_cpu_field_forward(C.byref(field.xyz_desc), _p(_f(field.xyz_table)), C.byref(field.rgb_desc),
                                     _p(_f(field.rgb_table)), _p(w[0]), _p(w[1]), _p(w[2]), _p(w[3]), _p(w[4]),
                                     int(field.Wr1.shape[1]), _p(w[5]), _p(w[6]), _p(w[7]), _p(w[8]), _p(w[9]),
                                     f32(field.scale), C7, _p(x), _p(d), C.c_int64(n), _p(sig), _p(rgb), _p(nr), _p(npd),
                                     _p(sem))
    if rc != 0:
        raise RuntimeError(f"ngp_cpu_field_forward failed: {rc}")
    return sig, rgb, nr, npd, sem


def num_threads():
    return int(lib().ngp_cpu_num_threads())


def set_num_threads(n):
    """OpenMP threads of the C restatement from now on (bench.py: the CPU baseline at 1 thread and at all of them)"""
    lib().ngp_cpu_set_num_threads(int(n))
