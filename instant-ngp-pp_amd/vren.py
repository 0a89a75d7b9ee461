"""`vren`-shaped module: the 15 entry points of the reference's CUDA extension
(models/csrc/binding.cpp:323-342) with identical names, argument order, return values and
in-place behaviour, implemented on libngp_hip.so.

Inputs must be CUDA + contiguous tensors (RuntimeError otherwise, as CHECK_INPUT in
models/csrc/include/utils.h:4-6).  Kernels run on torch's CURRENT stream (the reference uses
the legacy default stream, §8(b) of SURVEY.md).
"""
import torch

from ._lib import call, check_input

_f32 = torch.float32


def _chk(**tensors):
    for k, v in tensors.items():
        check_input(v, k)


def ray_aabb_intersect(rays_o, rays_d, centers, half_sizes, max_hits):
    """binding.cpp:4-16 -> [hit_cnt (N) i32, hits_t (N,max_hits,2), hits_voxel_idx (N,max_hits) i64]"""
    _chk(rays_o=rays_o, rays_d=rays_d, centers=centers, half_sizes=half_sizes)
    n, v = rays_o.shape[0], centers.shape[0]
    dev = rays_o.device
    hit_cnt = torch.empty(n, dtype=torch.int32, device=dev)
    hits_t = torch.empty(n, max_hits, 2, dtype=_f32, device=dev)
    hits_idx = torch.empty(n, max_hits, dtype=torch.int64, device=dev)
    call("ray_aabb_intersect", rays_o, rays_d, centers, half_sizes, n, v, int(max_hits), hit_cnt, hits_t, hits_idx)
    return [hit_cnt, hits_t, hits_idx]


def ray_sphere_intersect(rays_o, rays_d, centers, radii, max_hits):
    """binding.cpp:19-31"""
    _chk(rays_o=rays_o, rays_d=rays_d, centers=centers, radii=radii)
    n, v = rays_o.shape[0], centers.shape[0]
    dev = rays_o.device
    hit_cnt = torch.empty(n, dtype=torch.int32, device=dev)
    hits_t = torch.empty(n, max_hits, 2, dtype=_f32, device=dev)
    hits_idx = torch.empty(n, max_hits, dtype=torch.int64, device=dev)
    call("ray_sphere_intersect", rays_o, rays_d, centers, radii, n, v, int(max_hits), hit_cnt, hits_t, hits_idx)
    return [hit_cnt, hits_t, hits_idx]


def morton3D(coords):
    """binding.cpp:74-79: (N,3) i32 -> (N) i32"""
    _chk(coords=coords)
    out = torch.empty(coords.shape[0], dtype=coords.dtype, device=coords.device)
    call("morton3D", coords, coords.shape[0], out)
    return out


def morton3D_invert(indices):
    """binding.cpp:82-87: (N) i32 -> (N,3) i32"""
    _chk(indices=indices)
    out = torch.empty(indices.shape[0], 3, dtype=indices.dtype, device=indices.device)
    call("morton3D_invert", indices, indices.shape[0], out)
    return out


def packbits(density_grid, density_threshold, density_bitfield):
    """binding.cpp:90-101: writes density_bitfield in place, returns None"""
    _chk(density_grid=density_grid, density_bitfield=density_bitfield)
    call("packbits", density_grid, density_bitfield.shape[0], float(density_threshold), None, density_bitfield)


def _march_train(rays_o, rays_d, hits_t, density_bitfield, cascades, scale, exp_step_factor, noise, grid_size,
                 max_samples, zero_tail):
    _chk(rays_o=rays_o, rays_d=rays_d, hits_t=hits_t, density_bitfield=density_bitfield, noise=noise)
    n = rays_o.shape[0]
    dev = rays_o.device
    cap = n * max_samples
    rays_a = torch.empty(n, 3, dtype=torch.int64, device=dev)
    xyzs = torch.empty(cap, 3, dtype=_f32, device=dev)
    dirs = torch.empty(cap, 3, dtype=_f32, device=dev)
    deltas = torch.empty(cap, dtype=_f32, device=dev)
    ts = torch.empty(cap, dtype=_f32, device=dev)
    counter = torch.empty(2, dtype=torch.int32, device=dev)
    t_scratch = torch.empty(max(cap, 1), dtype=_f32, device=dev)
    counts = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    call("raymarching_train", rays_o, rays_d, hits_t, density_bitfield, int(cascades), float(scale),
         float(exp_step_factor), noise, int(grid_size), int(max_samples), n, t_scratch, counts, rays_a, xyzs, dirs,
         deltas, ts, counter, cap, int(zero_tail))
    return [rays_a, xyzs, dirs, deltas, ts, counter]


def raymarching_train(rays_o, rays_d, hits_t, density_bitfield, cascades, scale, exp_step_factor, noise, grid_size,
                      max_samples):
    """binding.cpp:104-131 -> [rays_a, xyzs, dirs, deltas, ts, counter], outputs sized
    N_rays*max_samples with zeros behind counter[0] exactly like the reference's torch::zeros
    buffers (raymarching.cu:298-305)."""
    return _march_train(rays_o, rays_d, hits_t, density_bitfield, cascades, scale, exp_step_factor, noise, grid_size,
                        max_samples, zero_tail=True)


def raymarching_train_untrimmed(rays_o, rays_d, hits_t, density_bitfield, cascades, scale, exp_step_factor, noise,
                                grid_size, max_samples):
    """Same, but rows behind counter[0] are left uninitialised (RayMarcher.forward slices them
    off immediately, custom_functions.py:93-98, so the 268 MB zero-fill is skipped)."""
    return _march_train(rays_o, rays_d, hits_t, density_bitfield, cascades, scale, exp_step_factor, noise, grid_size,
                        max_samples, zero_tail=False)


def raymarching_test(rays_o, rays_d, hits_t, alive_indices, density_bitfield, cascades, scale, exp_step_factor,
                     grid_size, max_samples, N_samples):
    """binding.cpp:134-163; hits_t[:,0] is advanced in place."""
    _chk(rays_o=rays_o, rays_d=rays_d, hits_t=hits_t, alive_indices=alive_indices, density_bitfield=density_bitfield)
    na = alive_indices.shape[0]
    dev = rays_o.device
    xyzs = torch.zeros(na, N_samples, 3, dtype=_f32, device=dev)
    dirs = torch.zeros(na, N_samples, 3, dtype=_f32, device=dev)
    deltas = torch.zeros(na, N_samples, dtype=_f32, device=dev)
    ts = torch.zeros(na, N_samples, dtype=_f32, device=dev)
    n_eff = torch.zeros(na, dtype=torch.int32, device=dev)
    call("raymarching_test", rays_o, rays_d, hits_t, alive_indices, density_bitfield, int(cascades), float(scale),
         float(exp_step_factor), int(grid_size), int(max_samples), int(N_samples), na, xyzs, dirs, deltas, ts, n_eff)
    return [xyzs, dirs, deltas, ts, n_eff]


def composite_alpha_fw(sigmas, deltas, rays_a, T_threshold):
    """binding.cpp:166-180 -> [alphas, ws]"""
    _chk(sigmas=sigmas, deltas=deltas, rays_a=rays_a)
    alphas = torch.zeros_like(sigmas)  # rows no ray covers stay zero (reference: torch::zeros)
    ws = torch.zeros_like(sigmas)
    call("composite_alpha_fw", sigmas, deltas, rays_a, float(T_threshold), rays_a.shape[0], alphas, ws)
    return [alphas, ws]


def composite_train_fw(sigmas, rgbs, normals_pred, sems, deltas, ts, rays_a, T_threshold, classes):
    """binding.cpp:183-208 -> [total_samples, opacity, depth, rgb, normal_pred, sem, ws]"""
    _chk(sigmas=sigmas, rgbs=rgbs, normals_pred=normals_pred, sems=sems, deltas=deltas, ts=ts, rays_a=rays_a)
    nr, N = rays_a.shape[0], sigmas.shape[0]
    dev = sigmas.device
    total = torch.zeros(nr, dtype=torch.int64, device=dev)   # torch::zeros, volumerendering.cu:137-143
    opacity = torch.zeros(nr, dtype=_f32, device=dev)
    depth = torch.zeros(nr, dtype=_f32, device=dev)
    rgb = torch.zeros(nr, 3, dtype=_f32, device=dev)
    normal = torch.zeros(nr, 3, dtype=_f32, device=dev)
    sem = torch.zeros(nr, classes, dtype=_f32, device=dev)
    ws = torch.zeros(N, dtype=_f32, device=dev)
    call("composite_train_fw", sigmas, rgbs, normals_pred, sems, deltas, ts, rays_a, float(T_threshold), int(classes),
         nr, total, opacity, depth, rgb, normal, sem, ws)
    return [total, opacity, depth, rgb, normal, sem, ws]


def composite_train_bw(dL_dopacity, dL_ddepth, dL_drgb, dL_dnormal_pred, dL_dsem, dL_dws, sigmas, rgbs, normals_pred,
                       ws, deltas, ts, rays_a, opacity, depth, rgb, normal_pred, T_threshold, classes):
    """binding.cpp:211-259 -> [dL_dsigmas, dL_drgbs, dL_dnormals_pred, dL_dsems]"""
    _chk(dL_dopacity=dL_dopacity, dL_ddepth=dL_ddepth, dL_drgb=dL_drgb, dL_dnormal_pred=dL_dnormal_pred,
         dL_dsem=dL_dsem, dL_dws=dL_dws, sigmas=sigmas, rgbs=rgbs, normals_pred=normals_pred, ws=ws, deltas=deltas,
         ts=ts, rays_a=rays_a, opacity=opacity, depth=depth, rgb=rgb, normal_pred=normal_pred)
    N = sigmas.shape[0]
    dev = sigmas.device
    d_sig = torch.zeros(N, dtype=_f32, device=dev)
    d_rgbs = torch.zeros(N, 3, dtype=_f32, device=dev)
    d_nrm = torch.zeros(N, 3, dtype=_f32, device=dev)
    d_sems = torch.zeros(N, classes, dtype=_f32, device=dev)
    call("composite_train_bw", dL_dopacity, dL_ddepth, dL_drgb, dL_dnormal_pred, dL_dsem, dL_dws, sigmas, rgbs,
         normals_pred, ws, deltas, ts, rays_a, opacity, depth, rgb, normal_pred, float(T_threshold), int(classes),
         rays_a.shape[0], d_sig, d_rgbs, d_nrm, d_sems)
    return [d_sig, d_rgbs, d_nrm, d_sems]


def composite_test_fw(sigmas, rgbs, normals, normals_raw, sems, deltas, ts, hits_t, alive_indices, T_threshold,
                      classes, N_eff_samples, opacity, depth, rgb, normal, normal_raw, sem):
    """binding.cpp:262-320; opacity/depth/rgb/normal/normal_raw/sem/alive_indices updated in place."""
    _chk(sigmas=sigmas, rgbs=rgbs, normals=normals, normals_raw=normals_raw, sems=sems, deltas=deltas, ts=ts,
         hits_t=hits_t, alive_indices=alive_indices, N_eff_samples=N_eff_samples, opacity=opacity, depth=depth,
         rgb=rgb, normal=normal, normal_raw=normal_raw, sem=sem)
    call("composite_test_fw", sigmas, rgbs, normals, normals_raw, sems, deltas, ts, hits_t, alive_indices,
         float(T_threshold), int(classes), N_eff_samples, sigmas.shape[0], sigmas.shape[1], opacity, depth, rgb,
         normal, normal_raw, sem)


def composite_refloss_fw(sigmas, normals_diff, normals_ori, deltas, ts, rays_a, T_threshold):
    """ref_loss.cu:41-73 -> [loss_o (N_rays), loss_p (N_rays,3)]"""
    _chk(sigmas=sigmas, normals_diff=normals_diff, normals_ori=normals_ori, deltas=deltas, ts=ts, rays_a=rays_a)
    nr = rays_a.shape[0]
    lo = torch.empty(nr, dtype=_f32, device=sigmas.device)
    lp = torch.empty(nr, 3, dtype=_f32, device=sigmas.device)
    call("composite_refloss_fw", sigmas, normals_diff, normals_ori, deltas, ts, rays_a, float(T_threshold), nr, lo, lp)
    return [lo, lp]


def composite_refloss_bw(dL_dloss_o, dL_dloss_p, sigmas, normals_diff, normals_ori, deltas, ts, rays_a, loss_o,
                         loss_p, T_threshold):
    """ref_loss.cu:133-175 -> [dL_dsigmas, dL_dnormals_diff, dL_dnormals_ori]"""
    _chk(dL_dloss_o=dL_dloss_o, dL_dloss_p=dL_dloss_p, sigmas=sigmas, normals_diff=normals_diff,
         normals_ori=normals_ori, deltas=deltas, ts=ts, rays_a=rays_a, loss_o=loss_o, loss_p=loss_p)
    N = sigmas.shape[0]
    dev = sigmas.device
    ds = torch.zeros(N, dtype=_f32, device=dev)
    dd = torch.zeros(N, 3, dtype=_f32, device=dev)
    do = torch.zeros(N, dtype=_f32, device=dev)
    call("composite_refloss_bw", dL_dloss_o, dL_dloss_p, sigmas, normals_diff, normals_ori, deltas, ts, rays_a,
         loss_o, loss_p, float(T_threshold), rays_a.shape[0], ds, dd, do)
    return [ds, dd, do]


def distortion_loss_fw(ws, deltas, ts, rays_a):
    """losses.cu:62-107 -> [loss (N_rays), ws_inclusive_scan (N), wts_inclusive_scan (N)]"""
    _chk(ws=ws, deltas=deltas, ts=ts, rays_a=rays_a)
    nr, N = rays_a.shape[0], ws.shape[0]
    loss = torch.empty(nr, dtype=_f32, device=ws.device)
    wi = torch.zeros(N, dtype=_f32, device=ws.device)
    wti = torch.zeros(N, dtype=_f32, device=ws.device)
    call("distortion_loss_fw", ws, deltas, ts, rays_a, nr, loss, wi, wti)
    return [loss, wi, wti]


def distortion_loss_bw(dL_dloss, ws_inclusive_scan, wts_inclusive_scan, ws, deltas, ts, rays_a):
    """losses.cu:143-173 -> dL_dws (N)"""
    _chk(dL_dloss=dL_dloss, ws_inclusive_scan=ws_inclusive_scan, wts_inclusive_scan=wts_inclusive_scan, ws=ws,
         deltas=deltas, ts=ts, rays_a=rays_a)
    out = torch.zeros(ws.shape[0], dtype=_f32, device=ws.device)
    call("distortion_loss_bw", dL_dloss, ws_inclusive_scan, wts_inclusive_scan, ws, deltas, ts, rays_a,
         rays_a.shape[0], out)
    return out
