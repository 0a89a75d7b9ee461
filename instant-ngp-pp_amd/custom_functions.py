"""Operator surface of the reference (models/custom_functions.py) — same class names, argument
order and return values — on the MI355X library.

RayAABBIntersector :9   RaySphereIntersector :33   RayMarcher :56   VolumeRenderer :117
RefLoss :165   TruncExp :200   ReLU :213   TruncTanh :231   sample_pdf :248   raw2outputs :280
"""
import torch

from . import vren
from ._lib import call
from .torch_scatter import segment_csr

_f32 = torch.float32


class RayAABBIntersector(torch.autograd.Function):
    """rays_o, rays_d (N_rays,3); center, half_size (N_voxels,3); max_hits
    -> hits_cnt (N_rays), hits_t (N_rays,max_hits,2) near->far (-1 = no hit), hits_voxel_idx"""

    @staticmethod
    def forward(ctx, rays_o, rays_d, center, half_size, max_hits):
        return tuple(vren.ray_aabb_intersect(rays_o, rays_d, center, half_size, max_hits))


class RaySphereIntersector(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rays_o, rays_d, center, radii, max_hits):
        return tuple(vren.ray_sphere_intersect(rays_o, rays_d, center, radii, max_hits))


class RayMarcher(torch.autograd.Function):
    """March the rays through the occupancy bitfield.
    -> rays_a (N_rays,3) [ray_idx, start_idx, N_samples], xyzs (N,3), dirs (N,3), deltas (N), ts (N),
       total_samples (0-dim int32 tensor)

    rays_a rows come out in ray order with monotone start indices (the reference's order is
    whatever its atomics produce, raymarching.cu:237-241), which also makes backward()'s CSR
    segments valid by construction."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, hits_t, density_bitfield, cascades, scale, exp_step_factor, grid_size,
                max_samples):
        noise = torch.rand_like(rays_o[:, 0])  # perturbs the first sample of each ray
        rays_a, xyzs, dirs, deltas, ts, counter = vren.raymarching_train_untrimmed(
            rays_o, rays_d, hits_t, density_bitfield, cascades, scale, exp_step_factor, noise, grid_size, max_samples)
        total_samples = counter[0]
        n = int(total_samples)  # the one host sync of the step (the reference has the same, §3.1)
        xyzs, dirs, deltas, ts = xyzs[:n], dirs[:n], deltas[:n], ts[:n]
        ctx.save_for_backward(rays_a, ts)
        return rays_a, xyzs, dirs, deltas, ts, total_samples

    @staticmethod
    def backward(ctx, dL_drays_a, dL_dxyzs, dL_ddirs, dL_ddeltas, dL_dts, dL_dtotal_samples):
        rays_a, ts = ctx.saved_tensors
        segments = torch.cat([rays_a[:, 1], rays_a[-1:, 1] + rays_a[-1:, 2]])
        dL_drays_o = segment_csr(dL_dxyzs.contiguous(), segments)
        dL_drays_d = segment_csr((dL_dxyzs * ts[:, None] + dL_ddirs).contiguous(), segments)
        return dL_drays_o, dL_drays_d, None, None, None, None, None, None, None


def mark_full_cover(rays_a):
    """Tags a rays_a tensor whose segments tile every sample row [0, N) and name every ray once — what the
    marcher produces.  VolumeRenderer then skips the zero-fill of its outputs (the kernels write every row);
    for any other rays_a (a subset of rays, trimmed segments) it zero-fills like the reference's
    torch::zeros (volumerendering.cu:137-143, 280-283)."""
    rays_a._ngp_full_cover = True
    return rays_a


def _out_alloc(rays_a):
    return torch.empty if getattr(rays_a, "_ngp_full_cover", False) else torch.zeros


class VolumeRenderer(torch.autograd.Function):
    """Front-to-back compositing with a variable number of samples per ray (training only).
    -> vr_samples (scalar), opacity (N_rays), depth (N_rays), rgb (N_rays,3), normal_pred (N_rays,3),
       sem (N_rays,classes), ws (N)"""

    @staticmethod
    def forward(ctx, sigmas, rgbs, normals_pred, sems, deltas, ts, rays_a, T_threshold, classes):
        nr, N = rays_a.shape[0], sigmas.shape[0]
        dev = sigmas.device
        new = _out_alloc(rays_a)   # empty when the marcher's rays_a guarantees that every row is written
        total = new(nr, dtype=torch.int64, device=dev)
        opacity = new(nr, dtype=_f32, device=dev)
        depth = new(nr, dtype=_f32, device=dev)
        rgb = new(nr, 3, dtype=_f32, device=dev)
        normal_pred = new(nr, 3, dtype=_f32, device=dev)
        sem = new(nr, classes, dtype=_f32, device=dev)
        ws = new(N, dtype=_f32, device=dev)
        ctx.full_cover = new is torch.empty
        call("composite_train_fw", sigmas, rgbs, normals_pred, sems, deltas, ts, rays_a, float(T_threshold),
             int(classes), nr, total, opacity, depth, rgb, normal_pred, sem, ws)
        ctx.save_for_backward(sigmas, rgbs, normals_pred, deltas, ts, rays_a, opacity, depth, rgb, normal_pred, ws)
        ctx.T_threshold = T_threshold
        ctx.classes = classes
        ctx.set_materialize_grads(False)  # outputs the loss never touched arrive as None, not zeros
        return total.sum(), opacity, depth, rgb, normal_pred, sem, ws

    @staticmethod
    def backward(ctx, dL_dtotal_samples, dL_dopacity, dL_ddepth, dL_drgb, dL_dnormal_pred, dL_dsem, dL_dws):
        sigmas, rgbs, normals_pred, deltas, ts, rays_a, opacity, depth, rgb, normal_pred, ws = ctx.saved_tensors
        N, classes = sigmas.shape[0], ctx.classes
        nr = rays_a.shape[0]
        dev = sigmas.device

        def z(t, *shape):   # the kernel reads a NULL upstream gradient as zeros
            return None if t is None else t.contiguous()

        # The reference back-propagates all-zero gradients through the normal / semantic heads when
        # the loss ignores those maps; here they are simply not produced (None) and the field's
        # backward skips the heads.
        new = torch.empty if ctx.full_cover else torch.zeros   # rows no segment covers get zero gradients
        d_sig = new(N, dtype=_f32, device=dev)
        d_rgbs = new(N, 3, dtype=_f32, device=dev)
        d_nrm = new(N, 3, dtype=_f32, device=dev) if dL_dnormal_pred is not None else None
        d_sems = new(N, classes, dtype=_f32, device=dev) if dL_dsem is not None else None
        if N == 0:   # no sample in the batch (every ray missed the occupied cells): nothing to propagate
            return d_sig, d_rgbs, d_nrm, d_sems, None, None, None, None, None
        call("composite_train_bw", z(dL_dopacity, nr), z(dL_ddepth, nr), z(dL_drgb, nr, 3),
             None if d_nrm is None else dL_dnormal_pred.contiguous(), None if d_sems is None else dL_dsem.contiguous(),
             z(dL_dws, N), sigmas, rgbs, normals_pred, ws, deltas, ts, rays_a, opacity, depth, rgb, normal_pred,
             float(ctx.T_threshold), int(classes), nr, d_sig, d_rgbs, d_nrm, d_sems)
        return d_sig, d_rgbs, d_nrm, d_sems, None, None, None, None, None


class RefLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sigmas, normals_diff, normals_ori, deltas, ts, rays_a, T_threshold):
        loss_o, loss_p = vren.composite_refloss_fw(sigmas, normals_diff, normals_ori, deltas, ts, rays_a, T_threshold)
        ctx.save_for_backward(sigmas, normals_diff, normals_ori, deltas, ts, rays_a, loss_o, loss_p)
        ctx.T_threshold = T_threshold
        return loss_o, loss_p

    @staticmethod
    def backward(ctx, dL_dloss_o, dL_dloss_p):
        sigmas, normals_diff, normals_ori, deltas, ts, rays_a, loss_o, loss_p = ctx.saved_tensors
        dL_dsigmas, dL_dnormals_diff, dL_dnormals_ori = vren.composite_refloss_bw(
            dL_dloss_o.contiguous(), dL_dloss_p.contiguous(), sigmas, normals_diff, normals_ori, deltas, ts, rays_a,
            loss_o, loss_p, ctx.T_threshold)
        # the reference discards dL_dsigmas here (custom_functions.py:198)
        return None, dL_dnormals_diff, dL_dnormals_ori, None, None, None, None


class TruncExp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, dL_dout):
        x = ctx.saved_tensors[0]
        return dL_dout * torch.exp(x.clamp(-7, 7))


class ReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        mask = x > 0
        ctx.save_for_backward(mask)
        return torch.where(mask, x, torch.zeros_like(x))

    @staticmethod
    def backward(ctx, dL_dout):
        (mask,) = ctx.saved_tensors
        # masked-out entries get 1e-6, as the reference (custom_functions.py:227)
        return torch.where(mask, dL_dout, torch.full_like(dL_dout, 1e-6))


class TruncTanh(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.tanh(x)

    @staticmethod
    def backward(ctx, dL_dout):
        x = ctx.saved_tensors[0]
        return dL_dout * (1 - torch.tanh(x.clamp(-15, 15)) ** 2)


def sample_pdf(bins, weights, N_samples, det=False, pytest=False):
    """Inverse-CDF (hierarchical) sampling of `N_samples` depths per ray from the piecewise-constant
    density `weights` (.., n) over the bin edges `bins` (.., n+1); det=True uses evenly spaced
    quantiles instead of uniform draws.  Same contract as custom_functions.py:248-278."""
    w = weights + 1e-5                                             # no empty bins
    cdf = torch.cumsum(w / w.sum(-1, keepdim=True), -1)
    cdf = torch.nn.functional.pad(cdf, (1, 0))                     # (.., n+1), starts at 0
    lead = list(cdf.shape[:-1])
    if det:
        u = torch.linspace(0., 1., steps=N_samples, device=cdf.device).expand(lead + [N_samples])
    else:
        u = torch.rand(lead + [N_samples], device=cdf.device)
    hi = torch.searchsorted(cdf, u.contiguous(), right=True)       # first edge whose cdf exceeds u
    lo = (hi - 1).clamp(min=0)
    hi = hi.clamp(max=cdf.shape[-1] - 1)
    c_lo, c_hi = cdf.gather(-1, lo), cdf.gather(-1, hi)
    b_lo, b_hi = bins.gather(-1, lo), bins.gather(-1, hi)
    span = c_hi - c_lo
    span = torch.where(span < 1e-5, torch.ones_like(span), span)   # flat stretch: stay on the lower edge
    return b_lo + (u - c_lo) / span * (b_hi - b_lo)


def raw2outputs(raw, z_vals, rays_d, raw_noise_std=0, classes=7):
    """Compositing of S dense samples per ray (custom_functions.py:280-321).
    raw (R, S, 10+classes) = [sigma | rgb 3 | normal_raw 3 | normal_pred 3 | semantic classes];
    the last interval is 1e10 long, so any density on the last sample closes the ray.
    -> opacity, rgb, normal_raw, normal_pred, semantic, weights (R,S), depth"""
    sigma, feats = raw[..., 0], raw[..., 1:]
    seg = torch.diff(z_vals, dim=-1)
    seg = torch.cat([seg, seg.new_full(seg[..., :1].shape, 1e10)], -1) * rays_d.norm(dim=-1, keepdim=True)
    if raw_noise_std > 0.:
        sigma = sigma + torch.randn_like(sigma) * raw_noise_std
    alpha = 1. - torch.exp(-sigma * seg)
    # transmittance in front of each sample: exclusive running product of (1 - alpha + 1e-10)
    trans = torch.cumprod(torch.nn.functional.pad(1. - alpha + 1e-10, (1, 0), value=1.0), -1)[..., :-1]
    weights = alpha * trans
    mixed = (weights[..., None] * feats).sum(-2)                   # every per-sample feature, composited
    opacity = weights.sum(-1)
    depth_map = (weights * z_vals).sum(-1)
    return opacity, mixed[..., 0:3], mixed[..., 3:6], mixed[..., 6:9], mixed[..., 9:], weights, depth_map
