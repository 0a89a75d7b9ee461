"""Loss terms of the reference (losses.py:7-151) on the MI355X operator surface."""
import math

import torch
import torch.nn.functional as F
from torch import nn

from . import vren


def compute_scale_and_shift(prediction, target):
    """Least-squares (scale, shift) with scale*prediction + shift ~ target (losses.py:7-30): the 2x2
    normal equations [[sum p^2, sum p], [sum p, n]] [s, t]^T = [sum p*target, sum target]^T solved by
    Cramer's rule; a singular system gives (0, 0)."""
    n = prediction.new_tensor(float(prediction.numel()))
    s_pp, s_p = (prediction * prediction).sum(), prediction.sum()
    s_pt, s_t = (prediction * target).sum(), target.sum()
    det = s_pp * n - s_p * s_p
    if det == 0:
        zero = prediction.new_zeros(())
        return zero, zero
    return (n * s_pt - s_p * s_t) / det, (s_pp * s_t - s_p * s_pt) / det


class DistortionLoss(torch.autograd.Function):
    """Mip-NeRF 360 distortion loss in DVGO-v2's prefix-sum form (losses.py:32-58).
    ws, deltas, ts (N); rays_a (N_rays,3) -> loss (N_rays)"""

    @staticmethod
    def forward(ctx, ws, deltas, ts, rays_a):
        loss, ws_inclusive_scan, wts_inclusive_scan = vren.distortion_loss_fw(ws.contiguous(), deltas, ts, rays_a)
        ctx.save_for_backward(ws_inclusive_scan, wts_inclusive_scan, ws, deltas, ts, rays_a)
        return loss

    @staticmethod
    def backward(ctx, dL_dloss):
        ws_inclusive_scan, wts_inclusive_scan, ws, deltas, ts, rays_a = ctx.saved_tensors
        dL_dws = vren.distortion_loss_bw(dL_dloss.contiguous(), ws_inclusive_scan, wts_inclusive_scan,
                                         ws.contiguous(), deltas, ts, rays_a)
        return dL_dws, None, None, None


_CONST = {}


def _const(value, n, device):
    """cached (n,) tensor filled with `value` (upstream gradient of a mean over the rays)"""
    key = (float(value), int(n), str(device))
    t = _CONST.get(key)
    if t is None:
        if len(_CONST) > 64:
            _CONST.clear()
        t = _CONST[key] = torch.full((n,), float(value), dtype=torch.float32, device=device)
    return t


@torch.no_grad()
def nerf_loss_and_grads(rgb, opacity, ws, deltas, ts, rays_a, target_rgb, lambda_opa, lambda_distortion):
    """sum(term.mean()) over NeRFLoss's default terms — rgb MSE, opacity entropy, distortion
    (losses.py:96-105, train.py:307) — together with its gradients, in four launches.
    -> terms (4) = [loss, rgb, opacity, distortion], (d_rgb (N_rays,3), d_opacity (N_rays), d_ws (N) or None)
    rays_a must cover every sample row (the marcher's output does)."""
    from ._lib import call
    nr, N = rgb.shape[0], ws.shape[0]
    dev = rgb.device
    rgb, opacity, ws = rgb.contiguous(), opacity.contiguous(), ws.contiguous()
    terms = torch.zeros(4, dtype=torch.float32, device=dev)
    d_rgb = torch.empty(nr, 3, dtype=torch.float32, device=dev)
    d_op = torch.empty(nr, dtype=torch.float32, device=dev)
    dist = d_ws = None
    if lambda_distortion > 0 and N == 0:
        # no sample in the whole batch (e.g. an empty occupancy grid): the distortion term and its gradient vanish
        dist = torch.zeros(nr, dtype=torch.float32, device=dev)
        d_ws = torch.empty(0, dtype=torch.float32, device=dev)
    elif lambda_distortion > 0:
        dist = torch.empty(nr, dtype=torch.float32, device=dev)
        wi = torch.empty(N, dtype=torch.float32, device=dev)
        wti = torch.empty(N, dtype=torch.float32, device=dev)
        call("distortion_loss_fw", ws, deltas, ts, rays_a, nr, dist, wi, wti)
        d_ws = torch.empty(N, dtype=torch.float32, device=dev)
        call("distortion_loss_bw", _const(lambda_distortion / nr, nr, dev), wi, wti, ws, deltas, ts, rays_a, nr, d_ws)
    call("nerf_loss", rgb, target_rgb.contiguous(), opacity, dist, nr, float(lambda_opa), float(lambda_distortion),
         terms, d_rgb, d_op)
    return terms, (d_rgb, d_op, d_ws)


class FusedNeRFLoss(torch.autograd.Function):
    """autograd wrapper of nerf_loss_and_grads.
    Returns (loss, rgb_mse_mean, opacity_mean, distortion_mean); only `loss` is differentiable."""

    @staticmethod
    def forward(ctx, rgb, opacity, ws, deltas, ts, rays_a, target_rgb, lambda_opa, lambda_distortion):
        terms, (d_rgb, d_op, d_ws) = nerf_loss_and_grads(rgb, opacity, ws, deltas, ts, rays_a, target_rgb, lambda_opa,
                                                         lambda_distortion)
        ctx.save_for_backward(d_rgb, d_op, d_ws)
        loss, t_rgb, t_op, t_dist = terms.unbind(0)
        ctx.mark_non_differentiable(t_rgb, t_op, t_dist)
        return loss, t_rgb, t_op, t_dist

    @staticmethod
    def backward(ctx, g, *_unused):
        d_rgb, d_op, d_ws = ctx.saved_tensors
        return d_rgb * g, d_op * g, (None if d_ws is None else d_ws * g), None, None, None, None, None, None


class ExponentialAnnealingWeight():
    def __init__(self, max, min, k):
        self.max, self.min, self.k = max, min, k

    def getWeight(self, Tcur):
        return max(self.min, self.max * math.exp(-Tcur * self.k))


class NeRFLoss(nn.Module):
    """Per-ray loss terms of the reference (losses.py:71-140) under the same dictionary keys and
    weights; the trainer reduces them with sum(term.mean()) (train.py:307).  Optional terms are
    switched on by the same keyword flags: embed_msk, normal_ref, normal_mono, semantic, depth_mono."""

    WEIGHTS = dict(lambda_opa=2e-4, lambda_distortion=3e-4, lambda_depth_mono=1, lambda_normal_mono=1e-3,
                   lambda_normal_ref_rp=1e-3, lambda_normal_ref_ro=1e-3, lambda_sky=1e-1, lambda_semantic=4e-2)

    def __init__(self):
        super().__init__()
        for name, value in self.WEIGHTS.items():
            setattr(self, name, value)
        self.Annealing = ExponentialAnnealingWeight(max=1, min=6e-2, k=1e-3)
        self.CrossEntropyLoss = nn.CrossEntropyLoss(ignore_index=256)

    # ---- the individual terms -------------------------------------------------------------
    @staticmethod
    def _colour(results, target, mask=None):
        err = torch.square(results['rgb'] - target['rgb'])
        return err if mask is None else (1 - mask) * err

    def _opacity_entropy(self, results):
        o = results['opacity'] + 1e-10            # -o log o: pushes the opacity of a ray towards 0 or 1
        return self.lambda_opa * (-o * torch.log(o))

    def _distortion(self, results):
        return self.lambda_distortion * DistortionLoss.apply(results['ws'], results['deltas'], results['ts'],
                                                             results['rays_a'])

    def _normal_mono(self, results, target):
        n_pred = F.normalize(results['normal_pred'], dim=-1)
        n_gt = F.normalize(target['normal'], dim=-1)
        return self.lambda_normal_mono * ((n_pred - n_gt).abs() - 0.1 * n_pred * n_gt)

    def _depth_mono(self, results, target, scene_scale):
        depth_2d = target['depth'] / 25
        valid = depth_2d > 0
        scale, shift = compute_scale_and_shift(results['depth'][valid].detach(), depth_2d[valid])
        falloff = torch.exp(-results['depth'].detach() / scene_scale)
        return valid.to(depth_2d.dtype) * self.lambda_depth_mono * falloff * \
            torch.square(scale * results['depth'] + shift - depth_2d)

    # ---- dictionary of terms ---------------------------------------------------------------
    def forward(self, results, target, **kwargs):
        d = {}
        if kwargs.get('embed_msk', False):
            d['r_ms'], _ = self.mask_regularize(kwargs['mask'], self.Annealing.getWeight(kwargs['step']), 0)
            d['rgb'] = self._colour(results, target, kwargs['mask'])
        else:
            d['rgb'] = self._colour(results, target)
        d['opacity'] = self._opacity_entropy(results)
        if self.lambda_distortion > 0:
            d['distortion'] = self._distortion(results)
        if kwargs.get('normal_ref', False):
            if torch.is_grad_enabled() and getattr(results['Ro'], '_ngp_normals_have_grad', True) is False:
                raise RuntimeError("normal_ref=True, but normals_raw carries no gradient: the Ro term would silently "
                                   "train nothing.  Set model.differentiable_normals = True before render() "
                                   "(NGPTrainer(loss_kwargs={'normal_ref': True}) does), see networks.NGP.forward")
            d['normal_ref_rp'] = self.lambda_normal_ref_rp * results['Rp']
            d['normal_ref_ro'] = self.lambda_normal_ref_ro * results['Ro']
        if kwargs.get('normal_mono', False):
            d['normal_mono'] = self._normal_mono(results, target)
        if kwargs.get('semantic', False):
            d['CELoss'] = self.lambda_semantic * self.CrossEntropyLoss(results['semantic'], target['label'])
            is_sky = (target['label'] == 4).to(results['depth'].dtype)
            d['sky_depth'] = self.lambda_sky * is_sky * torch.exp(-results['depth'])
        if kwargs.get('depth_mono', False):
            d['depth_mono'] = self._depth_mono(results, target, kwargs.get('scale', 1))
        return d

    def mask_regularize(self, mask, size_delta, digit_delta):
        """keeps the transient mask small (size term) and binary (digit term)"""
        eps = 0.02
        size_term = mask.pow(2).mean() * size_delta
        digit_term = (1 / ((mask - 0.5).pow(2) + eps)).mean() * digit_delta
        return size_term, digit_term
