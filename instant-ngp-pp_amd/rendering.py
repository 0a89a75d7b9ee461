"""render(model, rays_o, rays_d, **kwargs) — the L2 render API of the reference
(models/rendering.py:13-251) on the MI355X operator surface.

Train keys: deltas, ts, rays_a, total_samples, sigma, xyzs, vr_samples, opacity, depth, rgb,
            normal_pred, semantic, ws, Ro, Rp                     (rendering.py:205-251)
Test keys:  opacity, depth, rgb, normal_pred, normal_raw, semantic, total_samples, points, mask
                                                                   (rendering.py:176-185)
kwargs read: test_time, to_cpu, to_numpy, exp_step_factor, embedding_a, num_classes, max_samples,
             T_threshold, use_skybox, random_bg (+ passed through to the model).
"""
import torch
import torch.nn.functional as F

from . import vren
from ._lib import call
from .custom_functions import RayAABBIntersector, RayMarcher, RefLoss, VolumeRenderer, mark_full_cover

MAX_SAMPLES = 1024
NEAR_DISTANCE = 0.01


def render(model, rays_o, rays_d, **kwargs):
    rays_o = rays_o.contiguous()
    rays_d = rays_d.contiguous()
    marched = kwargs.get('marched', None)
    if marched is not None and not kwargs.get('test_time', False):
        hits_t = marched.hits_t   # AABB + marcher already ran for exactly these rays (MarchAhead)
    else:
        hits_t = intersect_scene(model, rays_o, rays_d)

    fn = _render_rays_test if kwargs.get('test_time', False) else _render_rays_train
    results = fn(model, rays_o, rays_d, hits_t, **kwargs)
    if kwargs.get('to_cpu', False):
        for k, v in results.items():
            v = v.cpu()
            if kwargs.get('to_numpy', False):
                v = v.numpy()
            results[k] = v
    return results


def intersect_scene(model, rays_o, rays_d):
    """ray / scene-AABB intersection with the near clamp of rendering.py:25-30 -> hits_t (N_rays,1,2)"""
    _, hits_t, _ = RayAABBIntersector.apply(rays_o, rays_d, model.center, model.half_size, 1)
    # 0 <= t1 < NEAR_DISTANCE -> NEAR_DISTANCE, one fused launch
    call("clamp_near", hits_t, hits_t.shape[0], 1, NEAR_DISTANCE)
    return hits_t


class MarchAhead:
    """The ray-only front of a training step — AABB test, near clamp, occupancy marcher — for the
    NEXT ray batch, on its own HIP stream.

    None of it reads a network parameter, so it can run under the current step's backward /
    optimizer.  The sample count comes back through a pinned host word and an event on that
    stream: the host waits for the marcher alone, never for the main stream, and is therefore a
    step ahead of the device when it enqueues the field kernels (the reference synchronises the
    whole device at this point every step, custom_functions.py:93).  The result is handed to
    render(..., marched=...) and is only valid for the same rays and the occupancy bitfield it
    was marched with (the trainer does not march across a density-grid update)."""

    def __init__(self, device):
        self.stream = torch.cuda.Stream(device=device, priority=int(_os.environ.get("NGP_MARCH_PRIO", "0")))
        self.count_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.pending = None

    def launch(self, model, rays_o, rays_d, exp_step_factor=0.):
        main = torch.cuda.current_stream()
        self.stream.wait_stream(main)   # the rays and the bitfield were produced on the main stream
        with torch.cuda.stream(self.stream), torch.no_grad():
            hits_t = intersect_scene(model, rays_o, rays_d)
            noise = torch.rand_like(rays_o[:, 0])
            out = vren.raymarching_train_untrimmed(rays_o, rays_d, hits_t[:, 0], model.density_bitfield,
                                                   model.cascades, model.scale, exp_step_factor, noise,
                                                   model.grid_size, MAX_SAMPLES)
            self.count_host.copy_(out[5][:1], non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.stream)
        self.pending = (rays_o, rays_d, exp_step_factor, hits_t, out, done)

    def take(self, rays_o, rays_d, exp_step_factor=0.):
        """-> the marched batch if it was launched for exactly these tensors, else None"""
        p, self.pending = self.pending, None
        if p is None or p[0] is not rays_o or p[1] is not rays_d or p[2] != exp_step_factor:
            return None
        _, _, _, hits_t, (rays_a, xyzs, dirs, deltas, ts, counter), done = p
        done.synchronize()
        n = int(self.count_host[0])
        main = torch.cuda.current_stream()
        main.wait_event(done)
        for t in (hits_t, rays_a, xyzs, dirs, deltas, ts, counter):
            t.record_stream(main)   # allocated on the side stream's pool, consumed on the main stream
        m = _Marched()
        m.hits_t, m.rays_a, m.total_samples = hits_t, rays_a, counter[0]
        m.xyzs, m.dirs, m.deltas, m.ts = xyzs[:n], dirs[:n], deltas[:n], ts[:n]
        return m


class _Marched:
    __slots__ = ("hits_t", "rays_a", "xyzs", "dirs", "deltas", "ts", "total_samples")


def render_chunks(model, rays_o, rays_d, chunk_size, **kwargs):
    """render() in chunks of `chunk_size` rays (render.py:33-48): per-ray results concatenated,
    `total_samples` kept as a list"""
    results = {}
    for i in range(0, rays_o.shape[0], chunk_size):
        ret = render(model, rays_o[i:i + chunk_size], rays_d[i:i + chunk_size], **dict(kwargs))
        for k, v in ret.items():
            results.setdefault(k, []).append(v)
    for k in results:
        if k != 'total_samples':
            results[k] = torch.cat(results[k], 0)
    return results


@torch.no_grad()
def render_dense(model, rays_o, rays_d, z_vals, **kwargs):
    """The dense-sample path of rendering_noCUDA.py:103-214 on the HIP kernels: the field at the
    caller's depths `z_vals` (N_rays, S), composited as raw2outputs does (custom_functions.py:280-321:
    dists = diff(z)·|d| with a last interval of 1e10, no early termination).  Used to compare this
    library with the reference's noCUDA path on identical rays AND identical samples.
    -> opacity, depth, rgb, normal_raw, normal_pred, semantic (logit sums), ws (N_rays, S)"""
    classes = kwargs.get('num_classes', 7)
    rays_o, rays_d, z_vals = rays_o.contiguous(), rays_d.contiguous(), z_vals.contiguous()
    n_rays, S = z_vals.shape
    dev = rays_o.device
    xyzs = (rays_o[:, None, :] + rays_d[:, None, :] * z_vals[:, :, None]).reshape(-1, 3).contiguous()
    dirs = rays_d[:, None, :].expand(-1, S, -1).reshape(-1, 3).contiguous()
    sigmas, rgbs, normals_raw, normals_pred, sems = model(xyzs, dirs, **kwargs)
    dists = torch.cat([z_vals[:, 1:] - z_vals[:, :-1], torch.full_like(z_vals[:, :1], 1e10)], -1)
    dists = (dists * torch.norm(rays_d, dim=-1, keepdim=True)).reshape(-1).contiguous()
    idx = torch.arange(n_rays, device=dev, dtype=torch.int64)
    rays_a = torch.stack([idx, idx * S, torch.full_like(idx, S)], -1).contiguous()
    out = {}
    ts = z_vals.reshape(-1).contiguous()
    _, out['opacity'], out['depth'], out['rgb'], out['normal_pred'], out['semantic'], ws = vren.composite_train_fw(
        sigmas.contiguous(), rgbs.contiguous(), normals_pred.contiguous(), sems.contiguous(), dists, ts, rays_a,
        0.0, classes)
    out['normal_raw'] = vren.composite_train_fw(
        sigmas.contiguous(), rgbs.contiguous(), normals_raw.contiguous(), sems.contiguous(), dists, ts, rays_a,
        0.0, classes)[4]
    out['ws'] = ws.view(n_rays, S)
    return out


import os as _os

_REFERENCE_TEST_LOOP = _os.environ.get("NGP_REFERENCE_TEST_LOOP", "0") == "1"
# NGP_DEVICE_ROUNDS=1 / render(..., device_rounds=True): loop head, alive compaction and sample count on the device, no
# host round trip per round (volume_render_device_rounds).  Off by default: an 800x800 frame is 34 rounds of ~1.7 ms of
# field kernels each (tools/rounds_probe.py), the one host sync per round costs ~2 % of the frame, and without the exact
# row count on the host the field evaluates up to N_rays rows per round instead of N_alive * N_samples (measured 61.6 ms
# per frame against 55.3 for the host-driven loop).
_DEVICE_ROUNDS = _os.environ.get("NGP_DEVICE_ROUNDS", "0") == "1"


def volume_render(model, rays_o, rays_d, hits_t, opacity, depth, rgb, normal_pred, normal_raw, sem, **kwargs):
    """Progressive test-time marching (rendering.py:46-133): the per-ray accumulators are updated
    in place; returns the total number of samples evaluated.

    Same rounds, same samples per round and same compositing as the reference, with less work per
    round: the reference compacts the valid samples with a boolean mask, evaluates the field on
    them and scatters five result tensors back into zero-filled padded ones (~110 launches and
    three host syncs per round); here the field runs on the padded block directly — the marcher's
    padding rows are zeros, 6 % of the slots on the proxy scene, the compositor reads the first
    N_eff samples of a ray only, and a field row does not depend on the other rows of the batch, so
    every per-ray result is bit-identical — which leaves ~25 launches and one sync per round.
    `volume_render_reference` keeps the literal loop (NGP_REFERENCE_TEST_LOOP=1 selects it)."""
    if _REFERENCE_TEST_LOOP or kwargs.get('reference_test_loop', False):
        return volume_render_reference(model, rays_o, rays_d, hits_t, opacity, depth, rgb, normal_pred, normal_raw,
                                       sem, **kwargs)
    if (_DEVICE_ROUNDS or kwargs.get('device_rounds', False)) and rays_o.is_cuda and len(rays_o) > 0:
        return volume_render_device_rounds(model, rays_o, rays_d, hits_t, opacity, depth, rgb, normal_pred, normal_raw,
                                           sem, **kwargs)
    N_rays = len(rays_o)
    device = rays_o.device
    exp_step_factor = kwargs.get('exp_step_factor', 0.)
    classes = kwargs.get('num_classes', 7)
    T_threshold = kwargs.get('T_threshold', 1e-4)
    samples = 0
    total_samples = torch.zeros((), dtype=torch.int64, device=device)
    alive_indices = torch.arange(N_rays, device=device)
    min_samples = 1 if exp_step_factor == 0 else 4
    f32 = torch.float32
    while samples < kwargs.get('max_samples', MAX_SAMPLES):
        N_alive = len(alive_indices)
        if N_alive == 0:
            break
        N_samples = max(min(N_rays // N_alive, 64), min_samples)
        samples += N_samples
        n_pts = N_alive * N_samples
        xyzs = torch.zeros(n_pts, 3, dtype=f32, device=device)      # padding rows must hold finite inputs
        dirs = torch.zeros(n_pts, 3, dtype=f32, device=device)
        deltas = torch.empty(N_alive, N_samples, dtype=f32, device=device)   # read up to N_eff only
        ts = torch.empty(N_alive, N_samples, dtype=f32, device=device)
        N_eff_samples = torch.empty(N_alive, dtype=torch.int32, device=device)
        call("raymarching_test", rays_o, rays_d, hits_t, alive_indices, model.density_bitfield, int(model.cascades),
             float(model.scale), float(exp_step_factor), int(model.grid_size), MAX_SAMPLES, int(N_samples), N_alive,
             xyzs, dirs, deltas, ts, N_eff_samples)
        total_samples += N_eff_samples.sum()
        sigmas, rgbs, normals_pred, normals_raw, sems = model.forward_test(xyzs, dirs, **kwargs)
        call("composite_test_fw", sigmas.contiguous(), rgbs.contiguous(), normals_pred.contiguous(),
             normals_raw.contiguous(), sems.contiguous(), deltas, ts, hits_t, alive_indices, float(T_threshold),
             int(classes), N_eff_samples, N_alive, int(N_samples), opacity, depth, rgb, normal_pred, normal_raw, sem)
        alive_indices = alive_indices[alive_indices >= 0]   # the one host sync of the round

    if kwargs.get('use_skybox', False):
        rgb_bg = model.forward_skybox(rays_d)
        rgb += rgb_bg * (1 - opacity)[:, None]
    return total_samples


_ROUND_LAG = 3      # rounds the host may be ahead of the device's round state


def volume_render_device_rounds(model, rays_o, rays_d, hits_t, opacity, depth, rgb, normal_pred, normal_raw, sem, **kwargs):
    """The same rounds with the loop head on the DEVICE (ngp_test_round_begin): N_alive, N_samples of the round, the
    running `samples` sum and the total sample count live in a device record, the alive list is compacted by a kernel
    (ngp_alive_compact, order kept), and the host enqueues round after round without waiting for any of it.  It reads
    the record back through pinned memory `_ROUND_LAG` rounds late: to stop (the rounds enqueued in between do nothing
    once the record says done) and to size its launches — a stale N_alive is an upper bound of the current one, the
    marcher / compositor take the exact sizes from the device, and the field evaluates the (few) padding rows beyond
    N_alive * N_samples on zero inputs whose results nobody reads.  Per-ray results are bit-identical to the host-driven
    loop above and to volume_render_reference (same schedule, same samples, row results independent of the batch).
    Opt-in (see _DEVICE_ROUNDS): the frame is bound by its field kernels, not by the round trips."""
    N_rays = len(rays_o)
    device = rays_o.device
    exp_step_factor = kwargs.get('exp_step_factor', 0.)
    classes = kwargs.get('num_classes', 7)
    T_threshold = kwargs.get('T_threshold', 1e-4)
    max_total = int(kwargs.get('max_samples', MAX_SAMPLES))
    min_samples = 1 if exp_step_factor == 0 else 4
    f32 = torch.float32
    state = torch.zeros(8, dtype=torch.int32, device=device)
    state[0:1].fill_(N_rays)
    alive = [torch.arange(N_rays, device=device), torch.empty(N_rays, dtype=torch.int64, device=device)]
    cap = max(N_rays, min_samples * N_rays)
    xyzs = torch.empty(cap, 3, dtype=f32, device=device)
    dirs = torch.empty(cap, 3, dtype=f32, device=device)
    deltas = torch.empty(cap, dtype=f32, device=device)
    ts = torch.empty(cap, dtype=f32, device=device)
    n_eff = torch.empty(N_rays, dtype=torch.int32, device=device)
    counts = torch.empty((N_rays + 1023) // 1024, dtype=torch.int32, device=device)
    host = [torch.zeros(8, dtype=torch.int32).pin_memory() for _ in range(_ROUND_LAG)]
    events = [None] * _ROUND_LAG
    stream = torch.cuda.current_stream()
    nh = N_rays                     # host's (stale, hence upper) bound of the number of alive rays
    cur = 0
    for rnd in range(max_total):    # every round adds at least one sample per ray to the running sum
        slot = rnd % _ROUND_LAG
        if events[slot] is not None:
            events[slot].synchronize()          # the record as it stood _ROUND_LAG rounds ago
            if int(host[slot][3]) or int(host[slot][5]) == 0:
                break
            nh = int(host[slot][5])
        n_ub = min(64 * nh, max(N_rays, min_samples * nh))   # >= N_alive * N_samples of this round
        call("test_round_begin", state, N_rays, min_samples, max_total)
        xyzs[:n_ub].zero_()         # padding rows must hold finite inputs
        dirs[:n_ub].zero_()
        call("raymarching_test_rounds", rays_o, rays_d, hits_t, alive[cur], model.density_bitfield, int(model.cascades),
             float(model.scale), float(exp_step_factor), int(model.grid_size), MAX_SAMPLES, state, nh,
             xyzs, dirs, deltas, ts, n_eff)
        sigmas, rgbs, normals_pred, normals_raw, sems = model.forward_test(xyzs[:n_ub], dirs[:n_ub], **kwargs)
        call("composite_test_fw_rounds", sigmas.contiguous(), rgbs.contiguous(), normals_pred.contiguous(),
             normals_raw.contiguous(), sems.contiguous(), deltas, ts, alive[cur], float(T_threshold), int(classes), n_eff,
             state, nh, opacity, depth, rgb, normal_pred, normal_raw, sem)
        call("alive_compact", alive[cur], state, nh, counts, alive[1 - cur])
        cur = 1 - cur
        host[slot].copy_(state, non_blocking=True)
        events[slot] = torch.cuda.Event()
        events[slot].record(stream)
    total_samples = state[6:8].view(torch.int64)[0]
    if kwargs.get('use_skybox', False):
        rgb_bg = model.forward_skybox(rays_d)
        rgb += rgb_bg * (1 - opacity)[:, None]
    return total_samples


def volume_render_reference(model, rays_o, rays_d, hits_t, opacity, depth, rgb, normal_pred, normal_raw, sem, **kwargs):
    """the literal loop of rendering.py:46-133 (mask, compact, evaluate, scatter back)"""
    N_rays = len(rays_o)
    device = rays_o.device
    exp_step_factor = kwargs.get('exp_step_factor', 0.)
    classes = kwargs.get('num_classes', 7)
    T_threshold = kwargs.get('T_threshold', 1e-4)
    samples = 0
    total_samples = 0
    alive_indices = torch.arange(N_rays, device=device)
    # synthetic scenes are mostly background: 1 sample per round retires those rays quickly
    min_samples = 1 if exp_step_factor == 0 else 4

    while samples < kwargs.get('max_samples', MAX_SAMPLES):
        N_alive = len(alive_indices)
        if N_alive == 0:
            break
        N_samples = max(min(N_rays // N_alive, 64), min_samples)
        samples += N_samples

        xyzs, dirs, deltas, ts, N_eff_samples = vren.raymarching_test(
            rays_o, rays_d, hits_t, alive_indices, model.density_bitfield, model.cascades, model.scale,
            exp_step_factor, model.grid_size, MAX_SAMPLES, N_samples)
        total_samples += N_eff_samples.sum()
        xyzs = xyzs.reshape(-1, 3)
        dirs = dirs.reshape(-1, 3)
        valid_mask = ~torch.all(dirs == 0, dim=1)
        if valid_mask.sum() == 0:
            break

        n_pts = len(xyzs)
        sigmas = torch.zeros(n_pts, device=device)
        rgbs = torch.zeros(n_pts, 3, device=device)
        normals_pred = torch.zeros(n_pts, 3, device=device)
        normals_raw = torch.zeros(n_pts, 3, device=device)
        sems = torch.zeros(n_pts, classes, device=device)

        _sigmas, _rgbs, _normals_pred, _normals_raw, _sems = \
            model.forward_test(xyzs[valid_mask], dirs[valid_mask], **kwargs)
        sigmas[valid_mask] = _sigmas.detach().float()
        rgbs[valid_mask] = _rgbs.detach().float()
        normals_pred[valid_mask] = _normals_pred.float()
        normals_raw[valid_mask] = _normals_raw.float()
        sems[valid_mask] = _sems.float()

        vren.composite_test_fw(
            sigmas.view(N_alive, N_samples), rgbs.view(N_alive, N_samples, 3),
            normals_pred.view(N_alive, N_samples, 3), normals_raw.view(N_alive, N_samples, 3),
            sems.view(N_alive, N_samples, classes), deltas, ts, hits_t, alive_indices, T_threshold, classes,
            N_eff_samples, opacity, depth, rgb, normal_pred, normal_raw, sem)
        alive_indices = alive_indices[alive_indices >= 0]

    if kwargs.get('use_skybox', False):
        rgb_bg = model.forward_skybox(rays_d)
    else:
        rgb_bg = torch.zeros(3, device=device)
    rgb += rgb_bg * (1 - opacity)[:, None]
    return total_samples


@torch.no_grad()
def _render_rays_test(model, rays_o, rays_d, hits_t, **kwargs):
    hits_t = hits_t[:, 0, :].contiguous()
    classes = kwargs.get('num_classes', 7)
    N_rays = len(rays_o)
    device = rays_o.device
    opacity = torch.zeros(N_rays, device=device)
    depth = torch.zeros(N_rays, device=device)
    rgb = torch.zeros(N_rays, 3, device=device)
    normal_pred = torch.zeros(N_rays, 3, device=device)
    normal_raw = torch.zeros(N_rays, 3, device=device)
    sem = torch.zeros(N_rays, classes, device=device)
    mask = torch.zeros(N_rays, device=device)

    total_samples = volume_render(model, rays_o, rays_d, hits_t, opacity, depth, rgb, normal_pred, normal_raw, sem,
                                  **kwargs)
    results = {
        'opacity': opacity, 'depth': depth, 'rgb': rgb,
        'normal_pred': F.normalize(normal_pred, dim=-1),
        'normal_raw': F.normalize(normal_raw, dim=-1),
        'semantic': torch.argmax(sem, dim=-1, keepdim=True),
        'total_samples': total_samples,
        'points': rays_o + rays_d * depth.unsqueeze(-1),
        'mask': mask,
    }
    return results


def _render_rays_train(model, rays_o, rays_d, hits_t, **kwargs):
    exp_step_factor = kwargs.get('exp_step_factor', 0.)
    T_threshold = kwargs.get('T_threshold', 1e-4)
    classes = kwargs.get('num_classes', 7)
    results = {}
    marched = kwargs.pop('marched', None)
    if marched is not None:
        rays_a, xyzs, dirs, total_samples = marched.rays_a, marched.xyzs, marched.dirs, marched.total_samples
        results['deltas'], results['ts'] = marched.deltas, marched.ts
    else:
        with torch.no_grad():
            rays_a, xyzs, dirs, results['deltas'], results['ts'], total_samples = RayMarcher.apply(
                rays_o, rays_d, hits_t[:, 0], model.density_bitfield, model.cascades, model.scale, exp_step_factor,
                model.grid_size, MAX_SAMPLES)
    mark_full_cover(rays_a)   # the marcher's segments tile [0, N): the compositor may skip its zero-fills
    results['rays_a'] = rays_a
    results['total_samples'] = total_samples

    # per-ray tensor kwargs (embedding_a, exposure, ...) are repeated per sample; like the
    # reference this rewrites kwargs in place (rendering.py:217-219)
    for k, v in kwargs.items():
        if isinstance(v, torch.Tensor):
            kwargs[k] = torch.repeat_interleave(v[rays_a[:, 0]], rays_a[:, 2], 0, output_size=xyzs.shape[0])
    # the rays' segments go along: the field evaluates its colour branch only on the samples the compositor below
    # will use (all up to a ray's early-termination point — same sigma, deltas and T_threshold, same decision)
    fused = kwargs.pop('_fused_loss', None)
    if fused is not None and _fused_tail_ok(model, kwargs, exp_step_factor, classes):
        return _render_loss_fused(model, results, xyzs, dirs, rays_a, T_threshold, classes, fused, kwargs)
    model._live_ctx = (rays_a, results['deltas'], T_threshold)
    try:
        sigmas, rgbs, normals_raw, normals_pred, sems = model(xyzs, dirs, **kwargs)
    finally:
        model._live_ctx = None
    results['sigma'] = sigmas
    results['xyzs'] = xyzs

    (results['vr_samples'], results['opacity'], results['depth'], results['rgb'], results['normal_pred'],
     results['semantic'], results['ws']) = VolumeRenderer.apply(
        sigmas.contiguous(), rgbs.contiguous(), normals_pred.contiguous(), sems.contiguous(),
        results['deltas'], results['ts'], rays_a, T_threshold, classes)

    rgb_bg = None  # black background (synthetic scenes): rgb + 0*(1-opacity) is rgb, skip the ops
    if kwargs.get('use_skybox', False):
        rgb_bg = model.forward_skybox(rays_d)
    elif exp_step_factor != 0 and kwargs.get('random_bg', False):
        rgb_bg = torch.rand(3, device=rays_o.device)
    if rgb_bg is not None:
        results['rgb'] = results['rgb'] + rgb_bg * (1 - results['opacity'])[:, None]

    # Ref-NeRF normal regularisers (rendering.py:243-249)
    normals_diff, normals_ori = _RefLossInputs.apply(normals_raw, normals_pred, dirs)
    results['Ro'], results['Rp'] = RefLoss.apply(
        sigmas.detach().contiguous(), normals_diff, normals_ori,
        results['deltas'], results['ts'], rays_a, T_threshold)
    # NeRFLoss(normal_ref=True) needs Ro to reach the density field through normals_raw (reference:
    # create_graph=True, networks.py:186-196); the default field returns detached analytic normals
    results['Ro']._ngp_normals_have_grad = bool(normals_raw.requires_grad)
    return results


def _fused_tail_ok(model, kwargs, exp_step_factor, classes):
    """the one-launch render + loss tail covers the default recipe: sigmoid colours (no tone mapper), black or random
    constant background (no skybox network), detached analytic normals, at most 8 classes"""
    return (getattr(model, 'rgb_act', 'Sigmoid') == 'Sigmoid' and not kwargs.get('use_skybox', False)
            and not getattr(model, 'differentiable_normals', False) and classes <= 8
            and not getattr(model, 'compact_dead_samples', None) and hasattr(model, '_field'))


class _RenderLossFn(torch.autograd.Function):
    """Default-recipe tail of a training step as ONE launch (ngp_render_loss_fused): normals, softmax, compositing,
    Ref-NeRF regularisers, distortion loss, NeRFLoss's default terms AND their gradients w.r.t. the field's outputs.
    forward returns (terms (4) = [loss, rgb, opacity, distortion], per-ray results ..., ws); only terms is
    differentiable, and only through terms[0] with a unit seed (NGPTrainer's use): backward hands the gradients
    computed in forward to the field."""

    @staticmethod
    def forward(ctx, sig, rgb_o, dsig_dx, np_raw, sem_logits, dirs, deltas, ts, rays_a, rgb_gt, scale3, T_thr, classes,
                lambda_opa, lambda_dist, rgb_bg=None):
        n, nr = sig.shape[0], rays_a.shape[0]
        dev = sig.device
        f32 = torch.float32
        total = torch.empty(nr, dtype=torch.int64, device=dev)
        E = lambda *shape: torch.empty(*shape, dtype=f32, device=dev)   # (the caching allocator launches nothing)
        opacity, depth, rgb, normal, Ro, Rp, sem = E(nr), E(nr), E(nr, 3), E(nr, 3), E(nr), E(nr, 3), E(nr, classes)
        ws, d_sig, d_rgb = E(n), E(n), E(n, 3)
        acc = E(8)                                   # [terms (4) | vr_samples (int64) | -]: adjacent, cleared by one memset
        terms, vr = acc[:4], acc[4:6].view(torch.int64)
        call("render_loss_fused", sig, rgb_o, dsig_dx, scale3, np_raw, np_raw.stride(0), sem_logits, sem_logits.stride(0),
             dirs, deltas, ts, rays_a, rgb_gt, rgb_bg, float(T_thr), int(classes), nr, float(lambda_opa), float(lambda_dist),
             total, vr, opacity, depth, rgb, normal, sem, ws, Ro, Rp, terms, d_sig, d_rgb)
        ctx.save_for_backward(d_sig, d_rgb)
        ctx.set_materialize_grads(False)             # no zero-filled gradient tensors for the ten other outputs
        ctx.mark_non_differentiable(total, vr, opacity, depth, rgb, normal, sem, ws, Ro, Rp)
        return terms, total, vr, opacity, depth, rgb, normal, sem, ws, Ro, Rp

    @staticmethod
    def backward(ctx, g_terms, *_unused):
        d_sig, d_rgb = ctx.saved_tensors
        return (d_sig, d_rgb) + (None,) * 14


def _render_loss_fused(model, results, xyzs, dirs, rays_a, T_threshold, classes, fused, kwargs):
    sig, rgb_o, dsig_dx, np_raw, sem_logits = model._field(xyzs, dirs, kwargs)
    rgb_gt, lambda_opa, lambda_dist = fused
    rgb_bg = None
    if kwargs.get('exp_step_factor', 0.) != 0 and kwargs.get('random_bg', False):
        rgb_bg = torch.rand(3, device=xyzs.device)      # rendering.py:239 (drawn at the same place in the RNG stream)
    (terms, total, vr, opacity, depth, rgb, normal, sem, ws, Ro, Rp) = _RenderLossFn.apply(
        sig, rgb_o, dsig_dx, np_raw, sem_logits, dirs.contiguous(), results['deltas'], results['ts'], rays_a,
        rgb_gt.contiguous(), model._inv_span(), T_threshold, classes, lambda_opa, lambda_dist, rgb_bg)
    results['sigma'] = sig
    results['xyzs'] = xyzs
    results['vr_samples'] = vr[0]
    results['opacity'], results['depth'], results['rgb'] = opacity, depth, rgb
    results['normal_pred'], results['semantic'], results['ws'] = normal, sem, ws
    results['Ro'], results['Rp'] = Ro, Rp
    results['Ro']._ngp_normals_have_grad = False
    results['_loss_terms'] = terms        # terms[0] carries the graph: NGPTrainer seeds its backward with 1
    return results


class _RefLossInputs(torch.autograd.Function):
    """normals_diff = (n_raw - n_pred)^2 (N,3), normals_ori = clamp(<n_raw, normalize(dir)>, 0)^2 (N)
    in one launch (the reference spends ~12 elementwise launches here, rendering.py:243-245)."""

    @staticmethod
    def forward(ctx, normals_raw, normals_pred, dirs):
        normals_raw, normals_pred, dirs = normals_raw.contiguous(), normals_pred.contiguous(), dirs.contiguous()
        n = normals_raw.shape[0]
        ndiff = torch.empty(n, 3, dtype=torch.float32, device=dirs.device)
        nori = torch.empty(n, dtype=torch.float32, device=dirs.device)
        call("refloss_inputs", normals_raw, normals_pred, dirs, n, ndiff, nori)
        ctx.save_for_backward(normals_raw, normals_pred, dirs)
        return ndiff, nori

    @staticmethod
    def backward(ctx, g_diff, g_ori):
        normals_raw, normals_pred, dirs = ctx.saved_tensors
        e = normals_raw - normals_pred
        d_raw = d_pred = None
        if g_diff is not None:
            d_raw = 2 * e * g_diff
            d_pred = -d_raw
        if g_ori is not None:
            dn = F.normalize(dirs, p=2, dim=-1, eps=1e-6)
            dot = torch.clamp(torch.sum(normals_raw * dn, dim=-1, keepdim=True), min=0.)
            t = 2 * dot * dn * g_ori[:, None]
            d_raw = t if d_raw is None else d_raw + t
        return d_raw, d_pred, None


# name-mangled aliases the reference module exposes internally
__render_rays_train = _render_rays_train
__render_rays_test = _render_rays_test
