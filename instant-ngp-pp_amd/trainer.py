"""Training schedule of the reference's NeRFSystem (train.py:82-345) without Lightning:

  every 16 steps update_density_grid(0.01*1024/sqrt(3), warmup = step < 256)   train.py:272-275
  render() -> NeRFLoss -> sum of term means -> backward                        train.py:279-307
  gradient clipping by global norm 50, Adam(lr, eps=1e-8)                      train.py:244,435
  CosineAnnealingLR over the epochs down to lr/30, stepped once per epoch      train.py:249-251
  ray-batch data parallel: every rank draws its own rays, gradients are averaged  train.py:430-432

MI355X-specific structure:
  * all parameters live in ONE flat fp32 buffer (rgb table | xyz table | MLPs) with matching flat
    gradient / Adam-state buffers: one fused Adam launch per step (which also applies the clip
    coefficient and zeroes the gradient), two large RCCL all-reduces instead of per-tensor ones;
  * the hash-grid scatter kernels accumulate straight into the flat gradient buffer (no
    zeros_like + add pass over 800 MB);
  * with world_size > 1 the optimizer is sharded (ZeRO-1 style): gradients are reduce-scattered
    over RCCL (the rgb-table bucket, 77 % of the bytes, as soon as its scatter-add is enqueued, so
    it overlaps the density-path backward), every rank runs clip + Adam on its 1/N slice only
    (Adam's 6.4 GB/step of HBM traffic becomes 6.4/N GB), and the updated parameter slices are
    all-gathered.  Same bytes on xGMI as an all-reduce, 1/N of the optimizer time and state.
"""
import math
import os

import torch
import torch.distributed as dist

from ._lib import call
from .losses import NeRFLoss, nerf_loss_and_grads
from .rendering import MAX_SAMPLES, MarchAhead, render

_f32 = torch.float32


class GradBuckets:
    """Collectives over contiguous buckets of the flat gradient / parameter buffers (device
    agnostic: the gloo CPU tests drive this class directly).

    all-reduce mode   : reduce_bucket(i) sums bucket i over the ranks (async), wait() joins.
    sharded mode      : reduce_scatter_bucket(i, out) gives rank r the summed slice r of bucket i
                        (RCCL reduce-scatter into the rank's own shard buffer),
                        all_gather_bucket(i, flat_param, shard) publishes each rank's updated
                        parameter slice to everyone.
    Buckets must be a multiple of world*4 elements long in sharded mode (the trainer pads).
    """

    def __init__(self, flat_grad, boundaries, group=None, solo=True):
        self.flat = flat_grad
        self.bounds = list(boundaries)  # [0, b1, ..., n]
        self.group = group
        self.works = []
        # solo=False: a ONE-rank process group still sends every collective of the N>1 path through
        # the backend (a rehearsal of the RCCL calls on a single-GPU box, see NGPTrainer.force_sharded)
        self.solo = solo
        # set to a list to record every collective issued from now on: (op, bucket, payload bytes, HIP stream the
        # call was enqueued behind) — bench.py --gpus N prints it per rank
        self.trace = None

    def _issue(self, op, i, nbytes, fn):
        """runs fn() (which issues one collective and returns its work object or None).  With `trace` set the
        collective is also TIMED: HIP events on the issuing stream before the call and behind the work's completion
        (the work is waited for at once, so the traced steps serialise their collectives: durations are those of
        each collective by itself beside whatever the other streams run, not the overlapped schedule)."""
        if self.trace is None:
            return fn()
        rec = {"op": op, "bucket": int(i), "bytes": int(nbytes),
               "stream": int(torch.cuda.current_stream().cuda_stream) if self.flat.is_cuda else 0}
        if self.flat.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            w = fn()
            if w is not None:
                w.wait()
            e1.record()
            rec["events"] = (e0, e1)
        else:
            w = fn()
        self.trace.append(rec)
        return w

    @property
    def world(self):
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    @property
    def rank(self):
        return dist.get_rank(self.group) if dist.is_available() and dist.is_initialized() else 0

    def _native_rs(self):
        return dist.get_backend(self.group) == "nccl"

    def shard_range(self, i):
        lo, hi = self.bounds[i], self.bounds[i + 1]
        s = (hi - lo) // self.world
        return lo + self.rank * s, lo + (self.rank + 1) * s

    def reduce_bucket(self, i):
        if self.world == 1 and self.solo:
            return
        lo, hi = self.bounds[i], self.bounds[i + 1]
        self.works.append(self._issue("all_reduce", i, 4 * (hi - lo), lambda: dist.all_reduce(
            self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)))

    def reduce_scatter_bucket(self, i, out):
        """out (own buffer, 1/world of the bucket) <- this rank's slice of the bucket summed over ranks"""
        if self.world == 1 and self.solo:
            return
        lo, hi = self.bounds[i], self.bounds[i + 1]
        if self._native_rs():
            self.works.append(self._issue("reduce_scatter", i, 4 * (hi - lo), lambda: dist.reduce_scatter_tensor(
                out, self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)))
        else:  # gloo has no reduce-scatter: all-reduce, then the owner copies its slice out
            a, b = self.shard_range(i)

            def via_all_reduce():
                dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True).wait()
                out.copy_(self.flat[a:b])
            self._issue("reduce_scatter", i, 4 * (hi - lo), via_all_reduce)
            self.works.append(None)       # one entry per bucket, in issue order (NGPTrainer sums each shard behind its own)

    def all_gather_bucket(self, i, flat_param, shard, detach=False):
        """flat_param[bucket i] <- concatenation over ranks of `shard` (own buffer)"""
        if self.world == 1 and self.solo:
            return None
        lo, hi = self.bounds[i], self.bounds[i + 1]
        if self._native_rs():
            w = self._issue("all_gather", i, 4 * (hi - lo), lambda: dist.all_gather_into_tensor(
                flat_param[lo:hi], shard, group=self.group, async_op=True))
        else:
            s = (hi - lo) // self.world
            views = [flat_param[lo + r * s: lo + (r + 1) * s] for r in range(self.world)]
            w = self._issue("all_gather", i, 4 * (hi - lo), lambda: dist.all_gather(views, shard, group=self.group, async_op=True))
        if detach:
            return w   # the caller waits for it where the gathered parameters are first read
        self.works.append(w)
        return None

    def wait(self):
        for w in self.works:
            if w is not None:
                w.wait()
        self.works = []

    def take_works(self):
        """hands the pending work objects (issue order) to the caller, who waits for them one by one"""
        ws, self.works = self.works, []
        return ws


def shard_seed(base_seed, rank):
    """per-rank decorrelated ray sampling (the reference leaves this to
    DataLoader worker seeding)"""
    return int(base_seed) + int(rank)


class NGPTrainer:
    def __init__(self, model, lr=1e-2, num_epochs=20, steps_per_epoch=1000, clip_norm=50.0,
                 exp_step_factor=0.0, num_classes=7, density_threshold=0.01, render_kwargs=None, group=None,
                 force_sharded=None, loss_kwargs=None):
        """loss_kwargs: the flags train.py:289-300 hands to NeRFLoss (normal_ref, normal_mono, semantic,
        depth_mono, embed_msk, scale, ...).  Any optional term switches the fused default-recipe loss off;
        normal_ref also switches the field to differentiable normals (the Ro term must reach the density
        table through normals_raw, reference networks.py:186-196).
        force_sharded: take the sharded-optimizer path (reduce-scatter / Adam on the slice / all-gather)
        even with ONE rank in the process group, so that a single-GPU box can rehearse every RCCL call of
        the N>1 path; None reads the environment variable NGP_FORCE_SHARDED."""
        self.model = model
        self.force_sharded = bool(os.environ.get("NGP_FORCE_SHARDED")) if force_sharded is None else bool(force_sharded)
        self.base_lr = lr
        self.num_epochs = num_epochs
        self.steps_per_epoch = steps_per_epoch
        self.clip_norm = clip_norm
        self.exp_step_factor = exp_step_factor
        self.num_classes = num_classes
        self.density_threshold = density_threshold
        self.render_kwargs = dict(render_kwargs or {})
        self.loss_fn = NeRFLoss()
        self.loss_kwargs = dict(loss_kwargs or {})
        optional = any(self.loss_kwargs.get(k) for k in ("normal_ref", "normal_mono", "semantic", "depth_mono", "embed_msk"))
        self.fused_loss = not optional   # default recipe (rgb + opacity + distortion); False -> NeRFLoss module
        # NGP_NO_FUSED_TAIL=1 (A/B): the launch-per-operation tail (normals, softmax, compositor, RefLoss, distortion, loss)
        self.fused_tail = os.environ.get("NGP_NO_FUSED_TAIL", "0") != "1" 
        if self.loss_kwargs.get("normal_ref"):
            model.differentiable_normals = True
        self.warmup_steps = 256
        self.update_interval = 16
        self.global_step = 0
        self.group = group
        # Launch width of the Adam sweep (ngp_adam_step_width): how the sweep shares the CUs with the next step's density path
        # decides a few per cent of the step, and which width wins depends on the loop around the trainer (resident batches
        # or a loader at the head of every step) and on the box.  None = measure: after the all-cells
        # warm-up, windows of `update_interval` steps alternate between the candidates (4 windows each, timed with HIP
        # events on the optimizer stream, no host synchronisation), then the faster one stays.  The result of a step does
        # not depend on the width.  NGP_ADAM_WIDTH=<n> / adam_width=<n> fixes it.
        w = os.environ.get("NGP_ADAM_WIDTH", "")
        self.adam_width = int(w) if w.isdigit() else None
        self.adam_candidates = (512, 256)
        self.adam_tune = (320, 4)            # first step of the measurement, windows per candidate
        self._tune_events, self._tune_widths = [], []
        self._grad_zeroed = None
        self._flatten()
        # NGP_SERIAL_OPT=1 (A/B): clip + Adam on the caller's stream instead of the optimizer stream
        serial = os.environ.get("NGP_SERIAL_OPT", "0") == "1"
        self._opt_stream = (torch.cuda.current_stream(self.flat_param.device) if serial
                            else torch.cuda.Stream(device=self.flat_param.device,
                                                   priority=int(os.environ.get("NGP_OPT_PRIO", "0")))
                            ) if self.flat_param.is_cuda else None
        self._march_ahead = MarchAhead(self.flat_param.device) if self.flat_param.is_cuda else None
        if self.flat_param.is_cuda and not serial and not self.sharded and os.environ.get("NGP_SIX_STREAMS", "0") != "1":
            # Four streams, one per hardware queue of the HIP runtime's default pool: the caller's, the colour forward's, the
            # optimizer's and the march-ahead's.  The backward's table scatters go on the optimizer stream (clip + Adam follow them
            # there anyway) and the two heads on the march-ahead stream (idle in the middle of the forward) instead of streams of
            # their own: with six streams two pairs share a queue, and which pairs depends on the order the streams were created
            # in — the scatter behind the caller's MLP kernels would cost 0.4 ms per step.  Same speed as the six-stream layout
            # that happened to come out right (profiles/r03_occupancy_shaping.txt (12); NGP_SIX_STREAMS=1 for the A/B).
            from . import networks
            networks._SIDE[self.flat_param.device.index] = self._opt_stream
            networks._HEADS[self.flat_param.device.index] = self._march_ahead.stream

    # ------------------------------------------------------------------ flat parameter store
    def _flatten(self):
        named = [(n, p) for n, p in self.model.named_parameters() if p.numel() > 0]
        order = {"rgb_encoder.params": 0, "xyz_encoder.params": 1}
        named.sort(key=lambda np_: order.get(np_[0], 2))
        self.names = [n for n, _ in named]
        world = dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1
        quantum = 4 * world   # every slice 16-byte aligned; every bucket divisible by the world size
        sizes = [(p.numel() + 3) // 4 * 4 for _, p in named]
        if named[0][0] == "rgb_encoder.params":           # bucket 0 = the colour table alone
            sizes[0] = (sizes[0] + quantum - 1) // quantum * quantum
        rest = sum(sizes[1:]) if named[0][0] == "rgb_encoder.params" else sum(sizes)
        sizes[-1] += (quantum - rest % quantum) % quantum
        total = sum(sizes)
        dev = named[0][1].device
        forced = self.force_sharded and dist.is_available() and dist.is_initialized()
        self.sharded = world > 1 or bool(forced)
        self.flat_param = torch.zeros(total, dtype=_f32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=_f32, device=dev)
        self.scalars = torch.zeros(2, dtype=_f32, device=dev)  # [sum of squares, clip coefficient]
        # clipping from a norm bound (ngp_clip_decide): sum_s ||dz2[s]|| of the two MLPs that feed the encoders, and the
        # device flag "the bound did not settle it: compute the exact norm"
        self.norm_acc = torch.zeros(2, dtype=_f32, device=dev)
        self.need_exact = torch.zeros(1, dtype=torch.int32, device=dev)
        off = 0
        self.slices = {}
        for (n, p), sz in zip(named, sizes):
            view = self.flat_param[off:off + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
            p.grad = self.flat_grad[off:off + p.numel()].view_as(p)
            self.slices[n] = (off, p.numel())
            off += sz
        # bucket 0 = rgb table, bucket 1 = everything else
        b0 = sizes[0] if named[0][0] == "rgb_encoder.params" else 0
        self.buckets = GradBuckets(self.flat_grad, [0, b0, total] if b0 else [0, total], group=self.group,
                                   solo=not forced)
        # Adam state: whole buffer on one GPU; with N ranks each rank keeps (and updates) only its
        # 1/N slice of every bucket — reduce-scatter gradients, Adam on the slice, all-gather params
        if self.sharded:
            self.shards = [self.buckets.shard_range(i) for i in range(len(self.buckets.bounds) - 1)]
            self.exp_avg = [torch.zeros(b - a, dtype=_f32, device=dev) for a, b in self.shards]
            self.exp_avg_sq = [torch.zeros(b - a, dtype=_f32, device=dev) for a, b in self.shards]
            # master copy of this rank's parameter slices and the landing buffers of the reduce-scatter
            self.param_shard = [self.flat_param[a:b].clone() for a, b in self.shards]
            self.grad_shard = [torch.zeros(b - a, dtype=_f32, device=dev) for a, b in self.shards]
        else:
            self.exp_avg = torch.zeros(total, dtype=_f32, device=dev)
            self.exp_avg_sq = torch.zeros(total, dtype=_f32, device=dev)
        # the field's weight products accumulate straight into these views of the flat gradient
        # (networks._FieldFn.backward); autograd then has nothing to add for them
        m = self.model
        import os as _os
        if hasattr(m, "xyz_net") and hasattr(m, "rgb_net") and _os.environ.get("NGP_NO_GRAD_SINKS", "0") != "1":
            lin1, lin2 = m.xyz_net[0], m.xyz_net[2]
            m._grad_sinks = {"W1": lin1.weight.grad, "b1": lin1.bias.grad, "W2": lin2.weight.grad, "b2": lin2.bias.grad,
                             "rgb_p": m.rgb_net.params.grad, "nrm_p": m.norm_pred_header.params.grad,
                             "sem_p": m.semantic_header.params.grad}
        # scatter kernels accumulate directly into the flat gradient (see tinycudann._GridFwd)
        for enc_name in ("rgb_encoder", "xyz_encoder"):
            enc = getattr(self.model, enc_name, None)
            if enc is not None:
                enc.grad_buffer = enc.params.grad
        # bucket 0's reduce-scatter is fired from the colour encoder's backward (overlaps the rest)
        self.hooked0 = bool(hasattr(self.model, "rgb_encoder") and self.sharded and b0)
        if self.hooked0:
            self.model.rgb_encoder.on_grad_ready = lambda: self.buckets.reduce_scatter_bucket(0, self.grad_shard[0])
            self.model.rgb_encoder.grad_ready_is_collective = True   # the field's backward then scatters colour first
        elif (hasattr(self.model, "rgb_encoder") and b0 and dev.type == "cuda" and not self.sharded
              and os.environ.get("NGP_NO_EARLY_NORM", "0") != "1"):
            # one GPU: the colour table's share of the gradient norm (77 % of the entries) is summed
            # right behind its scatter, beside the density head's backward, instead of on the path
            # between the last scatter and Adam
            self.model.rgb_encoder.on_grad_ready = self._early_norm_share
        # first element behind the two tables (the MLP parameters): their share of the norm is always summed exactly
        self._mlp_lo = 0
        if self.names[:2] == ["rgb_encoder.params", "xyz_encoder.params"] and len(self.names) > 2:
            self._mlp_lo = self.slices[self.names[2]][0]
        self.norm_bound = bool(self._mlp_lo and dev.type == "cuda" and not self.sharded and hasattr(self.model, "xyz_net")
                               and os.environ.get("NGP_NO_NORM_BOUND", "0") != "1")
        self._bound_step = False
        self._norm_share_armed = False   # step() arms it: exactly one backward per optimizer step
        self._norm_share_fired = 0

    def _unit_seed(self, terms):
        s = getattr(self, '_seed4', None)
        if s is None or s.device != terms.device:
            s = self._seed4 = torch.tensor([1.0, 0.0, 0.0, 0.0], device=terms.device)
        return s

    def _early_norm_share(self):
        if self._norm_share_armed and not self._bound_step:
            self._norm_share_fired += 1
            if self._norm_share_fired == 1:
                b0 = self.buckets.bounds[1]
                call("sumsq", self.flat_grad[0:b0], b0, self.scalars[0:1])

    # ------------------------------------------------------------------ schedule
    def lr_at(self, epoch):
        eta_min = self.base_lr / 30
        return eta_min + (self.base_lr - eta_min) * (1 + math.cos(math.pi * epoch / self.num_epochs)) / 2

    @property
    def lr(self):
        return self.lr_at(min(self.global_step // self.steps_per_epoch, self.num_epochs))

    def step(self, rays_o, rays_d, rgb_gt, next_rays=None, target=None, **loss_kwargs):
        """one training step on this rank's ray batch; returns (loss tensor, results dict).

        target: further per-ray supervision for NeRFLoss's optional terms ('normal', 'label', 'depth': the
        batch dictionary of train.py:299); loss_kwargs: per-step additions to the trainer's loss_kwargs
        (e.g. mask=..., step=...).

        next_rays = (rays_o, rays_d) of the FOLLOWING step, if the caller already has them (a
        data loader that is one batch ahead): their AABB test and occupancy march are then run on a
        side stream under this step's backward and the next call picks the result up, provided it
        is called with the very same tensors and no density-grid update lies in between."""
        model = self.model
        self._join_grad_zeroing()
        if self.global_step % self.update_interval == 0:
            model.update_density_grid(self.density_threshold * MAX_SAMPLES / 3 ** 0.5,
                                      warmup=self.global_step < self.warmup_steps)
        ahead = self._march_ahead
        marched = None
        launch_next_late = False
        if ahead is not None and next_rays is not None:
            marched = ahead.take(rays_o, rays_d, self.exp_step_factor)
            late = marched is None
            if late:   # nothing in flight for this batch: march it now, same route
                ahead.launch(model, rays_o, rays_d, self.exp_step_factor)
                marched = ahead.take(rays_o, rays_d, self.exp_step_factor)
            if (self.global_step + 1) % self.update_interval != 0:
                if late:   # the host has just waited for this batch's march (a step that starts with an occupancy update): the
                    launch_next_late = True   # device is idle until the forward is enqueued - the next batch's march goes behind it
                else:
                    ahead.launch(model, next_rays[0], next_rays[1], self.exp_step_factor)
        elif ahead is not None:
            marched = ahead.take(rays_o, rays_d, self.exp_step_factor)
        default_recipe = bool(self.fused_loss and not loss_kwargs and not target)
        extra = {}
        if default_recipe and self.fused_tail and rays_o.is_cuda:
            # render + loss + the loss's gradients as one launch behind the field (rendering._RenderLossFn)
            extra['_fused_loss'] = (rgb_gt, self.loss_fn.lambda_opa, self.loss_fn.lambda_distortion)
        results = render(model, rays_o, rays_d, exp_step_factor=self.exp_step_factor,
                         num_classes=self.num_classes, marched=marched, **self.render_kwargs, **extra)
        if launch_next_late:
            ahead.launch(model, next_rays[0], next_rays[1], self.exp_step_factor)
        self._norm_share_armed, self._norm_share_fired = True, 0   # one backward follows, then the optimizer step
        # clip_grad_norm_(50) from an upper bound of the norm (ngp_clip_decide) instead of the 0.8 GB sum-of-squares
        # pass: only on the default recipe, where the fused field backward is the one writer of the table gradients
        self._bound_step = bool(self.norm_bound and self.fused_loss and not loss_kwargs and not target
                                and not getattr(model, "differentiable_normals", False))
        model._norm_bound_acc = self.norm_acc if self._bound_step else None
        model._norm_bound_hits, model._norm_bound_ok = 0, True
        if self.norm_bound:
            model.rgb_encoder._bound_valid = model.xyz_encoder._bound_valid = True
        if '_loss_terms' in results:
            terms = results.pop('_loss_terms')
            loss = terms[0]
            torch.autograd.backward([terms], [self._unit_seed(terms)])
        elif self.fused_loss and not loss_kwargs and not target:
            # same value and gradients as sum(term.mean()) over NeRFLoss's default terms; the
            # gradients are seeded directly (no loss node, no multiplications by 1)
            terms, (d_rgb, d_op, d_ws) = nerf_loss_and_grads(
                results["rgb"], results["opacity"], results["ws"], results["deltas"], results["ts"],
                results["rays_a"], rgb_gt, self.loss_fn.lambda_opa, self.loss_fn.lambda_distortion)
            loss = terms[0]
            outs, seeds = [results["rgb"], results["opacity"]], [d_rgb, d_op]
            if d_ws is not None:
                outs.append(results["ws"])
                seeds.append(d_ws)
            torch.autograd.backward(outs, seeds)
        else:
            batch = {"rgb": rgb_gt}
            batch.update(target or {})
            kw = dict(self.loss_kwargs)
            kw.update(loss_kwargs)
            loss_d = self.loss_fn(results, batch, **kw)
            loss = sum(lo.mean() for lo in loss_d.values())
            loss.backward()
        self.optimizer_step()
        return loss.detach(), results

    def _adam_width_now(self, stream):
        """the sweep's launch width for this step; runs the measurement described in __init__ while it lasts"""
        if self.adam_width is not None:
            return self.adam_width
        start, rounds = self.adam_tune
        W, cands = self.update_interval, self.adam_candidates
        k = self.global_step - 1 - start          # (global_step was advanced already: 1-based here)
        n_win = rounds * len(cands)
        if k < 0:
            return cands[0]
        if not self._tune_events and k > 0:        # the start was missed (a resumed run): begin at the next window boundary
            self.adam_tune = (-(-(self.global_step - 1) // W) * W, rounds)
            return cands[0]
        if k % W == 0 and k // W <= n_win and len(self._tune_events) == k // W:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(stream)                      # window boundary: the same point of every step, on the optimizer stream
            self._tune_events.append(ev)
        if k < n_win * W:
            return cands[(k // W) % len(cands)]
        if len(self._tune_events) == n_win + 1 and self._tune_events[-1].query():
            ms = [a.elapsed_time(b) for a, b in zip(self._tune_events, self._tune_events[1:])]
            med = []
            for c in range(len(cands)):
                v = sorted(ms[c::len(cands)])
                med.append(0.5 * (v[(len(v) - 1) // 2] + v[len(v) // 2]))
            best = min(range(len(cands)), key=lambda c: med[c])
            # the default keeps its place unless another width is faster by more than the windows' own spread
            self.adam_width = cands[best] if med[best] < 0.99 * med[0] else cands[0]
            self.adam_tune_ms = dict(zip(cands, med))
            self._tune_events = []
            return self.adam_width
        return cands[0]

    def optimizer_step(self):
        world = self.buckets.world
        self.global_step += 1
        # lr of the epoch this step belongs to (the scheduler ticks at epoch boundaries)
        lr = self.lr_at(min((self.global_step - 1) // self.steps_per_epoch, self.num_epochs))
        if not self.sharded:
            # clip + Adam stream 6.4 GB and touch no ray data: they run on the optimizer stream, the
            # field waits on `_params_ready` / `_rgb_params_ready` where it first reads the respective
            # parameters (the ray-only front of the next step is on MarchAhead's stream anyway).
            n = self.flat_grad.numel()
            main = torch.cuda.current_stream()
            side = self._opt_stream
            side.wait_stream(main)
            b0 = self.buckets.bounds[1] if len(self.buckets.bounds) > 2 else 0
            early = self._norm_share_armed and self._norm_share_fired == 1   # scalars[0] holds the colour table's share
            self._norm_share_armed, self._norm_share_fired = False, 0
            m = self.model
            bounded = bool(self._bound_step and getattr(m, "_norm_bound_hits", 0) == 2 and getattr(m, "_norm_bound_ok", False)
                           and m.rgb_encoder._bound_valid and m.xyz_encoder._bound_valid)
            self._bound_step = False
            m._norm_bound_acc = None
            zero_after = False
            with torch.cuda.stream(side):
                if bounded:
                    lo = self._mlp_lo
                    Kp = m.rgb_net.padded_in
                    rgb_p, lin1, lin2 = m.rgb_net.params, m.xyz_net[0], m.xyz_net[2]
                    # (the MLP gradients' exact sum of squares is formed by the same launch, into scalars[0])
                    call("clip_decide_rest", self.norm_acc, rgb_p, 128 * Kp, rgb_p[128 * Kp:], rgb_p.numel() - 128 * Kp,
                         lin1.weight, lin1.weight.numel(), lin2.weight, lin2.weight.numel(), self.flat_grad[lo:n], n - lo,
                         self.scalars[0:1], float(self.clip_norm), 1.0, self.scalars[1:2], self.need_exact)
                    # bound >= clip_norm (not seen in training): the exact norm after all, decided on the device
                    call("sumsq_if", self.flat_grad[0:lo], lo, self.scalars[0:1], self.need_exact)
                    call("clip_coef_if", self.scalars[0:1], float(self.clip_norm), 1.0, self.scalars[1:2], self.need_exact)
                    zero_after = True      # the two accumulators are cleared behind the Adam launches, not before them
                elif early:
                    call("sumsq", self.flat_grad[b0:n], n - b0, self.scalars[0:1])
                else:
                    self.scalars[0:1].zero_()
                    call("sumsq", self.flat_grad, n, self.scalars[0:1])
                if not bounded:
                    call("clip_coef", self.scalars[0:1], float(self.clip_norm), 1.0, self.scalars[1:2])
                    self.scalars[0:1].zero_()   # ready for the next step's early share
                    self.norm_acc.zero_()
                # two pieces: [density table | MLPs] first — the next forward starts on them — then
                # the colour table, which the field does not read before its colour branch
                events = {}
                width = self._adam_width_now(side)
                for lo, hi in ((b0, n), (0, b0)):
                    if hi > lo:
                        call("adam_step_width", self.flat_param[lo:hi], self.flat_grad[lo:hi], self.exp_avg[lo:hi],
                             self.exp_avg_sq[lo:hi], hi - lo, float(lr), 0.9, 0.999, 1e-8, 0.0, self.global_step,
                             self.scalars[1:2], 1, width)
                    ev = torch.cuda.Event()
                    ev.record(side)
                    events[lo] = ev
                if zero_after:
                    self.scalars[0:1].zero_()
                    self.norm_acc.zero_()
                    # the next backward adds into both: it waits for this event (networks._FieldFn.backward)
                    ev = torch.cuda.Event()
                    ev.record(side)
                    self.model._acc_zeroed = ev
            self.model._params_ready, self.model._rgb_params_ready = events[b0], events[0]
            return
        self.scalars.zero_()
        # sharded: bucket 0's reduce-scatter was issued from the colour encoder's backward
        nb = len(self.buckets.bounds) - 1
        first = 1 if self.hooked0 else 0
        for i in range(first, nb):
            self.buckets.reduce_scatter_bucket(i, self.grad_shard[i])
        # global gradient norm = sqrt(sum over ranks of the shard sums).  Every shard is summed as soon as ITS
        # reduce-scatter has landed: the colour table's share (77 % of the entries) is read while the second bucket
        # is still on the wire (the one-GPU path's early norm share, kept in the sharded path).
        works = self.buckets.take_works()
        order = ([0] if self.hooked0 else []) + list(range(first, nb))
        assert len(works) == len(order)
        for w, i in zip(works, order):
            if w is not None:
                w.wait()
            g = self.grad_shard[i]
            call("sumsq", g, g.numel(), self.scalars[0:1])
        # all contributions are in the shard buffers now.  The scatter kernels add into flat_grad, so it has to be zero
        # again before the next backward: Adam clears only the rank's own 1/N (the shard buffers), the other (N-1)/N of
        # the accumulation buffer has no kernel that reads it on this rank — one 0.8 GB fill (0.15 ms of HBM writes) on
        # the optimizer stream, beside the norm / Adam of the slices and the all-gathers; the next backward joins it
        if self._opt_stream is not None:
            self._opt_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._opt_stream):
                self.flat_grad.zero_()
                self._grad_zeroed = torch.cuda.Event()
                self._grad_zeroed.record(self._opt_stream)
        else:
            self.flat_grad.zero_()
        dist.all_reduce(self.scalars[0:1], op=dist.ReduceOp.SUM, group=self.group)
        call("clip_coef", self.scalars[0:1], float(self.clip_norm), 1.0 / world, self.scalars[1:2])
        # per bucket: Adam on the slice, then publish it.  [density table | MLPs] first, the colour table
        # (77 % of the bytes) second; the field waits for each gather where it first reads those
        # parameters, so the small gather is in flight while the colour slice is still being updated and
        # the large one runs under the next step's marcher and density path
        works = {}
        for i in range(nb - 1, -1, -1):
            call("adam_step", self.param_shard[i], self.grad_shard[i], self.exp_avg[i], self.exp_avg_sq[i],
                 self.grad_shard[i].numel(), float(lr), 0.9, 0.999, 1e-8, 0.0, self.global_step, self.scalars[1:2], 0)
            works[i] = self.buckets.all_gather_bucket(i, self.flat_param, self.param_shard[i], detach=True)
        if nb == 2:
            self.model._params_ready, self.model._rgb_params_ready = works[1], works[0]
        else:
            self.model._params_ready = works[0]


    def _join_grad_zeroing(self):
        ev, self._grad_zeroed = self._grad_zeroed, None
        if ev is not None:
            ev.wait()

    def wait(self):
        """make the current stream wait for a pending side-stream optimizer step (call before reading
        parameters / gradients outside the model's own forward)"""
        self._join_grad_zeroing()
        for name in ("_params_ready", "_rgb_params_ready"):
            ev = getattr(self.model, name, None)
            if ev is not None:
                ev.wait()

    # ------------------------------------------------------------------ parameter store <-> shards
    def sync_shards(self):
        """Sharded optimizer only: the rank's master parameter slices (what Adam updates and the all-gather
        publishes) are re-read from the flat parameter buffer.  Call after ANY write into the model's
        parameters from outside the optimizer (checkpoint load, manual initialisation): otherwise the next
        all-gather overwrites those writes with the stale slices."""
        self.wait()
        if self.sharded:
            with torch.no_grad():
                for ps, (a, b) in zip(self.param_shard, self.shards):
                    ps.copy_(self.flat_param[a:b])

    def load_ckpt(self, ckpt_path, model_name='model', prefixes_to_ignore=()):
        """utils.load_ckpt into the trainer's flat parameter store (values are copied into the existing
        views, shapes are checked), followed by sync_shards()."""
        from .ckpt import load_ckpt
        self.wait()
        load_ckpt(self.model, ckpt_path, model_name, prefixes_to_ignore)
        self.sync_shards()

    # ------------------------------------------------------------------ multi-GPU helpers
    def broadcast_state(self, src=0):
        """start every rank from rank `src`'s parameters and occupancy grid (DDP does this at
        construction and re-broadcasts buffers every forward; train.py:431)"""
        if self.buckets.world == 1 and self.buckets.solo:
            return
        dist.broadcast(self.flat_param, src, group=self.group)
        self.sync_shards()
        for name in ("density_grid", "density_bitfield"):
            if hasattr(self.model, name):
                dist.broadcast(getattr(self.model, name), src, group=self.group)
