"""NGP field of the reference (models/networks.py:13-420) on the MI355X library.

Same constructor, buffers (`center, xyz_min, xyz_max, half_size, density_bitfield`; the trainer
adds `density_grid`, `grid_coords`), attributes (`cascades, scale, grid_size`), methods
(`density, grad, forward, forward_test, forward_skybox, get_all_cells,
sample_uniform_and_occupied_cells, mark_invisible_cells, update_density_grid`) and state-dict
keys (`xyz_encoder.params`, `xyz_net.0.weight`, `rgb_net.params`, ...), so reference
checkpoints map one to one.

Differences, all deliberate:
  * by default d(sigma)/dx is computed analytically in the forward pass (sigmoid-gated
    back-substitution through the 2-layer density MLP + the grid input-gradient kernel) instead of
    torch.autograd.grad(create_graph=True) (networks.py:186-196).  The values are identical; the
    result is detached, i.e. normals_raw carries no gradient — which is what every recipe without
    --normal_ref needs.  Setting `model.differentiable_normals = True` switches forward() to the
    reference's own formulation, with the grid double backward (H4) on ngp_grid_bwd_bwd_input.
"""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd import Function

from . import tinycudann as tcnn
from . import vren
from ._lib import call, call_host
from .custom_functions import TruncExp
from .rendering import NEAR_DISTANCE

_f32 = torch.float32
_SOFTPLUS = 3


class _LinearAct(Function):
    """y = act(x W^T + b) with nn.Linear parameter layout; also returns the pre-activation
    (non-differentiable output, needed by Softplus' backward and by the analytic d(sigma)/dx)."""

    @staticmethod
    def forward(ctx, x, W, b, act):
        x = x.contiguous()
        n, ni = x.shape
        no = W.shape[0]
        y = torch.empty(n, no, dtype=_f32, device=x.device)
        z = torch.empty(n, no, dtype=_f32, device=x.device)
        call("linear_fwd", x, ni, W, ni, b, n, ni, no, act, y, no, z)
        ctx.save_for_backward(x, W, y, z)
        ctx.act = act
        ctx.has_bias = b is not None
        ctx.mark_non_differentiable(z)
        return y, z

    @staticmethod
    def backward(ctx, dy, _dz):
        x, W, y, z = ctx.saved_tensors
        n, ni = x.shape
        no = W.shape[0]
        act = ctx.act
        dz = dy.contiguous()
        if act != 0:
            dz = torch.empty_like(dz)
            call("act_bwd", dy.contiguous(), y, dz.numel(), act, dz)
        dx = dW = db = None
        if ctx.needs_input_grad[1]:
            dW = torch.zeros_like(W)
            db = torch.zeros(no, dtype=_f32, device=x.device) if ctx.has_bias else None
            call("linear_bwd_weight", dz, no, x, ni, n, ni, no, dW, ni, db)
        if ctx.needs_input_grad[0]:
            dx = torch.empty(n, ni, dtype=_f32, device=x.device)
            call("linear_bwd_input", dz, no, W, ni, n, ni, no, dx, ni, 0)
        return dx, dW, db, None


_RELU, _NONE = 1, 0

import os as _os

_OVERLAP = _os.environ.get("NGP_NO_OVERLAP", "0") != "1"
_HEADS_BESIDE = _os.environ.get("NGP_NO_HEADS_BESIDE", "0") != "1"   # A/B switch: the two heads on a stream of their own beside rgb_net (-0.03 ms/step)
_FUSED_FWD = _os.environ.get("NGP_NO_FUSED_FWD", "0") != "1"   # A/B switch for ngp_mlp2_fwd
_FUSED_BWD = _os.environ.get("NGP_NO_FUSED_BWD", "0") != "1"   # A/B switch for the operand-transform products
# the library's streaming weight-gradient kernel (mlp_stream_wgrad_kernel) is on unless one of its A/B switches is set
_SORT_GRID_SAMPLES = _os.environ.get("NGP_NO_SORT_GRID_SAMPLES", "0") != "1"   # A/B: Morton-sorted occupancy-update points
_REUSE_DSIG_DFEAT = _os.environ.get("NGP_NO_REUSE_DFEAT", "0") != "1"           # A/B: second data-gradient product of the density head
_FUSED_GRID_UPDATE = _os.environ.get("NGP_GRID_UPDATE_TORCH", "0") != "1"       # A/B: the torch-op route of the sampled update
_STREAM_WGRAD = not (_os.environ.get("NGP_MLP_NO_STREAM") or _os.environ.get("NGP_MLP_NO_STREAM_WGRAD"))
# widest second layer that takes the fused route (tools/mlp_bwd_microbench.py, n = 433 k, MI355X):
# density head 0.54 -> 0.49 ms, rgb_net 0.60 -> 0.57 ms, 32-wide headers 0.22 -> 0.22 ms
_FUSED_BWD_MAX_OUT = int(_os.environ.get("NGP_FUSED_BWD_MAX_OUT", "3"))
_SIDE = {}
_SIDE_FWD = {}
_FWD_OVERLAP = _os.environ.get("NGP_NO_FWD_OVERLAP", "0") != "1"   # A/B: the colour branch of the forward on its own stream
# Colour branch on the live samples only (those up to their ray's early-termination point).  Exact and tested, but
# OFF by default: on the proxy scene 19 % of the samples are behind a stop, the step gains 0.05 ms in steady state
# (the forward gains nothing — the colour branch cannot start before sigma is known and becomes the longer chain —
# the backward loses 19 % of its colour-branch work) and the whole 20k-step schedule loses 1-4 % (no ray stops early
# in the first epochs, the host-side count and the three index launches cost the same).  NGP_COMPACT=1 or
# model.compact_dead_samples = True switches it on: scenes with solid interiors have far more dead samples.
_COMPACT = _os.environ.get("NGP_COMPACT", "0") == "1"
_SCATTER_AFTER_DGRAD = _os.environ.get("NGP_SCATTER_AFTER_DGRAD", "0") == "1"   # A/B: density scatter held back behind the colour data gradient


_PINNED = {}


def _pinned_word(dev):
    """one pinned int32 per device: the landing place of a device-side count the host has to read"""
    key = torch.device(dev).index
    t = _PINNED.get(key)
    if t is None:
        t = _PINNED[key] = torch.zeros(1, dtype=torch.int32).pin_memory()
    return t


def _fwd_stream(dev):
    key = torch.device(dev).index
    st = _SIDE_FWD.get(key)
    if st is None:
        st = _SIDE_FWD[key] = torch.cuda.Stream(device=dev)
    return st


_HEADS = {}


def _heads_stream(dev):
    key = torch.device(dev).index
    st = _HEADS.get(key)
    if st is None:
        st = _HEADS[key] = torch.cuda.Stream(device=dev)
    return st


def _side_stream(dev):
    key = torch.device(dev).index
    st = _SIDE.get(key)
    if st is None:
        st = _SIDE[key] = torch.cuda.Stream(device=dev, priority=int(_os.environ.get("NGP_SIDE_PRIO", "0")))
    return st


class _Mlp2Bwd:
    """Backward of a 2-layer MLP (x_in -> H hidden with act1 -> n_out with act2) on the library, in three stages
    the caller can interleave with other work: the constructor runs the elementwise stage (dz2, on the plain route
    also dz1), input_product() the data gradient dz1 . W1[:, cols], weight_products() dW1 / dW2 / db1 / db2.
    On the fused route dz1 = act1'(hidden) * (dz2 . W2) is formed inside the two first-layer products."""

    def __init__(self, d_out, out, ld_out, act2, W2, hidden, H, act1, n_out, x_in, ld_in, n_in, W1, ldw1, dW1, dW2,
                 db1, db2, norm_acc=None):
        """norm_acc: device scalar that takes sum_s ||dz2[s]|| (the trainer's norm-bound sums), accumulated by the pass
        that forms dz2 when that pass is the elementwise one (self.norm_noted says whether it was)"""
        self.a = (d_out, out, ld_out, act2, W2, hidden, H, act1, n_out, x_in, ld_in, n_in, W1, ldw1, dW1, dW2, db1, db2)
        self.norm_noted = False
        n = hidden.shape[0]
        dev = hidden.device
        self.n = n
        self.fused = bool(_FUSED_BWD and n_out <= _FUSED_BWD_MAX_OUT and d_out.stride(0) == n_out and ld_out == n_out)
        self.dz1 = None
        self.dw2_done = False
        if self.fused:
            self.dz2 = torch.empty(n, n_out, dtype=_f32, device=dev)
            if norm_acc is not None and n_out <= 4:
                call("act_bwd_rows", d_out, out, n, n_out, act2, self.dz2, norm_acc)
                self.norm_noted = True
            else:
                call("act_bwd", d_out, out, n * n_out, act2, self.dz2)
            return
        self.dz2 = torch.empty(n, 16 if n_out > 4 else 4, dtype=_f32, device=dev)
        self.dz1 = torch.empty(n, H, dtype=_f32, device=dev)
        if n_out <= 4:   # dW2 / db2 come out of the same pass over the hidden activations
            call("mlp_hidden_bwd", d_out, d_out.stride(0), out, ld_out, act2, W2, H, hidden, H, act1, n, H, n_out,
                 None, 0, self.dz1, H, dW2, H, db2)
            self.dw2_done = True
        else:
            call("mlp_hidden_bwd", d_out, d_out.stride(0), out, ld_out, act2, W2, H, hidden, H, act1, n, H, n_out,
                 self.dz2, self.dz2.shape[1], self.dz1, H, None, 0, None)

    def input_product(self, dx, ld_dx, dx_cols, w1_col0, accumulate):
        """dx (+)= dz1 . W1[:, w1_col0 : w1_col0 + dx_cols] (the input columns that need a gradient)"""
        (d_out, out, ld_out, act2, W2, hidden, H, act1, n_out, x_in, ld_in, n_in, W1, ldw1, dW1, dW2, db1, db2) = self.a
        w1_dx = W1[w1_col0:] if W1.dim() == 1 else W1
        if self.fused:
            call("mlp_bwd_input", self.dz2, n_out, W2, H, hidden, H, act1, w1_dx, ldw1, self.n, dx_cols, H, n_out, dx, ld_dx,
                 1 if accumulate else 0)
        else:
            call("linear_bwd_input", self.dz1, H, w1_dx, ldw1, self.n, dx_cols, H, dx, ld_dx, 1 if accumulate else 0)

    def weight_products(self):
        (d_out, out, ld_out, act2, W2, hidden, H, act1, n_out, x_in, ld_in, n_in, W1, ldw1, dW1, dW2, db1, db2) = self.a
        n = self.n
        wide = n_in > 128 and n_in % 128 <= 32
        # 128 + remainder columns: a second launch with 128x32 tiles instead of a mostly empty 128x128
        # tile (rgb_net's first layer is 128 x 144/160)
        rem = n_in - 128
        x_rem = x_in[:, 128:] if x_in.dim() == 2 else x_in
        dW1_rem = dW1[128:] if dW1.dim() == 1 else dW1[:, 128:]
        if self.fused:
            dz2 = self.dz2
            # the first-layer weight product streams dz2 and hidden anyway: it also leaves dW2 / db2
            if wide and _STREAM_WGRAD and H == 128 and n_in in (144, 160) and act1 in (_RELU, _SOFTPLUS):
                # the streaming kernel takes all 144 / 160 input columns (and dW2 / db2) in one pass
                call("mlp_bwd_weight", dz2, n_out, W2, H, hidden, H, act1, x_in, ld_in, n, n_in, H, n_out, dW1, ldw1, db1,
                     dW2, H, db2)
            elif wide:
                # dW2 / db2 ride with the narrow remainder launch (16 accumulator registers per lane; in the
                # 128x128 one the extra partial sums would spill at 3 workgroups per CU)
                call("mlp_bwd_weight", dz2, n_out, W2, H, hidden, H, act1, x_in, ld_in, n, 128, H, n_out, dW1, ldw1, db1,
                     None, 0, None)
                call("mlp_bwd_weight", dz2, n_out, W2, H, hidden, H, act1, x_rem, ld_in, n, rem, H, n_out, dW1_rem, ldw1,
                     None, dW2, H, db2)
            else:
                call("mlp_bwd_weight", dz2, n_out, W2, H, hidden, H, act1, x_in, ld_in, n, n_in, H, n_out, dW1, ldw1, db1,
                     dW2, H, db2)
            return
        if not self.dw2_done:
            call("linear_bwd_weight", self.dz2, self.dz2.shape[1], hidden, H, n, H, n_out, dW2, H, db2)
        if wide:
            call("linear_bwd_weight", self.dz1, H, x_in, ld_in, n, 128, H, dW1, ldw1, db1)
            call("linear_bwd_weight", self.dz1, H, x_rem, ld_in, n, rem, H, dW1_rem, ldw1, None)
        else:
            call("linear_bwd_weight", self.dz1, H, x_in, ld_in, n, n_in, H, dW1, ldw1, db1)


class _NegNormalize(Function):
    """y = -F.normalize(x * scale3, p=2, dim=-1, eps=1e-6) on (N,3) rows, one launch each way."""

    @staticmethod
    def forward(ctx, x, scale3):
        x = x.contiguous()
        y = torch.empty(x.shape[0], 3, dtype=_f32, device=x.device)
        call("neg_normalize", x, 3, scale3, x.shape[0], y)
        if scale3 is None:
            ctx.save_for_backward(x)
        ctx.set_materialize_grads(False)
        return y

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        call("neg_normalize_bwd", x, 3, g.contiguous(), x.shape[0], dx)
        return dx, None


def _wait_params(model, rgb_table=True):
    """The trainer runs clip + Adam on a side stream in two pieces — [density table | MLPs], then
    the colour table (77 % of the bytes) — and anything that reads parameters first waits for the
    piece it needs: the field starts on the density path while the colour table is still being
    updated.  rgb_table=False waits for the first piece only."""
    # either a HIP event (single GPU: Adam on the optimizer stream) or the handle of an async
    # all-gather of the updated parameter shards (sharded optimizer): both make the current stream
    # wait through .wait()
    ev = getattr(model, "_params_ready", None)
    if ev is not None:
        ev.wait()
        model._params_ready = None
    if rgb_table:
        ev = getattr(model, "_rgb_params_ready", None)
        if ev is not None:
            ev.wait()
            model._rgb_params_ready = None


def _bound_note(model, st, slot, n_out, n):
    """The trainer's clip decision (ngp_clip_decide) bounds a table gradient's norm by ||W1||_F ||W2||_F sum_s ||dz2[s]||:
    accumulate the sum for the MLP whose backward `st` is (slot 0: rgb_net, 1: density head)."""
    acc = getattr(model, "_norm_bound_acc", None)
    if acc is None:
        return
    if not st.fused:
        model._norm_bound_ok = False
        return
    if not st.norm_noted:      # (the elementwise stage adds the sum itself when it is handed the accumulator)
        call("row_norm_sum", st.dz2, n_out, n, n_out, acc[slot:slot + 1])
    model._norm_bound_hits = getattr(model, "_norm_bound_hits", 0) + 1


class _FieldFn(Function):
    """The whole NGP field (networks.py:198-240) as one autograd node: explicit kernel launches on
    preallocated buffers, no concat (both encoders write straight into rgb_net's input matrix),
    analytic d(sigma)/dx, and a backward that runs only the branches whose outputs received a
    gradient (normal / semantic heads are skipped when the loss does not use them).

    inputs : x (N,3) world, d (N,3), embed_a (N,E) or None, then the 9 parameter tensors
    outputs: sigma (N), rgb (N,3) [after rgb_net's output activation], dsigma/dxn (N,3) w.r.t. the
             normalised position (no grad; divide by xyz_max-xyz_min for world units),
             normal head (N,3) raw, semantic logits (N,C)
    """

    @staticmethod
    def forward(ctx, model, x, d, embed_a, xyz_table, W1, b1, W2, b2, rgb_table, rgb_p, nrm_p, sem_p):
        n = x.shape[0]
        dev = x.device
        xe, re = model.xyz_encoder, model.rgb_encoder
        C = model.semantic_header.n_output_dims
        E = 0 if embed_a is None else embed_a.shape[1]
        K = 144 + E
        Kp = model.rgb_net.padded_in
        span = model._span()
        xn = (x - model.xyz_min).div_(span)   # (before the waits below: it reads no parameter and runs under the Adam sweep)
        # the two pieces of the trainer's Adam sweep (or of the sharded optimizer's all-gather): every stream below waits
        # for a piece where it first reads that piece's parameters
        ev_p, ev_c = getattr(model, "_params_ready", None), getattr(model, "_rgb_params_ready", None)
        model._params_ready = model._rgb_params_ready = None
        # Samples behind their ray's early-termination point take no part in the image and get no gradient: when the
        # renderer hands over the rays' segments (model._live_ctx), the colour branch and its backward run on the
        # live samples only.  The density head needs every sample (the stop depends on sigma); the list of live rows
        # comes from the compositor's own bookkeeping (ngp_live_rows), its length reaches the host through a pinned
        # word while the device works on the analytic normals.
        live = getattr(model, "_live_ctx", None)
        want = getattr(model, "compact_dead_samples", None)
        compact = bool((_COMPACT if want is None else want) and live is not None and n > 0 and embed_a is None
                       and x.is_cuda and not ctx.needs_input_grad[1])
        # every buffer comes from the caller's stream (the allocator then knows them as that stream's; the colour
        # stream below only launches into them and is joined before anything is returned)
        feat = torch.empty(n, 128, dtype=_f32, device=dev)
        a1 = torch.empty(n, 128, dtype=_f32, device=dev)
        sig = torch.empty(n, 1, dtype=_f32, device=dev)
        dfeat = torch.empty(n, 128, dtype=_f32, device=dev)
        grads = torch.empty(n, 3, dtype=_f32, device=dev)   # d sigma / d xn (normalised coordinates)
        rgb_o = torch.empty(n, 3, dtype=_f32, device=dev)
        np_o = torch.empty(n, 3, dtype=_f32, device=dev)
        sem_o = torch.empty(n, C, dtype=_f32, device=dev)
        dz2 = torch.empty(n, 1, dtype=_f32, device=dev) if _FUSED_BWD else None
        dz1 = None if _FUSED_BWD else torch.empty(n, 128, dtype=_f32, device=dev)
        net = model.rgb_net
        bufs = {}

        def colour_buffers(m):
            bufs["rgb_in"] = torch.empty(m, Kp, dtype=_f32, device=dev)
            bufs["a_r"] = torch.empty(m, 128, dtype=_f32, device=dev)
            bufs["a_n"] = torch.empty(m, 32, dtype=_f32, device=dev)
            bufs["a_s"] = torch.empty(m, 32, dtype=_f32, device=dev)
            if compact:   # compacted outputs; rgb_o / np_o / sem_o are filled from them
                bufs["rgb_c"] = torch.empty(m, 3, dtype=_f32, device=dev)
                bufs["np_c"] = torch.empty(m, 3, dtype=_f32, device=dev)
                bufs["sem_c"] = torch.empty(m, C, dtype=_f32, device=dev)

        def colour_branch():
            # [SH(16) | rgb grid features (128) | appearance code (E) | ones-padding] -> rgb_net, the two heads
            n, rgb_in, a_r, a_n, a_s = bufs["n"], bufs["rgb_in"], bufs["a_r"], bufs["a_n"], bufs["a_s"]
            feat_rgb = rgb_in[:, 16:144]
            if compact:   # (positions and directions came over with the list, see ngp_live_rows)
                xn, d = bufs["xn_c"][:n], bufs["d_c"][:n]
                rgb_o, np_o, sem_o = bufs["rgb_c"], bufs["np_c"], bufs["sem_c"]
            else:
                xn, d = bufs["xn_full"], bufs["d_full"]
                rgb_o, np_o, sem_o = bufs["rgb_o"], bufs["np_o"], bufs["sem_o"]
            call("sh_fwd_dirs", d, n, 4, rgb_in, Kp)
            if ev_c is not None:
                ev_c.wait()   # the colour table's piece (the stream this runs on waits for it)
            call("grid_fwd", re.desc, rgb_table, xn, n, rgb_in[:, 16:], Kp)
            if ev_p is not None:
                ev_p.wait()   # the MLPs' piece
            if E:
                rgb_in[:, 144:K] = embed_a
            if Kp > K:
                rgb_in[:, K:] = 1.0
            heads = None
            if _HEADS_BESIDE and _FUSED_FWD and C <= 8 and x.is_cuda:
                # the two 32-wide heads read the same rows as rgb_net: on a stream of their own their 256-thread
                # workgroups (72 registers) fit beside the 8-wave rgb_net workgroup on every CU
                cur = torch.cuda.current_stream()
                heads = _heads_stream(dev)
                heads.wait_stream(cur)
                with torch.cuda.stream(heads):
                    call("mlp2_fwd", feat_rgb, Kp, nrm_p, 128, None, _RELU, nrm_p[32 * 128:], 32, None, _NONE,
                         n, 128, 32, 3, a_n, 32, np_o, 3)
                    call("mlp2_fwd", feat_rgb, Kp, sem_p, 128, None, _RELU, sem_p[32 * 128:], 32, None, _NONE,
                         n, 128, 32, C, a_s, 32, sem_o, C)
                call("mlp2_fwd", rgb_in, Kp, rgb_p, Kp, None, _RELU, rgb_p[128 * Kp:], 128, None, net.output_activation,
                     n, Kp, 128, 3, a_r, 128, rgb_o, 3)
                cur.wait_stream(heads)
            elif _FUSED_FWD:
                call("mlp2_fwd", rgb_in, Kp, rgb_p, Kp, None, _RELU, rgb_p[128 * Kp:], 128, None, net.output_activation,
                     n, Kp, 128, 3, a_r, 128, rgb_o, 3)
                call("mlp2_fwd", feat_rgb, Kp, nrm_p, 128, None, _RELU, nrm_p[32 * 128:], 32, None, _NONE,
                     n, 128, 32, 3, a_n, 32, np_o, 3)
            else:
                call("linear_fwd", rgb_in, Kp, rgb_p, Kp, None, n, Kp, 128, _RELU, a_r, 128, None)
                call("linear_fwd", a_r, 128, rgb_p[128 * Kp:], 128, None, n, 128, 3, net.output_activation, rgb_o, 3, None)
                call("linear_fwd", feat_rgb, Kp, nrm_p, 128, None, n, 128, 32, _RELU, a_n, 32, None)
                call("linear_fwd", a_n, 32, nrm_p[32 * 128:], 32, None, n, 32, 3, _NONE, np_o, 3, None)
            if heads is not None:
                pass
            elif _FUSED_FWD and C <= 8:
                call("mlp2_fwd", feat_rgb, Kp, sem_p, 128, None, _RELU, sem_p[32 * 128:], 32, None, _NONE,
                     n, 128, 32, C, a_s, 32, sem_o, C)
            else:
                call("linear_fwd", feat_rgb, Kp, sem_p, 128, None, n, 128, 32, _RELU, a_s, 32, None)
                call("linear_fwd", a_s, 32, sem_p[32 * 128:], 32, None, n, 32, C, _NONE, sem_o, C, None)
            if compact:   # back to sample order, zeros behind the stops
                call("spread_rows3", rgb_o, 3, bufs["rgb_o"], np_o, 3, bufs["np_o"], sem_o, C, bufs["sem_o"],
                     bufs["inv_idx"], bufs["n_full"])

        bufs.update(xn_full=xn, d_full=d, rgb_o=rgb_o, np_o=np_o, sem_o=sem_o, n_full=n, n=n)
        # The colour branch needs the positions and the colour table, nothing of the density path: it runs on a
        # stream of its own from the moment the colour table's Adam piece is done, beside the rest of the density
        # path and the analytic normals (which are stretched beyond that piece's end on one stream).  (Compacted: it
        # also needs the list of live rows, i.e. sigma — it starts behind the density head, beside the normals.)
        main = torch.cuda.current_stream()
        side = _fwd_stream(dev) if (_FWD_OVERLAP and x.is_cuda) else None
        if side is not None and not compact:
            colour_buffers(n)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                colour_branch()

        # density head
        if ev_p is not None:
            ev_p.wait()
        call("grid_fwd", xe.desc, xyz_table, xn, n, feat, 128)
        if _FUSED_FWD and _FUSED_BWD:
            # both layers in one launch, the 1-wide second layer in the MFMA epilogue, which also leaves
            # dz2 = softplus'(z2) = 1 - exp(-sigma): the start of the d(sigma)/dx pass below (no act_bwd launch)
            call("mlp2_fwd_dact", feat, 128, W1, 128, b1, _SOFTPLUS, W2, 128, b2, _SOFTPLUS, n, 128, 128, 1, a1, 128, sig, 1,
                 dz2)
        elif _FUSED_FWD:
            call("mlp2_fwd", feat, 128, W1, 128, b1, _SOFTPLUS, W2, 128, b2, _SOFTPLUS, n, 128, 128, 1, a1, 128, sig, 1)
        else:
            call("linear_fwd", feat, 128, W1, 128, b1, n, 128, 128, _SOFTPLUS, a1, 128, None)
            call("linear_fwd", a1, 128, W2, 128, b2, n, 128, 1, _SOFTPLUS, sig, 1, None)
        live_ev = None
        if compact:
            rays_a, deltas, T_thr = live
            n_rays = rays_a.shape[0]
            offsets = torch.empty(n_rays, dtype=torch.int32, device=dev)
            bufs["live_idx"] = torch.empty(n, dtype=torch.int32, device=dev)
            # -1 = "not in the list": ngp_live_rows writes only the rows its rays_a segments cover, the spread kernels read all n
            bufs["inv_idx"] = torch.full((n,), -1, dtype=torch.int32, device=dev)
            n_live_dev = torch.empty(1, dtype=torch.int32, device=dev)
            bufs["xn_c"] = torch.empty(n, 3, dtype=_f32, device=dev)   # the first n_live rows are used
            bufs["d_c"] = torch.empty(n, 3, dtype=_f32, device=dev)
            call("live_rows", sig, deltas.contiguous(), rays_a, float(T_thr), n_rays, offsets, bufs["live_idx"],
                 bufs["inv_idx"], n_live_dev, xn, bufs["xn_c"], d, bufs["d_c"])
            n_live_host = _pinned_word(dev)
            n_live_host.copy_(n_live_dev, non_blocking=True)
            live_ev = torch.cuda.Event()
            live_ev.record(main)
        # analytic d(sigma)/dx: back-substitute ones through the head, then the grid input gradient
        if _FUSED_BWD:
            if not _FUSED_FWD:
                call("act_bwd", None, sig, n, _SOFTPLUS, dz2)      # upstream gradient = ones
            call("mlp_bwd_input", dz2, 1, W2, 128, a1, 128, _SOFTPLUS, W1, 128, n, 128, 128, 1, dfeat, 128, 0)
        else:
            call("mlp_hidden_bwd", None, 0, sig, 1, _SOFTPLUS, W2, 128, a1, 128, _SOFTPLUS, n, 128, 1, None, 0, dz1, 128,
                 None, 0, None)
            call("linear_bwd_input", dz1, 128, W1, 128, n, 128, 128, dfeat, 128, 0)
        call("grid_bwd_input", xe.desc, xyz_table, xn, dfeat, 128, n, grads)
        # dfeat = d(sigma)/d(features) is kept: the density head has ONE output, so the gradient the backward
        # sends into the density encoder is d_sigma[s] * dfeat[s] — no second data-gradient product there

        if compact:
            # the host learns the number of live rows while the device is busy with the normals (and the optimizer
            # stream with the colour table); the colour branch is then enqueued with exactly that many rows
            live_ev.synchronize()
            n_c = int(n_live_host[0])
            bufs["n"] = n_c
            colour_buffers(n_c)
            if side is not None:
                side.wait_event(live_ev)
                with torch.cuda.stream(side):
                    colour_branch()
            else:
                colour_branch()
        if side is not None:
            main.wait_stream(side)
        elif not compact:
            colour_buffers(n)
            colour_branch()
        rgb_in, a_r, a_n, a_s = bufs["rgb_in"], bufs["a_r"], bufs["a_n"], bufs["a_s"]

        ctx.model = model
        ctx.E, ctx.K, ctx.Kp, ctx.C = E, K, Kp, C
        ctx.compact = compact
        if compact:
            ctx.live = (bufs["live_idx"], bufs["xn_c"][:bufs["n"]], bufs["rgb_c"], bufs["np_c"], bufs["sem_c"], bufs["n"])
        ctx.save_for_backward(xn, feat, a1, sig, rgb_in, a_r, rgb_o, a_n, np_o, a_s, sem_o,
                              xyz_table, W1, W2, rgb_table, rgb_p, nrm_p, sem_p, dfeat)
        ctx.mark_non_differentiable(grads)
        ctx.set_materialize_grads(False)
        return sig[:, 0], rgb_o, grads, np_o, sem_o

    @staticmethod
    def backward(ctx, d_sig, d_rgb, _d_grads, d_np, d_sem):
        (xn, feat, a1, sig, rgb_in, a_r, rgb_o, a_n, np_o, a_s, sem_o,
         xyz_table, W1, W2, rgb_table, rgb_p, nrm_p, sem_p, dsig_dfeat) = ctx.saved_tensors
        model = ctx.model
        E, K, Kp, C = ctx.E, ctx.K, ctx.Kp, ctx.C
        n = xn.shape[0]
        dev = xn.device
        xe, re = model.xyz_encoder, model.rgb_encoder
        need = ctx.needs_input_grad  # (model, x, d, embed_a, xyz_table, W1, b1, W2, b2, rgb_table, rgb_p, nrm_p, sem_p)
        g_x = g_emb = g_xyz = g_W1 = g_b1 = g_W2 = g_b2 = g_rgbt = g_rgbp = g_nrm = g_sem = None
        span = model._span()
        # A trainer that owns the gradient storage (NGPTrainer: one flat buffer, zeroed by its Adam
        # launch) registers the MLP gradients as sinks: the weight products accumulate straight into
        # them and autograd gets None — no zeros_like fill, no AccumulateGrad add per parameter.
        sinks = getattr(model, "_grad_sinks", None)

        def grad_buffer(name, like):
            t = None if sinks is None else sinks.get(name)
            if t is not None:
                return t, None          # (accumulate here, return nothing to autograd)
            z = torch.zeros(like, dtype=_f32, device=dev) if isinstance(like, tuple) else torch.zeros_like(like)
            return z, z

        # Schedule.  The two table scatters are bound by memory-side atomic requests, the MLP products by the
        # matrix pipe: the scatters go to a side stream, one behind the other, and the products run beside them.
        #   side: [density scatter] -> [colour scatter (+ its share of the gradient norm / its reduce-scatter)]
        #   main: colour data gradients -> (fork) -> colour weight gradients -> density weight gradients -> join
        # The density scatter can start at once: its input is d_sigma[s] * d(sigma)/d(features)[s], and the second
        # factor was computed (and kept) by the forward pass for the analytic normals.
        main = torch.cuda.current_stream()
        side = _side_stream(dev) if _OVERLAP else None
        forked = False
        ev = getattr(model, "_acc_zeroed", None)   # the trainer clears its norm accumulators behind the Adam launches
        if ev is not None:
            ev.wait()
            model._acc_zeroed = None

        def on_side(fn):
            nonlocal forked
            if side is None:
                fn()
                return
            side.wait_stream(main)
            with torch.cuda.stream(side):
                fn()
            forked = True

        def table_buffer(enc, table):
            buf = getattr(enc, "grad_buffer", None)
            if buf is None:
                buf = torch.zeros_like(table)
                return buf, buf
            return buf, None

        reuse = d_sig is not None and _REUSE_DSIG_DFEAT and not need[1]
        if reuse and need[4]:
            d_sig_c = d_sig.contiguous()
            buf, g_xyz = table_buffer(xe, xyz_table)

            def density_scatter(n=n):   # (bound now: the colour branch below may work on fewer rows)
                call("grid_bwd_param_scaled", xe.desc, xn, dsig_dfeat, 128, d_sig_c, n, buf)
                cb = getattr(xe, "on_grad_ready", None)
                if cb is not None:
                    cb()
            # Data parallel: the colour table's gradient (77 % of the bytes) goes into its reduce-scatter the moment
            # its scatter is enqueued, so there the colour scatter leads on the side stream and the density scatter
            # follows it — the collective then runs under the density scatter and the weight products.
            density_later = bool(getattr(re, "grad_ready_is_collective", False)) and d_rgb is not None and need[9]
            # One GPU: the density scatter leads on the side stream and is let loose at once.  It becomes ready together
            # with the colour branch's data gradient, and whichever reaches the CUs first keeps them (the product takes
            # 0.15 ms when its 256 large workgroups are placed first, 0.4-0.6 ms behind the scatter's thousands of
            # small ones).  Holding the scatter back behind the product (NGP_SCATTER_AFTER_DGRAD=1) makes that
            # deterministic and is still the slower schedule: +0.04 ms/step (A/B, 3 alternations of 100 steps), the
            # scatters are the longer path and every microsecond they start later is lost.
            density_after_dgrad = not density_later and d_rgb is not None and _SCATTER_AFTER_DGRAD
            if not density_later and not density_after_dgrad:
                on_side(density_scatter)
                density_scatter = None
        else:
            density_scatter = None
            density_after_dgrad = False

        # ---- colour branch (rgb_net + the two heads): data gradients w.r.t. [grid features | appearance code] first
        # (compacted forward: the branch's activations hold the live rows only; the upstream gradients are brought
        # into that order, rows behind a stop carry exact zeros anyway)
        xn_col, n_col = xn, n
        if ctx.compact:
            live_idx, xn_col, rgb_c, np_c, sem_c, n_col = ctx.live
            rgb_o, np_o, sem_o = rgb_c, np_c, sem_c

            def to_live(t, cols):
                if t is None:
                    return None
                out = torch.empty(n_col, cols, dtype=_f32, device=dev)
                call("gather_rows", t.contiguous(), cols, cols, live_idx, n_col, out, cols)
                return out
            d_rgb, d_np, d_sem = to_live(d_rgb, 3), to_live(d_np, 3), to_live(d_sem, C)
        n_full, n = n, n_col
        dfeat_rgb = None
        W_cols = 128 + E
        stages = []
        rgb_stage = None
        if d_rgb is not None:
            acc_rgbp, g_rgbp = grad_buffer("rgb_p", rgb_p)
            dfeat_rgb = torch.empty(n, W_cols, dtype=_f32, device=dev)
            nb_acc = getattr(model, "_norm_bound_acc", None)
            st = _Mlp2Bwd(d_rgb.contiguous(), rgb_o, 3, model.rgb_net.output_activation, rgb_p[128 * Kp:], a_r, 128,
                          _RELU, 3, rgb_in, Kp, Kp, rgb_p, Kp, acc_rgbp, acc_rgbp[128 * Kp:], None, None,
                          norm_acc=None if nb_acc is None else nb_acc[0:1])
            st.input_product(dfeat_rgb, W_cols, W_cols, 16, False)
            stages.append(st)
            rgb_stage = st
        for d_o, p, a_h, out, n_out, slot in ((d_np, nrm_p, a_n, np_o, 3, "nrm"), (d_sem, sem_p, a_s, sem_o, C, "sem")):
            if d_o is None:
                continue
            model._norm_bound_ok = False    # a head adds to the colour features' gradient: outside the norm bound
            acc_p, g_p = grad_buffer(slot + "_p", p)
            first = dfeat_rgb is None
            if first:
                dfeat_rgb = torch.zeros(n, W_cols, dtype=_f32, device=dev) if E else torch.empty(n, W_cols, dtype=_f32, device=dev)
            st = _Mlp2Bwd(d_o.contiguous(), out, n_out, _NONE, p[32 * 128:], a_h, 32, _RELU, n_out,
                          rgb_in[:, 16:], Kp, 128, p, 128, acc_p, acc_p[32 * 128:], None, None)
            st.input_product(dfeat_rgb, W_cols, 128, 0, not first)
            stages.append(st)
            if slot == "nrm":
                g_nrm = g_p
            else:
                g_sem = g_p

        if density_after_dgrad and density_scatter is not None:
            on_side(density_scatter)
            density_scatter = None
        if dfeat_rgb is not None and need[9]:
            buf_c, g_rgbt = table_buffer(re, rgb_table)

            def colour_scatter():
                call("grid_bwd_param", re.desc, xn_col, dfeat_rgb, W_cols, n, buf_c)
                cb = getattr(re, "on_grad_ready", None)
                if cb is not None:
                    cb()
            if d_sig is not None:
                on_side(colour_scatter)
            else:
                colour_scatter()
        if density_scatter is not None:
            on_side(density_scatter)
        if dfeat_rgb is not None:
            if E and need[3]:
                g_emb = dfeat_rgb[:, 128:]
            if need[1]:
                g_x = torch.empty(n, 3, dtype=_f32, device=dev)
                call("grid_bwd_input", re.desc, rgb_table, xn, dfeat_rgb, W_cols, n, g_x)
        for st in stages:
            st.weight_products()
        n = n_full

        # ---- density head
        if d_sig is not None:
            (acc_W1, g_W1), (acc_W2, g_W2) = grad_buffer("W1", W1), grad_buffer("W2", W2)
            (acc_b1, g_b1), (acc_b2, g_b2) = grad_buffer("b1", (128,)), grad_buffer("b2", (1,))
            nb_acc = getattr(model, "_norm_bound_acc", None)
            st = _Mlp2Bwd(d_sig.contiguous().view(n, 1), sig, 1, _SOFTPLUS, W2, a1, 128, _SOFTPLUS, 1,
                          feat, 128, 128, W1, 128, acc_W1, acc_W2, acc_b1, acc_b2,
                          norm_acc=None if nb_acc is None else nb_acc[1:2])
            if not reuse:
                dfeat = torch.empty(n, 128, dtype=_f32, device=dev)
                st.input_product(dfeat, 128, 128, 0, False)
            st.weight_products()
            _bound_note(model, st, 1, 1, n)
            if not reuse:
                if need[4]:
                    buf, g_xyz = table_buffer(xe, xyz_table)
                    call("grid_bwd_param", xe.desc, xn, dfeat, 128, n, buf)
                    cb = getattr(xe, "on_grad_ready", None)
                    if cb is not None:
                        cb()
                if need[1]:
                    gx2 = torch.empty(n, 3, dtype=_f32, device=dev)
                    call("grid_bwd_input", xe.desc, xyz_table, xn, dfeat, 128, n, gx2)
                    g_x = gx2 if g_x is None else g_x + gx2
        # the norm-bound sums feed the optimizer's clip decision only: behind the weight products, where they fill the
        # wait for the scatters instead of sitting between the data gradient and the weight gradient (67 us there)
        if rgb_stage is not None:
            _bound_note(model, rgb_stage, 0, 3, rgb_stage.n)
        if g_x is not None:
            g_x = g_x / span
        if forked:
            main.wait_stream(side)
        return (None, g_x, None, g_emb, g_xyz, g_W1, g_b1, g_W2, g_b2, g_rgbt, g_rgbp, g_nrm, g_sem)


class NGP(nn.Module):
    def __init__(self, scale, rgb_act='Sigmoid', use_skybox=False, embed_a=False, embed_a_len=12, classes=7):
        super().__init__()
        self.rgb_act = rgb_act
        self.scale = scale
        self.use_skybox = use_skybox
        self.embed_a = embed_a
        self.register_buffer('center', torch.zeros(1, 3))
        self.register_buffer('xyz_min', -torch.ones(1, 3) * scale)
        self.register_buffer('xyz_max', torch.ones(1, 3) * scale)
        self.register_buffer('half_size', (self.xyz_max - self.xyz_min) / 2)

        # cascade k of the occupancy grid covers [-2^(k-1), 2^(k-1)]^3
        self.cascades = max(1 + int(np.ceil(np.log2(2 * scale))), 1)
        self.grid_size = 128
        self.register_buffer('density_bitfield',
                             torch.zeros(self.cascades * self.grid_size ** 3 // 8, dtype=torch.uint8))

        L, Fdim, log2_T, N_min = 16, 8, 19, 16
        b = np.exp(np.log(2048 * scale / N_min) / (L - 1))
        self.xyz_encoder = tcnn.Encoding(3, {
            "otype": "Grid", "type": "Hash", "n_levels": L, "n_features_per_level": Fdim,
            "log2_hashmap_size": log2_T, "base_resolution": N_min, "per_level_scale": b,
            "interpolation": "Linear"})
        self.xyz_net = nn.Sequential(
            nn.Linear(self.xyz_encoder.n_output_dims, 128),
            nn.Softplus(),
            nn.Linear(128, 1))
        self.sigma_act = nn.Softplus()

        L_, F_, log2_T_, N_min_ = 16, 8, 21, 16
        b_ = np.exp(np.log(2048 * scale / N_min_) / (L_ - 1))
        self.rgb_encoder = tcnn.Encoding(3, {
            "otype": "HashGrid", "n_levels": L_, "n_features_per_level": F_,
            "log2_hashmap_size": log2_T_, "base_resolution": N_min_, "per_level_scale": b_,
            "interpolation": "Linear"}, seed=1338)
        self.dir_encoder = tcnn.Encoding(3, {"otype": "SphericalHarmonics", "degree": 4})

        rgb_input_dim = self.rgb_encoder.n_output_dims + self.dir_encoder.n_output_dims
        self.rgb_net = tcnn.Network(
            n_input_dims=rgb_input_dim + embed_a_len if embed_a else rgb_input_dim, n_output_dims=3,
            network_config={"otype": "CutlassMLP", "activation": "ReLU", "output_activation": rgb_act,
                            "n_neurons": 128, "n_hidden_layers": 1}, seed=1339)
        self.norm_pred_header = tcnn.Network(
            n_input_dims=self.rgb_encoder.n_output_dims, n_output_dims=3,
            network_config={"otype": "CutlassMLP", "activation": "ReLU", "output_activation": "None",
                            "n_neurons": 32, "n_hidden_layers": 1}, seed=1340)
        self.semantic_header = tcnn.Network(
            n_input_dims=self.rgb_encoder.n_output_dims, n_output_dims=classes,
            network_config={"otype": "CutlassMLP", "activation": "ReLU", "output_activation": "None",
                            "n_neurons": 32, "n_hidden_layers": 1}, seed=1341)
        self.semantic_act = nn.Softmax(dim=-1)

        if use_skybox:
            self.skybox_dir_encoder = tcnn.Encoding(3, {"otype": "SphericalHarmonics", "degree": 3})
            self.skybox_rgb_net = tcnn.Network(
                n_input_dims=9, n_output_dims=3,
                network_config={"otype": "CutlassMLP", "activation": "ReLU", "output_activation": rgb_act,
                                "n_neurons": 32, "n_hidden_layers": 1}, seed=1342)

        if self.rgb_act == 'None':  # rgb_net outputs log-radiance: one tonemapper per channel
            for i in range(3):
                setattr(self, f'tonemapper_net_{i}', tcnn.Network(
                    n_input_dims=1, n_output_dims=1,
                    network_config={"otype": "CutlassMLP", "activation": "ReLU", "output_activation": "Sigmoid",
                                    "n_neurons": 64, "n_hidden_layers": 1}, seed=1343 + i))

    # ------------------------------------------------------------------ density / normals
    def _density_head(self, feat):
        """feat (N,128) -> sigma (N) through autograd-visible layers (used when a caller wants
        gradients of density() alone)."""
        lin1, lin2 = self.xyz_net[0], self.xyz_net[2]
        a1, _ = _LinearAct.apply(feat, lin1.weight, lin1.bias, _SOFTPLUS)
        s, _ = _LinearAct.apply(a1, lin2.weight, lin2.bias, _SOFTPLUS)
        return s[:, 0]

    def density(self, x, return_feat=False, grad=True, grad_feat=True):
        """x (N,3) in [-scale, scale] -> sigmas (N) [, feat_rgb (N,128)]"""
        x = ((x - self.xyz_min) / (self.xyz_max - self.xyz_min)).contiguous()
        _wait_params(self, rgb_table=return_feat)
        if not (grad and torch.is_grad_enabled()):
            # inference (update_density_grid runs this on 1-2 M points): three launches, no graph
            with torch.no_grad():
                n = x.shape[0]
                lin1, lin2 = self.xyz_net[0], self.xyz_net[2]
                feat = torch.empty(n, 128, dtype=_f32, device=x.device)
                call("grid_fwd", self.xyz_encoder.desc, self.xyz_encoder.params, x, n, feat, 128)
                a1 = torch.empty(n, 128, dtype=_f32, device=x.device)
                sig = torch.empty(n, 1, dtype=_f32, device=x.device)
                call("mlp2_fwd", feat, 128, lin1.weight, 128, lin1.bias, _SOFTPLUS, lin2.weight, 128, lin2.bias,
                     _SOFTPLUS, n, 128, 128, 1, a1, 128, sig, 1)
                sigmas = sig[:, 0]
        else:
            sigmas = self._density_head(self.xyz_encoder(x))
        if return_feat:
            with torch.set_grad_enabled(grad_feat and torch.is_grad_enabled()):
                feat_rgb = self.rgb_encoder(x)
            return sigmas, feat_rgb
        return sigmas

    def _field(self, x, d, kwargs):
        """-> sigmas, rgbs (after rgb_net's own output activation), dsigma/dx, raw normal head, semantic logits"""
        embed_a = None
        if self.embed_a:
            embed_a = kwargs['embedding_a']
            if embed_a.size(0) < x.size(0):
                embed_a = torch.repeat_interleave(embed_a, int(x.size(0) / embed_a.size(0)), 0)
            embed_a = embed_a.contiguous()
        lin1, lin2 = self.xyz_net[0], self.xyz_net[2]
        return _FieldFn.apply(self, x.contiguous(), d.contiguous(), embed_a,
                              self.xyz_encoder.params, lin1.weight, lin1.bias, lin2.weight, lin2.bias,
                              self.rgb_encoder.params, self.rgb_net.params, self.norm_pred_header.params,
                              self.semantic_header.params)

    def grad(self, x):
        """-> sigmas (N), feat_rgb (N,128), d(sigma)/dx (N,3) (detached, see module docstring).
        Kept for API parity (networks.py:186-196); forward()/forward_test() use the fused node."""
        sigmas, _, grads, _, _ = self._field(x, torch.zeros_like(x), {'embedding_a': torch.zeros(
            x.shape[0], self.rgb_net.n_input_dims - 144, device=x.device)} if self.embed_a else {})
        span = self.xyz_max - self.xyz_min
        feat_rgb = self.rgb_encoder(((x - self.xyz_min) / span).contiguous())
        return sigmas, feat_rgb, grads / span

    # ------------------------------------------------------------------ full field
    def _span(self):
        sp = getattr(self, '_span_t', None)
        if sp is None or sp.device != self.xyz_min.device:
            sp = self._span_t = self.xyz_max - self.xyz_min
        return sp

    def _inv_span(self):
        inv = getattr(self, '_inv_span_t', None)
        if inv is None or inv.device != self.xyz_min.device:
            inv = self._inv_span_t = (1.0 / (self.xyz_max - self.xyz_min)).reshape(3).contiguous()
        return inv

    def _tone(self, rgbs, kwargs):
        if self.rgb_act == 'None':  # rgb_net outputs log-radiance
            if kwargs.get('output_radiance', False):
                rgbs = TruncExp.apply(rgbs)
            else:
                rgbs = self.log_radiance_to_rgb(rgbs, **kwargs)
        return rgbs

    def _forward_differentiable_normals(self, x, d, kwargs):
        """The reference's own formulation (networks.py:186-240): d(sigma)/dx by
        torch.autograd.grad(create_graph=True), so that normals_raw carries gradients back into the
        density table and MLP (H4, needed by --normal_ref).  The grid's double backward runs on
        ngp_grid_bwd_bwd_input; the 17 k-parameter density MLP uses torch ops here."""
        _wait_params(self)
        x = x.detach().clone().requires_grad_(True)
        xn = (x - self.xyz_min) / (self.xyz_max - self.xyz_min)
        with torch.enable_grad():
            h = self.xyz_net(self.xyz_encoder(xn))
            sigmas = self.sigma_act(h[:, 0])
            grads = torch.autograd.grad(sigmas, x, torch.ones_like(sigmas), create_graph=True)[0]
        feat_rgb = self.rgb_encoder(xn.detach())
        dn = F.normalize(d, p=2, dim=-1, eps=1e-6)
        cols = [self.dir_encoder((dn + 1) / 2), feat_rgb]
        if self.embed_a:
            embed_a = kwargs['embedding_a']
            if embed_a.size(0) < feat_rgb.size(0):
                embed_a = torch.repeat_interleave(embed_a, int(feat_rgb.size(0) / embed_a.size(0)), 0)
            cols.append(embed_a)
        rgbs = self.rgb_net(torch.cat(cols, 1))
        return sigmas, rgbs, grads, self.norm_pred_header(feat_rgb), self.semantic_header(feat_rgb)

    def forward(self, x, d, **kwargs):
        """x, d (N,3) -> sigmas (N), rgbs (N,3), normals_raw (N,3), normals_pred (N,3), semantic (N,C)"""
        if getattr(self, 'differentiable_normals', False) and torch.is_grad_enabled():
            sigmas, rgbs, grads, np_raw, sem_logits = self._forward_differentiable_normals(x, d, kwargs)
            normals_raw = -F.normalize(grads, p=2, dim=-1, eps=1e-6)
        else:
            sigmas, rgbs, grads, np_raw, sem_logits = self._field(x, d, kwargs)
            normals_raw = _NegNormalize.apply(grads, self._inv_span())
        normals_pred = _NegNormalize.apply(np_raw, None)
        semantic = self.semantic_act(sem_logits)
        return sigmas, self._tone(rgbs, kwargs), normals_raw, normals_pred, semantic

    def forward_test(self, x, d, **kwargs):
        """same as forward but returns (sigmas, rgbs, normals_pred, normals_raw, semantic) — the
        reference's test path swaps the two normals (networks.py:282) and detaches the heads."""
        sigmas, rgbs, grads, np_raw, sem_logits = self._field(x, d, kwargs)
        normals_raw = _NegNormalize.apply(grads, self._inv_span())
        normals_pred = _NegNormalize.apply(np_raw.detach(), None)
        semantic = self.semantic_act(sem_logits.detach())
        return sigmas, self._tone(rgbs, kwargs), normals_pred, normals_raw, semantic

    def forward_skybox(self, d):
        if not self.use_skybox:
            return None
        d = d / torch.norm(d, dim=1, keepdim=True)
        d = self.skybox_dir_encoder((d + 1) / 2)
        return self.skybox_rgb_net(d)

    def log_radiance_to_rgb(self, log_radiances, **kwargs):
        """HDR -> LDR with the per-channel tonemappers (models/networks_noCUDA.py:238-251; the
        reference's CUDA model calls this without defining it, networks.py:238)."""
        out = []
        for i in range(3):
            inp = log_radiances[:, i:i + 1]
            if 'exposure' in kwargs:
                inp = inp + torch.log(kwargs['exposure'])
            out.append(getattr(self, f'tonemapper_net_{i}')(inp))
        return torch.cat(out, 1)

    # ------------------------------------------------------------------ occupancy grid
    @torch.no_grad()
    def get_all_cells(self):
        indices = vren.morton3D(self.grid_coords).long()
        return [(indices, self.grid_coords)] * self.cascades

    @torch.no_grad()
    def sample_uniform_and_occupied_cells(self, M, density_threshold):
        cells = []
        for c in range(self.cascades):
            gen = getattr(self, 'grid_rng', None)  # shared-seed generator keeps DDP ranks' grids identical
            coords1 = torch.randint(self.grid_size, (M, 3), dtype=torch.int32, device=self.density_grid.device,
                                    generator=gen)
            indices1 = vren.morton3D(coords1).long()
            # M random occupied cells without a host round trip (the reference's nonzero() +
            # randint(len) syncs): rank r ~ U[0, n_occ) is mapped to the r-th occupied cell through
            # the running count of the occupancy mask.  With no occupied cell at all the reference
            # samples none; here those M draws fall back onto the uniform cells (same coverage).
            occ = self.density_grid[c] > density_threshold
            csum = torch.cumsum(occ, 0, dtype=torch.int32)
            n_occ = csum[-1]
            u = torch.rand(M, device=occ.device, generator=gen)
            rank = torch.clamp((u * n_occ).to(torch.int32), max=n_occ - 1) + 1
            pos = torch.searchsorted(csum, rank.clamp(min=1), right=False)
            indices2 = torch.where(n_occ > 0, pos.clamp(max=occ.numel() - 1), indices1)
            coords2 = vren.morton3D_invert(indices2.int())
            indices, coords = torch.cat([indices1, indices2]), torch.cat([coords1, coords2])
            if _SORT_GRID_SAMPLES:
                # Morton order: neighbouring points share hash-grid cells at the coarse levels, so the 1 M
                # point gather of density() runs out of L2 instead of HBM (which cell receives which jitter
                # draw changes, their distribution does not)
                indices, perm = torch.sort(indices)
                coords = coords[perm]
            cells += [(indices, coords)]
        return cells

    @torch.no_grad()
    def mark_invisible_cells(self, K, poses, img_wh, chunk=64 ** 3):
        """Cells that no camera sees, or that lie in front of a camera but closer than NEAR_DISTANCE,
        get density -1 and are never updated again; `count_grid` keeps the fraction of cameras that
        see each cell (networks.py:336-377).  Run once before training.
        K (3,3) intrinsics, poses (N,3,4) camera-to-world, img_wh (w, h)."""
        n_cams = poses.shape[0]
        w, h = img_wh
        rot_t = poses[:, :3, :3].transpose(1, 2)                       # world -> camera rotations
        # one 3x4 projection per camera: pixel-homogeneous coordinates = proj @ [x_world; 1]
        proj = K @ torch.cat([rot_t, -rot_t @ poses[:, :3, 3:]], 2)
        self.count_grid = torch.zeros_like(self.density_grid)
        for c, (indices, coords) in enumerate(self.get_all_cells()):
            s = min(2 ** (c - 1), self.scale)
            centres = (coords.to(_f32) / (self.grid_size - 1) * 2 - 1) * (s - s / self.grid_size)
            for lo in range(0, len(indices), chunk):
                sel = indices[lo:lo + chunk]
                uvd = torch.einsum('nij,mj->nim', proj[:, :, :3], centres[lo:lo + chunk]) + proj[:, :, 3:]
                depth = uvd[:, 2]
                u, v = uvd[:, 0] / depth, uvd[:, 1] / depth
                inside = (depth >= 0) & (u >= 0) & (u < w) & (v >= 0) & (v < h)
                seen = (inside & (depth >= NEAR_DISTANCE)).sum(0) / n_cams
                too_close = (inside & (depth < NEAR_DISTANCE)).any(0)
                self.count_grid[c, sel] = seen
                self.density_grid[c, sel] = torch.where((seen > 0) & ~too_close, 0.0, -1.0)

    @torch.no_grad()
    def _update_density_grid_sampled(self, density_threshold, decay):
        """The sampled branch of update_density_grid on the fused kernels (ngp_grid_sample_cells ->
        density() -> ngp_density_grid_scatter_max -> ngp_density_grid_ema_threshold -> ngp_packbits): per
        cascade M = G^3/4 uniform cells + M occupied ones, jittered, evaluated, max-combined into the grid; EMA,
        mean threshold and bit packing.  The draws are a counter-based hash of (seed, update number, cascade):
        identical on every data-parallel rank (the seed comes from `grid_rng` when the trainer set one)."""
        G = self.grid_size
        M = G ** 3 // 4
        dev = self.density_grid.device
        ws = getattr(self, '_grid_ws', None)
        if ws is None or ws[0].device != dev:
            n_ws = call_host("grid_sample_workspace", G, M)
            ws = self._grid_ws = (torch.empty(n_ws, dtype=torch.int32, device=dev),
                                  torch.empty(2 * M, dtype=torch.int32, device=dev),
                                  torch.empty(2 * M, 3, dtype=_f32, device=dev),
                                  torch.empty(1024, dtype=_f32, device=dev), torch.empty(2, dtype=_f32, device=dev))
            gen = getattr(self, 'grid_rng', None)
            self._grid_seed = int(gen.initial_seed() if gen is not None else torch.initial_seed()) & 0x7FFFFFFFFFFF
            self._grid_updates = 0
        work, indices, xyzs_w, partials, thr = ws
        density_grid_tmp = torch.zeros_like(self.density_grid)
        for c in range(self.cascades):
            s = min(2 ** (c - 1), self.scale)
            seed = self._grid_seed + 1000003 * self._grid_updates + 7919 * c
            call("grid_sample_cells", self.density_grid[c], G, float(density_threshold), M, seed, float(s), work, indices,
                 xyzs_w)
            sig = self.density(xyzs_w)
            call("density_grid_scatter_max", density_grid_tmp[c], indices, sig.contiguous(), 2 * M)
        self._grid_updates += 1
        call("density_grid_ema_threshold", self.density_grid, density_grid_tmp, self.density_grid.numel(), float(decay),
             float(density_threshold), partials, thr)
        call("packbits", self.density_grid.view(-1), self.density_bitfield.shape[0], 0.0, thr, self.density_bitfield)

    @torch.no_grad()
    def update_density_grid(self, density_threshold, warmup=False, decay=0.95, erode=False):
        if not warmup and not erode and _FUSED_GRID_UPDATE and self.density_grid.is_cuda:
            return self._update_density_grid_sampled(density_threshold, decay)
        density_grid_tmp = torch.zeros_like(self.density_grid)
        if warmup:
            cells = self.get_all_cells()
        else:
            cells = self.sample_uniform_and_occupied_cells(self.grid_size ** 3 // 4, density_threshold)
        for c in range(self.cascades):
            indices, coords = cells[c]
            s = min(2 ** (c - 1), self.scale)
            noise = torch.rand(coords.shape[0], 3, dtype=_f32, device=coords.device,
                               generator=getattr(self, 'grid_rng', None))
            xyzs_w = torch.empty(coords.shape[0], 3, dtype=_f32, device=coords.device)
            call("grid_cell_points", coords.contiguous(), noise, coords.shape[0], self.grid_size, float(s), xyzs_w)
            density_grid_tmp[c, indices] = self.density(xyzs_w)
        if erode:
            if not hasattr(self, 'count_grid'):
                raise RuntimeError("erode=True needs mark_invisible_cells() to have been called "
                                   "(the reference crashes here too: networks.py:399)")
            decay_t = torch.clamp(decay ** (1 / self.count_grid), 0.1, 0.95)
            self.density_grid = torch.where(self.density_grid < 0, self.density_grid,
                                            torch.maximum(self.density_grid * decay_t, density_grid_tmp))
        else:
            call("density_grid_ema", self.density_grid, density_grid_tmp, self.density_grid.numel(), float(decay))
        # threshold = min(mean of the positive cells, density_threshold), kept on the device
        pos = self.density_grid > 0
        mean_density = (self.density_grid * pos).sum() / pos.sum().clamp(min=1)
        thr = torch.clamp(mean_density, max=density_threshold).reshape(1)
        call("packbits", self.density_grid.view(-1), self.density_bitfield.shape[0], 0.0, thr, self.density_bitfield)

    def uniform_sample(self, resolution=128):
        half_grid_size = self.scale / resolution
        lin = torch.linspace(0, 1 - half_grid_size, resolution, device=self.xyz_min.device)
        samples = torch.stack(torch.meshgrid(lin, lin, lin, indexing='ij'), -1)
        dense_xyz = self.xyz_min * (1 - samples) + self.xyz_max * samples
        dense_xyz += half_grid_size * torch.rand_like(dense_xyz)
        return self.density(dense_xyz.view(-1, 3))
