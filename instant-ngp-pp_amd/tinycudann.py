"""`tinycudann`-shaped module (the subset the reference uses: models/networks.py:40-163,
models/implicit_mask.py:15-27, models/networks_noCUDA.py:17-30) on libngp_hip.so.

  Encoding(n_input_dims, encoding_config)            otype Grid/HashGrid, SphericalHarmonics, Frequency, Identity
  Network(n_input_dims, n_output_dims, network_config)   otype CutlassMLP / FullyFusedMLP
  NetworkWithInputEncoding(n_input_dims, n_output_dims, encoding_config, network_config)

Each is an nn.Module with `n_output_dims`, one flat fp32 `params` nn.Parameter and
`forward(x (N, n_in)) -> (N, n_out)` fp32 (the reference builds tcnn with
TCNN_HALF_PRECISION=0, README.md:24).  Semantics follow SURVEY.md Appendix B (tcnn's source is
not part of the reference tree: parity with the real tiny-cuda-nn is unpinned).
"""
import math

import torch
from torch import nn
from torch.autograd import Function

from ._lib import GridDesc, call, call_host

_ACT = {"None": 0, "ReLU": 1, "Sigmoid": 2, "Softplus": 3, "Exponential": 4}
_f32 = torch.float32


def _pad16(n):
    return (n + 15) // 16 * 16


# ------------------------------------------------------------------------------ hash grid
class _GridFwd(Function):
    @staticmethod
    def forward(ctx, x, params, enc):
        x = x.contiguous()
        y = torch.empty(x.shape[0], enc.n_output_dims, dtype=_f32, device=x.device)
        call("grid_fwd", enc.desc, params, x, x.shape[0], y, enc.n_output_dims)
        ctx.save_for_backward(x, params)
        ctx.enc = enc
        return y

    @staticmethod
    def backward(ctx, dy):
        x, params = ctx.saved_tensors
        enc = ctx.enc
        dy = dy.contiguous()
        dx = dparams = None
        if ctx.needs_input_grad[0]:
            dx = _GridBwdInput.apply(dy, x, params, enc)
        if ctx.needs_input_grad[1]:
            buf = getattr(enc, "grad_buffer", None)
            if buf is not None:
                # trainer-owned flat gradient: scatter-add straight into it (autograd sees None)
                call("grid_bwd_param", enc.desc, x, dy, enc.n_output_dims, x.shape[0], buf)
                enc._bound_valid = False   # a table-gradient contribution the trainer's norm bound does not cover
            else:
                dparams = torch.zeros_like(params)
                call("grid_bwd_param", enc.desc, x, dy, enc.n_output_dims, x.shape[0], dparams)
            cb = getattr(enc, "on_grad_ready", None)
            if cb is not None:
                cb()
        return dx, dparams, None


class _GridBwdInput(Function):
    """dL_dx = J(x, table)^T dL_dy; differentiable w.r.t. dL_dy and the table (H4)."""

    @staticmethod
    def forward(ctx, dy, x, params, enc):
        dx = torch.empty(x.shape[0], 3, dtype=_f32, device=x.device)
        call("grid_bwd_input", enc.desc, params, x, dy, enc.n_output_dims, x.shape[0], dx)
        ctx.save_for_backward(dy, x, params)
        ctx.enc = enc
        return dx

    @staticmethod
    def backward(ctx, ddx):
        dy, x, params = ctx.saved_tensors
        enc = ctx.enc
        ddx = ddx.contiguous()
        d_dy = torch.empty_like(dy) if ctx.needs_input_grad[0] else None
        d_params = torch.zeros_like(params) if ctx.needs_input_grad[2] else None
        if d_params is not None:
            enc._bound_valid = False       # the double backward adds to the table gradient outside the norm bound
        call("grid_bwd_bwd_input", enc.desc, params, x, dy, enc.n_output_dims, ddx, x.shape[0], d_params, d_dy)
        return d_dy, None, d_params, None


class _SHFwd(Function):
    @staticmethod
    def forward(ctx, x, degree):
        x = x.contiguous()
        y = torch.empty(x.shape[0], degree * degree, dtype=_f32, device=x.device)
        call("sh_fwd", x, x.shape[0], degree, y, degree * degree)
        ctx.save_for_backward(x)
        ctx.degree = degree
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            call("sh_bwd_input", x, dy.contiguous(), x.shape[0], ctx.degree, dx)
        return dx, None


class Encoding(nn.Module):
    def __init__(self, n_input_dims, encoding_config, dtype=None, seed=1337):
        super().__init__()
        self.n_input_dims = n_input_dims
        self.encoding_config = dict(encoding_config)
        ot = encoding_config["otype"]
        self.otype = ot
        if ot in ("Grid", "HashGrid", "DenseGrid"):
            if n_input_dims != 3:
                raise ValueError("grid encoding: only 3-D inputs are supported")
            cfg = encoding_config
            if cfg.get("interpolation", "Linear") != "Linear":
                raise ValueError("grid encoding: only Linear interpolation is supported")
            if ot == "Grid" and cfg.get("type", "Hash") != "Hash":
                raise ValueError("grid encoding: only type Hash is supported")
            self.n_levels = int(cfg.get("n_levels", 16))
            self.n_features = int(cfg.get("n_features_per_level", 2))
            self.desc = GridDesc()
            n = call_host("grid_layout", self.n_levels, self.n_features, int(cfg.get("log2_hashmap_size", 19)),
                          int(cfg.get("base_resolution", 16)), float(cfg.get("per_level_scale", 2.0)), self.desc)
            if n <= 0:
                raise ValueError(f"unsupported grid configuration {cfg}")
            self.n_output_dims = self.n_levels * self.n_features
            g = torch.Generator().manual_seed(seed)
            self.params = nn.Parameter((torch.rand(int(n), generator=g, dtype=_f32) * 2 - 1) * 1e-4)
        elif ot == "SphericalHarmonics":
            self.degree = int(encoding_config.get("degree", 4))
            if not 1 <= self.degree <= 4:
                raise ValueError("SphericalHarmonics: degree must be in 1..4")
            self.n_output_dims = self.degree ** 2
            self.params = nn.Parameter(torch.zeros(0, dtype=_f32))
        elif ot == "Frequency":
            self.n_frequencies = int(encoding_config.get("n_frequencies", 12))
            self.n_output_dims = n_input_dims * self.n_frequencies * 2
            self.params = nn.Parameter(torch.zeros(0, dtype=_f32))
        elif ot == "Identity":
            self.n_output_dims = n_input_dims
            self.params = nn.Parameter(torch.zeros(0, dtype=_f32))
        else:
            raise ValueError(f"unsupported encoding otype {ot}")

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("x must be a CUDA tensor")
        x = x.float()
        if self.otype in ("Grid", "HashGrid", "DenseGrid"):
            return _GridFwd.apply(x, self.params, self)
        if self.otype == "SphericalHarmonics":
            return _SHFwd.apply(x, self.degree)
        if self.otype == "Frequency":
            # tcnn: for each input dim, for each frequency k: sin(2^k pi x), cos(2^k pi x)
            k = torch.arange(self.n_frequencies, device=x.device, dtype=_f32)
            ang = x[:, :, None] * (2.0 ** k)[None, None, :] * math.pi
            return torch.stack([torch.sin(ang), torch.cos(ang)], -1).reshape(x.shape[0], -1)
        return x


# ------------------------------------------------------------------------------ MLP
class _MLPFn(Function):
    """Bias-free MLP (tcnn CutlassMLP layout): params = concat of row-major (n_out_l, n_in_l)."""

    @staticmethod
    def forward(ctx, x, params, net):
        x = x.contiguous()
        n = x.shape[0]
        acts = [x]
        off = 0
        h = x
        for li, (no, ni) in enumerate(net.layer_shapes):
            W = params[off:off + no * ni]
            off += no * ni
            act = net.activation if li < len(net.layer_shapes) - 1 else net.output_activation
            y = torch.empty(n, no, dtype=_f32, device=x.device)
            call("linear_fwd", h, h.shape[1], W, ni, None, n, ni, no, act, y, no, None)
            acts.append(y)
            h = y
        ctx.net = net
        ctx.save_for_backward(params, *acts)
        return h

    @staticmethod
    def backward(ctx, dy):
        net = ctx.net
        params, *acts = ctx.saved_tensors
        n = acts[0].shape[0]
        need_x, need_p = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dparams = torch.zeros_like(params) if need_p else None
        shapes = net.layer_shapes
        offs = [0]
        for (no, ni) in shapes:
            offs.append(offs[-1] + no * ni)
        g = dy.contiguous()
        dx = None
        for li in range(len(shapes) - 1, -1, -1):
            no, ni = shapes[li]
            act = net.activation if li < len(shapes) - 1 else net.output_activation
            if act == _ACT["Softplus"]:
                raise NotImplementedError("Softplus inside tinycudann.Network is not supported")
            dz = g
            if act != 0:
                dz = torch.empty_like(g)
                call("act_bwd", g, acts[li + 1], g.numel(), act, dz)
            W = params[offs[li]:offs[li + 1]]
            if need_p:
                call("linear_bwd_weight", dz, no, acts[li], acts[li].shape[1], n, ni, no,
                     dparams[offs[li]:offs[li + 1]], ni, None)
            if li > 0 or need_x:
                g = torch.empty(n, ni, dtype=_f32, device=dy.device)
                call("linear_bwd_input", dz, no, W, ni, n, ni, no, g, ni, 0)
                if li == 0:
                    dx = g
        return dx, dparams, None


class Network(nn.Module):
    def __init__(self, n_input_dims, n_output_dims, network_config, seed=1337):
        super().__init__()
        ot = network_config.get("otype", "CutlassMLP")
        if ot not in ("CutlassMLP", "FullyFusedMLP"):
            raise ValueError(f"unsupported network otype {ot}")
        self.n_input_dims = n_input_dims
        self.n_output_dims = n_output_dims
        self.network_config = dict(network_config)
        self.activation = _ACT[network_config.get("activation", "ReLU")]
        self.output_activation = _ACT[network_config.get("output_activation", "None")]
        width = int(network_config.get("n_neurons", 128))
        n_hidden = int(network_config.get("n_hidden_layers", 5))
        self.padded_in = _pad16(n_input_dims)
        self.padded_out = _pad16(n_output_dims)
        dims = [self.padded_in] + [width] * n_hidden + [self.padded_out]
        self.layer_shapes = [(dims[i + 1], dims[i]) for i in range(len(dims) - 1)]
        g = torch.Generator().manual_seed(seed)
        chunks = []
        for (no, ni) in self.layer_shapes:  # tcnn: xavier uniform
            bound = math.sqrt(6.0 / (ni + no))
            chunks.append((torch.rand(no * ni, generator=g, dtype=_f32) * 2 - 1) * bound)
        self.params = nn.Parameter(torch.cat(chunks))

    def layer_weight(self, i):
        off = sum(a * b for a, b in self.layer_shapes[:i])
        no, ni = self.layer_shapes[i]
        return self.params[off:off + no * ni].view(no, ni)

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("x must be a CUDA tensor")
        x = x.float()
        if x.shape[1] != self.padded_in:  # tcnn pads the input width to a multiple of 16 with ones
            pad = torch.ones(x.shape[0], self.padded_in - x.shape[1], dtype=_f32, device=x.device)
            x = torch.cat([x, pad], 1)
        y = _MLPFn.apply(x, self.params, self)
        return y[:, :self.n_output_dims]


class NetworkWithInputEncoding(nn.Module):
    def __init__(self, n_input_dims, n_output_dims, encoding_config, network_config, seed=1337):
        super().__init__()
        self.encoding = Encoding(n_input_dims, encoding_config, seed=seed)
        self.network = Network(self.encoding.n_output_dims, n_output_dims, network_config, seed=seed + 1)
        self.n_input_dims = n_input_dims
        self.n_output_dims = n_output_dims
        # tcnn exposes one flat vector: network parameters first, then the encoding's
        self._n_net = self.network.params.numel()
        flat = torch.cat([self.network.params.data, self.encoding.params.data])
        del self.network.params
        del self.encoding.params
        self.params = nn.Parameter(flat)

    def forward(self, x):
        self.network.params = self.params[:self._n_net]
        self.encoding.params = self.params[self._n_net:]
        try:
            return self.network(self.encoding(x))
        finally:
            del self.network.params
            del self.encoding.params
