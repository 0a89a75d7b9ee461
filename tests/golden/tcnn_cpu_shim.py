"""A pure-torch, CPU, twice-differentiable stand-in for the `tinycudann` objects the reference's
models/networks.py instantiates — used ONLY by make_golden.py to run the reference's own NGP class
(its forward / forward_test / density / grad wiring) in the build container and record what it
returns.  It follows the tiny-cuda-nn semantics restated in SURVEY.md Appendix B, i.e. the same
rules as oracle/ngp_oracle.c (make_golden.py asserts that this shim and the C oracle agree on the
encoder before it records anything), so the fixture pins the reference's PYTHON wiring: what is
normalised how, which tensor feeds which network in which column order, which activation follows.
"""
import torch
from torch import nn


def _grid_levels(n_levels, n_features, log2_T, base, per_level_scale):
    """level geometry from the C oracle (ngp_cpu_grid_layout: scale_l = exp2f(l*log2f(b))*base - 1 in
    fp32 with the C library's exp2f/log2f, res = ceil(scale)+1, rows = min(align8(res^3), 2^T)); numpy's
    float32 exp2 differs from glibc's by an ulp at some levels, so it is not recomputed here"""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if root not in sys.path:
        sys.path.insert(0, root)
    import oracle
    desc, n_params = oracle.grid_layout(n_levels, n_features, log2_T, base, float(per_level_scale))
    out = [(float(desc.scale[l]), int(desc.resolution[l]), int(desc.offsets[l]), int(desc.offsets[l + 1] - desc.offsets[l]))
           for l in range(n_levels)]
    return out, int(desc.offsets[n_levels])


class _Grid(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.L, self.F = cfg["n_levels"], cfg["n_features_per_level"]
        self.levels, rows = _grid_levels(self.L, self.F, cfg["log2_hashmap_size"], cfg["base_resolution"],
                                         cfg["per_level_scale"])
        self.n_output_dims = self.L * self.F
        self.params = nn.Parameter(torch.empty(rows * self.F).uniform_(-1e-4, 1e-4))

    def forward(self, x):
        table = self.params.view(-1, self.F)
        outs = []
        for sc, res, off, size in self.levels:
            pos = (x.double() * sc + 0.5).to(x.dtype)   # one rounding, as tcnn's / the oracle's fmaf(scale, x, 0.5)
            g0 = torch.floor(pos).detach()
            w = pos - g0
            g0 = g0.to(torch.int64)
            dense = res ** 3 <= size     # oracle grid_row: the running stride x, x*res, x*res^2 fits the level
            acc = 0
            for c in range(8):
                cx, cy, cz = c & 1, (c >> 1) & 1, (c >> 2) & 1
                gx, gy, gz = g0[:, 0] + cx, g0[:, 1] + cy, g0[:, 2] + cz
                if dense:
                    idx = gx + gy * res + gz * res * res
                else:
                    idx = (gx & 0xFFFFFFFF) ^ ((gy * 2654435761) & 0xFFFFFFFF) ^ ((gz * 805459861) & 0xFFFFFFFF)
                idx = off + idx % size
                wt = (w[:, 0] if cx else 1 - w[:, 0]) * (w[:, 1] if cy else 1 - w[:, 1]) * (w[:, 2] if cz else 1 - w[:, 2])
                acc = acc + wt[:, None] * table[idx]
            outs.append(acc)
        return torch.cat(outs, 1)


class _SH(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.degree = cfg["degree"]
        self.n_output_dims = self.degree ** 2
        self.params = nn.Parameter(torch.zeros(0))

    def forward(self, x01):
        v = x01 * 2 - 1
        x, y, z = v[:, 0], v[:, 1], v[:, 2]
        xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
        o = [torch.full_like(x, 0.28209479177387814)]
        if self.degree > 1:
            o += [-0.48860251190291987 * y, 0.48860251190291987 * z, -0.48860251190291987 * x]
        if self.degree > 2:
            o += [1.0925484305920792 * xy, -1.0925484305920792 * yz, 0.94617469575755997 * z2 - 0.31539156525251999,
                  -1.0925484305920792 * xz, 0.54627421529603959 * x2 - 0.54627421529603959 * y2]
        if self.degree > 3:
            o += [0.59004358992664352 * y * (-3.0 * x2 + y2), 2.8906114426405538 * xy * z,
                  0.45704579946446572 * y * (1.0 - 5.0 * z2), 0.3731763325901154 * z * (5.0 * z2 - 3.0),
                  0.45704579946446572 * x * (1.0 - 5.0 * z2), 1.4453057213202769 * z * (x2 - y2),
                  0.59004358992664352 * x * (-x2 + 3.0 * y2)]
        return torch.stack(o, 1)


def Encoding(n_input_dims, encoding_config, **_):
    ot = encoding_config["otype"]
    if ot in ("Grid", "HashGrid"):
        return _Grid(encoding_config)
    if ot == "SphericalHarmonics":
        return _SH(encoding_config)
    raise NotImplementedError(ot)


_ACT = {"None": lambda v: v, "ReLU": torch.relu, "Sigmoid": torch.sigmoid}


class Network(nn.Module):
    """CutlassMLP: bias-free, input padded to a multiple of 16 with ones, output padded to 16 and
    sliced, weights (out, in) row-major, concatenated in one flat `params`"""

    def __init__(self, n_input_dims, n_output_dims, network_config, **_):
        super().__init__()
        self.n_input_dims, self.n_output_dims = n_input_dims, n_output_dims
        self.width, self.n_hidden = network_config["n_neurons"], network_config["n_hidden_layers"]
        self.act, self.out_act = network_config["activation"], network_config["output_activation"]
        self.padded_in = (n_input_dims + 15) // 16 * 16
        self.padded_out = (n_output_dims + 15) // 16 * 16
        dims = [self.padded_in] + [self.width] * self.n_hidden + [self.padded_out]
        self.shapes = [(dims[i + 1], dims[i]) for i in range(len(dims) - 1)]
        n = sum(a * b for a, b in self.shapes)
        self.params = nn.Parameter(torch.empty(n).uniform_(-0.1, 0.1))

    def forward(self, x):
        if x.shape[1] < self.padded_in:
            x = torch.cat([x, torch.ones(x.shape[0], self.padded_in - x.shape[1], dtype=x.dtype)], 1)
        off = 0
        for i, (o, k) in enumerate(self.shapes):
            W = self.params[off:off + o * k].view(o, k)
            off += o * k
            x = x @ W.T
            x = _ACT[self.act](x) if i < len(self.shapes) - 1 else _ACT[self.out_act](x)
        return x[:, :self.n_output_dims]
