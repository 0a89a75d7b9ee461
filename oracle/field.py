"""CPU restatement of the NGP field (models/networks.py:165-240) on the oracle's C kernels —
TEST INFRASTRUCTURE ONLY (parity checks of the GPU field and the timed cpu_baseline leg).

The hash grid / SH / MLP semantics are tiny-cuda-nn's (parity with the real
tcnn is unpinned, see ngp_oracle.c).  The WIRING of the field is pinned: tests/test_oracle_golden.py
checks this class against the outputs of the reference's own models/networks.py::NGP (fixture G6).  Weights are taken from a state dict with the reference's
key names (`xyz_encoder.params`, `xyz_net.0.weight`, ..., `rgb_net.params`).
"""
import numpy as np

import oracle

F32 = np.float32


def _softplus(v):
    return np.where(v > 20, v, np.log1p(np.exp(np.minimum(v, 20)))).astype(F32)


def _sigmoid(v):
    return (1.0 / (1.0 + np.exp(-v))).astype(F32)


def _normalize(v, eps=1e-6):
    n = np.maximum(np.linalg.norm(v, axis=-1, keepdims=True), eps)
    return (v / n).astype(F32)


class CpuNGP:
    def __init__(self, state, scale, classes=7, L=16, F=8, log2_T_xyz=19, log2_T_rgb=21, n_min=16):
        s = {k: np.asarray(v, F32) for k, v in state.items()}
        self.scale = scale
        self.classes = classes
        self.center = np.zeros((1, 3), F32)
        self.half_size = np.full((1, 3), scale, F32)
        b = float(np.exp(np.log(2048 * scale / n_min) / (L - 1)))
        self.xyz_desc, n_xyz = oracle.grid_layout(L, F, log2_T_xyz, n_min, b)
        self.rgb_desc, n_rgb = oracle.grid_layout(L, F, log2_T_rgb, n_min, b)
        self.xyz_table = s["xyz_encoder.params"]
        self.rgb_table = s["rgb_encoder.params"]
        assert self.xyz_table.size == n_xyz and self.rgb_table.size == n_rgb
        self.W1, self.b1 = s["xyz_net.0.weight"], s["xyz_net.0.bias"]
        self.W2, self.b2 = s["xyz_net.2.weight"], s["xyz_net.2.bias"]
        rp = s["rgb_net.params"]
        n_in = (rp.size - 16 * 128) // 128
        self.Wr1, self.Wr2 = rp[:128 * n_in].reshape(128, n_in), rp[128 * n_in:].reshape(16, 128)
        npar = s["norm_pred_header.params"]
        self.Wn1, self.Wn2 = npar[:32 * 128].reshape(32, 128), npar[32 * 128:].reshape(16, 32)
        sp = s["semantic_header.params"]
        self.Ws1, self.Ws2 = sp[:32 * 128].reshape(32, 128), sp[32 * 128:].reshape(16, 32)

    def density(self, x, with_grad=False):
        xn = ((x - (-self.scale)) / (2 * self.scale)).astype(F32)
        feat = oracle.grid_fwd(self.xyz_desc, self.xyz_table, xn)
        z1 = oracle.linear_fwd(feat, self.W1, self.b1, "None")
        a1 = _softplus(z1)
        h = oracle.linear_fwd(a1, self.W2, self.b2, "None")
        sigma = _softplus(h)[:, 0]
        if not with_grad:
            return sigma, xn
        dh = _sigmoid(h)                                   # d softplus
        dz1 = (dh @ self.W2) * _sigmoid(z1)                # (N,128)
        dfeat = (dz1 @ self.W1).astype(F32)                # (N,128)
        grads = oracle.grid_bwd_input(self.xyz_desc, self.xyz_table, xn, dfeat) / (2 * self.scale)
        return sigma, xn, grads.astype(F32)

    def __call__(self, x, d, embed_a=None):
        """-> (sigmas, rgbs, normals_raw, normals_pred, sems, None) like NGP.forward (+ the 6th slot
        rendering_noCUDA expects)."""
        x = np.ascontiguousarray(x, F32)
        sigma, xn, grads = self.density(x, with_grad=True)
        feat_rgb = oracle.grid_fwd(self.rgb_desc, self.rgb_table, xn)
        normals_raw = -_normalize(grads)
        hn = oracle.linear_fwd(feat_rgb, self.Wn1, None, "ReLU")
        normals_pred = -_normalize(oracle.linear_fwd(hn, self.Wn2, None, "None")[:, :3])
        hs = oracle.linear_fwd(feat_rgb, self.Ws1, None, "ReLU")
        logits = oracle.linear_fwd(hs, self.Ws2, None, "None")[:, :self.classes]
        e = np.exp(logits - logits.max(-1, keepdims=True))
        sems = (e / e.sum(-1, keepdims=True)).astype(F32)
        dn = _normalize(np.asarray(d, F32))
        sh = oracle.sh_fwd((dn + 1) / 2, 4)
        inp = np.concatenate([sh, feat_rgb] + ([np.asarray(embed_a, F32)] if embed_a is not None and self.Wr1.shape[1] > 144 else []), 1)
        if inp.shape[1] < self.Wr1.shape[1]:
            # tcnn pads the network input to a multiple of 16 with ones:
            # 16 + 128 + 8 appearance dims = 152 -> 160
            inp = np.concatenate([inp, np.ones((inp.shape[0], self.Wr1.shape[1] - inp.shape[1]), F32)], 1)
        hr = oracle.linear_fwd(inp, self.Wr1, None, "ReLU")
        rgbs = oracle.linear_fwd(hr, self.Wr2, None, "Sigmoid")[:, :3]
        return sigma, rgbs, normals_raw, normals_pred, sems, None
