"""NGP field of the reference (models/networks.py:13-420) on the MI355X library.

Same constructor, buffers (`center, xyz_min, xyz_max, half_size, density_bitfield`; the trainer
adds `density_grid`, `grid_coords`), attributes (`cascades, scale, grid_size`), methods
(`density, grad, forward, forward_test, forward_skybox, get_all_cells,
sample_uniform_and_occupied_cells, mark_invisible_cells, update_density_grid`) and state-dict
keys (`xyz_encoder.params`, `xyz_net.0.weight`, `rgb_net.params`, ...), so reference
checkpoints map one to one.

Differences, all deliberate:
  * d(sigma)/dx is computed analytically in the forward pass (sigmoid-gated back-substitution
    through the 2-layer density MLP + the grid input-gradient kernel) instead of
    torch.autograd.grad(create_graph=True) (networks.py:186-196).  The values are identical; the
    result is detached, i.e. normals_raw carries no gradient (the double backward H4 is only
    needed with --normal_ref and is exposed through tinycudann.Encoding, not wired in here yet).
"""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd import Function

from . import tinycudann as tcnn
from . import vren
from ._lib import call
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
            call("act_bwd", dy.contiguous(), z if act == _SOFTPLUS else y, dz.numel(), act, dz)
        dx = dW = db = None
        if ctx.needs_input_grad[1]:
            dW = torch.zeros_like(W)
            db = torch.zeros(no, dtype=_f32, device=x.device) if ctx.has_bias else None
            call("linear_bwd_weight", dz, no, x, ni, n, ni, no, dW, ni, db)
        if ctx.needs_input_grad[0]:
            dx = torch.empty(n, ni, dtype=_f32, device=x.device)
            call("linear_bwd_input", dz, no, W, ni, n, ni, no, dx, ni)
        return dx, dW, db, None


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
        """feat (N,128) -> sigma (N), plus the pre-activations of both layers."""
        lin1, lin2 = self.xyz_net[0], self.xyz_net[2]
        a1, z1 = _LinearAct.apply(feat, lin1.weight, lin1.bias, _SOFTPLUS)
        s, h = _LinearAct.apply(a1, lin2.weight, lin2.bias, _SOFTPLUS)
        return s[:, 0], z1, h

    def density(self, x, return_feat=False, grad=True, grad_feat=True):
        """x (N,3) in [-scale, scale] -> sigmas (N) [, feat_rgb (N,128)]"""
        x = (x - self.xyz_min) / (self.xyz_max - self.xyz_min)
        with torch.set_grad_enabled(grad and torch.is_grad_enabled()):
            h = self.xyz_encoder(x)
            sigmas, _, _ = self._density_head(h)
        if return_feat:
            with torch.set_grad_enabled(grad_feat and torch.is_grad_enabled()):
                feat_rgb = self.rgb_encoder(x)
            return sigmas, feat_rgb
        return sigmas

    def grad(self, x):
        """-> sigmas (N), feat_rgb (N,128), d(sigma)/dx (N,3) (detached, see module docstring)."""
        span = self.xyz_max - self.xyz_min
        xn = ((x - self.xyz_min) / span).contiguous()
        feat = self.xyz_encoder(xn)
        sigmas, z1, h = self._density_head(feat)
        feat_rgb = self.rgb_encoder(xn)
        with torch.no_grad():
            n = xn.shape[0]
            lin1, lin2 = self.xyz_net[0], self.xyz_net[2]
            ones = torch.ones(n, 1, dtype=_f32, device=x.device)
            dh = torch.empty_like(ones)
            call("act_bwd", ones, h, n, _SOFTPLUS, dh)                      # d sigma / d h
            da1 = torch.empty(n, 128, dtype=_f32, device=x.device)
            call("linear_bwd_input", dh, 1, lin2.weight, 128, n, 128, 1, da1, 128)
            dz1 = torch.empty_like(da1)
            call("act_bwd", da1, z1, da1.numel(), _SOFTPLUS, dz1)
            dfeat = torch.empty(n, feat.shape[1], dtype=_f32, device=x.device)
            call("linear_bwd_input", dz1, 128, lin1.weight, feat.shape[1], n, feat.shape[1], 128, dfeat,
                 feat.shape[1])
            grads = torch.empty(n, 3, dtype=_f32, device=x.device)
            call("grid_bwd_input", self.xyz_encoder.desc, self.xyz_encoder.params, xn, dfeat, n, grads)
            grads = grads / span
        return sigmas, feat_rgb, grads

    # ------------------------------------------------------------------ full field
    def _color(self, d, feat_rgb, kwargs):
        d = F.normalize(d, p=2, dim=-1, eps=1e-6)
        d = self.dir_encoder((d + 1) / 2)
        if self.embed_a:
            embed_a = kwargs['embedding_a']
            if embed_a.size(0) < feat_rgb.size(0):
                repeat = int(feat_rgb.size(0) / embed_a.size(0))
                embed_a = torch.repeat_interleave(embed_a, repeat, 0)
            rgbs = self.rgb_net(torch.cat([d, feat_rgb, embed_a], 1))
        else:
            rgbs = self.rgb_net(torch.cat([d, feat_rgb], 1))
        if self.rgb_act == 'None':  # log-radiance
            if kwargs.get('output_radiance', False):
                rgbs = TruncExp.apply(rgbs)
            else:
                rgbs = self.log_radiance_to_rgb(rgbs, **kwargs)
        return rgbs

    def forward(self, x, d, **kwargs):
        """x, d (N,3) -> sigmas (N), rgbs (N,3), normals_raw (N,3), normals_pred (N,3), semantic (N,C)"""
        sigmas, feat_rgb, grads = self.grad(x)
        normals_raw = -F.normalize(grads, p=2, dim=-1, eps=1e-6)
        normals_pred = -F.normalize(self.norm_pred_header(feat_rgb), p=2, dim=-1, eps=1e-6)
        semantic = self.semantic_act(self.semantic_header(feat_rgb))
        rgbs = self._color(d, feat_rgb, kwargs)
        return sigmas, rgbs, normals_raw, normals_pred, semantic

    def forward_test(self, x, d, **kwargs):
        """same as forward but returns (sigmas, rgbs, normals_pred, normals_raw, semantic) — the
        reference's test path swaps the two normals (networks.py:282)."""
        sigmas, feat_rgb, grads = self.grad(x)
        normals_raw = -F.normalize(grads, p=2, dim=-1, eps=1e-6)
        with torch.no_grad():
            normals_pred = -F.normalize(self.norm_pred_header(feat_rgb), p=2, dim=-1, eps=1e-6)
            semantic = self.semantic_act(self.semantic_header(feat_rgb))
        rgbs = self._color(d, feat_rgb, kwargs)
        return sigmas, rgbs, normals_pred, normals_raw, semantic

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
            indices2 = torch.nonzero(self.density_grid[c] > density_threshold)[:, 0]
            if len(indices2) > 0:
                rand_idx = torch.randint(len(indices2), (M,), device=self.density_grid.device, generator=gen)
                indices2 = indices2[rand_idx]
            coords2 = vren.morton3D_invert(indices2.int())
            cells += [(torch.cat([indices1, indices2]), torch.cat([coords1, coords2]))]
        return cells

    @torch.no_grad()
    def mark_invisible_cells(self, K, poses, img_wh, chunk=64 ** 3):
        """cells no camera sees get density -1 (never updated); run once before training."""
        N_cams = poses.shape[0]
        self.count_grid = torch.zeros_like(self.density_grid)
        w2c_R = poses[:, :3, :3].transpose(1, 2)
        w2c_T = -w2c_R @ poses[:, :3, 3:]
        cells = self.get_all_cells()
        for c in range(self.cascades):
            indices, coords = cells[c]
            for i in range(0, len(indices), chunk):
                xyzs = coords[i:i + chunk] / (self.grid_size - 1) * 2 - 1
                s = min(2 ** (c - 1), self.scale)
                half_grid_size = s / self.grid_size
                xyzs_w = (xyzs * (s - half_grid_size)).T
                xyzs_c = w2c_R @ xyzs_w + w2c_T
                uvd = K @ xyzs_c
                uv = uvd[:, :2] / uvd[:, 2:]
                in_image = (uvd[:, 2] >= 0) & (uv[:, 0] >= 0) & (uv[:, 0] < img_wh[0]) & \
                           (uv[:, 1] >= 0) & (uv[:, 1] < img_wh[1])
                covered_by_cam = (uvd[:, 2] >= NEAR_DISTANCE) & in_image
                self.count_grid[c, indices[i:i + chunk]] = count = covered_by_cam.sum(0) / N_cams
                too_near_to_any_cam = ((uvd[:, 2] < NEAR_DISTANCE) & in_image).any(0)
                valid_mask = (count > 0) & (~too_near_to_any_cam)
                self.density_grid[c, indices[i:i + chunk]] = torch.where(valid_mask, 0., -1.)

    @torch.no_grad()
    def update_density_grid(self, density_threshold, warmup=False, decay=0.95, erode=False):
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
        mean_density = self.density_grid[self.density_grid > 0].mean().item()
        vren.packbits(self.density_grid.view(-1), min(mean_density, density_threshold), self.density_bitfield)

    def uniform_sample(self, resolution=128):
        half_grid_size = self.scale / resolution
        lin = torch.linspace(0, 1 - half_grid_size, resolution, device=self.xyz_min.device)
        samples = torch.stack(torch.meshgrid(lin, lin, lin, indexing='ij'), -1)
        dense_xyz = self.xyz_min * (1 - samples) + self.xyz_max * samples
        dense_xyz += half_grid_size * torch.rand_like(dense_xyz)
        return self.density(dense_xyz.view(-1, 3))
