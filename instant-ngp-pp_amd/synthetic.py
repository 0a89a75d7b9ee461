"""Synthetic stand-in for NeRF-Synthetic lego (no dataset is available offline): the
"S-lego-proxy" scene of SURVEY.md §8(d).

An analytic field in [-0.5,0.5]^3 — union of three boxes and a sphere shell, sigma = 60 inside,
rgb = 0.5+0.5*sin(8*pi*x) — seen by pinhole cameras with lego's intrinsics (800x800,
camera_angle_x = 0.6911, datasets/nerf.py:33-41) on the upper hemisphere at radius 1.5
(datasets/nerf.py:48-60 normalises poses to that radius), looking at the origin.  Rays follow
datasets/ray_utils.py:8-74: camera-space directions ((u-cx+.5)/fx, (v-cy+.5)/fy, 1), rotated by
c2w, not normalised.  Ground-truth colours come from dense quadrature of the analytic field.
"""
import math

import torch

SIGMA_IN = 60.0


def analytic_sigma(x):
    """x (...,3) world coordinates -> sigma (...)"""
    ax = x.abs()
    box1 = (ax[..., 0] < 0.35) & (ax[..., 1] < 0.10) & (ax[..., 2] < 0.10)
    box2 = (ax[..., 0] < 0.10) & (ax[..., 1] < 0.30) & ((x[..., 2] + 0.15).abs() < 0.08)
    box3 = ((x[..., 0] - 0.15).abs() < 0.08) & ((x[..., 1] + 0.1).abs() < 0.08) & (ax[..., 2] < 0.33)
    r = torch.linalg.norm(x - torch.tensor([-0.15, 0.12, 0.12], device=x.device), dim=-1)
    shell = (r < 0.2) & (r > 0.15)
    return torch.where(box1 | box2 | box3 | shell, SIGMA_IN, 0.0)


def analytic_rgb(x):
    return 0.5 + 0.5 * torch.sin(8 * math.pi * x)


class LegoProxy:
    def __init__(self, n_images=100, img_wh=(800, 800), device="cuda", seed=20220806, radius=1.5):
        self.device = torch.device(device)
        self.img_wh = img_wh
        w, h = img_wh
        fx = fy = 0.5 * w / math.tan(0.5 * 0.6911112070083618)
        self.K = torch.tensor([[fx, 0, w / 2], [0, fy, h / 2], [0, 0, 1]], dtype=torch.float32, device=self.device)
        g = torch.Generator().manual_seed(seed)
        # upper-hemisphere camera centres, look-at origin, OpenCV axes (x right, y down, z forward)
        phi = torch.rand(n_images, generator=g) * 2 * math.pi
        cos_t = torch.rand(n_images, generator=g) * 0.9 + 0.05
        sin_t = torch.sqrt(1 - cos_t ** 2)
        pos = torch.stack([sin_t * torch.cos(phi), sin_t * torch.sin(phi), cos_t], -1) * radius
        fwd = -pos / pos.norm(dim=-1, keepdim=True)
        up = torch.tensor([0.0, 0.0, 1.0]).expand_as(fwd)
        right = torch.cross(fwd, up, dim=-1)
        right = right / right.norm(dim=-1, keepdim=True)
        down = torch.cross(fwd, right, dim=-1)
        self.poses = torch.stack([right, down, fwd, pos], -1).to(self.device)  # (n,3,4) c2w
        v, u = torch.meshgrid(torch.arange(h, device=self.device, dtype=torch.float32),
                              torch.arange(w, device=self.device, dtype=torch.float32), indexing="ij")
        self.directions = torch.stack([(u - w / 2 + 0.5) / fx, (v - h / 2 + 0.5) / fy, torch.ones_like(u)],
                                      -1).reshape(-1, 3)

    def rays(self, img_idxs, pix_idxs):
        """get_rays (ray_utils.py:50-74) for (image, pixel) pairs"""
        c2w = self.poses[img_idxs]
        d = self.directions[pix_idxs]
        rays_d = (d[:, None, :] @ c2w[:, :, :3].transpose(1, 2))[:, 0]
        rays_o = c2w[:, :, 3]
        return rays_o.contiguous(), rays_d.contiguous()

    def sample_batch(self, batch_size, generator=None):
        """random images + random pixels, as BaseDataset.__getitem__ (datasets/base.py:22-31)"""
        n_img = self.poses.shape[0]
        w, h = self.img_wh
        img = torch.randint(n_img, (batch_size,), device=self.device, generator=generator)
        pix = torch.randint(w * h, (batch_size,), device=self.device, generator=generator)
        return img, pix

    @torch.no_grad()
    def ground_truth(self, rays_o, rays_d, n_quad=1024, white_bg=False):
        """dense quadrature of the analytic field inside the unit box -> rgb (N,3), opacity (N)"""
        inv = 1.0 / rays_d
        a, b = (-0.5 - rays_o) * inv, (0.5 - rays_o) * inv
        t1 = torch.minimum(a, b).amax(-1).clamp(min=0)
        t2 = torch.maximum(a, b).amin(-1)
        hit = t2 > t1
        t2 = torch.where(hit, t2, t1)
        out_rgb = torch.zeros(len(rays_o), 3, device=rays_o.device)
        out_op = torch.zeros(len(rays_o), device=rays_o.device)
        step = (t2 - t1) / n_quad
        chunk = 16384
        for s in range(0, len(rays_o), chunk):
            sl = slice(s, s + chunk)
            k = torch.arange(n_quad, device=rays_o.device, dtype=torch.float32) + 0.5
            t = t1[sl, None] + step[sl, None] * k[None, :]
            x = rays_o[sl, None, :] + rays_d[sl, None, :] * t[..., None]
            sig = analytic_sigma(x)
            dl = step[sl, None] * rays_d[sl].norm(dim=-1, keepdim=True)
            alpha = 1 - torch.exp(-sig * dl)
            T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1 - alpha], 1), 1)[:, :-1]
            w = alpha * T
            out_rgb[sl] = (w[..., None] * analytic_rgb(x)).sum(1)
            out_op[sl] = w.sum(1)
        if white_bg:
            out_rgb = out_rgb + (1 - out_op)[:, None]
        return out_rgb, out_op

    @torch.no_grad()
    def occupancy_from_analytic(self, model, supersample=2):
        """density_grid (K, G^3, morton order) from the analytic sigma: the steady-state occupancy
        the schedule converges to, used to put the micro-benchmark straight into that regime."""
        from . import vren
        G = model.grid_size
        coords = model.grid_coords
        idx = vren.morton3D(coords).long()
        grid = torch.zeros(model.cascades, G ** 3, device=coords.device)
        for c in range(model.cascades):
            s = min(2 ** (c - 1), model.scale)
            best = torch.zeros(G ** 3, device=coords.device)
            for ox in range(supersample):
                for oy in range(supersample):
                    for oz in range(supersample):
                        off = (torch.tensor([ox, oy, oz], device=coords.device, dtype=torch.float32) + 0.5) / supersample
                        x = ((coords.float() + off) / G * 2 - 1) * s
                        best = torch.maximum(best, analytic_sigma(x))
            grid[c, idx] = best
        return grid
