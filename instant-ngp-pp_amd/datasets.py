"""NeRF-Synthetic (Blender) data on the ray side of the hot path: the loader of the reference
(datasets/nerf.py:13-71, datasets/base.py:5-66, datasets/ray_utils.py:8-74,
datasets/color_utils.py:19-28) with the same attribute names and sample dictionaries, plus a writer
that exports the analytic lego-proxy scene in that on-disk format (no dataset ships with this
repository or exists on the GPU box).

On-disk format (what `NeRFDataset` reads):
  <root>/transforms_{train,val,test}.json : {"camera_angle_x": float,
        "frames": [{"file_path": "./train/r_0", "transform_matrix": 4x4 camera-to-world in Blender axes
                    [right, up, back]}, ...]}
  <root>/<file_path>.png                  : RGBA (blended on white) or RGB, 8 bit

Deviations from the reference, all deliberate:
  * images are decoded with PIL (imageio / cv2 are not available); when the stored size differs from
    int(800*downsample) the resize is PIL bilinear, which is not bit-identical to cv2.resize —
    parity unpinned for downsample != stored size;
  * everything can live on the GPU (`device=`): ray sampling then costs two randint launches instead
    of a DataLoader worker round trip.
"""
import json
import math
import os

import numpy as np
import torch


def get_ray_directions(H, W, K, device='cpu', random=False, return_uv=False, flatten=True):
    """ray directions of all pixels in camera coordinates [right down front] (ray_utils.py:8-47):
    ((u - cx + 0.5)/fx, (v - cy + 0.5)/fy, 1), u = column, v = row; `random` jitters inside the pixel"""
    v, u = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=device),
                          torch.arange(W, dtype=torch.float32, device=device), indexing="ij")
    fx, fy, cx, cy = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])
    if random:
        directions = torch.stack([(u - cx + torch.rand_like(u)) / fx, (v - cy + torch.rand_like(v)) / fy,
                                  torch.ones_like(u)], -1)
    else:
        directions = torch.stack([(u - cx + 0.5) / fx, (v - cy + 0.5) / fy, torch.ones_like(u)], -1)
    grid = torch.stack([u, v], -1)
    if flatten:
        directions = directions.reshape(-1, 3)
        grid = grid.reshape(-1, 2)
    if return_uv:
        return directions, grid
    return directions


def get_rays(directions, c2w):
    """world-space origins and (unnormalised) directions (ray_utils.py:50-74).
    directions (N,3); c2w (3,4) or (N,3,4)"""
    if c2w.ndim == 2:
        rays_d = directions @ c2w[:, :3].T
    else:
        rays_d = (directions[:, None, :] @ c2w[..., :3].transpose(1, 2))[:, 0]
    rays_o = c2w[..., 3].expand_as(rays_d)
    return rays_o, rays_d


def read_image(img_path, img_wh):
    """(h*w, 3) float32 in [0,1]; RGBA is blended on white (color_utils.py:19-28)"""
    from PIL import Image
    im = Image.open(img_path)
    if im.mode not in ("RGB", "RGBA"):
        im = im.convert("RGBA" if "A" in im.mode else "RGB")
    img = np.asarray(im, dtype=np.float32) / 255.0
    if img.shape[2] == 4:
        img = img[..., :3] * img[..., -1:] + (1 - img[..., -1:])
    if (img.shape[1], img.shape[0]) != tuple(img_wh):
        chans = [np.asarray(Image.fromarray(img[..., c]).resize(tuple(img_wh), Image.BILINEAR)) for c in range(3)]
        img = np.stack(chans, -1)
    return np.ascontiguousarray(img.reshape(-1, 3), dtype=np.float32)


class BaseDataset(torch.utils.data.Dataset):
    """length and sampling of datasets/base.py:5-66 (train: 1000 random batches per epoch)"""

    def __init__(self, root_dir, split='train', downsample=1.0):
        self.root_dir = root_dir
        self.split = split
        self.downsample = downsample
        self.batch_size = 8192
        self.ray_sampling_strategy = 'all_images'

    def __len__(self):
        if self.split.startswith('train'):
            return 1000
        return len(self.poses)

    def __getitem__(self, idx):
        if self.split.startswith('train'):
            dev = self.rays.device
            n_img = len(self.poses)
            if self.ray_sampling_strategy == 'all_images':
                img_idxs = torch.randint(n_img, (self.batch_size,), device=dev)
            else:  # 'same_image'
                img_idxs = torch.randint(n_img, (1,), device=dev).expand(self.batch_size).contiguous()
            w, h = self.img_wh
            pix_idxs = torch.randint(w * h, (self.batch_size,), device=dev)
            rays = self.rays[img_idxs, pix_idxs]
            # the reference names these the other way round (u = pix // w is the row), kept as is
            uv = torch.stack([pix_idxs // w, pix_idxs % w], -1)
            sample = {'img_idxs': img_idxs, 'pix_idxs': pix_idxs, 'uv': uv, 'rgb': rays[:, :3]}
            if self.rays.shape[-1] == 4:
                sample['exposure'] = rays[:, 3:]
        else:
            sample = {'pose': self.poses[idx], 'img_idxs': idx}
            if len(self.rays) > 0:
                rays = self.rays[idx]
                sample['rgb'] = rays[:, :3]
                if rays.shape[1] == 4:
                    sample['exposure'] = rays[0, 3]
        return sample


class NeRFDataset(BaseDataset):
    """NeRF-Synthetic scene directory (datasets/nerf.py:13-71): attributes K (3,3), directions
    (h*w,3), img_wh, rays (N_images, h*w, 3), poses (N_images, 3, 4) [right down front], camera
    centres scaled to radius 1.5."""

    def __init__(self, root_dir, split='train', downsample=1.0, device='cpu', **kwargs):
        super().__init__(root_dir, split, downsample)
        self.device = torch.device(device)
        self.read_intrinsics()
        if kwargs.get('read_meta', True):
            self.read_meta(split)

    def read_intrinsics(self):
        with open(os.path.join(self.root_dir, "transforms_train.json"), 'r') as f:
            meta = json.load(f)
        w = h = int(800 * self.downsample)
        fx = fy = 0.5 * 800 / np.tan(0.5 * meta['camera_angle_x']) * self.downsample
        K = np.float32([[fx, 0, w / 2], [0, fy, h / 2], [0, 0, 1]])
        self.K = torch.from_numpy(K)
        self.directions = get_ray_directions(h, w, self.K, device=self.device)
        self.img_wh = (w, h)

    def read_meta(self, split):
        with open(os.path.join(self.root_dir, f"transforms_{split}.json"), 'r') as f:
            meta = json.load(f)
        pose_radius_scale = 1.5
        rays, poses = [], []
        for frame in meta['frames']:
            c2w = np.array(frame['transform_matrix'], dtype=np.float64)[:3, :4]
            c2w[:, 1:3] *= -1  # [right up back] -> [right down front]
            c2w[:, 3] /= np.linalg.norm(c2w[:, 3]) / pose_radius_scale
            poses.append(c2w)
            rays.append(read_image(os.path.join(self.root_dir, f"{frame['file_path']}.png"), self.img_wh))
        self.rays = torch.from_numpy(np.stack(rays)).to(self.device)
        self.poses = torch.from_numpy(np.stack(poses).astype(np.float32)).to(self.device)

    def batch_rays(self, sample):
        """(rays_o, rays_d) of a train sample, as NeRFSystem.forward does (train.py:136-155)"""
        poses = self.poses[sample['img_idxs']]
        directions = self.directions[sample['pix_idxs']]
        return get_rays(directions, poses)


def write_synthetic_dataset(root_dir, scene, n_train=20, n_test=4, rgba=True, n_quad=256):
    """Exports views of an analytic scene (synthetic.LegoProxy) as a NeRF-Synthetic directory:
    PNGs and Blender-convention transforms.  rgba=True stores un-premultiplied colour + alpha =
    accumulated opacity (what Blender writes; the loader blends it on white); rgba=False stores the
    scene composited on black as RGB, which is the background the reference's renderer adds for
    synthetic scenes (rendering.py:231-232).  The image size must be int(800*downsample) for the
    downsample used at load time."""
    from PIL import Image
    w, h = scene.img_wh
    fx = float(scene.K[0, 0])
    angle_x = 2 * math.atan(0.5 * w / fx)
    n_total = scene.poses.shape[0]
    assert n_train + n_test <= n_total
    pix = torch.arange(w * h, device=scene.device)
    splits = {"train": range(0, n_train), "test": range(n_train, n_train + n_test), "val": range(n_train, n_train + n_test)}
    for split, idxs in splits.items():
        os.makedirs(os.path.join(root_dir, split), exist_ok=True)
        frames = []
        for k, i in enumerate(idxs):
            img_idx = torch.full((w * h,), i, dtype=torch.long, device=scene.device)
            o, d = scene.rays(img_idx, pix)
            rgb, opacity = scene.ground_truth(o, d, n_quad=n_quad)
            a = opacity.clamp(0, 1)[:, None]
            colour = torch.where(a > 1e-6, rgb / a.clamp(min=1e-6), torch.zeros_like(rgb)).clamp(0, 1)
            name = f"r_{k}"
            if rgba:
                arr = (torch.cat([colour, a], -1).reshape(h, w, 4).cpu().numpy() * 255.0 + 0.5).astype(np.uint8)
                Image.fromarray(arr, "RGBA").save(os.path.join(root_dir, split, name + ".png"))
            else:
                arr = (rgb.clamp(0, 1).reshape(h, w, 3).cpu().numpy() * 255.0 + 0.5).astype(np.uint8)
                Image.fromarray(arr, "RGB").save(os.path.join(root_dir, split, name + ".png"))
            c2w = scene.poses[i].cpu().numpy().astype(np.float64)  # [right down front], |t| = radius
            blender = c2w.copy()
            blender[:, 1:3] *= -1
            m = np.eye(4)
            m[:3, :4] = blender
            frames.append({"file_path": f"./{split}/{name}", "transform_matrix": m.tolist()})
        with open(os.path.join(root_dir, f"transforms_{split}.json"), "w") as f:
            json.dump({"camera_angle_x": angle_x, "frames": frames}, f)
    return root_dir
