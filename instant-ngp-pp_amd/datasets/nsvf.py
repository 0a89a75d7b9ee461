"""NSVF-format scenes: Synthetic_NSVF, Synthetic_NeRF, BlendedMVS, TanksAndTemple
(reference: datasets/nsvf.py:13-99).

On disk:
  <root>/intrinsics.txt   Synthetic / Ignatius: first number = focal length; otherwise a 4x4 matrix
  <root>/bbox.txt         xmin ymin zmin xmax ymax zmax [voxel size]
  <root>/pose/<p>_*.txt   4x4 camera-to-world [right down front]; p = 0 train, 1 val / real test, 2 synthetic test
  <root>/rgb/<p>_*.png    images
  <root>/test_traj.txt or test_pose/*.txt   fly-through poses ('test_traj' split), [left down front]
The scene is recentred on the bounding-box centre and scaled so the (5 % enlarged) box spans [-0.5, 0.5].
"""
import glob
import os

import numpy as np
import torch

from .base import BaseDataset
from .color_utils import read_image
from .ray_utils import get_ray_directions

# image sizes are not stored in the files: they are known per collection (matched on the path)
_FULL_SIZE = (('Synthetic', (800, 800)), ('Ignatius', (1920, 1080)), ('BlendedMVS', (768, 576)), ('Tanks', (1920, 1080)))
# hand-tuned enlargements of the bounding box for two scenes
_BOUND_FIX = (('Mic', 1.2), ('Lego', 1.1))


def _first_match(table, path, default=None):
    return next((val for word, val in table if word in path), default)


class NSVFDataset(BaseDataset):
    def __init__(self, root_dir, split='train', downsample=1.0, device='cpu', **kwargs):
        super().__init__(root_dir, split, downsample)
        self.device = torch.device(device)
        self.read_intrinsics()
        if not kwargs.get('read_meta', True):
            return
        box = np.loadtxt(os.path.join(root_dir, 'bbox.txt'))[:6].reshape(2, 3)
        self.shift = box.sum(0) / 2
        self.scale = (box[1] - box[0]).max() / 2 * 1.05
        self.scale *= _first_match(_BOUND_FIX, root_dir, 1.0)
        self.read_meta(split)

    def read_intrinsics(self):
        path = os.path.join(self.root_dir, 'intrinsics.txt')
        size = _first_match(_FULL_SIZE, self.root_dir)
        if size is None:
            raise ValueError("image size is only known for Synthetic / Ignatius / BlendedMVS / Tanks scenes")
        w, h = (int(s * self.downsample) for s in size)
        if 'Synthetic' in self.root_dir or 'Ignatius' in self.root_dir:
            with open(path) as f:
                focal = float(f.readline().split()[0]) * self.downsample
            K = np.float32([[focal, 0, w / 2], [0, focal, h / 2], [0, 0, 1]])
        else:
            K = np.loadtxt(path, dtype=np.float32)[:3, :3]
            K[:2] *= self.downsample
        self.K = torch.from_numpy(K)
        self.img_wh = (w, h)
        self.directions = get_ray_directions(h, w, self.K, device=self.device)

    def _into_unit_box(self, pose44, mirror_x=False):
        c2w = np.array(pose44, dtype=np.float64)[:3]
        if mirror_x:
            c2w[:, 0] = -c2w[:, 0]
        c2w[:, 3] = (c2w[:, 3] - self.shift) / (2 * self.scale)
        return c2w

    def _prefix(self, split):
        known = {'train': '0_', 'trainval': '[0-1]_', 'val': '1_'}
        if split in known:
            return known[split]
        if 'Synthetic' in self.root_dir:
            return '2_'
        if split == 'test':
            return '1_'
        raise ValueError(f'{split} split not recognized!')

    def read_meta(self, split):
        root = self.root_dir
        self.rays = []
        if split == 'test_traj':
            if 'Ignatius' in root:
                raw = [np.loadtxt(p) for p in sorted(glob.glob(os.path.join(root, 'test_pose/*.txt')))]
            else:
                raw = np.loadtxt(os.path.join(root, 'test_traj.txt')).reshape(-1, 4, 4)
            poses = [self._into_unit_box(m, mirror_x=True) for m in raw]
        else:
            pattern = self._prefix(split) + '*'
            pose_files = sorted(glob.glob(os.path.join(root, 'pose', pattern + '.txt')))
            image_files = sorted(glob.glob(os.path.join(root, 'rgb', pattern + '.png')))
            n = min(len(pose_files), len(image_files))
            poses = [self._into_unit_box(np.loadtxt(p)) for p in pose_files[:n]]
            black_bg = 'Jade' in root or 'Fountain' in root
            for f in image_files[:n]:
                px = read_image(f, self.img_wh)
                if black_bg:  # these two scenes come on black; the recipe trains on white
                    px[(px <= 0.1).all(-1)] = 1.0
                self.rays.append(px)
            self.rays = torch.from_numpy(np.stack(self.rays)).to(self.device)
        self.poses = torch.from_numpy(np.stack(poses).astype(np.float32)).to(self.device)
