"""NeRF++ scenes: tat_*, lf_* (reference: datasets/nerfpp.py:13-59).

On disk:
  <root>/<split>/rgb/*            images
  <root>/<split>/pose/*.txt       16 numbers: 4x4 camera-to-world [right down front]
  <root>/train/intrinsics/*.txt   16 numbers: 4x4 pinhole matrix (the first file is used for all frames)
  <root>/camera_path/pose/*.txt   fly-through ('test_traj' split)
Poses are used as stored (the data is pre-normalised).
"""
import glob
import os

import numpy as np
import torch

from .base import BaseDataset
from .color_utils import read_image
from .ray_utils import get_ray_directions


class NeRFPPDataset(BaseDataset):
    def __init__(self, root_dir, split='train', downsample=1.0, device='cpu', **kwargs):
        super().__init__(root_dir, split, downsample)
        self.device = torch.device(device)
        self.read_intrinsics()
        if kwargs.get('read_meta', True):
            self.read_meta(split)

    def read_intrinsics(self):
        from PIL import Image
        K = np.loadtxt(glob.glob(os.path.join(self.root_dir, 'train/intrinsics/*.txt'))[0],
                       dtype=np.float32).reshape(4, 4)[:3, :3]
        K[:2] *= self.downsample
        with Image.open(glob.glob(os.path.join(self.root_dir, 'train/rgb/*'))[0]) as im:
            w, h = int(im.size[0] * self.downsample), int(im.size[1] * self.downsample)
        self.K = torch.from_numpy(K)
        self.directions = get_ray_directions(h, w, self.K, device=self.device)
        self.img_wh = (w, h)

    def read_meta(self, split):
        root = self.root_dir
        listing = lambda *parts: sorted(glob.glob(os.path.join(root, *parts)))
        self.rays = []
        if split == 'test_traj':
            pose_paths = listing('camera_path/pose/*.txt')
        else:
            parts = ('train', 'val') if split == 'trainval' else (split,)
            img_paths = [p for s in parts for p in listing(s, 'rgb/*')]
            pose_paths = [p for s in parts for p in listing(s, 'pose/*.txt')][:len(img_paths)]
            self.rays = torch.from_numpy(np.stack([read_image(p, self.img_wh) for p in img_paths])).to(self.device)
        poses = np.stack([np.loadtxt(p).reshape(4, 4)[:3] for p in pose_paths])
        self.poses = torch.from_numpy(poses.astype(np.float32)).to(self.device)
