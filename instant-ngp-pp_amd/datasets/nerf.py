"""NeRF-Synthetic (Blender) scenes (reference: datasets/nerf.py:13-68).

On disk:
  <root>/transforms_{train,val,test}.json : {"camera_angle_x": float,
        "frames": [{"file_path": "./train/r_0", "transform_matrix": 4x4 camera-to-world in Blender axes
                    [right, up, back]}, ...]}
  <root>/<file_path>.png                  : RGBA (blended on white) or RGB, 8 bit
"""
import json
import os

import numpy as np
import torch

from .base import BaseDataset
from .color_utils import read_image
from .ray_utils import get_ray_directions

FULL_RES = 800          # the renders are 800 x 800; `downsample` scales that
CAMERA_RADIUS = 1.5     # every camera centre is moved to this distance from the origin


class NeRFDataset(BaseDataset):
    """attributes K (3,3), directions (h*w,3), img_wh, rays (N_images, h*w, 3), poses (N_images, 3, 4)
    [right down front]"""

    def __init__(self, root_dir, split='train', downsample=1.0, device='cpu', **kwargs):
        super().__init__(root_dir, split, downsample)
        self.device = torch.device(device)
        self.read_intrinsics()
        if kwargs.get('read_meta', True):
            self.read_meta(split)

    def _transforms(self, split):
        with open(os.path.join(self.root_dir, f"transforms_{split}.json")) as f:
            return json.load(f)

    def read_intrinsics(self):
        """focal length from the horizontal field of view of the TRAIN split, principal point at the centre"""
        side = int(FULL_RES * self.downsample)
        focal = 0.5 * FULL_RES / np.tan(0.5 * self._transforms("train")['camera_angle_x']) * self.downsample
        self.K = torch.from_numpy(np.float32([[focal, 0, side / 2], [0, focal, side / 2], [0, 0, 1]]))
        self.img_wh = (side, side)
        self.directions = get_ray_directions(side, side, self.K, device=self.device)

    def read_meta(self, split):
        frames = self._transforms(split)['frames']
        c2w = np.array([fr['transform_matrix'] for fr in frames], dtype=np.float64)[:, :3, :4]
        c2w[:, :, 1:3] *= -1  # Blender [right up back] -> [right down front]
        c2w[:, :, 3] *= CAMERA_RADIUS / np.linalg.norm(c2w[:, :, 3], axis=-1, keepdims=True)
        pixels = [read_image(os.path.join(self.root_dir, fr['file_path'] + ".png"), self.img_wh) for fr in frames]
        self.rays = torch.from_numpy(np.stack(pixels)).to(self.device)
        self.poses = torch.from_numpy(c2w.astype(np.float32)).to(self.device)
