"""Tanks-and-Temples style scenes — Playground etc. (reference: datasets/tnt.py:18-266).

On disk:
  <root>/intrinsics.txt                 3x3 or 4x4 (or 16 numbers) pinhole matrix at full resolution
  <root>/pose/<p>_<name>.txt            4x4 (or 3x4) camera-to-world, OpenCV axes [right down front];
                                        p = 0 train, 1 val / real-scene test, 2 synthetic-scene test
  <root>/{images|rgb}/<p>_<name>.png    images (RGBA is blended on white)
  <root>/semantic/<p>_<name>.pgm        optional labels (use_sem)
  <root>/depth/<p>_<name>.npy           optional monocular depth (depth_mono)
  <root>/camera_path/pose/*.txt         optional render trajectory (test split)
"""
import glob
import os

import numpy as np
import torch

from .base import BaseDataset
from .color_utils import read_image, read_semantic
from .ray_utils import get_ray_directions


def _name_key(x):
    # "<p>_<8 chars>.<ext>"-style names sort by their last 9 characters, anything else by the whole name
    return x[-9:] if len(x) > 2 and x[-10] == "_" else x


def _load_pose(path):
    m = np.loadtxt(path).reshape(-1, 4)
    return m if len(m) == 4 else np.concatenate([m, [[0.0, 0.0, 0.0, 1.0]]], 0)


class tntDataset(BaseDataset):
    """Translations are divided by the largest camera distance from the origin over ALL pose files
    (every split), rotations are kept; `up` = -mean camera y axis of this split.  Test split: rays of
    `camera_path/` (or, with render_train, of the train poses with three blends inserted between
    neighbours, at most 600) are precomputed as `render_traj_rays`."""

    def __init__(self, root_dir, split='train', downsample=1.0, cam_scale_factor=0.95, render_train=False,
                 device='cpu', **kwargs):
        super().__init__(root_dir, split, downsample)
        self.device = torch.device(device)
        img_dir = 'images' if os.path.exists(os.path.join(root_dir, 'images')) else 'rgb'
        if split == 'train':
            prefix = '0_'
        elif split == 'val':
            prefix = '1_'
        elif 'Synthetic' in root_dir:
            prefix = '2_'
        elif split == 'test':
            prefix = '1_'
        else:
            raise ValueError(f'{split} split not recognized!')

        def listing(sub, ext):
            return sorted(glob.glob(os.path.join(root_dir, sub, prefix + '*' + ext)), key=_name_key)

        imgs = listing(img_dir, '.png')
        semantics = listing('semantic', '.pgm') if kwargs.get('use_sem', False) else []
        depths = listing('depth', '.npy') if kwargs.get('depth_mono', False) else []
        pose_files = listing('pose', '.txt')

        from PIL import Image
        first = sorted(os.listdir(os.path.join(root_dir, img_dir)), key=_name_key)[0]
        with Image.open(os.path.join(root_dir, img_dir, first)) as im:
            w, h = int(im.width * downsample), int(im.height * downsample)
        K = np.loadtxt(os.path.join(root_dir, 'intrinsics.txt'), dtype=np.float32)
        if K.shape[0] > 4:
            K = K.reshape(4, 4)
        self.K = torch.from_numpy(K[:3, :3] * downsample)  # the whole matrix is scaled, K[2,2] included
        self.img_wh = (w, h)
        self.directions = get_ray_directions(h, w, self.K, device=self.device,
                                             anti_aliasing_factor=kwargs.get('anti_aliasing_factor', 1.0))

        c2w = np.stack([_load_pose(p) for p in pose_files])  # (n,4,4) float64
        up = -c2w[:, :3, 1].mean(0)
        self.up = torch.from_numpy(up / np.linalg.norm(up))
        every = sorted(os.listdir(os.path.join(root_dir, 'pose')), key=_name_key)
        scale = max(np.linalg.norm(np.loadtxt(os.path.join(root_dir, 'pose', f)).reshape(-1, 4)[..., 3]) for f in every)
        self.scene_scale = scale

        self.has_render_traj = split == "test" and not render_train and os.path.exists(os.path.join(root_dir, 'camera_path'))
        path_c2w = None
        if self.has_render_traj or render_train:
            path_dir = os.path.join(root_dir, "pose" if render_train else "camera_path/pose")
            names = sorted([x for x in os.listdir(path_dir) if x.endswith(".txt")], key=lambda x: int(x[-9:-4]))
            keys = [_load_pose(os.path.join(path_dir, x)) for x in names]
            if render_train:  # three blends (1/4, 1/2, 3/4) after every interior pose, capped at 600
                dense = []
                for i, pose in enumerate(keys):
                    if len(dense) >= 600:
                        break
                    dense.append(pose)
                    if 0 < i < len(keys) - 1:
                        dense += [(pose * (4 - k) + keys[i + 1] * k) / 4 for k in (1, 2, 3)]
                keys = dense
            path_c2w = np.stack(keys)
            path_c2w[..., 3] /= scale
            self.c2w = torch.from_numpy(path_c2w)
        c2w[..., 3] /= scale  # note: divides the homogeneous 1 as well; only rows 0..2 are used below

        self.imgs = imgs
        self.poses = torch.from_numpy(c2w[:, :3].astype(np.float32)).to(self.device)
        white_out = 'Jade' in root_dir or 'Fountain' in root_dir  # black-background scenes
        rays = []
        for path in imgs:
            img = read_image(path, self.img_wh)
            if white_out:
                img[np.all(img <= 0.1, axis=-1)] = 1.0
            rays.append(img)
        self.rays = torch.from_numpy(np.stack(rays)).to(self.device) if rays else torch.zeros(0)
        if semantics:
            classes = kwargs.get('num_classes', 7)
            self.labels = torch.from_numpy(np.stack([read_semantic(p, self.img_wh, classes) for p in semantics])
                                           .astype(np.int64))
        if depths:
            self.depths_2d = torch.from_numpy(np.stack([np.load(p).reshape(-1) for p in depths]).astype(np.float32))
        if not split.startswith('train') and path_c2w is not None:
            self.render_traj_rays = self.get_path_rays(path_c2w)
