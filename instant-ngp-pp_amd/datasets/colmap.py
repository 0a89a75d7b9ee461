"""COLMAP-reconstructed scenes: mip-NeRF-360, HDR-NeRF (reference: datasets/colmap.py:47-248).

On disk:
  <root>/sparse/0/{cameras,images,points3D}.bin   one shared camera (id 1), world-to-camera poses
  <root>/images[_<1/downsample>]/<name>           one image per registered frame
  <root>/semantic[_<1/downsample>]/<stem>.pgm     optional labels (use_sem)
"""
import glob
import os

import numpy as np
import torch

from .base import BaseDataset
from .color_utils import read_image, read_semantic
from .colmap_utils import read_cameras_binary, read_images_binary, read_points3d_binary
from .ray_utils import create_spheric_poses, generate_interpolated_path, get_ray_directions, normalize

# HDR-NeRF exposure times per scene: image suffix digit -> seconds (colmap.py:202-218)
_HDR_EXPOSURES = {}
for _scenes, _table in (
        (('bathroom', 'bear', 'chair', 'desk'), {e: 1 / 8 * 4 ** e for e in range(5)}),
        (('diningroom', 'dog'), {e: 1 / 16 * 4 ** e for e in range(5)}),
        (('sofa',), {0: 0.25, 1: 1, 2: 2, 3: 4, 4: 16}),
        (('sponza',), {0: 0.5, 1: 2, 2: 4, 3: 8, 4: 32}),
        (('box',), {0: 2 / 3, 1: 1 / 3, 2: 1 / 6, 3: 0.1, 4: 0.05}),
        (('computer',), {0: 1 / 3, 1: 1 / 8, 2: 1 / 15, 3: 1 / 30, 4: 1 / 60}),
        (('flower',), {0: 1 / 3, 1: 1 / 6, 2: 0.1, 3: 0.05, 4: 1 / 45}),
        (('luckycat',), {0: 2, 1: 1, 2: 0.5, 3: 0.25, 4: 0.125})):
    for _s in _scenes:
        _HDR_EXPOSURES[_s] = _table


class ColmapDataset(BaseDataset):
    """K from the shared camera (SIMPLE_RADIAL / PINHOLE / OPENCV; distortion ignored), poses =
    inverse of the stored world-to-camera matrices in file-name order, translations (and the point
    cloud) divided by the largest camera distance from the origin, `up` = -mean camera y axis.
    Splits: every 8th frame (by name) is test, the rest train; 'test_traj' is a synthetic fly-around."""

    def __init__(self, root_dir, split='train', downsample=1.0, device='cpu', **kwargs):
        super().__init__(root_dir, split, downsample)
        self.device = torch.device(device)
        self.read_intrinsics(**kwargs)
        if kwargs.get('read_meta', True):
            self.read_meta(split, **kwargs)

    def read_intrinsics(self, **kwargs):
        cam = read_cameras_binary(os.path.join(self.root_dir, 'sparse/0/cameras.bin'))[1]
        h, w = int(cam.height * self.downsample), int(cam.width * self.downsample)
        self.img_wh = (w, h)
        if cam.model == 'SIMPLE_RADIAL':
            fx = fy = cam.params[0] * self.downsample
            cx, cy = cam.params[1] * self.downsample, cam.params[2] * self.downsample
        elif cam.model in ('PINHOLE', 'OPENCV'):
            fx, fy, cx, cy = (cam.params[i] * self.downsample for i in range(4))
        else:
            raise ValueError(f"Please parse the intrinsics for camera model {cam.model}!")
        self.K = torch.tensor([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=torch.float32)
        self.directions = get_ray_directions(h, w, self.K, device=self.device,
                                             anti_aliasing_factor=kwargs.get('anti_aliasing_factor', 1.0))

    def _hdr_split(self, split, poses):
        """HDR-NeRF layout: several exposures per pose (colmap.py:134-166)"""
        pick = lambda pattern: sorted(glob.glob(os.path.join(self.root_dir, pattern)))
        if split not in ('train', 'test'):
            raise ValueError(f"split {split} is invalid for HDR-NeRF!")
        if 'syndata' in self.root_dir:  # synthetic: first 17 poses are test, last 18 train
            self.unit_exposure_rgb = 0.73
            if split == 'train':
                return pick('train/*[024].png'), np.repeat(poses[-18:], 3, 0)
            return pick('test/*[13].png'), np.repeat(poses[:17], 2, 0)
        self.unit_exposure_rgb = 0.5  # real: even poses train, odd poses test
        if split == 'train':
            paths = [p for d in '024' for p in pick(f'input_images/*{d}.jpg')[::2]]
            return paths, np.tile(poses[::2], (3, 1, 1))
        paths = [p for d in '13' for p in pick(f'input_images/*{d}.jpg')[1::2]]
        return paths, np.tile(poses[1::2], (2, 1, 1))

    def read_meta(self, split, **kwargs):
        imdata = read_images_binary(os.path.join(self.root_dir, 'sparse/0/images.bin'))
        records = list(imdata.values())
        names = [im.name for im in records]
        order = np.argsort(names)
        use_sem = kwargs.get('use_sem', False)
        suffix = f'_{int(1 / self.downsample)}' if '360' in self.root_dir and self.downsample < 1 else ''
        img_paths = [os.path.join(self.root_dir, 'images' + suffix, n) for n in sorted(names)]
        sem_paths = [os.path.join(self.root_dir, 'semantic' + suffix, os.path.splitext(n)[0] + '.pgm')
                     for n in sorted(names)] if use_sem else []

        w2c = np.tile(np.eye(4), (len(records), 1, 1))
        for i, im in enumerate(records):
            w2c[i, :3, :3] = im.qvec2rotmat()
            w2c[i, :3, 3] = im.tvec
        poses = np.linalg.inv(w2c)[order, :3]  # (N_images, 3, 4) camera-to-world
        cloud = read_points3d_binary(os.path.join(self.root_dir, 'sparse/0/points3D.bin'))
        pts3d = np.array([p.xyz for p in cloud.values()])

        self.up = torch.tensor(-normalize(poses[:, :3, 1].mean(0)), dtype=torch.float32)
        scale = np.linalg.norm(poses[..., 3], axis=-1).max()
        poses[..., 3] /= scale
        self.poses, self.pts3d = poses, pts3d / scale

        self.rays = []
        if use_sem:
            self.labels = []
        if split == 'test_traj':
            self.poses = torch.tensor(create_spheric_poses(1.2, self.poses[:, 1, 3].mean()), dtype=torch.float32)
            return

        hdr = 'HDR-NeRF' in self.root_dir
        path_poses = None
        if hdr:
            img_paths, self.poses = self._hdr_split(split, self.poses)
        elif split == 'train':
            keep = [i for i in range(len(img_paths)) if i % 8 != 0]
            img_paths, self.poses = [img_paths[i] for i in keep], self.poses[keep]
        elif split == 'test':
            path_poses = torch.tensor(self.poses, dtype=torch.float32)
            keep = [i for i in range(len(img_paths)) if i % 8 == 0]
            img_paths, self.poses = [img_paths[i] for i in keep], self.poses[keep]
            if kwargs.get('render_traj', False):
                path_poses = generate_interpolated_path(self.poses, 120)[400:800]

        for img_path in img_paths:
            img = torch.from_numpy(read_image(img_path, self.img_wh))
            if hdr:
                scene = [p for p in self.root_dir.split('/') if p][-1]
                digit = int(img_path.split('.')[0][-1])
                img = torch.cat([img, _HDR_EXPOSURES[scene][digit] * torch.ones_like(img[:, :1])], 1)
            self.rays.append(img)
        self.rays = torch.stack(self.rays).to(self.device)  # (N_images, hw, 3|4)
        self.poses = torch.tensor(self.poses, dtype=torch.float32).to(self.device)

        if use_sem:  # as upstream: labels of ALL registered frames, not only this split's
            classes = kwargs.get('num_classes', 7)
            self.labels = torch.from_numpy(np.stack([read_semantic(p, self.img_wh, classes) for p in sem_paths])
                                           .astype(np.int64))
        if split == 'test' and path_poses is not None:
            self.render_traj_rays = self.get_path_rays(path_poses)
