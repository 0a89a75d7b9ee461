"""Dataset loaders on the ray side of the hot path: the reference's
`datasets` package surface — `dataset_dict`, the loader classes, `ray_utils`, `color_utils`,
`colmap_utils` — with the same attribute names and sample dictionaries, plus `export`, which
writes synthetic scenes in each on-disk format.

Not provided: the KITTI-360, Mega-NeRF and Highbay loaders (datasets/kitti360.py, mega_nerf/,
highbay.py) — site-specific formats with their own calibration files, outside §8.
"""
from .base import BaseDataset
from .color_utils import read_image, read_normal, read_normal_up, read_semantic
from .colmap import ColmapDataset
from .export import write_synthetic_dataset
from .nerf import NeRFDataset
from .nerfpp import NeRFPPDataset
from .nsvf import NSVFDataset
from .ray_utils import (average_poses, axisangle_to_R, center_poses, create_spheric_poses,
                        generate_interpolated_path, get_ray_directions, get_rays)
from .tnt import tntDataset

dataset_dict = {'nerf': NeRFDataset,
                'nsvf': NSVFDataset,
                'colmap': ColmapDataset,
                'nerfpp': NeRFPPDataset,
                'tnt': tntDataset}
