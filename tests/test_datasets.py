"""NeRF-Synthetic loader — host logic, runs without a GPU.

The reference loader cannot be imported here (kornia / cv2 / imageio are absent), so there are no
golden vectors from it: parity unpinned; the tests below are known-answer tests of the documented
conventions (datasets/nerf.py:22-71, ray_utils.py:8-74, color_utils.py:19-28, base.py:18-66) and a
round trip through the on-disk format."""
import json
import math
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ngp_amd  # noqa: E402,F401
from ngp_amd import datasets  # noqa: E402
from ngp_amd.synthetic import LegoProxy  # noqa: E402


def test_ray_directions_pixel_centres():
    K = torch.tensor([[100.0, 0, 4.0], [0, 50.0, 3.0], [0, 0, 1]])
    d, uv = datasets.get_ray_directions(6, 8, K, return_uv=True)
    assert d.shape == (48, 3) and uv.shape == (48, 2)
    # pixel (row v=2, column u=5) -> ((5 - 4 + .5)/100, (2 - 3 + .5)/50, 1), row-major flattening
    i = 2 * 8 + 5
    assert torch.allclose(d[i], torch.tensor([1.5 / 100, -0.5 / 50, 1.0]))
    assert uv[i].tolist() == [5.0, 2.0]
    d2 = datasets.get_ray_directions(6, 8, K, flatten=False)
    assert d2.shape == (6, 8, 3) and torch.equal(d2.reshape(-1, 3), d)
    torch.manual_seed(0)
    dr = datasets.get_ray_directions(6, 8, K, random=True)
    lo = torch.stack([(uv[:, 0] - 4) / 100, (uv[:, 1] - 3) / 50], -1)
    hi = torch.stack([(uv[:, 0] - 4 + 1) / 100, (uv[:, 1] - 3 + 1) / 50], -1)
    assert ((dr[:, :2] >= lo) & (dr[:, :2] <= hi)).all()


def test_get_rays_single_and_per_ray_poses():
    g = torch.Generator().manual_seed(1)
    dirs = torch.randn(50, 3, generator=g)
    c2w = torch.randn(3, 4, generator=g)
    o, d = datasets.get_rays(dirs, c2w)
    assert torch.allclose(d, dirs @ c2w[:, :3].T) and torch.equal(o, c2w[:, 3].expand(50, 3))
    many = torch.randn(50, 3, 4, generator=g)
    o2, d2 = datasets.get_rays(dirs, many)
    for i in (0, 17, 49):
        assert torch.allclose(d2[i], many[i, :, :3] @ dirs[i], atol=1e-6)
        assert torch.equal(o2[i], many[i, :, 3])


@pytest.fixture(scope="module")
def tiny_scene(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("lego_proxy"))
    scene = LegoProxy(n_images=6, img_wh=(40, 40), device="cpu")     # int(800 * 0.05) = 40
    datasets.write_synthetic_dataset(root, scene, n_train=4, n_test=2, rgba=True, n_quad=96)
    return root, scene


def test_dataset_round_trip(tiny_scene):
    root, scene = tiny_scene
    meta = json.load(open(os.path.join(root, "transforms_train.json")))
    assert abs(meta["camera_angle_x"] - 0.6911112070083618) < 1e-6 and len(meta["frames"]) == 4
    ds = datasets.NeRFDataset(root, "train", downsample=0.05)
    assert ds.img_wh == (40, 40) and ds.rays.shape == (4, 1600, 3) and ds.poses.shape == (4, 3, 4)
    # intrinsics: fx = 0.5*800/tan(0.5*angle) * downsample, principal point at the image centre
    assert abs(float(ds.K[0, 0]) - float(scene.K[0, 0])) < 1e-3 and float(ds.K[0, 2]) == 20.0
    assert torch.allclose(ds.directions, scene.directions, atol=1e-6)
    # Blender [right up back] -> [right down front], camera centres at radius 1.5
    assert torch.allclose(ds.poses, scene.poses[:4], atol=1e-6)
    assert torch.allclose(ds.poses[:, :, 3].norm(dim=-1), torch.full((4,), 1.5), atol=1e-6)
    # pixels: RGBA blended on white = rgb_on_black + (1 - opacity), 8-bit quantisation on both factors
    pix = torch.arange(1600)
    o, d = scene.rays(torch.zeros(1600, dtype=torch.long), pix)
    rgb, opacity = scene.ground_truth(o, d, n_quad=96)
    want = rgb + (1 - opacity.clamp(0, 1))[:, None]
    assert (ds.rays[0] - want).abs().max() < 2.5 / 255
    assert ds.rays.min() >= 0 and ds.rays.max() <= 1
    test = datasets.NeRFDataset(root, "test", downsample=0.05)
    assert len(test) == 2 and torch.allclose(test.poses, scene.poses[4:6], atol=1e-6)


def test_rgb_export_is_black_background(tmp_path):
    scene = LegoProxy(n_images=3, img_wh=(40, 40), device="cpu")
    root = datasets.write_synthetic_dataset(str(tmp_path), scene, n_train=2, n_test=1, rgba=False, n_quad=96)
    ds = datasets.NeRFDataset(root, "train", downsample=0.05)
    o, d = scene.rays(torch.ones(1600, dtype=torch.long), torch.arange(1600))
    rgb, _ = scene.ground_truth(o, d, n_quad=96)
    assert (ds.rays[1] - rgb.clamp(0, 1)).abs().max() < 1.0 / 255
    assert float(ds.rays[1].min()) == 0.0        # background pixels stay black


def test_sampling_contract(tiny_scene):
    root, _ = tiny_scene
    ds = datasets.NeRFDataset(root, "train", downsample=0.05)
    assert len(ds) == 1000                         # 1000 random batches per "epoch" (base.py:18-21)
    ds.batch_size = 512
    torch.manual_seed(3)
    s = ds[0]
    assert set(s) == {"img_idxs", "pix_idxs", "uv", "rgb"}
    assert s["rgb"].shape == (512, 3) and s["img_idxs"].max() < 4 and s["pix_idxs"].max() < 1600
    assert torch.equal(s["rgb"], ds.rays[s["img_idxs"], s["pix_idxs"]])
    assert torch.equal(s["uv"][:, 0], s["pix_idxs"] // 40) and torch.equal(s["uv"][:, 1], s["pix_idxs"] % 40)
    assert len(torch.unique(s["img_idxs"])) > 1
    ds.ray_sampling_strategy = "same_image"
    s = ds[1]
    assert len(torch.unique(s["img_idxs"])) == 1
    o, d = ds.batch_rays(s)
    assert o.shape == (512, 3) and torch.allclose(o, ds.poses[s["img_idxs"]][:, :, 3])
    want = (ds.directions[s["pix_idxs"]][:, None, :] @ ds.poses[s["img_idxs"]][:, :, :3].transpose(1, 2))[:, 0]
    assert torch.allclose(d, want)
    test = datasets.NeRFDataset(root, "test", downsample=0.05)
    t = test[1]
    assert set(t) == {"pose", "img_idxs", "rgb"} and t["rgb"].shape == (1600, 3) and t["img_idxs"] == 1


def test_read_image_resizes_when_needed(tiny_scene):
    root, _ = tiny_scene
    img = datasets.read_image(os.path.join(root, "train", "r_0.png"), (20, 20))
    assert img.shape == (400, 3) and img.dtype == np.float32 and 0 <= img.min() and img.max() <= 1
