"""Dataset loaders — host logic, runs without a GPU.

Two layers: known-answer tests of the documented conventions (datasets/nerf.py:22-71,
ray_utils.py:8-74, color_utils.py:19-28, base.py:18-66) with a round trip through the on-disk
format, and G12 — the reference's OWN loaders run on directories written by `datasets.export`
(tests/golden/make_golden_datasets.py; kornia / imageio / cv2 are absent here and are replaced by
minimal stand-ins for the three calls the loaders make), compared array by array.  Image resizing
(stored size != requested size) is not covered by G12: the reference uses cv2.resize, this package
PIL — parity unpinned for that case."""
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


# ---- G12: the reference's OWN loaders, run on directories written by datasets.export ----------------
# (tests/golden/make_golden_datasets.py; kornia / imageio / cv2 replaced by minimal stand-ins, every
# other line — file discovery, ordering, pose conventions, scaling, splits, COLMAP parsing — is the
# reference's).  The same directories are rebuilt here from the same integer rules.
G12_SPLITS = {"nerf": ["train", "test"], "colmap": ["train", "test", "test_traj"], "colmap_radial": ["train"],
              "tnt": ["train", "val", "test"], "nsvf": ["train", "trainval", "test"],
              "nsvf_mvs": ["train", "test", "test_traj"], "nerfpp": ["train", "trainval", "test"]}


@pytest.fixture(scope="module")
def g12():
    import shutil
    import helpers
    from ngp_amd.datasets import export
    root = helpers.dataset_tmp_root()
    dirs = helpers.write_dataset_dirs(root, export)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "g12_datasets.npz"))
    yield dirs, gold
    shutil.rmtree(root, ignore_errors=True)


def _n(v):
    return v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)


@pytest.mark.parametrize("name", list(G12_SPLITS))
def test_loader_matches_reference_loader_golden(g12, name):
    dirs, gold = g12
    key, path, kwargs = dirs[name]
    for split in G12_SPLITS[name]:
        ds = datasets.dataset_dict[key](path, split=split, **kwargs)
        tag = f"{name}.{split}"
        assert tuple(ds.img_wh) == tuple(gold[tag + ".img_wh"])
        np.testing.assert_allclose(_n(ds.K), gold[tag + ".K"], rtol=1e-6, atol=0)
        np.testing.assert_allclose(_n(ds.directions), gold[tag + ".directions"], rtol=1e-6, atol=1e-7)
        assert _n(ds.poses).dtype == np.float32
        np.testing.assert_allclose(_n(ds.poses), gold[tag + ".poses"], rtol=1e-6, atol=1e-7)
        if tag + ".rays" in gold:
            assert np.array_equal(_n(ds.rays), gold[tag + ".rays"])      # same decode, same blend: bit-exact
        else:
            assert len(ds.rays) == 0
        for attr in ("up", "labels", "depths_2d", "pts3d", "shift", "scale", "c2w"):
            if f"{tag}.{attr}" in gold:
                got = _n(getattr(ds, attr))
                if attr == "labels" and gold[f"{tag}.{attr}"].size == 0:
                    assert len(got) == 0
                elif attr == "labels":
                    assert np.array_equal(got, gold[f"{tag}.{attr}"].astype(np.int64))
                else:
                    np.testing.assert_allclose(got, gold[f"{tag}.{attr}"], rtol=1e-6, atol=1e-7)
        if tag + ".n_traj" in gold:
            assert len(ds.render_traj_rays) == int(gold[tag + ".n_traj"])
            for j in (0, len(ds.render_traj_rays) - 1):
                np.testing.assert_allclose(_n(ds.render_traj_rays[j]), gold[f"{tag}.traj{j}"], rtol=1e-5, atol=1e-6)
        if tag + ".item_keys" in gold:
            item = ds[len(ds) - 1]
            assert sorted(item.keys()) == list(gold[tag + ".item_keys"])
            assert np.array_equal(_n(item["rgb"]), gold[tag + ".item_rgb"])
            np.testing.assert_allclose(_n(item["pose"]), gold[tag + ".item_pose"], rtol=1e-6, atol=1e-7)
        if split.startswith("train"):
            ds.batch_size = 64
            s = ds[0]
            assert s["rgb"].shape == (64, 3) and int(s["img_idxs"].max()) < len(ds.poses)
            assert torch.equal(s["rgb"], ds.rays[s["img_idxs"], s["pix_idxs"]][:, :3])
            if hasattr(ds, "labels"):
                assert torch.equal(s["label"], ds.labels[s["img_idxs"], s["pix_idxs"]])
            if hasattr(ds, "depths_2d"):
                assert torch.equal(s["depth"], ds.depths_2d[s["img_idxs"], s["pix_idxs"]])
            o, d = ds.batch_rays(s)
            assert o.shape == d.shape == (64, 3)


def test_tnt_render_train_path_matches_reference(g12):
    dirs, gold = g12
    key, path, kwargs = dirs["tnt"]
    ds = datasets.tntDataset(path, split="test", render_train=True, **kwargs)
    np.testing.assert_allclose(_n(ds.c2w), gold["tnt.render_train.c2w"], rtol=1e-12, atol=1e-12)
    assert len(ds.render_traj_rays) == int(gold["tnt.render_train.n_traj"])
    np.testing.assert_allclose(_n(ds.render_traj_rays[3]), gold["tnt.render_train.traj3"], rtol=1e-5, atol=1e-6)


def test_ray_utils_match_reference_golden(g12):
    import helpers
    from ngp_amd.datasets import ray_utils as ru
    _, gold = g12
    _, c2w, K = helpers.dataset_inputs(9, 12, 16)
    pts = np.stack([np.sin(np.arange(30) * 0.3), np.cos(np.arange(30) * 0.7), np.sin(np.arange(30) * 0.11)], 1)
    tight = dict(rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(ru.average_poses(c2w, pts), gold["ru.average_poses"], **tight)
    np.testing.assert_allclose(ru.average_poses(c2w), gold["ru.average_poses_nopts"], **tight)
    cp, cpts = ru.center_poses(c2w, pts)
    np.testing.assert_allclose(cp, gold["ru.center_poses"], **tight)
    np.testing.assert_allclose(cpts, gold["ru.center_pts"], **tight)
    np.testing.assert_allclose(ru.center_poses(c2w), gold["ru.center_poses_nopts"], **tight)
    np.testing.assert_allclose(ru.create_spheric_poses(1.2, -0.3, n_poses=7), gold["ru.spheric"], **tight)
    np.testing.assert_allclose(ru.generate_interpolated_path(c2w, 4), gold["ru.interp"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(ru.generate_interpolated_path(c2w[:3], 5), gold["ru.interp_few"], rtol=1e-9, atol=1e-9)
    v = torch.tensor(gold["ru.axisangle_in"])
    np.testing.assert_allclose(_n(ru.axisangle_to_R(v)), gold["ru.axisangle"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(_n(ru.axisangle_to_R(v[0])), gold["ru.axisangle_single"], rtol=1e-6, atol=1e-7)
    Kt = torch.tensor(K, dtype=torch.float32)
    Kaa = Kt.clone()
    np.testing.assert_allclose(_n(ru.get_ray_directions(6, 8, Kaa, anti_aliasing_factor=2.0)), gold["ru.dirs_aa2"], rtol=1e-6)
    assert float(Kaa[0, 0]) == 2 * float(Kt[0, 0]) and float(Kaa[2, 2]) == 1.0     # K is scaled in place, as upstream
    d, uv = ru.get_ray_directions(6, 8, Kt.clone(), return_uv=True, flatten=False)
    np.testing.assert_allclose(_n(d), gold["ru.dirs_grid"], rtol=1e-6)
    assert np.array_equal(_n(uv), gold["ru.uv_grid"])
    ro, rd = ru.get_rays(torch.tensor(gold["ru.dirs_aa2"][:9]), torch.tensor(c2w, dtype=torch.float32))
    np.testing.assert_allclose(_n(ro), gold["ru.rays_o"], rtol=1e-6)
    np.testing.assert_allclose(_n(rd), gold["ru.rays_d"], rtol=1e-5, atol=1e-6)


def test_colmap_binary_io_matches_reference_reader(g12):
    """our reader on our files == the reference's (stock COLMAP) reader on the same files"""
    from ngp_amd.datasets import colmap_utils as cu
    dirs, gold = g12
    base = os.path.join(dirs["colmap"][1], "sparse/0")
    cam = cu.read_cameras_binary(os.path.join(base, "cameras.bin"))[1]
    assert cam.model == str(gold["cu.camera_model"])
    assert np.array_equal(np.concatenate([[cam.width, cam.height], cam.params]), gold["cu.camera"])
    ims = cu.read_images_binary(os.path.join(base, "images.bin"))
    assert list(ims.keys()) == list(gold["cu.image_ids"])
    assert [ims[k].name for k in ims] == list(gold["cu.image_names"])
    assert np.array_equal(np.stack([ims[k].qvec for k in ims]), gold["cu.qvecs"])
    assert np.array_equal(np.stack([ims[k].tvec for k in ims]), gold["cu.tvecs"])
    np.testing.assert_allclose(np.stack([ims[k].qvec2rotmat() for k in ims]), gold["cu.rotmats"], rtol=0, atol=1e-15)
    pts = cu.read_points3d_binary(os.path.join(base, "points3D.bin"))
    assert np.array_equal(np.stack([pts[k].xyz for k in pts]), gold["cu.xyz"])
    # quaternion <-> matrix round trip incl. a half-turn (w = 0)
    for R in (np.diag([1.0, -1.0, -1.0]), np.diag([-1.0, -1.0, 1.0]), cu.qvec2rotmat(np.array([0.5, -0.5, 0.5, 0.5]))):
        np.testing.assert_allclose(cu.qvec2rotmat(cu.rotmat2qvec(R)), R, atol=1e-12)


def test_color_utils_curves_and_unsupported_camera(tmp_path):
    from ngp_amd.datasets import color_utils, export
    x = np.linspace(0, 1, 101)
    np.testing.assert_allclose(color_utils.linear_to_srgb(color_utils.srgb_to_linear(x)), x, atol=1e-6)
    assert color_utils.linear_to_srgb(np.array([4.0]))[0] == 1.0
    import helpers
    img, c2w, K = helpers.dataset_inputs(3, 6, 8)
    root = export.export_colmap(str(tmp_path / "scene"), img[..., :3], c2w, K, model="FOV")
    with pytest.raises(ValueError, match="camera model FOV"):
        datasets.ColmapDataset(root)
