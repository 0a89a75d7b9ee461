#!/usr/bin/env python3
"""Generates tests/golden/g12_datasets.npz by RUNNING THE REFERENCE'S OWN dataset loaders
(datasets/{nerf,colmap,tnt,nsvf,nerfpp}.py, ray_utils.py, colmap_utils.py) on small directories
written by this package's exporters (tests/helpers.py::write_dataset_dirs).

Runs only in the build container (needs /root/reference).  The loaders import three libraries that
are absent here; they get minimal stand-ins for exactly the calls the loaders make:
  kornia.create_meshgrid(H, W, False, device)  -> (1,H,W,2) pixel lattice, x (column) first
  imageio.imread(path)                         -> PIL decode to a numpy array
  cv2.resize(img, (w,h))                       -> identity (the fixture images are stored at the
                                                  requested size; anything else raises)
and the reference's `datasets/__init__.py` (which imports the KITTI / Mega-NeRF loaders and their
own dependencies) is bypassed by registering `datasets` as a bare package path.  Everything else —
file discovery and ordering, pose conventions, scaling, splits, COLMAP binary parsing, the path
generators — is the reference's code.  The fixture holds numbers only.
"""
import importlib
import os
import shutil
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT))


def import_reference_datasets():
    kornia = types.ModuleType("kornia")

    def create_meshgrid(height, width, normalized_coordinates=True, device="cpu"):
        assert not normalized_coordinates
        ys, xs = torch.meshgrid(torch.arange(height, dtype=torch.float32, device=device),
                                torch.arange(width, dtype=torch.float32, device=device), indexing="ij")
        return torch.stack([xs, ys], -1)[None]

    kornia.create_meshgrid = create_meshgrid
    imageio = types.ModuleType("imageio")

    def imread(path):
        from PIL import Image
        return np.asarray(Image.open(path))

    imageio.imread = imread
    cv2 = types.ModuleType("cv2")

    def resize(img, wh):
        if (img.shape[1], img.shape[0]) != tuple(wh):
            raise RuntimeError("fixture images must be stored at the requested size")
        return img

    cv2.resize = resize
    sys.modules.update(kornia=kornia, imageio=imageio, cv2=cv2)
    pkg = types.ModuleType("datasets")
    pkg.__path__ = [os.path.join(REF, "datasets")]
    sys.modules["datasets"] = pkg
    return {name: importlib.import_module(f"datasets.{name}") for name in
            ("ray_utils", "colmap_utils", "nerf", "colmap", "tnt", "nsvf", "nerfpp")}


def t2n(v):
    if torch.is_tensor(v):
        return v.detach().cpu().numpy()
    return np.asarray(v)


def main():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    import ngp_amd  # noqa: F401
    from ngp_amd.datasets import export
    ref = import_reference_datasets()
    classes = {"nerf": ref["nerf"].NeRFDataset, "colmap": ref["colmap"].ColmapDataset, "tnt": ref["tnt"].tntDataset,
               "nsvf": ref["nsvf"].NSVFDataset, "nerfpp": ref["nerfpp"].NeRFPPDataset}
    root = helpers.dataset_tmp_root()
    out = {}
    try:
        dirs = helpers.write_dataset_dirs(root, export)
        splits = {"nerf": ["train", "test"], "colmap": ["train", "test", "test_traj"], "colmap_radial": ["train"],
                  "tnt": ["train", "val", "test"], "nsvf": ["train", "trainval", "test"], "nsvf_mvs": ["train", "test", "test_traj"],
                  "nerfpp": ["train", "trainval", "test"]}
        for name, (key, path, kwargs) in dirs.items():
            for split in splits[name]:
                ds = classes[key](path, split=split, **kwargs)
                tag = f"{name}.{split}"
                out[tag + ".K"] = t2n(ds.K)
                out[tag + ".img_wh"] = np.array(ds.img_wh)
                out[tag + ".directions"] = t2n(ds.directions)
                out[tag + ".poses"] = t2n(ds.poses)
                if len(ds.rays) > 0:
                    out[tag + ".rays"] = t2n(ds.rays)
                for attr in ("up", "labels", "depths_2d", "pts3d", "shift", "scale", "c2w"):
                    if hasattr(ds, attr):
                        out[f"{tag}.{attr}"] = t2n(getattr(ds, attr))
                if hasattr(ds, "render_traj_rays"):
                    out[tag + ".n_traj"] = np.array(len(ds.render_traj_rays))
                    for j in (0, len(ds.render_traj_rays) - 1):
                        out[f"{tag}.traj{j}"] = t2n(ds.render_traj_rays[j])
                if not split.startswith("train") and len(ds.rays) > 0:
                    item = ds[len(ds) - 1]
                    out[tag + ".item_keys"] = np.array(sorted(item.keys()))
                    out[tag + ".item_rgb"] = t2n(item["rgb"])
                    out[tag + ".item_pose"] = t2n(item["pose"])
                print(tag, "poses", out[tag + ".poses"].shape, "rays", getattr(ds.rays, "shape", None))
        # tnt: interpolated train path
        key, path, kwargs = dirs["tnt"]
        ds = classes["tnt"](path, split="test", render_train=True, **kwargs)
        out["tnt.render_train.c2w"] = t2n(ds.c2w)
        out["tnt.render_train.n_traj"] = np.array(len(ds.render_traj_rays))
        out["tnt.render_train.traj3"] = t2n(ds.render_traj_rays[3])

        # the free functions of ray_utils
        ru = ref["ray_utils"]
        _, c2w, K = helpers.dataset_inputs(9, 12, 16)
        pts = np.stack([np.sin(np.arange(30) * 0.3), np.cos(np.arange(30) * 0.7), np.sin(np.arange(30) * 0.11)], 1)
        out["ru.average_poses"] = ru.average_poses(c2w, pts)
        out["ru.average_poses_nopts"] = ru.average_poses(c2w)
        cp, cpts = ru.center_poses(c2w, pts)
        out["ru.center_poses"], out["ru.center_pts"] = cp, cpts
        out["ru.center_poses_nopts"] = ru.center_poses(c2w)
        out["ru.spheric"] = ru.create_spheric_poses(1.2, -0.3, n_poses=7)
        out["ru.interp"] = ru.generate_interpolated_path(c2w, 4)
        out["ru.interp_few"] = ru.generate_interpolated_path(c2w[:3], 5)
        v = torch.tensor([[0.3, -0.2, 0.5], [0.0, 0.0, 0.0], [1.5, 2.0, -0.7]])
        out["ru.axisangle_in"], out["ru.axisangle"] = t2n(v), t2n(ru.axisangle_to_R(v))
        out["ru.axisangle_single"] = t2n(ru.axisangle_to_R(v[0]))
        Kt = torch.tensor(K, dtype=torch.float32)
        out["ru.dirs_aa2"] = t2n(ru.get_ray_directions(6, 8, Kt.clone(), anti_aliasing_factor=2.0))
        d, uv = ru.get_ray_directions(6, 8, Kt.clone(), return_uv=True, flatten=False)
        out["ru.dirs_grid"], out["ru.uv_grid"] = t2n(d), t2n(uv)
        ro, rd = ru.get_rays(torch.tensor(out["ru.dirs_aa2"][:9]), torch.tensor(c2w, dtype=torch.float32))
        out["ru.rays_o"], out["ru.rays_d"] = t2n(ro), t2n(rd)
        # colmap_utils: what the reference's reader sees in the files our writer produced
        cu = ref["colmap_utils"]
        cams = cu.read_cameras_binary(os.path.join(dirs["colmap"][1], "sparse/0/cameras.bin"))
        ims = cu.read_images_binary(os.path.join(dirs["colmap"][1], "sparse/0/images.bin"))
        p3d = cu.read_points3d_binary(os.path.join(dirs["colmap"][1], "sparse/0/points3D.bin"))
        out["cu.camera"] = np.concatenate([[cams[1].width, cams[1].height], cams[1].params])
        out["cu.camera_model"] = np.array(cams[1].model)
        out["cu.image_ids"] = np.array(list(ims.keys()))
        out["cu.image_names"] = np.array([ims[k].name for k in ims])
        out["cu.qvecs"] = np.stack([ims[k].qvec for k in ims])
        out["cu.tvecs"] = np.stack([ims[k].tvec for k in ims])
        out["cu.rotmats"] = np.stack([ims[k].qvec2rotmat() for k in ims])
        out["cu.xyz"] = np.stack([p3d[k].xyz for k in p3d])
    finally:
        shutil.rmtree(root, ignore_errors=True)
    np.savez_compressed(os.path.join(OUT, "g12_datasets.npz"), **out)
    print("wrote g12_datasets.npz with", len(out), "arrays,", os.path.getsize(os.path.join(OUT, "g12_datasets.npz")), "bytes")


if __name__ == "__main__":
    main()
