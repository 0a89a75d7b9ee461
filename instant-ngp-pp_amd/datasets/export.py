"""Writers for the on-disk formats the loaders read, so that synthetic scenes can stand in for the
datasets this repository cannot ship (no NeRF-Synthetic / T&T / mip-NeRF-360 files exist here or on
the GPU box).  Generic writers take images (n,h,w,3|4) uint8, camera-to-world poses (n,3,4) in
[right down front] axes and a 3x3 pinhole matrix; `write_synthetic_dataset` renders the analytic
lego-proxy scene and writes it as a NeRF-Synthetic directory."""
import json
import math
import os

import numpy as np
import torch

from . import colmap_utils as cu


def _save_png(arr, path):
    from PIL import Image
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(arr, {2: "L", 3: "RGB", 4: "RGBA"}[arr.shape[2] if arr.ndim == 3 else 2]).save(path)


def _homog(c2w):
    m = np.eye(4)
    m[:3, :4] = c2w
    return m


def export_blender(root, images, c2w, camera_angle_x, splits):
    """NeRF-Synthetic: splits = {"train": [frame indices], ...}; poses are stored in Blender axes"""
    for split, idxs in splits.items():
        frames = []
        for k, i in enumerate(idxs):
            _save_png(images[i], os.path.join(root, split, f"r_{k}.png"))
            m = _homog(c2w[i])
            m[:3, 1:3] *= -1
            frames.append({"file_path": f"./{split}/r_{k}", "transform_matrix": m.tolist()})
        with open(os.path.join(root, f"transforms_{split}.json"), "w") as f:
            json.dump({"camera_angle_x": camera_angle_x, "frames": frames}, f)
    return root


def export_tnt(root, images, c2w, K, split_of, img_dir="images", labels=None, depths=None, camera_path=None,
               flat_intrinsics=False):
    """T&T / NSVF layout: split_of[i] in {0,1,2} is the file-name prefix of frame i.
    labels (n,h,w) uint8 -> semantic/*.pgm, depths (n,h,w) float -> depth/*.npy,
    camera_path (m,3,4) -> camera_path/pose/<5 digits>.txt"""
    os.makedirs(os.path.join(root, "pose"), exist_ok=True)
    K4 = np.eye(4)
    K4[:3, :3] = K
    np.savetxt(os.path.join(root, "intrinsics.txt"), K4.reshape(1, 16) if flat_intrinsics else K4)
    for i in range(len(images)):
        stem = f"{split_of[i]}_{i:08d}"
        _save_png(images[i], os.path.join(root, img_dir, stem + ".png"))
        np.savetxt(os.path.join(root, "pose", stem + ".txt"), _homog(c2w[i]))
        if labels is not None:
            _save_png(labels[i], os.path.join(root, "semantic", stem + ".pgm"))
        if depths is not None:
            os.makedirs(os.path.join(root, "depth"), exist_ok=True)
            np.save(os.path.join(root, "depth", stem + ".npy"), depths[i])
    if camera_path is not None:
        os.makedirs(os.path.join(root, "camera_path", "pose"), exist_ok=True)
        for j, pose in enumerate(camera_path):
            np.savetxt(os.path.join(root, "camera_path", "pose", f"path_{j:05d}.txt"), _homog(pose))
    return root


def export_nsvf(root, images, c2w, K, split_of, bbox):
    """NSVF layout = the T&T one with rgb/ and bbox.txt"""
    export_tnt(root, images, c2w, K, split_of, img_dir="rgb")
    np.savetxt(os.path.join(root, "bbox.txt"), np.asarray(bbox, dtype=np.float64).reshape(1, -1))
    return root


def export_nerfpp(root, images, c2w, K, splits):
    """NeRF++ layout: <split>/{rgb,pose,intrinsics}/, 16 numbers per text file"""
    K4 = np.eye(4)
    K4[:3, :3] = K
    for split, idxs in splits.items():
        for sub in ("rgb", "pose", "intrinsics"):
            os.makedirs(os.path.join(root, split, sub), exist_ok=True)
        for i in idxs:
            _save_png(images[i], os.path.join(root, split, "rgb", f"{i:05d}.png"))
            np.savetxt(os.path.join(root, split, "pose", f"{i:05d}.txt"), _homog(c2w[i]).reshape(1, 16))
            np.savetxt(os.path.join(root, split, "intrinsics", f"{i:05d}.txt"), K4.reshape(1, 16))
    return root


def export_colmap(root, images, c2w, K, names=None, points=None, model="PINHOLE", labels=None, shuffle_seed=None):
    """COLMAP layout: sparse/0/*.bin with ONE camera (id 1) and world-to-camera poses, images/<name>.
    `shuffle_seed` stores the image records in a permuted order (COLMAP registers images in
    reconstruction order, not name order — the loader has to sort)."""
    n, h, w = images.shape[0], images.shape[1], images.shape[2]
    names = names or [f"frame_{i:04d}.png" for i in range(n)]
    os.makedirs(os.path.join(root, "sparse", "0"), exist_ok=True)
    if model == "PINHOLE":
        params = [K[0, 0], K[1, 1], K[0, 2], K[1, 2]]
    elif model == "SIMPLE_RADIAL":
        params = [K[0, 0], K[0, 2], K[1, 2], 0.0]
    elif model == "OPENCV":
        params = [K[0, 0], K[1, 1], K[0, 2], K[1, 2], 0.0, 0.0, 0.0, 0.0]
    else:
        params = [0.0] * cu.CAMERA_MODELS[cu.CAMERA_MODEL_IDS[model]][1]
    cu.write_cameras_binary({1: cu.Camera(1, model, w, h, np.array(params, dtype=np.float64))},
                            os.path.join(root, "sparse/0/cameras.bin"))
    order = list(range(n))
    if shuffle_seed is not None:
        order = list(np.random.default_rng(shuffle_seed).permutation(n))
    records = {}
    for i in order:
        w2c = np.linalg.inv(_homog(c2w[i]))
        records[i + 1] = cu.Image(i + 1, cu.rotmat2qvec(w2c[:3, :3]), w2c[:3, 3], 1, names[i],
                                  np.zeros((0, 2)), np.zeros(0, dtype=np.int64))
        _save_png(images[i], os.path.join(root, "images", names[i]))
        if labels is not None:
            _save_png(labels[i], os.path.join(root, "semantic", os.path.splitext(names[i])[0] + ".pgm"))
    cu.write_images_binary(records, os.path.join(root, "sparse/0/images.bin"))
    points = np.zeros((1, 3)) if points is None else np.asarray(points, dtype=np.float64)
    cloud = {j + 1: cu.Point3D(j + 1, p, np.array([128, 128, 128], dtype=np.uint8), 0.5,
                               np.array([1], dtype=np.int32), np.array([0], dtype=np.int32)) for j, p in enumerate(points)}
    cu.write_points3d_binary(cloud, os.path.join(root, "sparse/0/points3D.bin"))
    return root


@torch.no_grad()
def render_scene_views(scene, idxs, rgba=True, n_quad=256):
    """uint8 views of an analytic scene (synthetic.LegoProxy): rgba=True stores un-premultiplied
    colour + alpha = accumulated opacity (what Blender writes; loaders blend it on white);
    rgba=False stores the scene composited on black, the background the renderer adds for synthetic
    scenes (rendering.py:231-232)."""
    w, h = scene.img_wh
    pix = torch.arange(w * h, device=scene.device)
    out = []
    for i in idxs:
        o, d = scene.rays(torch.full((w * h,), int(i), dtype=torch.long, device=scene.device), pix)
        rgb, opacity = scene.ground_truth(o, d, n_quad=n_quad)
        if rgba:
            a = opacity.clamp(0, 1)[:, None]
            colour = torch.where(a > 1e-6, rgb / a.clamp(min=1e-6), torch.zeros_like(rgb)).clamp(0, 1)
            img = torch.cat([colour, a], -1).reshape(h, w, 4)
        else:
            img = rgb.clamp(0, 1).reshape(h, w, 3)
        out.append((img.cpu().numpy() * 255.0 + 0.5).astype(np.uint8))
    return np.stack(out)


def write_synthetic_dataset(root_dir, scene, n_train=20, n_test=4, rgba=True, n_quad=256):
    """Exports views of `scene` as a NeRF-Synthetic directory.  The image size must be
    int(800*downsample) for the downsample used at load time."""
    w, _ = scene.img_wh
    assert n_train + n_test <= scene.poses.shape[0]
    idxs = list(range(n_train + n_test))
    images = render_scene_views(scene, idxs, rgba=rgba, n_quad=n_quad)
    c2w = scene.poses[:n_train + n_test].cpu().numpy().astype(np.float64)
    held_out = list(range(n_train, n_train + n_test))
    return export_blender(root_dir, images, c2w, 2 * math.atan(0.5 * w / float(scene.K[0, 0])),
                          {"train": list(range(n_train)), "test": held_out, "val": held_out})
