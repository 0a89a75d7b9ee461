"""Seeded synthetic inputs shared by the CPU and GPU tests (no datasets are available)."""
import numpy as np

SEED = 20220806


def rng(offset=0):
    return np.random.default_rng(SEED + offset)


def make_rays(n, scale=0.5, seed=0, miss_fraction=0.1):
    """Pinhole-like rays from cameras on a sphere of radius ~3*scale looking at the scene box;
    a fraction points away (misses).  Directions are NOT unit length (as get_rays output)."""
    g = rng(seed)
    o = g.normal(size=(n, 3))
    o = o / np.linalg.norm(o, axis=1, keepdims=True) * (3.0 * scale) * (1 + 0.2 * g.random((n, 1)))
    tgt = (g.random((n, 3)) - 0.5) * 1.6 * scale
    d = tgt - o
    d = d / np.linalg.norm(d, axis=1, keepdims=True) * (1.0 + 0.2 * g.random((n, 1)))
    miss = g.random(n) < miss_fraction
    d[miss] = -d[miss]
    return o.astype(np.float32), d.astype(np.float32)


def make_bitfield(cascades, grid_size=128, fill=0.05, seed=1, blobs=True):
    """Occupancy bitfield (cascades*G^3/8 bytes) in morton order: random blobs + noise."""
    g = rng(seed)
    n = cascades * grid_size ** 3
    if fill >= 1.0:
        return np.full(n // 8, 255, np.uint8)
    bits = g.random(n) < fill
    if blobs:  # contiguous morton ranges = spatially compact blocks
        for _ in range(40 * cascades):
            s = int(g.integers(0, n - 4096))
            bits[s:s + int(g.integers(256, 4096))] = True
    return np.packbits(bits.reshape(-1, 8)[:, ::-1], axis=1).reshape(-1).astype(np.uint8)


def make_segments(n_rays, max_len, seed=2, empty_fraction=0.15):
    """rays_a (n_rays,3) int64 with rows in ray order, some empty rays; returns (rays_a, N)."""
    g = rng(seed)
    counts = g.integers(1, max_len + 1, n_rays)
    counts[g.random(n_rays) < empty_fraction] = 0
    starts = np.cumsum(counts) - counts
    rays_a = np.stack([np.arange(n_rays), starts, counts], 1).astype(np.int64)
    return rays_a, int(counts.sum())


def borderline_rays(sigmas, deltas, rays_a, T_thr, rel=1e-4):
    """Rays whose running transmittance comes within `rel` of T_threshold: the stop decision of a
    parallel product scan may legitimately differ from the serial walk there, so parity tests
    compare those rays loosely ('early termination')."""
    bad = np.zeros(len(rays_a), bool)
    for i, (_, s, n) in enumerate(rays_a):
        if n == 0:
            continue
        a = 1.0 - np.exp(-sigmas[s:s + n].astype(np.float64) * deltas[s:s + n].astype(np.float64))
        T = np.cumprod(1.0 - a)
        bad[i] = np.any(np.abs(T - T_thr) <= rel * max(T_thr, 1e-30))
    return bad


def table_rule(n, amp=0.6):
    """the deterministic table of tests/golden/make_golden.py:table_rule (value_i = frac(i * 2654435761 / 2^32)
    - 0.5, times amp) — the G6 fixture was recorded with the hash tables filled this way"""
    i = np.arange(n, dtype=np.uint64)
    return (((i * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)).astype(np.float64) / 4294967296.0 - 0.5).astype(
        np.float32) * np.float32(amp)


def g6_state(g, tag, n_xyz, n_rgb):
    """state dict (numpy) of the model the G6 fixture was recorded with"""
    state = {"xyz_encoder.params": table_rule(n_xyz), "rgb_encoder.params": table_rule(n_rgb)}
    for k in ("xyz_net.0.weight", "xyz_net.0.bias", "xyz_net.2.weight", "xyz_net.2.bias", "rgb_net.params",
              "norm_pred_header.params", "semantic_header.params"):
        state[k] = g[f"{tag}_{k}"]
    return state


def noise_rule(shape, call_index):
    """tests/golden/make_golden.py:noise_rule — the jitter draws the G9 fixture was recorded with"""
    n = int(np.prod(shape))
    i = np.arange(1, n + 1, dtype=np.float64) + 7919.0 * call_index
    return np.mod(i * 0.6180339887498949, 1.0).astype(np.float32).reshape(shape)


# ---- dataset fixtures (G12): procedural images and cameras, written by the package's exporters ----
def dataset_inputs(n, h, w):
    """n RGBA images (n,h,w,4) uint8 from an integer rule, camera-to-world poses (n,3,4) float64 in
    [right down front] axes looking roughly at the origin from a wobbly ring, and a pinhole K"""
    i, r, c, ch = np.meshgrid(np.arange(n), np.arange(h), np.arange(w), np.arange(4), indexing="ij")
    img = (37 * i + 11 * r + 7 * c + 53 * ch + (r * c) % 17 * 5) % 256
    img[..., 3] = (r[..., 3] * 16 + c[..., 3] * 9 + i[..., 3] * 40) % 256
    poses = []
    for k in range(n):
        az, el, rad = 0.7 * k + 0.2, 0.35 + 0.15 * np.sin(1.3 * k), 2.0 + 0.4 * np.cos(0.9 * k)
        pos = rad * np.array([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)])
        target = 0.1 * np.array([np.sin(k), np.cos(2 * k), 0.3])
        fwd = (target - pos) / np.linalg.norm(target - pos)
        right = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
        right /= np.linalg.norm(right)
        poses.append(np.stack([right, np.cross(fwd, right), fwd, pos], 1))
    K = np.array([[1.1 * w, 0, w / 2 - 0.25], [0, 1.05 * w, h / 2 + 0.5], [0, 0, 1]])
    return img.astype(np.uint8), np.stack(poses), K


def write_dataset_dirs(root, export):
    """One directory per on-disk format under `root` (whose path must not contain the words the
    loaders branch on).  `export` = the package's datasets.export module.  Returns
    {name: (dataset key, directory, constructor kwargs)}."""
    import os
    out = {}
    img, c2w, K = dataset_inputs(10, 16, 16)
    angle = 2 * np.arctan(0.5 * 800 / 1111.0)  # NeRF-Synthetic: focal 1111 px at 800 px
    export.export_blender(os.path.join(root, "blender"), img, c2w, float(angle),
                          {"train": [0, 1, 2, 3, 4, 5], "val": [6, 7], "test": [8, 9]})
    out["nerf"] = ("nerf", os.path.join(root, "blender"), dict(downsample=0.02))

    img, c2w, K = dataset_inputs(19, 12, 16)
    pts = np.stack([np.sin(np.arange(40) * 0.37), np.cos(np.arange(40) * 0.61), np.sin(np.arange(40) * 0.13 + 1)], 1) * 0.8
    names = [f"view_{(7 * i) % 19:03d}.png" for i in range(19)]   # name order != record order
    labels = (img[..., 0] % 7).astype(np.uint8)
    export.export_colmap(os.path.join(root, "colmap_pinhole"), img[..., :3], c2w, K, names=names, points=pts,
                         model="PINHOLE", labels=labels, shuffle_seed=5)
    out["colmap"] = ("colmap", os.path.join(root, "colmap_pinhole"), dict(downsample=1.0, use_sem=True))
    export.export_colmap(os.path.join(root, "colmap_radial"), img, c2w, K, names=names, points=pts, model="SIMPLE_RADIAL")
    out["colmap_radial"] = ("colmap", os.path.join(root, "colmap_radial"), dict(downsample=1.0))

    img, c2w, K = dataset_inputs(9, 12, 16)
    split_of = [0, 0, 1, 0, 0, 1, 0, 0, 0]
    depth = (np.arange(9 * 12 * 16, dtype=np.float32).reshape(9, 12, 16) % 97) / 10
    path = dataset_inputs(5, 12, 16)[1][:, :, :] * np.array([1, 1, 1, 0.9])
    export.export_tnt(os.path.join(root, "tnt_scene"), img, c2w, K, split_of, img_dir="images", labels=(img[..., 1] % 5).astype(np.uint8),
                      depths=depth, camera_path=path, flat_intrinsics=True)
    out["tnt"] = ("tnt", os.path.join(root, "tnt_scene"), dict(downsample=1.0, use_sem=True, depth_mono=True))

    img, c2w, K = dataset_inputs(8, 16, 16)
    export.export_nsvf(os.path.join(root, "Synthetic_NSVF", "Wineholder"), img, c2w, K, [0, 0, 0, 0, 1, 1, 2, 2],
                       bbox=[-0.7, -0.6, -0.5, 0.9, 0.8, 0.7, 0.1])
    with open(os.path.join(root, "Synthetic_NSVF", "Wineholder", "intrinsics.txt"), "w") as f:
        f.write("1111.0 400.0 400.0 0.\n0. 0. 0.\n0.\n1.\n800 800\n")
    out["nsvf"] = ("nsvf", os.path.join(root, "Synthetic_NSVF", "Wineholder"), dict(downsample=0.02))
    img, c2w, K = dataset_inputs(6, 12, 16)
    K48 = K.copy()
    K48[:2] *= 48
    export.export_nsvf(os.path.join(root, "BlendedMVS", "Character"), img[..., :3], c2w, K48, [0, 0, 0, 1, 1, 1],
                       bbox=[-1.0, -1.1, -0.9, 1.2, 0.8, 1.0])
    np.savetxt(os.path.join(root, "BlendedMVS", "Character", "test_traj.txt"),
               np.concatenate([np.concatenate([p, [[0, 0, 0, 1.0]]], 0) for p in c2w[:3]], 0))
    out["nsvf_mvs"] = ("nsvf", os.path.join(root, "BlendedMVS", "Character"), dict(downsample=1 / 48))

    img, c2w, K = dataset_inputs(7, 12, 16)
    export.export_nerfpp(os.path.join(root, "nerfpp_scene"), img[..., :3], c2w, K, {"train": [0, 1, 2, 3], "val": [4], "test": [5, 6]})
    out["nerfpp"] = ("nerfpp", os.path.join(root, "nerfpp_scene"), dict(downsample=1.0))
    return out


def dataset_tmp_root():
    """a scratch directory whose path avoids the substrings the loaders branch on"""
    import tempfile
    while True:
        d = tempfile.mkdtemp(prefix="ngpds_", dir="/tmp")
        if not any(s in d for s in ("360", "HDR", "Synthetic", "Tanks", "Blended")):
            return d
