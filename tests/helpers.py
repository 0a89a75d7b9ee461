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
