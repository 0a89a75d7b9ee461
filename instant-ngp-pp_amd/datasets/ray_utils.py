"""Camera-ray geometry on the input side of the hot path (reference: datasets/ray_utils.py).

Conventions: camera space is [right down front]; a pose is a (3,4) camera-to-world matrix
[R | t]; pixel (column u, row v) looks along ((u-cx+.5)/fx, (v-cy+.5)/fy, 1).
"""
import numpy as np
import torch


def get_ray_directions(H, W, K, device='cpu', random=False, return_uv=False, flatten=True, anti_aliasing_factor=1.0):
    """Camera-space directions of all pixels (ray_utils.py:8-47).  `random` jitters inside the
    pixel; `anti_aliasing_factor` > 1 renders on a finer lattice and — as the reference does —
    scales the caller's K IN PLACE (K[2,2] is reset to 1)."""
    if anti_aliasing_factor > 1.0:
        H, W = int(H * anti_aliasing_factor), int(W * anti_aliasing_factor)
        K *= anti_aliasing_factor
        K[2, 2] = 1
    v, u = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=device),
                          torch.arange(W, dtype=torch.float32, device=device), indexing="ij")
    fx, fy, cx, cy = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])
    if random:
        ju, jv = torch.rand_like(u), torch.rand_like(v)   # u first, then v: the reference's draw order
    else:
        ju = jv = 0.5
    directions = torch.stack([(u - cx + ju) / fx, (v - cy + jv) / fy, torch.ones_like(u)], -1)
    grid = torch.stack([u, v], -1)
    if flatten:
        directions, grid = directions.reshape(-1, 3), grid.reshape(-1, 2)
    return (directions, grid) if return_uv else directions


def get_rays(directions, c2w):
    """World-space origins and (unnormalised) directions (ray_utils.py:50-74).
    directions (N,3); c2w (3,4) or (N,3,4)."""
    if c2w.ndim == 2:
        rays_d = directions @ c2w[:, :3].T
    else:
        rays_d = (directions[:, None, :] @ c2w[..., :3].transpose(1, 2))[:, 0]
    rays_o = c2w[..., 3].expand_as(rays_d)
    return rays_o, rays_d


def axisangle_to_R(v):
    """Rodrigues' formula for a batch of axis-angle vectors (ray_utils.py:78-104), used by the
    pose-refinement option: R = I + sin|v|/|v| [v]x + (1-cos|v|)/|v|^2 [v]x^2, |v| padded by 1e-7."""
    single = v.ndim == 1
    v = v.reshape(-1, 3)
    x, y, z = v.unbind(-1)
    o = torch.zeros_like(x)
    skew = torch.stack([o, -z, y, z, o, -x, -y, x, o], -1).reshape(-1, 3, 3)
    n = (v.norm(dim=1) + 1e-7)[:, None, None]
    R = torch.eye(3, device=v.device, dtype=v.dtype) + torch.sin(n) / n * skew + (1 - torch.cos(n)) / n ** 2 * (skew @ skew)
    return R[0] if single else R


def normalize(v):
    return v / np.linalg.norm(v)


def average_poses(poses, pts3d=None):
    """The 'mean camera' (ray_utils.py:112-156): origin = centroid of the point cloud (or of the
    camera centres), z = mean viewing axis, x = normalize(mean-y × z), y = z × x."""
    origin = pts3d.mean(0) if pts3d is not None else poses[..., 3].mean(0)
    z = normalize(poses[..., 2].mean(0))
    x = normalize(np.cross(poses[..., 1].mean(0), z))
    return np.stack([x, np.cross(z, x), z, origin], 1)


def _homogeneous(m34):
    out = np.zeros(m34.shape[:-2] + (4, 4), dtype=m34.dtype)
    out[..., :3, :] = m34
    out[..., 3, 3] = 1
    return out


def center_poses(poses, pts3d=None):
    """Expresses poses (and points) in the frame of `average_poses` (ray_utils.py:159-188)."""
    to_avg = np.linalg.inv(_homogeneous(average_poses(poses, pts3d)))
    centred = (to_avg @ _homogeneous(poses))[:, :3]
    if pts3d is None:
        return centred
    return centred, pts3d @ to_avg[:, :3].T + to_avg[:, 3:].T


def create_spheric_poses(radius, mean_h, n_poses=120):
    """Circular fly-around (ray_utils.py:190-225): n_poses azimuths over a full turn, camera tilted
    by -pi/12, pushed back by `radius`, lifted by 2*mean_h, then re-expressed with axes
    (-x, z, y).  Rows are built as R_y(theta) R_x(phi) [I | (0, 2 mean_h, -radius)]."""
    phi = -np.pi / 12
    cp, sp = np.cos(phi), np.sin(phi)
    rot_phi = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    shift = np.array([[1, 0, 0, 0], [0, 1, 0, 2 * mean_h], [0, 0, 1, -radius]])
    swap = np.array([[-1, 0, 0], [0, 0, 1], [0, 1, 0]])
    out = []
    for th in np.linspace(0, 2 * np.pi, n_poses + 1)[:-1]:
        c, s = np.cos(th), np.sin(th)
        rot_theta = np.array([[c, 0, -s], [0, 1, 0], [s, 0, c]])
        out.append(swap @ (rot_theta @ rot_phi @ shift))
    return np.stack(out, 0)


def viewmatrix(lookdir, up, position):
    """look-at pose with columns (x, y, z, position), z along `lookdir` (ray_utils.py:227-234)"""
    z = normalize(lookdir)
    x = normalize(np.cross(up, z))
    y = normalize(np.cross(z, x))
    return np.stack([x, y, z, position], axis=1)


def generate_interpolated_path(poses, n_interp, spline_degree=5, smoothness=.03, rot_weight=.1):
    """Smooth camera path through key poses (ray_utils.py:236-277, after multinerf): every pose
    becomes three points (centre, centre - w·z, centre + w·y), one B-spline of degree
    min(spline_degree, n-1) with smoothing `smoothness` is fitted through the 9-D points
    (scipy.interpolate.splprep) and sampled at n_interp*(n-1) parameters in [0,1); poses are
    rebuilt by look-at.  Returns (n_interp*(n-1), 3, 4)."""
    import scipy.interpolate
    centre = poses[:, :3, -1]
    pts = np.stack([centre, centre - rot_weight * poses[:, :3, 2], centre + rot_weight * poses[:, :3, 1]], 1)
    n_key = pts.shape[0]
    n_out = n_interp * (n_key - 1)
    tck, _ = scipy.interpolate.splprep(pts.reshape(n_key, -1).T, k=min(spline_degree, n_key - 1), s=smoothness)
    sampled = np.array(scipy.interpolate.splev(np.linspace(0, 1, n_out, endpoint=False), tck)).T.reshape(n_out, 3, 3)
    return np.array([viewmatrix(p - look, up - p, p) for p, look, up in sampled])
