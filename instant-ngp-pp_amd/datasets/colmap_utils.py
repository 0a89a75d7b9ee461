"""COLMAP sparse-model binary files: cameras.bin, images.bin, points3D.bin.

Same entry points and record attributes as the reader the reference vendors
(datasets/colmap_utils.py: read_cameras_binary / read_images_binary / read_points3d_binary returning
{id: record}); written here against the published file layout (all little endian):

  cameras.bin   u64 n | n x { i32 camera_id, i32 model_id, u64 width, u64 height, f64 params[P(model)] }
  images.bin    u64 n | n x { i32 image_id, f64 qvec[4] (w,x,y,z), f64 tvec[3], i32 camera_id,
                              char name[] NUL-terminated, u64 m, m x { f64 x, f64 y, i64 point3D_id } }
  points3D.bin  u64 n | n x { i64 point3D_id, f64 xyz[3], u8 rgb[3], f64 error,
                              u64 track, track x { i32 image_id, i32 point2D_idx } }

plus writers, used to export synthetic scenes in this format for the tests.
"""
import struct
from dataclasses import dataclass

import numpy as np

# model_id -> (name, number of parameters)
CAMERA_MODELS = {0: ("SIMPLE_PINHOLE", 3), 1: ("PINHOLE", 4), 2: ("SIMPLE_RADIAL", 4), 3: ("RADIAL", 5),
                 4: ("OPENCV", 8), 5: ("OPENCV_FISHEYE", 8), 6: ("FULL_OPENCV", 12), 7: ("FOV", 5),
                 8: ("SIMPLE_RADIAL_FISHEYE", 4), 9: ("RADIAL_FISHEYE", 5), 10: ("THIN_PRISM_FISHEYE", 12)}
CAMERA_MODEL_IDS = {name: mid for mid, (name, _) in CAMERA_MODELS.items()}


def qvec2rotmat(q):
    """rotation matrix of a unit quaternion (w, x, y, z)"""
    w, x, y, z = q
    return np.array([[1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * z * x + 2 * w * y],
                     [2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x],
                     [2 * z * x - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y]])


def rotmat2qvec(R):
    """unit quaternion (w, x, y, z), w >= 0, of a rotation matrix: the dominant eigenvector of the
    symmetric 4x4 matrix built from R (robust for every rotation)"""
    (xx, yx, zx), (xy, yy, zy), (xz, yz, zz) = np.asarray(R, dtype=np.float64)
    Kmat = np.array([[xx - yy - zz, 0, 0, 0],
                     [yx + xy, yy - xx - zz, 0, 0],
                     [zx + xz, zy + yz, zz - xx - yy, 0],
                     [yz - zy, zx - xz, xy - yx, xx + yy + zz]]) / 3.0
    vals, vecs = np.linalg.eigh(Kmat)
    q = vecs[[3, 0, 1, 2], np.argmax(vals)]
    return -q if q[0] < 0 else q


@dataclass
class Camera:
    id: int
    model: str
    width: int
    height: int
    params: np.ndarray


@dataclass
class Image:
    id: int
    qvec: np.ndarray
    tvec: np.ndarray
    camera_id: int
    name: str
    xys: np.ndarray
    point3D_ids: np.ndarray

    def qvec2rotmat(self):
        return qvec2rotmat(self.qvec)


@dataclass
class Point3D:
    id: int
    xyz: np.ndarray
    rgb: np.ndarray
    error: float
    image_ids: np.ndarray
    point2D_idxs: np.ndarray


class _Cursor:
    def __init__(self, path):
        with open(path, "rb") as f:
            self.buf = f.read()
        self.pos = 0

    def take(self, fmt):
        vals = struct.unpack_from("<" + fmt, self.buf, self.pos)
        self.pos += struct.calcsize("<" + fmt)
        return vals

    def array(self, dtype, count):
        a = np.frombuffer(self.buf, dtype=dtype, count=count, offset=self.pos)
        self.pos += a.nbytes
        return a

    def cstring(self):
        end = self.buf.index(b"\x00", self.pos)
        s = self.buf[self.pos:end].decode("utf-8")
        self.pos = end + 1
        return s


def read_cameras_binary(path_to_model_file):
    c = _Cursor(path_to_model_file)
    cameras = {}
    for _ in range(c.take("Q")[0]):
        cam_id, model_id, width, height = c.take("iiQQ")
        name, n_params = CAMERA_MODELS[model_id]
        cameras[cam_id] = Camera(cam_id, name, width, height, c.array("<f8", n_params).copy())
    return cameras


def read_images_binary(path_to_model_file):
    c = _Cursor(path_to_model_file)
    images = {}
    for _ in range(c.take("Q")[0]):
        image_id = c.take("i")[0]
        qvec = c.array("<f8", 4).copy()
        tvec = c.array("<f8", 3).copy()
        camera_id = c.take("i")[0]
        name = c.cstring()
        n_obs = c.take("Q")[0]
        obs = c.array(np.dtype([("xy", "<f8", 2), ("pid", "<i8")]), n_obs)
        images[image_id] = Image(image_id, qvec, tvec, camera_id, name, obs["xy"].copy(), obs["pid"].copy())
    return images


def read_points3d_binary(path_to_model_file):
    c = _Cursor(path_to_model_file)
    points = {}
    for _ in range(c.take("Q")[0]):
        pid = c.take("q")[0]
        xyz = c.array("<f8", 3).copy()
        rgb = c.array("u1", 3).copy()
        error = c.take("d")[0]
        track = c.array("<i4", 2 * c.take("Q")[0]).reshape(-1, 2)
        points[pid] = Point3D(pid, xyz, rgb, error, track[:, 0].copy(), track[:, 1].copy())
    return points


def write_cameras_binary(cameras, path):
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(cameras)))
        for cam in cameras.values():
            n_params = CAMERA_MODELS[CAMERA_MODEL_IDS[cam.model]][1]
            assert len(cam.params) == n_params
            f.write(struct.pack("<iiQQ", cam.id, CAMERA_MODEL_IDS[cam.model], cam.width, cam.height))
            f.write(np.asarray(cam.params, dtype="<f8").tobytes())


def write_images_binary(images, path):
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(images)))
        for im in images.values():
            f.write(struct.pack("<i", im.id))
            f.write(np.asarray(im.qvec, dtype="<f8").tobytes())
            f.write(np.asarray(im.tvec, dtype="<f8").tobytes())
            f.write(struct.pack("<i", im.camera_id))
            f.write(im.name.encode("utf-8") + b"\x00")
            f.write(struct.pack("<Q", len(im.point3D_ids)))
            obs = np.zeros(len(im.point3D_ids), dtype=np.dtype([("xy", "<f8", 2), ("pid", "<i8")]))
            obs["xy"], obs["pid"] = np.asarray(im.xys).reshape(-1, 2), im.point3D_ids
            f.write(obs.tobytes())


def write_points3d_binary(points, path):
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(points)))
        for p in points.values():
            f.write(struct.pack("<q", p.id))
            f.write(np.asarray(p.xyz, dtype="<f8").tobytes())
            f.write(np.asarray(p.rgb, dtype="u1").tobytes())
            f.write(struct.pack("<dQ", p.error, len(p.image_ids)))
            f.write(np.stack([p.image_ids, p.point2D_idxs], 1).astype("<i4").tobytes())
