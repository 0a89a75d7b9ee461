"""Image decoding for the loaders (reference: datasets/color_utils.py).

Deviation: files are decoded with PIL (imageio / cv2 are not available).  When the stored size
equals the requested size the result is bit-identical to the reference's; otherwise the resize is
PIL bilinear (labels: nearest), which differs from cv2.resize — parity unpinned for that case.
"""
import numpy as np


def srgb_to_linear(img):
    """inverse sRGB transfer curve (color_utils.py:7-9)"""
    return np.where(img > 0.04045, ((img + 0.055) / 1.055) ** 2.4, img / 12.92)


def linear_to_srgb(img):
    """sRGB transfer curve, clamped at 1 (color_utils.py:12-16)"""
    out = np.where(img > 0.0031308, 1.055 * img ** (1 / 2.4) - 0.055, 12.92 * img)
    return np.minimum(out, 1)


def _resized(arr, wh, resample):
    from PIL import Image
    if (arr.shape[1], arr.shape[0]) == tuple(wh):
        return arr
    if arr.ndim == 2:
        return np.asarray(Image.fromarray(arr).resize(tuple(wh), resample))
    return np.stack([np.asarray(Image.fromarray(np.ascontiguousarray(arr[..., c])).resize(tuple(wh), resample))
                     for c in range(arr.shape[2])], -1)


def _load(path):
    from PIL import Image
    im = Image.open(path)
    if im.mode in ("P", "PA", "LA", "CMYK", "1"):
        im = im.convert("RGBA" if "A" in im.mode or "transparency" in im.info else "RGB")
    return np.asarray(im)


def read_image(img_path, img_wh):
    """(h*w, 3) float32 in [0,1]; RGBA is blended on white (color_utils.py:19-28)"""
    from PIL import Image
    img = _load(img_path).astype(np.float32) / 255.0
    if img.ndim == 3 and img.shape[2] == 4:
        img = img[..., :3] * img[..., -1:] + (1 - img[..., -1:])
    img = _resized(img, img_wh, Image.BILINEAR)
    return np.ascontiguousarray(img.reshape(-1, img.shape[-1]), dtype=np.float32)


def read_normal_up(img_path, img_wh):
    """(h*w,) mask: 1 where the stored grey image is non-zero (color_utils.py:30-39)"""
    from PIL import Image
    img = _resized(_load(img_path).astype(np.float32) / 255.0, img_wh, Image.BILINEAR).reshape(-1).copy()
    img[img > 0] = 1
    return img


def read_normal(norm_path, norm_wh):
    """Monocular normal map -> (unit normals (h*w,3) with y and z flipped, mask of normals within
    60 degrees of +y) (color_utils.py:41-63)"""
    from PIL import Image
    n = _load(norm_path).astype(np.float32) / 255.0
    if n.shape[2] == 4:
        n = n[..., :3] * n[..., -1:] + (1 - n[..., -1:])
    n = _resized(n, norm_wh, Image.BILINEAR).reshape(-1, 3)
    n = (n + 1e-6) * 2. - 1.
    n[:, 1:] = -n[:, 1:]
    n = n / np.linalg.norm(n, ord=2, axis=-1, keepdims=True)
    cos_up = n[:, 1] / (np.linalg.norm(n, axis=-1) + 1e-6)
    return n, (cos_up > .5).astype(n.dtype)


def read_semantic(sem_path, sem_wh, classes=7):
    """(h*w,) integer labels from a .pgm (color_utils.py:65-71)"""
    from PIL import Image
    label = _load(sem_path)
    label = _resized(label, sem_wh, Image.NEAREST)
    return label.reshape(-1).astype(np.uint64)
