"""Image metrics with the reference's names and arguments (metrics.py:4-15)."""
import torch


def mse(image_pred, image_gt, valid_mask=None, reduction="mean"):
    """squared error, optionally restricted to `valid_mask`; reduction "mean" -> scalar, else per element"""
    err = torch.square(image_pred - image_gt)
    err = err if valid_mask is None else err[valid_mask]
    return err.mean() if reduction == "mean" else err


@torch.no_grad()
def psnr(image_pred, image_gt, valid_mask=None, reduction="mean"):
    """-10 log10(MSE) for images in [0, 1]"""
    return torch.log10(mse(image_pred, image_gt, valid_mask, reduction)).mul(-10.0)
