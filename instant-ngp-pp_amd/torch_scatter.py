"""`torch_scatter`-shaped module: segment_csr with sum reduction, the only entry point the
reference uses (models/custom_functions.py:4,110-112)."""
import torch

from ._lib import call, check_input


def segment_csr(src, indptr, out=None, reduce="sum"):
    if reduce != "sum":
        raise NotImplementedError("segment_csr: only reduce='sum' is implemented")
    src = src.contiguous()
    indptr = indptr.contiguous()
    check_input(src, "src")
    check_input(indptr, "indptr")
    n_seg = indptr.shape[0] - 1
    width = 1
    for s in src.shape[1:]:
        width *= s
    res = torch.empty((n_seg,) + tuple(src.shape[1:]), dtype=torch.float32, device=src.device) if out is None else out
    call("segment_csr_sum", src, indptr, n_seg, width, res)
    return res
