"""Argument plumbing shared by the reference-named wrappers: accept NumPy arrays or torch tensors
(the reference passes NumPy at these seams, evaluation.py:306-309) and hand device tensors to the
C ABI.  There is no CPU compute path: a HIP device is required."""
import numpy as np
import torch

from .. import _lib


def to_dev(a, dtype=torch.float32):
    _lib.require_gpu()
    if isinstance(a, torch.Tensor):
        return a.detach().to(device="cuda", dtype=dtype).contiguous()
    return torch.from_numpy(np.ascontiguousarray(a)).to(device="cuda", dtype=dtype).contiguous()
