"""DTCWT filter preparation (reference pytorch_wavelets/dtcwt/lowlevel.py:58-67).  The 1-D filter
primitives of the reference (colfilter, coldfilt, colifilt, q2c, ...) have no standalone counterpart
here: they only exist fused inside the per-level kernels (csrc/wl_dtcwt_kernels.h)."""
import numpy as np
import torch


def prep_filt(h, c, transpose=False):
    """Column-vector filter, REVERSED, shape (c,1,L,1) (or (c,1,1,L) with transpose), default dtype."""
    h = np.asarray(h.detach().cpu().numpy() if isinstance(h, torch.Tensor) else h, dtype=np.float64)
    h = h.reshape(-1)[::-1].reshape(1, 1, -1, 1)
    h = np.repeat(h, repeats=c, axis=0)
    if transpose:
        h = h.transpose((0, 1, 3, 2))
    return torch.tensor(np.copy(h), dtype=torch.get_default_dtype())
