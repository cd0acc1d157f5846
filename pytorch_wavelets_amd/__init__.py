"""pytorch_wavelets_amd - MI355X-native (gfx950) 2-D wavelet filterbank engine.

Drop-in for the DWT / DTCWT hot path of fbcotter/pytorch_wavelets: the same nn.Module API and
(yl, yh) tensor layout, computed by hand-written HIP kernels behind a C ABI
(include/wavelets_hip.h).  There is no CPU path: tensors must live on a cuda (ROCm) device.
"""
__version__ = '0.1.0'

from .dwt.transform2d import DWTForward, DWTInverse   # noqa: E402,F401
from .dwt.transform1d import DWT1DForward, DWT1DInverse   # noqa: E402,F401
from .dtcwt.transform2d import DTCWTForward, DTCWTInverse   # noqa: E402,F401
from .scatternet import ScatLayer, ScatLayerj2   # noqa: E402,F401

from . import parallel                                  # noqa: E402,F401


def last_kernel():
    """Name of the kernel functor the engine dispatched last on this thread (wl_last_kernel of the C ABI): what actually
    ran for the last forward / inverse / backward, after every dispatch decision - bench.py and the tests quote it."""
    from . import ops
    raw = ops._backend().wl_last_kernel().decode()
    return raw.split('K = ')[-1].rstrip(']') if 'K = ' in raw else raw


def launch_count():
    """Kernels the engine has launched in this process so far (wl_launch_count of the C ABI)."""
    from . import ops
    return int(ops._backend().wl_launch_count())


def kernels_since(count):
    """The kernel functors launched since ``count = launch_count()`` was read, oldest first (the engine remembers the last
    32): every launch of a multi-launch transform by name."""
    from . import ops
    be = ops._backend()
    n = min(int(be.wl_launch_count()) - count, 32)
    out = []
    for back in range(n - 1, -1, -1):
        raw = be.wl_kernel_history(back).decode()
        name = raw.split('K = ')[-1].rstrip(']') if 'K = ' in raw else raw
        # the two-bank variant queued behind a variant that relies on a relation between the filter banks: it returns at once
        # unless the device finds the relation broken (csrc/wl_common.h, tap-relation guards)
        # ' (aux)': a helper launch in front of the chosen kernel (WlTapPrep: one thread that examines the filter banks on the device)
        out.append(name + ' (armed fallback)' if 'wl_launch_armed' in raw else name + ' (aux)' if 'wl_launch_aux' in raw else name)
    return out


DTCWT = DTCWTForward
IDTCWT = DTCWTInverse
DWT = DWTForward
IDWT = DWTInverse
DWT2D = DWT
IDWT2D = IDWT
DWT1D = DWT1DForward
IDWT1D = DWT1DInverse

__all__ = ['__version__', 'last_kernel', 'launch_count', 'kernels_since', 'DTCWTForward', 'DTCWTInverse', 'DWTForward', 'DWTInverse', 'DTCWT', 'IDTCWT',
           'DWT', 'IDWT', 'DWT2D', 'IDWT2D', 'DWT1DForward', 'DWT1DInverse', 'DWT1D', 'IDWT1D', 'ScatLayer', 'ScatLayerj2']
