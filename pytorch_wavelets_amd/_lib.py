"""Loader for the gfx950 engine, libwavelets_hip.so (built in-tree by __graft_entry__.build()).

There is no CPU or PyTorch fallback: if the library is missing every operator raises.
"""
import ctypes
import os

from . import _capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('WL_LIB') or os.path.join(_HERE, 'csrc', 'libwavelets_hip.so')   # WL_LIB: A/B builds
_LIB = None


class EngineUnavailable(RuntimeError):
    pass


def get():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise EngineUnavailable(
                'pytorch_wavelets_amd: %s not found. Build it with `python -c "import __graft_entry__ as g; '
                'g.build()"` (hipcc --offload-arch=gfx950). This engine has no CPU/PyTorch fallback.'
                % LIB_PATH)
        _LIB = _capi.bind(ctypes.CDLL(LIB_PATH))
    return _LIB


def check(rc, what):
    if rc == 0:
        return
    names = {-1: 'WL_ERR_MODE', -2: 'WL_ERR_SHAPE', -3: 'WL_ERR_UNSUPPORTED', -4: 'WL_ERR_DTYPE',
             -5: 'WL_ERR_TAPS'}
    if rc == -3:
        raise NotImplementedError('%s: configuration not supported by the gfx950 engine '
                                  '(WL_ERR_UNSUPPORTED)' % what)
    raise RuntimeError('%s failed: %s' % (what, names.get(rc, 'hipError_t %d' % rc)))
