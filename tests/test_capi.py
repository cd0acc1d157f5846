"""The C-ABI library must load without a GPU and export every symbol include/wavelets_hip.h
declares (no compute calls here)."""
import os
import re

import __graft_entry__ as ge
from pytorch_wavelets_amd import _capi, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'wavelets_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(wl_[a-z0-9_]+)\s*\(', src)))


def test_library_builds_loads_and_exports_header_symbols():
    ge.build()
    lib = _lib.get()
    names = _declared()
    assert len(names) >= 5
    for n in names:
        assert hasattr(lib, n), 'missing export: ' + n
    assert sorted(_capi.PROTOTYPES) == names, 'ctypes prototypes out of sync with the header'
    assert lib.wl_backend() == b'hip-gfx950'
    assert lib.wl_dwt_coeff_len(512, 8, 1) == 259 and lib.wl_dwt_coeff_len(63, 6, 2) == 32
