"""Test-only: load the HOST EMULATION build of the kernel sources (tests/emu/libwl_emu.so) and
install it as the ops backend, so the Python layer and the kernels' index arithmetic can be
exercised in the GPU-less container.  Never used by the product or by the -m gpu tests."""
import contextlib
import ctypes
import os
import subprocess

from pytorch_wavelets_amd import _capi, ops

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emu')
_SO = os.path.join(_DIR, 'libwl_emu.so')
_H = None


def handle():
    global _H
    if _H is None:
        srcs = [os.path.join(_DIR, f) for f in os.listdir(_DIR) if f.endswith(('.cpp', '.h', '.sh'))]
        csrc = os.path.join(os.path.dirname(_DIR), '..', 'pytorch_wavelets_amd', 'csrc')
        srcs += [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(('.h', '.inc'))]
        if not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
            subprocess.check_call([os.path.join(_DIR, 'build.sh')])
        _H = _capi.bind(ctypes.CDLL(_SO))
        assert _H.wl_backend() == b'emu'
    return _H


@contextlib.contextmanager
def emulated():
    prev = ops._TEST_BACKEND
    ops._TEST_BACKEND = handle()
    try:
        yield
    finally:
        ops._TEST_BACKEND = prev


@contextlib.contextmanager
def chip_of(cus):
    """The emulated chip with `cus` compute units (default 2) - for launcher policies that depend on the chip's size."""
    handle()
    lib = ctypes.CDLL(_SO)
    lib.wl_emu_set_cus(int(cus))
    try:
        yield
    finally:
        lib.wl_emu_set_cus(2)
