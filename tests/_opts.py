"""Test helper: switch the engine between the specialised and the generic kernel families through wl_set_option
(the product reads no environment variable on the launch path).  On a GPU box this addresses the real library, in
the GPU-less container the host emulation (and whichever of the two is already loaded)."""
import torch


def set_generic(flag):
    import emu_backend
    from pytorch_wavelets_amd import _lib
    v = 1 if str(flag) not in ('0', 'False', '') else 0
    if torch.cuda.is_available() or _lib._LIB is not None:
        _lib.get().wl_set_option(b'generic_only', v)
    if not torch.cuda.is_available() or emu_backend._H is not None:
        emu_backend.handle().wl_set_option(b'generic_only', v)
