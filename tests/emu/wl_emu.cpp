// libwl_emu.so: the same kernel bodies and C ABI as libwavelets_hip.so, executed on the host.
#include "wl_backend_emu.h"
#include "../../pytorch_wavelets_amd/csrc/wl_api.inc"
#include "../../pytorch_wavelets_amd/csrc/wl_rows_api.inc"
#include "../../pytorch_wavelets_amd/csrc/wl_strip_api.inc"
#include "../../pytorch_wavelets_amd/csrc/wl_dtinv_api.inc"
#include "../../pytorch_wavelets_amd/csrc/wl_dwt1d_api.inc"
