// libwl_emu.so, unit 'rows': the same kernel bodies and C ABI as the matching unit of libwavelets_hip.so, executed on the host.
#define WL_ROWS_UNIT_ANALYSIS 1   // (the synthesis half: wl_emu_irows.cpp)
#include "wl_backend_emu.h"
#include "../../pytorch_wavelets_amd/csrc/wl_rows_api.inc"
