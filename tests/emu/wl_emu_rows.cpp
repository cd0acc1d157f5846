// libwl_emu.so, unit 'rows': the same kernel bodies and C ABI as the matching unit of libwavelets_hip.so, executed on the host.
#include "wl_backend_emu.h"
#include "../../pytorch_wavelets_amd/csrc/wl_rows_api.inc"
