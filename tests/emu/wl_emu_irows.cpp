// libwl_emu.so, unit 'irows': the synthesis half of wl_rows_api.inc (the matching unit of libwavelets_hip.so is wl_irows_hip.hip), executed on the host.
#define WL_ROWS_UNIT_SYNTHESIS 1
#include "wl_backend_emu.h"
#include "../../pytorch_wavelets_amd/csrc/wl_rows_api.inc"
