// libwl_emu.so, unit 'istrip': the synthesis strip kernels of wl_strip_api.inc (HIP build: wl_istrip_hip.hip), executed on the host.
#define WL_STRIP_PARTS 2
#include "wl_backend_emu.h"
#include "../../pytorch_wavelets_amd/csrc/wl_strip_api.inc"
