// libwl_emu.so, unit 'strip': the analysis strip kernels of wl_strip_api.inc (HIP build: wl_strip_hip.hip), executed on the host.
#define WL_STRIP_PARTS 1
#include "wl_backend_emu.h"
#include "../../pytorch_wavelets_amd/csrc/wl_strip_api.inc"
