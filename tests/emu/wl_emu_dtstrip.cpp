// libwl_emu.so, unit 'dtstrip': the streaming DTCWT / ScatLayer kernels of wl_strip_api.inc (HIP build: wl_dtstrip_hip.hip), executed on the host.
#define WL_STRIP_PARTS 4
#include "wl_backend_emu.h"
#include "../../pytorch_wavelets_amd/csrc/wl_strip_api.inc"
