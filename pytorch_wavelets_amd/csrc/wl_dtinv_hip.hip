// libwavelets_hip.so, fourth translation unit: the fused DTCWT inverse (levels 2 + 1 in one launch) and the fused multi-level
// 1-D analysis.
#include "wl_backend_hip.h"
#include "wl_dtinv_api.inc"
#include "wl_dwt1d_api.inc"
