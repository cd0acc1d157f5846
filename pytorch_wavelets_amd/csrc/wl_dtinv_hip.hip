// libwavelets_hip.so, fourth translation unit: the fused DTCWT inverse (levels 2 + 1 in one launch).
#include "wl_backend_hip.h"
#include "wl_dtinv_api.inc"
