// libwavelets_hip.so: the one-level streaming ANALYSIS strip kernels (wl_dwt_strip.h; built with -fno-slp-vectorize, like the other
// streaming kernels).  The synthesis strips are wl_istrip_hip.hip, the streaming DTCWT / ScatLayer kernels wl_dtstrip_hip.hip.
#define WL_STRIP_PARTS 1
#include "wl_backend_hip.h"
#include "wl_strip_api.inc"
