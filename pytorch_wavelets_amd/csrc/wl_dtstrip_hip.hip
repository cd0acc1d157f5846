// libwavelets_hip.so: the streaming DTCWT / ScatLayer kernels over column strips (wl_dtcwt_strip.h, wl_dtcwt_fused.h): the third part of
// wl_strip_api.inc.
#define WL_STRIP_PARTS 4
#include "wl_backend_hip.h"
#include "wl_strip_api.inc"
