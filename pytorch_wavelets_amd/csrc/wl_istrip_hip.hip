// libwavelets_hip.so: the one-level streaming SYNTHESIS strip kernels (wl_idwt_strip.h): the second part of wl_strip_api.inc.
#define WL_STRIP_PARTS 2
#include "wl_backend_hip.h"
#include "wl_strip_api.inc"
