// libwavelets_hip.so, third translation unit: the one-level streaming analysis over column strips (built with
// -fno-slp-vectorize, like the other streaming kernels).
#include "wl_backend_hip.h"
#include "wl_strip_api.inc"
