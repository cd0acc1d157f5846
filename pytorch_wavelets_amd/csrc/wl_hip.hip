// libwavelets_hip.so: gfx950 build of the kernels + the C ABI of include/wavelets_hip.h.
#include "wl_backend_hip.h"
#include "wl_api.inc"
