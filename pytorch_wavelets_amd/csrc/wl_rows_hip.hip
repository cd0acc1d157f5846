// libwavelets_hip.so, second translation unit: the streaming multi-level analysis and synthesis kernels (built with
// -fno-slp-vectorize, see wl_rows_api.inc).
#include "wl_backend_hip.h"
#include "wl_rows_api.inc"
