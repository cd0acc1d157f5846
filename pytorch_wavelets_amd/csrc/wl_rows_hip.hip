// libwavelets_hip.so, second translation unit: the streaming multi-level analysis kernels (built with
// -fno-slp-vectorize, see wl_rows_api.inc).
#define WL_ROWS_UNIT_ANALYSIS 1   // (the synthesis half of wl_rows_api.inc is wl_irows_hip.hip)
#include "wl_backend_hip.h"
#include "wl_rows_api.inc"
