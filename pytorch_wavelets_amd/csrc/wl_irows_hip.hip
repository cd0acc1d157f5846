// libwavelets_hip.so, translation unit of the streaming multi-level SYNTHESIS kernels (wl_idwt_rows.h): the second half of wl_rows_api.inc
// (built with -fno-slp-vectorize like the analysis half, wl_rows_hip.hip; the two compile side by side).
#define WL_ROWS_UNIT_SYNTHESIS 1
#include "wl_backend_hip.h"
#include "wl_rows_api.inc"
