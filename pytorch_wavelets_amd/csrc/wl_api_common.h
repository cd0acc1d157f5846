// Host-side helpers shared by the translation units of the C ABI (wl_api.inc, wl_rows_api.inc).
#pragma once
#include <stdlib.h>
#include "../../include/wavelets_hip.h"
#include "wl_common.h"

// ---- helpers ---------------------------------------------------------------------------------
static int wl_mode_to_ext(int mode) {
    switch (mode) {
        case 0: return WL_EXT_ZERO;
        case 1: return WL_EXT_SYM;
        case 2: return WL_EXT_PER;
        case 4: return WL_EXT_REFL;
        case 6: return WL_EXT_PERIODIC;
        default: return -1;
    }
}

static inline int wl_coeff_len(int n, int L, int mode) { return mode == 2 ? (n + 1) / 2 : (n + L - 1) / 2; }

// base of  y[k] = sum_j h[j] * ext(x, 2k + base + j)   (stored taps h; see wl_dwt_kernels.h)
static int wl_afb_base(int n, int L, int mode) {
    if (mode == 2) return L / 2 - L + 1;
    const int K = (n + L - 1) / 2;
    const int p = 2 * (K - 1) - n + L;
    return -(p / 2);
}

// A launch of the variant HINT that relies on a relation between the filter banks, guarded on the device (wl_common.h), with
// the plain variant PLAIN queued behind it as its armed fallback: exactly one of the two does the work.  The fallback's own
// checks run first (dry), so that it cannot decline behind a hinted variant that is already on the stream.
#define WL_GUARDED_PAIR(HINT, PLAIN, ...)                         \
    do {                                                          \
        int rc_ = PLAIN(__VA_ARGS__, 2, 1);                       \
        if (rc_ != 0) return rc_;                                 \
        rc_ = HINT(__VA_ARGS__, 1, 0);                            \
        if (rc_ != 0) return rc_;                                 \
        return PLAIN(__VA_ARGS__, 2, 2);                          \
    } while (0)
