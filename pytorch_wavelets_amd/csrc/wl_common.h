// Common definitions for the gfx950 wavelet filterbank kernels.
//
// The kernel BODIES in this directory are plain C++ templates over a small execution context
// (WlCtx: thread id, block id, LDS pointer, barrier).  They are compiled
//   * by hipcc for gfx950 (wl_hip.hip) - the product, and
//   * by a host compiler into tests/emu/libwl_emu.so, where a fibre scheduler runs one workgroup
//     at a time with exact barrier semantics.  The emulator exists so the index arithmetic of the
//     kernels can be debugged in the GPU-less authoring container; it is test infrastructure and
//     is never loaded by the pytorch_wavelets_amd package.
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define WL_DEV __device__ __forceinline__
#define WL_HD __host__ __device__ __forceinline__
#else
#define WL_DEV inline __attribute__((always_inline))
#define WL_HD inline __attribute__((always_inline))
#endif

// dtype codes of the C ABI (include/wavelets_hip.h)
#define WL_F32 0
#define WL_F16 1
#define WL_F64 2

// boundary-extension codes used INSIDE kernels (the ABI takes the reference's mode ints)
#define WL_EXT_ZERO 0
#define WL_EXT_SYM 1       // half-sample symmetric
#define WL_EXT_REFL 2      // whole-sample reflect
#define WL_EXT_PERIODIC 3  // wrap
#define WL_EXT_PER 4       // periodization: odd length repeats the last sample, then wraps

typedef _Float16 wl_half;

template <typename T> struct WlAcc { typedef float type; };
template <> struct WlAcc<double> { typedef double type; };

// value of `v` in the previous lane of the wave (lane 0 gets its own).  The host emulation runs one fibre per
// lane to the next barrier, so it emulates the shuffle through a per-workgroup exchange array + two barriers.
#if defined(__HIPCC__)
WL_DEV float wl_shfl_up1(float v) { return __shfl_up(v, 1); }
WL_DEV float wl_shfl(float v, int src_lane) { return __shfl(v, src_lane, 64); }   // value of `v` in lane src_lane & 63
#else
float wl_shfl_up1(float v);
float wl_shfl(float v, int src_lane);
#endif

struct WlCtx {
    int tid;        // thread index in the workgroup
    int nthreads;   // workgroup size
    int64_t bid;    // linear workgroup index
    char* smem;     // dynamic LDS base (16-byte aligned)
#if defined(__HIPCC__)
    WL_DEV void sync() const { __syncthreads(); }
#else
    void (*sync_fn)(void*);
    void* sync_arg;
    inline void sync() const { sync_fn(sync_arg); }
#endif
};

// ---------------------------------------------------------------------------------------------
// Boundary extension: extended position i of a length-n signal -> source position, or -1 for a
// zero sample.  Closed forms of SURVEY.md §8 (reference: dwt/lowlevel.py:28-88, utils.py:146-174).
// ---------------------------------------------------------------------------------------------
WL_HD int wl_pmod(int i, int p) {
    int r = i % p;
    return r < 0 ? r + p : r;
}

WL_HD int wl_ext(int i, int n, int ext) {
    if ((unsigned)i < (unsigned)n) return i;
    switch (ext) {
        case WL_EXT_ZERO:
            return -1;
        case WL_EXT_SYM: {
            int j = wl_pmod(i, 2 * n);
            return j < n ? j : 2 * n - 1 - j;
        }
        case WL_EXT_REFL: {
            if (n == 1) return 0;
            int p = 2 * n - 2;
            int j = wl_pmod(i, p);
            return j < n ? j : p - j;
        }
        case WL_EXT_PERIODIC:
            return wl_pmod(i, n);
        default: {  // WL_EXT_PER
            int ne = n + (n & 1);
            int j = wl_pmod(i, ne);
            return j == n ? n - 1 : j;
        }
    }
}

// Replicate-padded view used by the DTCWT modules (dtcwt/transform2d.py:116-120, :131-135):
// a virtual signal of length n + pad_lo + pad_hi whose first pad_lo / last pad_hi samples
// replicate the edge.  Returns the source position of virtual position v (after extension).
WL_HD int wl_ext_padded(int v, int n, int pad_lo, int pad_hi, int ext) {
    if ((unsigned)(v - pad_lo) < (unsigned)n) return v - pad_lo;   // interior: no extension, no padding
    int nv = n + pad_lo + pad_hi;
    int j = wl_ext(v, nv, ext);
    if (j < 0) return -1;
    j -= pad_lo;
    return j < 0 ? 0 : (j >= n ? n - 1 : j);
}

WL_HD int wl_cdiv(int a, int b) { return (a + b - 1) / b; }
WL_HD int wl_align_up(int a, int b) { return (a + b - 1) / b * b; }

// XCD-aware block remap (MI355X: workgroup b is dispatched to XCD b % 8, each XCD has its own L2).  Returns the
// LOGICAL block a hardware block should work on so that every XCD walks one contiguous range of logical blocks:
// neighbouring tiles (which share halo rows / cache lines) then meet in the same L2.  Placement only affects speed.
WL_HD int64_t wl_xcd_remap(int64_t bid, int64_t nblocks) {
    if (nblocks <= 0) return bid;   // remap disabled
    const int64_t q = nblocks / 8, r = nblocks % 8;
    const int64_t x = bid % 8, i = bid / 8;
    return x * q + (x < r ? x : r) + i;
}
