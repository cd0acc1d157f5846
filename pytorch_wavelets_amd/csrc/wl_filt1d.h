// Single-axis building blocks: one strided / dilated correlation with boundary extension along ANY axis of a dense
// tensor, and the 1-D two-channel synthesis bank.  They carry the parts of the reference's API that are not the fused
// 2-D hot path: DWT1DForward / DWT1DInverse (dwt/transform1d.py:7-115, AFB1D / SFB1D dwt/lowlevel.py:368-424, :697-743),
// the stationary transform's a-trous bank (afb1d_atrous, dwt/lowlevel.py:175-223), and the DTCWT primitives the
// reference exposes and tests on their own (colfilter ... rowifilt, dtcwt/lowlevel.py:70-239).
//
//   correlation:  y[o, q0 + qs*k, i] = sum_{t<nt} h[t0 + ts*t] * ext(x[o, :, i], start + step*k + dstep*t),  k in [0,K)
//                 x is (outer, n, inner) dense, y is (outer, ny, inner) dense; optionally a second tap set / output
//                 (the lo and hi channel of a filter bank share the samples).
//   synthesis:    full[m] = sum_k lo[k] g0[m-2k] + hi[k] g1[m-2k];  y[p] = full[p + L - 2]                (crop), or
//                 z[m] = full[m] + [m < L-2] full[m + 2K],  y[p] = z[(p + L/2 - 1) mod 2K]                (periodization)
// One thread per output sample, samples read straight from global memory (consecutive lanes read consecutive
// addresses along the fastest axis; the L-fold re-reads hit L1/L2).  These are API-completeness kernels, not the
// roofline path.
#pragma once
#include "wl_common.h"

#define WL_EXT_REPLICATE 5   // edge replication (only the stationary transform's 'replicate' padding uses it)
#define WL_EXT_PER_FOLD1 6   // periodization as the reference evaluates it (roll, zero-padded convolution, ONE fold of the
                             // wrapped tail: dwt/lowlevel.py:134-150) - differs from the circular form (WL_EXT_PER) when the
                             // signal is shorter than the filter.  step 2, tap_step 1, start = 1 - ntaps.

WL_HD int wl_ext_any(int i, int n, int ext) {
    if (ext == WL_EXT_REPLICATE) return i < 0 ? 0 : (i >= n ? n - 1 : i);
    return wl_ext(i, n, ext);
}

template <typename T>
struct WlCorr1dArgs {
    typedef typename WlAcc<T>::type A;
    const T* x; T* y0; T* y1;          // y1 may be nullptr
    const A* h0; const A* h1;
    int64_t outer, inner, y_outer_stride;   // y index = o * y_outer_stride + q * inner + i
    int n, K, nt, t0, ts, start, step, dstep, ext, q0, qs;
};

template <typename T>
struct WlCorr1d {
    typedef WlCorr1dArgs<T> Args;
    typedef typename WlAcc<T>::type A;
    static const int kThreads = 256;
    static const int kMinWaves = 1;
    static WL_DEV void run(const Args& a, const WlCtx& ctx) {
        const int64_t per = (int64_t)a.K * a.inner;
        const int64_t idx = ctx.bid * kThreads + ctx.tid;
        if (idx >= a.outer * per) return;
        int64_t o, i;
        int k;
        if (a.outer * per < (1LL << 31)) {       // (uniform) 32-bit index arithmetic: a 64-bit division costs ~40 instructions
            const unsigned u = (unsigned)idx, up = (unsigned)per, ui = (unsigned)a.inner;
            const unsigned uo = u / up, ur = u - uo * up;
            const unsigned uk = ui == 1 ? ur : ur / ui;
            o = uo; k = (int)uk; i = ur - uk * ui;
        } else {
            o = idx / per;
            const int64_t rem = idx - o * per;
            k = (int)(rem / a.inner);
            i = rem - (int64_t)k * a.inner;
        }
        const T* xp = a.x + o * a.n * a.inner + i;
        A acc0 = 0, acc1 = 0;
        const int p0 = a.start + a.step * k;
        if (a.ext != WL_EXT_PER_FOLD1 && a.dstep > 0 && p0 >= 0 && p0 + a.dstep * (a.nt - 1) < a.n) {
            // interior: every tap meets a sample of the signal (all but the first / last few outputs of a row): no extension
            // arithmetic per tap
            const T* xq = xp + (int64_t)p0 * a.inner;
            const int64_t st = (int64_t)a.dstep * a.inner;
            const A* h0 = a.h0 + a.t0;
            const A* h1 = a.h1 + a.t0;
            if (a.y1) {
                for (int t = 0; t < a.nt; ++t) {
                    const A v = (A)xq[t * st];
                    acc0 += h0[a.ts * t] * v;
                    acc1 += h1[a.ts * t] * v;
                }
            } else {
                for (int t = 0; t < a.nt; ++t) acc0 += h0[a.ts * t] * (A)xq[t * st];
            }
        } else if (a.ext == WL_EXT_PER_FOLD1) {
            const int ne = a.n + (a.n & 1), L2 = a.nt / 2;
            const int nfold = k < (L2 < ne / 2 ? L2 : ne / 2) ? 2 : 1;
            for (int f = 0; f < nfold; ++f)
                for (int t = 0; t < a.nt; ++t) {
                    const int s = wl_per_rolled_src(p0 + t + f * ne, a.n, L2);
                    if (s < 0) continue;
                    const A v = (A)xp[(int64_t)s * a.inner];
                    acc0 += a.h0[a.t0 + a.ts * t] * v;
                    if (a.y1) acc1 += a.h1[a.t0 + a.ts * t] * v;
                }
        } else
        for (int t = 0; t < a.nt; ++t) {
            const int s = wl_ext_any(p0 + a.dstep * t, a.n, a.ext);
            if (s < 0) continue;
            const A v = (A)xp[(int64_t)s * a.inner];
            acc0 += a.h0[a.t0 + a.ts * t] * v;
            if (a.y1) acc1 += a.h1[a.t0 + a.ts * t] * v;
        }
        const int64_t yo = o * a.y_outer_stride + (int64_t)(a.q0 + a.qs * k) * a.inner + i;
        a.y0[yo] = (T)acc0;
        if (a.y1) a.y1[yo] = (T)acc1;
    }
};

template <typename T>
struct WlSynth1dArgs {
    typedef typename WlAcc<T>::type A;
    const T* lo; const T* hi; T* y;    // lo / hi: (outer, K, inner); hi may be nullptr (zeros); y: (outer, ny, inner)
    const A* g0; const A* g1;
    int64_t outer, inner;
    int K, ny, L, circ;
};

template <typename T>
struct WlSynth1d {
    typedef WlSynth1dArgs<T> Args;
    typedef typename WlAcc<T>::type A;
    static const int kThreads = 256;
    static const int kMinWaves = 1;
    static WL_DEV A full(const Args& a, const T* lp, const T* hp, int m) {   // full[m] of the header
        A acc = 0;
        int k1 = m / 2;
        if (k1 > a.K - 1) k1 = a.K - 1;
        int k0 = (m - a.L + 2) / 2;          // ceil((m - L + 1) / 2) for m - L + 1 >= 0
        if (m - a.L + 1 < 0) k0 = 0;
        for (int k = k0; k <= k1; ++k) {
            const int t = m - 2 * k;
            if (t < 0 || t >= a.L) continue;
            acc += a.g0[t] * (A)lp[(int64_t)k * a.inner];
            if (hp) acc += a.g1[t] * (A)hp[(int64_t)k * a.inner];
        }
        return acc;
    }
    static WL_DEV void run(const Args& a, const WlCtx& ctx) {
        const int64_t per = (int64_t)a.ny * a.inner;
        const int64_t idx = ctx.bid * kThreads + ctx.tid;
        if (idx >= a.outer * per) return;
        int64_t o, i;
        int p;
        if (a.outer * per < (1LL << 31)) {       // (uniform) 32-bit index arithmetic
            const unsigned u = (unsigned)idx, up = (unsigned)per, ui = (unsigned)a.inner;
            const unsigned uo = u / up, ur = u - uo * up;
            const unsigned uq = ui == 1 ? ur : ur / ui;
            o = uo; p = (int)uq; i = ur - uq * ui;
        } else {
            o = idx / per;
            const int64_t rem = idx - o * per;
            p = (int)(rem / a.inner);
            i = rem - (int64_t)p * a.inner;
        }
        const T* lp = a.lo + o * a.K * a.inner + i;
        const T* hp = a.hi ? a.hi + o * a.K * a.inner + i : nullptr;
        A v;
        if (a.circ) {
            const int N = 2 * a.K;
            const int m = (p + (a.L / 2 - 1 < 2 * N ? a.L / 2 - 1 : 0)) % N;   // the reference's roll(): identity from 2N on
            v = full(a, lp, hp, m);
            if (m < a.L - 2) v += full(a, lp, hp, m + N);
        } else {
            v = full(a, lp, hp, p + a.L - 2);
        }
        a.y[idx] = (T)v;
    }
};
