// One level of the 2-D stationary (undecimated, "a trous") analysis bank in ONE launch: afb2d_atrous of the reference
// (dwt/lowlevel.py:475-521 = afb1d_atrous :175-223 along W, then along H; SWTForward.forward's level, dwt/transform2d.py:
// 186-212).  The taps are dilated by d = 2^j, the signal is extended by (L d) / 2 - d samples before and (L d) / 2 after with
// the pad mode, the output keeps the input size:
//     lohi_r[i][k] = sum_t hw_r[t] * X(i, k - ((Lw d) / 2 - d) + d t)                   r = 0 (low), 1 (high) along W
//     y[4c + 2r + b][i][k] = sum_t hh_b[t] * LOHI_r(i - ((Lh d) / 2 - d) + d t, k)      b = 0 (low), 1 (high) along H
// (stored taps = the reference's conv2d weights: already reversed; X / LOHI = the plane under the extension rule).
// On the single-axis kernels (wl_corr1d, the round-2 path) a level is two launches plus the two stack().reshape() copies of
// the reference's formulation and moves ~21 plane-sizes through memory for the 5 it has to (x in, four sub-bands out).
// Here a workgroup owns a TH x 64 tile of the output: the extended input tile goes to LDS once, the row-filtered (lo, hi)
// pairs of its TH + (Lh - 1) d rows to LDS, the column filter reads them back; the four sub-bands are written straight
// into the (N, 4C, H, W) layout the reference returns.  The input may be every 4th channel of a previous level's output
// (plane stride x_ps): no copy of the ll channels between levels.
#pragma once
#include "wl_common.h"
#include "wl_filt1d.h"   // wl_ext_any: the extension rules incl. 'replicate'

template <typename T>
struct WlSwtArgs {
    typedef typename WlAcc<T>::type A;
    const T* x;                    // (planes, H, W) through x_ps (rows dense)
    T* y;                          // (planes, 4, H, W) dense
    const A* hw0; const A* hw1;    // taps along W (low, high), Lw each
    const A* hh0; const A* hh1;    // taps along H, Lh each
    int64_t planes, x_ps, nblocks;
    int H, W, Lw, Lh, d, ext;
    int sw, sh;                    // first extended column / row a correlation reads relative to its output: -((L d)/2 - d)
    int TH, tiles_x, tiles_y;
    int in_pitch;                  // elements per row of the staged input tile
    int mid_off;                   // byte offset of the (lo, hi) rows in LDS
    int lds_bytes;
};

// LT = compile-time tap count of both axes (loops unrolled, taps in registers), 0 = any (Lw, Lh) at run time
template <typename T, int LT>
struct WlSwtLevel {
    typedef WlSwtArgs<T> Args;
    typedef typename WlAcc<T>::type A;
    static const int kThreads = 256;
    static const int kMinWaves = 2;
    static const int TW = 64;
    struct Pair { A lo, hi; };
    static WL_DEV void run(const Args& a, const WlCtx& ctx) {
        const int tx = ctx.tid & 63, ty = ctx.tid >> 6;
        const int per_plane = a.tiles_x * a.tiles_y;
        const int64_t plane = ctx.bid / per_plane;
        const int rem = (int)(ctx.bid - plane * per_plane);
        const int tyi = rem / a.tiles_x, txi = rem - tyi * a.tiles_x;
        const int r0 = tyi * a.TH, c0 = txi * TW;
        const int Lw = LT ? LT : a.Lw, Lh = LT ? LT : a.Lh, d = a.d;
        const int hw = (Lw - 1) * d, hh = (Lh - 1) * d;
        const int th = r0 + a.TH <= a.H ? a.TH : a.H - r0;         // output rows / columns of this tile
        const int tw = c0 + TW <= a.W ? TW : a.W - c0;
        const int rows = th + hh, cols = tw + hw;
        A* const in = reinterpret_cast<A*>(ctx.smem);
        Pair* const mid = reinterpret_cast<Pair*>(ctx.smem + a.mid_off);
        const T* const xp = a.x + (size_t)plane * a.x_ps;
        // ---- the extended input tile: rows r0 + sh .., columns c0 + sw ..
        // (eight rows of a column at a time: eight independent loads in flight per thread, then the eight LDS writes - one
        // load per loop iteration waits for memory once per element)
        for (int c = tx; c < cols; c += 64) {
            const int gc = wl_ext_any(c0 + a.sw + c, a.W, a.ext);
            for (int rb = ty; rb < rows; rb += 32) {
                T v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int r = rb + 4 * u;
                    const int gr = r < rows ? wl_ext_any(r0 + a.sh + r, a.H, a.ext) : -1;
                    v[u] = xp[(gr < 0 || gc < 0) ? 0 : (size_t)gr * a.W + gc];
                    if (gr < 0 || gc < 0) v[u] = (T)0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int r = rb + 4 * u;
                    if (r < rows) in[r * a.in_pitch + c] = (A)v[u];
                }
            }
        }
        ctx.sync();
        // ---- row filter: (lo, hi) of every staged row at the tile's output columns
        if (tx < tw) {
            A w0[LT ? LT : 1], w1[LT ? LT : 1];
            if (LT) {
#pragma unroll
                for (int t = 0; t < (LT ? LT : 1); ++t) { w0[t] = a.hw0[t]; w1[t] = a.hw1[t]; }
            }
            for (int r = ty; r < rows; r += 4) {
                const A* p = in + r * a.in_pitch + tx;
                A lo = 0, hi = 0;
                if (LT) {
#pragma unroll
                    for (int t = 0; t < (LT ? LT : 1); ++t) { const A s = p[t * d]; lo += w0[t] * s; hi += w1[t] * s; }
                } else {
                    for (int t = 0; t < Lw; ++t) { const A s = p[t * d]; lo += a.hw0[t] * s; hi += a.hw1[t] * s; }
                }
                Pair q; q.lo = lo; q.hi = hi;
                mid[r * TW + tx] = q;
            }
        }
        ctx.sync();
        // ---- column filter and the four sub-bands: channel 4c + 2 (band along W) + (band along H)
        if (tx < tw) {
            A v0[LT ? LT : 1], v1[LT ? LT : 1];
            if (LT) {
#pragma unroll
                for (int t = 0; t < (LT ? LT : 1); ++t) { v0[t] = a.hh0[t]; v1[t] = a.hh1[t]; }
            }
            const size_t per = (size_t)a.H * a.W;
            T* const yp = a.y + (size_t)plane * 4 * per + (size_t)c0 + tx;
            for (int r = ty; r < th; r += 4) {
                const Pair* p = mid + r * TW + tx;
                A ll = 0, lh = 0, hl = 0, hhv = 0;
                if (LT) {
#pragma unroll
                    for (int t = 0; t < (LT ? LT : 1); ++t) {
                        const Pair s = p[t * d * TW];
                        ll += v0[t] * s.lo; hl += v0[t] * s.hi; lh += v1[t] * s.lo; hhv += v1[t] * s.hi;
                    }
                } else {
                    for (int t = 0; t < Lh; ++t) {
                        const Pair s = p[t * d * TW];
                        ll += a.hh0[t] * s.lo; hl += a.hh0[t] * s.hi; lh += a.hh1[t] * s.lo; hhv += a.hh1[t] * s.hi;
                    }
                }
                T* o = yp + (size_t)(r0 + r) * a.W;
                o[0] = (T)ll; o[per] = (T)lh; o[2 * per] = (T)hl; o[3 * per] = (T)hhv;
            }
        }
    }
};
