// Compile-time specialised single-level 2-D DWT synthesis tile kernel.
//
//   y[n] = sum_{k'} B(k') g[n + s - 2k'],  s = L-2 (zero/symmetric/reflect/periodic) or L/2-1 (periodization,
//   circular band index).  Polyphase form used here (L even): outputs come in pairs (n0, n0+1) with n0+s even,
//   c = (n0+s)/2:   (y[n0], y[n0+1]) = sum_{j<L/2} B(c-j) * (g[2j], g[2j+1])
//   i.e. one packed FMA per tap and band with the band sample broadcast - no zero stuffing, no transposed
//   convolution, half the MACs of the reference's conv_transpose2d formulation.
//
// One 256-thread workgroup produces a 32 x 64 tile of y:
//   stage : the four band tiles (ll through explicit strides, lh/hl/hh from `highs`) are loaded with coalesced
//           dword loads (all loads of a thread issued before the first LDS write) and stored interleaved as one
//           float4 (ll,lh,hl,hh) per cell; cells outside the bands are zero (or wrap, periodization);
//   column: item = (row pair, band column): L/2 ds_read_b128, 4 packed FMAs per tap -> (lo,hi) for both rows;
//   row   : item = (row, TWO column pairs): (L/2+2)/2 ds_read_b128 (a sliding window of (lo,hi) samples), 4 packed
//           FMAs per tap -> 16 contiguous bytes of y per lane.
// The parity of the shift s is a template parameter: for even s (every non-periodization mode with L = 2 mod 4
// excluded - e.g. db4 symmetric) a 32 x 64 tile needs exactly 16 x 32 output pairs, no spare pair row / column.
//
// Restates SFB2D.forward (reference dwt/lowlevel.py:671-680 = 3 x sfb1d = 6 x conv_transpose2d + 3 adds,
// :226-271) and, with a cropped OH/OW, AFB2D.backward (:350-365).
#pragma once
#include "wl_common.h"
#include "wl_dtcwt_tile.h"   // WlPair / WlQuad

template <typename T>
struct WlSfbTileArgs {
    const T* ll;      // (NC, Kh, Kw) through strides
    const T* highs;   // (NC, 3, Kh, Kw) dense or nullptr
    T* y;             // (NC, OH, OW)
    const float* g_w_lo;
    const float* g_w_hi;
    const float* g_h_lo;
    const float* g_h_hi;
    int64_t NC;
    int64_t ll_plane_stride;
    int ll_row_stride;
    int Kh, Kw, OH, OW;
    int s, circ;
    int tiles_x, tiles_y;
    int64_t nblocks;  // grid size (for the XCD-aware block remap)
};

template <typename T, int LT, int SODD>   // SODD = s & 1 (parity of the synthesis shift)
struct WlSfbTile {
    typedef WlSfbTileArgs<T> Args;
    static const int kThreads = 256;
    static const int kMinWaves = 4;
    static const int TH = 32, TW = 64;
    static const int HL = LT / 2;                 // taps per phase
    static const int NPR = TH / 2 + SODD;         // row pairs per tile (odd s shifts the pairing by one)
    static const int NPC = TW / 2 + 2 * SODD;     // column pairs per tile (even: the row pass takes two at a time)
    static const int NKR = NPR + HL - 1;          // staged band rows
    static const int NKC = (NPC + HL - 1 + 1) & ~1;   // staged band cols = LDS pitch in cells (even: b128 reads of U)
    static const int UR = 2 * NPR;                // intermediate rows
    static const int kTapFloats = 4 * LT;
    static const int kLdsFloats = kTapFloats + 4 * NKR * NKC + 2 * UR * NKC;
    typedef WlPair<T> Pair;

    static WL_DEV void run(const Args& a, const WlCtx& ctx) {
        const int tid = ctx.tid;
        const int tiles = a.tiles_x * a.tiles_y;
        const int64_t lbid = wl_xcd_remap(ctx.bid, a.nblocks);
        const int64_t plane = lbid / tiles;
        const int tile = (int)(lbid - plane * tiles);
        const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
        constexpr int sodd = SODD;
        // first pair of the tile: n0 = 2*m0 - sodd with 2*m0 = tile origin; c = (n0 + s)/2
        const int nrow0 = ty * TH - sodd, ncol0 = tx * TW - sodd;
        const int cr0 = (nrow0 + a.s) >> 1, cc0 = (ncol0 + a.s) >> 1;   // exact (even), may be negative
        const int kr0 = cr0 - (HL - 1), kc0 = cc0 - (HL - 1);           // first staged band row / col
        float* lds = reinterpret_cast<float*>(ctx.smem);
        float* tl = lds;                                  // g_w pairs, then g_h pairs
        wl_f4* B = reinterpret_cast<wl_f4*>(lds + kTapFloats);
        wl_f2* U = reinterpret_cast<wl_f2*>(lds + kTapFloats + 4 * NKR * NKC);
        if (tid < LT) {
            tl[tid] = a.g_w_lo[tid]; tl[LT + tid] = a.g_w_hi[tid];
            tl[2 * LT + tid] = a.g_h_lo[tid]; tl[3 * LT + tid] = a.g_h_hi[tid];
        }
        // ---- stage the four band tiles -------------------------------------------------------------------------
        {
            const unsigned bplane = (unsigned)a.Kh * (unsigned)a.Kw;
            const T* llp = a.ll + (size_t)plane * a.ll_plane_stride;
            const T* hp = a.highs ? a.highs + (size_t)plane * 3 * bplane : nullptr;
            // even geometry (band width, strides, first staged column): ALIGNED two-cell loads per lane - half the load
            // instructions.  (Odd band widths, e.g. 259, keep element loads: misaligned 8-byte loads measured slower.)
            const bool pairs2 = !(a.Kw & 1) && !(a.ll_row_stride & 1) && !(kc0 & 1) && !(a.ll_plane_stride & 1) &&
                                ((uintptr_t)a.ll % (2 * sizeof(T)) == 0) &&
                                (!a.highs || (uintptr_t)a.highs % (2 * sizeof(T)) == 0) && !(bplane & 1);
            if (pairs2) {
                constexpr int NPAIR = NKC / 2, NIT2 = (NKR * NPAIR + kThreads - 1) / kThreads;
                WlPair<T> pv[NIT2][4];
#pragma unroll
                for (int it = 0; it < NIT2; ++it) {
                    const int f = tid + it * kThreads;
#pragma unroll
                    for (int b = 0; b < 4; ++b) pv[it][b].a = pv[it][b].b = (T)0;
                    if (f < NKR * NPAIR) {
                        const int i = f / NPAIR, j = 2 * (f - i * NPAIR);
                        int r = kr0 + i, c = kc0 + j;
                        if (a.circ) {
                            if ((unsigned)r >= (unsigned)a.Kh) r = wl_pmod(r, a.Kh);
                            if ((unsigned)c >= (unsigned)a.Kw) c = wl_pmod(c, a.Kw);   // even c, even Kw: the pair stays together
                        }
                        if ((unsigned)r < (unsigned)a.Kh && (unsigned)c < (unsigned)a.Kw) {
                            pv[it][0] = *reinterpret_cast<const WlPair<T>*>(llp + (unsigned)(r * a.ll_row_stride + c));
                            if (hp) {
                                const T* q = hp + ((unsigned)r * (unsigned)a.Kw + (unsigned)c);
                                pv[it][1] = *reinterpret_cast<const WlPair<T>*>(q);
                                pv[it][2] = *reinterpret_cast<const WlPair<T>*>(q + bplane);
                                pv[it][3] = *reinterpret_cast<const WlPair<T>*>(q + 2 * bplane);
                            }
                        }
                    }
                }
#pragma unroll
                for (int it = 0; it < NIT2; ++it) {
                    const int f = tid + it * kThreads;
                    if (f < NKR * NPAIR) {
                        wl_f4 w0, w1;
                        w0.x = (float)pv[it][0].a; w0.y = (float)pv[it][1].a; w0.z = (float)pv[it][2].a; w0.w = (float)pv[it][3].a;
                        w1.x = (float)pv[it][0].b; w1.y = (float)pv[it][1].b; w1.z = (float)pv[it][2].b; w1.w = (float)pv[it][3].b;
                        B[2 * f] = w0; B[2 * f + 1] = w1;   // cell (i, j) = i*NKC + j = 2f
                    }
                }
            } else {
            constexpr int NIT = (NKR * NKC + kThreads - 1) / kThreads;
            float v[NIT][4];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int f = tid + it * kThreads;
                v[it][0] = v[it][1] = v[it][2] = v[it][3] = 0.f;
                if (f < NKR * NKC) {
                    const int i = f / NKC, j = f - i * NKC;
                    int r = kr0 + i, c = kc0 + j;
                    if (a.circ) {   // wrap only the out-of-range (border) cells: the modulo is slow
                        if ((unsigned)r >= (unsigned)a.Kh) r = wl_pmod(r, a.Kh);
                        if ((unsigned)c >= (unsigned)a.Kw) c = wl_pmod(c, a.Kw);
                    }
                    if ((unsigned)r < (unsigned)a.Kh && (unsigned)c < (unsigned)a.Kw) {
                        v[it][0] = (float)llp[(unsigned)(r * a.ll_row_stride + c)];
                        if (hp) {
                            const unsigned o = (unsigned)r * (unsigned)a.Kw + (unsigned)c;
                            v[it][1] = (float)hp[o];
                            v[it][2] = (float)hp[bplane + o];
                            v[it][3] = (float)hp[2 * bplane + o];
                        }
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int f = tid + it * kThreads;
                if (f < NKR * NKC) {
                    wl_f4 w; w.x = v[it][0]; w.y = v[it][1]; w.z = v[it][2]; w.w = v[it][3];
                    B[f] = w;
                }
            }
            }
        }
        ctx.sync();
        // ---- column synthesis (along H): row pair p -> U rows 2p, 2p+1 of (lo,hi) ------------------------------
        {
            wl_v2 g0[HL], g1[HL];
#pragma unroll
            for (int j = 0; j < HL; ++j) {
                g0[j].x = tl[2 * LT + 2 * j]; g0[j].y = tl[2 * LT + 2 * j + 1];
                g1[j].x = tl[3 * LT + 2 * j]; g1[j].y = tl[3 * LT + 2 * j + 1];
            }
            _Pragma("nounroll") for (int f = tid; f < NPR * NKC; f += kThreads) {
                const int p = f / NKC, kc = f - p * NKC;
                wl_v2 lo = {0.f, 0.f}, hi = {0.f, 0.f};   // (row n0, row n0+1)
                // c = cr0 + p;  band row c - j  ->  staged row (c - j) - kr0 = p + HL-1 - j
#pragma unroll
                for (int j = 0; j < HL; ++j) {
                    const wl_f4 b = B[(p + HL - 1 - j) * NKC + kc];
                    lo += g0[j] * b.x; lo += g1[j] * b.y;
                    hi += g0[j] * b.z; hi += g1[j] * b.w;
                }
                wl_f2 u0, u1;
                u0.x = lo.x; u0.y = hi.x; u1.x = lo.y; u1.y = hi.y;
                U[(2 * p) * NKC + kc] = u0;
                U[(2 * p + 1) * NKC + kc] = u1;
            }
        }
        ctx.sync();
        // ---- row synthesis (along W) + store ---------------------------------------------------------------------
        {
            wl_v2 g0[HL], g1[HL];
#pragma unroll
            for (int j = 0; j < HL; ++j) {
                g0[j].x = tl[2 * j]; g0[j].y = tl[2 * j + 1];
                g1[j].x = tl[LT + 2 * j]; g1[j].y = tl[LT + 2 * j + 1];
            }
            T* yp = a.y + (size_t)plane * a.OH * a.OW;
            const int row_lo = ty * TH, row_hi = (ty * TH + TH) < a.OH ? (ty * TH + TH) : a.OH;
            const int col_lo = tx * TW, col_hi = (tx * TW + TW) < a.OW ? (tx * TW + TW) : a.OW;
            // item = (intermediate row, two column pairs): window of HL+1 (lo,hi) samples -> 4 contiguous outputs
            constexpr int NQ2 = NPC / 2;
            constexpr int NW = (HL + 1 + 1) / 2;      // ds_read_b128 per item (two (lo,hi) samples each)
            _Pragma("nounroll") for (int f = tid; f < UR * NQ2; f += kThreads) {
                const int ur = f / NQ2, q2 = f - ur * NQ2;
                const int n = nrow0 + ur;
                if (n < row_lo || n >= row_hi) continue;
                // pairs q = 2*q2 and q+1 read U[q + HL-1 - j] and U[q + HL - j], j < HL: cells 2*q2 .. 2*q2 + HL
                float ul[2 * NW], uh[2 * NW];
                const wl_f4* u4 = reinterpret_cast<const wl_f4*>(U + ur * NKC + 2 * q2);
#pragma unroll
                for (int u = 0; u < NW; ++u) {
                    const wl_f4 t = u4[u];
                    ul[2 * u] = t.x; uh[2 * u] = t.y; ul[2 * u + 1] = t.z; uh[2 * u + 1] = t.w;
                }
                wl_v2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};   // (col w0, w0+1), (w0+2, w0+3)
#pragma unroll
                for (int j = 0; j < HL; ++j) {
                    acc0 += g0[j] * ul[HL - 1 - j]; acc0 += g1[j] * uh[HL - 1 - j];
                    acc1 += g0[j] * ul[HL - j];     acc1 += g1[j] * uh[HL - j];
                }
                const int w0 = ncol0 + 4 * q2;
                T* dst = yp + (n * a.OW + w0);   // (w0 may be -1: signed offset)
                if (w0 >= col_lo && w0 + 3 < col_hi) {
                    if (SODD == 0) {
                        WlQuad<T> o; o.a = (T)acc0.x; o.b = (T)acc0.y; o.c = (T)acc1.x; o.d = (T)acc1.y;
                        *reinterpret_cast<WlQuad<T>*>(dst) = o;
                    } else {
                        Pair p0, p1; p0.a = (T)acc0.x; p0.b = (T)acc0.y; p1.a = (T)acc1.x; p1.b = (T)acc1.y;
                        *reinterpret_cast<Pair*>(dst) = p0;
                        *reinterpret_cast<Pair*>(dst + 2) = p1;
                    }
                } else {
                    const float v[4] = {acc0.x, acc0.y, acc1.x, acc1.y};
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (w0 + u >= col_lo && w0 + u < col_hi) dst[u] = (T)v[u];
                }
            }
        }
    }
};
