// Compile-time specialised DTCWT tile kernels (float / half data, float accumulation) for the tap counts of
// the reference's filter tables (biort: 5/7 near_sym_a, 13/19 near_sym_b, 9/7 antonini, 5/3 legall and their
// synthesis counterparts; qshift: 10, 14, 16, 18 taps).  Other tap counts and double data use the generic
// kernels of wl_dtcwt_kernels.h; the arithmetic is the same, only the work mapping differs:
//   * every LDS access is a 16-byte (or 8-byte) vector access with lanes on consecutive addresses,
//   * each staged sample feeds a register sliding window (4 outputs per thread along the filtered axis), so a
//     tap costs one FMA, not one FMA + two LDS reads,
//   * global loads are issued in a batch before the first LDS write; rows/columns resolve their boundary
//     extension once per lane, not once per element,
//   * sub-band stores are 8 contiguous bytes per lane (one complex value / two lowpass samples).
#pragma once
#include "wl_common.h"
#include "wl_dwt_stream.h"      // wl_f4 / wl_f2 / wl_v2
#include "wl_dtcwt_kernels.h"   // argument structs, wl_dtfwd1_quad_out, WlPair

// ---------------------------------------------------------------------------------------------------------
// level 1 forward (+ ScatLayer epilogue): fwd_j1, reference dtcwt/transform_funcs.py:98-121 and
// scatternet/lowlevel.py:86-109.  One workgroup = one TH x TW tile of the (padded-to-even) full-res plane.
//   stage : (TH+2M) x SP input cells, origin (r0-M, c0-MA) with MA = M rounded up to even so that interior
//           lanes read aligned pairs;
//   row   : item = (staged row, 4 output columns): NV ds_read_b128 -> 4 x (lo, hi) -> 2 ds_write_b128;
//   column: item = (column pair, 4 output rows) = two 2x2 quads: 4+2M ds_read_b128 of (lo,hi,lo,hi), all four
//           bands of both quads in registers, q2c + stores through wl_dtfwd1_quad_out.
// ---------------------------------------------------------------------------------------------------------
template <typename T, int L0, int L1, int TH_ = 32, int TW_ = 64>
struct WlDtFwd1Tile {
    typedef WlDtFwd1Args<T> Args;
    static const int kThreads = 256;
    static const int kMinWaves = 2;
    static const int TH = TH_, TW = TW_;
    static const int M0 = L0 / 2, M1 = L1 / 2, M = M0 > M1 ? M0 : M1, MA = (M + 1) & ~1;
    static const int NR = TH + 2 * M;                    // staged rows
    static const int NV = (MA + M + 4 + 3) / 4;          // float4 reads per row item
    static const int SP = 4 * (TW / 4 - 1) + 4 * NV;     // staged row pitch (floats)
    static const int TP = 2 * TW;                        // (lo,hi) row pitch (floats)
    static const int kTapFloats = (L0 + L1 + 3) & ~3;
    static const int kLdsFloats = kTapFloats + NR * SP + NR * TP;
    static const int NP = SP / 2;                        // staged pairs per row
    static const int RPI = kThreads / NP;                // staged rows per staging iteration
    static const int NIT = (NR + RPI - 1) / RPI;
    static const int NQT = (TH / 4) * (TW / 2);          // column items per tile
    static const int NQI = (NQT + kThreads - 1) / kThreads;
    typedef T Pair2 __attribute__((ext_vector_type(2)));

    static WL_DEV void run(const Args& a, const WlCtx& ctx) {
        const int tid = ctx.tid;
        const int tiles = a.tiles_x * a.tiles_y;
        const int64_t unit = ctx.bid / tiles;            // plane, or image when combining colour
        const int tile = (int)(ctx.bid - unit * tiles);
        const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
        const int r0 = ty * TH, c0 = tx * TW;
        float* lds = reinterpret_cast<float*>(ctx.smem);
        float* tl = lds;
        float* S = lds + kTapFloats;
        float* Tm = S + NR * SP;
        if (tid < L0) tl[tid] = a.h0[tid];
        if (tid < L1) tl[L0 + tid] = a.h1[tid];
        // ---- staging geometry: resolved once per workgroup ---------------------------------------------------
        const int s_row = tid / NP, p_own = tid - s_row * NP;
        const bool lane_on = s_row < RPI;
        const int padr = a.He - a.H, padc = a.We - a.W;
        const int cs0 = lane_on ? wl_ext_padded(c0 - MA + 2 * p_own, a.W, 0, padc, a.ext) : -1;
        const int cs1 = lane_on ? wl_ext_padded(c0 - MA + 2 * p_own + 1, a.W, 0, padc, a.ext) : -1;
        const bool vec_ok = (a.W % 2 == 0) && ((uintptr_t)a.x % (2 * sizeof(T)) == 0);
        const bool pair_ld = vec_ok && cs0 >= 0 && cs1 == cs0 + 1 && (cs0 & 1) == 0;
        int rsrc[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = it * RPI + s_row;
            rsrc[it] = (lane_on && i < NR) ? wl_ext_padded(r0 - M + i, a.H, 0, padr, a.ext) : -1;
        }
        const int nch = a.combine ? 3 : 1;
        float msum[NQI][2][6];
        for (int ch = 0; ch < nch; ++ch) {
            const int64_t plane = a.combine ? unit * 3 + ch : unit;
            const T* xp = a.x + (size_t)plane * a.H * a.W;
            // ---- stage: all loads first, then the LDS writes ------------------------------------------------------
            {
                Pair2 pf[NIT];
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    pf[it] = Pair2{(T)0, (T)0};
                    const int r = rsrc[it];
                    if (r >= 0) {
                        const T* src = xp + r * a.W;
                        if (pair_ld) pf[it] = *reinterpret_cast<const Pair2*>(src + cs0);
                        else {
                            if (cs0 >= 0) pf[it].x = src[cs0];
                            if (cs1 >= 0) pf[it].y = src[cs1];
                        }
                    }
                }
                if (lane_on) {
                    float* d = S + s_row * SP + 2 * p_own;
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        if (it * RPI + s_row < NR) {
                            wl_f2 w; w.x = (float)pf[it].x; w.y = (float)pf[it].y;
                            *reinterpret_cast<wl_f2*>(d + it * RPI * SP) = w;
                        }
                    }
                }
            }
            ctx.sync();
            // ---- row bank: lo[j] = sum h0[t] s[j+MA-M0+t], hi[j] = sum h1[t] s[j+MA-M1+t] -----------------------
            {
                float t0[L0], t1[L1];
#pragma unroll
                for (int t = 0; t < L0; ++t) t0[t] = tl[t];
#pragma unroll
                for (int t = 0; t < L1; ++t) t1[t] = tl[L0 + t];
                _Pragma("nounroll") for (int f = tid; f < NR * (TW / 4); f += kThreads) {
                    const int i = f / (TW / 4), g = f - i * (TW / 4);
                    float v[NV * 4];
                    const wl_f4* s4 = reinterpret_cast<const wl_f4*>(S + i * SP) + g;
#pragma unroll
                    for (int u = 0; u < NV; ++u) {
                        const wl_f4 q = s4[u];
                        v[4 * u] = q.x; v[4 * u + 1] = q.y; v[4 * u + 2] = q.z; v[4 * u + 3] = q.w;
                    }
                    float lo[4] = {0.f, 0.f, 0.f, 0.f}, hi[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
#pragma unroll
                        for (int t = 0; t < L0; ++t) lo[u] += t0[t] * v[u + MA - M0 + t];
#pragma unroll
                        for (int t = 0; t < L1; ++t) hi[u] += t1[t] * v[u + MA - M1 + t];
                    }
                    wl_f4 o0, o1;
                    o0.x = lo[0]; o0.y = hi[0]; o0.z = lo[1]; o0.w = hi[1];
                    o1.x = lo[2]; o1.y = hi[2]; o1.z = lo[3]; o1.w = hi[3];
                    wl_f4* d = reinterpret_cast<wl_f4*>(Tm + i * TP) + 2 * g;
                    d[0] = o0; d[1] = o1;
                }
            }
            ctx.sync();
            // ---- column bank + q2c + stores -----------------------------------------------------------------------
            {
                float t0[L0], t1[L1];
#pragma unroll
                for (int t = 0; t < L0; ++t) t0[t] = tl[t];
#pragma unroll
                for (int t = 0; t < L1; ++t) t1[t] = tl[L0 + t];
#pragma unroll
                for (int qi = 0; qi < NQI; ++qi) {
                    const int f = tid + qi * kThreads;
                    if (f >= NQT) break;
                    const int rg = f / (TW / 2), cp = f - rg * (TW / 2);
                    // acc[row][col][band]: band 0 ll, 1 lh, 2 hl, 3 hh
                    float acc[4][2][4];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int c = 0; c < 2; ++c) acc[r][c][0] = acc[r][c][1] = acc[r][c][2] = acc[r][c][3] = 0.f;
                    const float* col = Tm + (4 * rg) * TP + 4 * cp;
#pragma unroll
                    for (int w = 0; w < 4 + 2 * M; ++w) {
                        const wl_f4 p = *reinterpret_cast<const wl_f4*>(col + w * TP);   // lo_c0, hi_c0, lo_c1, hi_c1
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int ta = w - r - (M - M0), tb = w - r - (M - M1);
                            if (ta >= 0 && ta < L0) {
                                acc[r][0][0] += t0[ta] * p.x; acc[r][0][2] += t0[ta] * p.y;
                                acc[r][1][0] += t0[ta] * p.z; acc[r][1][2] += t0[ta] * p.w;
                            }
                            if (tb >= 0 && tb < L1) {
                                acc[r][0][1] += t1[tb] * p.x; acc[r][0][3] += t1[tb] * p.y;
                                acc[r][1][1] += t1[tb] * p.z; acc[r][1][3] += t1[tb] * p.w;
                            }
                        }
                    }
#pragma unroll
                    for (int qd = 0; qd < 2; ++qd) {
                        const int R = r0 + 4 * rg + 2 * qd, Cc = c0 + 2 * cp;
                        if (R >= a.He || Cc >= a.We) continue;
                        float ll[4], lh[4], hl[4], hh[4];
#pragma unroll
                        for (int p = 0; p < 4; ++p) {
                            const int r = 2 * qd + (p >> 1), c = p & 1;
                            ll[p] = acc[r][c][0]; lh[p] = acc[r][c][1]; hl[p] = acc[r][c][2]; hh[p] = acc[r][c][3];
                        }
                        wl_dtfwd1_quad_out<T>(a, plane, ch, R, Cc, ll, lh, hl, hh, msum[qi][qd]);
                    }
                }
            }
            // (the barrier after the next colour plane's staging orders this column bank before its row bank)
        }
    }
};
