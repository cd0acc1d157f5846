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
#include "wl_dtcwt_kernels.h"   // argument structs, wl_dtfwd1_quad_out, WlPair

// ---------------------------------------------------------------------------------------------------------
// level 1 forward (+ ScatLayer epilogue): fwd_j1, reference dtcwt/transform_funcs.py:98-121 and
// scatternet/lowlevel.py:86-109.  One workgroup walks a horizontal run of TH x TW (32 x 64) tiles of the (padded-to-even) full-res plane.
//   stage : (TH+2M) x SP input cells, origin (r0-M, c0-MA) with MA = M rounded up to even so that interior
//           lanes read aligned pairs;
//   row   : item = (staged row, 4 output columns): NV ds_read_b128 -> 4 x (lo, hi) -> 2 ds_write_b128;
//   column: item = one 2x2 quad: 2+2M ds_read_b128 of (lo,hi,lo,hi), packed FMAs ((ll,hl) and (lh,hh) pairs), all
//           four bands in registers, q2c + stores through wl_dtfwd1_quad_out.
// ---------------------------------------------------------------------------------------------------------
template <typename T, int L0, int L1, int COMB = 0, int TH_ = 32, int TW_ = 64>
struct WlDtFwd1Tile {
    typedef WlDtFwd1Args<T> Args;
    static const int kThreads = 256;
    static const int kMinWaves = (L0 + L1 <= 16) ? 4 : 2;
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
    static const int NQT = (TH / 2) * (TW / 2);          // column items (2x2 quads) per tile
    static const int NQI = (NQT + kThreads - 1) / kThreads;
    typedef T Pair2 __attribute__((ext_vector_type(2)));

    static WL_DEV void run(const Args& a, const WlCtx& ctx) {
        const int tid = ctx.tid;
        // A workgroup walks a horizontal run of tiles (x colour planes when combining colour); the loads of step
        // s+1 are in registers while step s is computed and are committed to LDS between its two passes, the loads
        // of step s+2 are issued right after (same software pipeline as the DWT analysis kernel, wl_dwt_tile.h).
        const int per_unit = a.tiles_y * a.runs_x;
        const int64_t lbid = wl_xcd_remap(ctx.bid, a.nblocks);
        const int64_t unit = lbid / per_unit;            // plane, or image when combining colour
        const int rem = (int)(lbid - unit * per_unit);
        const int ty = rem / a.runs_x, rx = rem - ty * a.runs_x;
        const int tx_begin = rx * a.run_len;
        const int tx_end = tx_begin + a.run_len < a.tiles_x ? tx_begin + a.run_len : a.tiles_x;
        const int r0 = ty * TH;
        float* lds = reinterpret_cast<float*>(ctx.smem);
        float* tl = lds;
        float* S = lds + kTapFloats;
        float* Tm = S + NR * SP;
        if (tid < L0) tl[tid] = a.h0[tid];
        if (tid < L1) tl[L0 + tid] = a.h1[tid];
        // ---- staging geometry: rows resolved once per workgroup, columns once per tile ------------------------
        const int s_row = tid / NP, p_own = tid - s_row * NP;
        const bool lane_on = s_row < RPI;
        const int padr = a.He - a.H, padc = a.We - a.W;
        const bool vec_ok = (a.W % 2 == 0) && ((uintptr_t)a.x % (2 * sizeof(T)) == 0);
        int roff[NIT];   // element offset of the source row inside a plane, or -1
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = it * RPI + s_row;
            const int r = (lane_on && i < NR) ? wl_ext_padded(r0 - M + i, a.H, 0, padr, a.ext) : -1;
            roff[it] = r < 0 ? -1 : r * a.W;
        }
        constexpr int nch = COMB ? 3 : 1;
        const int nsteps = (tx_end - tx_begin) * nch;    // step = (tile column, colour plane)
        Pair2 pf[NIT];
        auto issue = [&](int step) {
            const int tx = tx_begin + step / nch, ch = step - (step / nch) * nch;
            const T* xp = a.x + (size_t)(COMB ? unit * 3 + ch : unit) * a.H * a.W;
            const int c0 = tx * TW;
            const int cs0 = lane_on ? wl_ext_padded(c0 - MA + 2 * p_own, a.W, 0, padc, a.ext) : -1;
            const int cs1 = lane_on ? wl_ext_padded(c0 - MA + 2 * p_own + 1, a.W, 0, padc, a.ext) : -1;
            const bool pair_ld = vec_ok && cs0 >= 0 && cs1 == cs0 + 1 && (cs0 & 1) == 0;
            const int cpair = pair_ld ? cs0 : -1;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                pf[it] = Pair2{(T)0, (T)0};
                if ((cpair | roff[it]) >= 0) pf[it] = *reinterpret_cast<const Pair2*>(xp + (unsigned)(roff[it] + cpair));
            }
            if (lane_on && !pair_ld) {   // border lanes (mirrored / replicated / odd widths): element loads
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    if (roff[it] >= 0) {
                        if (cs0 >= 0) pf[it].x = xp[(unsigned)(roff[it] + cs0)];
                        if (cs1 >= 0) pf[it].y = xp[(unsigned)(roff[it] + cs1)];
                    }
                }
            }
        };
        auto commit = [&]() {
            if (!lane_on) return;
            float* d = S + s_row * SP + 2 * p_own;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                if (it * RPI + s_row < NR) {
                    wl_f2 w; w.x = (float)pf[it].x; w.y = (float)pf[it].y;
                    *reinterpret_cast<wl_f2*>(d + it * RPI * SP) = w;
                }
            }
        };
        float msum[COMB ? NQI : 1][6];
        issue(0);
        commit();
        ctx.sync();
        if (1 < nsteps) issue(1);
        for (int step = 0; step < nsteps; ++step) {
            const int tx = tx_begin + step / nch, ch = step - (step / nch) * nch;
            const int64_t plane = COMB ? unit * 3 + ch : unit;
            const int c0 = tx * TW;
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
            if (step + 1 < nsteps) {
                commit();                                  // step+1 -> S (the row bank above was its last reader)
                if (step + 2 < nsteps) issue(step + 2);
            }
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
                    const int qr = f / (TW / 2), cp = f - qr * (TW / 2);
                    // [row][col] of the quad: aL = (ll, hl) (lowpass along H), aH = (lh, hh) (highpass along H)
                    wl_v2 aL[2][2], aH[2][2];
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int c = 0; c < 2; ++c) { aL[r][c] = wl_v2{0.f, 0.f}; aH[r][c] = wl_v2{0.f, 0.f}; }
                    const float* col = Tm + (2 * qr) * TP + 4 * cp;
#pragma unroll
                    for (int w = 0; w < 2 + 2 * M; ++w) {
                        const wl_f4 p = *reinterpret_cast<const wl_f4*>(col + w * TP);   // lo_c0, hi_c0, lo_c1, hi_c1
                        const wl_v2 s0 = {p.x, p.y}, s1 = {p.z, p.w};
#pragma unroll
                        for (int r = 0; r < 2; ++r) {
                            const int ta = w - r - (M - M0), tb = w - r - (M - M1);
                            if (ta >= 0 && ta < L0) { aL[r][0] += s0 * t0[ta]; aL[r][1] += s1 * t0[ta]; }
                            if (tb >= 0 && tb < L1) { aH[r][0] += s0 * t1[tb]; aH[r][1] += s1 * t1[tb]; }
                        }
                    }
                    const int R = r0 + 2 * qr, Cc = c0 + 2 * cp;
                    if (R >= a.He || Cc >= a.We) continue;
                    float ll[4], lh[4], hl[4], hh[4];
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int r = p >> 1, c = p & 1;
                        ll[p] = aL[r][c].x; hl[p] = aL[r][c].y; lh[p] = aH[r][c].x; hh[p] = aH[r][c].y;
                    }
                    wl_dtfwd1_quad_out<T, COMB>(a, plane, ch, R, Cc, ll, lh, hl, hh, msum[COMB ? qi : 0]);
                }
            }
            ctx.sync();   // S(step+1) visible; Tm free for the next row bank
        }
    }
};

// ---------------------------------------------------------------------------------------------------------
// level >= 2 forward: fwd_j2plus, reference dtcwt/transform_funcs.py:226-249 (coldfilt/rowdfilt,
// dtcwt/lowlevel.py:99-151).  With E_t = X[4k+2-L+2t], O_t = X[4k+3-L+2t]  (t < L):
//   lo[2k] = sum h0b[t] E_t   lo[2k+1] = sum h0a[t] O_t   hi[2k] = sum h1a[t] O_t   hi[2k+1] = sum h1b[t] E_t
// so each staged sample meets exactly one tap PAIR: (h0b,h1b)[t] for even offsets, (h0a,h1a)[t] for odd ones
// -> one packed FMA per sample and output pair.  One workgroup = THo x TWo half-resolution outputs.
//   stage : (2*THo+2L-4) x (2*TWo+2L-4) cells of the (replicate-padded, symmetric-extended) input;
//   row   : item = (staged row, output pair kk): L/2 ds_read_b128 -> (lo,hi) of outputs 2kk, 2kk+1;
//   column: item = (column pair, output row pair) = one 2x2 quad: 2L ds_read_b128, 4 packed FMAs each.
// ---------------------------------------------------------------------------------------------------------
template <typename T, int L, int THO_ = 16, int TWO_ = 64>
struct WlDtFwd2Tile {
    typedef WlDtFwd2Args<T> Args;
    static const int kThreads = 256;
    static const int kMinWaves = 2;
    static const int THO = THO_, TWO = TWO_;
    static const int NR = 2 * THO + 2 * L - 4;
    static const int SP = 2 * TWO + 2 * L - 4;           // multiple of 4 (L even)
    static const int TP = 2 * TWO;
    static const int kTapFloats = 4 * L;
    static const int kLdsFloats = kTapFloats + NR * SP + NR * TP;
    static const int NP = SP / 2;
    static const int RPI = kThreads / NP;
    static const int NIT = (NR + RPI - 1) / RPI;
    static const int NQT = (THO / 2) * (TWO / 2);
    typedef T Pair2 __attribute__((ext_vector_type(2)));
    typedef WlPair<T> Pair;

    static WL_DEV void run(const Args& a, const WlCtx& ctx) {
        const int tid = ctx.tid;
        const int tiles = a.tiles_x * a.tiles_y;
        const int64_t lbid = wl_xcd_remap(ctx.bid, a.nblocks);   // neighbouring tiles share their halo in one L2
        const int64_t plane = lbid / tiles;
        const int tile = (int)(lbid - plane * tiles);
        const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
        const int r0 = ty * THO, c0 = tx * TWO;          // half-res output origin
        float* lds = reinterpret_cast<float*>(ctx.smem);
        float* tl = lds;                                 // (h0b,h1b)[t] pairs, then (h0a,h1a)[t] pairs
        float* S = lds + kTapFloats;
        float* Tm = S + NR * SP;
        if (tid < L) {
            tl[2 * tid] = a.h0b[tid]; tl[2 * tid + 1] = a.h1b[tid];
            tl[2 * L + 2 * tid] = a.h0a[tid]; tl[2 * L + 2 * tid + 1] = a.h1a[tid];
        }
        const T* xp = a.x + (size_t)plane * a.H * a.W;
        const int er0 = 2 * r0 + 2 - L, ec0 = 2 * c0 + 2 - L;
        // ---- stage ------------------------------------------------------------------------------------------------
        {
            const int s_row = tid / NP, p_own = tid - s_row * NP;
            const bool lane_on = s_row < RPI;
            const int cs0 = wl_ext_padded(ec0 + 2 * p_own, a.W, a.padc, a.padc, WL_EXT_SYM);
            const int cs1 = wl_ext_padded(ec0 + 2 * p_own + 1, a.W, a.padc, a.padc, WL_EXT_SYM);
            const bool vec_ok = (a.W % 2 == 0) && ((uintptr_t)a.x % (2 * sizeof(T)) == 0);
            const bool pair_ld = vec_ok && cs1 == cs0 + 1 && (cs0 & 1) == 0;
            Pair2 pf[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int i = it * RPI + s_row;
                pf[it] = Pair2{(T)0, (T)0};
                if (lane_on && i < NR) {
                    const int r = wl_ext_padded(er0 + i, a.H, a.padr, a.padr, WL_EXT_SYM);
                    const T* src = xp + r * a.W;
                    if (pair_ld) pf[it] = *reinterpret_cast<const Pair2*>(src + cs0);
                    else { pf[it].x = src[cs0]; pf[it].y = src[cs1]; }
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
        wl_v2 tE[L], tO[L];
#pragma unroll
        for (int t = 0; t < L; ++t) {
            tE[t].x = tl[2 * t]; tE[t].y = tl[2 * t + 1];
            tO[t].x = tl[2 * L + 2 * t]; tO[t].y = tl[2 * L + 2 * t + 1];
        }
        // ---- row bank ---------------------------------------------------------------------------------------------
        _Pragma("nounroll") for (int f = tid; f < NR * (TWO / 2); f += kThreads) {
            const int i = f / (TWO / 2), kk = f - i * (TWO / 2);
            const wl_f4* s4 = reinterpret_cast<const wl_f4*>(S + i * SP) + kk;
            wl_v2 aE = {0.f, 0.f}, aO = {0.f, 0.f};   // (lo[2kk], hi[2kk+1]), (lo[2kk+1], hi[2kk])
#pragma unroll
            for (int u = 0; u < L / 2; ++u) {
                const wl_f4 q = s4[u];
                aE += tE[2 * u] * q.x; aO += tO[2 * u] * q.y;
                aE += tE[2 * u + 1] * q.z; aO += tO[2 * u + 1] * q.w;
            }
            wl_f4 o;
            o.x = aE.x; o.y = aO.y; o.z = aO.x; o.w = aE.y;
            reinterpret_cast<wl_f4*>(Tm + i * TP)[kk] = o;
        }
        ctx.sync();
        // ---- column bank + q2c + stores ---------------------------------------------------------------------------
        const int h2 = a.He / 2, w2 = a.We / 2, w4 = a.We / 4;
        const size_t qplane = (size_t)(a.He / 4) * w4;
        _Pragma("nounroll") for (int f = tid; f < NQT; f += kThreads) {
            const int kr = f / (TWO / 2), cp = f - kr * (TWO / 2);
            const int R = r0 + 2 * kr, Cc = c0 + 2 * cp;
            if (R >= h2 || Cc >= w2) continue;
            // e?[c] = (row 2kr lowpass-H, row 2kr+1 highpass-H), o?[c] = (row 2kr+1 lowpass-H, row 2kr highpass-H)
            wl_v2 eL[2] = {{0.f, 0.f}, {0.f, 0.f}}, eH[2] = {{0.f, 0.f}, {0.f, 0.f}};
            wl_v2 oL[2] = {{0.f, 0.f}, {0.f, 0.f}}, oH[2] = {{0.f, 0.f}, {0.f, 0.f}};
            const float* col = Tm + (4 * kr) * TP + 4 * cp;
#pragma unroll
            for (int t = 0; t < L; ++t) {
                const wl_f4 pe = *reinterpret_cast<const wl_f4*>(col + (2 * t) * TP);
                const wl_f4 po = *reinterpret_cast<const wl_f4*>(col + (2 * t + 1) * TP);
                eL[0] += tE[t] * pe.x; eH[0] += tE[t] * pe.y; eL[1] += tE[t] * pe.z; eH[1] += tE[t] * pe.w;
                oL[0] += tO[t] * po.x; oH[0] += tO[t] * po.y; oL[1] += tO[t] * po.z; oH[1] += tO[t] * po.w;
            }
            T* lp = a.ll + (size_t)plane * h2 * w2 + (size_t)R * w2 + Cc;
            Pair p0, p1;
            p0.a = (T)eL[0].x; p0.b = (T)eL[1].x; p1.a = (T)oL[0].x; p1.b = (T)oL[1].x;
            *reinterpret_cast<Pair*>(lp) = p0;
            *reinterpret_cast<Pair*>(lp + w2) = p1;
            if (!a.highs) continue;
            const float lh[4] = {oL[0].y, oL[1].y, eL[0].y, eL[1].y};
            const float hl[4] = {eH[0].x, eH[1].x, oH[0].x, oH[1].x};
            const float hh[4] = {oH[0].y, oH[1].y, eH[0].y, eH[1].y};
            float re[6], im[6];
            wl_q2c(lh, re[0], im[0], re[5], im[5]);
            wl_q2c(hh, re[1], im[1], re[4], im[4]);
            wl_q2c(hl, re[2], im[2], re[3], im[3]);
            const size_t q = (size_t)(R / 2) * w4 + (Cc / 2);
            T* hp = a.highs + (size_t)plane * 12 * qplane;
#pragma unroll
            for (int o = 0; o < 6; ++o) {
                Pair p; p.a = (T)re[o]; p.b = (T)im[o];
                *reinterpret_cast<Pair*>(hp + ((size_t)o * qplane + q) * 2) = p;
            }
        }
    }
};

// ---------------------------------------------------------------------------------------------------------
// shared by the inverse kernels: stage an (2*NRQ) x (2*NCQ) block of the four real bands as one float4
// (ll, hl, lh, hh) per cell - the two bands that meet the lowpass column filter first, then the two highpass ones.  The unit of work is a 2x2 QUAD: c2q (dtcwt/lowlevel.py:263-295) turns the two complex
// orientations of a band into the four samples of a quad, and the symmetric extension of an even-sized band
// maps quads onto quads (mirrored ones with their rows / columns swapped).  Per quad: 2 + 6 eight-byte loads,
// all issued before the first LDS write.   origin (pr_org, pc_org) must be even.
// ---------------------------------------------------------------------------------------------------------
template <typename T> struct __attribute__((packed, aligned(sizeof(T)), may_alias)) WlQuad { T a, b, c, d; };

// Where a quad's values come from.  A loader fills  l0,l1 = the two lowpass row pairs of source quad (qr,qc)  and
// z[o] = (re, im) of orientation o.
template <typename T>
struct WlDtQuadLoaderPlain {      // (ll plane through strides, highs (6, h/2, w/2, 2)); either may be nullptr
    const T* llp; int ll_rs; const T* hp; size_t qplane; int w2;
    WL_DEV void load(int qr, int qc, WlPair<T>& l0, WlPair<T>& l1, WlPair<T>* z) const {
        typedef WlPair<T> Pair;
        if (llp) {
            const T* p = llp + (size_t)(2 * qr) * ll_rs + 2 * qc;
            l0 = *reinterpret_cast<const Pair*>(p);
            l1 = *reinterpret_cast<const Pair*>(p + ll_rs);
        }
        if (hp) {
            const size_t q = (size_t)qr * w2 + qc;
#pragma unroll
            for (int o = 0; o < 6; ++o) z[o] = *reinterpret_cast<const Pair*>(hp + ((size_t)o * qplane + q) * 2);
        }
    }
};
// ScatLayer backward prologue (scatternet/lowlevel.py:114-137) fused into the staging:
//   ll = 1/4 nearest-upsample(dZ_ll),  w_o = dZ_o * (re/r, im/r)
template <typename T>
struct WlDtQuadLoaderScat {
    const T* zl;     // dZ lowpass plane of this (n,c): (h/2, w/2)
    const T* zr;     // dZ magnitude planes of this (n,c): orientation stride zr_os
    const T* dx;     // re/r, im/r planes of this (n,c): orientation stride d_os
    const T* dy;
    size_t zr_os, d_os;
    int w2;
    WL_DEV void load(int qr, int qc, WlPair<T>& l0, WlPair<T>& l1, WlPair<T>* z) const {
        const size_t q = (size_t)qr * w2 + qc;
        const T v = (T)((float)zl[q] * 0.25f);
        l0.a = l0.b = l1.a = l1.b = v;
#pragma unroll
        for (int o = 0; o < 6; ++o) {
            const float dr = (float)zr[o * zr_os + q];
            z[o].a = (T)(dr * (float)dx[o * d_os + q]);
            z[o].b = (T)(dr * (float)dy[o * d_os + q]);
        }
    }
};

template <typename T, int NRQ, int NCQ, int kThreads, typename Loader>
WL_DEV void wl_dt_stage_quads(wl_f4* B, int tid, int pr_org, int pc_org, int h, int w, int ext, const Loader& ld) {
    typedef WlPair<T> Pair;
    constexpr int NQ = NRQ * NCQ, NQI = (NQ + kThreads - 1) / kThreads, BP = 2 * NCQ;
    Pair l0[NQI], l1[NQI], z[NQI][6];
    int flip[NQI];
#pragma unroll
    for (int it = 0; it < NQI; ++it) {
        const int f = tid + it * kThreads;
        flip[it] = -1;
        l0[it].a = l0[it].b = l1[it].a = l1[it].b = (T)0;
#pragma unroll
        for (int o = 0; o < 6; ++o) z[it][o].a = z[it][o].b = (T)0;
        if (f < NQ) {
            const int Qr = f / NCQ, Qc = f - Qr * NCQ;
            const int sr = wl_ext(pr_org + 2 * Qr, h, ext), sc = wl_ext(pc_org + 2 * Qc, w, ext);
            flip[it] = 0;
            if (sr >= 0 && sc >= 0) {
                flip[it] = ((sr & 1) << 1) | (sc & 1);
                ld.load(sr >> 1, sc >> 1, l0[it], l1[it], z[it]);
            }
        }
    }
#pragma unroll
    for (int it = 0; it < NQI; ++it) {
        const int f = tid + it * kThreads;
        if (flip[it] < 0) continue;
        const int Qr = f / NCQ, Qc = f - Qr * NCQ;
        const float k = (float)WL_SQRT1_2;
        // natural-orientation cells n[sub], sub = 2*(row&1) + (col&1); pairs lh<-(0,5) hl<-(2,3) hh<-(1,4)
        wl_f4 n[4];
        n[0].x = (float)l0[it].a; n[1].x = (float)l0[it].b; n[2].x = (float)l1[it].a; n[3].x = (float)l1[it].b;
#define WL_C2Q(FIELD, O1, O2)                                                                         \
        {                                                                                             \
            const float w1r = (float)z[it][O1].a, w1i = (float)z[it][O1].b;                           \
            const float w2r = (float)z[it][O2].a, w2i = (float)z[it][O2].b;                           \
            n[0].FIELD = (w1r + w2r) * k; n[1].FIELD = (w1i + w2i) * k;                               \
            n[2].FIELD = (w1i - w2i) * k; n[3].FIELD = (w2r - w1r) * k;                               \
        }
        WL_C2Q(z, 0, 5) WL_C2Q(y, 2, 3) WL_C2Q(w, 1, 4)
#undef WL_C2Q
        if (flip[it] & 1) { wl_f4 t = n[0]; n[0] = n[1]; n[1] = t; t = n[2]; n[2] = n[3]; n[3] = t; }
        if (flip[it] & 2) { wl_f4 t = n[0]; n[0] = n[2]; n[2] = t; t = n[1]; n[1] = n[3]; n[3] = t; }
        wl_f4* d = B + (2 * Qr) * BP + 2 * Qc;
        d[0] = n[0]; d[1] = n[1]; d[BP] = n[2]; d[BP + 1] = n[3];
    }
}

// ---------------------------------------------------------------------------------------------------------
// level 1 inverse: inv_j1, reference dtcwt/transform_funcs.py:152-184.
//   lo = cf(ll,g0) + cf(lh,g1),  hi = cf(hl,g0) + cf(hh,g1),  y = rf(lo,g0) + rf(hi,g1)
//   stage : quads (see above), origin (r0-ME, c0-ME), ME = M rounded up to even;
//   column: item = (staged column, 4 output rows): 4+2M ds_read_b128 -> 4 x (lo,hi) -> ds_write_b64;
//   row   : item = (row, 4 output columns): ds_read_b128 of (lo,hi) pairs -> 16 contiguous bytes of y per lane.
// ---------------------------------------------------------------------------------------------------------
template <typename T, int L0, int L1, int SCAT = 0, int TH_ = 16, int TW_ = 32>   // SCAT: fused ScatLayer backward
struct WlDtInv1Tile {
    typedef WlDtInv1Args<T> Args;
    static const int kThreads = 256;
    static const int kMinWaves = 2;
    static const int TH = TH_, TW = TW_;
    static const int M0 = L0 / 2, M1 = L1 / 2, M = M0 > M1 ? M0 : M1, ME = (M + 1) & ~1;
    static const int NRS = TH + 2 * ME, NCS = TW + 2 * ME;       // staged cells (even; NCS multiple of 4)
    static const int NVU = (ME + M + 4 + 1) / 2;                 // ds_read_b128 (2 x (lo,hi)) per row item
    static const int kTapFloats = (L0 + L1 + 3) & ~3;
    static const int kLdsFloats = kTapFloats + 4 * NRS * NCS + 2 * TH * NCS;
    static_assert(4 * (TW / 4 - 1) + 2 * NVU <= NCS, "row window exceeds the staged width");

    static WL_DEV void run(const Args& a, const WlCtx& ctx) {
        const int tid = ctx.tid;
        const int tiles = a.tiles_x * a.tiles_y;
        const int64_t lbid = wl_xcd_remap(ctx.bid, a.nblocks);   // neighbouring tiles share their halo in one L2
        const int64_t plane = lbid / tiles;
        const int tile = (int)(lbid - plane * tiles);
        const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
        const int r0 = ty * TH, c0 = tx * TW;
        float* lds = reinterpret_cast<float*>(ctx.smem);
        float* tl = lds;
        wl_f4* B = reinterpret_cast<wl_f4*>(lds + kTapFloats);
        wl_f2* U = reinterpret_cast<wl_f2*>(lds + kTapFloats + 4 * NRS * NCS);
        if (tid < L0) tl[tid] = a.g0[tid];
        if (tid < L1) tl[L0 + tid] = a.g1[tid];
        const size_t qplane = (size_t)(a.H / 2) * (a.W / 2);
        if (SCAT) {
            const int64_t n = plane / a.C;
            const int c = (int)(plane - n * a.C);
            WlDtQuadLoaderScat<T> ld;
            ld.w2 = a.W / 2;
            if (!a.combine) {   // dZ (N,7,C,h,w)
                ld.zl = a.sz + ((size_t)n * 7 * a.C + c) * qplane;
                ld.zr = ld.zl + (size_t)a.C * qplane;
                ld.zr_os = (size_t)a.C * qplane;
            } else {            // dZ (N,3+6,h,w): the six magnitudes are shared by the three colours
                ld.zl = a.sz + ((size_t)n * 9 + c) * qplane;
                ld.zr = a.sz + ((size_t)n * 9 + 3) * qplane;
                ld.zr_os = qplane;
            }
            ld.dx = a.sdx + ((size_t)n * 6 * a.C + c) * qplane;
            ld.dy = a.sdy + ((size_t)n * 6 * a.C + c) * qplane;
            ld.d_os = (size_t)a.C * qplane;
            wl_dt_stage_quads<T, NRS / 2, NCS / 2, kThreads>(B, tid, r0 - ME, c0 - ME, a.H, a.W, a.ext, ld);
        } else {
            WlDtQuadLoaderPlain<T> ld;
            ld.llp = a.ll ? a.ll + (size_t)plane * a.ll_plane_stride : nullptr;
            ld.ll_rs = a.ll_row_stride;
            ld.hp = a.highs ? a.highs + (size_t)plane * 12 * qplane : nullptr;
            ld.qplane = qplane; ld.w2 = a.W / 2;
            wl_dt_stage_quads<T, NRS / 2, NCS / 2, kThreads>(B, tid, r0 - ME, c0 - ME, a.H, a.W, a.ext, ld);
        }
        ctx.sync();
        float t0[L0], t1[L1];
#pragma unroll
        for (int t = 0; t < L0; ++t) t0[t] = tl[t];
#pragma unroll
        for (int t = 0; t < L1; ++t) t1[t] = tl[L0 + t];
        // ---- column bank ------------------------------------------------------------------------------------------
        _Pragma("nounroll") for (int f = tid; f < (TH / 4) * NCS; f += kThreads) {
            const int rg = f / NCS, j = f - rg * NCS;
            wl_v2 lh[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};   // (lo, hi) of the four output rows
            const wl_f4* col = B + (4 * rg) * NCS + j;
#pragma unroll
            for (int w = ME - M; w < ME + M + 4; ++w) {
                const wl_f4 p = col[w * NCS];   // ll, hl, lh, hh
                const wl_v2 pl = {p.x, p.y}, ph = {p.z, p.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ta = w - r - (ME - M0), tb = w - r - (ME - M1);
                    if (ta >= 0 && ta < L0) lh[r] += pl * t0[ta];
                    if (tb >= 0 && tb < L1) lh[r] += ph * t1[tb];
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                wl_f2 u; u.x = lh[r].x; u.y = lh[r].y;
                U[(4 * rg + r) * NCS + j] = u;
            }
        }
        ctx.sync();
        // ---- row bank + store ---------------------------------------------------------------------------------------
        T* yp = a.y + (size_t)plane * a.H * a.W;
        _Pragma("nounroll") for (int f = tid; f < TH * (TW / 4); f += kThreads) {
            const int i = f / (TW / 4), g = f - i * (TW / 4);
            const int R = r0 + i, Cc = c0 + 4 * g;
            if (R >= a.H || Cc >= a.W) continue;
            float ulo[2 * NVU], uhi[2 * NVU];
            const wl_f4* u4 = reinterpret_cast<const wl_f4*>(U + i * NCS + 4 * g);
#pragma unroll
            for (int u = 0; u < NVU; ++u) {
                const wl_f4 q = u4[u];
                ulo[2 * u] = q.x; uhi[2 * u] = q.y; ulo[2 * u + 1] = q.z; uhi[2 * u + 1] = q.w;
            }
            float y[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int t = 0; t < L0; ++t) y[u] += t0[t] * ulo[u + ME - M0 + t];
#pragma unroll
                for (int t = 0; t < L1; ++t) y[u] += t1[t] * uhi[u + ME - M1 + t];
            }
            T* dst = yp + (size_t)R * a.W + Cc;
            if (Cc + 3 < a.W) {
                WlQuad<T> o; o.a = (T)y[0]; o.b = (T)y[1]; o.c = (T)y[2]; o.d = (T)y[3];
                *reinterpret_cast<WlQuad<T>*>(dst) = o;
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (Cc + u < a.W) dst[u] = (T)y[u];
            }
        }
    }
};

// ---------------------------------------------------------------------------------------------------------
// level >= 2 inverse: inv_j2plus, reference dtcwt/transform_funcs.py:279-307 (colifilt/rowifilt,
// dtcwt/lowlevel.py:154-239).   Y[4q+s] = sum_{t<m2} h_s[e_s+2t] X[o_s - m2 + 2(q+t)],  m2 = L/2:
//   m2 even: (h,e,o)_s = (ha,0,0) (hb,0,1) (ha,1,2) (hb,1,3), highpass o = (1,0,3,2)
//   m2 odd : (h,e,o)_s = (ha,1,1) (hb,1,2) (ha,0,1) (hb,0,2), highpass o = (2,1,2,1)
//   lowpass streams (ll, hl; lo) use (ha,hb) = (g0b,g0a), highpass streams (lh, hh; hi) use (g1b,g1a).
// One workgroup = TH x TW outputs (multiples of 4) from a (TH/2 + 2*m2e) x (TW/2 + 2*m2e) staged block.
// ---------------------------------------------------------------------------------------------------------
template <int L, bool HP>
struct WlIfiltMap {   // compile-time (e, o) of output phase s
    static const int m2 = L / 2;
    static constexpr int e(int s) { return (m2 & 1) ? ((s >> 1) ^ 1) : (s >> 1); }
    static constexpr int o(int s) {
        return (m2 & 1) ? (HP ? ((s & 1) ? 1 : 2) : ((s & 1) ? 2 : 1)) : (HP ? (s ^ 1) : s);
    }
};

template <typename T, int L, int TH_ = 16, int TW_ = 64>
struct WlDtInv2Tile {
    typedef WlDtInv2Args<T> Args;
    static const int kThreads = 256;
    static const int kMinWaves = 2;
    static const int TH = TH_, TW = TW_;
    static const int m2 = L / 2, m2e = m2 + (m2 & 1), D = m2e - m2;
    static const int NRS = TH / 2 + 2 * m2e, NCS = TW / 2 + 2 * m2e;   // staged cells (even)
    static const int WR = 2 * m2 + 2;                                  // window length of one q
    static const int NVU = (D + WR + 1) / 2;                           // ds_read_b128 per row item
    static const int kTapFloats = 4 * L;
    static const int kLdsFloats = kTapFloats + 4 * NRS * NCS + 2 * TH * NCS;
    static_assert(2 * (TW / 4 - 1) + 2 * NVU <= NCS, "row window exceeds the staged width");

    static WL_DEV void run(const Args& a, const WlCtx& ctx) {
        const int tid = ctx.tid;
        const int tiles = a.tiles_x * a.tiles_y;
        const int64_t lbid = wl_xcd_remap(ctx.bid, a.nblocks);   // neighbouring tiles share their halo in one L2
        const int64_t plane = lbid / tiles;
        const int tile = (int)(lbid - plane * tiles);
        const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
        const int R0 = ty * TH, C0 = tx * TW;
        float* lds = reinterpret_cast<float*>(ctx.smem);
        float* tl = lds;   // g0b, g0a, g1b, g1a  (= ha_lp, hb_lp, ha_hp, hb_hp)
        wl_f4* B = reinterpret_cast<wl_f4*>(lds + kTapFloats);
        wl_f2* U = reinterpret_cast<wl_f2*>(lds + kTapFloats + 4 * NRS * NCS);
        if (tid < L) {
            tl[tid] = a.g0b[tid]; tl[L + tid] = a.g0a[tid]; tl[2 * L + tid] = a.g1b[tid]; tl[3 * L + tid] = a.g1a[tid];
        }
        const size_t qplane = (size_t)(a.h / 2) * (a.w / 2);
        {
            WlDtQuadLoaderPlain<T> ld;
            ld.llp = a.ll ? a.ll + (size_t)plane * a.ll_plane_stride : nullptr;
            ld.ll_rs = a.ll_row_stride;
            ld.hp = a.highs ? a.highs + (size_t)plane * 12 * qplane : nullptr;
            ld.qplane = qplane; ld.w2 = a.w / 2;
            wl_dt_stage_quads<T, NRS / 2, NCS / 2, kThreads>(B, tid, R0 / 2 - m2e, C0 / 2 - m2e, a.h, a.w, WL_EXT_SYM, ld);
        }
        ctx.sync();
        float tp[4][L];   // [0] ha_lp [1] hb_lp [2] ha_hp [3] hb_hp
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int t = 0; t < L; ++t) tp[k][t] = tl[k * L + t];
        typedef WlIfiltMap<L, false> LP;
        typedef WlIfiltMap<L, true> HPm;
        // ---- column interpolation: item = (staged column, q) -> output rows 4q .. 4q+3 ---------------------------
        _Pragma("nounroll") for (int f = tid; f < (TH / 4) * NCS; f += kThreads) {
            const int ql = f / NCS, j = f - ql * NCS;
            wl_v2 lh[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};   // (lo, hi) of output rows 4q+s
            const wl_f4* col = B + (2 * ql + D) * NCS + j;
#pragma unroll
            for (int w = 0; w < WR; ++w) {
                const wl_f4 p = col[w * NCS];   // ll, hl, lh, hh
                const wl_v2 pl = {p.x, p.y}, ph = {p.z, p.w};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int tl_ = w - LP::o(s), th_ = w - HPm::o(s);
                    if (tl_ >= 0 && (tl_ & 1) == 0 && tl_ / 2 < m2) lh[s] += pl * tp[s & 1][LP::e(s) + tl_];
                    if (th_ >= 0 && (th_ & 1) == 0 && th_ / 2 < m2) lh[s] += ph * tp[2 + (s & 1)][HPm::e(s) + th_];
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                wl_f2 u; u.x = lh[s].x; u.y = lh[s].y;
                U[(4 * ql + s) * NCS + j] = u;
            }
        }
        ctx.sync();
        // ---- row interpolation + store: item = (row, q) -> 16 contiguous bytes of y -----------------------------------
        const int OH = 2 * a.h, OW = 2 * a.w;
        T* yp = a.y + (size_t)plane * OH * OW;
        _Pragma("nounroll") for (int f = tid; f < TH * (TW / 4); f += kThreads) {
            const int i = f / (TW / 4), ql = f - i * (TW / 4);
            const int R = R0 + i, Cc = C0 + 4 * ql;
            if (R >= OH || Cc >= OW) continue;
            float ulo[2 * NVU], uhi[2 * NVU];
            const wl_f4* u4 = reinterpret_cast<const wl_f4*>(U + i * NCS + 2 * ql);
#pragma unroll
            for (int u = 0; u < NVU; ++u) {
                const wl_f4 q = u4[u];
                ulo[2 * u] = q.x; uhi[2 * u] = q.y; ulo[2 * u + 1] = q.z; uhi[2 * u + 1] = q.w;
            }
            float y[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int t = 0; t < m2; ++t) {
                    y[s] += tp[s & 1][LP::e(s) + 2 * t] * ulo[D + LP::o(s) + 2 * t];
                    y[s] += tp[2 + (s & 1)][HPm::e(s) + 2 * t] * uhi[D + HPm::o(s) + 2 * t];
                }
            }
            WlQuad<T> o; o.a = (T)y[0]; o.b = (T)y[1]; o.c = (T)y[2]; o.d = (T)y[3];
            *reinterpret_cast<WlQuad<T>*>(yp + (size_t)R * OW + Cc) = o;   // OW is a multiple of 4
        }
    }
};
