// Compile-time specialised single-level 2-D DWT analysis tile kernel ("tile v2").
//
// One 256-thread workgroup produces a 16 x 64 output tile (all four sub-bands) of one (n,c) plane:
//   stage : the (2*16+L-2) x (2*64+L-2) input tile is copied to LDS with the boundary extension
//           applied as index math (interior tiles: aligned 8-byte loads; border tiles: per-element);
//   row   : item = (row, q): three ds_read_b128 (16-byte lane stride, conflict free), (lo,hi) of outputs
//           k=2q,2q+1 from v_pk_fma_f32 with the sample broadcast, one ds_write_b128;
//   column: item = (out row, q): L ds_read_b128 = (lo,hi) of columns 2q,2q+1, 4 packed FMAs per tap; every
//           band row leaves as 8 contiguous bytes per lane (512 B per wave) straight into yl / yh[j].
// A workgroup walks a horizontal run of tiles (the last tile of a row absorbs a remainder of up to XT columns: a
// band is (W+L-2)/2 wide, e.g. 259 = 4 x 64 + 3, and a 3-column tile would cost a whole pipeline step); the 8-byte loads of tile t+1 are issued into registers before
// tile t's row bank starts, so ~20 KB per workgroup (four workgroups per CU, ~40 KB LDS each) are in flight
// during all arithmetic.  Boundary extension costs nothing on the hot path: each lane resolves the source
// columns of its two cells once per workgroup, out-of-range ROWS are just other source rows (or zeros).
//
// Restates AFB2D.forward (reference dwt/lowlevel.py:336-347 = afb1d along W, afb1d along H, reshape, two
// .contiguous() copies) for even tap counts; other tap counts use wl_dwt_kernels.h.
#pragma once
#include "wl_common.h"

template <typename T>
struct WlAfbTileArgs {
    const T* x;       // (NC, H, W)
    T* ll;            // (NC, Kh, Kw)
    T* highs;         // (NC, 3, Kh, Kw)
    const float* h_w_lo;
    const float* h_w_hi;
    const float* h_h_lo;
    const float* h_h_hi;
    int64_t NC;
    int64_t x_ps, ll_ps;   // plane strides of x and ll (elements)
    int x_rs, ll_rs;       // row strides of x and ll (elements; rows are unit-stride): LL_j of the inner levels is
                           //   kept at a cache-line-aligned pitch, only the last level's yl is dense
    int H, W, Kh, Kw;
    int base, ext;
    int tiles_x, tiles_y;
    int vec_ok;       // rows can be read as aligned 8-byte pairs (W even, base pointer 8-byte aligned)
    int run_len;      // tiles (horizontally adjacent) per workgroup
    int runs_x;       // ceil(tiles_x / run_len)
    int64_t nblocks;  // grid size (for the XCD-aware block remap)
};

// V4_ = 1: the input tile is staged with FOUR-element loads (used for float16: 8 instead of 4 bytes per lane); needs
// W % 4 == 0, an aligned base pointer and the staged origin (2*kw0 + base - SH_) on a multiple of four columns.
template <typename T, int LT, int TH_ = 16, int TW_ = 64, int SH_ = 0, int V4_ = 0>
struct WlAfbTile {
    typedef WlAfbTileArgs<T> Args;
    static const int kThreads = 256;
    // long windows need more than the 168 registers of three waves per SIMD (they spilled)
    static const int kMinWaves = (LT >= 20 || (LT >= 16 && sizeof(T) == 4)) ? 2 : 3;
    static const int TH = TH_, TW = TW_;
    static const int NROWS = 2 * TH + LT - 2;            // staged input rows
    static const int NCOLS = 2 * TW + LT - 2;            // staged input cols actually needed
    static const int SH = SH_;                           // 1: odd `base` (periodization) - the staged origin is moved one
                                                         //    column left so that lanes still read aligned pairs
    static const int NV = (LT + 2 + SH + 3) / 4;         // float4 reads per row item
    static const int NQ = TW / 2;                        // k-pairs per regular tile row
    // the LAST tile of a row may be up to XT columns wider, so the L/2-1 excess columns of a band do not cost a
    // tile - unless the wider LDS rows would push the kernel from three to two workgroups per CU (long filters)
    static constexpr int lds_bytes(int xt) {
        return 4 * (4 * LT + NROWS * (4 * (TW / 2 + xt / 2 - 1) + 4 * NV) + NROWS * 2 * (TW + xt));
    }
    static const int XT = lds_bytes(8) <= 160 * 1024 / 3 ? 8 : 0;
    static const int NQX = XT / 2;
    static const int NQXD = NQX > 0 ? NQX : 1;           // divisor for the extra-pair loops (dead when XT == 0)
    static const int SP = 4 * (NQ + NQX - 1) + 4 * NV;   // staged row pitch (floats, multiple of 4)
    static const int TP = 2 * (TW + XT);                 // (lo,hi) row pitch in floats
    static const int kTapFloats = 4 * LT;
    static const int kLdsFloats = kTapFloats + NROWS * SP + NROWS * TP;
    typedef T Pair2 __attribute__((ext_vector_type(2)));
    typedef T Quad4 __attribute__((ext_vector_type(4)));
    static const int V4 = V4_;
    struct __attribute__((packed, aligned(sizeof(T)), may_alias)) Pair { T a, b; };   // element-aligned pair

    static WL_DEV void run(const Args& a, const WlCtx& ctx) {
        const int tid = ctx.tid;
        // a workgroup walks a HORIZONTAL run of tiles (same tile row, consecutive tile columns): whole output
        // rows are then written by one workgroup, so cache lines are not split between L2s of different XCDs
        const int per_plane = a.tiles_y * a.runs_x;
        const int64_t lbid = wl_xcd_remap(ctx.bid, a.nblocks);
        const int64_t plane = lbid / per_plane;
        const int rem = (int)(lbid - plane * per_plane);
        const int ty = rem / a.runs_x, rx = rem - ty * a.runs_x;
        const int tx_begin = rx * a.run_len;
        const int tx_end = tx_begin + a.run_len < a.tiles_x ? tx_begin + a.run_len : a.tiles_x;
        const int kh0 = ty * TH;
        float* lds = reinterpret_cast<float*>(ctx.smem);
        float* tl = lds;
        float* S = lds + kTapFloats;
        float* Tm = S + NROWS * SP;
        if (tid < LT) {
            tl[2 * tid] = a.h_w_lo[tid]; tl[2 * tid + 1] = a.h_w_hi[tid];
            tl[2 * LT + 2 * tid] = a.h_h_lo[tid]; tl[2 * LT + 2 * tid + 1] = a.h_h_hi[tid];
        }
        const T* xp = a.x + (size_t)plane * a.x_ps;
        constexpr int NP = V4 ? SP / 4 : SP / 2;          // staged cells (pairs, or quads) per row
        constexpr int RPI = kThreads / NP;                // staged rows per iteration (lanes: RPI x NP)
        constexpr int NIT = (NROWS + RPI - 1) / RPI;
        Pair2 pf[V4 ? 1 : NIT];
        Quad4 pq[V4 ? NIT : 1];
        // Rows: this tile row needs input rows er0 .. er0+nr_need-1; the lane owns staged rows s_row, s_row+RPI, ..
        // whose SOURCE rows under the boundary extension are resolved once per workgroup.
        const int s_row = tid / NP, p_own = tid - s_row * NP;
        const int er0 = 2 * kh0 + a.base;
        const int nrows_out = (a.Kh - kh0) < TH ? (a.Kh - kh0) : TH;
        const int nr_need = 2 * nrows_out + LT - 2;
        int rsrc[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = it * RPI + s_row;
            rsrc[it] = (s_row < RPI && i < nr_need) ? wl_ext(er0 + i, a.H, a.ext) : -1;
        }
        // per-tile state (set by issue): which cells this lane stages for the tile being prefetched
        int nq_need = 0;

        // issue the loads of tile column `tx` (registers only).  A border cell is simply loaded from its source
        // column (an L1/L2 hit), so every mode, odd widths and multiple reflections need no special path.
        auto issue = [&](int tx) {
            if (V4) {
                const int kw0 = tx * TW;
                const int ec0 = 2 * kw0 + a.base - SH;                 // multiple of 4
                const int ncols_out = tx == a.tiles_x - 1 ? a.Kw - kw0 : TW;
                const int nq = (ncols_out + 1) / 2;
                const int nq4 = (nq * 2 + NV * 2 - 2 + 1) / 2;         // staged quads per row actually needed
                const bool lane_on = s_row < RPI && p_own < (nq4 < NP ? nq4 : NP);
                int cs[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) cs[e] = lane_on ? wl_ext(ec0 + 4 * p_own + e, a.W, a.ext) : -1;
                const bool quad_ld = cs[0] >= 0 && (cs[0] & 3) == 0 && cs[3] == cs[0] + 3;
                const int cq = quad_ld ? cs[0] : -1;
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    pq[it] = Quad4{(T)0, (T)0, (T)0, (T)0};
                    const int r = rsrc[it];
                    if ((cq | r) >= 0) pq[it] = *reinterpret_cast<const Quad4*>(xp + r * a.x_rs + cq);
                }
                if (lane_on && !quad_ld) {   // border lanes: element loads from the source columns
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        const int r = rsrc[it];
                        if (r >= 0) {
                            const T* src = xp + r * a.x_rs;
                            if (cs[0] >= 0) pq[it].x = src[cs[0]];
                            if (cs[1] >= 0) pq[it].y = src[cs[1]];
                            if (cs[2] >= 0) pq[it].z = src[cs[2]];
                            if (cs[3] >= 0) pq[it].w = src[cs[3]];
                        }
                    }
                }
                return;
            }
            const int kw0 = tx * TW;
            const int ec0 = 2 * kw0 + a.base - SH;
            const int ncols_out = tx == a.tiles_x - 1 ? a.Kw - kw0 : TW;
            const int nq = (ncols_out + 1) / 2;
            const int np_need = nq * 2 + NV * 2 - 2 < NP ? nq * 2 + NV * 2 - 2 : NP;   // staged pairs per row
            const bool lane_on = s_row < RPI && p_own < np_need;
            const int cs0 = lane_on ? wl_ext(ec0 + 2 * p_own, a.W, a.ext) : -1;
            const int cs1 = lane_on ? wl_ext(ec0 + 2 * p_own + 1, a.W, a.ext) : -1;
            const bool pair_ld = a.vec_ok && cs0 >= 0 && cs1 == cs0 + 1 && (cs0 & 1) == 0;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                pf[it] = Pair2{(T)0, (T)0};
                const int r = rsrc[it];
                if (lane_on && r >= 0) {
                    const T* src = xp + r * a.x_rs;
                    if (pair_ld) pf[it] = *reinterpret_cast<const Pair2*>(src + cs0);
                    else {
                        if (cs0 >= 0) pf[it].x = src[cs0];
                        if (cs1 >= 0) pf[it].y = src[cs1];
                    }
                }
            }
        };
        // registers -> LDS (cells outside the needed part of a partial tile are written too: zeros)
        auto commit = [&]() {
            if (s_row >= RPI) return;
            if (V4) {
                float* d4 = S + s_row * SP + 4 * p_own;
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    if (it * RPI + s_row < NROWS) {
                        wl_f4 w; w.x = (float)pq[it].x; w.y = (float)pq[it].y; w.z = (float)pq[it].z; w.w = (float)pq[it].w;
                        *reinterpret_cast<wl_f4*>(d4 + it * RPI * SP) = w;
                    }
                }
                return;
            }
            float* d = S + s_row * SP + 2 * p_own;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                if (it * RPI + s_row < NROWS) {
                    wl_f2 w; w.x = (float)pf[it].x; w.y = (float)pf[it].y;
                    *reinterpret_cast<wl_f2*>(d + it * RPI * SP) = w;
                }
            }
        };

        // Software pipeline (gfx9 has ONE in-order counter for loads and stores, so a wait for loads also waits
        // for every store issued before it): tile t+1 is committed to LDS between the two banks of tile t, i.e.
        // the wait sits a whole row bank after the youngest stores, and the loads of tile t+2 are issued right
        // after it - they fly during the column bank of t and the row bank of t+1.
        issue(tx_begin);
        commit();
        ctx.sync();
        if (tx_begin + 1 < tx_end) issue(tx_begin + 1);
        for (int tx = tx_begin; tx < tx_end; ++tx) {
            const int kw0 = tx * TW;
            {
                const int ncols_out = tx == a.tiles_x - 1 ? a.Kw - kw0 : TW;
                nq_need = (ncols_out + 1) / 2;
            }
        // ---- row bank -----------------------------------------------------------------------------------------
        {
            wl_v2 tw[LT];
#pragma unroll
            for (int j = 0; j < LT; ++j) { tw[j].x = tl[2 * j]; tw[j].y = tl[2 * j + 1]; }
            auto row_item = [&](int i, int q) {
                float v[NV * 4];
                const wl_vf4* s4 = reinterpret_cast<const wl_vf4*>(S + i * SP) + q;
#pragma unroll
                for (int u = 0; u < NV; ++u) {
                    const wl_vf4 t = s4[u];
                    v[4 * u] = t.x; v[4 * u + 1] = t.y; v[4 * u + 2] = t.z; v[4 * u + 3] = t.w;
                }
                wl_v2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
                for (int j = 0; j < LT; ++j) {
                    a0 += tw[j] * v[j + SH];
                    a1 += tw[j] * v[j + 2 + SH];
                }
                wl_f4 o;
                o.x = a0.x; o.y = a0.y; o.z = a1.x; o.w = a1.y;
                reinterpret_cast<wl_f4*>(Tm + i * TP)[q] = o;
            };
            _Pragma("nounroll") for (int f = tid; f < nr_need * NQ; f += kThreads) {
                const int i = f / NQ, q = f - i * NQ;
                if (q < nq_need) row_item(i, q);
            }
            if (NQX > 0 && nq_need > NQ) {   // the wider last tile: pairs NQ .. nq_need-1
                _Pragma("nounroll") for (int f = tid; f < nr_need * NQX; f += kThreads) {
                    const int i = f / NQXD, q = NQ + (f - i * NQXD);
                    if (q < nq_need) row_item(i, q);
                }
            }
        }
        ctx.sync();
        if (tx + 1 < tx_end) {
            commit();                                  // tile tx+1 -> S (the row bank above was its last reader)
            if (tx + 2 < tx_end) issue(tx + 2);
        }
        // ---- column bank + band stores ---------------------------------------------------------------------------
        {
            wl_v2 th[LT];
#pragma unroll
            for (int j = 0; j < LT; ++j) { th[j].x = tl[2 * LT + 2 * j]; th[j].y = tl[2 * LT + 2 * j + 1]; }
            const unsigned bplane = (unsigned)a.Kh * (unsigned)a.Kw;
            T* llp = a.ll + (size_t)plane * a.ll_ps;
            T* hp = a.highs + (size_t)plane * 3 * bplane;
            auto col_item = [&](int kh, int q) {
                const int k = kh0 + kh, kw = kw0 + 2 * q;
                if (k >= a.Kh || kw >= a.Kw) return;
                wl_v2 cl0 = {0.f, 0.f}, ch0 = {0.f, 0.f}, cl1 = {0.f, 0.f}, ch1 = {0.f, 0.f};
                const float* col = Tm + (2 * kh) * TP + 4 * q;
#pragma unroll
                for (int j = 0; j < LT; ++j) {
                    const wl_vf4 p = *reinterpret_cast<const wl_vf4*>(col + j * TP);   // one 128-bit load (a struct of four floats is
                                                                                       // split and re-paired into 8-byte reads with a 2-way bank conflict)
                    cl0 += th[j] * p.x; ch0 += th[j] * p.y;
                    cl1 += th[j] * p.z; ch1 += th[j] * p.w;
                }
                const unsigned o = (unsigned)k * (unsigned)a.Kw + (unsigned)kw;
                const unsigned ol = (unsigned)k * (unsigned)a.ll_rs + (unsigned)kw;
                if (kw + 1 < a.Kw) {
                    Pair p0, p1, p2, p3;
                    p0.a = (T)cl0.x; p0.b = (T)cl1.x;   // LL
                    p1.a = (T)cl0.y; p1.b = (T)cl1.y;   // W-lo / H-hi
                    p2.a = (T)ch0.x; p2.b = (T)ch1.x;   // W-hi / H-lo
                    p3.a = (T)ch0.y; p3.b = (T)ch1.y;   // HH
                    *reinterpret_cast<Pair*>(llp + ol) = p0;
                    *reinterpret_cast<Pair*>(hp + o) = p1;
                    *reinterpret_cast<Pair*>(hp + bplane + o) = p2;
                    *reinterpret_cast<Pair*>(hp + 2 * bplane + o) = p3;
                } else {
                    llp[ol] = (T)cl0.x;
                    hp[o] = (T)cl0.y;
                    hp[bplane + o] = (T)ch0.x;
                    hp[2 * bplane + o] = (T)ch0.y;
                }
            };
            _Pragma("nounroll") for (int f = tid; f < TH * NQ; f += kThreads) {
                const int kh = f / NQ;
                col_item(kh, f - kh * NQ);
            }
            if (NQX > 0 && nq_need > NQ) {
                _Pragma("nounroll") for (int f = tid; f < TH * NQX; f += kThreads) {
                    const int kh = f / NQXD;
                    col_item(kh, NQ + (f - kh * NQXD));
                }
            }
        }
        ctx.sync();   // S(tx+1) visible; Tm free for the next row bank
        }   // tile loop
    }
};
