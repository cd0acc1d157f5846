// Compile-time specialised single-level 2-D DWT analysis tile kernel ("tile v2").
//
// One 256-thread workgroup produces a 16 x 64 output tile (all four sub-bands) of one (n,c) plane:
//   stage : the (2*16+L-2) x (2*64+L-2) input tile is copied to LDS with the boundary extension
//           applied as index math (interior tiles: aligned 8-byte loads; border tiles: per-element);
//   row   : item = (row, q): three ds_read_b128 (16-byte lane stride, conflict free), (lo,hi) of outputs
//           k=2q,2q+1 from v_pk_fma_f32 with the sample broadcast, one ds_write_b128;
//   column: item = (out row, q): L ds_read_b128 = (lo,hi) of columns 2q,2q+1, 4 packed FMAs per tap; every
//           band row leaves as 8 contiguous bytes per lane (512 B per wave) straight into yl / yh[j].
// A workgroup walks a vertical run of tiles; the 8-byte loads of tile t+1 are issued into registers before
// tile t's row bank starts, so ~20 KB per workgroup (four workgroups per CU, ~40 KB LDS each) are in flight
// during all arithmetic.  Out-of-range cells are never loaded: the lane that holds a border sample also
// writes its mirrored / wrapped images, out-of-range ROWS are just other source rows (or zeros).
//
// Restates AFB2D.forward (reference dwt/lowlevel.py:336-347 = afb1d along W, afb1d along H, reshape, two
// .contiguous() copies) for even tap counts; other tap counts use wl_dwt_kernels.h.
#pragma once
#include "wl_common.h"
#include "wl_dwt_stream.h"   // wl_f4 / wl_f2 / wl_v2

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
    int H, W, Kh, Kw;
    int base, ext;
    int tiles_x, tiles_y;
    int vec_ok;       // rows can be read as aligned 8-byte pairs (W even, base pointer 8-byte aligned)
    int run_len;      // tiles (vertically adjacent) per workgroup
    int runs_y;       // ceil(tiles_y / run_len)
};

template <typename T, int LT>
struct WlAfbTile {
    typedef WlAfbTileArgs<T> Args;
    static const int kThreads = 256;
    static const int kMinWaves = 3;
    static const int TH = 16, TW = 64;
    static const int NROWS = 2 * TH + LT - 2;            // staged input rows
    static const int NCOLS = 2 * TW + LT - 2;            // staged input cols actually needed
    static const int NV = (LT + 2 + 3) / 4;              // float4 reads per row item
    static const int NQ = TW / 2;                        // k-pairs per tile row
    static const int SP = 4 * (NQ - 1) + 4 * NV;         // staged row pitch (floats, multiple of 4, >= NCOLS)
    static const int TP = 2 * TW;                        // (lo,hi) row pitch in floats
    static const int kTapFloats = 4 * LT;
    static const int kLdsFloats = kTapFloats + NROWS * SP + NROWS * TP;
    typedef T Pair2 __attribute__((ext_vector_type(2)));
    struct __attribute__((packed, aligned(sizeof(T)), may_alias)) Pair { T a, b; };   // element-aligned pair

    // images of source column s (value v) inside the staged row whose first column is ec0
    static WL_DEV void mirror_cols(float* srow, int s, float v, int W, int ec0, int ext) {
        int e1, e2;
        if (ext == WL_EXT_SYM) { e1 = -1 - s; e2 = 2 * W - 1 - s; }
        else if (ext == WL_EXT_REFL) { e1 = s >= 1 ? -s : -0x40000000; e2 = s <= W - 2 ? 2 * W - 2 - s : -0x40000000; }
        else { e1 = s - W; e2 = s + W; }   // periodic / periodization (even W)
        const int j1 = e1 - ec0, j2 = e2 - ec0;
        if ((unsigned)j1 < (unsigned)SP) srow[j1] = v;
        if ((unsigned)j2 < (unsigned)SP) srow[j2] = v;
    }

    static WL_DEV void run(const Args& a, const WlCtx& ctx) {
        const int tid = ctx.tid;
        const int per_plane = a.tiles_x * a.runs_y;
        const int64_t plane = ctx.bid / per_plane;
        const int rem = (int)(ctx.bid - plane * per_plane);
        const int ry = rem / a.tiles_x, tx = rem - ry * a.tiles_x;
        const int ty_begin = ry * a.run_len;
        const int ty_end = ty_begin + a.run_len < a.tiles_y ? ty_begin + a.run_len : a.tiles_y;
        const int kw0 = tx * TW;
        float* lds = reinterpret_cast<float*>(ctx.smem);
        float* tl = lds;
        float* S = lds + kTapFloats;
        float* Tm = S + NROWS * SP;
        if (tid < LT) {
            tl[2 * tid] = a.h_w_lo[tid]; tl[2 * tid + 1] = a.h_w_hi[tid];
            tl[2 * LT + 2 * tid] = a.h_h_lo[tid]; tl[2 * LT + 2 * tid + 1] = a.h_h_hi[tid];
        }
        const T* xp = a.x + (size_t)plane * a.H * a.W;
        const int ec0 = 2 * kw0 + a.base;
        const bool fast = a.vec_ok && (ec0 & 1) == 0;     // aligned pair loads + mirrored borders
        const bool edge_x = ec0 < 0 || ec0 + SP > a.W;    // this tile column touches the left / right border
        constexpr int NP = SP / 2;                        // 8-byte pairs per staged row
        constexpr int NIT = (NROWS * NP + kThreads - 1) / kThreads;
        Pair2 pf[NIT];

        // issue the loads of tile row `ty` (registers only)
        auto issue = [&](int ty) {
            const int er0 = 2 * ty * TH + a.base;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int f = tid + it * kThreads;
                pf[it] = Pair2{(T)0, (T)0};
                if (f < NROWS * NP) {
                    const int i = f / NP, p = f - i * NP;
                    const int r = wl_ext(er0 + i, a.H, a.ext);
                    const int c = ec0 + 2 * p;
                    if (r >= 0 && c >= 0 && c < a.W)
                        pf[it] = *reinterpret_cast<const Pair2*>(xp + (unsigned)(r * a.W + c));
                }
            }
        };
        // registers -> LDS, plus the border images of the samples this lane holds
        auto commit = [&]() {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int f = tid + it * kThreads;
                if (f < NROWS * NP) {
                    const int i = f / NP, p = f - i * NP;
                    const int c = ec0 + 2 * p;
                    if ((c >= 0 && c < a.W) || a.ext == WL_EXT_ZERO) {   // (out-of-range cells hold zeros)
                        wl_f2 w; w.x = (float)pf[it].x; w.y = (float)pf[it].y;
                        *reinterpret_cast<wl_f2*>(S + i * SP + 2 * p) = w;
                    }
                }
            }
            if (edge_x && a.ext != WL_EXT_ZERO) {
                // border images of the samples this lane just wrote (own LDS writes: program order suffices)
                _Pragma("nounroll") for (int f = tid; f < NROWS * NP; f += kThreads) {
                    const int i = f / NP, p = f - i * NP;
                    const int c = ec0 + 2 * p;
                    if (c >= 0 && c < a.W && (c < SP || c + SP >= a.W)) {
                        float* srow = S + i * SP;
                        mirror_cols(srow, c, srow[2 * p], a.W, ec0, a.ext);
                        mirror_cols(srow, c + 1, srow[2 * p + 1], a.W, ec0, a.ext);
                    }
                }
            }
        };
        // generic staging (odd widths, unaligned bases, odd periodization offsets): per-element extension
        auto stage_slow = [&](int ty) {
            const int er0 = 2 * ty * TH + a.base;
            constexpr int NITS = (NROWS * SP + kThreads - 1) / kThreads;
            constexpr int G = 8;
            _Pragma("nounroll") for (int g0 = 0; g0 < NITS; g0 += G) {
                float v[G];
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const int f = tid + (g0 + u) * kThreads;
                    v[u] = 0.f;
                    if (f < NROWS * SP) {
                        const int i = f / SP, j = f - i * SP;
                        const int r = wl_ext(er0 + i, a.H, a.ext);
                        const int c = wl_ext(ec0 + j, a.W, a.ext);
                        if (r >= 0 && c >= 0) v[u] = (float)xp[(unsigned)(r * a.W + c)];
                    }
                }
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const int f = tid + (g0 + u) * kThreads;
                    if (f < NROWS * SP) S[f] = v[u];
                }
            }
        };

        if (fast) issue(ty_begin);
        for (int ty = ty_begin; ty < ty_end; ++ty) {
            const int kh0 = ty * TH;
            if (fast) commit(); else stage_slow(ty);
            ctx.sync();
            if (fast && ty + 1 < ty_end) issue(ty + 1);
        // ---- row bank -----------------------------------------------------------------------------------------
        {
            wl_v2 tw[LT];
#pragma unroll
            for (int j = 0; j < LT; ++j) { tw[j].x = tl[2 * j]; tw[j].y = tl[2 * j + 1]; }
            _Pragma("nounroll") for (int f = tid; f < NROWS * NQ; f += kThreads) {
                const int i = f / NQ, q = f - i * NQ;
                float v[NV * 4];
                const wl_f4* s4 = reinterpret_cast<const wl_f4*>(S + i * SP) + q;
#pragma unroll
                for (int u = 0; u < NV; ++u) {
                    const wl_f4 t = s4[u];
                    v[4 * u] = t.x; v[4 * u + 1] = t.y; v[4 * u + 2] = t.z; v[4 * u + 3] = t.w;
                }
                wl_v2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
                for (int j = 0; j < LT; ++j) {
                    a0 += tw[j] * v[j];
                    a1 += tw[j] * v[j + 2];
                }
                wl_f4 o;
                o.x = a0.x; o.y = a0.y; o.z = a1.x; o.w = a1.y;
                reinterpret_cast<wl_f4*>(Tm + i * TP)[q] = o;
            }
        }
        ctx.sync();
        // ---- column bank + band stores ---------------------------------------------------------------------------
        {
            wl_v2 th[LT];
#pragma unroll
            for (int j = 0; j < LT; ++j) { th[j].x = tl[2 * LT + 2 * j]; th[j].y = tl[2 * LT + 2 * j + 1]; }
            const unsigned bplane = (unsigned)a.Kh * (unsigned)a.Kw;
            T* llp = a.ll + (size_t)plane * bplane;
            T* hp = a.highs + (size_t)plane * 3 * bplane;
            _Pragma("nounroll") for (int f = tid; f < TH * NQ; f += kThreads) {
                const int kh = f / NQ, q = f - kh * NQ;
                const int k = kh0 + kh, kw = kw0 + 2 * q;
                if (k >= a.Kh || kw >= a.Kw) continue;
                wl_v2 cl0 = {0.f, 0.f}, ch0 = {0.f, 0.f}, cl1 = {0.f, 0.f}, ch1 = {0.f, 0.f};
                const float* col = Tm + (2 * kh) * TP + 4 * q;
#pragma unroll
                for (int j = 0; j < LT; ++j) {
                    const wl_f4 p = *reinterpret_cast<const wl_f4*>(col + j * TP);
                    cl0 += th[j] * p.x; ch0 += th[j] * p.y;
                    cl1 += th[j] * p.z; ch1 += th[j] * p.w;
                }
                const unsigned o = (unsigned)k * (unsigned)a.Kw + (unsigned)kw;
                if (kw + 1 < a.Kw) {
                    Pair p0, p1, p2, p3;
                    p0.a = (T)cl0.x; p0.b = (T)cl1.x;   // LL
                    p1.a = (T)cl0.y; p1.b = (T)cl1.y;   // W-lo / H-hi
                    p2.a = (T)ch0.x; p2.b = (T)ch1.x;   // W-hi / H-lo
                    p3.a = (T)ch0.y; p3.b = (T)ch1.y;   // HH
                    *reinterpret_cast<Pair*>(llp + o) = p0;
                    *reinterpret_cast<Pair*>(hp + o) = p1;
                    *reinterpret_cast<Pair*>(hp + bplane + o) = p2;
                    *reinterpret_cast<Pair*>(hp + 2 * bplane + o) = p3;
                } else {
                    llp[o] = (T)cl0.x;
                    hp[o] = (T)cl0.y;
                    hp[bplane + o] = (T)ch0.x;
                    hp[2 * bplane + o] = (T)ch0.y;
                }
            }
        }
        }   // tile loop (the barrier after the next commit orders this column bank before the next row bank)
    }
};
