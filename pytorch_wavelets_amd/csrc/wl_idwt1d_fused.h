// Multi-level 1-D DWT synthesis in ONE launch: DWT1DInverse.forward's level loop (reference dwt/transform1d.py:97-115 = J x
// SFB1D.forward, dwt/lowlevel.py:697-727 -> sfb1d :226-271, incl. the 'unpad' of a lowpass one sample longer than the next
// highpass) and the backward of the 1-D analysis (AFB1D.backward :409-424, with the analysis taps and the crop to the input
// length) for signals along the last axis of dense (rows, n) tensors.  One level on the single-axis kernel (wl_synth1d) reads
// its two inputs through the caches and round-trips the reconstructed lowpass: J = 3 ran at 0.075 of the HBM roofline
// (64 x 16 x 65536 float32).  Here a workgroup owns a chunk of `chunk` OUTPUT samples of a row (a multiple of 2^J) and keeps the
// levels in LDS.  Per axis, as wl_synth1d evaluates it (every mode but periodization):
//     y[p] = sum_k g0[p + L - 2 - 2k] lo[k] + g1[p + L - 2 - 2k] hi[k]   =>   the output pair (2q, 2q + 1) reads k = q .. q + L/2 - 1:
//     (y[2q], y[2q+1]) = sum_{i < L/2} (g[L-2-2i], g[L-1-2i]) c[q + i]          - one packed FMA per coefficient and bank.
// Level j needs coefficients [base_j, base_j + span_j) with base_j = c chunk / 2^j and span_j = span_{j-1} / 2 + L/2 - 1 (rounded
// up to even): all highpass ranges and the coarsest lowpass range are loaded into LDS first (zeros beyond a level's end: no tap
// ever tests a boundary), then level J .. 1 run out of LDS, each writing the next finer level's lowpass (zeros from that level's
// highpass length on: the 'unpad') - and level 1 the output.  HBM traffic = every coefficient once (+ the (L/2 - 1)-coefficient
// halos per level and chunk) + every output once.
#pragma once
#include "wl_common.h"
#include "wl_dwt_rows.h"   // wl_pk_fma_x, wl_uniform_v2

#define WL_IDWT1D_MAXJ 4
#ifndef WL_IDWT1D_CHUNK
#define WL_IDWT1D_CHUNK 4096   // output samples per chunk
#endif

template <typename T>
struct WlIdwt1dArgs {
    const T* lo;                       // (rows, n_lo) dense: the coarsest lowpass (n_lo = n_hi[J-1] or one more)
    const T* hi[WL_IDWT1D_MAXJ];       // (rows, n_hi[j]) dense, finest first; nullptr = zeros
    T* y;                              // (rows, out_len) dense
    const float* g0; const float* g1;  // stored synthesis taps, L each
    int64_t rows, nblocks;
    int J, n_lo, out_len, chunk, nchunks;
    int n_hi[WL_IDWT1D_MAXJ];
    int span[WL_IDWT1D_MAXJ + 1];      // span[0] = chunk, span[j] = coefficients of level j a chunk needs (even)
    int hi_off[WL_IDWT1D_MAXJ], lo_off[WL_IDWT1D_MAXJ];   // LDS byte offsets of level j + 1's highpass / lowpass buffers
    int lds_bytes;
};

template <typename T, int LT>
struct WlIdwt1dFused {
    typedef WlIdwt1dArgs<T> Args;
    static const int kThreads = 256;
    static const int kMinWaves = 2;
    static const int HL = LT / 2;
    // n floats of src[base ..] -> dst, zeros from position `len` on (src may be null = zeros)
    // (n is even and dst 8-byte aligned: two samples per lane and load - element-aligned 8-byte loads - while both exist)
    typedef T Pair2 __attribute__((ext_vector_type(2), aligned(sizeof(T)), may_alias));
    static WL_DEV void load_range(float* dst, const T* src, int base, int n, int len, int tid) {
        // (round 6: four samples per lane and load - element-aligned 16-byte loads, 16-byte LDS writes - while all four exist; the
        // buffers are 16-byte aligned and padded to whole 16-byte groups by the launcher)
        typedef T Quad4 __attribute__((ext_vector_type(4), aligned(sizeof(T)), may_alias));
        for (int i = 4 * tid; i < n; i += 4 * kThreads) {
            const int k = base + i;
            wl_vf4 v; v.x = v.y = v.z = v.w = 0.f;
            if (src && k + 3 < len) {
                const Quad4 t = *reinterpret_cast<const Quad4*>(src + k);
                v.x = (float)t.x; v.y = (float)t.y; v.z = (float)t.z; v.w = (float)t.w;
            } else if (src) {
                if (k < len) v.x = (float)src[k];
                if (k + 1 < len) v.y = (float)src[k + 1];
                if (k + 2 < len) v.z = (float)src[k + 2];
            }
            *reinterpret_cast<wl_vf4*>(dst + i) = v;
        }
    }
    static WL_DEV void run(const Args& a, const WlCtx& ctx) {
        const int tid = ctx.tid;
        const int64_t row = ctx.bid / a.nchunks;
        const int c = (int)(ctx.bid - row * a.nchunks);
        const int J = a.J;
        wl_v2 p0[HL], p1[HL];                                    // (g[L-2-2i], g[L-1-2i]) of either bank: scalar registers
#pragma unroll
        for (int i = 0; i < HL; ++i) {
            p0[i] = wl_uniform_v2(wl_v2{a.g0[LT - 2 - 2 * i], a.g0[LT - 1 - 2 * i]});
            p1[i] = wl_uniform_v2(wl_v2{a.g1[LT - 2 - 2 * i], a.g1[LT - 1 - 2 * i]});
        }
        // ---- every coefficient the chunk needs
        for (int j = 0; j < J; ++j) {
            const int base = (c * a.chunk) >> (j + 1);
            float* hb = reinterpret_cast<float*>(ctx.smem + a.hi_off[j]);
            load_range(hb, a.hi[j] ? a.hi[j] + (size_t)row * a.n_hi[j] : nullptr, base, a.span[j + 1], a.n_hi[j], tid);
            if (j == J - 1) {
                float* lb = reinterpret_cast<float*>(ctx.smem + a.lo_off[j]);
                // (the coarsest lowpass may be one sample longer than its highpass: the surplus sample is dropped)
                load_range(lb, a.lo + (size_t)row * a.n_lo, base, a.span[j + 1], a.n_hi[j] < a.n_lo ? a.n_hi[j] : a.n_lo, tid);
            }
        }
        ctx.sync();
        // ---- levels J .. 1
        for (int j = J - 1; j >= 0; --j) {
            const float* lb = reinterpret_cast<const float*>(ctx.smem + a.lo_off[j]);
            const float* hb = reinterpret_cast<const float*>(ctx.smem + a.hi_off[j]);
            const int npairs = a.span[j] >> 1;                   // output pairs of this level
            const int obase = (c * a.chunk) >> j;                // position of output 0 in the level's output signal
            float* nxt = j > 0 ? reinterpret_cast<float*>(ctx.smem + a.lo_off[j - 1]) : nullptr;
            const int lim = j > 0 ? a.n_hi[j - 1] : a.out_len;   // outputs from here on are dropped ('unpad' / crop)
            T* const yp = a.y + (size_t)row * a.out_len;
            if (!nxt) {
                // level 1 -> y: TWO output pairs per lane, one element-aligned 16-byte store (round 6: half the store instructions)
                typedef T Quad4 __attribute__((ext_vector_type(4), aligned(sizeof(T)), may_alias));
                for (int q = 2 * tid; q < npairs; q += 2 * kThreads) {
                    wl_v2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f}, b0 = {0.f, 0.f}, b1 = {0.f, 0.f};
                    float sl = lb[q], sh = hb[q];
#pragma unroll
                    for (int i = 0; i < HL; ++i) {
                        const float nl = lb[q + i + 1], nh = hb[q + i + 1];   // (one group of padding behind every buffer: readable)
                        wl_pk_fma_x(a0, p0[i], wl_v2{sl, 0.f}); wl_pk_fma_x(a1, p1[i], wl_v2{sh, 0.f});
                        wl_pk_fma_x(b0, p0[i], wl_v2{nl, 0.f}); wl_pk_fma_x(b1, p1[i], wl_v2{nh, 0.f});
                        sl = nl; sh = nh;
                    }
                    const float v0 = a0.x + a1.x, v1 = a0.y + a1.y, v2 = b0.x + b1.x, v3 = b0.y + b1.y;
                    const int p = obase + 2 * q;
                    if (q + 1 < npairs && p + 3 < lim) {
                        Quad4 t; t.x = (T)v0; t.y = (T)v1; t.z = (T)v2; t.w = (T)v3;
                        *reinterpret_cast<Quad4*>(yp + p) = t;
                    } else {
                        if (p < lim) yp[p] = (T)v0;
                        if (p + 1 < lim) yp[p + 1] = (T)v1;
                        if (q + 1 < npairs && p + 2 < lim) yp[p + 2] = (T)v2;
                        if (q + 1 < npairs && p + 3 < lim) yp[p + 3] = (T)v3;
                    }
                }
                ctx.sync();
                continue;
            }
            for (int q = tid; q < npairs; q += kThreads) {
                wl_v2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
#pragma unroll
                for (int i = 0; i < HL; ++i) {
                    const wl_v2 sl = {lb[q + i], 0.f}, sh = {hb[q + i], 0.f};
                    wl_pk_fma_x(acc0, p0[i], sl);
                    wl_pk_fma_x(acc1, p1[i], sh);
                }
                const float v0 = acc0.x + acc1.x, v1 = acc0.y + acc1.y;
                const int p = obase + 2 * q;
                if (nxt) {
                    wl_f2 t; t.x = p < lim ? v0 : 0.f; t.y = p + 1 < lim ? v1 : 0.f;
                    *reinterpret_cast<wl_f2*>(nxt + 2 * q) = t;
                } else if (p + 1 < lim) {                        // (one element-aligned 8-byte store per lane)
                    Pair2 t; t.x = (T)v0; t.y = (T)v1;
                    *reinterpret_cast<Pair2*>(yp + p) = t;
                } else if (p < lim) yp[p] = (T)v0;
            }
            ctx.sync();
        }
    }
};
