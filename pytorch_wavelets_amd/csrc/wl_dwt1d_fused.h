// Multi-level 1-D DWT analysis in ONE launch: DWT1DForward.forward's level loop (reference dwt/transform1d.py:44-59 = J x
// AFB1D.forward, dwt/lowlevel.py:368-424 -> afb1d :91-172) for signals along the last axis of a dense (rows, N) tensor.
// The per-level path (wl_corr1d) reads every level's input from memory L times through the caches and writes / re-reads the
// intermediate lowpass signals: 0.17 of the HBM roofline at 64 x 16 x 65536 float32.  Here a workgroup owns one (row, chunk
// of the coarsest level's outputs) and keeps everything in LDS:
//   * the input samples the chunk depends on (its own 2^J x chunk samples + (L - 2)(2^J - 1) + ... halo samples either side,
//     fetched through the boundary-extension rule at the signal's ends) are loaded ONCE, as 16-byte groups, converted to
//     float32;
//   * level j + 1 is computed from level j's LDS buffer: a thread owns output positions k, k + 256, ..; its L samples are L/2
//     aligned 8-byte LDS reads (consecutive lanes 8 bytes apart: conflict-free), every tap one packed FMA on the (lo, hi)
//     pair with the tap pair in scalar registers;  the lowpass goes to the next level's LDS buffer, the highpass (and the
//     coarsest lowpass) of the positions the chunk OWNS to memory, consecutive lanes on consecutive addresses;
//   * positions beyond a level's ends are read through the extension rule of the mode (zero / symmetric / reflect: the
//     mirrored source positions lie inside the chunk that holds the end; periodic / periodization wrap to the other end, so
//     those modes are taken only when one chunk holds the whole row).
// HBM traffic = every input sample once (+ the halo, ~1 % at 8192-sample chunks) + every output once.
#pragma once
#include "wl_common.h"
#include "wl_dwt_rows.h"   // wl_pk_fma_x / _y, wl_uniform_v2

#define WL_DWT1D_MAXJ 4
#ifndef WL_DWT1D_UNROLL
#define WL_DWT1D_UNROLL 1      // output positions a thread works on at a time (A/B builds)
#endif
#ifndef WL_DWT1D_STAGES
#define WL_DWT1D_STAGES 1      // chunks whose input samples are in flight (in registers) ahead of the one being computed
#endif
#ifndef WL_DWT1D_SPAN
#define WL_DWT1D_SPAN 4096     // input samples per chunk
#endif

template <typename T>
struct WlDwt1dArgs {
    const T* x;                        // (rows, N) dense
    T* lo;                             // (rows, n[J]) dense: the coarsest lowpass
    T* hi[WL_DWT1D_MAXJ];              // (rows, n[j + 1]) dense: highpass of level j + 1
    const float* h0; const float* h1;  // stored (reversed) taps
    int64_t rows, nblocks;
    int J, ext;
    int n[WL_DWT1D_MAXJ + 1];          // n[0] = N, n[j] = coefficients of level j
    int base[WL_DWT1D_MAXJ];           // level j + 1: y[k] = sum_t h[t] ext(x_j, 2 k + base[j] + t)
    int chunk, nchunks;                // coarsest outputs per chunk, chunks per row
    int cpw, ngroups;                  // consecutive chunks per workgroup, workgroups per row
    int buf_off[WL_DWT1D_MAXJ];        // LDS byte offset of level j's buffer (j = 0: the input)
    int lds_bytes;
};

template <typename T, int LT>
struct WlDwt1dFused {
    typedef WlDwt1dArgs<T> Args;
    static const int kThreads = 256;
    static const int kMinWaves = 2;
    static const int SZ = (int)sizeof(T);

    // ranges of one chunk: level j = 0 .. J; [clo, chi) = positions held in LDS (computed, or loaded for j = 0: may extend
    // beyond [0, n) there, those cells hold extension samples), [olo, ohi) = positions whose outputs this chunk stores,
    // org = position of LDS cell 0 (a multiple of 4, <= clo)
    struct Geo { int clo[WL_DWT1D_MAXJ + 1], chi[WL_DWT1D_MAXJ + 1], olo[WL_DWT1D_MAXJ + 1], ohi[WL_DWT1D_MAXJ + 1], org[WL_DWT1D_MAXJ + 1]; };
    static WL_HD int floor4(int v) { return v >= 0 ? v & ~3 : -((-v + 3) & ~3); }
    // level j of the chunk (j = MAXJ .. 1, template recursion: every index into the local arrays is a compile-time
    // constant - indexed by a run-time level they would live in scratch memory)
    template <int j> static WL_HD void geo_level(const Args& a, Geo& g, int k0, bool last) {
        if constexpr (j >= 1) {
            // (every element of g is assigned at ONE place with a compile-time index: assigned in both arms of a branch the
            // compiler merged the stores into one with a run-time offset and the arrays went to scratch memory)
            const int J = a.J;
            int clo = 0, chi = 0, olo = 0, ohi = 0, org = 0;
            if (j <= J) {
                const int sh = J - j;
                int lo = k0 << sh, hi = last ? a.n[j] : ((k0 + a.chunk) << sh);
                if (lo > a.n[j]) lo = a.n[j];
                if (hi > a.n[j]) hi = a.n[j];
                olo = lo; ohi = hi;
                clo = lo; chi = hi;
                if constexpr (j < WL_DWT1D_MAXJ) {
                    if (j < J) {
                        // what level j + 1 reads of level j, inside the signal
                        int ilo = 2 * g.clo[j + 1] + a.base[j], ihi = 2 * (g.chi[j + 1] - 1) + a.base[j] + LT;
                        if (g.chi[j + 1] <= g.clo[j + 1]) { ilo = lo; ihi = hi; }
                        if (ilo < 0) ilo = 0;
                        if (ihi > a.n[j]) ihi = a.n[j];
                        clo = lo < ilo ? lo : ilo;
                        chi = hi > ihi ? hi : ihi;
                        if (a.nchunks == 1) { clo = 0; chi = a.n[j]; }
                    }
                }
                org = floor4(clo);
            }
            g.clo[j] = clo; g.chi[j] = chi; g.olo[j] = olo; g.ohi[j] = ohi; g.org[j] = org;
            geo_level<j - 1>(a, g, k0, last);
        }
    }
    static WL_HD Geo geometry(const Args& a, int c) {
        Geo g;
        geo_level<WL_DWT1D_MAXJ>(a, g, c * a.chunk, c == a.nchunks - 1);
        // the input: extended positions level 1 reads
        g.olo[0] = g.ohi[0] = 0;
        g.clo[0] = 2 * g.clo[1] + a.base[0];
        g.chi[0] = 2 * (g.chi[1] - 1) + a.base[0] + LT;
        if (g.chi[1] <= g.clo[1]) g.chi[0] = g.clo[0];
        g.org[0] = floor4(g.clo[0]);
        return g;
    }

    typedef T Quad4 __attribute__((ext_vector_type(4), aligned(sizeof(T)), may_alias));

    // the input samples of a chunk on their way from memory to LDS buffer 0: whole 4-sample groups inside the signal as
    // vector loads (MAXV per thread), the rest (the signal's ends through the extension rule, the ragged edges of the range)
    // one sample per thread
    static const int MAXV = (WL_DWT1D_SPAN + 1023) / 1024 + 1;
    static const int NST = WL_DWT1D_STAGES;
    struct Stage { Quad4 q[MAXV]; T s; };
    // real = false: behind the workgroup's last chunk - the same number of loads (the compiler counts them to wait for exactly
    // the oldest stage), all of the row's first group: one cache line
    static WL_DEV void issue(const Args& a, const Geo& g, const T* xr, int tid, Stage& st, bool real) {
        const int N = a.n[0];
        int v_lo = g.clo[0] < 0 ? 0 : (g.clo[0] + 3) & ~3;
        int v_hi = g.chi[0] > N ? N & ~3 : g.chi[0] & ~3;
        if (v_hi < v_lo) v_hi = v_lo;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int p = v_lo + 4 * (tid + kThreads * i);
            st.q[i] = *reinterpret_cast<const Quad4*>(xr + (real && p < v_hi ? p : 0));   // (off lanes: a group that exists)
        }
        const int n_head = v_lo - g.clo[0], n_tail = g.chi[0] - v_hi;
        const int p = tid < n_head ? g.clo[0] + tid : v_hi + (tid - n_head);
        const int sp = tid < n_head + n_tail ? wl_ext(p, N, a.ext) : 0;
        st.s = xr[sp < 0 || !real ? 0 : sp];
    }
    static WL_DEV void commit(const Args& a, const Geo& g, const WlCtx& ctx, int tid, const Stage& st) {
        float* const b0 = reinterpret_cast<float*>(ctx.smem + a.buf_off[0]);
        const int N = a.n[0];
        int v_lo = g.clo[0] < 0 ? 0 : (g.clo[0] + 3) & ~3;
        int v_hi = g.chi[0] > N ? N & ~3 : g.chi[0] & ~3;
        if (v_hi < v_lo) v_hi = v_lo;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int p = v_lo + 4 * (tid + kThreads * i);
            if (p < v_hi) {
                wl_vf4 w; w.x = (float)st.q[i].x; w.y = (float)st.q[i].y; w.z = (float)st.q[i].z; w.w = (float)st.q[i].w;
                *reinterpret_cast<wl_vf4*>(b0 + (p - g.org[0])) = w;
            }
        }
        const int n_head = v_lo - g.clo[0], n_tail = g.chi[0] - v_hi;
        if (tid < n_head + n_tail) {
            const int p = tid < n_head ? g.clo[0] + tid : v_hi + (tid - n_head);
            b0[p - g.org[0]] = wl_ext(p, N, a.ext) < 0 ? 0.f : (float)st.s;
        }
    }

    static WL_DEV void run(const Args& a, const WlCtx& ctx) {
        const int tid = ctx.tid;
        const int64_t row = ctx.bid / a.ngroups;
        const int grp = (int)(ctx.bid - row * a.ngroups);
        const int c0 = grp * a.cpw, c1 = c0 + a.cpw < a.nchunks ? c0 + a.cpw : a.nchunks;
        wl_v2 tp[LT];                                           // (h0[t], h1[t]): scalar registers
#pragma unroll
        for (int t = 0; t < LT; ++t) tp[t] = wl_uniform_v2(wl_v2{a.h0[t], a.h1[t]});
        const T* const xr = a.x + (size_t)row * a.n[0];
        // a workgroup streams over its chunks: while the levels of chunk c run out of LDS, the input samples of chunk c + 1 are
        // already on their way into registers (the first version - one chunk per workgroup, load, then compute - spent most of
        // its time waiting for memory: 0.27 of the roofline)
        // NST register stages: the samples of chunks c + 1 .. c + NST are in flight while chunk c is computed (one stage was
        // not enough: the levels of a chunk take less time than a load under load, 0.40 of the roofline)
        Stage st[NST];
        Geo gq[NST];
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            gq[u] = geometry(a, c0 + u < c1 ? c0 + u : c1 - 1);
            issue(a, gq[u], xr, tid, st[u], c0 + u < c1);
        }
        for (int cb = c0; cb < c1; cb += NST) {
#pragma unroll
        for (int u = 0; u < NST; ++u) {
        const int c = cb + u;
        if (c >= c1) break;
        commit(a, gq[u], ctx, tid, st[u]);
        ctx.sync();
        const Geo g = gq[u];
        gq[u] = geometry(a, c + NST < c1 ? c + NST : c1 - 1);
        issue(a, gq[u], xr, tid, st[u], c + NST < c1);
        // ---- levels
#pragma unroll
        for (int j = 0; j < WL_DWT1D_MAXJ; ++j) {               // level j + 1 from buffer j
            if (j >= a.J) break;
            const float* const src = reinterpret_cast<const float*>(ctx.smem + a.buf_off[j]);
            float* const dst = j + 1 < a.J ? reinterpret_cast<float*>(ctx.smem + a.buf_off[j + 1]) : nullptr;
            const int nj = a.n[j], base = a.base[j];
            const int klo = g.clo[j + 1], khi = g.chi[j + 1];
            const int olo = g.olo[j + 1], ohi = g.ohi[j + 1];
            T* const hp = a.hi[j] + (size_t)row * a.n[j + 1];
            T* const lp = a.lo + (size_t)row * a.n[j + 1];
            const int sorg = g.org[j], dorg = g.org[j + 1];
            // buffer 0 holds extended positions (the extension samples were materialised by the loader); the buffers of
            // the inner levels hold positions inside the signal only: beyond its ends the extension rule picks the source
            const int in_lo = j == 0 ? g.clo[0] : 0, in_hi = j == 0 ? g.chi[0] : nj;
            const bool even = ((base - sorg) & 1) == 0;         // (uniform) the pairs (2k + base + 2u, + 1) are 8-byte aligned
#if WL_DWT1D_UNROLL > 1
#pragma unroll WL_DWT1D_UNROLL
#endif
            for (int k = klo + tid; k < khi; k += kThreads) {
                const int w0 = 2 * k + base;
                wl_v2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
                if (w0 >= in_lo && w0 + LT <= in_hi) {
                    const float* p = src + (w0 - sorg);
                    if (even) {
#pragma unroll
                        for (int u = 0; u < LT / 2; ++u) {
                            const wl_f2 t2 = *reinterpret_cast<const wl_f2*>(p + 2 * u);
                            const wl_v2 s = {t2.x, t2.y};
                            wl_pk_fma_x(acc0, tp[2 * u], s);
                            wl_pk_fma_y(acc1, tp[2 * u + 1], s);
                        }
                    } else {
                        wl_v2 s = {p[0], 0.f};
                        wl_pk_fma_x(acc0, tp[0], s);
#pragma unroll
                        for (int u = 0; u < LT / 2 - 1; ++u) {
                            const wl_f2 t2 = *reinterpret_cast<const wl_f2*>(p + 1 + 2 * u);
                            s = wl_v2{t2.x, t2.y};
                            wl_pk_fma_x(acc1, tp[2 * u + 1], s);
                            wl_pk_fma_y(acc0, tp[2 * u + 2], s);
                        }
                        s = wl_v2{p[LT - 1], 0.f};
                        wl_pk_fma_x(acc1, tp[LT - 1], s);
                    }
                } else {
                    for (int t = 0; t < LT; ++t) {
                        int sp = w0 + t;
                        if (sp < in_lo || sp >= in_hi) sp = wl_ext(sp, nj, a.ext);   // (inner levels; buffer 0 never gets here)
                        const float v = sp < 0 && j > 0 ? 0.f : src[sp - sorg];
                        acc0.x = __builtin_fmaf(a.h0[t], v, acc0.x);
                        acc0.y = __builtin_fmaf(a.h1[t], v, acc0.y);
                    }
                }
                const float lo = acc0.x + acc1.x, hi = acc0.y + acc1.y;
                if (dst) dst[k - dorg] = lo;
                if (k >= olo && k < ohi) {
                    hp[k] = (T)hi;
                    if (!dst) lp[k] = (T)lo;
                }
            }
            ctx.sync();
        }
        }
        }
    }
};
