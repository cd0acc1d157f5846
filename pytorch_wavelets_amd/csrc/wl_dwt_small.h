// Multi-level 2-D DWT analysis of SMALL planes (up to 64 x 64: feature maps of a CNN, CIFAR / Tiny-ImageNet images - the
// shapes wavelet pooling and the scattering front ends are used on), several planes per workgroup: DWTForward.forward's
// level loop (reference dwt/transform2d.py:63-74 = J x AFB2D.forward, dwt/lowlevel.py:336-347 -> afb1d :91-172).
// The streaming kernel (wl_dwt_rows.h) gives a plane a workgroup of 12 waves: on 32 x 32 planes most of its lanes idle and
// it reaches 0.08-0.10 of the HBM roofline (round 4, 2048 x 3 x 32 x 32).  Here a workgroup of four waves owns G consecutive
// planes - one contiguous chunk of memory, loaded with 16-byte loads - and keeps them in LDS through all levels:
//   per level:  row pass    (lo, hi)[p][r][k] = sum_t (hw_lo, hw_hi)[t] * X_p(r, 2k + base_w + t)          -> LDS
//               column pass  (ll, W-hi/H-lo) = sum_t hh_lo[t] * (lo, hi)[p][2i + base_h + t][k],
//                            (W-lo/H-hi, hh) = sum_t hh_hi[t] * (lo, hi)[p][2i + base_h + t][k]
//               the three band planes to memory (consecutive threads on consecutive addresses of a band plane), ll to the
//               LDS buffer the next level reads - or to memory after the last level.
// Positions beyond a plane go through the extension rule of the mode (every mode of the reference incl. periodization);
// outputs whose L samples lie inside the plane skip it.  HBM traffic = every input element once + every output once.
#pragma once
#include "wl_common.h"

#define WL_SMALL_MAXLEV 4

template <typename T>
struct WlSmallArgs {
    const T* x;                        // (planes, H, W) dense
    T* yl;                             // (planes, h[nlev], w[nlev]) dense
    T* yh[WL_SMALL_MAXLEV];            // (planes, 3, h[j+1], w[j+1]) dense
    const float* hw_lo; const float* hw_hi; const float* hh_lo; const float* hh_hi;   // stored (reversed) taps, L each
    int64_t planes, nblocks;
    int nlev, L, ext, G;
    int h[WL_SMALL_MAXLEV + 1], w[WL_SMALL_MAXLEV + 1];   // h[0] x w[0] = the plane, h[j] x w[j] = level j's sub-bands
    int base_h[WL_SMALL_MAXLEV], base_w[WL_SMALL_MAXLEV];
    unsigned mg_k[WL_SMALL_MAXLEV];    // ceil(2^32 / d) (0 for d = 1): d = w[j+1]
    // The outputs of an axis whose L samples lie inside the plane (k in [kf, kf + ki)) and the others (nb = K - ki of them, at
    // either end) are separate loops: a wave runs either the unrolled interior form or the extension form, never both.
    int kf_w[WL_SMALL_MAXLEV], ki_w[WL_SMALL_MAXLEV], kf_h[WL_SMALL_MAXLEV], ki_h[WL_SMALL_MAXLEV];
    unsigned mg_r1[WL_SMALL_MAXLEV], mg_k1[WL_SMALL_MAXLEV], mg_r2[WL_SMALL_MAXLEV], mg_k2[WL_SMALL_MAXLEV];   // d = h ki_w, ki_w, h nb_w, nb_w
    unsigned mg_c1[WL_SMALL_MAXLEV], mg_c2[WL_SMALL_MAXLEV];   // d = ki_h w[j+1], nb_h w[j+1]
    int buf_off[2], mid_off, tap_off, lds_bytes;
    int vec_ok;                        // the chunk of a workgroup starts on a 16-byte boundary and H W is a multiple of 4 elements
};

// LT = compile-time tap count (2, 4, 6, 8: the taps in registers, unrolled loops), 0 = any L at run time (taps in LDS)
template <typename T, int LT>
struct WlAfbSmall {
    typedef WlSmallArgs<T> Args;
    static const int kThreads = 256;
    static const int kMinWaves = 2;
    // Source position of extended position i under ONE fold / wrap (no division): what every level of a plane needs as long as
    // the filter is not longer than the signal (the launcher checks, and declines otherwise: the per-level kernels fold any
    // number of times).  A wave
    // whose lanes straddle a plane's edge runs both the interior and this path: it has to be cheap (the first version called
    // wl_ext - two integer divisions - per tap: 0.10 of the roofline on 32 x 32 planes).
    static WL_DEV int ext_once(int i, int n, int ext) {
        if ((unsigned)i < (unsigned)n) return i;
        if (ext == WL_EXT_ZERO) return -1;
        if (ext == WL_EXT_SYM) return i < 0 ? -1 - i : 2 * n - 1 - i;
        if (ext == WL_EXT_REFL) return i < 0 ? -i : 2 * n - 2 - i;
        if (ext == WL_EXT_PERIODIC) return i < 0 ? i + n : i - n;
        const int ne = n + (n & 1);                                 // periodization: the last sample repeated when n is odd, then wrapped
        const int j = i < 0 ? i + ne : (i >= ne ? i - ne : i);
        return j == n ? n - 1 : j;
    }
    // n / d by multiplication with magic = ceil(2^32 / d) (exact for n, d < 2^16); d = 1 has no 32-bit magic: stored as 0
    static WL_HD unsigned divm(unsigned n, unsigned magic) { return magic ? (unsigned)(((unsigned long long)n * magic) >> 32) : n; }

    // (a run-time loop over the levels: unrolled four times with the extension paths inlined the kernel spilled its scalar
    // registers 8000 times over)
    static WL_DEV void levels(const Args& a, const WlCtx& ctx, int np, int64_t plane0, const float* tw0, const float* tw1,
                              const float* th0, const float* th1) {
        for (int j = 0; j < a.nlev; ++j) {
            const int tid = ctx.tid;
            const int L = LT ? LT : a.L;
            const int h = a.h[j], w = a.w[j], Kh = a.h[j + 1], Kw = a.w[j + 1];
            const float* const src = reinterpret_cast<const float*>(ctx.smem + a.buf_off[j & 1]);
            float* const nxt = reinterpret_cast<float*>(ctx.smem + a.buf_off[(j + 1) & 1]);
            wl_f2* const mid = reinterpret_cast<wl_f2*>(ctx.smem + a.mid_off);
            const int hK = h * Kw;
            // ---- row pass, interior outputs: every sample inside the row
            {
                const int ki = a.ki_w[j], kf = a.kf_w[j], hki = h * ki;
                for (int idx = tid; idx < np * hki; idx += kThreads) {
                    const int p = (int)divm(idx, a.mg_r1[j]);
                    const int rem = idx - p * hki;
                    const int r = (int)divm(rem, a.mg_k1[j]);
                    const int k = kf + (rem - r * ki);
                    const float* row = src + (p * h + r) * w + 2 * k + a.base_w[j];
                    float lo = 0.f, hi = 0.f;
                    if (LT) {
#pragma unroll
                        for (int t = 0; t < (LT ? LT : 1); ++t) { const float v = row[t]; lo = __builtin_fmaf(tw0[t], v, lo); hi = __builtin_fmaf(tw1[t], v, hi); }
                    } else {
                        for (int t = 0; t < L; ++t) { const float v = row[t]; lo = __builtin_fmaf(tw0[t], v, lo); hi = __builtin_fmaf(tw1[t], v, hi); }
                    }
                    wl_f2 q; q.x = lo; q.y = hi;
                    mid[p * hK + r * Kw + k] = q;                   // [p][r][k]
                }
                // ---- row pass, the outputs at either end of a row: samples through the extension rule
                const int nb = Kw - ki, hnb = h * nb;
                for (int idx = tid; idx < np * hnb; idx += kThreads) {
                    const int p = (int)divm(idx, a.mg_r2[j]);
                    const int rem = idx - p * hnb;
                    const int r = (int)divm(rem, a.mg_k2[j]);
                    const int kb = rem - r * nb;
                    const int k = kb < kf ? kb : kb + ki;
                    const float* row = src + (p * h + r) * w;
                    const int s = 2 * k + a.base_w[j];
                    float lo = 0.f, hi = 0.f;
                    for (int t = 0; t < L; ++t) {
                        const int c = ext_once(s + t, w, a.ext);
                        const float v = c < 0 ? 0.f : row[c];
                        lo = __builtin_fmaf(tw0[t], v, lo); hi = __builtin_fmaf(tw1[t], v, hi);
                    }
                    wl_f2 q; q.x = lo; q.y = hi;
                    mid[p * hK + r * Kw + k] = q;
                }
            }
            ctx.sync();
            // ---- column pass, stores: interior rows, then the rows at either end
            const bool last = j + 1 == a.nlev;
            const int KK = Kh * Kw;
            T* const hp = a.yh[j] + (size_t)plane0 * 3 * KK;
            T* const lp = a.yl + (size_t)plane0 * KK;
            {
                const int ki = a.ki_h[j], kf = a.kf_h[j], kiK = ki * Kw;
                for (int idx = tid; idx < np * kiK; idx += kThreads) {
                    const int p = (int)divm(idx, a.mg_c1[j]);
                    const int rem = idx - p * kiK;
                    const int ii = (int)divm(rem, a.mg_k[j]);
                    const int k = rem - ii * Kw, i = kf + ii;
                    const wl_f2* col = mid + p * hK + (2 * i + a.base_h[j]) * Kw + k;
                    float ll = 0.f, hl = 0.f, lh = 0.f, hh = 0.f;     // hl: W-hi / H-lo, lh: W-lo / H-hi
                    if (LT) {
#pragma unroll
                        for (int t = 0; t < (LT ? LT : 1); ++t) {
                            const wl_f2 v = col[t * Kw];
                            ll = __builtin_fmaf(th0[t], v.x, ll); hl = __builtin_fmaf(th0[t], v.y, hl);
                            lh = __builtin_fmaf(th1[t], v.x, lh); hh = __builtin_fmaf(th1[t], v.y, hh);
                        }
                    } else {
                        for (int t = 0; t < L; ++t) {
                            const wl_f2 v = col[t * Kw];
                            ll = __builtin_fmaf(th0[t], v.x, ll); hl = __builtin_fmaf(th0[t], v.y, hl);
                            lh = __builtin_fmaf(th1[t], v.x, lh); hh = __builtin_fmaf(th1[t], v.y, hh);
                        }
                    }
                    const int o1 = i * Kw + k;
                    T* o = hp + (size_t)p * 3 * KK + o1;
                    o[0] = (T)lh; o[KK] = (T)hl; o[2 * KK] = (T)hh;
                    if (last) lp[p * KK + o1] = (T)ll; else nxt[p * KK + o1] = ll;     // [p][i][k]: the next level's planes, dense
                }
                const int nb = Kh - ki, nbK = nb * Kw;
                for (int idx = tid; idx < np * nbK; idx += kThreads) {
                    const int p = (int)divm(idx, a.mg_c2[j]);
                    const int rem = idx - p * nbK;
                    const int ib = (int)divm(rem, a.mg_k[j]);
                    const int k = rem - ib * Kw, i = ib < kf ? ib : ib + ki;
                    const wl_f2* col = mid + p * hK + k;
                    const int s = 2 * i + a.base_h[j];
                    float ll = 0.f, hl = 0.f, lh = 0.f, hh = 0.f;
                    for (int t = 0; t < L; ++t) {
                        const int r = ext_once(s + t, h, a.ext);
                        wl_f2 v; v.x = v.y = 0.f;
                        if (r >= 0) v = col[r * Kw];
                        ll = __builtin_fmaf(th0[t], v.x, ll); hl = __builtin_fmaf(th0[t], v.y, hl);
                        lh = __builtin_fmaf(th1[t], v.x, lh); hh = __builtin_fmaf(th1[t], v.y, hh);
                    }
                    const int o1 = i * Kw + k;
                    T* o = hp + (size_t)p * 3 * KK + o1;
                    o[0] = (T)lh; o[KK] = (T)hl; o[2 * KK] = (T)hh;
                    if (last) lp[p * KK + o1] = (T)ll; else nxt[p * KK + o1] = ll;
                }
            }
            ctx.sync();
        }
    }

    // A workgroup walks over groups bid, bid + nblocks, ..: while the levels of one group run out of LDS, the planes of the next
    // are already on their way into registers (PRE 16-byte loads per thread: groups of up to 6144 elements) - a workgroup per
    // group, load then compute, spent most of its time waiting for its one load (0.20 of the roofline at 2048 x 3 x 32 x 32).
    static const int PRE = 6;
    typedef T Quad4 __attribute__((ext_vector_type(4)));
    struct Pre { Quad4 q[PRE]; };
    static WL_DEV void fetch(const Args& a, int tid, int64_t grp, Pre& pr) {
        const int64_t plane0 = grp * a.G;
        const int np = plane0 + a.G <= a.planes ? a.G : (int)(a.planes - plane0);
        const int HW = a.h[0] * a.w[0], n = np * HW;
        const T* const xp = a.x + (size_t)plane0 * HW;
#pragma unroll
        for (int u = 0; u < PRE; ++u) {
            const int i = 4 * (tid + kThreads * u);
            if (a.vec_ok) {
                pr.q[u] = *reinterpret_cast<const Quad4*>(xp + (i + 3 < n ? i : 0));       // (off lanes: an element group that exists)
            } else {
                Quad4 v;
                v.x = xp[i < n ? i : 0]; v.y = xp[i + 1 < n ? i + 1 : 0]; v.z = xp[i + 2 < n ? i + 2 : 0]; v.w = xp[i + 3 < n ? i + 3 : 0];
                pr.q[u] = v;
            }
        }
    }
    static WL_DEV void commit(const Args& a, const WlCtx& ctx, int tid, int n, const Pre& pr) {
        float* const b0 = reinterpret_cast<float*>(ctx.smem + a.buf_off[0]);
#pragma unroll
        for (int u = 0; u < PRE; ++u) {
            const int i = 4 * (tid + kThreads * u);
            if (i + 3 < n) {
                wl_vf4 f; f.x = (float)pr.q[u].x; f.y = (float)pr.q[u].y; f.z = (float)pr.q[u].z; f.w = (float)pr.q[u].w;
                *reinterpret_cast<wl_vf4*>(b0 + i) = f;
            } else {
                if (i < n) b0[i] = (float)pr.q[u].x;
                if (i + 1 < n) b0[i + 1] = (float)pr.q[u].y;
                if (i + 2 < n) b0[i + 2] = (float)pr.q[u].z;
            }
        }
    }

    static WL_DEV void run(const Args& a, const WlCtx& ctx) {
        const int tid = ctx.tid;
        const int HW = a.h[0] * a.w[0];
        const int64_t ngroups = (a.planes + a.G - 1) / a.G;
        float tw0[LT ? LT : 1], tw1[LT ? LT : 1], th0[LT ? LT : 1], th1[LT ? LT : 1];
        float* const tp = reinterpret_cast<float*>(ctx.smem + a.tap_off);
        if (LT) {
#pragma unroll
            for (int t = 0; t < (LT ? LT : 1); ++t) { tw0[t] = a.hw_lo[t]; tw1[t] = a.hw_hi[t]; th0[t] = a.hh_lo[t]; th1[t] = a.hh_hi[t]; }
        } else if (tid < a.L) {
            tp[tid] = a.hw_lo[tid]; tp[a.L + tid] = a.hw_hi[tid]; tp[2 * a.L + tid] = a.hh_lo[tid]; tp[3 * a.L + tid] = a.hh_hi[tid];
        }
        Pre pr;
        fetch(a, tid, ctx.bid, pr);
        for (int64_t grp = ctx.bid; grp < ngroups; grp += a.nblocks) {
            const int64_t plane0 = grp * a.G;
            const int np = plane0 + a.G <= a.planes ? a.G : (int)(a.planes - plane0);
            commit(a, ctx, tid, np * HW, pr);
            ctx.sync();
            if (grp + a.nblocks < ngroups) fetch(a, tid, grp + a.nblocks, pr);
            if (LT) levels(a, ctx, np, plane0, tw0, tw1, th0, th1);
            else levels(a, ctx, np, plane0, tp, tp + a.L, tp + 2 * a.L, tp + 3 * a.L);
        }
    }
};

// ---- synthesis ---------------------------------------------------------------------------------------------------------
// Multi-level 2-D DWT synthesis of small planes, several planes per workgroup: DWTInverse.forward's level loop (reference
// dwt/transform2d.py:131-148 = J x SFB2D.forward, dwt/lowlevel.py:671-680 -> sfb1d :226-271, incl. the 'unpad' crop of :141-146)
// and the backward of the analysis (AFB2D.backward :350-365, with the analysis taps).  1-D, per axis, as wl_synth1d evaluates it:
//     full[m] = sum_k g0[m - 2k] lo[k] + g1[m - 2k] hi[k]            (0 <= k < K, 0 <= m - 2k < L)
//     y[p] = full[p + L - 2], p < 2K - L + 2;    periodization: m = (p + L/2 - 1) mod 2K, y[p] = full[m] (+ full[m + 2K] for m < L - 2)
// along H with the column pair on (ll, W-lo/H-hi) and (W-hi/H-lo, hh), then along W with the row pair on the two results.
// All coefficients of a group of planes are loaded into LDS at the start (one exposed memory latency), the lowpass of a level
// becomes the next finer level's ll in LDS (cropped to the size of that level's high-pass planes where it is one larger).
template <typename T>
struct WlSmallSynArgs {
    const T* yl;                       // (planes, llh, llw) dense: the coarsest lowpass (may be one row / column larger than yh[nlev-1])
    const T* yh[WL_SMALL_MAXLEV];      // (planes, 3, Kh[j], Kw[j]) dense, finest first; nullptr = zeros
    T* y;                              // (planes, OH[0], OW[0]) dense
    const float* gw_lo; const float* gw_hi; const float* gh_lo; const float* gh_hi;
    int64_t planes, nblocks;
    int nlev, L, per, G;
    int llh, llw;
    int Kh[WL_SMALL_MAXLEV], Kw[WL_SMALL_MAXLEV], OH[WL_SMALL_MAXLEV], OW[WL_SMALL_MAXLEV];
    unsigned mg_pk[WL_SMALL_MAXLEV], mg_k[WL_SMALL_MAXLEV], mg_pq[WL_SMALL_MAXLEV], mg_q[WL_SMALL_MAXLEV];   // d = OH Kw, Kw, OH OW, OW
    unsigned mg2_pk[WL_SMALL_MAXLEV], mg2_pq[WL_SMALL_MAXLEV], mg2_q[WL_SMALL_MAXLEV];   // pair form: d = ceil(OH/2) Kw, OH ceil(OW/2), ceil(OW/2)
    int h_off[WL_SMALL_MAXLEV];        // LDS byte offset of level j's high-pass planes ([p][3][Kh][Kw])
    int ll_off[2], mid_off, tap_off, lds_bytes;
};

// LT = compile-time tap count (2, 4, 6, 8; every mode but periodization): the PAIR form - the output pair (2q, 2q + 1) of an axis
// reads coefficients q .. q + L/2 - 1, (y[2q], y[2q+1]) += (g[L-2-2i], g[L-1-2i]) c[q + i], taps in registers, no search for the
// tap range per output (the general form below spent more on its loop bounds than on its two or four taps: the synthesis ran
// at half the fraction of the analysis on feature-map shapes).  LT = 0: any even L, periodization included.
template <typename T, int LT = 0>
struct WlSfbSmall {
    typedef WlSmallSynArgs<T> Args;
    static const int HL = LT / 2;
    static const int kThreads = 256;
    static const int kMinWaves = 2;
    static WL_HD unsigned divm(unsigned n, unsigned magic) { return magic ? (unsigned)(((unsigned long long)n * magic) >> 32) : n; }
    // full[m] over K coefficients spaced `cs` floats apart (lo, hi: two planes; hi may be null), taps g0 / g1 in LDS
    static WL_DEV float full(const float* lo, const float* hi, int cs, int K, int L, const float* g0, const float* g1, int m) {
        int k1 = m >> 1;
        if (k1 > K - 1) k1 = K - 1;
        int k0 = m - L + 1 > 0 ? (m - L + 2) >> 1 : 0;
        float acc = 0.f;
        for (int k = k0; k <= k1; ++k) {
            const int t = m - 2 * k;
            acc = __builtin_fmaf(g0[t], lo[k * cs], acc);
            if (hi) acc = __builtin_fmaf(g1[t], hi[k * cs], acc);
        }
        return acc;
    }
    static WL_DEV float synth(const float* lo, const float* hi, int cs, int K, int L, const float* g0, const float* g1, int p, int per) {
        if (!per) return full(lo, hi, cs, K, L, g0, g1, p + L - 2);
        const int N = 2 * K;
        int m = p + (L / 2 - 1 < 2 * N ? L / 2 - 1 : 0);
        m -= (m / N) * N;
        float v = full(lo, hi, cs, K, L, g0, g1, m);
        if (m < L - 2) v += full(lo, hi, cs, K, L, g0, g1, m + N);
        return v;
    }
    template <typename S>
    static WL_DEV void load_block(float* dst, const S* src, int n, int tid) {
        for (int i = tid; i < n; i += kThreads) dst[i] = (float)src[i];
    }
    static WL_DEV void run(const Args& a, const WlCtx& ctx) {
        const int tid = ctx.tid;
        const int64_t plane0 = ctx.bid * a.G;
        const int np = plane0 + a.G <= a.planes ? a.G : (int)(a.planes - plane0);
        const int L = a.L;
        float* const tp = reinterpret_cast<float*>(ctx.smem + a.tap_off);
        if (tid < L) { tp[tid] = a.gw_lo[tid]; tp[L + tid] = a.gw_hi[tid]; tp[2 * L + tid] = a.gh_lo[tid]; tp[3 * L + tid] = a.gh_hi[tid]; }
        // every coefficient of the group: the coarsest lowpass and the high-pass planes of all levels (contiguous chunks)
        const int J = a.nlev;
        float* llbuf = reinterpret_cast<float*>(ctx.smem + a.ll_off[0]);
        load_block(llbuf, a.yl + (size_t)plane0 * a.llh * a.llw, np * a.llh * a.llw, tid);
        for (int j = 0; j < J; ++j) {
            const int n3 = 3 * a.Kh[j] * a.Kw[j];
            float* hb = reinterpret_cast<float*>(ctx.smem + a.h_off[j]);
            if (a.yh[j]) load_block(hb, a.yh[j] + (size_t)plane0 * n3, np * n3, tid);
            else for (int i = tid; i < np * n3; i += kThreads) hb[i] = 0.f;
        }
        ctx.sync();
        int llh = a.llh, llw = a.llw;                            // size (pitch) of the current ll buffer
        for (int j = J - 1; j >= 0; --j) {
            const int Kh = a.Kh[j], Kw = a.Kw[j], OH = a.OH[j], OW = a.OW[j], KK = Kh * Kw;
            const float* const ll = reinterpret_cast<const float*>(ctx.smem + a.ll_off[(J - 1 - j) & 1]);
            float* const nxt = reinterpret_cast<float*>(ctx.smem + a.ll_off[(J - j) & 1]);
            const float* const hb = reinterpret_cast<const float*>(ctx.smem + a.h_off[j]);
            wl_f2* const mid = reinterpret_cast<wl_f2*>(ctx.smem + a.mid_off);
            const int OK = OH * Kw, OO = OH * OW;
            T* const yp = a.y + (size_t)plane0 * OO;
            if (LT) {
                float ch0[LT ? LT : 1], ch1[LT ? LT : 1], cw0[LT ? LT : 1], cw1[LT ? LT : 1];
#pragma unroll
                for (int t = 0; t < (LT ? LT : 1); ++t) { ch0[t] = tp[2 * L + t]; ch1[t] = tp[3 * L + t]; cw0[t] = tp[t]; cw1[t] = tp[L + t]; }
                // ---- along H, two output rows per item
                const int OH2 = (OH + 1) >> 1, O2K = OH2 * Kw;
                for (int idx = tid; idx < np * O2K; idx += kThreads) {
                    const int p = (int)divm(idx, a.mg2_pk[j]);
                    const int rem = idx - p * O2K;
                    const int rq = (int)divm(rem, a.mg_k[j]);
                    const int k = rem - rq * Kw;
                    const float* lp = ll + p * llh * llw + k;
                    const float* hp = hb + p * 3 * KK + k;
                    float lo0 = 0.f, lo1 = 0.f, hi0 = 0.f, hi1 = 0.f;
#pragma unroll
                    for (int i = 0; i < (LT ? HL : 1); ++i) {
                        const int kk = rq + i;
                        if (kk < Kh) {
                            const float vll = lp[kk * llw], vlh = hp[kk * Kw], vhl = hp[KK + kk * Kw], vhh = hp[2 * KK + kk * Kw];
                            const float a0 = ch0[LT - 2 - 2 * i], a1 = ch0[LT - 1 - 2 * i], b0 = ch1[LT - 2 - 2 * i], b1 = ch1[LT - 1 - 2 * i];
                            lo0 += a0 * vll + b0 * vlh; lo1 += a1 * vll + b1 * vlh;
                            hi0 += a0 * vhl + b0 * vhh; hi1 += a1 * vhl + b1 * vhh;
                        }
                    }
                    wl_f2 u0; u0.x = lo0; u0.y = hi0;
                    mid[p * OK + (2 * rq) * Kw + k] = u0;
                    if (2 * rq + 1 < OH) { wl_f2 u1; u1.x = lo1; u1.y = hi1; mid[p * OK + (2 * rq + 1) * Kw + k] = u1; }
                }
                ctx.sync();
                // ---- along W, two output columns per item
                const int OW2 = (OW + 1) >> 1, OHW2 = OH * OW2;
                for (int idx = tid; idx < np * OHW2; idx += kThreads) {
                    const int p = (int)divm(idx, a.mg2_pq[j]);
                    const int rem = idx - p * OHW2;
                    const int r = (int)divm(rem, a.mg2_q[j]);
                    const int cq = rem - r * OW2;
                    const wl_f2* m2 = mid + p * OK + r * Kw + cq;
                    float y0 = 0.f, y1 = 0.f;
#pragma unroll
                    for (int i = 0; i < (LT ? HL : 1); ++i) {
                        if (cq + i < Kw) {
                            const wl_f2 v = m2[i];
                            y0 += cw0[LT - 2 - 2 * i] * v.x + cw1[LT - 2 - 2 * i] * v.y;
                            y1 += cw0[LT - 1 - 2 * i] * v.x + cw1[LT - 1 - 2 * i] * v.y;
                        }
                    }
                    const int o = p * OO + r * OW + 2 * cq;
                    if (j == 0) { yp[o] = (T)y0; if (2 * cq + 1 < OW) yp[o + 1] = (T)y1; }
                    else { nxt[o] = y0; if (2 * cq + 1 < OW) nxt[o + 1] = y1; }
                }
                ctx.sync();
                llh = OH; llw = OW;
                continue;
            }
            // ---- along H: (ll, lh) -> lo, (hl, hh) -> hi at every column k < Kw (the crop of a larger ll is the index range)
            const int ncol = np * OH * Kw;
            for (int idx = tid; idx < ncol; idx += kThreads) {
                const int p = (int)divm(idx, a.mg_pk[j]);
                const int rem = idx - p * OK;
                const int r = (int)divm(rem, a.mg_k[j]);
                const int k = rem - r * Kw;
                const float* hp = hb + p * 3 * KK + k;
                wl_f2 q;
                // (ll rows are llw floats apart, band rows Kw: the two sources of the first synthesis as two single-source sums)
                q.x = synth(ll + p * llh * llw + k, nullptr, llw, Kh, L, tp + 2 * L, tp + 3 * L, r, a.per)
                    + synth(hp, nullptr, Kw, Kh, L, tp + 3 * L, tp + 3 * L, r, a.per);
                q.y = synth(hp + KK, hp + 2 * KK, Kw, Kh, L, tp + 2 * L, tp + 3 * L, r, a.per);
                mid[idx] = q;                                       // [p][r][k]
            }
            ctx.sync();
            // ---- along W
            const int nout = np * OH * OW;
            for (int idx = tid; idx < nout; idx += kThreads) {
                const int p = (int)divm(idx, a.mg_pq[j]);
                const int rem = idx - p * OO;
                const int r = (int)divm(rem, a.mg_q[j]);
                const int q = rem - r * OW;
                const float* m2 = reinterpret_cast<const float*>(mid + p * OK + r * Kw);
                const float v = synth(m2, m2 + 1, 2, Kw, L, tp, tp + L, q, a.per);
                if (j == 0) yp[idx] = (T)v; else nxt[idx] = v;      // [p][r][q]: the next level's ll, dense at OH x OW
            }
            ctx.sync();
            llh = OH; llw = OW;
        }
    }
};
