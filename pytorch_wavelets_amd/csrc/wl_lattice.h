// The two-channel orthogonal filter bank as a LATTICE of plane rotations (Vaidyanathan's factorisation of a paraunitary
// polyphase matrix) - the arithmetic of the column pass of the long-filter strip kernels (wl_dwt_strip.h), which are bound by
// the vector instructions they issue, not by HBM.
//
// One analysis level along an axis, as the kernels evaluate it (reference dwt/lowlevel.py:91-172, stored = reversed taps):
//     y_lo[o] = sum_t lo[t] P[e0 + t],  y_hi[o] = sum_t hi[t] P[e0 + t],  e0 = base + 2 o,  L = 2 K taps
// is, in terms of the row pairs ("feeds") a[m] = P[e + 2m], b[m] = P[e + 2m + 1], the causal 2 x 2 polynomial matrix
//     (y_lo, y_hi)[m] = sum_{i < K} E_i (a, b)[m - i],   E_i = [[lo[L-2-2i], lo[L-1-2i]], [hi[L-2-2i], hi[L-1-2i]]]
// For the decomposition pair of an orthogonal wavelet E(z) is paraunitary and factors into K rotations and K - 1 delays,
//     E(z) = R_{K-1} D(z) R_{K-2} .. D(z) R_0,   D(z) = diag(1, z^-1),   R_k = [[c_k, s_k], [-s_k, c_k]],  R_0 a reflection,
// which, with every cos factored out into ONE gain g, is the recurrence the kernel runs per feed:
//     (u, v) = (a + T_0 b, b - T_0 a);   for k = 1 .. K-1:  (u, v) <- (u + T_k v', v' - T_k u),  v' = v of stage k-1 one feed ago;
//     y_lo = g u,  y_hi = -g v                                              (g is folded into the row filter's taps)
// 2 K fused multiply-adds per output pair instead of the 2 L of the direct form, K - 1 delayed values of state instead of an
// L-row window.  In float32 the recurrence is as accurate as the direct sum (1.5 - 3 e-7 of the largest coefficient against
// 0.7 e-7: every stage is a scaled rotation, nothing cancels; measured for db2 .. db10, sym4 .. sym8, coif1 .. coif3).
//
// The factorisation is computed ON THE DEVICE from the taps as they are at call time (wl_lattice_factor, float64, by the
// one-thread kernel WlTapPrep in front of a lattice launch) and ACCEPTED only if the bank it realises is the bank in the
// buffers to within `tol` of the largest tap: the entries every peeling step must annihilate are the measure (they bound
// the difference between the taps and what the lattice computes; 1 - 6 e-8 for the float32 tables).  tol = 2^-22 for
// float32 data; 2^-12 - a quarter of a unit in the last place of the storage type - for float16 data, whose modules
// usually hold float16-ROUNDED taps (`.half()`): those are no exact orthogonal pair any more (residue 0.2 - 1.2 e-4 for
// db6 / db8 / sym8 / coif3: accepted; db10: 9 e-4, rejected), the lattice then computes the orthogonal bank nearest to them,
// which is as close to the float32 reference (2.0 e-4) as the rounded taps themselves (2.3 e-4).  Anything else - custom
// banks, edited buffers, NaNs, a rotation of 90 degrees (cos = 0) - is rejected, and the launch behind the lattice kernel,
// its armed two-bank fallback, does the work (wl_common.h, tap-relation guards).
#pragma once
#include "wl_common.h"

#define WL_LAT_MAXK 10                 // L <= 20 taps
#define WL_TAP_SCRATCH_FLOATS 16       // device scratch of a lattice launch: [0] verdict, [1] g, [2 .. 2 + K) T_k
#define WL_LAT_OK 0x4c415431u          // verdict word of an accepted factorisation

// The SYNTHESIS bank of the same wavelet, as the synthesis kernels evaluate it (reference dwt/lowlevel.py:226-271): the output row
// pair of feed m is  (z[2m], z[2m+1]) = sum_{t < K} (g0[2t], g0[2t+1]) a[m-t] + (g1[2t], g1[2t+1]) b[m-t]  (a / b = the lowpass /
// highpass coefficient rows), the polynomial matrix R(z) whose TRANSPOSE F(z) = sum_t [[g0[2t], g0[2t+1]], [g1[2t], g1[2t+1]]] z^-t
// factors exactly like E(z) above (`syn` = true builds F's coefficients).  Transposing the product reverses it:
//     (p, q) = (a, -b);   for k = K-1 .. 1:  (p, q) <- (p - T_k q, q + T_k p),  q <- q one feed ago;
//     (z[2m], z[2m+1]) = g (p - T_0 q, q + T_0 p)                                  (g again folded into the row synthesis' taps)
//
// lo / hi: the L stored taps of one axis.  true: *g and T[0 .. L/2) hold the recurrence.
// (L is a template parameter: every loop unrolls and the matrices live in registers - with run-time bounds they sat in scratch
// memory and the one-thread kernel took 25 us, as long as a small analysis launch)
template <int L>
WL_HD bool wl_lattice_factor(const float* lo, const float* hi, double tol, float* g, float* T, bool syn = false) {
    static_assert(L >= 4 && L % 2 == 0 && L / 2 <= WL_LAT_MAXK, "even tap counts from 4 to 20");
    const int K = L / 2;
    double E[K][2][2], n0[K][2], n1[K][2], Td[K];
    double scale = 0.0;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const int e = syn ? 2 * i : L - 2 - 2 * i;
        E[i][0][0] = lo[e]; E[i][0][1] = lo[e + 1];
        E[i][1][0] = hi[e]; E[i][1][1] = hi[e + 1];
    }
#pragma unroll
    for (int t = 0; t < L; ++t) {
        const double a = lo[t] < 0 ? -(double)lo[t] : (double)lo[t], b = hi[t] < 0 ? -(double)hi[t] : (double)hi[t];
        if (!(a <= 1e30) || !(b <= 1e30)) return false;            // NaN / Inf taps
        scale = a > scale ? a : scale; scale = b > scale ? b : scale;
    }
    if (!(scale > 0.0)) return false;
    auto ab = [](double v) { return v < 0 ? -v : v; };
    double res = 0.0, gain = 1.0;
#pragma unroll
    for (int k = K - 1; k >= 1; --k) {
        // E(z) = R_k D(z) E'(z): the rotation that annihilates the z^-k coefficient of row 0 (and with it, for a paraunitary
        // E, the z^0 coefficient of row 1)
        const int j = ab(E[k][0][0]) + ab(E[k][1][0]) >= ab(E[k][0][1]) + ab(E[k][1][1]) ? 0 : 1;
        const double x = E[k][0][j], y = E[k][1][j];
        const double r = __builtin_sqrt(x * x + y * y);
        if (!(r > 1e-12 * scale)) return false;
        const double s = x / r, c = y / r;
        if (!(ab(c) > 1e-7)) return false;                         // a rotation by 90 degrees has no finite tangent
#pragma unroll
        for (int i = 0; i < K; ++i)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (i > k) continue;
                n0[i][q] = c * E[i][0][q] - s * E[i][1][q];
                n1[i][q] = s * E[i][0][q] + c * E[i][1][q];
            }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            res = ab(n0[k][q]) > res ? ab(n0[k][q]) : res;
            res = ab(n1[0][q]) > res ? ab(n1[0][q]) : res;
        }
#pragma unroll
        for (int i = 0; i < K - 1; ++i)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (i >= k) continue;
                E[i][0][q] = n0[i][q]; E[i][1][q] = n1[i + 1][q];
            }
        Td[k] = -(s / c);
        gain *= c;
    }
    // what is left is R_0 = [[-c0, s0], [s0, c0]] (a reflection: det E = -z^-(K-1) for a mirror pair hi[t] = (-1)^t lo[L-1-t])
    const double c0 = E[0][1][1], s0 = E[0][1][0];
    res = ab(E[0][0][0] + c0) > res ? ab(E[0][0][0] + c0) : res;
    res = ab(E[0][0][1] - s0) > res ? ab(E[0][0][1] - s0) : res;
    if (!(ab(c0) > 1e-7)) return false;
    Td[0] = -(s0 / c0);
    gain *= c0;
    if (!(res <= tol * scale)) return false;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (!(ab(Td[k]) < 1e6)) return false;
        T[k] = (float)Td[k];
    }
    *g = (float)(-gain);
    return ab(gain) > 1e-30;
}

// d = w * (c, c) + z, c = the low (HI = 0) / high (HI = 1) half of a scalar-register pair; NZ: - z instead of + z
template <int HI, int NZ> WL_DEV wl_v2 wl_fma_s(wl_v2 w, wl_v2 pair, wl_v2 z) {
    wl_v2 d;
#if defined(__HIPCC__)
    if (!HI && !NZ) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(w), "s"(pair), "v"(z));
    else if (HI && !NZ) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(w), "s"(pair), "v"(z));
    else if (!HI && NZ) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(d) : "v"(w), "s"(pair), "v"(z));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(d) : "v"(w), "s"(pair), "v"(z));
#else
    const float c = HI ? pair.y : pair.x;
    d.x = __builtin_fmaf(w.x, c, NZ ? -z.x : z.x); d.y = __builtin_fmaf(w.y, c, NZ ? -z.y : z.y);
#endif
    return d;
}

// ---- the quadrature-mirror form of a tap pair -------------------------------------------------------------------------
// bank[u] = (lo[u], lo[L-1-u]), u < L/2, in scalar registers; the pair of tap t is (lo[t], hi[t]) with hi[t] = (-1)^t lo[L-1-t]:
// bank[u] or bank[u] with its halves swapped, the high half negated for odd t - operand modifiers of the packed FMA.
// acc (+)= (lo[t], hi[t]) * s.x (xy = 0) / s.y (xy = 1); t and xy are compile-time after unrolling: exactly one form survives.
template <int LT> WL_DEV void wl_qmf_fma(wl_v2& acc, const wl_v2 (&bank)[LT / 2], int t, int xy, wl_v2 s) {
    const int u = t < LT / 2 ? t : LT - 1 - t;
    const bool sw = t >= LT / 2, neg = t & 1;
#if defined(__HIPCC__)
    if (!sw && !neg) { if (xy) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(bank[u]), "v"(s));
                       else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "s"(bank[u]), "v"(s)); }
    else if (!sw && neg) { if (xy) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_hi:[1,0,0]" : "+v"(acc) : "s"(bank[u]), "v"(s));
                           else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1] neg_hi:[1,0,0]" : "+v"(acc) : "s"(bank[u]), "v"(s)); }
    else if (sw && !neg) { if (xy) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(bank[u]), "v"(s));
                           else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,0,1]" : "+v"(acc) : "s"(bank[u]), "v"(s)); }
    else { if (xy) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "+v"(acc) : "s"(bank[u]), "v"(s));
           else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "+v"(acc) : "s"(bank[u]), "v"(s)); }
#else
    const float c = xy ? s.y : s.x;
    const float lo = sw ? bank[u].y : bank[u].x, hi = (sw ? bank[u].x : bank[u].y) * (neg ? -1.f : 1.f);
    acc.x = __builtin_fmaf(lo, c, acc.x); acc.y = __builtin_fmaf(hi, c, acc.y);
#endif
}
template <int LT> WL_DEV wl_v2 wl_qmf_mul(const wl_v2 (&bank)[LT / 2], int t, int xy, wl_v2 s) {
    const int u = t < LT / 2 ? t : LT - 1 - t;
    const bool sw = t >= LT / 2, neg = t & 1;
    wl_v2 r;
#if defined(__HIPCC__)
    if (!sw && !neg) { if (xy) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "s"(bank[u]), "v"(s));
                       else asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "s"(bank[u]), "v"(s)); }
    else if (!sw && neg) { if (xy) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_hi:[1,0]" : "=v"(r) : "s"(bank[u]), "v"(s));
                           else asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0] neg_hi:[1,0]" : "=v"(r) : "s"(bank[u]), "v"(s)); }
    else if (sw && !neg) { if (xy) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(r) : "s"(bank[u]), "v"(s));
                           else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,0]" : "=v"(r) : "s"(bank[u]), "v"(s)); }
    else { if (xy) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1] neg_hi:[1,0]" : "=v"(r) : "s"(bank[u]), "v"(s));
           else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(r) : "s"(bank[u]), "v"(s)); }
#else
    const float c = xy ? s.y : s.x;
    const float lo = sw ? bank[u].y : bank[u].x, hi = (sw ? bank[u].x : bank[u].y) * (neg ? -1.f : 1.f);
    r.x = lo * c; r.y = hi * c;
#endif
    return r;
}

// synthesis: the highpass pair of tap pair j is the lowpass pair read backwards with its halves swapped and one half negated,
// (g1[2j], g1[2j+1]) = (g0[L-1-2j], -g0[L-2-2j]) = q(pair[L/2-1-j]), q(p) = (p.y, -p.x):  acc += q(pair) * s.x (Y = 0) / s.y (Y = 1)
template <int Y> WL_DEV void wl_qmf_syn_fma(wl_v2& acc, wl_v2 pair, wl_v2 sv) {
#if defined(__HIPCC__)
    if (Y) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "+v"(acc) : "s"(pair), "v"(sv));
    else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "+v"(acc) : "s"(pair), "v"(sv));
#else
    const float c = Y ? sv.y : sv.x;
    acc.x = __builtin_fmaf(pair.y, c, acc.x); acc.y = __builtin_fmaf(-pair.x, c, acc.y);
#endif
}

// ---- the one-thread kernel in front of a lattice launch ---------------------------------------------------------------
// Verdict + factorisation of the COLUMN bank (h_h_*) into `out` (WL_TAP_SCRATCH_FLOATS floats of device memory the caller
// owns for the duration of the launches that read it): accepted iff both highpass banks are the quadrature mirrors of their
// lowpass banks (the row pass of the lattice kernels is the QMF form) and the column bank factors to within `tol`.
struct WlTapPrepArgs {
    const float* h_w_lo; const float* h_w_hi; const float* h_h_lo; const float* h_h_hi;
    float* out;
    int L;
    float tol;
    int syn;               // the banks are a synthesis pair (WlSfbStrip) / an analysis pair (WlAfbStrip)
    int same;              // the kernel behind holds ONE bank for both axes (the fused multi-level kernels): w and h banks must be equal
};
// The CALLER-OWNED record of what a scratch block holds (`int* tap_state` of the *_ex entry points; zero for a fresh block, for
// other banks, another tap count or the other direction): bit 0 = this library ran WlTapPrep on exactly these banks into this
// block, bit 1 = that examination included "one bank for both axes" (`same`, what the fused multi-level kernels rely on).  Only
// the launcher that queues WlTapPrep sets the bits - a launch that never looked at the scratch (8 / 10 taps on the strip kernels,
// the direct-form variants) leaves them alone, so a later launch of the same transform never trusts an unexamined block.
static inline bool wl_tap_examined(const int* st, int need) { return st && (*st & need) == need; }
static inline void wl_tap_mark(int* st, int bits) { if (st) *st |= bits; }

template <int LT>
struct WlTapPrep {
    typedef WlTapPrepArgs Args;
    static const int kThreads = 64;
    static const int kMinWaves = 1;
    static WL_DEV void run(const Args& a, const WlCtx& ctx) {
        if (ctx.bid != 0) return;
        // (every lane of the one wave computes the same: the taps arrive by scalar loads, nothing diverges)
        float g = 0.f, T[LT / 2];
#pragma unroll
        for (int k = 0; k < LT / 2; ++k) T[k] = 0.f;
        bool ok = wl_taps_qmf(a.h_w_lo, a.h_w_hi, LT) && wl_taps_qmf(a.h_h_lo, a.h_h_hi, LT);
        if (a.same) ok = ok & wl_taps_same(a.h_w_lo, a.h_h_lo, LT) & wl_taps_same(a.h_w_hi, a.h_h_hi, LT);
        ok = wl_lattice_factor<LT>(a.h_h_lo, a.h_h_hi, (double)a.tol, &g, T, a.syn != 0) && ok;
        if (ctx.tid != 0) return;
        unsigned* flag = reinterpret_cast<unsigned*>(a.out);
        a.out[1] = g;
#pragma unroll
        for (int k = 0; k < LT / 2; ++k) a.out[2 + k] = T[k];
        flag[0] = ok ? WL_LAT_OK : 0u;
    }
};
