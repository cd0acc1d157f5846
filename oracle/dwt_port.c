/* CPU port of the oracle's DWT forward + inverse (TEST / BASELINE INFRASTRUCTURE ONLY).
 *
 * Plain C restatement of oracle/wavelet_oracle.py::dwt_forward / dwt_inverse (which restate the reference's
 * DWTForward / DWTInverse, dwt/transform2d.py:63-74, :131-148; 1-D banks dwt/lowlevel.py:91-172, :226-271),
 * float32, one OpenMP task per (n,c) plane.  Used only by bench.py's `cpu_baseline` leg ("kind": "port") and
 * by tests/test_oracle_port.py, which pins it against the numpy oracle.  Never linked into the product.
 *
 *   analysis : y_b[k] = sum_j h_b[j] * ext(x, 2k + base + j)      (h = stored/reversed taps)
 *   synthesis: y[n]   = sum_k' B(k') * g[n + s - 2k']
 */
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { EXT_ZERO = 0, EXT_SYM = 1, EXT_REFL = 2, EXT_PERIODIC = 3, EXT_PER = 4 };

static int pmod(int i, int p) { int r = i % p; return r < 0 ? r + p : r; }

static int ext_idx(int i, int n, int ext) {
    if ((unsigned)i < (unsigned)n) return i;
    switch (ext) {
        case EXT_ZERO: return -1;
        case EXT_SYM: { int j = pmod(i, 2 * n); return j < n ? j : 2 * n - 1 - j; }
        case EXT_REFL: { if (n == 1) return 0; int p = 2 * n - 2, j = pmod(i, p); return j < n ? j : p - j; }
        case EXT_PERIODIC: return pmod(i, n);
        default: { int ne = n + (n & 1), j = pmod(i, ne); return j == n ? n - 1 : j; }
    }
}

static int mode_ext(int mode) {
    switch (mode) { case 0: return EXT_ZERO; case 1: return EXT_SYM; case 2: return EXT_PER;
                    case 4: return EXT_REFL; case 6: return EXT_PERIODIC; default: return -1; }
}
static int coeff_len(int n, int L, int mode) { return mode == 2 ? (n + 1) / 2 : (n + L - 1) / 2; }
static int afb_base(int n, int L, int mode) {
    if (mode == 2) return L / 2 - L + 1;
    int K = (n + L - 1) / 2, p = 2 * (K - 1) - n + L;
    return -(p / 2);
}

/* one analysis level of one plane: x (H,W) -> ll (Kh,Kw), highs (3,Kh,Kw) */
static void afb2d_plane(const float* x, int H, int W, float* ll, float* highs, const float* h0, const float* h1,
                        int L, int mode, float* tmp /* H * Kw * 2 */, int* idx /* max(H,W)+L scratch */) {
    const int ext = mode_ext(mode);
    const int Kh = coeff_len(H, L, mode), Kw = coeff_len(W, L, mode);
    const int bw = afb_base(W, L, mode), bh = afb_base(H, L, mode);
    /* row bank: tmp[r][k][0..1] */
    for (int r = 0; r < H; ++r) {
        const float* xr = x + (size_t)r * W;
        for (int k = 0; k < Kw; ++k) {
            float lo = 0.f, hi = 0.f;
            const int e0 = 2 * k + bw;
            if (e0 >= 0 && e0 + L <= W) {
                for (int j = 0; j < L; ++j) { lo += h0[j] * xr[e0 + j]; hi += h1[j] * xr[e0 + j]; }
            } else {
                for (int j = 0; j < L; ++j) {
                    const int c = ext_idx(e0 + j, W, ext);
                    if (c >= 0) { lo += h0[j] * xr[c]; hi += h1[j] * xr[c]; }
                }
            }
            tmp[((size_t)r * Kw + k) * 2] = lo;
            tmp[((size_t)r * Kw + k) * 2 + 1] = hi;
        }
    }
    (void)idx;
    const size_t bp = (size_t)Kh * Kw;
    for (int kh = 0; kh < Kh; ++kh) {
        float* o_ll = ll + (size_t)kh * Kw;
        float* o_lh = highs + (size_t)kh * Kw;
        float* o_hl = highs + bp + (size_t)kh * Kw;
        float* o_hh = highs + 2 * bp + (size_t)kh * Kw;
        for (int k = 0; k < Kw; ++k) o_ll[k] = o_lh[k] = o_hl[k] = o_hh[k] = 0.f;
        for (int j = 0; j < L; ++j) {
            const int r = ext_idx(2 * kh + bh + j, H, ext);
            if (r < 0) continue;
            const float* t = tmp + (size_t)r * Kw * 2;
            const float a = h0[j], b = h1[j];
            for (int k = 0; k < Kw; ++k) {
                const float lo = t[2 * k], hi = t[2 * k + 1];
                o_ll[k] += a * lo; o_lh[k] += b * lo; o_hl[k] += a * hi; o_hh[k] += b * hi;
            }
        }
    }
}

/* one synthesis level of one plane: ll (Kh,Kw; row stride lls), highs (3,Kh,Kw) -> y (OH,OW) */
static void sfb2d_plane(const float* ll, int lls, const float* highs, int Kh, int Kw, float* y, int OH, int OW,
                        const float* g0, const float* g1, int L, int mode, float* tmp /* OH * Kw * 2 */) {
    const int circ = mode == 2;
    const int s = circ ? L / 2 - 1 : L - 2;
    const size_t bp = (size_t)Kh * Kw;
    for (int n = 0; n < OH; ++n) {
        float* t = tmp + (size_t)n * Kw * 2;
        for (int k = 0; k < 2 * Kw; ++k) t[k] = 0.f;
        const int nn = n + s;
        for (int tap = nn & 1; tap < L; tap += 2) {
            int kr = (nn - tap) / 2;            /* exact: nn - tap even */
            if (nn - tap < 0) kr = -((tap - nn) / 2);
            if (circ) kr = pmod(kr, Kh); else if (kr < 0 || kr >= Kh) continue;
            const float a = g0[tap], b = g1[tap];
            const float* r_ll = ll + (size_t)kr * lls;
            const float* r_lh = highs + (size_t)kr * Kw;
            const float* r_hl = highs + bp + (size_t)kr * Kw;
            const float* r_hh = highs + 2 * bp + (size_t)kr * Kw;
            for (int k = 0; k < Kw; ++k) {
                t[2 * k] += a * r_ll[k] + b * r_lh[k];
                t[2 * k + 1] += a * r_hl[k] + b * r_hh[k];
            }
        }
    }
    for (int n = 0; n < OH; ++n) {
        const float* t = tmp + (size_t)n * Kw * 2;
        float* yr = y + (size_t)n * OW;
        for (int w = 0; w < OW; ++w) {
            const int ww = w + s;
            float acc = 0.f;
            for (int tap = ww & 1; tap < L; tap += 2) {
                int kc = (ww - tap) / 2;
                if (ww - tap < 0) kc = -((tap - ww) / 2);
                if (circ) kc = pmod(kc, Kw); else if (kc < 0 || kc >= Kw) continue;
                acc += g0[tap] * t[2 * kc] + g1[tap] * t[2 * kc + 1];
            }
            yr[w] = acc;
        }
    }
}

/* Forward: x (planes,H,W) -> yl, yh[j] packed back to back in `out` in the order yh_0, yh_1, .., yh_{J-1}, yl.
 * Returns the number of floats written per plane (or -1).  `out` may be NULL to query the size. */
long wl_port_dwt_forward(const float* x, float* out, long planes, int H, int W, int J, const float* h0,
                         const float* h1, int L, int mode, int threads) {
    if (mode_ext(mode) < 0 || J < 1 || J > 16) return -1;
    int hs[17], ws[17];
    hs[0] = H; ws[0] = W;
    long per = 0;
    for (int j = 0; j < J; ++j) {
        hs[j + 1] = coeff_len(hs[j], L, mode); ws[j + 1] = coeff_len(ws[j], L, mode);
        per += 3L * hs[j + 1] * ws[j + 1];
    }
    per += (long)hs[J] * ws[J];
    if (!out) return per;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        float* tmp = (float*)malloc(sizeof(float) * (size_t)H * ws[1] * 2);
        float* llA = (float*)malloc(sizeof(float) * (size_t)hs[1] * ws[1]);
        float* llB = (float*)malloc(sizeof(float) * (size_t)hs[1] * ws[1]);
#pragma omp for schedule(dynamic, 1)
        for (long p = 0; p < planes; ++p) {
            const float* src = x + (size_t)p * H * W;
            float* o = out + (size_t)p * per;
            float* cur = llA, *nxt = llB;
            for (int j = 0; j < J; ++j) {
                float* lldst = (j == J - 1) ? o + 3L * hs[j + 1] * ws[j + 1] : cur;
                afb2d_plane(src, hs[j], ws[j], lldst, o, h0, h1, L, mode, tmp, 0);
                o += 3L * hs[j + 1] * ws[j + 1];
                src = lldst;
                float* t = cur; cur = nxt; nxt = t;
            }
        }
        free(tmp); free(llA); free(llB);
    }
    return per;
}

/* Inverse of the packing above: coeffs (planes, per) -> y (planes,H,W). */
long wl_port_dwt_inverse(const float* coeffs, float* y, long planes, int H, int W, int J, const float* g0,
                         const float* g1, int L, int mode, int threads) {
    if (mode_ext(mode) < 0 || J < 1 || J > 16) return -1;
    int hs[17], ws[17];
    long offs[17];
    hs[0] = H; ws[0] = W;
    long per = 0;
    for (int j = 0; j < J; ++j) {
        hs[j + 1] = coeff_len(hs[j], L, mode); ws[j + 1] = coeff_len(ws[j], L, mode);
        offs[j] = per;
        per += 3L * hs[j + 1] * ws[j + 1];
    }
    offs[J] = per;
    per += (long)hs[J] * ws[J];
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        const size_t big = (size_t)(2 * hs[1] + 2) * (2 * ws[1] + 2);
        float* tmp = (float*)malloc(sizeof(float) * big * 2);
        float* bufA = (float*)malloc(sizeof(float) * big);
        float* bufB = (float*)malloc(sizeof(float) * big);
#pragma omp for schedule(dynamic, 1)
        for (long p = 0; p < planes; ++p) {
            const float* c = coeffs + (size_t)p * per;
            const float* ll = c + offs[J];
            int llh = hs[J], llw = ws[J], lls = ws[J];
            float* cur = bufA, *nxt = bufB;
            for (int j = J - 1; j >= 0; --j) {
                const int Kh = hs[j + 1], Kw = ws[j + 1];
                /* 'unpad': the ll coming from the coarser level may be one row/col larger (transform2d.py:142-145) */
                (void)llh; (void)llw;
                const int OHf = mode == 2 ? 2 * Kh : 2 * Kh - L + 2, OWf = mode == 2 ? 2 * Kw : 2 * Kw - L + 2;
                float* dst = j == 0 ? y + (size_t)p * H * W : cur;
                const int OH = j == 0 ? (OHf < H ? OHf : H) : OHf, OW = j == 0 ? (OWf < W ? OWf : W) : OWf;
                sfb2d_plane(ll, lls, c + offs[j], Kh, Kw, dst, OH, OW, g0, g1, L, mode, tmp);
                ll = dst; llh = OH; llw = OW; lls = OW;
                float* t = cur; cur = nxt; nxt = t;
            }
        }
        free(tmp); free(bufA); free(bufB);
    }
    return per;
}
