// DTCWT per-level tile kernels (any tap count, float / half / double data).
//
// Each 256-thread workgroup produces one output tile of one (n,c) plane; boundary extension (symmetric
// half-sample, or zero padding at level 1) and the replicate padding the reference applies to odd /
// non-multiple-of-4 sizes are index math while the input tile is staged into LDS; the row and the column
// filter banks run out of LDS and the quad -> complex packing (q2c) / its inverse (c2q) are fused into the
// epilogue / the staging, so the six oriented complex sub-bands are written / read directly in the
// reference's (N, C, 6, H, W, 2) layout.
//
//   wl_dtcwt_fwd1_body : fwd_j1   (reference dtcwt/transform_funcs.py:98-121: 2 x rowfilter + 4 x colfilter +
//                                  3 x q2c + 2 x stack + stack)  and, with `scat`, the ScatLayer epilogue
//                                  (scatternet/lowlevel.py:86-109: avg_pool2d(ll,2), sqrt(re^2+im^2+b^2)-b, cat)
//   wl_dtcwt_fwd2_body : fwd_j2plus (transform_funcs.py:226-249: 2 x rowdfilt + 4 x coldfilt + q2c)
//   wl_dtcwt_inv1_body : inv_j1   (transform_funcs.py:152-184: c2q x 3, 4 x colfilter, 2 x rowfilter, 3 adds)
//   wl_dtcwt_inv2_body : inv_j2plus (transform_funcs.py:279-307: c2q x 3, 4 x colifilt, 2 x rowifilt, 3 adds)
// Filters are the reference's stored buffers (reversed columns), 1-D formulas from dtcwt/lowlevel.py:70-239.
#pragma once
#include "wl_common.h"
#include <math.h>

#define WL_SQRT1_2 0.70710678118654752440
// The magnitudes of the scattering epilogues: float32 square root and reciprocal as ONE hardware instruction each (v_sqrt_f32 /
// v_rcp_f32, 1 ulp).  `sqrtf` / `/` compile to the correctly rounded sequences (scaling for denormals, the instruction, two
// fused multiply-adds and three selects to fix the last bit: 16 instructions per root - six roots per quad made a third of what a
// level-1 lane of the ScatLayer kernel issues, and that kernel is bound by the instructions it issues).  The parity tolerance is
// 1e-5 of the largest value; float64 planes keep the exact forms.
#if defined(__HIPCC__)
WL_DEV float wl_sqrt(float v) { return __builtin_amdgcn_sqrtf(v); }
WL_DEV float wl_rcp(float v) { return __builtin_amdgcn_rcpf(v); }
#else
WL_DEV float wl_sqrt(float v) { return sqrtf(v); }
WL_DEV float wl_rcp(float v) { return 1.0f / v; }
#endif
WL_DEV double wl_sqrt(double v) { return sqrt(v); }
WL_DEV double wl_rcp(double v) { return 1.0 / v; }

// orientation slots: lh -> (0, 5), hh -> (1, 4), hl -> (2, 3)   (transform_funcs.py:61-72)

// ---------------------------------------------------------------------------------------------------------
// level 1 forward
// ---------------------------------------------------------------------------------------------------------
template <typename T>
struct WlDtFwd1Args {
    typedef typename WlAcc<T>::type A;
    const T* x;      // (NC, H, W)
    T* ll;           // (NC, He, We) or nullptr
    T* highs;        // (NC, 6, He/2, We/2, 2) or nullptr (skip_hps)
    T* z;            // ScatLayer output (N, 7, C, He/2, We/2) [or (N, 3+6, ..) when combine] or nullptr
    // where the entries of image n go (elements; not when combining colour): the averaged lowpass of channel c at
    // z + n z_bs + z_ll_off + c q, magnitude o at z + n z_bs + z_mag_off + (o C + c) q, q = (He/2)(We/2).  The layer's own
    // layout is z_bs = 7 C q, z_ll_off = 0, z_mag_off = C q; ScatLayerj2 points them into its 49-entry output (z_ll_off < 0:
    // no lowpass entry).
    int64_t z_bs, z_ll_off, z_mag_off;
    T* drdx;         // ScatLayer saved re/r, im/r: (N, 6, C, He/2, We/2) or nullptr
    T* drdy;
    const A* h0;     // lowpass taps (L0, odd), stored order
    const A* h1;     // highpass taps (L1, odd)
    int64_t NC;
    int C, H, W, He, We;
    int L0, L1, M, ext;   // M = max(L0,L1)/2
    int TH, TW, tiles_x, tiles_y;
    int64_t nblocks;       // specialised kernels: grid size for the XCD-aware block remap (0 = off)
    int run_len, runs_x;   // specialised kernel: tiles per workgroup along x, ceil(tiles_x / run_len)
    int combine;     // ScatLayer combine_colour (C == 3)
    A magbias;
    // small-plane kernel (wl_dtcwt_small.h): ceil(2^32 / d) (0 for d = 1) for d = (H + 2M)(W + 2M), W + 2M, W, (H/2)(W/2), W/2
    unsigned mg_q, mg_w, mg_w2, nblocks_q, mg_qc;
};

template <typename T>
WL_HD size_t wl_dtfwd1_lds(const WlDtFwd1Args<T>& a) {
    typedef typename WlAcc<T>::type A;
    const size_t nr = a.TH + 2 * a.M, nc = a.TW + 2 * a.M;
    return sizeof(A) * (wl_align_up(a.L0 + a.L1, 4) + nr * (nc | 1) + nr * a.TW * 2);
}

// element-aligned pair (8-byte stores of two adjacent samples / one complex value)
template <typename T> struct __attribute__((packed, aligned(sizeof(T)), may_alias)) WlPair { T a, b; };

// q2c (transform_funcs.py:61-72): quad samples p = 0:(r,c) 1:(r,c+1) 2:(r+1,c) 3:(r+1,c+1) of the band `v` ->
// orientations (o1, o2):  z1 = ((a-d) + i(b+c))/sqrt2,  z2 = ((a+d) + i(b-c))/sqrt2
template <typename A>
WL_DEV void wl_q2c(const A* v, A& re1, A& im1, A& re2, A& im2) {
    const A k = (A)WL_SQRT1_2;
    re1 = (v[0] - v[3]) * k; im1 = (v[1] + v[2]) * k;
    re2 = (v[0] + v[3]) * k; im2 = (v[1] - v[2]) * k;
}

// Output of one 2x2 quad (full-res origin R, Cc; both even) of plane `plane`, shared by the generic and the
// specialised level-1 kernels: lowpass, the six complex orientations and / or the ScatLayer epilogue.
// `msum` = this quad's six running sums over colour planes (combine_colour), `ch` the colour index.
// COMB: compile-time combine_colour (0 / 1), or -1 = decided at run time by a.combine.
template <typename T, int COMB = -1, typename A>
WL_DEV void wl_dtfwd1_quad_out(const WlDtFwd1Args<T>& a, int64_t plane, int ch, int R, int Cc, const A* ll,
                               const A* lh, const A* hl, const A* hh, A* msum) {
    const bool combine = COMB < 0 ? (a.combine != 0) : (COMB != 0);
    typedef WlPair<T> Pair;
    const int w2 = a.We / 2;
    const size_t qplane = (size_t)(a.He / 2) * w2;
    if (a.ll) {
        T* lp = a.ll + (size_t)plane * a.He * a.We + (size_t)R * a.We + Cc;
        Pair p0, p1;
        p0.a = (T)ll[0]; p0.b = (T)ll[1]; p1.a = (T)ll[2]; p1.b = (T)ll[3];
        *reinterpret_cast<Pair*>(lp) = p0;
        *reinterpret_cast<Pair*>(lp + a.We) = p1;
    }
    if (!a.highs && !a.z) return;
    A re[6], im[6];
    wl_q2c(lh, re[0], im[0], re[5], im[5]);
    wl_q2c(hh, re[1], im[1], re[4], im[4]);
    wl_q2c(hl, re[2], im[2], re[3], im[3]);
    const size_t q = (size_t)(R / 2) * w2 + (Cc / 2);
    if (a.highs) {
        T* hp = a.highs + (size_t)plane * 12 * qplane;
#pragma unroll
        for (int o = 0; o < 6; ++o) {
            Pair p; p.a = (T)re[o]; p.b = (T)im[o];
            *reinterpret_cast<Pair*>(hp + ((size_t)o * qplane + q) * 2) = p;
        }
    }
    if (!a.z) return;
    const int64_t n = plane / a.C;
    const int c = (int)(plane - n * a.C);
    const A b2 = a.magbias * a.magbias;
    const A llavg = (ll[0] + ll[1] + ll[2] + ll[3]) * (A)0.25;
    if (!combine) {
        T* const zn = a.z + (size_t)n * a.z_bs + (size_t)c * qplane + q;
        if (a.z_ll_off >= 0) zn[a.z_ll_off] = (T)llavg;
        T* const zp = zn + a.z_mag_off;
#pragma unroll
        for (int o = 0; o < 6; ++o) {
            const A r = wl_sqrt(re[o] * re[o] + im[o] * im[o] + b2);
            zp[(size_t)o * a.C * qplane] = (T)(r - a.magbias);
            if (a.drdx) {
                const size_t so = (((size_t)n * 6 + o) * a.C + c) * qplane + q;
                const A ir = wl_rcp(r);
                a.drdx[so] = (T)(re[o] * ir);
                a.drdy[so] = (T)(im[o] * ir);
            }
        }
    } else {
        T* zp = a.z + (size_t)n * 9 * qplane + q;
        zp[(size_t)c * qplane] = (T)llavg;
#pragma unroll
        for (int o = 0; o < 6; ++o) {
            const A e = re[o] * re[o] + im[o] * im[o];
            msum[o] = ch == 0 ? e : msum[o] + e;
            if (a.drdx) {   // numerators now, divided by r in the last pass
                const size_t so = (((size_t)n * 6 + o) * 3 + c) * qplane + q;
                a.drdx[so] = (T)re[o];
                a.drdy[so] = (T)im[o];
            }
        }
        if (ch == 2) {
#pragma unroll
            for (int o = 0; o < 6; ++o) {
                const A r = wl_sqrt(msum[o] + b2);
                zp[(size_t)(3 + o) * qplane] = (T)(r - a.magbias);
                if (a.drdx) {
                    for (int c3 = 0; c3 < 3; ++c3) {
                        const size_t so = (((size_t)n * 6 + o) * 3 + c3) * qplane + q;
                        a.drdx[so] = (T)((A)a.drdx[so] * wl_rcp(r));
                        a.drdy[so] = (T)((A)a.drdy[so] * wl_rcp(r));
                    }
                }
            }
        }
    }
}

template <typename T>
WL_DEV void wl_dtcwt_fwd1_body(const WlDtFwd1Args<T>& a, const WlCtx& ctx) {
    typedef typename WlAcc<T>::type A;
    const int tx = ctx.tid & 63, ty = ctx.tid >> 6, ny = ctx.nthreads >> 6;
    const int tiles = a.tiles_x * a.tiles_y;
    const int64_t unit = ctx.bid / tiles;            // plane, or image when combining colour
    const int tile = (int)(ctx.bid - unit * tiles);
    const int r0 = (tile / a.tiles_x) * a.TH, c0 = (tile % a.tiles_x) * a.TW;
    const int nr = a.TH + 2 * a.M, nc = a.TW + 2 * a.M, sp = nc | 1;
    A* taps = reinterpret_cast<A*>(ctx.smem);
    A* S = taps + wl_align_up(a.L0 + a.L1, 4);
    A* Tm = S + (size_t)nr * sp;
    A* t0 = taps, *t1 = taps + a.L0;
    for (int i = ctx.tid; i < a.L0; i += ctx.nthreads) t0[i] = a.h0[i];
    for (int i = ctx.tid; i < a.L1; i += ctx.nthreads) t1[i] = a.h1[i];
    const int m0 = a.L0 / 2, m1 = a.L1 / 2;
    const int nch = a.combine ? 3 : 1;
    // combine_colour accumulates re^2+im^2 over the 3 colour planes; each thread owns the same quads for all
    A msum[4][6];
    for (int ch = 0; ch < nch; ++ch) {
        const int64_t plane = a.combine ? unit * 3 + ch : unit;
        const T* xp = a.x + (size_t)plane * a.H * a.W;
        if (ch) ctx.sync();
        for (int i = ty; i < nr; i += ny) {
            const int r = wl_ext_padded(r0 - a.M + i, a.H, 0, a.He - a.H, a.ext);
            for (int j = tx; j < nc; j += 64) {
                const int c = wl_ext_padded(c0 - a.M + j, a.W, 0, a.We - a.W, a.ext);
                S[i * sp + j] = (r < 0 || c < 0) ? (A)0 : (A)xp[(size_t)r * a.W + c];
            }
        }
        ctx.sync();
        for (int i = ty; i < nr; i += ny) {
            const A* s = S + i * sp;
            for (int j = tx; j < a.TW; j += 64) {
                A lo = 0, hi = 0;
                for (int t = 0; t < a.L0; ++t) lo += t0[t] * s[j + a.M - m0 + t];
                for (int t = 0; t < a.L1; ++t) hi += t1[t] * s[j + a.M - m1 + t];
                Tm[(i * a.TW + j) * 2] = lo;
                Tm[(i * a.TW + j) * 2 + 1] = hi;
            }
        }
        ctx.sync();
        // one thread per 2x2 quad: 4 positions x 4 bands, then q2c
        int slot = 0;
        for (int qr = ty; qr < a.TH / 2; qr += ny) {
            for (int qc = tx; qc < a.TW / 2; qc += 64, ++slot) {
                const int R = r0 + 2 * qr, Cc = c0 + 2 * qc;
                if (R >= a.He || Cc >= a.We) continue;
                A ll[4], lh[4], hl[4], hh[4];
                for (int p = 0; p < 4; ++p) {
                    const int rr = 2 * qr + (p >> 1), cc = 2 * qc + (p & 1);
                    A vll = 0, vlh = 0, vhl = 0, vhh = 0;
                    for (int t = 0; t < a.L0; ++t) {
                        const A* e = Tm + ((rr + a.M - m0 + t) * a.TW + cc) * 2;
                        vll += t0[t] * e[0];
                        vhl += t0[t] * e[1];
                    }
                    for (int t = 0; t < a.L1; ++t) {
                        const A* e = Tm + ((rr + a.M - m1 + t) * a.TW + cc) * 2;
                        vlh += t1[t] * e[0];
                        vhh += t1[t] * e[1];
                    }
                    ll[p] = vll; lh[p] = vlh; hl[p] = vhl; hh[p] = vhh;
                }
                wl_dtfwd1_quad_out<T>(a, plane, ch, R, Cc, ll, lh, hl, hh, msum[slot & 3]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// level >= 2 forward (dual-tree decimating filters)
//   Y[2k+s] = sum_t f_s[t] * sym(X, 4k + 2t + d_s - L);  lowpass: (f0,d0,f1,d1) = (h0b,2,h0a,3)
//                                                        highpass:               (h1a,3,h1b,2)
// ---------------------------------------------------------------------------------------------------------
template <typename T>
struct WlDtFwd2Args {
    typedef typename WlAcc<T>::type A;
    const T* x;       // (NC, H, W)
    T* ll;            // (NC, He/2, We/2)
    T* highs;         // (NC, 6, He/4, We/4, 2) or nullptr
    const A* h0a; const A* h0b; const A* h1a; const A* h1b;   // L taps each (even L)
    int64_t NC;
    int H, W, He, We, padr, padc;   // virtual size (multiple of 4) and replicate pad (0/1) on both sides
    int L;
    int TH, TW, tiles_x, tiles_y;   // output (half-res) tile, multiples of 4
    int64_t nblocks;                // specialised kernel: grid size for the XCD-aware block remap (0 = off)
    // ScatLayerj2's second scale (streaming kernel only): instead of ll / highs the 2x2-averaged lowpass and the smoothed
    // magnitudes go to z, addressed like WlDtFwd1Args::z (q = (H/4)(W/4)); nullptr = off
    T* z;
    int64_t z_bs, z_ll_off, z_mag_off;
    int C;
    A magbias;
};

template <typename T>
WL_HD size_t wl_dtfwd2_lds(const WlDtFwd2Args<T>& a) {
    typedef typename WlAcc<T>::type A;
    const size_t nr = 2 * a.TH + 2 * a.L - 4, nc = 2 * a.TW + 2 * a.L - 4;
    return sizeof(A) * (4 * (size_t)a.L + nr * (nc | 1) + nr * a.TW * 2);
}

template <typename T>
WL_DEV void wl_dtcwt_fwd2_body(const WlDtFwd2Args<T>& a, const WlCtx& ctx) {
    typedef typename WlAcc<T>::type A;
    const int tx = ctx.tid & 63, ty = ctx.tid >> 6, ny = ctx.nthreads >> 6;
    const int tiles = a.tiles_x * a.tiles_y;
    const int64_t plane = ctx.bid / tiles;
    const int tile = (int)(ctx.bid - plane * tiles);
    const int r0 = (tile / a.tiles_x) * a.TH, c0 = (tile % a.tiles_x) * a.TW;   // half-res output origin
    const int L = a.L;
    const int nr = 2 * a.TH + 2 * L - 4, nc = 2 * a.TW + 2 * L - 4, sp = nc | 1;
    A* taps = reinterpret_cast<A*>(ctx.smem);
    A* S = taps + 4 * L;
    A* Tm = S + (size_t)nr * sp;
    for (int i = ctx.tid; i < L; i += ctx.nthreads) {
        taps[i] = a.h0a[i]; taps[L + i] = a.h0b[i]; taps[2 * L + i] = a.h1a[i]; taps[3 * L + i] = a.h1b[i];
    }
    const A* h0a = taps, *h0b = taps + L, *h1a = taps + 2 * L, *h1b = taps + 3 * L;
    const T* xp = a.x + (size_t)plane * a.H * a.W;
    // input origin: output k0 = r0/2 needs X[4*k0 + 2 - L ...]
    const int er0 = 2 * r0 + 2 - L, ec0 = 2 * c0 + 2 - L;
    for (int i = ty; i < nr; i += ny) {
        const int r = wl_ext_padded(er0 + i, a.H, a.padr, a.padr, WL_EXT_SYM);
        for (int j = tx; j < nc; j += 64) {
            const int c = wl_ext_padded(ec0 + j, a.W, a.padc, a.padc, WL_EXT_SYM);
            S[i * sp + j] = (A)xp[(size_t)r * a.W + c];
        }
    }
    ctx.sync();
    // row pass: output column jo = 2kk + s reads S[i][4kk + 2t + (d_s - 2)]
    for (int i = ty; i < nr; i += ny) {
        const A* s = S + i * sp;
        for (int jo = tx; jo < a.TW; jo += 64) {
            const int kk = jo >> 1, par = jo & 1;
            const A* fl = par ? h0a : h0b;          // lowpass: (h0b, d=2), (h0a, d=3)
            const A* fh = par ? h1b : h1a;          // highpass: (h1a, d=3), (h1b, d=2)
            const int dl = par ? 1 : 0, dh = par ? 0 : 1;
            A lo = 0, hi = 0;
            for (int t = 0; t < L; ++t) {
                lo += fl[t] * s[4 * kk + 2 * t + dl];
                hi += fh[t] * s[4 * kk + 2 * t + dh];
            }
            Tm[(i * a.TW + jo) * 2] = lo;
            Tm[(i * a.TW + jo) * 2 + 1] = hi;
        }
    }
    ctx.sync();
    const int h2 = a.He / 2, w2 = a.We / 2, h4 = a.He / 4, w4 = a.We / 4;
    const size_t qplane = (size_t)h4 * w4;
    for (int qr = ty; qr < a.TH / 2; qr += ny) {
        for (int qc = tx; qc < a.TW / 2; qc += 64) {
            const int R = r0 + 2 * qr, Cc = c0 + 2 * qc;   // half-res coordinates of the quad
            if (R >= h2 || Cc >= w2) continue;
            A ll[4], lh[4], hl[4], hh[4];
            for (int p = 0; p < 4; ++p) {
                const int ro = 2 * qr + (p >> 1), co = 2 * qc + (p & 1);
                const int kk = ro >> 1, par = ro & 1;
                const A* fl = par ? h0a : h0b;
                const A* fh = par ? h1b : h1a;
                const int dl = par ? 1 : 0, dh = par ? 0 : 1;
                A vll = 0, vlh = 0, vhl = 0, vhh = 0;
                for (int t = 0; t < L; ++t) {
                    const A* el = Tm + ((4 * kk + 2 * t + dl) * a.TW + co) * 2;
                    const A* eh = Tm + ((4 * kk + 2 * t + dh) * a.TW + co) * 2;
                    vll += fl[t] * el[0];
                    vhl += fl[t] * el[1];
                    vlh += fh[t] * eh[0];
                    vhh += fh[t] * eh[1];
                }
                ll[p] = vll; lh[p] = vlh; hl[p] = vhl; hh[p] = vhh;
            }
            T* lp = a.ll + (size_t)plane * h2 * w2 + (size_t)R * w2 + Cc;
            lp[0] = (T)ll[0]; lp[1] = (T)ll[1]; lp[w2] = (T)ll[2]; lp[w2 + 1] = (T)ll[3];
            if (!a.highs) continue;
            A re[6], im[6];
            const A k = (A)WL_SQRT1_2;
            re[0] = (lh[0] - lh[3]) * k; im[0] = (lh[1] + lh[2]) * k; re[5] = (lh[0] + lh[3]) * k; im[5] = (lh[1] - lh[2]) * k;
            re[1] = (hh[0] - hh[3]) * k; im[1] = (hh[1] + hh[2]) * k; re[4] = (hh[0] + hh[3]) * k; im[4] = (hh[1] - hh[2]) * k;
            re[2] = (hl[0] - hl[3]) * k; im[2] = (hl[1] + hl[2]) * k; re[3] = (hl[0] + hl[3]) * k; im[3] = (hl[1] - hl[2]) * k;
            const size_t q = (size_t)(R / 2) * w4 + (Cc / 2);
            T* hp = a.highs + (size_t)plane * 12 * qplane;
            for (int o = 0; o < 6; ++o) {
                hp[((size_t)o * qplane + q) * 2] = (T)re[o];
                hp[((size_t)o * qplane + q) * 2 + 1] = (T)im[o];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// shared by the inverse kernels: value of the quad-band `b` (0 lh, 1 hl, 2 hh) at full-res (r, c) from the
// complex orientations (c2q, dtcwt/lowlevel.py:263-295; pairs lh<-(0,5) hl<-(2,3) hh<-(1,4))
// ---------------------------------------------------------------------------------------------------------
template <typename T, typename A>
WL_DEV A wl_c2q_at(const T* hp, size_t qplane, int w2, int b, int r, int c) {
    const int o1 = b == 0 ? 0 : (b == 1 ? 2 : 1), o2 = b == 0 ? 5 : (b == 1 ? 3 : 4);
    const size_t q = (size_t)(r >> 1) * w2 + (c >> 1);
    const T* p1 = hp + ((size_t)o1 * qplane + q) * 2;
    const T* p2 = hp + ((size_t)o2 * qplane + q) * 2;
    const A w1r = (A)p1[0], w1i = (A)p1[1], w2r = (A)p2[0], w2i = (A)p2[1];
    const int sub = ((r & 1) << 1) | (c & 1);
    const A v = sub == 0 ? w1r + w2r : (sub == 1 ? w1i + w2i : (sub == 2 ? w1i - w2i : w2r - w1r));
    return v * (A)WL_SQRT1_2;
}

// ---------------------------------------------------------------------------------------------------------
// level 1 inverse:  lo = cf(lh,g1)+cf(ll,g0); hi = cf(hh,g1)+cf(hl,g0); y = rf(hi,g1)+rf(lo,g0)
// ---------------------------------------------------------------------------------------------------------
template <typename T>
struct WlDtInv1Args {
    typedef typename WlAcc<T>::type A;
    const T* ll;      // (NC, >=H, >=W) through strides (+ crop offset already applied) or nullptr
    const T* highs;   // (NC, 6, H/2, W/2, 2) or nullptr
    T* y;             // (NC, H, W)
    const A* g0; const A* g1;   // L0, L1 taps (odd)
    int64_t NC;
    int64_t ll_plane_stride;
    int ll_row_stride;
    int H, W, L0, L1, M, ext;
    int TH, TW, tiles_x, tiles_y;
    int64_t nblocks;       // specialised kernel: grid size for the XCD-aware block remap (0 = off)
    // fused ScatLayer backward (specialised kernel only): dZ, re/r, im/r instead of ll / highs
    const T* sz; const T* sdx; const T* sdy;
    int C, combine;
};

template <typename T>
WL_HD size_t wl_dtinv1_lds(const WlDtInv1Args<T>& a) {
    typedef typename WlAcc<T>::type A;
    const size_t nr = a.TH + 2 * a.M, nc = a.TW + 2 * a.M;
    return sizeof(A) * (wl_align_up(a.L0 + a.L1, 4) + 4 * nr * nc + 2 * (size_t)a.TH * nc);
}

template <typename T>
WL_DEV void wl_dtcwt_inv1_body(const WlDtInv1Args<T>& a, const WlCtx& ctx) {
    typedef typename WlAcc<T>::type A;
    const int tx = ctx.tid & 63, ty = ctx.tid >> 6, ny = ctx.nthreads >> 6;
    const int tiles = a.tiles_x * a.tiles_y;
    const int64_t plane = ctx.bid / tiles;
    const int tile = (int)(ctx.bid - plane * tiles);
    const int r0 = (tile / a.tiles_x) * a.TH, c0 = (tile % a.tiles_x) * a.TW;
    const int nr = a.TH + 2 * a.M, nc = a.TW + 2 * a.M;
    A* taps = reinterpret_cast<A*>(ctx.smem);
    A* B = taps + wl_align_up(a.L0 + a.L1, 4);     // [4][nr][nc]: ll, lh, hl, hh
    A* U = B + (size_t)4 * nr * nc;                // [2][TH][nc]: lo, hi
    A* t0 = taps, *t1 = taps + a.L0;
    for (int i = ctx.tid; i < a.L0; i += ctx.nthreads) t0[i] = a.g0[i];
    for (int i = ctx.tid; i < a.L1; i += ctx.nthreads) t1[i] = a.g1[i];
    const int m0 = a.L0 / 2, m1 = a.L1 / 2;
    const int w2 = a.W / 2;
    const size_t qplane = (size_t)(a.H / 2) * w2;
    const T* llp = a.ll ? a.ll + (size_t)plane * a.ll_plane_stride : nullptr;
    const T* hp = a.highs ? a.highs + (size_t)plane * 12 * qplane : nullptr;
    const size_t bs = (size_t)nr * nc;
    for (int i = ty; i < nr; i += ny) {
        const int r = wl_ext(r0 - a.M + i, a.H, a.ext);
        for (int j = tx; j < nc; j += 64) {
            const int c = wl_ext(c0 - a.M + j, a.W, a.ext);
            A v0 = 0, v1 = 0, v2 = 0, v3 = 0;
            if (r >= 0 && c >= 0) {
                if (llp) v0 = (A)llp[(size_t)r * a.ll_row_stride + c];
                if (hp) {
                    v1 = wl_c2q_at<T, A>(hp, qplane, w2, 0, r, c);
                    v2 = wl_c2q_at<T, A>(hp, qplane, w2, 1, r, c);
                    v3 = wl_c2q_at<T, A>(hp, qplane, w2, 2, r, c);
                }
            }
            const size_t o = (size_t)i * nc + j;
            B[o] = v0; B[bs + o] = v1; B[2 * bs + o] = v2; B[3 * bs + o] = v3;
        }
    }
    ctx.sync();
    for (int i = ty; i < a.TH; i += ny) {
        for (int j = tx; j < nc; j += 64) {
            A lo = 0, hi = 0;
            for (int t = 0; t < a.L0; ++t) {
                const size_t o = (size_t)(i + a.M - m0 + t) * nc + j;
                lo += t0[t] * B[o];            // cf(ll, g0)
                hi += t0[t] * B[2 * bs + o];   // cf(hl, g0)
            }
            for (int t = 0; t < a.L1; ++t) {
                const size_t o = (size_t)(i + a.M - m1 + t) * nc + j;
                lo += t1[t] * B[bs + o];       // cf(lh, g1)
                hi += t1[t] * B[3 * bs + o];   // cf(hh, g1)
            }
            U[(size_t)i * nc + j] = lo;
            U[(size_t)(a.TH + i) * nc + j] = hi;
        }
    }
    ctx.sync();
    T* yp = a.y + (size_t)plane * a.H * a.W;
    for (int i = ty; i < a.TH; i += ny) {
        if (r0 + i >= a.H) break;
        const A* ulo = U + (size_t)i * nc;
        const A* uhi = U + (size_t)(a.TH + i) * nc;
        for (int j = tx; j < a.TW; j += 64) {
            if (c0 + j >= a.W) break;
            A acc = 0;
            for (int t = 0; t < a.L0; ++t) acc += t0[t] * ulo[j + a.M - m0 + t];
            for (int t = 0; t < a.L1; ++t) acc += t1[t] * uhi[j + a.M - m1 + t];
            yp[(size_t)(r0 + i) * a.W + (c0 + j)] = (T)acc;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// level >= 2 inverse (dual-tree interpolating filters, dtcwt/lowlevel.py:154-239)
//   Y[4q+s] = sum_{t<m2} f_s[t] * sym(X, o_s - m2 + 2(q+t)),   f_s[t] = h_s[e_s + 2t]
//   m2 even: (h,e,o)_s = (ha,0,0) (hb,0,1) (ha,1,2) (hb,1,3)   highpass: o = (1,0,3,2)
//   m2 odd : (h,e,o)_s = (ha,1,1) (hb,1,2) (ha,0,1) (hb,0,2)   highpass: o = (2,1,2,1)
//   lowpass streams use (ha,hb) = (g0b,g0a), highpass streams (g1b,g1a)   (transform_funcs.py:299-306)
// ---------------------------------------------------------------------------------------------------------
template <typename T>
struct WlDtInv2Args {
    typedef typename WlAcc<T>::type A;
    const T* ll;      // (NC, h, w) through strides, or nullptr
    const T* highs;   // (NC, 6, h/2, w/2, 2) or nullptr
    T* y;             // (NC, 2h, 2w)
    const A* g0a; const A* g0b; const A* g1a; const A* g1b;
    int64_t NC;
    int64_t ll_plane_stride;
    int ll_row_stride;
    int h, w, L;
    int TH, TW, tiles_x, tiles_y;   // output tile, multiples of 4
    int64_t nblocks;                // specialised kernel: grid size for the XCD-aware block remap (0 = off)
};

template <typename T>
WL_HD size_t wl_dtinv2_lds(const WlDtInv2Args<T>& a) {
    typedef typename WlAcc<T>::type A;
    const size_t nr = a.TH / 2 + a.L + 2, nc = a.TW / 2 + a.L + 2;
    return sizeof(A) * (4 * (size_t)a.L + 4 * nr * nc + 2 * (size_t)a.TH * nc);
}

// interpolated sample: output index `n` (= 4q+s) of the stream pair (ha,hb,highpass) read from a staged line
// `x` (stride `st`) whose element 0 is input position `base`
template <typename A>
WL_DEV A wl_ifilt_at(const A* x, int st, int base, int n, const A* ha, const A* hb, int m2, bool highpass) {
    const int q = n >> 2, s = n & 3;
    int e, o;
    const A* h;
    if ((m2 & 1) == 0) {
        h = (s & 1) ? hb : ha;
        e = s >> 1;
        o = highpass ? (s ^ 1) : s;
    } else {
        h = (s & 1) ? hb : ha;
        e = (s >> 1) ^ 1;
        o = highpass ? ((s & 1) ? 1 : 2) : ((s & 1) ? 2 : 1);
    }
    A acc = 0;
    const A* p = x + (size_t)(o - m2 + 2 * q - base) * st;
    for (int t = 0; t < m2; ++t) acc += h[e + 2 * t] * p[(size_t)(2 * t) * st];
    return acc;
}

template <typename T>
WL_DEV void wl_dtcwt_inv2_body(const WlDtInv2Args<T>& a, const WlCtx& ctx) {
    typedef typename WlAcc<T>::type A;
    const int tx = ctx.tid & 63, ty = ctx.tid >> 6, ny = ctx.nthreads >> 6;
    const int tiles = a.tiles_x * a.tiles_y;
    const int64_t plane = ctx.bid / tiles;
    const int tile = (int)(ctx.bid - plane * tiles);
    const int R0 = (tile / a.tiles_x) * a.TH, C0 = (tile % a.tiles_x) * a.TW;   // output origin (multiples of 4)
    const int L = a.L, m2 = L / 2;
    const int nr = a.TH / 2 + L + 2, nc = a.TW / 2 + L + 2;
    A* taps = reinterpret_cast<A*>(ctx.smem);
    A* B = taps + 4 * L;
    A* U = B + (size_t)4 * nr * nc;
    for (int i = ctx.tid; i < L; i += ctx.nthreads) {
        taps[i] = a.g0a[i]; taps[L + i] = a.g0b[i]; taps[2 * L + i] = a.g1a[i]; taps[3 * L + i] = a.g1b[i];
    }
    const A* g0a = taps, *g0b = taps + L, *g1a = taps + 2 * L, *g1b = taps + 3 * L;
    // staged input origin: q0 = R0/4 -> X[o - m2 + 2*q0 ...], smallest o is 0
    const int rb = R0 / 2 - m2, cb = C0 / 2 - m2;
    const int w2 = a.w / 2;
    const size_t qplane = (size_t)(a.h / 2) * w2;
    const T* llp = a.ll ? a.ll + (size_t)plane * a.ll_plane_stride : nullptr;
    const T* hp = a.highs ? a.highs + (size_t)plane * 12 * qplane : nullptr;
    const size_t bs = (size_t)nr * nc;
    for (int i = ty; i < nr; i += ny) {
        const int r = wl_ext(rb + i, a.h, WL_EXT_SYM);
        for (int j = tx; j < nc; j += 64) {
            const int c = wl_ext(cb + j, a.w, WL_EXT_SYM);
            A v0 = 0, v1 = 0, v2 = 0, v3 = 0;
            if (llp) v0 = (A)llp[(size_t)r * a.ll_row_stride + c];
            if (hp) {
                v1 = wl_c2q_at<T, A>(hp, qplane, w2, 0, r, c);
                v2 = wl_c2q_at<T, A>(hp, qplane, w2, 1, r, c);
                v3 = wl_c2q_at<T, A>(hp, qplane, w2, 2, r, c);
            }
            const size_t o = (size_t)i * nc + j;
            B[o] = v0; B[bs + o] = v1; B[2 * bs + o] = v2; B[3 * bs + o] = v3;
        }
    }
    ctx.sync();
    // column interpolation: lo = ci(lh; g1b,g1a,hp) + ci(ll; g0b,g0a,lp);  hi = ci(hh; g1b,g1a,hp) + ci(hl; g0b,g0a,lp)
    for (int i = ty; i < a.TH; i += ny) {
        const int n = R0 + i;
        for (int j = tx; j < nc; j += 64) {
            const A lo = wl_ifilt_at<A>(B + bs + j, nc, rb, n, g1b, g1a, m2, true) +
                         wl_ifilt_at<A>(B + j, nc, rb, n, g0b, g0a, m2, false);
            const A hi = wl_ifilt_at<A>(B + 3 * bs + j, nc, rb, n, g1b, g1a, m2, true) +
                         wl_ifilt_at<A>(B + 2 * bs + j, nc, rb, n, g0b, g0a, m2, false);
            U[(size_t)i * nc + j] = lo;
            U[(size_t)(a.TH + i) * nc + j] = hi;
        }
    }
    ctx.sync();
    const int OH = 2 * a.h, OW = 2 * a.w;
    T* yp = a.y + (size_t)plane * OH * OW;
    for (int i = ty; i < a.TH; i += ny) {
        if (R0 + i >= OH) break;
        for (int j = tx; j < a.TW; j += 64) {
            if (C0 + j >= OW) break;
            const A v = wl_ifilt_at<A>(U + (size_t)(a.TH + i) * nc, 1, cb, C0 + j, g1b, g1a, m2, true) +
                        wl_ifilt_at<A>(U + (size_t)i * nc, 1, cb, C0 + j, g0b, g0a, m2, false);
            yp[(size_t)(R0 + i) * OW + (C0 + j)] = (T)v;
        }
    }
}
