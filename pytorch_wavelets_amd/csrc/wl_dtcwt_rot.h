// Level 1 of the rotationally symmetric DTCWT ('near_sym_b_bp': a third, band-pass filter h2 for the diagonal sub-band) in
// ONE launch: fwd_j1_rot of the reference (dtcwt/transform_funcs.py:124-149 = three rowfilter + four colfilter
// (dtcwt/lowlevel.py:70-94) + three q2c (:243-260) + two stack), and - SCAT = 1 - ScatLayerj1_rot_f.forward on top of it
// (scatternet/lowlevel.py:140-182: 2x2 average of the lowpass, smoothed magnitudes of the six orientations):
//     lo = R_h0 x, hi = R_h1 x, ba = R_h2 x                      (R: along W, C: along H; Y[i] = sum_j h[j] X(i + j - L/2))
//     ll = C_h0 lo, lh = C_h1 lo, hl = C_h0 hi, hh = C_h2 ba
//     (15, 165 deg) = q2c(lh), (45, 135) = q2c(hh), (75, 105) = q2c(hl);  q2c of a quad (a b / c d) / sqrt 2: (a - d, b + c), (a + d, b - c)
// On the single-axis kernels (the round-2 path) this is seven launches plus ~25 tensor-library kernels for q2c / stack and the
// magnitudes: 1.5 ms for 64 x 3 x 256 x 256, 43 times the time per pixel of the plain ScatLayer.
// A workgroup owns a TH x 64 pixel tile: the symmetrically extended (or zero padded) input tile goes to LDS once, the three
// row-filtered planes of its TH + 2 m rows to LDS, and a thread owns 2 x 2 quads: the column filters of its four pixels
// share their 20 rows, q2c happens in registers, the outputs leave in the layouts the reference returns ((N, C, H, W) lowpass,
// (N, 6, C, H/2, W/2) real and imaginary parts; SCAT: (N, 7, C, H/2, W/2)).  The taps are centred and zero padded to LM = 19
// (the longest odd filter of the tables) so that every loop is unrolled and the taps sit in scalar registers.
#pragma once
#include "wl_common.h"
#include "wl_dtcwt_kernels.h"   // wl_sqrt

template <typename T>
struct WlDtRotArgs {
    typedef typename WlAcc<T>::type A;
    const T* x;                    // (planes, H, W) dense
    T* ll;                         // (planes, H, W); SCAT: unused
    T* re; T* im;                  // (N, 6, C, H/2, W/2); SCAT: re = Z (N, 7, C, H/2, W/2), im unused
    const A* h0; const A* h1; const A* h2;
    int64_t planes, nblocks;
    int C, H, W, L0, L1, L2, ext;  // ext: WL_EXT_SYM or WL_EXT_ZERO
    int TH, tiles_x, tiles_y;
    int in_pitch, mid_off, lds_bytes;
    double bias;
};

template <typename T, int SCAT>
struct WlDtFwd1Rot {
    typedef WlDtRotArgs<T> Args;
    typedef typename WlAcc<T>::type A;
    static const int kThreads = 256;
    static const int kMinWaves = 2;
    static const int TW = 64, LM = 19, M = LM / 2;
    struct A2 { A x, y; };
    // tap s of the centred, zero-padded form of an L-tap filter (wave-uniform: scalar loads)
    static WL_DEV A padded(const A* h, int L, int s) {
        const int t = s - (M - L / 2);
        const A v = h[t < 0 ? 0 : (t >= L ? L - 1 : t)];
        return (t >= 0 && t < L) ? v : (A)0;
    }
    static WL_DEV void run(const Args& a, const WlCtx& ctx) {
        const int tid = ctx.tid;
        const int per_plane = a.tiles_x * a.tiles_y;
        const int64_t plane = ctx.bid / per_plane;
        const int rem = (int)(ctx.bid - plane * per_plane);
        const int tyi = rem / a.tiles_x, txi = rem - tyi * a.tiles_x;
        const int r0 = tyi * a.TH, c0 = txi * TW;
        const int th = r0 + a.TH <= a.H ? a.TH : a.H - r0;          // pixel rows / columns of this tile (even)
        const int tw = c0 + TW <= a.W ? TW : a.W - c0;
        const int rows = th + 2 * M, cols = tw + 2 * M;
        A* const in = reinterpret_cast<A*>(ctx.smem);
        A* const mid = reinterpret_cast<A*>(ctx.smem + a.mid_off);  // three planes of rows x 64: lo, hi, ba
        const int mplane = (a.TH + 2 * M) * TW;
        const T* const xp = a.x + (size_t)plane * a.H * a.W;
        A p0[LM], p1[LM], p2[LM];
#pragma unroll
        for (int s = 0; s < LM; ++s) { p0[s] = padded(a.h0, a.L0, s); p1[s] = padded(a.h1, a.L1, s); p2[s] = padded(a.h2, a.L2, s); }
        // ---- the extended input tile: pixel rows r0 - M .., columns c0 - M .. (eight loads in flight per thread)
        {
            const int tx = tid & 63, ty = tid >> 6;
            for (int c = tx; c < cols; c += 64) {
                const int gc = wl_ext(c0 - M + c, a.W, a.ext);
                for (int rb = ty; rb < rows; rb += 32) {
                    T v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int r = rb + 4 * u;
                        const int gr = r < rows ? wl_ext(r0 - M + r, a.H, a.ext) : -1;
                        v[u] = xp[(gr < 0 || gc < 0) ? 0 : (size_t)gr * a.W + gc];
                        if (gr < 0 || gc < 0) v[u] = (T)0;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int r = rb + 4 * u;
                        if (r < rows) in[r * a.in_pitch + c] = (A)v[u];
                    }
                }
            }
        }
        ctx.sync();
        // ---- the three row filters of every staged row at the tile's columns
        {
            const int tx = tid & 63, ty = tid >> 6;
            if (tx < tw) {
                for (int r = ty; r < rows; r += 4) {
                    const A* p = in + r * a.in_pitch + tx;
                    A lo = 0, hi = 0, ba = 0;
#pragma unroll
                    for (int s = 0; s < LM; ++s) { const A v = p[s]; lo += p0[s] * v; hi += p1[s] * v; ba += p2[s] * v; }
                    mid[r * TW + tx] = lo; mid[mplane + r * TW + tx] = hi; mid[2 * mplane + r * TW + tx] = ba;
                }
            }
        }
        ctx.sync();
        // ---- column filters of 2 x 2 quads, q2c, stores
        const int qx = tid & 31, qy0 = tid >> 5;
        const int h2 = a.H / 2, w2 = a.W / 2;
        const size_t qper = (size_t)h2 * w2;
        const int64_t n = plane / a.C;
        const int ch = (int)(plane - n * a.C);
        if (2 * qx < tw) {
            for (int qy = qy0; 2 * qy < th; qy += 8) {
                A2 ll0 = {0, 0}, ll1 = {0, 0}, lh0 = {0, 0}, lh1 = {0, 0}, hl0 = {0, 0}, hl1 = {0, 0}, hh0 = {0, 0}, hh1 = {0, 0};
                const A* q = mid + (2 * qy) * TW + 2 * qx;        // staged row of pixel row 2 qy - M
#pragma unroll
                for (int s = 0; s <= LM; ++s) {
                    const A2 lo = *reinterpret_cast<const A2*>(q + s * TW);
                    const A2 hi = *reinterpret_cast<const A2*>(q + mplane + s * TW);
                    const A2 ba = *reinterpret_cast<const A2*>(q + 2 * mplane + s * TW);
                    if (s < LM) {
                        ll0.x += p0[s] * lo.x; ll0.y += p0[s] * lo.y; lh0.x += p1[s] * lo.x; lh0.y += p1[s] * lo.y;
                        hl0.x += p0[s] * hi.x; hl0.y += p0[s] * hi.y; hh0.x += p2[s] * ba.x; hh0.y += p2[s] * ba.y;
                    }
                    if (s >= 1) {
                        ll1.x += p0[s - 1] * lo.x; ll1.y += p0[s - 1] * lo.y; lh1.x += p1[s - 1] * lo.x; lh1.y += p1[s - 1] * lo.y;
                        hl1.x += p0[s - 1] * hi.x; hl1.y += p0[s - 1] * hi.y; hh1.x += p2[s - 1] * ba.x; hh1.y += p2[s - 1] * ba.y;
                    }
                }
                const int gq = (r0 / 2 + qy), gp = (c0 / 2 + qx);
                const A k = (A)0.70710678118654752440;
                // quad (a b / c d) = (X0.x X0.y / X1.x X1.y): z1 = (a - d, b + c), z2 = (a + d, b - c), both / sqrt 2
                const A o_re[6] = {(lh0.x - lh1.y) * k, (hh0.x - hh1.y) * k, (hl0.x - hl1.y) * k,
                                   (hl0.x + hl1.y) * k, (hh0.x + hh1.y) * k, (lh0.x + lh1.y) * k};
                const A o_im[6] = {(lh0.y + lh1.x) * k, (hh0.y + hh1.x) * k, (hl0.y + hl1.x) * k,
                                   (hl0.y - hl1.x) * k, (hh0.y - hh1.x) * k, (lh0.y - lh1.x) * k};
                const size_t qoff = (size_t)gq * w2 + gp;
                if (SCAT) {
                    T* z = a.re + ((size_t)n * 7 * a.C + ch) * qper + qoff;
                    const size_t os = (size_t)a.C * qper;
                    z[0] = (T)((ll0.x + ll0.y + ll1.x + ll1.y) * (A)0.25);
                    const A b = (A)a.bias;
#pragma unroll
                    for (int o = 0; o < 6; ++o) {
                        const A e = o_re[o] * o_re[o] + o_im[o] * o_im[o] + b * b;
                        z[(size_t)(o + 1) * os] = (T)(wl_sqrt(e) - b);
                    }
                } else {
                    T* lp = a.ll + (size_t)plane * a.H * a.W + (size_t)(r0 + 2 * qy) * a.W + c0 + 2 * qx;
                    lp[0] = (T)ll0.x; lp[1] = (T)ll0.y; lp[a.W] = (T)ll1.x; lp[a.W + 1] = (T)ll1.y;
                    const size_t os = (size_t)a.C * qper;
                    T* pr = a.re + ((size_t)n * 6 * a.C + ch) * qper + qoff;
                    T* pi = a.im + ((size_t)n * 6 * a.C + ch) * qper + qoff;
#pragma unroll
                    for (int o = 0; o < 6; ++o) { pr[(size_t)o * os] = (T)o_re[o]; pi[(size_t)o * os] = (T)o_im[o]; }
                }
            }
        }
    }
};
