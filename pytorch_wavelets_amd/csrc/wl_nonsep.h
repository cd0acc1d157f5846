// Non-separable one-level 2-D analysis / synthesis with four arbitrary Ly x Lx point-spread functions: the reference's
// afb2d_nonsep (dwt/lowlevel.py:524-597) and sfb2d_nonsep (:746-798).  One thread per output sample, O(Ly*Lx) taps
// each - the direct evaluation of the reference's strided conv2d / conv_transpose2d.  Not a tuned path: the separable
// kernels are the product path, this closes the reference's function-level API.
//
//   analysis : y[b][i][j] = sum_{u<Ly, v<Lx} f[b][u][v] * X(2i + oy + u, 2j + ox + v)        (cross-correlation: the
//              stored filters are already mirrored by prep_filt_afb2d_nonsep), X = x under the per-axis extension;
//              zero / symmetric / reflect: o = -(p // 2) = -(L - 2) (:575-591), K = (N + L - 1) // 2 outputs;
//              periodization: xe = x with its last sample repeated when N is odd, o = ceil(L / 2) - (L - 1) on the
//              periodic xe, K = Ne / 2 (:558-570: roll, zero-padded conv, one fold - the same numbers whenever L - 1 <= Ne)
//   synthesis: y[p][q] = sum_b sum_{u,v} g[b][u][v] * c[b][(p + sy - u) / 2][(q + sx - v) / 2]  over the even numerators;
//              zero / symmetric / reflect / periodic: s = L - 2, indices outside the coefficient plane dropped (:793-796);
//              periodization: s = L / 2 - 1 and the numerator taken modulo 2K (:787-791: fold + roll).
//   gradients (what autograd gives upstream, where both banks are plain differentiable ATen chains):
//     analysis:  dx[r][c] = sum over the extended positions (er, ec) that the boundary rule maps onto (r, c) of
//                sum_b sum_{u,v} f[b][u][v] * dy[b][(er - oy - u) / 2][(ec - ox - v) / 2]   (even numerators, in range)
//                - the position itself plus the mirrored / wrapped / repeated ones in the two pads (WlAfbNonsepAdj);
//     synthesis: dc[b][i][j] = sum_{u,v} g[b][u][v] * DY(2i + u - sy, 2j + v - sx), DY = dy outside its support zero
//                (periodization: periodic) - which IS the analysis kernel with other offsets (WlAfbNonsep).
#pragma once
#include "wl_common.h"

template <typename T>
struct WlNonsepArgs {
    typedef typename WlAcc<T>::type A;
    const T* x;      // analysis: (planes, H, W); synthesis: coefficients (planes, 4, Kh, Kw)
    T* y;            // analysis: (planes, 4, Kh, Kw); synthesis: (planes, OH, OW)
    const A* f;      // (4, Ly, Lx)
    int64_t planes;
    int H, W, Kh, Kw, OH, OW, Ly, Lx, oy, ox, ext, per;
};

template <typename T>
struct WlAfbNonsep {
    typedef WlNonsepArgs<T> Args;
    typedef typename WlAcc<T>::type A;
    static const int kThreads = 256;
    static const int kMinWaves = 1;
    static WL_DEV void run(const Args& a, const WlCtx& ctx) {
        const int64_t per = (int64_t)a.Kh * a.Kw;
        const int64_t idx = ctx.bid * kThreads + ctx.tid;
        if (idx >= a.planes * per) return;
        const int64_t plane = idx / per;
        const int rem = (int)(idx - plane * per);
        const int i = rem / a.Kw, j = rem - i * a.Kw;
        const T* xp = a.x + (size_t)plane * a.H * a.W;
        const int taps = a.Ly * a.Lx;
        A acc[4] = {0, 0, 0, 0};
        for (int u = 0; u < a.Ly; ++u) {
            const int r = wl_ext(2 * i + a.oy + u, a.H, a.ext);
            if (r < 0) continue;
            for (int v = 0; v < a.Lx; ++v) {
                const int c = wl_ext(2 * j + a.ox + v, a.W, a.ext);
                if (c < 0) continue;
                const A s = (A)xp[(size_t)r * a.W + c];
                const A* fp = a.f + u * a.Lx + v;
                acc[0] += fp[0] * s; acc[1] += fp[taps] * s; acc[2] += fp[2 * taps] * s; acc[3] += fp[3 * taps] * s;
            }
        }
        T* yp = a.y + (size_t)plane * 4 * per + rem;
        yp[0] = (T)acc[0]; yp[per] = (T)acc[1]; yp[2 * per] = (T)acc[2]; yp[3 * per] = (T)acc[3];
    }
};

template <typename T>
struct WlSfbNonsep {
    typedef WlNonsepArgs<T> Args;
    typedef typename WlAcc<T>::type A;
    static const int kThreads = 256;
    static const int kMinWaves = 1;
    static WL_DEV void run(const Args& a, const WlCtx& ctx) {
        const int64_t per = (int64_t)a.OH * a.OW;
        const int64_t idx = ctx.bid * kThreads + ctx.tid;
        if (idx >= a.planes * per) return;
        const int64_t plane = idx / per;
        const int rem = (int)(idx - plane * per);
        const int p = rem / a.OW, q = rem - p * a.OW;
        const size_t cper = (size_t)a.Kh * a.Kw;
        const T* cp = a.x + (size_t)plane * 4 * cper;
        const int taps = a.Ly * a.Lx;
        A acc = 0;
        for (int u = 0; u < a.Ly; ++u) {
            int P = p + a.oy - u;
            if (a.per) P = wl_pmod(P, 2 * a.Kh);
            if (P < 0 || (P & 1) || P / 2 >= a.Kh) continue;
            for (int v = 0; v < a.Lx; ++v) {
                int Q = q + a.ox - v;
                if (a.per) Q = wl_pmod(Q, 2 * a.Kw);
                if (Q < 0 || (Q & 1) || Q / 2 >= a.Kw) continue;
                const T* s = cp + (size_t)(P / 2) * a.Kw + Q / 2;
                const A* gp = a.f + u * a.Lx + v;
                acc += gp[0] * (A)s[0] + gp[taps] * (A)s[cper] + gp[2 * taps] * (A)s[2 * cper] + gp[3 * taps] * (A)s[3 * cper];
            }
        }
        a.y[(size_t)plane * per + rem] = (T)acc;
    }
};

// Adjoint of WlAfbNonsep: x = dy (planes, 4, Kh, Kw), y = dx (planes, H, W); one thread per input sample.
template <typename T>
struct WlAfbNonsepAdj {
    typedef WlNonsepArgs<T> Args;
    typedef typename WlAcc<T>::type A;
    static const int kThreads = 256;
    static const int kMinWaves = 1;
    // contribution of extended position (er, ec)
    static WL_DEV A at(const Args& a, const T* dp, int er, int ec) {
        const size_t per = (size_t)a.Kh * a.Kw;
        const int taps = a.Ly * a.Lx;
        A acc = 0;
        for (int u = 0; u < a.Ly; ++u) {
            const int P = er - a.oy - u;
            if (P < 0 || (P & 1) || P / 2 >= a.Kh) continue;
            for (int v = 0; v < a.Lx; ++v) {
                const int Q = ec - a.ox - v;
                if (Q < 0 || (Q & 1) || Q / 2 >= a.Kw) continue;
                const T* s = dp + (size_t)(P / 2) * a.Kw + Q / 2;
                const A* fp = a.f + u * a.Lx + v;
                acc += fp[0] * (A)s[0] + fp[taps] * (A)s[per] + fp[2 * taps] * (A)s[2 * per] + fp[3 * taps] * (A)s[3 * per];
            }
        }
        return acc;
    }
    // k-th extended position mapped onto r: k = 0 is r itself, then the positions of the two pads that the rule folds
    // onto r, in order; -1000000 when there are no more.  lo / hi = first / last extended position the bank reads.
    static WL_DEV int cand(int k, int r, int n, int lo, int hi, int ext) {
        if (k == 0) return r;
        for (int e = lo; e <= hi; ++e) {
            if (e == 0) e = n;                       // skip the interior
            if (e > hi) break;
            if (wl_ext(e, n, ext) == r && --k == 0) return e;
        }
        return -1000000;
    }
    static WL_DEV void run(const Args& a, const WlCtx& ctx) {
        const int64_t per = (int64_t)a.H * a.W;
        const int64_t idx = ctx.bid * kThreads + ctx.tid;
        if (idx >= a.planes * per) return;
        const int64_t plane = idx / per;
        const int rem = (int)(idx - plane * per);
        const int r = rem / a.W, c = rem - r * a.W;
        const T* dp = a.x + (size_t)plane * 4 * a.Kh * a.Kw;
        const int rlo = a.oy < 0 ? a.oy : 0, rhi = 2 * (a.Kh - 1) + a.oy + a.Ly - 1;
        const int clo = a.ox < 0 ? a.ox : 0, chi = 2 * (a.Kw - 1) + a.ox + a.Lx - 1;
        A acc = 0;
        for (int kr = 0;; ++kr) {
            const int er = cand(kr, r, a.H, rlo, rhi, a.ext);
            if (er == -1000000) break;
            for (int kc = 0;; ++kc) {
                const int ec = cand(kc, c, a.W, clo, chi, a.ext);
                if (ec == -1000000) break;
                acc += at(a, dp, er, ec);
            }
        }
        a.y[idx] = (T)acc;
    }
};
