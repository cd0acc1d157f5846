// Direct (one thread per output sample, O(L^2) taps) analysis / synthesis for ONE degenerate corner of the reference:
// periodization when a level is shorter than the filter.  There the reference's afb1d / sfb1d fold the wrapped
// tail only ONCE (dwt/lowlevel.py:146-150 `x[:L2] = x[:L2] + x[N2:N2+L2]`, :256-260 `y[:L-2] = y[:L-2] + y[N:N+L-2]`),
// which is not a circular convolution - so the tile kernels (circular index math) do not apply.  These kernels
// evaluate the reference's formulas literally; the planes concerned have fewer than L samples per axis, so cost is
// irrelevant.  (They are valid for every periodization size - for long signals both forms coincide.)
//
//   analysis, one axis:  xe = x (+ its last sample again when N is odd), Ne = len(xe), N2 = Ne/2, L2 = L/2,
//       Z(i) = xe[(i + S) mod Ne] for 0 <= i < Ne, else 0           (the rolled signal, zero padded by conv2d)
//       S = L2 mod Ne while L2 < 2 Ne, else 0: the reference's roll() (dwt/lowlevel.py:9-25) is built from two slices and
//       degenerates into the identity once the shift reaches twice the length (signals of 1-4 samples under 10-20 taps)
//       y[k] = sum_m h[L-1-m] * ( Z(2k-m) + [k < min(L2,N2)] * Z(2k+Ne-m) )
//   synthesis, one axis: full[n] = sum_k lo[k] g0[n-2k] + hi[k] g1[n-2k], n in [0, 2K+L-2), N = 2K,
//       z[n] = full[n] + [n < L-2] * full[n+N]   (n < N),   y[i] = z[(i + S) mod N],  S = L/2-1 (0 once L/2-1 >= 2N: roll() again)
#pragma once
#include "wl_common.h"

template <typename T>
struct WlDirectArgs {
    typedef typename WlAcc<T>::type A;
    const T* x;        // analysis: (NC,H,W) input through x_ps/x_rs; synthesis: ll (NC,Kh,Kw) through x_ps/x_rs
    const T* highs;    // synthesis: (NC,3,Kh,Kw) or nullptr
    T* ll;             // analysis: (NC,Kh,Kw) through ll_ps/ll_rs; synthesis: y (NC,OH,OW) dense
    T* hout;           // analysis: (NC,3,Kh,Kw)
    const A* w_lo; const A* w_hi; const A* h_lo; const A* h_hi;   // taps along W / along H
    int64_t NC, x_ps, ll_ps;
    int x_rs, ll_rs;
    int H, W, Kh, Kw, Lw, Lh, OH, OW;
};

template <typename T>
struct WlAfbDirect {
    typedef WlDirectArgs<T> Args;
    typedef typename WlAcc<T>::type A;
    static const int kThreads = 256;
    static const int kMinWaves = 1;
    static WL_DEV int zsrc(int i, int n, int L2) { return wl_per_rolled_src(i, n, L2); }   // Z(i) of the header
    static WL_DEV void run(const Args& a, const WlCtx& ctx) {
        const int64_t per = (int64_t)a.Kh * a.Kw;
        const int64_t idx = ctx.bid * kThreads + ctx.tid;
        if (idx >= a.NC * per) return;
        const int64_t plane = idx / per;
        const int rem = (int)(idx - plane * per);
        const int kh = rem / a.Kw, kw = rem - kh * a.Kw;
        const int neh = a.H + (a.H & 1), new_ = a.W + (a.W & 1);
        const int l2h = a.Lh / 2, l2w = a.Lw / 2;
        const bool fold_h = kh < (l2h < neh / 2 ? l2h : neh / 2), fold_w = kw < (l2w < new_ / 2 ? l2w : new_ / 2);
        const T* xp = a.x + (size_t)plane * a.x_ps;
        A acc[4] = {0, 0, 0, 0};   // (H-lo,W-lo) (H-hi,W-lo) (H-lo,W-hi) (H-hi,W-hi)
        for (int mh = 0; mh < a.Lh; ++mh) {
            const A hl = a.h_lo[a.Lh - 1 - mh], hh = a.h_hi[a.Lh - 1 - mh];
            for (int fh = 0; fh < (fold_h ? 2 : 1); ++fh) {
                const int r = zsrc(2 * kh - mh + fh * neh, a.H, l2h);
                if (r < 0) continue;
                A rl = 0, rh = 0;   // row-filtered (W-lo, W-hi) sample of source row r at output column kw
                for (int mw = 0; mw < a.Lw; ++mw) {
                    A s = 0;
                    for (int fw = 0; fw < (fold_w ? 2 : 1); ++fw) {
                        const int c = zsrc(2 * kw - mw + fw * new_, a.W, l2w);
                        if (c >= 0) s += (A)xp[(size_t)r * a.x_rs + c];
                    }
                    rl += a.w_lo[a.Lw - 1 - mw] * s;
                    rh += a.w_hi[a.Lw - 1 - mw] * s;
                }
                acc[0] += hl * rl; acc[1] += hh * rl; acc[2] += hl * rh; acc[3] += hh * rh;
            }
        }
        a.ll[(size_t)plane * a.ll_ps + (size_t)kh * a.ll_rs + kw] = (T)acc[0];
        T* hp = a.hout + (size_t)plane * 3 * per + rem;
        hp[0] = (T)acc[1]; hp[per] = (T)acc[2]; hp[2 * per] = (T)acc[3];
    }
};

template <typename T>
struct WlSfbDirect {
    typedef WlDirectArgs<T> Args;
    typedef typename WlAcc<T>::type A;
    static const int kThreads = 256;
    static const int kMinWaves = 1;
    // coefficient of coefficient k in output sample i of one synthesis axis (taps g, length L, K coefficients)
    static WL_DEV A coef(const A* g, int L, int K, int i, int k) {
        const int N = 2 * K;
        const int n = (i + (L / 2 - 1 < 2 * N ? L / 2 - 1 : 0)) % N;
        A c = 0;
        int t = n - 2 * k;
        if (t >= 0 && t < L) c += g[t];
        if (n < L - 2) { t = n + N - 2 * k; if (t >= 0 && t < L) c += g[t]; }
        return c;
    }
    static WL_DEV void run(const Args& a, const WlCtx& ctx) {
        const int64_t per = (int64_t)a.OH * a.OW;
        const int64_t idx = ctx.bid * kThreads + ctx.tid;
        if (idx >= a.NC * per) return;
        const int64_t plane = idx / per;
        const int rem = (int)(idx - plane * per);
        const int i = rem / a.OW, j = rem - i * a.OW;
        const T* lp = a.x + (size_t)plane * a.x_ps;
        const int64_t bp = (int64_t)a.Kh * a.Kw;
        const T* hp = a.highs ? a.highs + (size_t)plane * 3 * bp : nullptr;
        A y = 0;
        for (int kh = 0; kh < a.Kh; ++kh) {
            const A al = coef(a.h_lo, a.Lh, a.Kh, i, kh), ah = coef(a.h_hi, a.Lh, a.Kh, i, kh);
            if (al == 0 && ah == 0) continue;
            for (int kw = 0; kw < a.Kw; ++kw) {
                const A bl = coef(a.w_lo, a.Lw, a.Kw, j, kw), bh = coef(a.w_hi, a.Lw, a.Kw, j, kw);
                const A ll = (A)lp[(size_t)kh * a.x_rs + kw];
                A lh = 0, hl = 0, hh = 0;
                if (hp) { const size_t o = (size_t)kh * a.Kw + kw; lh = (A)hp[o]; hl = (A)hp[bp + o]; hh = (A)hp[2 * bp + o]; }
                // along H: lo = (ll, lh), hi = (hl, hh); along W: (lo, hi)   (SFB2D.forward, dwt/lowlevel.py:671-680)
                y += bl * (al * ll + ah * lh) + bh * (al * hl + ah * hh);
            }
        }
        a.ll[(size_t)plane * per + rem] = (T)y;
    }
};
