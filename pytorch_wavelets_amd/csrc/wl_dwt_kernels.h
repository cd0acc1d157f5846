// Generic (any tap count, any boundary mode, any size) single-level 2-D DWT analysis and
// synthesis workgroup kernels.  One 256-thread workgroup produces one TH x TW output tile of one
// (n,c) plane: boundary extension is index math while the input tile is staged into LDS, the row
// bank runs LDS->LDS, the column bank runs LDS->registers and the four sub-bands are written
// straight into the final `yl` / `yh[j]` (N,C,3,H',W') buffers.
//
// Restates (fused): afb1d x2 + reshape/contiguous  = AFB2D.forward  (reference dwt/lowlevel.py:91-172, :336-347)
//                   sfb1d x3                       = SFB2D.forward  (reference dwt/lowlevel.py:226-271, :671-680)
// The specialised streaming kernels in wl_dwt_stream.h cover the hot configurations; these
// kernels are the general path (and the reference the streaming kernels are tested against).
#pragma once
#include "wl_common.h"

// ---------------------------------------------------------------------------------------------
// analysis
//   y_b[k] = sum_j h_b[j] * ext(x, 2k + base + j),  base = off - (L-1), h = stored (reversed) taps
// ---------------------------------------------------------------------------------------------
template <typename T>
struct WlAfb2dArgs {
    typedef typename WlAcc<T>::type A;
    const T* x;      // (NC, H, W)
    T* ll;           // (NC, Kh, Kw)
    T* highs;        // (NC, 3, Kh, Kw)
    const A* h_w_lo; // taps along W (device pointers), Lw each
    const A* h_w_hi;
    const A* h_h_lo; // taps along H, Lh each
    const A* h_h_hi;
    int64_t NC;
    int64_t x_ps, ll_ps;   // plane strides of x and ll (elements)
    int x_rs, ll_rs;       // row strides of x and ll (elements; rows are unit-stride)
    int H, W, Kh, Kw;
    int Lw, Lh, basew, baseh, extw, exth;
    int TH, TW, tiles_x, tiles_y;
    int nrows, ncols, spitch;   // staged tile geometry
};

template <typename T>
WL_HD size_t wl_afb2d_tile_lds(const WlAfb2dArgs<T>& a) {
    typedef typename WlAcc<T>::type A;
    size_t taps = (size_t)wl_align_up(2 * a.Lw + 2 * a.Lh, 4);
    return sizeof(A) * (taps + (size_t)a.nrows * a.spitch + (size_t)a.nrows * a.TW * 2);
}

template <typename T>
WL_DEV void wl_afb2d_tile_body(const WlAfb2dArgs<T>& a, const WlCtx& ctx) {
    typedef typename WlAcc<T>::type A;
    const int tx = ctx.tid & 63, ty = ctx.tid >> 6, ny = ctx.nthreads >> 6;
    const int tiles = a.tiles_x * a.tiles_y;
    const int64_t plane = ctx.bid / tiles;
    const int tile = (int)(ctx.bid - plane * tiles);
    const int kh0 = (tile / a.tiles_x) * a.TH, kw0 = (tile % a.tiles_x) * a.TW;

    A* taps = reinterpret_cast<A*>(ctx.smem);
    A* S = taps + wl_align_up(2 * a.Lw + 2 * a.Lh, 4);
    A* Tm = S + (size_t)a.nrows * a.spitch;
    A* hwl = taps, *hwh = taps + a.Lw, *hhl = taps + 2 * a.Lw, *hhh = taps + 2 * a.Lw + a.Lh;
    for (int i = ctx.tid; i < a.Lw; i += ctx.nthreads) { hwl[i] = a.h_w_lo[i]; hwh[i] = a.h_w_hi[i]; }
    for (int i = ctx.tid; i < a.Lh; i += ctx.nthreads) { hhl[i] = a.h_h_lo[i]; hhh[i] = a.h_h_hi[i]; }

    // phase 1: stage the boundary-extended input tile
    const T* xp = a.x + (size_t)plane * a.x_ps;
    const int er0 = 2 * kh0 + a.baseh, ec0 = 2 * kw0 + a.basew;
    for (int i = ty; i < a.nrows; i += ny) {
        const int r = wl_ext(er0 + i, a.H, a.exth);
        const T* xr = xp + (size_t)(r < 0 ? 0 : r) * a.x_rs;
        for (int j = tx; j < a.ncols; j += 64) {
            const int c = wl_ext(ec0 + j, a.W, a.extw);
            S[i * a.spitch + j] = (r < 0 || c < 0) ? (A)0 : (A)xr[c];
        }
    }
    ctx.sync();
    // phase 2: row bank (along W), both bands, decimate by 2
    for (int i = ty; i < a.nrows; i += ny) {
        const A* s = S + i * a.spitch;
        for (int k = tx; k < a.TW; k += 64) {
            A lo = 0, hi = 0;
            for (int j = 0; j < a.Lw; ++j) {
                const A v = s[2 * k + j];
                lo += hwl[j] * v;
                hi += hwh[j] * v;
            }
            Tm[(i * a.TW + k) * 2] = lo;
            Tm[(i * a.TW + k) * 2 + 1] = hi;
        }
    }
    ctx.sync();
    // phase 3: column bank (along H) and sub-band scatter
    const size_t bplane = (size_t)a.Kh * a.Kw;
    T* llp = a.ll + (size_t)plane * a.ll_ps;
    T* hp = a.highs + (size_t)plane * 3 * bplane;
    for (int kh = ty; kh < a.TH; kh += ny) {
        if (kh0 + kh >= a.Kh) break;
        for (int kw = tx; kw < a.TW; kw += 64) {
            if (kw0 + kw >= a.Kw) break;
            A ll = 0, lh = 0, hl = 0, hh = 0;
            for (int j = 0; j < a.Lh; ++j) {
                const A lo = Tm[((2 * kh + j) * a.TW + kw) * 2];
                const A hi = Tm[((2 * kh + j) * a.TW + kw) * 2 + 1];
                ll += hhl[j] * lo;
                lh += hhh[j] * lo;
                hl += hhl[j] * hi;
                hh += hhh[j] * hi;
            }
            const size_t o = (size_t)(kh0 + kh) * a.Kw + (kw0 + kw);
            llp[(size_t)(kh0 + kh) * a.ll_rs + (kw0 + kw)] = (T)ll;
            hp[o] = (T)lh;
            hp[bplane + o] = (T)hl;
            hp[2 * bplane + o] = (T)hh;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// synthesis
//   y[n] = sum_{k'} B(k') * g[n + s - 2k'],  B = band (zero outside [0,K), or circular for
//   periodization);  s = L-2 (non-periodization), L/2-1 (periodization)
// ---------------------------------------------------------------------------------------------
template <typename T>
struct WlSfb2dArgs {
    typedef typename WlAcc<T>::type A;
    const T* ll;      // (NC, >=Kh, >=Kw) with explicit strides (may be a cropped view)
    const T* highs;   // (NC, 3, Kh, Kw) contiguous, or nullptr (= zeros)
    T* y;             // (NC, OH, OW)
    const A* g_w_lo;  // taps along W
    const A* g_w_hi;
    const A* g_h_lo;  // taps along H
    const A* g_h_hi;
    int64_t NC;
    int64_t ll_plane_stride;
    int ll_row_stride;
    int Kh, Kw, OH, OW;
    int Lw, Lh, sw, sh, circ;   // circ=1: periodization (circular band index)
    int TH, TW, tiles_x, tiles_y;
    int nkr, nkc;               // staged band tile geometry (upper bounds)
};

template <typename T>
WL_HD size_t wl_sfb2d_tile_lds(const WlSfb2dArgs<T>& a) {
    typedef typename WlAcc<T>::type A;
    size_t taps = (size_t)wl_align_up(2 * a.Lw + 2 * a.Lh, 4);
    return sizeof(A) * (taps + (size_t)4 * a.nkr * a.nkc + (size_t)2 * a.TH * a.nkc);
}

WL_HD int wl_floordiv2(int v) { return v >= 0 ? v / 2 : -((-v + 1) / 2); }

template <typename T>
WL_DEV void wl_sfb2d_tile_body(const WlSfb2dArgs<T>& a, const WlCtx& ctx) {
    typedef typename WlAcc<T>::type A;
    const int tx = ctx.tid & 63, ty = ctx.tid >> 6, ny = ctx.nthreads >> 6;
    const int tiles = a.tiles_x * a.tiles_y;
    const int64_t plane = ctx.bid / tiles;
    const int tile = (int)(ctx.bid - plane * tiles);
    const int n0 = (tile / a.tiles_x) * a.TH, w0 = (tile % a.tiles_x) * a.TW;

    A* taps = reinterpret_cast<A*>(ctx.smem);
    A* B = taps + wl_align_up(2 * a.Lw + 2 * a.Lh, 4);   // [4][nkr][nkc]
    A* U = B + (size_t)4 * a.nkr * a.nkc;                // [2][TH][nkc]
    A* gwl = taps, *gwh = taps + a.Lw, *ghl = taps + 2 * a.Lw, *ghh = taps + 2 * a.Lw + a.Lh;
    for (int i = ctx.tid; i < a.Lw; i += ctx.nthreads) { gwl[i] = a.g_w_lo[i]; gwh[i] = a.g_w_hi[i]; }
    for (int i = ctx.tid; i < a.Lh; i += ctx.nthreads) { ghl[i] = a.g_h_lo[i]; ghh[i] = a.g_h_hi[i]; }

    // band index ranges needed by this output tile
    const int kr0 = wl_floordiv2(n0 + a.sh - a.Lh + 2);
    const int kc0 = wl_floordiv2(w0 + a.sw - a.Lw + 2);
    const size_t bplane = (size_t)a.Kh * a.Kw;
    const T* llp = a.ll + (size_t)plane * a.ll_plane_stride;
    const T* hp = a.highs ? a.highs + (size_t)plane * 3 * bplane : nullptr;
    // phase 1: stage the four band tiles
    for (int i = ty; i < a.nkr; i += ny) {
        int r = kr0 + i;
        if (a.circ) r = wl_pmod(r, a.Kh);
        const bool rok = (unsigned)r < (unsigned)a.Kh;
        for (int j = tx; j < a.nkc; j += 64) {
            int c = kc0 + j;
            if (a.circ) c = wl_pmod(c, a.Kw);
            const bool ok = rok && (unsigned)c < (unsigned)a.Kw;
            A v0 = 0, v1 = 0, v2 = 0, v3 = 0;
            if (ok) {
                v0 = (A)llp[(size_t)r * a.ll_row_stride + c];
                if (hp) {
                    const size_t o = (size_t)r * a.Kw + c;
                    v1 = (A)hp[o];
                    v2 = (A)hp[bplane + o];
                    v3 = (A)hp[2 * bplane + o];
                }
            }
            const int o = i * a.nkc + j;
            B[o] = v0;
            B[a.nkr * a.nkc + o] = v1;
            B[2 * a.nkr * a.nkc + o] = v2;
            B[3 * a.nkr * a.nkc + o] = v3;
        }
    }
    ctx.sync();
    // phase 2: column synthesis (along H): (ll,lh)->lo, (hl,hh)->hi
    const int bs = a.nkr * a.nkc;
    for (int n = ty; n < a.TH; n += ny) {
        const int nn = n0 + n + a.sh;                 // tap t = nn - 2k'
        int k_lo = wl_floordiv2(nn - a.Lh + 2) - kr0; // smallest k' with t <= L-1 ... (t = nn-2k' < L)
        int k_hi = wl_floordiv2(nn) - kr0;            // largest k' with t >= 0
        if (k_lo < 0) k_lo = 0;
        if (k_hi > a.nkr - 1) k_hi = a.nkr - 1;
        for (int j = tx; j < a.nkc; j += 64) {
            A lo = 0, hi = 0;
            for (int kk = k_lo; kk <= k_hi; ++kk) {
                const int t = nn - 2 * (kr0 + kk);
                const A g0 = ghl[t], g1 = ghh[t];
                const int o = kk * a.nkc + j;
                lo += B[o] * g0 + B[bs + o] * g1;
                hi += B[2 * bs + o] * g0 + B[3 * bs + o] * g1;
            }
            U[n * a.nkc + j] = lo;
            U[(a.TH + n) * a.nkc + j] = hi;
        }
    }
    ctx.sync();
    // phase 3: row synthesis (along W) and store
    T* yp = a.y + (size_t)plane * a.OH * a.OW;
    for (int n = ty; n < a.TH; n += ny) {
        if (n0 + n >= a.OH) break;
        const A* ulo = U + n * a.nkc;
        const A* uhi = U + (a.TH + n) * a.nkc;
        for (int w = tx; w < a.TW; w += 64) {
            if (w0 + w >= a.OW) break;
            const int ww = w0 + w + a.sw;
            int k_lo = wl_floordiv2(ww - a.Lw + 2) - kc0;
            int k_hi = wl_floordiv2(ww) - kc0;
            if (k_lo < 0) k_lo = 0;
            if (k_hi > a.nkc - 1) k_hi = a.nkc - 1;
            A acc = 0;
            for (int kk = k_lo; kk <= k_hi; ++kk) {
                const int t = ww - 2 * (kc0 + kk);
                acc += ulo[kk] * gwl[t] + uhi[kk] * gwh[t];
            }
            yp[(size_t)(n0 + n) * a.OW + (w0 + w)] = (T)acc;
        }
    }
}
