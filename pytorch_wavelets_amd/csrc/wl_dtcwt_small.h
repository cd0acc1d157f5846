// Level 1 of the DTCWT / the ScatLayer on SMALL planes (up to 64 x 64: CIFAR / Tiny-ImageNet images, where the scattering
// layers are used upstream), several planes per workgroup: FWD_J1.forward (reference dtcwt/transform_funcs.py:98-121,
// 346-358) and ScatLayerj1_f.forward (scatternet/lowlevel.py:76-111).  The streaming kernels (wl_dtcwt_fused.h) need rows of
// 1 KiB; below that the tile kernel gives every 16 x 64 tile a workgroup and its halo: 0.16 of the HBM roofline on 32 x 32
// planes (round 4).  Here a workgroup of four waves owns G consecutive planes - one contiguous chunk of memory:
//   * the planes go to LDS EXTENDED by M = max(L0, L1) / 2 cells on every side (symmetric extension or zeros, resolved once per
//     cell while loading): no pass below ever tests a boundary;
//   * row pass     (lo, hi)[p][r][c] = sum_t (h0[t] X_p(r, c + t - L0/2), h1[t] X_p(r, c + t - L1/2)) for all H + 2M rows;
//   * column pass  per 2 x 2 quad: its two columns' (lo, hi) rows 2q - M .. 2q + 1 + M (shared by the quad's two rows), the
//     four sub-bands at its four pixels with compile-time taps, then the epilogue shared with every level-1 kernel
//     (wl_dtfwd1_quad_out: lowpass, q2c + the six orientations and / or the ScatLayer output with the saved (re, im) / r).
// A first version that tested the boundary per tap and summed the columns per pixel ran slower than the tile kernel.
#pragma once
#include "wl_common.h"
#include "wl_dtcwt_kernels.h"

template <typename T, int LA, int LB>
struct WlDtFwd1Small {
    typedef WlDtFwd1Args<T> Args;      // run_len = planes per workgroup; mg_* = division magics (see wl_dwt_small.h)
    typedef typename WlAcc<T>::type A;
    static const int kThreads = 256;
    static const int kMinWaves = 2;
    static const int M0 = LA / 2, M1 = LB / 2, M = M0 > M1 ? M0 : M1;
    static WL_HD unsigned divm(unsigned n, unsigned magic) { return magic ? (unsigned)(((unsigned long long)n * magic) >> 32) : n; }
    static WL_DEV int ext1(int i, int n, int ext) {              // one fold: symmetric (half-sample) or zero
        if ((unsigned)i < (unsigned)n) return i;
        if (ext != WL_EXT_SYM) return -1;
        return i < 0 ? -1 - i : 2 * n - 1 - i;
    }
    static WL_DEV void run(const Args& a, const WlCtx& ctx) {
        const int tid = ctx.tid;
        const int G = a.run_len, H = a.H, W = a.W, HW = H * W;
        const int He = H + 2 * M, We = W + 2 * M, EP = He * We;     // extended plane
        const int64_t plane0 = ctx.bid * G;
        const int np = plane0 + G <= a.NC ? G : (int)(a.NC - plane0);
        A* const in = reinterpret_cast<A*>(ctx.smem);              // [p][He][We]
        A* const mid = in + wl_align_up(G * EP, 4);                 // [p][He][W] (lo, hi) pairs
        A c0[LA], c1[LB];
#pragma unroll
        for (int t = 0; t < LA; ++t) c0[t] = a.h0[t];
#pragma unroll
        for (int t = 0; t < LB; ++t) c1[t] = a.h1[t];
        // ---- the extended planes
        const T* const xp = a.x + (size_t)plane0 * HW;
        const int ne = np * EP;
        for (int idx = tid; idx < ne; idx += kThreads) {
            const int p = (int)divm(idx, a.mg_q);                   // / EP
            const int rem = idx - p * EP;
            const int re = (int)divm(rem, a.mg_w);                  // / We
            const int ce = rem - re * We;
            const int r = ext1(re - M, H, a.ext), c = ext1(ce - M, W, a.ext);
            in[idx] = (r < 0 || c < 0) ? (A)0 : (A)xp[p * HW + r * W + c];
        }
        ctx.sync();
        // ---- row pass: every extended row, the plane's own columns
        const int nm = np * He * W;
        for (int idx = tid; idx < nm; idx += kThreads) {
            const int pr = (int)divm(idx, a.mg_w2);                 // / W  = p * He + re
            const int c = idx - pr * W;
            const A* row = in + pr * We + c;                        // extended column c + M is the centre
            A lo = 0, hi = 0;
#pragma unroll
            for (int t = 0; t < LA; ++t) lo += c0[t] * row[M - M0 + t];
#pragma unroll
            for (int t = 0; t < LB; ++t) hi += c1[t] * row[M - M1 + t];
            mid[2 * idx] = lo; mid[2 * idx + 1] = hi;
        }
        ctx.sync();
        // ---- column pass per quad + epilogue
        const int h2 = H / 2, w2 = W / 2, Q = h2 * w2;
        const int nq = np * Q;
        for (int idx = tid; idx < nq; idx += kThreads) {
            const int p = (int)divm(idx, a.nblocks_q);              // / Q
            const int rem = idx - p * Q;
            const int qr = (int)divm(rem, a.mg_qc);                 // / w2
            const int qc = rem - qr * w2;
            A ll[4] = {0, 0, 0, 0}, lh[4] = {0, 0, 0, 0}, hl[4] = {0, 0, 0, 0}, hh[4] = {0, 0, 0, 0};
            // extended rows 2 qr .. 2 qr + 1 + 2 M of the quad's two columns: row s meets tap s - (M - M0) of the upper pixel
            // row and tap s - 1 - (M - M0) of the lower one
            const A* base = mid + 2 * ((p * He + 2 * qr) * W + 2 * qc);
#pragma unroll
            for (int s = 0; s < 2 * M + 2; ++s) {
                const A* v = base + 2 * s * W;
                const A lo0 = v[0], hi0 = v[1], lo1 = v[2], hi1 = v[3];
                const int ta = s - (M - M0), tb = s - (M - M1);     // taps for the upper row (compile-time after unrolling)
                if (ta >= 0 && ta < LA) { ll[0] += c0[ta] * lo0; ll[1] += c0[ta] * lo1; hl[0] += c0[ta] * hi0; hl[1] += c0[ta] * hi1; }
                if (tb >= 0 && tb < LB) { lh[0] += c1[tb] * lo0; lh[1] += c1[tb] * lo1; hh[0] += c1[tb] * hi0; hh[1] += c1[tb] * hi1; }
                if (ta - 1 >= 0 && ta - 1 < LA) { ll[2] += c0[ta - 1] * lo0; ll[3] += c0[ta - 1] * lo1; hl[2] += c0[ta - 1] * hi0; hl[3] += c0[ta - 1] * hi1; }
                if (tb - 1 >= 0 && tb - 1 < LB) { lh[2] += c1[tb - 1] * lo0; lh[3] += c1[tb - 1] * lo1; hh[2] += c1[tb - 1] * hi0; hh[3] += c1[tb - 1] * hi1; }
            }
            wl_dtfwd1_quad_out<T, 0>(a, plane0 + p, 0, 2 * qr, 2 * qc, ll, lh, hl, hh, (A*)nullptr);
        }
    }
};
