// Streaming level-1 DTCWT forward (+ ScatLayer epilogue) over column strips: fwd_j1 (reference
// dtcwt/transform_funcs.py:98-121) and ScatLayerj1_f.forward (scatternet/lowlevel.py:76-111) on the streaming design of
// wl_dwt_strip.h.  Level 1 is an UNDECIMATED separable filter pair (odd tap counts, 'same' size), so the kernel is
// memory-bound by a wide margin (19 packed FMAs per pixel against 20 bytes): what counts is that every input row is read
// once, by LDS-DMA, and that nothing but the outputs goes back.
//
// A workgroup owns one (plane, strip of 2-column quads, segment of rows):
//   * four stager waves load one input row each per half-batch (4 rows) by LDS-DMA, three half-batches ahead, and stage
//     them as float32 rows whose mirrored (symmetric extension of the replicate-padded signal) or zero halo cells are
//     materialised, laid out so that every compute lane reads aligned 8-byte words;
//   * a compute lane owns the TWO columns of a quad column.  Per new row it reads 2 + 2M samples, runs the row filter
//     pair for both columns ((lo, hi) packed: one v_pk_fma_f32 per tap and column, the sample picked by op_sel) into
//     circular windows of 2M+1 rows in registers, and - M rows later - the column filter pair of the completed row:
//     (ll, hl) and (lh, hh) packed.  Every second row a 2x2 quad is complete: q2c and the stores (or the scattering
//     magnitudes) go through wl_dtfwd1_quad_out, shared with the tile kernels.
#pragma once
#include "wl_common.h"
#include "wl_dwt_rows.h"        // packed-FMA helpers
#include "wl_dwt_strip.h"       // WlStage, WL_STRIP_* constants
#include "wl_dtcwt_kernels.h"   // WlDtFwd1Args, wl_dtfwd1_quad_out

// acc += w * (c, c): both halves of the packed pair meet the same tap
#if defined(__HIPCC__)
WL_DEV void wl_pk_fma_vs(wl_v2& acc, wl_v2 w, wl_v2 tap2) {
    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "s"(tap2));
}
WL_DEV wl_v2 wl_pk_mul_vs(wl_v2 w, wl_v2 tap2) {
    wl_v2 r;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(w), "s"(tap2));
    return r;
}
#else
inline void wl_pk_fma_vs(wl_v2& acc, wl_v2 w, wl_v2 tap2) { acc.x = __builtin_fmaf(w.x, tap2.x, acc.x); acc.y = __builtin_fmaf(w.y, tap2.y, acc.y); }
inline wl_v2 wl_pk_mul_vs(wl_v2 w, wl_v2 tap2) { return wl_v2{w.x * tap2.x, w.y * tap2.y}; }
#endif

template <typename T>
struct WlDtStripArgs {
    WlDtFwd1Args<T> f;             // tensors, taps, sizes and the epilogue switches of the level-1 forward
    int64_t nblocks;
    int nstrips, strip_quads;      // quad columns per strip; the last strip may be narrower
    int nseg, seg_rows;            // output rows per segment (a multiple of 4)
    int dma_off, dma_pitch, st_off, st_pitch, lds_bytes;
};

template <typename T, int L0, int L1>
struct WlDtFwd1Strip {
    typedef WlDtStripArgs<T> Args;
    static const int kWaves = WL_STRIP_CWAVES + 4;
    static const int kThreads = 64 * kWaves;
    static const int kMinWaves = 4;
    static const int SZ = (int)sizeof(T);
    static const int A = 16 / SZ;
    static const int M0 = L0 / 2, M1 = L1 / 2, M = M0 > M1 ? M0 : M1;
    static const int LW = (2 * M + 1 + 3) / 4 * 4;     // window slots: a multiple of the 4 rows of a half-batch
    static const int PERIOD = LW / 4;
    static const int NS = 2 + 2 * M;                   // samples a lane reads per row
    static const int NC2 = NS / 2;                     // as 8-byte words
    static const int D = WL_STRIP_D;

    struct Strip {
        int q0, q1;            // quad columns [q0, q1): pixel columns [2 q0, 2 q1)
        int e_lo, c0a, np, ppr, ng, dm, lane_off;
        int r_lo, r_hi;        // output rows of this segment
        int e_first, nfeeds, nhb;
    };
    static WL_HD Strip geometry(const Args& a, int strip, int seg) {
        Strip s;
        const int quads = a.f.We / 2;
        s.q0 = strip * a.strip_quads;
        s.q1 = s.q0 + a.strip_quads < quads ? s.q0 + a.strip_quads : quads;
        s.e_lo = 2 * s.q0 - M;
        const int e_hi = 2 * s.q1 - 1 + M;
        s.c0a = s.e_lo >= 0 ? s.e_lo / A * A : -((-s.e_lo + A - 1) / A * A);
        const int last = e_hi > a.f.W - 1 ? a.f.W - 1 : e_hi;
        s.np = (last - s.c0a) / A + 1;
        s.ppr = (s.np + 63) / 64;
        s.ng = s.np * A / 4;
        const int d = s.e_lo - s.c0a;
        s.dm = d & 1;
        s.lane_off = (d + s.dm) * 4;
        s.r_lo = seg * a.seg_rows;
        s.r_hi = s.r_lo + a.seg_rows < a.f.He ? s.r_lo + a.seg_rows : a.f.He;
        s.e_first = s.r_lo - M;
        s.nfeeds = s.r_hi - s.r_lo + 2 * M;
        s.nhb = (s.nfeeds + 3) / 4;
        return s;
    }

    // ---- stager wave: row `sidx` of every half-batch ----------------------------------------------------------------
    static WL_DEV void stager(const Args& a, const Strip& s, const WlCtx& ctx, int64_t plane, int lane, int sidx) {
        const WlDtFwd1Args<T>& f = a.f;
        const char* xp = reinterpret_cast<const char*>(f.x + (size_t)plane * f.H * f.W);
        const int row_stride = f.W * SZ;
        const int padr = f.He - f.H, padc = f.We - f.W;
        const int e_last = s.e_first + s.nfeeds - 1;
        int gbyte[WL_STRIP_MAXPPR];
#pragma unroll
        for (int q = 0; q < WL_STRIP_MAXPPR; ++q) {
            const int p = q * 64 + lane;
            const int col = s.c0a + p * A;
            const bool on = q < s.ppr && p < s.np && (unsigned)col < (unsigned)f.W;
            gbyte[q] = on ? col * SZ : -1;
        }
        const int ngl = (s.ng + 63) >> 6;
        int imin, imax;
        {
            int g_lo = 0, g_hi = s.ng;
            if (s.c0a < 0) g_lo = (-s.c0a + 3) / 4;
            const int lim = (f.W - s.c0a) / 4;
            if (lim < g_hi) g_hi = lim;
            imin = g_lo > lane ? (g_lo - lane + 63) / 64 : 0;
            imax = g_hi > lane ? (g_hi - lane + 63) / 64 : 0;
        }
        // halo cells left / right of the row (symmetric extension of the replicate-padded row): one item per lane
        int hdst = -1, hsrc = -1;
        {
            const int e_hi = 2 * s.q1 - 1 + M;
            const int nl = s.e_lo < 0 ? -s.e_lo : 0, nr = e_hi > f.W - 1 ? e_hi - (f.W - 1) : 0;
            if (f.ext != WL_EXT_ZERO && lane < nl + nr) {
                const int e = lane < nl ? s.e_lo + lane : f.W + (lane - nl);
                const int src = wl_ext_padded(e, f.W, 0, padc, f.ext);
                hdst = (e - s.c0a + s.dm) * 4;
                hsrc = (src - s.c0a) * SZ;
            }
        }
        auto src_row = [&](int h) {
            int e = s.e_first + 4 * h + sidx;
            e = e < e_last ? e : e_last;
            return wl_ext_padded(e, f.H, 0, padr, f.ext);     // -1: a row of zeros
        };
        auto issue = [&](int h) {
            const int slot = a.dma_off + ((h % D) * 4 + sidx) * a.dma_pitch;
            int r = src_row(h);
            r = r < 0 ? 0 : r;                                // a dummy row keeps the DMA count exact
            const char* grow = xp + (size_t)r * row_stride;
#pragma unroll
            for (int q = 0; q < WL_STRIP_MAXPPR; ++q)
                if (q < s.ppr) wl_dma16(ctx, (unsigned)(slot + q * 1024), grow + gbyte[q], gbyte[q] >= 0);
        };
        for (int h = 0; h < D && h < s.nhb; ++h) issue(h);
        int inflight = D < s.nhb ? D : s.nhb;
        for (int hb = 0; hb < s.nhb; ++hb) {
            wl_wait_vm_dyn((inflight - 1) * s.ppr);
            --inflight;
            const char* drow0 = ctx.smem + a.dma_off + ((hb % D) * 4 + sidx) * a.dma_pitch;
            char* srow0 = ctx.smem + a.st_off + ((hb & 1) * 4 + sidx) * a.st_pitch;
            const char* srow = drow0 + lane * 4 * SZ;
            char* drow = srow0 + lane * 16 + s.dm * 4;
            if (src_row(hb) < 0) WlStage<T>::template stage_row<4>(srow, drow, imin, imax, ngl);
            else if (s.dm == 0) WlStage<T>::template stage_row<0>(srow, drow, imin, imax, ngl);
            else WlStage<T>::template stage_row<1>(srow, drow, imin, imax, ngl);
            if (hdst >= 0) *reinterpret_cast<float*>(srow0 + hdst) = src_row(hb) < 0 ? 0.f : (float)*reinterpret_cast<const T*>(drow0 + hsrc);
            ctx.sync();
            if (hb + D < s.nhb) { issue(hb + D); ++inflight; }
        }
        wl_wait_vm<0>();
    }

    // ---- compute wave ---------------------------------------------------------------------------------------------
    struct Wave {
        wl_v2 tr[2 * M + 1];       // row-filter tap pairs (h0[t], h1[t]), both centred in 2M+1 slots (zeros outside)
        wl_v2 c0[L0], c1[L1];      // column-filter taps, duplicated into both halves: (h0[t], h0[t]), (h1[t], h1[t])
    };
    template <int COL> static WL_DEV wl_v2 row_filter(const Wave& R, const wl_v2 (&s)[NC2]) {
        // column COL of the quad: samples COL .. COL + 2M
        wl_v2 a0 = (COL & 1) ? wl_pk_mul_y(R.tr[0], s[0]) : wl_pk_mul_x(R.tr[0], s[0]);
        wl_v2 a1 = ((COL + 1) & 1) ? wl_pk_mul_y(R.tr[1], s[(COL + 1) / 2]) : wl_pk_mul_x(R.tr[1], s[(COL + 1) / 2]);
#pragma unroll
        for (int t = 2; t < 2 * M + 1; ++t) {
            wl_v2& acc = (t & 1) ? a1 : a0;
            if ((COL + t) & 1) wl_pk_fma_y(acc, R.tr[t], s[(COL + t) / 2]); else wl_pk_fma_x(acc, R.tr[t], s[(COL + t) / 2]);
        }
        return a0 + a1;
    }
    // column filters of the row whose window is centred on slot `c`: aL = (ll, hl), aH = (lh, hh)
    static WL_DEV void col_filter(const Wave& R, const wl_v2 (&w)[LW], int c, wl_v2& aL, wl_v2& aH) {
        aL = wl_pk_mul_vs(w[(c + LW - M0) % LW], R.c0[0]);
        aH = wl_pk_mul_vs(w[(c + LW - M1) % LW], R.c1[0]);
#pragma unroll
        for (int t = 1; t < L0; ++t) wl_pk_fma_vs(aL, w[(c + LW - M0 + t) % LW], R.c0[t]);
#pragma unroll
        for (int t = 1; t < L1; ++t) wl_pk_fma_vs(aH, w[(c + LW - M1 + t) % LW], R.c1[t]);
    }

    static WL_DEV void compute(const Args& a, const Strip& s, const WlCtx& ctx, int64_t plane, int cw, int lane) {
        const WlDtFwd1Args<T>& f = a.f;
        const int q = s.q0 + 64 * cw + lane;                  // quad column
        const bool active = q < s.q1;
        Wave R;
#pragma unroll
        for (int t = 0; t < 2 * M + 1; ++t) {
            const int t0 = t - (M - M0), t1 = t - (M - M1);
            const float v0 = t0 >= 0 && t0 < L0 ? (float)f.h0[t0 >= 0 && t0 < L0 ? t0 : 0] : 0.f;
            const float v1 = t1 >= 0 && t1 < L1 ? (float)f.h1[t1 >= 0 && t1 < L1 ? t1 : 0] : 0.f;
            R.tr[t] = wl_uniform_v2(wl_v2{v0, v1});
        }
#pragma unroll
        for (int t = 0; t < L0; ++t) R.c0[t] = wl_uniform_v2(wl_v2{(float)f.h0[t], (float)f.h0[t]});
#pragma unroll
        for (int t = 0; t < L1; ++t) R.c1[t] = wl_uniform_v2(wl_v2{(float)f.h1[t], (float)f.h1[t]});
        const int soff = s.lane_off + 8 * (active ? q - s.q0 : 0);
        wl_v2 wa[LW], wb[LW];                                 // circular windows of (lo, hi) rows: the quad's two columns
#pragma unroll
        for (int t = 0; t < LW; ++t) wa[t] = wb[t] = wl_v2{0.f, 0.f};
        float msum[6];
        char* const smem = ctx.smem;
        for (int hb0 = 0; hb0 < s.nhb; hb0 += PERIOD) {
#pragma unroll
            for (int ph = 0; ph < PERIOD; ++ph) {
                const int hb = hb0 + ph;
                if (hb >= s.nhb) break;
                ctx.sync();
                if (!active) continue;
                const char* slot = smem + a.st_off + (hb & 1) * 4 * a.st_pitch + soff;
                wl_v2 sr[4][NC2];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int u = 0; u < NC2; ++u) {
                        const wl_f2 t = *reinterpret_cast<const wl_f2*>(slot + i * a.st_pitch + 8 * u);
                        sr[i][u] = wl_v2{t.x, t.y};
                    }
                float ll[4], lh[4], hl[4], hh[4];             // the quad being completed: p = 2 * (row & 1) + column
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int w = (4 * ph + i) % LW;          // slot of the new row e = e_first + 4 hb + i
                    wa[w] = row_filter<0>(R, sr[i]);
                    wb[w] = row_filter<1>(R, sr[i]);
                    // the row that is complete now: o = e - M, its window centred on slot w - M
                    const int o = s.e_first + 4 * hb + i - M;
                    wl_v2 aL, aH, bL, bH;
                    col_filter(R, wa, (w + LW - M) % LW, aL, aH);
                    col_filter(R, wb, (w + LW - M) % LW, bL, bH);
                    const int p = 2 * (i & 1);                // (o is even exactly when i is: r_lo and 2M are even)
                    ll[p] = aL.x; hl[p] = aL.y; lh[p] = aH.x; hh[p] = aH.y;
                    ll[p + 1] = bL.x; hl[p + 1] = bL.y; lh[p + 1] = bH.x; hh[p + 1] = bH.y;
                    if ((i & 1) && o - 1 >= s.r_lo && o < s.r_hi)
                        wl_dtfwd1_quad_out<T, 0>(f, plane, 0, o - 1, 2 * q, ll, lh, hl, hh, msum);
                }
            }
        }
    }

    static WL_DEV void run(const Args& a, const WlCtx& ctx) {
        const int tid = ctx.tid;
        const int wave = wl_uniform(tid >> 6), lane = tid & 63;
        const int64_t lbid = wl_xcd_remap(ctx.bid, a.nblocks);
        const int per_plane = a.nstrips * a.nseg;
        const int64_t plane = lbid / per_plane;
        const int rem = (int)(lbid - plane * per_plane);
        const int seg = rem / a.nstrips, strip = rem - seg * a.nstrips;
        const Strip s = geometry(a, strip, seg);
        for (int i = tid * 16; i < a.lds_bytes; i += kThreads * 16) {
            wl_f4 z; z.x = z.y = z.z = z.w = 0.f;
            *reinterpret_cast<wl_f4*>(ctx.smem + i) = z;
        }
        ctx.sync();
        if (wave >= WL_STRIP_CWAVES) {
#if defined(__HIPCC__)
            __builtin_amdgcn_s_setprio(2);
#endif
            stager(a, s, ctx, plane, lane, wave - WL_STRIP_CWAVES);
        } else if (64 * wave < s.q1 - s.q0) {
            compute(a, s, ctx, plane, wave, lane);
        } else {
            for (int hb = 0; hb < s.nhb; ++hb) ctx.sync();
        }
    }
};
