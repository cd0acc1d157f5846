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

// =================================================================================================================
// Streaming level-1 DTCWT inverse over column strips: inv_j1 (reference dtcwt/transform_funcs.py:152-184 = c2q x 3,
// 4 colfilter, 2 rowfilter, 3 adds) with both inputs present.  The separable filters commute, so
//     y = C_g0 (R_g0 ll + R_g1 hl) + C_g1 (R_g0 lh + R_g1 hh)          (R: along rows, C: along columns)
// is evaluated row filter first: a compute lane owns the two columns of a quad column, keeps circular windows of the
// row-filtered pair (A, B) = (R_g0 ll + R_g1 hl, R_g0 lh + R_g1 hh) in registers and emits an output row M rows later.
//   * half-batch = one quad row.  Every lane of the four stager waves owns ONE quad of it: it loads the quad's eight
//     sources (2 x 2 ll pixels, 6 orientation (re, im) pairs: 8-byte loads, consecutive lanes on consecutive addresses)
//     straight into registers one half-batch ahead, runs c2q, and writes the quad's four pixels as (ll, lh, hl, hh)
//     16-byte cells into the staged rows - plus the mirrored copies of the pixels it owns (symmetric extension; rows
//     above / below the plane are whole quad rows flipped) - so that a compute lane reads a pixel's four channels as ONE
//     aligned word whose halves are the packed operands of the row filter.  (A first version brought the rows in by
//     LDS-DMA like the other strip kernels: with 16 bytes read per output pixel, and DMA instructions that cost the CU the
//     same ~100 cycles full or nearly empty, it was bound by the DMA instruction rate at 0.44 of the HBM peak.)
// =================================================================================================================
template <typename T>
struct WlDtIStripArgs {
    WlDtInv1Args<T> f;             // tensors, taps, sizes
    int64_t nblocks;
    int nstrips, strip_quads, nseg, seg_rows;
    int st_off, st_pitch, lds_bytes;
};

// SCAT = 1: ScatLayerj1_f.backward (reference scatternet/lowlevel.py:114-137) - the same inverse fed by the scattering
// prologue: the lowpass quad is the 2 x 2 nearest-neighbour upsampling of dZ[:, 0] / 4, the band-pass pair of orientation o is
// dZ[:, 1 + o] * (re / r, im / r) with the saved quotients; the stager lanes form them from 19 coalesced 4-byte loads per
// quad (dZ (N, 7, C, h, w), drdx / drdy (N, 6, C, h, w)) instead of the eight 8-byte loads of the plain inverse.
// PP = 2 (planes of up to 256 columns): a workgroup owns TWO consecutive planes - compute waves 0, 1 and stager waves 0, 1 the
// first, 2, 3 the second, the staged rows side by side - so that all eight waves work (wl_dtcwt_fused.h does the same for the
// lean forward kernels).  PP = 4: planes of up to 128 columns, one compute wave and one stager wave each.
template <typename T, int L0, int L1, int SCAT = 0, int PP = 1>
struct WlDtInv1Strip {
    static_assert(PP == 1 || PP == 2 || PP == 4, "planes per workgroup");
    typedef WlDtIStripArgs<T> Args;
    static const int CW = 4;                           // compute waves: up to 256 quad columns per strip
    static const int kWaves = CW + 4;
    static const int kThreads = 64 * kWaves;
    static const int SZ = (int)sizeof(T);
    static const int M0 = L0 / 2, M1 = L1 / 2, M = M0 > M1 ? M0 : M1, ME = (M + 1) & ~1;
    static const int kMinWaves = M > 4 ? 2 : 4;        // (13 / 19 taps: 127 registers when the allocator is given room, 6 spilled at a bound of 128 - two workgroups per CU either way)
    static const int LW = 2 * M + 2;                   // window slots (even): rotation by the 2 rows of a half-batch
    static const int PERIOD = LW / 2;
#ifndef WL_DTI1_PF
#define WL_DTI1_PF 2
#endif
    static const int PF = WL_DTI1_PF;                  // register sets of a stager
    static const int NPX = 2 + 2 * M;                  // pixels a lane reads per row

    struct Strip {
        int q0, q1;            // quad columns [q0, q1) -> output pixel columns [2 q0, 2 q1)
        int e_lo, px0;         // first extended pixel column a lane reads (2 q0 - M); pixel column of staged cell 0 (even)
        int Qa, nq;            // quads [Qa, Qa + nq) are staged (those inside the plane), one per stager lane
        int r_lo, r_hi, e_first, nhb;
    };
    static WL_HD Strip geometry(const Args& a, int strip, int seg) {
        Strip s;
        const int W2 = a.f.W / 2;
        s.q0 = strip * a.strip_quads;
        s.q1 = s.q0 + a.strip_quads < W2 ? s.q0 + a.strip_quads : W2;
        s.e_lo = 2 * s.q0 - M;
        const int e_hi = 2 * s.q1 - 1 + M;
        s.px0 = s.e_lo >= 0 ? s.e_lo & ~1 : -((-s.e_lo + 1) & ~1);
        int qa = s.px0 / 2, qb = e_hi / 2;
        if (qa < 0) qa = 0;
        if (qb > W2 - 1) qb = W2 - 1;
        s.Qa = qa; s.nq = qb - qa + 1;
        s.r_lo = seg * a.seg_rows;
        s.r_hi = s.r_lo + a.seg_rows < a.f.H ? s.r_lo + a.seg_rows : a.f.H;
        s.e_first = s.r_lo - ME;
        s.nhb = ((s.r_hi - 1 + M - s.e_first) / 2 + 1 + PF - 1) / PF * PF;   // (a multiple of the stagers' register sets)
        return s;
    }
    // extended quad row (pair of extended pixel rows 2 eq, 2 eq + 1) -> source quad row; flip: its two rows swap
    // (symmetric extension of an even number of rows mirrors whole quad rows); -1: zeros
    static WL_HD int src_quad_row(int eq, int H2, int ext, bool& flip) {
        flip = false;
        if ((unsigned)eq < (unsigned)H2) return eq;
        if (ext == WL_EXT_ZERO) return -1;
        flip = true;
        int m = eq < 0 ? -1 - eq : 2 * H2 - 1 - eq;
        return m < 0 ? 0 : (m >= H2 ? H2 - 1 : m);     // (one fold: the launcher requires H2 > M)
    }

    // byte offset of staged cell c (pixel column px0 + c) inside a staged row: even and odd cells live in the two halves of
    // the row, so that the lanes of a wave (one quad = two pixels each) read and write consecutive 16-byte words (with the
    // cells simply side by side every access was a 2-way bank conflict: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.50)
    static WL_HD int cell_off(const Args& a, int c) { return (c & 1) * (a.st_pitch / (2 * PP)) + (c >> 1) * 16; }

    typedef T Pair2 __attribute__((ext_vector_type(2), may_alias));
    struct Quad { Pair2 l0, l1, b[6]; T z0, z[6], dx[6], dy[6]; };   // the sources of one quad, as loaded (SCAT: z0, z, dx, dy)

    static WL_DEV void stager(const Args& a, const Strip& s, const WlCtx& ctx, int64_t plane0, int lane, int sidx0) {
        const WlDtInv1Args<T>& f = a.f;
        const int H2 = f.H / 2, W2 = f.W / 2;
        const size_t qplane = (size_t)H2 * W2;
        const int sub = PP > 1 ? sidx0 / (4 / PP) : 0;        // PP = 2: stager waves 0, 1 the first plane, 2, 3 the second; PP = 4: a plane each
        const int sidx = PP > 1 ? sidx0 % (4 / PP) : sidx0;
        const int64_t plane = plane0 + sub;
        const int sub_off = sub * (a.st_pitch / PP);
        const int j = 64 * sidx + lane;                       // this lane's quad of every quad row
        const int Q = s.Qa + j;
        const bool qon = j < s.nq && plane < f.NC;
        const T* llp = SCAT ? nullptr : f.ll + (size_t)plane * f.ll_plane_stride + 2 * Q;
        const T* hp = SCAT ? nullptr : f.highs + (size_t)plane * 6 * qplane * 2 + 2 * Q;
        // SCAT: my quad column of dZ[n, k, c], drdx[n, o, c], drdy[n, o, c]; entries k / o are C planes apart
        const int64_t n_img = SCAT ? plane / f.C : 0;
        const int c_img = SCAT ? (int)(plane - n_img * f.C) : 0;
        const T* zp = SCAT ? f.sz + ((size_t)n_img * 7 * f.C + c_img) * qplane + Q : nullptr;
        const T* dxp = SCAT ? f.sdx + ((size_t)n_img * 6 * f.C + c_img) * qplane + Q : nullptr;
        const T* dyp = SCAT ? f.sdy + ((size_t)n_img * 6 * f.C + c_img) * qplane + Q : nullptr;
        const size_t eplane = (size_t)f.C * qplane;
        // staged cells (16 bytes per pixel) of its two pixel columns, and of their mirror images inside the strip's range
        const int e_hi = 2 * s.q1 - 1 + M;
        int cdst[2], mdst[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int p = 2 * Q + c;
            cdst[c] = cell_off(a, p - s.px0) + sub_off;
            int e = -1000000;
            if (f.ext != WL_EXT_ZERO && qon) {
                if (p < M && -1 - p >= s.e_lo) e = -1 - p;
                if (p >= f.W - M && 2 * f.W - 1 - p <= e_hi) e = 2 * f.W - 1 - p;
            }
            mdst[c] = e == -1000000 ? -1 : cell_off(a, e - s.px0) + sub_off;
        }
        auto load = [&](int h, Quad& qd) {
            bool flip;
            int sq = src_quad_row(s.e_first / 2 + h, H2, f.ext, flip);
            sq = sq < 0 ? 0 : sq;
            if (!qon) return;
            if (SCAT) {
                const size_t ro = (size_t)sq * W2;
                qd.z0 = zp[ro];
#pragma unroll
                for (int o = 0; o < 6; ++o) {
                    qd.z[o] = zp[(size_t)(o + 1) * eplane + ro];
                    qd.dx[o] = dxp[(size_t)o * eplane + ro];
                    qd.dy[o] = dyp[(size_t)o * eplane + ro];
                }
                return;
            }
            qd.l0 = *reinterpret_cast<const Pair2*>(llp + (size_t)(2 * sq) * f.ll_row_stride);
            qd.l1 = *reinterpret_cast<const Pair2*>(llp + (size_t)(2 * sq + 1) * f.ll_row_stride);
#pragma unroll
            for (int o = 0; o < 6; ++o) qd.b[o] = *reinterpret_cast<const Pair2*>(hp + ((size_t)o * qplane + (size_t)sq * W2) * 2);
        };
        const float k = (float)WL_SQRT1_2;
        auto stage = [&](int hb, const Quad& qd) {
            bool flip;
            const int sq = src_quad_row(s.e_first / 2 + hb, H2, f.ext, flip);
            char* sslot = ctx.smem + a.st_off + (hb & 1) * 2 * a.st_pitch;
            if (!qon) return;
            float re[6], im[6];
#pragma unroll
            for (int o = 0; o < 6; ++o) {
                if (SCAT) { re[o] = (float)qd.z[o] * (float)qd.dx[o]; im[o] = (float)qd.z[o] * (float)qd.dy[o]; }
                else { re[o] = (float)qd.b[o].x; im[o] = (float)qd.b[o].y; }
            }
            // c2q (dtcwt/lowlevel.py:263-295): orientation pairs (0,5) -> lh, (2,3) -> hl, (1,4) -> hh
            float v[2][2][4];                                 // [row][col][channel]
#pragma unroll
            for (int ch = 1; ch < 4; ++ch) {
                const int o1 = ch == 1 ? 0 : (ch == 2 ? 2 : 1), o2 = ch == 1 ? 5 : (ch == 2 ? 3 : 4);
                v[0][0][ch] = (re[o1] + re[o2]) * k; v[0][1][ch] = (im[o1] + im[o2]) * k;
                v[1][0][ch] = (im[o1] - im[o2]) * k; v[1][1][ch] = (re[o2] - re[o1]) * k;
            }
            if (SCAT) v[0][0][0] = v[0][1][0] = v[1][0][0] = v[1][1][0] = 0.25f * (float)qd.z0;
            else { v[0][0][0] = (float)qd.l0.x; v[0][1][0] = (float)qd.l0.y; v[1][0][0] = (float)qd.l1.x; v[1][1][0] = (float)qd.l1.y; }
            const bool zero = sq < 0;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                char* drow = sslot + (flip ? 1 - i : i) * a.st_pitch;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    wl_vf4 w;
                    w.x = zero ? 0.f : v[i][c][0]; w.y = zero ? 0.f : v[i][c][1];
                    w.z = zero ? 0.f : v[i][c][2]; w.w = zero ? 0.f : v[i][c][3];
                    *reinterpret_cast<wl_vf4*>(drow + cdst[c]) = w;
                    if (mdst[c] >= 0) *reinterpret_cast<wl_vf4*>(drow + mdst[c]) = w;
                }
            }
        };
        // software pipeline in registers: PF sets, the loads of half-batches hb + 1 .. hb + PF - 1 are in flight while hb is
        // staged (measured: PF = 2 is the fastest; 3 is 1 % and 4 is 5 % slower on the whole inverse).
        // No branch in the loop (nhb is a multiple of PF; behind the last half-batch: that one again), so that the compiler
        // counts the loads and waits for exactly the oldest set.
        Quad qq[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) load(u < s.nhb ? u : s.nhb - 1, qq[u]);
        for (int hb = 0; hb < s.nhb; hb += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                stage(hb + u, qq[u]);
                ctx.sync();
                load(hb + u + PF < s.nhb ? hb + u + PF : s.nhb - 1, qq[u]);
            }
        }
    }

    // The 13 / 19-tap pair (near_sym_b): 51 duplicated tap pairs do not fit the scalar file (129 spilled scalars, a v_readlane
    // per tap).  Its row filter reads the column filter's pairs instead: the packed FMA takes the pair's low / high half for
    // both of its lanes (op_sel) - 19 pairs in all, the same products in the same order.
    static const bool PAIRS = M > 4;
    struct Wave {
        wl_v2 r0[PAIRS ? 1 : L0], r1[PAIRS ? 1 : L1];   // row-filter taps, duplicated: (g0[t], g0[t]) meets (ll, lh), (g1[t], g1[t]) meets (hl, hh)
        wl_v2 cc[2 * M + 1];       // column-filter tap pairs (g0[t], g1[t]), both centred in 2M+1 slots
    };
    // acc (+)= w * (c, c), c = the low / high half of a scalar tap pair
    template <int HI> static WL_DEV void fma_half(wl_v2& acc, wl_v2 w, wl_v2 pair) {
#if defined(__HIPCC__)
        if (HI) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w), "s"(pair));
        else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(w), "s"(pair));
#else
        const float c = HI ? pair.y : pair.x;
        acc.x = __builtin_fmaf(w.x, c, acc.x); acc.y = __builtin_fmaf(w.y, c, acc.y);
#endif
    }
    template <int HI> static WL_DEV wl_v2 mul_half(wl_v2 w, wl_v2 pair) {
#if defined(__HIPCC__)
        wl_v2 r;
        if (HI) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(w), "s"(pair));
        else asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(w), "s"(pair));
        return r;
#else
        const float c = HI ? pair.y : pair.x;
        return wl_v2{w.x * c, w.y * c};
#endif
    }
    // (A, B) of both output columns of the quad from the lane's pixels px[0..NPX): (x, y) = (ll, lh), (z, w) = (hl, hh).
    // Four accumulator chains, interleaved: on gfx950 a packed FMA that reads the result of one of the two instructions
    // before it costs a wait state (the compiler pads with s_nop, which takes an issue slot like any instruction).
    static WL_DEV void row_filter2(const Wave& R, const wl_vf4 (&px)[NPX], wl_v2& ra, wl_v2& rb) {
        if (PAIRS) {
            wl_v2 a0 = mul_half<0>(wl_v2{px[M - M0].x, px[M - M0].y}, R.cc[M - M0]);
            wl_v2 b0 = mul_half<0>(wl_v2{px[1 + M - M0].x, px[1 + M - M0].y}, R.cc[M - M0]);
            wl_v2 a1 = mul_half<1>(wl_v2{px[M - M1].z, px[M - M1].w}, R.cc[M - M1]);
            wl_v2 b1 = mul_half<1>(wl_v2{px[1 + M - M1].z, px[1 + M - M1].w}, R.cc[M - M1]);
#pragma unroll
            for (int t = 1; t < (L0 > L1 ? L0 : L1); ++t) {
                if (t < L0) {
                    fma_half<0>(a0, wl_v2{px[M - M0 + t].x, px[M - M0 + t].y}, R.cc[M - M0 + (t < L0 ? t : 0)]);
                    fma_half<0>(b0, wl_v2{px[1 + M - M0 + t].x, px[1 + M - M0 + t].y}, R.cc[M - M0 + (t < L0 ? t : 0)]);
                }
                if (t < L1) {
                    fma_half<1>(a1, wl_v2{px[M - M1 + t].z, px[M - M1 + t].w}, R.cc[M - M1 + (t < L1 ? t : 0)]);
                    fma_half<1>(b1, wl_v2{px[1 + M - M1 + t].z, px[1 + M - M1 + t].w}, R.cc[M - M1 + (t < L1 ? t : 0)]);
                }
            }
            ra = a0 + a1; rb = b0 + b1;
            return;
        }
        wl_v2 a0 = wl_pk_mul_vs(wl_v2{px[M - M0].x, px[M - M0].y}, R.r0[0]);
        wl_v2 b0 = wl_pk_mul_vs(wl_v2{px[1 + M - M0].x, px[1 + M - M0].y}, R.r0[0]);
        wl_v2 a1 = wl_pk_mul_vs(wl_v2{px[M - M1].z, px[M - M1].w}, R.r1[0]);
        wl_v2 b1 = wl_pk_mul_vs(wl_v2{px[1 + M - M1].z, px[1 + M - M1].w}, R.r1[0]);
#pragma unroll
        for (int t = 1; t < (L0 > L1 ? L0 : L1); ++t) {
            if (t < L0) {
                wl_pk_fma_vs(a0, wl_v2{px[M - M0 + t].x, px[M - M0 + t].y}, R.r0[t]);
                wl_pk_fma_vs(b0, wl_v2{px[1 + M - M0 + t].x, px[1 + M - M0 + t].y}, R.r0[t]);
            }
            if (t < L1) {
                wl_pk_fma_vs(a1, wl_v2{px[M - M1 + t].z, px[M - M1 + t].w}, R.r1[t]);
                wl_pk_fma_vs(b1, wl_v2{px[1 + M - M1 + t].z, px[1 + M - M1 + t].w}, R.r1[t]);
            }
        }
        ra = a0 + a1; rb = b0 + b1;
    }
    // output samples of the row centred on slot c, both columns (even / odd taps in chains of their own)
    static WL_DEV void col_filter2(const Wave& R, const wl_v2 (&wa)[LW], const wl_v2 (&wb)[LW], int c, float& ya, float& yb) {
        wl_v2 a0 = wl_pk_mul_vs(wa[(c + LW - M) % LW], R.cc[0]), b0 = wl_pk_mul_vs(wb[(c + LW - M) % LW], R.cc[0]);
        wl_v2 a1 = wl_pk_mul_vs(wa[(c + LW - M + 1) % LW], R.cc[1]), b1 = wl_pk_mul_vs(wb[(c + LW - M + 1) % LW], R.cc[1]);
#pragma unroll
        for (int t = 2; t < 2 * M + 1; ++t) {
            if (t & 1) { wl_pk_fma_vs(a1, wa[(c + LW - M + t) % LW], R.cc[t]); wl_pk_fma_vs(b1, wb[(c + LW - M + t) % LW], R.cc[t]); }
            else { wl_pk_fma_vs(a0, wa[(c + LW - M + t) % LW], R.cc[t]); wl_pk_fma_vs(b0, wb[(c + LW - M + t) % LW], R.cc[t]); }
        }
        a0 += a1; b0 += b1;
        ya = a0.x + a0.y; yb = b0.x + b0.y;
    }

    static WL_DEV void compute(const Args& a, const Strip& s, const WlCtx& ctx, int64_t plane0, int cw0, int lane) {
        const WlDtInv1Args<T>& f = a.f;
        const int sub = PP > 1 ? cw0 / (CW / PP) : 0;
        const int cw = PP > 1 ? cw0 % (CW / PP) : cw0;
        const int64_t plane = plane0 + sub;
        const int q = s.q0 + 64 * cw + lane;
        const bool active = q < s.q1 && plane < f.NC;
        Wave R;
#pragma unroll
        for (int t = 0; t < (PAIRS ? 0 : L0); ++t) R.r0[t] = wl_uniform_v2(wl_v2{(float)f.g0[t], (float)f.g0[t]});
#pragma unroll
        for (int t = 0; t < (PAIRS ? 0 : L1); ++t) R.r1[t] = wl_uniform_v2(wl_v2{(float)f.g1[t], (float)f.g1[t]});
#pragma unroll
        for (int t = 0; t < 2 * M + 1; ++t) {
            const int t0 = t - (M - M0), t1 = t - (M - M1);
            const float v0 = t0 >= 0 && t0 < L0 ? (float)f.g0[t0 >= 0 && t0 < L0 ? t0 : 0] : 0.f;
            const float v1 = t1 >= 0 && t1 < L1 ? (float)f.g1[t1 >= 0 && t1 < L1 ? t1 : 0] : 0.f;
            R.cc[t] = wl_uniform_v2(wl_v2{v0, v1});
        }
        const int soff = (active ? q - s.q0 : 0) * 16 + sub * (a.st_pitch / PP);   // cell (e_lo - px0) + 2 (q - q0) + u: see cell_off (e_lo - px0 = M & 1)
        const int hp2 = a.st_pitch / (2 * PP);
        char* const yp = reinterpret_cast<char*>(f.y + (size_t)plane * f.H * f.W);
        const unsigned rowb = (unsigned)f.W * SZ, colb = (unsigned)(2 * q) * SZ;
        wl_v2 wa[LW], wb[LW];
#pragma unroll
        for (int t = 0; t < LW; ++t) wa[t] = wb[t] = wl_v2{0.f, 0.f};
        char* const smem = ctx.smem;
        for (int hb0 = 0; hb0 < s.nhb; hb0 += PERIOD) {
#pragma unroll
            for (int ph = 0; ph < PERIOD; ++ph) {
                const int hb = hb0 + ph;
                if (hb >= s.nhb) break;
                ctx.sync();
                if (!active) continue;
                const char* slot = smem + a.st_off + (hb & 1) * 2 * a.st_pitch + soff;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    wl_vf4 px[NPX];
#pragma unroll
                    for (int u = 0; u < NPX; ++u)
                        px[u] = *reinterpret_cast<const wl_vf4*>(slot + i * a.st_pitch + (((M & 1) + u) & 1) * hp2 + (((M & 1) + u) >> 1) * 16);
                    const int w = (2 * ph + i) % LW;          // slot of the new row e = e_first + 2 hb + i
                    row_filter2(R, px, wa[w], wb[w]);
                    const int o = s.e_first + 2 * hb + i - M;   // the row that is complete now
                    float ya, yb;
                    col_filter2(R, wa, wb, (w + LW - M) % LW, ya, yb);
                    if (o >= s.r_lo && o < s.r_hi) {
                        typedef T Vec2 __attribute__((ext_vector_type(2)));
                        Vec2 v = {(T)ya, (T)yb};
                        *reinterpret_cast<Vec2*>(yp + (unsigned)o * rowb + colb) = v;
                    }
                }
            }
        }
    }

    static WL_DEV void run(const Args& a, const WlCtx& ctx) {
        const int tid = ctx.tid;
        const int wave = wl_uniform(tid >> 6), lane = tid & 63;
        const int64_t lbid = wl_xcd_remap(ctx.bid, a.nblocks);
        const int per_plane = a.nstrips * a.nseg;
        const int64_t pidx = lbid / per_plane;
        const int64_t plane = pidx * PP;                      // (PP = 2: the first of the workgroup's two planes)
        const int rem = (int)(lbid - pidx * per_plane);
        const int seg = rem / a.nstrips, strip = rem - seg * a.nstrips;
        const Strip s = geometry(a, strip, seg);
        for (int i = tid * 16; i < a.lds_bytes; i += kThreads * 16) {
            wl_f4 z; z.x = z.y = z.z = z.w = 0.f;
            *reinterpret_cast<wl_f4*>(ctx.smem + i) = z;
        }
        ctx.sync();
        if (wave >= CW) {
#if defined(__HIPCC__)
            __builtin_amdgcn_s_setprio(2);
#endif
            stager(a, s, ctx, plane, lane, wave - CW);
        } else if (64 * (PP > 1 ? wave % (CW / PP) : wave) < s.q1 - s.q0) {
            compute(a, s, ctx, plane, wave, lane);
        } else {
            for (int hb = 0; hb < s.nhb; ++hb) ctx.sync();
        }
    }
};
