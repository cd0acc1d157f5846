// Streaming multi-level 2-D DWT synthesis: the mirror image of wl_dwt_rows.h.  One workgroup owns one (n,c) plane (or
// the top / bottom half of one) and marches down it, coarsest level first:
//   * every coefficient of yl / yh[j] is read from HBM exactly once by LDS-DMA (global_load_lds) into small per-band
//     LDS rings: a band plane is contiguous, so it is copied in whole 1024-byte chunks that ignore the row boundaries
//     (band rows are only 4-byte aligned and 1036 bytes long at the benchmark size), a few rows ahead of their use;
//   * a lane owns one PAIR of output columns (2c, 2c+1) of its level.  Per new coefficient row k ("feed") it runs the
//     polyphase row synthesis  (a,b)[n..n+1] = sum_t (ll|hl, lh|hh)[c + L/2-1 - t] * (g[2t], g[2t+1])  straight from the
//     rings, pushes the result into an L/2-row window in registers and - once the window is full - the polyphase
//     column synthesis of output rows 2(k - (L-2)/2) and the one below: v_pk_fma_f32 on (even,odd) tap pairs with the
//     broadcast sample picked by op_sel.  No zero stuffing, no transposed convolution, no boundary handling at all
//     (every sample a valid lane reads exists);
//   * the output rows of level j+1 (= LL_j) go to an LDS ring that the waves of level j consume; LL_1 .. LL_{J-1}
//     never touch HBM.  HBM traffic = yl, yh[j] in + x out = the algorithmic minimum of SURVEY.md 8(d);
//   * x leaves as 8 contiguous bytes per lane, whole rows written by consecutive lanes of consecutive waves.
// Roles are per wave: compute waves of every level and loader waves (one per band at the finest level, one per coarser
// level); the schedule (who runs how many feeds in which half-batch, with the rings as back-pressure) is simulated by
// the launcher and handed over as a table, exactly as for the analysis kernel.
//
// Restates DWTInverse.forward's level loop (reference dwt/transform2d.py:131-148) = J x SFB2D.forward
// (dwt/lowlevel.py:671-680) = 3J x sfb1d (:226-271) incl. the 'unpad' crop of an LL that is one row / column larger
// than the next finer high-pass; zero / symmetric / reflect / periodic (synthesis is the same for all of them).
#pragma once
#include "wl_common.h"
#include "wl_dwt_rows.h"   // wl_pk_fma_x / _y, wl_pk_mul_x / _y, wl_uniform_v2
#include "wl_lattice.h"

#define WL_IROWS_MAXLEV 3
#define WL_IROWS_WAVES 14
#define WL_IROWS_MAXHB 640
#ifndef WL_IROWS_ABLATE
#define WL_IROWS_ABLATE 0       // measurement builds only (tools/build_ab.sh): 1 no global stores, 2 no DMA, 4 no arithmetic,
#endif                          // 8 per-wave cycle counters into row 0 of x, 64 a barrier in every other half-batch only
#if (WL_IROWS_ABLATE & 8) && defined(__HIPCC__)
#define WL_ITICK() __builtin_readcyclecounter()
#else
#define WL_ITICK() 0ull
#endif
#define WL_IROWS_CHUNK 1024     // bytes one LDS-DMA instruction moves: 64 lanes x 16 bytes

struct WlIRowsLevel {
    int Kh, Kw;         // coefficient rows / cols (size of yh[j]; the LL source is cropped to it)
    int OH, OW;         // output rows / cols = 2K - L + 2
    int src_off[4];     // LDS byte offsets of the four source rings: [0] = LL, [1..3] = lh, hl, hh
    int ll_rows;        // rows of the low-pass ring (power of two, row k in k & (rows - 1)): chosen by the launcher when
    int ll_pitch;       // level j+1 writes it (pitch padded to 16 bytes); = dma_rows and rbytes when it arrives by DMA
    // DMA rings are flat images of dma_rows consecutive coefficient rows (row k at (k & (dma_rows - 1)) * rbytes, no
    // padding): a band plane is contiguous in HBM, so the loader copies it in whole 1024-byte chunks that ignore the
    // row boundaries - an LDS-DMA instruction costs the CU the same ~64 cycles whether it moves 4 bytes or 1024
    int rbytes;         // bytes of a coefficient row = Kw * sizeof(T)
    int dma_rows;       // power of two >= 1024 / rbytes + 4; dma_rows * rbytes is a multiple of 16
    int dma_shift;      // log2(dma_rows)
    int cpr;            // chunks per revolution of the ring = ceil(dma_rows * rbytes / 1024)
    int nwaves;         // compute waves: 64 column pairs each
    // periodization (PER = 1, round 6): the DMA rings are ROWS of `rbytes` (a pitch, 16-byte multiple) - [.. | WARM halo cells | Kw coefficients],
    // the coefficients at byte per_d0 (16-byte aligned), in front of them copies of the row's LAST WARM coefficients (the wrap of the
    // reference's periodized synthesis, dwt/lowlevel.py:252-261) - row k of the level's feed numbering at (k & (dma_rows - 1)) * rbytes;
    // an LDS-fed low-pass ring row holds the level above's output in the order it was produced (z order: x column m = z column m + WARM),
    // WARM copies of its first cells behind it, and head_off = WARM more rows: copies of the ring's first WARM rows, which the last WARM feeds read
    int per_d0, head_off;
    int per_grp;        // G: ring rows one DMA instruction fills (a power of two; G * rbytes <= 1024 - short rows, several to an instruction; 1: row by row)
};

struct WlIRowsSeg {
    int nhb;
    int f0[WL_IROWS_MAXLEV], fend[WL_IROWS_MAXLEV];   // feeds [f0, fend) of each level
    int rho[WL_IROWS_MAXLEV];                         // PER: the segment's frame is the periodic plane ROTATED by rho[j] coefficient rows of level j
                                                      // (feed i of level j reads coefficient row (i - WARM + rho[j]) mod Kh): both halves of a cut
                                                      // plane are a segment that starts at feed 0 and never meets its own wrap
    int own_lo, own_hi;                               // rows of x this segment stores (PER: z rows of the rotated frame)
    unsigned sched[WL_IROWS_MAXHB / 4];               // per half-batch: bits 2j+1:2j = feeds of level j
};

template <typename T>
struct WlIRowsArgs {
    const T* yl;                   // coarsest low-pass, (NC, Kh, Kw) planes ll_ps apart, rows dense
    const T* yh[WL_IROWS_MAXLEV];  // (NC, 3, Kh_j, Kw_j) dense, j = 0 finest
    T* y;                          // (NC, OH_0, OW_0) dense
    const float* g_w_lo;
    const float* g_w_hi;
    const float* g_h_lo;
    const float* g_h_hi;
    int64_t NC, ll_ps;
    int nlev, nwhole, lds_bytes;
    // role of every wave: role_level >= 0: compute wave of that level, role_arg = first column pair;
    //                     role_level == -1: loader, role_arg = level * 16 + first_source * 4 + nsources;  -2: spare
    signed char role_level[WL_IROWS_WAVES];
    short role_arg[WL_IROWS_WAVES];
    // narrow planes: a workgroup owns pp consecutive planes, every one with its own compute and loader waves and its own
    // lds_plane bytes of rings (role_sub = which of them a wave serves); the schedule is the same for all of them
    signed char role_sub[WL_IROWS_WAVES];
    int pp, lds_plane;
    WlIRowsLevel g[WL_IROWS_MAXLEV];
    WlIRowsSeg seg[3];             // 0: whole plane, 1: top half, 2: bottom half
    int per;                       // periodization (the PER instantiations)
    int guard;                     // tap-relation guard (wl_common.h) of the lattice variant (1) and its armed four-bank fallback (2); 0 = no check
    const float* lat;              // WlTapPrep's verdict (+ the column lattice for LAT = 1) in device scratch (wl_lattice.h)
};

// Host-side schedule of one segment (also documents the rules the table encodes).
struct WlIRowsSched {
    int fed[WL_IROWS_MAXLEV];
    WL_HD void init(const WlIRowsSeg& sg) { for (int j = 0; j < WL_IROWS_MAXLEV; ++j) fed[j] = sg.f0[j]; }
    // output rows [2 f0, made(j)) of level j have been produced
    WL_HD int made(const WlIRowsSeg& sg, int j, int LT) const {
        const int warm = (LT - 2) / 2;
        return fed[j] - sg.f0[j] > warm ? 2 * (fed[j] - warm) : 2 * sg.f0[j];
    }
    template <typename A> WL_HD int feeds_now(const A& a, const WlIRowsSeg& sg, int j, int LT) const {
        const int warm = (LT - 2) / 2;
        int n = 0;
        while (n < 2 && fed[j] + n < sg.fend[j]) {
            const int k = fed[j] + n;
            // its LL row must exist (the coarsest level's arrives by DMA like the bands) ...
            // (periodization: feeds Kh .. Kh + WARM - 1 read the head copies of the ring's first rows, made long ago)
            if (j + 1 < a.nlev && k >= made(sg, j + 1, LT) && !(a.per && k >= a.g[j].Kh)) break;
            // ... and its two output rows must fit into the ring of the level below without evicting an unread row
            if (j > 0 && k - sg.f0[j] >= warm && 2 * (k - warm) + 1 - a.g[j - 1].ll_rows >= fed[j - 1]) break;
            ++n;
        }
        return n;
    }
};

// LAT = 1 (round 5): ONE orthogonal bank for both axes - the row synthesis in the quadrature-mirror form (the lowpass pairs only),
// the column synthesis as the transposed lattice of wl_lattice.h: L packed FMAs and L/2 - 1 delayed values per coefficient row
// where the direct form needs 2L FMAs, two (L/2 - 1)-row windows and four banks of tap pairs in scalar registers (at 16 taps
// more than the scalar file holds: the direct form stops at 12).
#ifndef WL_IROWS_LAT_MIN
#define WL_IROWS_LAT_MIN 8      // tap counts from which the lattice variant of the fused synthesis exists.  8 = the metric's db4: same-box
                                // A/B of the inverse 0.1887-0.1899 -> 0.1768-0.1782 ms (-6 %) WITH the examination kernel and the armed
                                // fallback in the figure (the analysis kernel at 8 taps measured +2 %: its lattice starts at 10)
#endif
// PER = 1 (round 6): periodization.  y[n] = sum_k c[k mod N] g[n - 2k] (the reference's transposed convolution + ONE wrap-add, for a level of
// at least L/2 coefficients) and x[m] = y[(m + L/2 - 1) mod 2N] (its roll): the polyphase arithmetic of the other modes on coefficient rows /
// columns that carry L/2 - 1 wrapped coefficients in front, all 2N outputs kept, stored L/2 - 1 places earlier (address arithmetic).  The roll
// makes the order in which a level PRODUCES its rows the order in which the level below CONSUMES them - the rows it needs first are the
// producer's first - so nothing is recomputed: the low-pass rings keep a copy of their first L/2 - 1 rows and cells (WlIRowsLevel).
template <typename T, int LT, int LAT = 0, int PER = 0>
struct WlSfbRows {
    typedef WlIRowsArgs<T> Args;
    static const int kThreads = 64 * WL_IROWS_WAVES;
    static const int kMinWaves = 7;        // two workgroups per CU: 28 waves on 4 SIMDs
    static const int HL = LT / 2;          // taps per polyphase component = rows of the window
    static const int WARM = HL - 1;        // feeds that only fill the window
    static const int SZ = (int)sizeof(T);

    // ---- loader wave: sources [s0, s0+ns) of level j ---------------------------------------------------------------
    // A "step" = the next 1024-byte chunk of every source (one DMA instruction each, plus dword pieces where a plane ends
    // off a 16-byte boundary).  A step may go out once the rows it overwrites (the previous revolution's rows under the
    // same ring bytes) were consumed at least one half-batch ago; before a half-batch the wave waits until the steps
    // under the rows about to be consumed have landed.
    static WL_DEV void loader(const Args& a, const WlIRowsSeg& sg, const WlCtx& ctx, int64_t plane, int lane, int j, int s0, int ns, int soff) {
        const WlIRowsLevel& g = a.g[j];
        const int rb = g.rbytes, R = g.dma_rows, cpr = g.cpr;
        const int plane_bytes = g.Kh * rb, ring_bytes = R * rb;
        const int f0 = sg.f0[j], fend = sg.fend[j];
        const int r0 = f0 >> g.dma_shift;                // the segment starts with the revolution that holds its first row
        // everything a step needs sits in scalar registers: a load from the argument block per chunk would stall the wave
        // for longer than the chunk takes
        const char* src[4];
        int dst[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int s = s0 + (i < ns ? i : 0);
            src[i] = s == 0 ? reinterpret_cast<const char*>(a.yl + (size_t)plane * a.ll_ps)
                            : reinterpret_cast<const char*>(a.yh[j] + ((size_t)plane * 3 + (s - 1)) * ((size_t)g.Kh * g.Kw));
            dst[i] = g.src_off[s] + soff;
        }
        // steps (counted from the segment's first) under rows [.., k]
        auto need = [&](int k) { return ((k >> g.dma_shift) - r0) * cpr + (((k & (R - 1)) + 1) * rb - 1) / WL_IROWS_CHUNK + 1; };
        const int total = fend > f0 ? need(fend - 1) : 0;
        const int max_steps = WL_IROWS_MAX_VM / ns - 1;  // in flight at once
        const int p16 = plane_bytes & ~15, ptail = (plane_bytes & 15) / 4;   // whole 16-byte pieces; dwords behind them
        int issued = 0, landed = 0;                      // steps issued / known to have landed
        int r = r0, c = 0;                               // next step: revolution r, chunk c ...
        int rbyte = 0, pbyte = r0 * ring_bytes;          // ... = ring bytes [rbyte, +1024), plane bytes [pbyte, +1024)
        int fed = f0;
        auto try_issue = [&]() {
            if (issued >= total || issued - landed >= max_steps) return false;
            const int cend = rbyte + WL_IROWS_CHUNK < ring_bytes ? rbyte + WL_IROWS_CHUNK : ring_bytes;
            // the ring bytes it overwrites held rows of the previous revolution: all of them consumed, and not in the
            // half-batch that is running now (rows >= fed - 2)
            if (r != r0 && cend > (fed - 2 - (r - 1) * R) * rb) return false;
            if (!(WL_IROWS_ABLATE & 2)) {
                const bool whole = pbyte + 16 <= plane_bytes;                 // at least lane 0 has a whole piece
                const bool on = rbyte + lane * 16 < ring_bytes && pbyte + lane * 16 + 16 <= plane_bytes;
                const bool tail = ptail && p16 >= pbyte && p16 < pbyte + (cend - rbyte);   // the plane's last dwords are in this step
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i >= ns) break;
                    if (whole) wl_dma16_s(ctx, (unsigned)(dst[i] + rbyte), src[i], (unsigned)(pbyte + lane * 16), on);
                    if (tail) wl_dma4_s(ctx, (unsigned)(dst[i] + rbyte + (p16 - pbyte)), src[i], (unsigned)(p16 + lane * 4), lane < ptail);
                }
            }
            ++issued;
            rbyte += WL_IROWS_CHUNK; pbyte += WL_IROWS_CHUNK;
            if (++c == cpr) { c = 0; ++r; rbyte = 0; pbyte = r * ring_bytes; }
            return true;
        };
        while (try_issue()) {}
        unsigned long long tw = 0, tb = 0, ti = 0;
        unsigned word = sg.sched[0];
        for (int hb = 0; hb < sg.nhb; ++hb) {
            const unsigned long long c0 = WL_ITICK();
            const int n = (int)(word >> (8 * (hb & 3) + 2 * j)) & 3;
            if ((hb & 3) == 3) word = sg.sched[(hb >> 2) + 1 < WL_IROWS_MAXHB / 4 ? (hb >> 2) + 1 : 0];
            if (n) {   // the steps under rows fed .. fed+n-1 have landed
                const int nd = need(fed + n - 1);
                if (issued < nd) wl_fail();   // the launcher's geometry checks rule this out
                if (!(WL_IROWS_ABLATE & 2)) wl_wait_vm_dyn((issued - nd) * ns);
                landed = nd;
            }
            const unsigned long long c1 = WL_ITICK();
            if (!(WL_IROWS_ABLATE & 64) || !(hb & 1)) ctx.sync();
            const unsigned long long c2 = WL_ITICK();
            fed += n;
            while (try_issue()) {}
            tw += c1 - c0; tb += c2 - c1; ti += WL_ITICK() - c2;
        }
        wl_wait_vm<0>();   // nothing may land after the workgroup has released its LDS
        if ((WL_IROWS_ABLATE & 8) && lane == 0) {
            T* o = a.y + (size_t)plane * a.g[0].OH * a.g[0].OW + 4 * (ctx.tid >> 6);
            o[0] = (T)(float)(tw >> 6); o[1] = (T)(float)(tb >> 6); o[2] = (T)(float)(ti >> 6);
        }
    }

    // ---- loader wave, periodization: one coefficient ROW of every source per step, into ring rows with the wrapped cells in front ------
    static WL_DEV void loader_per(const Args& a, const WlIRowsSeg& sg, const WlCtx& ctx, int64_t plane, int lane, int j, int s0, int ns, int soff) {
        const WlIRowsLevel& g = a.g[j];
        const int P = g.rbytes, R = g.dma_rows, Kh = g.Kh, rowb = g.Kw * SZ;
        const int rowb16 = rowb & ~15, tail = (rowb & 15) / 4;
        const int nseg = (rowb16 + WL_IROWS_CHUNK - 1) / WL_IROWS_CHUNK;      // dwordx4 instructions per row and source
        const int per_row = ns * (nseg + (tail ? 1 : 0));                        // DMA instructions of one step
        const int f0 = sg.f0[j], fend = sg.fend[j], rho = sg.rho[j];
        const char* src[4];
        int dst[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int s = s0 + (i < ns ? i : 0);
            src[i] = s == 0 ? reinterpret_cast<const char*>(a.yl + (size_t)plane * a.ll_ps)
                            : reinterpret_cast<const char*>(a.yh[j] + ((size_t)plane * 3 + (s - 1)) * ((size_t)g.Kh * g.Kw));
            dst[i] = g.src_off[s] + soff;
        }
        // Short rows: G ring rows per DMA instruction (an instruction costs the CU the same ~64 cycles whatever it moves: row by row, a 224-column
        // plane issued three times the instructions of the flat images of the other modes).  The instruction covers the LDS bytes of G consecutive
        // ring rows; a lane's 16 bytes lie in ring row lrg at data byte lo - or in a row's halo cells / pitch padding: that lane is off.
        const int G = g.per_grp;
        const int per_step = G > 1 ? ns : per_row;            // DMA instructions of one step (G rows of every source)
        const int lrg = (16 * lane) / P, lo = 16 * lane - lrg * P - g.per_d0;
        const bool lon = lrg < G && lo >= 0 && lo + 16 <= rowb;
        const int fend_g = (fend + G - 1) & ~(G - 1);         // (the last group may reach past the segment's feeds: rows nobody reads)
        const int max_rows = (WL_IROWS_MAX_VM / per_step - 1) * G;   // rows in flight at once
        int next = f0, landed = f0;                           // next row to issue / rows [f0, landed) known to have landed
        int fed = f0;
        auto try_issue = [&]() {
            if (next >= fend_g || next - landed >= max_rows) return false;
            if (next + G - 1 - R >= fed - 2) return false;    // the ring rows it overwrites: consumed, and not in the half-batch that is running now
            if (G > 1) {
                int sr = next + lrg - WARM + rho;             // this lane's coefficient row, wrapped into the plane
                sr = sr < 0 ? sr + Kh : sr;
                sr = sr >= Kh ? sr - Kh : sr;
                sr = sr >= Kh ? sr - Kh : sr;
                const unsigned gro = (unsigned)sr * (unsigned)rowb + (unsigned)lo;
                const unsigned slot = (unsigned)((next & (R - 1)) * P);
                if (!(WL_IROWS_ABLATE & 2)) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (i >= ns) break;
                        wl_dma16_s(ctx, (unsigned)dst[i] + slot, src[i], gro, lon);
                    }
                }
                next += G;
                return true;
            }
            int sr = next - WARM + rho;                       // the coefficient row of feed `next`, wrapped into the plane
            sr = sr < 0 ? sr + Kh : sr;
            sr = sr >= Kh ? sr - Kh : sr;
            const unsigned gro = (unsigned)sr * (unsigned)rowb;
            const unsigned slot = (unsigned)((next & (R - 1)) * P + g.per_d0);
            if (!(WL_IROWS_ABLATE & 2)) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i >= ns) break;
                    for (int q = 0; q < nseg; ++q) {
                        const int byte = (q * 64 + lane) * 16;
                        wl_dma16_s(ctx, (unsigned)dst[i] + slot + (unsigned)(q * WL_IROWS_CHUNK), src[i], gro + (unsigned)byte, byte + 16 <= rowb);
                    }
                    if (tail) wl_dma4_s(ctx, (unsigned)dst[i] + slot + (unsigned)rowb16, src[i], gro + (unsigned)(rowb16 + lane * 4), lane < tail);
                }
            }
            ++next;
            return true;
        };
        while (try_issue()) {}
        unsigned word = sg.sched[0];
        for (int hb = 0; hb < sg.nhb; ++hb) {
            const int n = (int)(word >> (8 * (hb & 3) + 2 * j)) & 3;
            if ((hb & 3) == 3) word = sg.sched[(hb >> 2) + 1 < WL_IROWS_MAXHB / 4 ? (hb >> 2) + 1 : 0];
            if (n) {   // rows fed .. fed+n-1 have landed; their wrapped cells: copies of the row's last WARM coefficients
                const int nd = (fed + n + G - 1) & ~(G - 1);   // (whole groups)
                if (next < nd) wl_fail();   // the launcher's geometry checks rule this out
                if (!(WL_IROWS_ABLATE & 2)) wl_wait_vm_dyn(G > 1 ? ((next - nd) / G) * per_step : (next - nd) * per_step);
                landed = nd;
                if (WARM > 0) {
                    for (int it = lane; it < n * ns * WARM; it += 64) {
                        const int h = it % WARM, rs = it / WARM, i = rs % ns, rr = rs / ns;
                        char* row = ctx.smem + (dst[i < 4 ? i : 0] + ((fed + rr) & (R - 1)) * P + g.per_d0);
                        *reinterpret_cast<T*>(row + (h - WARM) * SZ) = *reinterpret_cast<const T*>(row + (g.Kw - WARM + h) * SZ);
                    }
                }
            }
            ctx.sync();
            fed += n;
            while (try_issue()) {}
        }
        wl_wait_vm<0>();   // nothing may land after the workgroup has released its LDS
    }

    // ---- compute waves of level j -------------------------------------------------------------------------------------
    struct __attribute__((packed, aligned(sizeof(T)), may_alias)) WlPairT { T a, b; };
    static const int NP = (HL + 1) / 2;        // coefficient pairs a lane reads per band row
    static const int NW = HL > 1 ? HL - 1 : 1; // rows of history in the window
    struct Wave {                              // per-wave / per-lane constants of a compute wave
        wl_v2 gw0[HL], gw1[LAT ? 1 : HL], gh0[LAT ? 1 : HL], gh1[LAT ? 1 : HL];   // (g[2t], g[2t+1]) tap pairs, wave-uniform (LAT: gw0 g only)
        wl_v2 lt[LAT ? HL : 1];                // LAT: (T_k, -T_k) of the column lattice
        int coff;                              // byte offset of this lane's first coefficient in a ring row
        int coff_ll;                           // PER: the same in the low-pass source's row (an LDS-fed ring starts at its cell 0, a DMA ring at its halo)
        int ycol1;                             // PER, level 0: byte offset of this lane's SECOND output column in a row of x (the roll may wrap between the two)
        int hcol0, hcol1;                      // PER, levels > 0: the wrapped cells behind the ring row this lane is the source of (or -1)
        int rot2, xoh;                         // PER, level 0: 2 rho - WARM and OH: x row of z row m = (m + rot2) mod OH
        int lhead;                             // PER, levels > 0: LDS offset of the head rows of the ring this level writes
        int llmask;                            // row mask of this level's low-pass source ring
        bool two;                              // the second column of the pair exists (odd widths: not for the last pair)
        bool odd_wave;                         // wave-uniform: some lane of this wave has !two
        char* yp; unsigned yrowb, ycol;        // level 0: x
        int lring, lpitch, lmask, lrows, lcol; // levels > 0: the low-pass ring of the level below
    };

    // the HL coefficients c .. c+HL-1 of one ring row as (even, odd) pairs: one ds_read2 / ds_read_b64 per pair, and the
    // packed FMAs pick the half they need (op_sel) - no register shuffling
    static WL_DEV void load_coeffs(const char* p, wl_v2 (&v)[NP]) {
#pragma unroll
        for (int i = 0; i < HL / 2; ++i)
            v[i] = wl_v2{(float)*reinterpret_cast<const T*>(p + 2 * i * SZ), (float)*reinterpret_cast<const T*>(p + (2 * i + 1) * SZ)};
        if (HL & 1) v[NP - 1] = wl_v2{(float)*reinterpret_cast<const T*>(p + (HL - 1) * SZ), 0.f};
    }
    // polyphase row synthesis of coefficient row k for this lane's column pair: a = (ll, hl) along W (the H-low
    // intermediate), b = (lh, hh) (the H-high one); tap pair t meets coefficient c + HL-1 - t
    static WL_DEV void row_syn(const WlIRowsLevel& g, const Wave& R, const char* smem, int k, wl_v2& na, wl_v2& nb) {
        // (all scalar already: no readfirstlane here - it would drag the producers of its operand into vector registers)
        int rll = g.src_off[0] + (k & R.llmask) * g.ll_pitch;
        if (PER && g.head_off >= 0 && k >= g.Kh) rll = g.head_off + (k - g.Kh) * g.ll_pitch;   // (wave-uniform: the last WARM feeds of a whole plane)
        const int rb = (k & (g.dma_rows - 1)) * g.rbytes;
        wl_v2 vll[NP], vlh[NP], vhl[NP], vhh[NP];
        load_coeffs(smem + (rll + (PER ? R.coff_ll : R.coff)), vll);
        load_coeffs(smem + (g.src_off[1] + rb + R.coff), vlh);
        load_coeffs(smem + (g.src_off[2] + rb + R.coff), vhl);
        load_coeffs(smem + (g.src_off[3] + rb + R.coff), vhh);
        na = wl_pk_mul_x(R.gw0[HL - 1], vll[0]);
        nb = wl_pk_mul_x(R.gw0[HL - 1], vlh[0]);
        if constexpr (LAT != 0) {   // the highpass pair of tap pair t = q(lowpass pair HL-1-t): the lowpass bank alone
            wl_qmf_syn_fma<0>(na, R.gw0[0], vhl[0]);
            wl_qmf_syn_fma<0>(nb, R.gw0[0], vhh[0]);
#pragma unroll
            for (int u = 1; u < HL; ++u) {
                const int t = HL - 1 - u;
                if (u & 1) {
                    wl_pk_fma_y(na, R.gw0[t], vll[u / 2]); wl_pk_fma_y(nb, R.gw0[t], vlh[u / 2]);
                    wl_qmf_syn_fma<1>(na, R.gw0[u], vhl[u / 2]); wl_qmf_syn_fma<1>(nb, R.gw0[u], vhh[u / 2]);
                } else {
                    wl_pk_fma_x(na, R.gw0[t], vll[u / 2]); wl_pk_fma_x(nb, R.gw0[t], vlh[u / 2]);
                    wl_qmf_syn_fma<0>(na, R.gw0[u], vhl[u / 2]); wl_qmf_syn_fma<0>(nb, R.gw0[u], vhh[u / 2]);
                }
            }
            return;
        }
        wl_pk_fma_x(na, R.gw1[LAT ? 0 : HL - 1], vhl[0]);
        wl_pk_fma_x(nb, R.gw1[LAT ? 0 : HL - 1], vhh[0]);
#pragma unroll
        for (int u = 1; u < HL; ++u) {
            const int t = LAT ? 0 : HL - 1 - u;
            if (u & 1) {
                wl_pk_fma_y(na, R.gw0[t], vll[u / 2]); wl_pk_fma_y(nb, R.gw0[t], vlh[u / 2]);
                wl_pk_fma_y(na, R.gw1[t], vhl[u / 2]); wl_pk_fma_y(nb, R.gw1[t], vhh[u / 2]);
            } else {
                wl_pk_fma_x(na, R.gw0[t], vll[u / 2]); wl_pk_fma_x(nb, R.gw0[t], vlh[u / 2]);
                wl_pk_fma_x(na, R.gw1[t], vhl[u / 2]); wl_pk_fma_x(nb, R.gw1[t], vhh[u / 2]);
            }
        }
    }
    // one coefficient row of the column lattice (wl_lattice.h, the transposed recurrence): (a, b) = the row-synthesised lowpass /
    // highpass row, packed over the lane's two columns -> y0 = (row m, row m+1) of column n, y1 of column n+1; the K - 1 delayed
    // values (S = the wa window) move along by one stage
    static WL_DEV void lat_feed(const Wave& R, wl_v2 (&S)[NW], wl_v2 a, wl_v2 b, wl_v2& y0, wl_v2& y1) {
        wl_v2 q = wl_fma_s<0, 1>(a, R.lt[HL - 1], b);              // T a - b
        wl_v2 p = wl_fma_s<0, 0>(b, R.lt[HL - 1], a);              // a + T b
#pragma unroll
        for (int k = HL - 1; k >= 1; --k) {
            const wl_v2 d = S[k - 1];
            S[k - 1] = q;
            const wl_v2 n = wl_fma_s<1, 0>(d, R.lt[k - 1], p);     // p - T q'
            q = wl_fma_s<0, 0>(p, R.lt[k - 1], d);                 // q' + T p
            p = n;
        }
        y0 = wl_v2{p.x, q.x}; y1 = wl_v2{p.y, q.y};                // (row m, row m+1) per column: a renaming of registers
    }
    // polyphase column synthesis from HL consecutive window rows e[OFF .. OFF+HL-1] (oldest first): y0 = (row m, row m+1)
    // of column n (.x of the window pairs), y1 of column n+1 (.y)
    template <int OFF, int NE>
    static WL_DEV void col_syn(const Wave& R, const wl_v2 (&ea)[NE], const wl_v2 (&eb)[NE], wl_v2& y0, wl_v2& y1) {
        y0 = wl_pk_mul_x(R.gh0[0], ea[OFF + HL - 1]); y1 = wl_pk_mul_y(R.gh0[0], ea[OFF + HL - 1]);
        wl_pk_fma_x(y0, R.gh1[0], eb[OFF + HL - 1]);
        wl_pk_fma_y(y1, R.gh1[0], eb[OFF + HL - 1]);
#pragma unroll
        for (int t = 1; t < HL; ++t) {
            wl_pk_fma_x(y0, R.gh0[LAT ? 0 : t], ea[OFF + HL - 1 - t]);
            wl_pk_fma_y(y1, R.gh0[LAT ? 0 : t], ea[OFF + HL - 1 - t]);
            wl_pk_fma_x(y0, R.gh1[LAT ? 0 : t], eb[OFF + HL - 1 - t]);
            wl_pk_fma_y(y1, R.gh1[LAT ? 0 : t], eb[OFF + HL - 1 - t]);
        }
    }
    // periodization: z rows m, m+1 (y0 = the two rows of z column 2q, y1 of column 2q+1) - level 0: to x, L/2 - 1 rows / columns earlier
    // (mod the plane, and by the segment's rotation); other levels: into the ring of the level below in z order, with the copies it wraps to
    template <int j>
    static WL_DEV void emit_per(const Wave& R, char* smem, int m, wl_v2 y0, wl_v2 y1) {
        if (j == 0) {
            if ((WL_IROWS_ABLATE & 1) && y0.x + y0.y + y1.x + y1.y != 1.2345e30f) return;
            int r0 = m + R.rot2;
            r0 = r0 < 0 ? r0 + R.xoh : r0;
            r0 = r0 >= R.xoh ? r0 - R.xoh : r0;
            int r1 = r0 + 1;
            r1 = r1 >= R.xoh ? r1 - R.xoh : r1;
            char* p0 = R.yp + (unsigned)r0 * R.yrowb;
            char* p1 = R.yp + (unsigned)r1 * R.yrowb;
            if (!R.odd_wave) {
                T v[2] = {(T)y0.x, (T)y1.x}, w[2] = {(T)y0.y, (T)y1.y};
                *reinterpret_cast<WlPairT*>(p0 + R.ycol) = *reinterpret_cast<WlPairT*>(v);
                *reinterpret_cast<WlPairT*>(p1 + R.ycol) = *reinterpret_cast<WlPairT*>(w);
            } else {   // the wave that holds the pair the roll splits between the row's last and first column
                *reinterpret_cast<T*>(p0 + R.ycol) = (T)y0.x; *reinterpret_cast<T*>(p0 + R.ycol1) = (T)y1.x;
                *reinterpret_cast<T*>(p1 + R.ycol) = (T)y0.y; *reinterpret_cast<T*>(p1 + R.ycol1) = (T)y1.y;
            }
        } else {
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int row = m + rr;
                if (row >= R.lrows) continue;
                const T va = (T)(rr ? y0.y : y0.x), vb = (T)(rr ? y1.y : y1.x);
                char* d = smem + (R.lring + (row & R.lmask) * R.lpitch);
                T v[2] = {va, vb};
                *reinterpret_cast<WlPairT*>(d + R.lcol) = *reinterpret_cast<WlPairT*>(v);
                if (R.hcol0 >= 0) *reinterpret_cast<T*>(d + R.hcol0) = va;
                if (R.hcol1 >= 0) *reinterpret_cast<T*>(d + R.hcol1) = vb;
                if (row < WARM) {   // (wave-uniform) the ring's first rows, kept for the feeds that wrap around the plane's bottom
                    char* h = smem + (R.lhead + row * R.lpitch);
                    *reinterpret_cast<WlPairT*>(h + R.lcol) = *reinterpret_cast<WlPairT*>(v);
                    if (R.hcol0 >= 0) *reinterpret_cast<T*>(h + R.hcol0) = va;
                    if (R.hcol1 >= 0) *reinterpret_cast<T*>(h + R.hcol1) = vb;
                }
            }
        }
    }
    // output rows m, m+1: to x (8 contiguous bytes per lane and row), or into the low-pass ring of the level below
    template <int j>
    static WL_DEV void emit(const Wave& R, char* smem, int m, wl_v2 y0, wl_v2 y1) {
        if constexpr (PER != 0) { emit_per<j>(R, smem, m, y0, y1); return; }
        if (j == 0) {
            // every row a level-0 feed produces is one this segment owns: its feeds start at own_lo / 2 and end with the
            // feed that makes row own_hi - 1
            if ((WL_IROWS_ABLATE & 1) && y0.x + y0.y + y1.x + y1.y != 1.2345e30f) return;   // keeps the arithmetic alive
            char* r0 = R.yp + ((unsigned)m * R.yrowb + R.ycol);
            char* r1 = r0 + R.yrowb;
            if (!R.odd_wave) {
#if (WL_STREAM_NT & 2) && defined(__HIPCC__)
                typedef T WlVec2 __attribute__((ext_vector_type(2), aligned(sizeof(T))));
                WlVec2 v = {(T)y0.x, (T)y1.x}, w = {(T)y0.y, (T)y1.y};
                __builtin_nontemporal_store(v, reinterpret_cast<WlVec2*>(r0));
                __builtin_nontemporal_store(w, reinterpret_cast<WlVec2*>(r1));
#else
                T v[2] = {(T)y0.x, (T)y1.x}, w[2] = {(T)y0.y, (T)y1.y};
                *reinterpret_cast<WlPairT*>(r0) = *reinterpret_cast<WlPairT*>(v);
                *reinterpret_cast<WlPairT*>(r1) = *reinterpret_cast<WlPairT*>(w);
#endif
            } else {   // the wave that holds the last pair of an odd-width row: that lane has one column only
                *reinterpret_cast<T*>(r0) = (T)y0.x;
                *reinterpret_cast<T*>(r1) = (T)y0.y;
                if (R.two) { *reinterpret_cast<T*>(r0 + SZ) = (T)y1.x; *reinterpret_cast<T*>(r1 + SZ) = (T)y1.y; }
            }
        } else {
            if (m < R.lrows) {
                T v[2] = {(T)y0.x, (T)y1.x};
                *reinterpret_cast<WlPairT*>(smem + (R.lring + (m & R.lmask) * R.lpitch + R.lcol)) = *reinterpret_cast<WlPairT*>(v);
            }
            if (m + 1 < R.lrows) {
                T v[2] = {(T)y0.y, (T)y1.y};
                *reinterpret_cast<WlPairT*>(smem + (R.lring + ((m + 1) & R.lmask) * R.lpitch + R.lcol)) = *reinterpret_cast<WlPairT*>(v);
            }
        }
    }

    template <int j>
    static WL_DEV void compute(const Args& a, const WlIRowsSeg& sg, const WlCtx& ctx, int64_t plane, int c0, int lane, int soff) {
        const WlIRowsLevel& g = a.g[j];
        char* const smem = ctx.smem + soff;
        const int c = c0 + lane;                         // column pair: output columns 2c, 2c+1
        const bool active = 2 * c < g.OW;
        Wave R;
        if constexpr (LAT != 0) {
            const float gsc = a.lat[1];                       // the lattice's gain rides on the row-synthesis taps
#pragma unroll
            for (int t = 0; t < HL; ++t) {
                R.gw0[t] = wl_uniform_v2(wl_v2{a.g_w_lo[2 * t] * gsc, a.g_w_lo[2 * t + 1] * gsc});
                R.lt[t] = wl_uniform_v2(wl_v2{a.lat[2 + t], -a.lat[2 + t]});
            }
        } else {
#pragma unroll
            for (int t = 0; t < HL; ++t) {
                R.gw0[t] = wl_uniform_v2(wl_v2{a.g_w_lo[2 * t], a.g_w_lo[2 * t + 1]});
                R.gw1[t] = wl_uniform_v2(wl_v2{a.g_w_hi[2 * t], a.g_w_hi[2 * t + 1]});
                R.gh0[t] = wl_uniform_v2(wl_v2{a.g_h_lo[2 * t], a.g_h_lo[2 * t + 1]});
                R.gh1[t] = wl_uniform_v2(wl_v2{a.g_h_hi[2 * t], a.g_h_hi[2 * t + 1]});
            }
        }
        R.coff = (active ? c : 0) * SZ;
        R.llmask = g.ll_rows - 1;
        R.two = 2 * c + 1 < g.OW;
        R.odd_wave = wl_uniform((g.OW & 1) && 2 * (c0 + 63) + 1 >= g.OW) != 0;
        R.yp = reinterpret_cast<char*>(a.y + (size_t)plane * g.OH * g.OW);
        R.yrowb = (unsigned)g.OW * SZ; R.ycol = (unsigned)(2 * c) * SZ;
        R.lmask = j > 0 ? a.g[j > 0 ? j - 1 : 0].ll_rows - 1 : 0;
        R.lring = j > 0 ? a.g[j > 0 ? j - 1 : 0].src_off[0] : 0;
        R.lpitch = j > 0 ? a.g[j > 0 ? j - 1 : 0].ll_pitch : 0;
        R.lrows = j > 0 ? a.g[j > 0 ? j - 1 : 0].Kh : 0;   // rows the level below reads ('unpad': it may drop the last one)
        R.lcol = 2 * c * SZ;
        R.coff_ll = R.coff; R.ycol1 = 0; R.hcol0 = R.hcol1 = -1; R.rot2 = 0; R.xoh = g.OH; R.lhead = 0;
        if constexpr (PER != 0) {
            const int q = active ? c : 0;
            R.coff = g.per_d0 - WARM * SZ + q * SZ;
            R.coff_ll = j + 1 < a.nlev ? q * SZ : R.coff;
            if (j == 0) {
                int x0 = 2 * q - WARM;
                x0 = x0 < 0 ? x0 + g.OW : x0;
                int x1 = x0 + 1;
                x1 = x1 >= g.OW ? x1 - g.OW : x1;
                R.ycol = (unsigned)x0 * SZ; R.ycol1 = x1 * SZ;
                const int qs = (WARM - 1) / 2;               // the pair (2 qs, 2 qs + 1) lands on columns (OW - 1, 0) when WARM is odd
                R.odd_wave = (WARM & 1) && c0 <= qs && qs < c0 + 64;
                R.rot2 = 2 * sg.rho[0] - WARM;
            } else {
                const WlIRowsLevel& gl = a.g[j > 0 ? j - 1 : 0];
                R.lhead = gl.head_off;
                R.hcol0 = active && 2 * q < WARM ? (gl.Kw + 2 * q) * SZ : -1;
                R.hcol1 = active && 2 * q + 1 < WARM ? (gl.Kw + 2 * q + 1) * SZ : -1;
            }
        }
        // window: the row-synthesised (a, b) of the previous HL-1 coefficient rows as (col n, col n+1) pairs, oldest first
        wl_v2 wa[NW], wb[NW];
#pragma unroll
        for (int t = 0; t < NW; ++t) wa[t] = wb[t] = wl_v2{0.f, 0.f};
        const int f0 = sg.f0[j];
        int fed = f0;
        unsigned long long tb = 0, tf = 0, c2 = WL_ITICK();
        unsigned word = sg.sched[0];                     // the schedule of four half-batches; the next one is fetched early
        for (int hb = 0; hb < sg.nhb; ++hb) {
            const int n = (int)(word >> (8 * (hb & 3) + 2 * j)) & 3;
            if ((hb & 3) == 3) word = sg.sched[(hb >> 2) + 1 < WL_IROWS_MAXHB / 4 ? (hb >> 2) + 1 : 0];
            const unsigned long long c0 = WL_ITICK();
            if (!(WL_IROWS_ABLATE & 64) || !(hb & 1)) ctx.sync();
            const unsigned long long c1 = WL_ITICK();
            tf += c0 - c2; tb += c1 - c0; c2 = c1;
            if (n == 0) continue;
            const int k = fed;
            fed += n;
            if (!active || (WL_IROWS_ABLATE & 4)) continue;
            const int m = 2 * (k - WARM);
            wl_v2 y0, y1;
            if constexpr (LAT != 0) {                    // (every feed: the lattice's state; its first K - 1 outputs of a segment are the warm-up)
                wl_v2 na, nb;
                row_syn(g, R, smem, k, na, nb);
                lat_feed(R, wa, na, nb, y0, y1);
                if (k - f0 >= WARM) emit<j>(R, smem, m, y0, y1);
                if (n == 2) {
                    row_syn(g, R, smem, k + 1, na, nb);
                    lat_feed(R, wa, na, nb, y0, y1);
                    if (k + 1 - f0 >= WARM) emit<j>(R, smem, m + 2, y0, y1);
                }
                continue;
            }
            if (n == 2) {                                // two coefficient rows: one window move for both
                wl_v2 ea[HL + 1], eb[HL + 1];
                row_syn(g, R, smem, k, ea[HL - 1], eb[HL - 1]);
                row_syn(g, R, smem, k + 1, ea[HL], eb[HL]);
#pragma unroll
                for (int t = 0; t < HL - 1; ++t) { ea[t] = wa[t]; eb[t] = wb[t]; }
                if (k - f0 >= WARM) { col_syn<0>(R, ea, eb, y0, y1); emit<j>(R, smem, m, y0, y1); }
                if (k + 1 - f0 >= WARM) { col_syn<1>(R, ea, eb, y0, y1); emit<j>(R, smem, m + 2, y0, y1); }
#pragma unroll
                for (int t = 0; t < HL - 1; ++t) { wa[t] = ea[t + 2]; wb[t] = eb[t + 2]; }
            } else {
                wl_v2 ea[HL], eb[HL];
                row_syn(g, R, smem, k, ea[HL - 1], eb[HL - 1]);
#pragma unroll
                for (int t = 0; t < HL - 1; ++t) { ea[t] = wa[t]; eb[t] = wb[t]; }
                if (k - f0 >= WARM) { col_syn<0>(R, ea, eb, y0, y1); emit<j>(R, smem, m, y0, y1); }
#pragma unroll
                for (int t = 0; t < HL - 1; ++t) { wa[t] = ea[t + 1]; wb[t] = eb[t + 1]; }
            }
        }
        if ((WL_IROWS_ABLATE & 8) && lane == 0) {
            T* o = a.y + (size_t)plane * a.g[0].OH * a.g[0].OW + 4 * (ctx.tid >> 6);
            o[0] = (T)(float)(tb >> 6); o[1] = (T)(float)(tf >> 6); o[2] = (T)0;
        }
    }

    static WL_DEV void run(const Args& a, const WlCtx& ctx) {
        const int tid = ctx.tid;
        const int wave = wl_uniform(tid >> 6), lane = tid & 63;
        if (LT >= WL_IROWS_LAT_MIN && a.guard) {   // the lattice variant / its armed fallback: WlTapPrep's verdict on the banks as they are now
            const bool holds = a.lat && *reinterpret_cast<const unsigned*>(a.lat) == WL_LAT_OK;
            if (!wl_guard_pass(a.guard, holds)) return;
        }
        const int64_t bid = ctx.bid;
        const int sub = wl_uniform(a.role_sub[wave]);
        const int64_t plane = (bid < a.nwhole ? bid : a.nwhole + (bid - a.nwhole) / 2) * a.pp + sub;
        const WlIRowsSeg& sg = a.seg[bid < a.nwhole ? 0 : 1 + (int)((bid - a.nwhole) & 1)];
        const int soff = sub * a.lds_plane;
        const int lev = plane < a.NC ? wl_uniform(a.role_level[wave]) : -2, arg = wl_uniform(a.role_arg[wave]);   // (the last workgroup may hold fewer planes)
        if (lev == -1) {
#if defined(__HIPCC__)
            __builtin_amdgcn_s_setprio(3);   // its few instructions go first: every other wave waits for it at the barrier
#endif
            if constexpr (PER != 0) loader_per(a, sg, ctx, plane, lane, arg >> 4, (arg >> 2) & 3, (arg & 3) + 1, soff);
            else loader(a, sg, ctx, plane, lane, arg >> 4, (arg >> 2) & 3, (arg & 3) + 1, soff);
        }
        else if (lev == 0) compute<0>(a, sg, ctx, plane, arg, lane, soff);
        else if (lev == 1) compute<1>(a, sg, ctx, plane, arg, lane, soff);
        else if (lev == 2) compute<2>(a, sg, ctx, plane, arg, lane, soff);
        else
            for (int hb = 0; hb < sg.nhb; ++hb) if (!(WL_IROWS_ABLATE & 64) || !(hb & 1)) ctx.sync();   // spare wave: keeps the barrier count
    }
};
