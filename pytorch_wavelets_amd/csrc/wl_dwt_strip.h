// Streaming ONE-level 2-D DWT analysis over column strips: the streaming design of wl_dwt_rows.h for everything its
// fused multi-level form cannot take - rows wider than one workgroup's lanes (1024, 2048, 4096 columns), 14 to 20 taps,
// periodization (rows and columns wrap), odd filter-bank offsets, float16 planes of any width, few planes.
//
// A workgroup owns one (plane, column strip, row segment) and marches down it once:
//   * two STAGER waves bring every input row of the strip in by LDS-DMA (global_load_lds_dwordx4, 16-byte pieces, three
//     half-batches ahead, counted waits) into a private ring in the input's own type.  A piece is a run of 16 bytes of
//     the EXTENDED row: for the wrapping modes (periodic, periodization) the pieces beyond either end of the row simply
//     come from the other end - a row is a whole number of pieces, so wrapping is per-piece address arithmetic, resolved
//     once per lane - and a strip in the middle of the row gets its halo columns from its neighbours' columns;
//   * the same waves then STAGE the rows they loaded: float32 copies (the float16 -> float32 conversion happens here,
//     once per sample instead of once per tap) into a two-slot ring whose origin is chosen so that every compute lane
//     reads its samples as whole, conflict-free 16-byte words; the mirrored cells of symmetric / reflect extension
//     are written here too, zero padding is cells (and rows) that stay zero;
//   * a compute lane owns TWO adjacent output columns.  Per pair of new rows it reads 2 x ceil((L+2)/4) 16-byte words (its
//     L+2 samples - the two columns share L-2 of them), runs the row filter for both columns (v_pk_fma_f32 on (lo,hi)
//     tap pairs held in scalar registers, the sample picked by op_sel) into an L-row window held in registers, and the
//     column filter from the window: 4L packed FMAs per output sample pair of each sub-band quadruple and no other
//     arithmetic.  The window is a circular buffer whose rotation is compile-time: the loop over half-batches is
//     unrolled by the rotation period, so no register is ever moved;
//   * the four sub-bands leave as pairs (8 / 4 contiguous bytes per lane), whole strip rows per half-batch.
// HBM traffic = the strip's columns once + (L-2) halo columns per strip and (L-2) halo rows per segment + the outputs.
// One level per launch: LL goes to HBM (a J-level transform on this kernel moves 1.33x the bytes of the fused kernel,
// which remains the path for up to 12 taps on rows of up to ~630 outputs).
//
// Restates AFB2D.forward (reference dwt/lowlevel.py:336-347 = afb1d along W, then along H, :91-172), every mode.
#pragma once
#include "wl_common.h"
#include "wl_dwt_rows.h"   // wl_pk_fma_x / _y, wl_pk_mul_x / _y, wl_uniform_v2
#include "wl_lattice.h"

#ifndef WL_STRIP_CWAVES
#define WL_STRIP_CWAVES 4
#endif
// compute waves (two output columns per lane: up to 512 output columns per strip)
#ifndef WL_STRIP_SWAVES
#define WL_STRIP_SWAVES 4       // stager waves: each loads and stages 4 / SWAVES of the 4 rows of a half-batch (one wave alone is
                                // latency-bound: with two stagers the staging took longer than the arithmetic, with four it hides)
#endif
#ifndef WL_STRIP_ABLATE
#define WL_STRIP_ABLATE 0       // measurement builds only (tools/build_ab_strip.sh): 1 = no arithmetic / stores in the compute
#endif                          // waves, 2 = the stagers load but do not stage
#ifndef WL_STRIP_DIRECT
#define WL_STRIP_DIRECT 1        // stagers load straight into registers (0: through an LDS-DMA ring, the first version; A/B builds)
#endif
#ifndef WL_STRIP_PF
#define WL_STRIP_PF 2           // register sets of a direct stager (rows in flight + the one being staged)
#endif
#ifndef WL_STRIP_D
#define WL_STRIP_D 3            // half-batches of LDS-DMA in flight = slots of the DMA ring
#endif
#if (WL_STRIP_ABLATE & 8) && defined(__HIPCC__)
#define WL_STICK() __builtin_readcyclecounter()
#else
#define WL_STICK() 0ull
#endif
#define WL_STRIP_MAXPPR 5       // 1 KiB DMA instructions per row at most (strip rows of up to 5 KiB)

// One row, DMA ring -> staged float32 ring, by one stager wave (shared by the analysis and the synthesis strip kernels).
template <typename T>
struct WlStage {
    static const int SZ = (int)sizeof(T);
    // One row, DMA ring -> staged ring: this lane's groups i in [imin, imax) (group = 4 cells; source at srow + i * 256 SZ,
    // destination at drow + i * 1024), six at a time: the LDS reads of a batch are all in flight before the first
    // conversion (a wave alone hides no latency otherwise), reads and conversions are unconditional (a read beyond the row
    // stays inside the rings), only the writes are predicated.  DM = (e_lo - c0a) mod 4 decides how a group meets the
    // 16-byte words of the staged row: 0 one ds_write_b128, 2 two ds_write_b64, odd b32 + b64 + b32; DM 4 = a row of zeros.
    template <int DM>
    static WL_DEV void stage_row(const char* srow, char* drow, int imin, int imax, int ngl) {
        typedef T Quad4 __attribute__((ext_vector_type(4), may_alias));
        for (int i0 = 0; i0 < ngl; i0 += 6) {
            Quad4 raw[6];
            if (DM != 4) {
    #pragma unroll
                for (int u = 0; u < 6; ++u)
                    if (i0 + u < ngl) raw[u] = *reinterpret_cast<const Quad4*>(srow + (i0 + u) * 256 * SZ);
            }
    #pragma unroll
            for (int u = 0; u < 6; ++u) {
                if (i0 + u >= ngl) break;
                float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
                if (DM != 4) { v0 = (float)raw[u].x; v1 = (float)raw[u].y; v2 = (float)raw[u].z; v3 = (float)raw[u].w; }
                if (i0 + u >= imin && i0 + u < imax) {
                    float* dst = reinterpret_cast<float*>(drow + (i0 + u) * 1024);
                    if (DM == 0) {
                        wl_vf4 w; w.x = v0; w.y = v1; w.z = v2; w.w = v3;
                        *reinterpret_cast<wl_vf4*>(dst) = w;
                    } else if (DM == 2) {
                        wl_f2 w0, w1; w0.x = v0; w0.y = v1; w1.x = v2; w1.y = v3;
                        *reinterpret_cast<wl_f2*>(dst) = w0; *reinterpret_cast<wl_f2*>(dst + 2) = w1;
                    } else if (DM == 1) {                     // odd: the middle two cells are 8-byte aligned
                        wl_f2 w; w.x = v1; w.y = v2;
                        dst[0] = v0; *reinterpret_cast<wl_f2*>(dst + 1) = w; dst[3] = v3;
                    } else {                                  // zeros, any alignment
                        dst[0] = 0.f; dst[1] = 0.f; dst[2] = 0.f; dst[3] = 0.f;
                    }
                }
            }
        }
    }
};

template <typename T>
struct WlStripArgs {
    const T* x;                    // (NC, H, W) through x_ps / x_rs
    T* ll;                         // (NC, Kh, Kw) through ll_ps / ll_rs
    T* highs;                      // (NC, 3, Kh, Kw) dense
    const float* h_w_lo;
    const float* h_w_hi;
    const float* h_h_lo;
    const float* h_h_hi;
    int64_t NC, x_ps, ll_ps, nblocks;
    int x_rs, ll_rs;
    int H, W, Kh, Kw, ext, base;
    int nstrips, strip_cols;       // output columns per strip (even); the last strip may be narrower
    int nseg, seg_rows;            // output rows per segment; the last segment may be shorter
    int dma_off, dma_pitch;        // DMA ring: WL_STRIP_D slots x 4 rows x dma_pitch bytes (input type)
    int st_off, st_pitch;          // staged ring: 2 slots x 4 rows x st_pitch bytes (float32)
    int lds_bytes;
    int pp, ring;                  // planes per workgroup (1, 2, 4: narrow strips, see run()) and the bytes of one plane's staged ring
    int pair_ok;                   // every output-column pair of every band row is one aligned 2-element store (even Kw, ll_rs)
    int guard;                     // tap-relation guard (wl_common.h): 1 = run only if both highpass banks are the quadrature mirrors of
                                   // their lowpass banks (the QMF variant), 2 = only if not (its armed two-bank fallback), 0 = no check
    const float* lat;              // LAT variant and its fallback: the verdict + lattice of the column bank that WlTapPrep left in
                                   // device scratch (wl_lattice.h); the guard then reads the verdict word instead of the taps
};

// QMF = 1: the caller vouches that each highpass bank is the quadrature mirror of its lowpass bank, hi[t] = (-1)^t lo[L-1-t]
// (the decomposition pair of every orthogonal wavelet, in the order the reference stores it).  The (lo, hi) tap pair of tap t
// is then (lo[t], +-lo[L-1-t]): with the lowpass bank held as the L/2 pairs P[u] = (lo[u], lo[L-1-u]) every tap pair is P[u]
// or P[u] with its halves swapped, the high half negated for odd t - operand modifiers of the packed FMA (op_sel, neg_hi).
// Half the scalar registers: at 16 taps the two banks of rows and columns were 64 of them and the scalar file overflowed
// (25-34 spilled scalars, read back through v_readlane in the half-batch loop).
// LAT = 1 (with QMF = 1): the COLUMN pass runs the orthogonal bank as a lattice of rotations (wl_lattice.h): per pair of new
// rows and output column 2 K packed FMAs and K - 1 delayed values instead of the 2 L FMAs and the L-row window of the direct
// form - this kernel is bound by the vector instructions it issues (16 taps: 285 per half-batch, 272 of them FMAs; with the
// lattice 72 of the 136 column FMAs go).  The recurrence's coefficients and the verdict that they reproduce the bank in the
// buffers come from device scratch (a.lat, written by WlTapPrep in front of this launch).
template <typename T, int LT, int QMF = 0, int LAT = 0>
struct WlAfbStrip {
    typedef WlStripArgs<T> Args;
    static_assert(!LAT || QMF, "the lattice variant runs its row pass in the QMF form");
    static const int KL = LT / 2;                      // rotations of the lattice
    static const int NB = QMF ? LT / 2 : LT;           // tap pairs held per bank
    static const int kWaves = WL_STRIP_CWAVES + WL_STRIP_SWAVES;
    static const int kThreads = 64 * kWaves;
#ifndef WL_STRIP_LAT_MINW
#define WL_STRIP_LAT_MINW 4     // waves per SIMD the lattice variants are compiled for (6: three workgroups per CU, at most 80 registers)
#endif
    // two 8-wave workgroups per CU need four waves per SIMD: at most 128 registers (18, 20 taps: 168)
    static const int kMinWaves = (LAT && LT <= 16) ? WL_STRIP_LAT_MINW : LT >= 18 ? 3 : 4;
    static const int SZ = (int)sizeof(T);
    static const int A = 16 / SZ;          // elements per 16-byte piece
    static const int WARM = (LT - 2) / 2;  // feeds that only fill the window
    static const int LW = (LT + 3) / 4 * 4;            // window slots: a multiple of the 4 rows of a half-batch
    // half-batches after which the circular window is back where it started (lattice: the K delay slots, two feeds per half-batch)
    static const int PERIOD = LAT ? (KL % 2 ? KL : KL / 2) : LW / 4;
    static const int NV4 = (LT + 2 + 3) / 4;           // 16-byte words a lane reads per row: its L+2 samples
    static const int D = WL_STRIP_D;
    static const bool kPipe = LT >= 14;    // see compute(): when the rows of a half-batch's second feed are requested
    static const int LROWS = 4 / WL_STRIP_SWAVES;

    // geometry of one workgroup's strip, derived by every wave from the strip index (wave-uniform)
    struct Strip {
        int k0, k1;            // output columns [k0, k1)
        int e_lo;              // first extended column a lane reads = 2 k0 + base
        int c0a;               // extended column of DMA-ring cell 0 (a multiple of A, <= e_lo)
        int np;                // 16-byte pieces per DMA-ring row (up to the last one that is loaded)
        int ppr;               // DMA instructions per row
        int ng;                // 4-cell groups per DMA-ring row
        int dm;                // (e_lo - c0a) mod 4
        int lane_off;          // staged-ring byte offset of the first sample of output-column pair 0
        int o_lo, o_hi;        // output rows [o_lo, o_hi) of this segment
        int nfeeds, nhb;
    };
    static WL_HD bool wraps(int ext) { return ext == WL_EXT_PERIODIC || ext == WL_EXT_PER; }
    static WL_HD Strip geometry(const Args& a, int strip, int seg) {
        Strip s;
        s.k0 = strip * a.strip_cols;
        s.k1 = s.k0 + a.strip_cols < a.Kw ? s.k0 + a.strip_cols : a.Kw;
        s.e_lo = 2 * s.k0 + a.base;
        const int e_hi = 2 * (s.k1 - 1) + a.base + LT - 1 + 2;          // (+2: a lane always reads both its columns' samples)
        s.c0a = s.e_lo >= 0 ? s.e_lo / A * A : -((-s.e_lo + A - 1) / A * A);
        int last = e_hi;                                                 // last extended column that arrives by DMA
        if (!wraps(a.ext) && last > a.W - 1) last = a.W - 1;
        s.np = (last - s.c0a) / A + 1;
        s.ppr = (s.np + 63) / 64;
        s.ng = s.np * A / 4;
        const int d = s.e_lo - s.c0a;
        s.dm = d & 3;
        s.lane_off = (d - s.dm + 4) * 4;
        s.o_lo = seg * a.seg_rows;
        s.o_hi = s.o_lo + a.seg_rows < a.Kh ? s.o_lo + a.seg_rows : a.Kh;
        s.nfeeds = s.o_hi - s.o_lo + WARM;
        s.nhb = (s.nfeeds + 1) / 2;
        return s;
    }
    // staged-ring cell index of DMA-ring cell c (both along one row)
    static WL_HD int staged_of(const Strip& s, int c) { return c + 4 - s.dm; }

    // ---- stager wave --------------------------------------------------------------------------------------------
    static WL_DEV void stager(const Args& a, const Strip& s, const WlCtx& ctx, int64_t plane, int lane, int sidx) {
        const char* xp = reinterpret_cast<const char*>(a.x + (size_t)plane * a.x_ps);
        const int row_stride = a.x_rs * SZ;
        const bool wrap = wraps(a.ext);
        const int We = a.W;                                   // (wrapping modes need an even W: the launcher checks)
        const int e_first = a.base + 2 * s.o_lo;              // extended row of the segment's first feed
        const int e_last = a.base + 2 * (s.o_lo + s.nfeeds) - 1;
        const int rfirst = sidx * LROWS;
        // per lane: source byte (inside a row) of its piece in each of the ppr DMA instructions of a row; -1 = off
        int gbyte[WL_STRIP_MAXPPR];
#pragma unroll
        for (int q = 0; q < WL_STRIP_MAXPPR; ++q) {
            const int p = q * 64 + lane;
            int col = s.c0a + p * A;
            bool on = q < s.ppr && p < s.np;
            if (on && (unsigned)col >= (unsigned)a.W) {
                if (wrap) col = wl_pmod(col, We); else on = false;
            }
            gbyte[q] = on ? col * SZ : -1;
        }
        // mirrored cells (symmetric / reflect, strips at either edge): item = (row r of my rows, cell); a lane handles
        // items lane and lane + 64.  dst = staged cell, src = DMA-ring cell, both as byte offsets inside my rows' block.
        int hdst[2], hsrc[2];
        {
            const int e_hi = 2 * (s.k1 - 1) + a.base + LT - 1 + 2;
            const int nl = s.e_lo < 0 ? -s.e_lo : 0;                              // extended columns left of the row
            const int nr = e_hi > a.W - 1 ? e_hi - (a.W - 1) : 0;                 // and right of it
            const int NH = (a.ext == WL_EXT_SYM || a.ext == WL_EXT_REFL) ? nl + nr : 0;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int it = lane + 64 * u;
                hdst[u] = hsrc[u] = -1;
                if (it < LROWS * NH) {
                    const int r = it / NH, c = it - r * NH;
                    const int e = c < nl ? s.e_lo + c : a.W + (c - nl);
                    const int src = wl_ext(e, a.W, a.ext);
                    hdst[u] = r * a.st_pitch + staged_of(s, e - s.c0a) * 4;
                    hsrc[u] = r * a.dma_pitch + (src - s.c0a) * SZ;
                }
            }
        }
        auto issue = [&](int h) {
            const int slot = a.dma_off + (h % D) * 4 * a.dma_pitch;
#pragma unroll
            for (int rr = 0; rr < LROWS; ++rr) {
                const int r = rfirst + rr;
                int e = e_first + 4 * h + r;
                e = e < e_last ? e : e_last;
                int src = e;
                if ((unsigned)e >= (unsigned)a.H) {
                    src = wl_ext(e, a.H, a.ext);
                    src = src < 0 ? 0 : src;                  // zero rows: a dummy row keeps the DMA count exact
                }
                const char* grow = xp + (size_t)src * row_stride;
#pragma unroll
                for (int q = 0; q < WL_STRIP_MAXPPR; ++q)
                    if (q < s.ppr) wl_dma16(ctx, (unsigned)(slot + r * a.dma_pitch + q * 1024), grow + gbyte[q], gbyte[q] >= 0);
            }
        };
        const int nl_inst = LROWS * s.ppr;                    // DMA instructions per half-batch of this wave
        // the 4-cell groups of a DMA-ring row that hold loaded data: [g_lo, g_hi); this lane owns groups lane + 64 i,
        // i in [imin, imax)
        const int ngl = (s.ng + 63) >> 6;
        int imin, imax;
        {
            int g_lo = 0, g_hi = s.ng;
            if (!wrap) {
                if (s.c0a < 0) g_lo = (-s.c0a + 3) / 4;
                const int lim = (a.W - s.c0a) / 4;            // (W and c0a are multiples of 4)
                if (lim < g_hi) g_hi = lim;
            }
            imin = g_lo > lane ? (g_lo - lane + 63) / 64 : 0;
            imax = g_hi > lane ? (g_hi - lane + 63) / 64 : 0;
        }
        for (int h = 0; h < D && h < s.nhb; ++h) issue(h);
        int inflight = (D < s.nhb ? D : s.nhb);               // half-batches issued and not yet waited for
        unsigned long long tw = 0, tg = 0, tb = 0, ti = 0;
        for (int hb = 0; hb < s.nhb; ++hb) {
            const unsigned long long c0 = WL_STICK();
            wl_wait_vm_dyn((inflight - 1) * nl_inst);         // my rows of this half-batch have landed
            --inflight;
            const unsigned long long c1 = WL_STICK();
            // stage my rows: DMA slot -> staged slot (float32, origin shifted so that lanes read aligned 16-byte words)
            char* dslot = ctx.smem + a.dma_off + (hb % D) * 4 * a.dma_pitch + rfirst * a.dma_pitch;
            char* sslot = ctx.smem + a.st_off + (hb & 1) * 4 * a.st_pitch + rfirst * a.st_pitch;
            bool zrow[LROWS];
#pragma unroll
            for (int rr = 0; rr < LROWS; ++rr) {
                int e = e_first + 4 * hb + rfirst + rr;
                e = e < e_last ? e : e_last;
                zrow[rr] = a.ext == WL_EXT_ZERO && (unsigned)e >= (unsigned)a.H;
            }
            // a lane stages the 4-cell groups lane, lane + 64, .. of each of its rows (see stage_row)
            if (!(WL_STRIP_ABLATE & 2)) {
#pragma unroll
                for (int rr = 0; rr < LROWS; ++rr) {
                    const char* srow = dslot + rr * a.dma_pitch + lane * 4 * SZ;
                    char* drow = sslot + rr * a.st_pitch + lane * 16 + (4 - s.dm) * 4;
                    if (zrow[rr]) WlStage<T>::template stage_row<4>(srow, drow, imin, imax, ngl);
                    else if (s.dm == 0) WlStage<T>::template stage_row<0>(srow, drow, imin, imax, ngl);
                    else if (s.dm == 2) WlStage<T>::template stage_row<2>(srow, drow, imin, imax, ngl);
                    else WlStage<T>::template stage_row<1>(srow, drow, imin, imax, ngl);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (hdst[u] >= 0) *reinterpret_cast<float*>(sslot + hdst[u]) = (float)*reinterpret_cast<const T*>(dslot + hsrc[u]);
            const unsigned long long c2 = WL_STICK();
            ctx.sync();
            const unsigned long long c3 = WL_STICK();
            if (hb + D < s.nhb) { issue(hb + D); ++inflight; }   // into the slot staged just now (nobody else reads it)
            tw += c1 - c0; tg += c2 - c1; tb += c3 - c2; ti += WL_STICK() - c3;
        }
        wl_wait_vm<0>();
        if ((WL_STRIP_ABLATE & 8) && lane == 0 && sidx == 0 && s.k0 == 0 && s.o_lo == 0) {
            T* o = a.ll + (size_t)plane * a.ll_ps;
            o[8] = (T)(float)(tw >> 10); o[9] = (T)(float)(tg >> 10); o[10] = (T)(float)(tb >> 10); o[11] = (T)(float)(ti >> 10);
        }
    }

    // ---- stager wave, direct form: row `sidx` of every half-batch ------------------------------------------------------
    // The stager needs every sample in a register anyway (conversion, alignment): instead of LDS-DMA into a ring and an
    // LDS read, the lane loads its 4-cell groups straight from global memory (8 / 16 bytes per lane, consecutive lanes on
    // consecutive addresses, wrapped columns resolved once per lane) one half-batch ahead, into one of two register
    // sets, and stages the other.  No DMA ring in LDS, no counted waits (the compiler tracks ordinary loads), no DMA
    // instruction issue - which cost a stager ~250 cycles apiece under load.
    static const int MAXG = 6;             // 4-cell groups per lane and row: strips of up to 6 x 64 x 4 cells
    static const int PF = WL_STRIP_PF;
    // (element-aligned: rows of any width and pitch; gfx950 takes unaligned vector loads)
    typedef T Quad4 __attribute__((ext_vector_type(4), aligned(sizeof(T)), may_alias));
    struct RowRegs { Quad4 g[MAXG]; T h[2]; };
    template <int DM, int NGL>
    static WL_DEV void stage_regs(const RowRegs& rr, char* drow, int imin, int imax, bool zero) {
#pragma unroll
        for (int i = 0; i < NGL; ++i) {
            if (i < imin || i >= imax) continue;
            float v0 = (float)rr.g[i].x, v1 = (float)rr.g[i].y, v2 = (float)rr.g[i].z, v3 = (float)rr.g[i].w;
            if (zero) v0 = v1 = v2 = v3 = 0.f;
            float* dst = reinterpret_cast<float*>(drow + i * 1024);
            if (DM == 0) {
                wl_vf4 w; w.x = v0; w.y = v1; w.z = v2; w.w = v3;
                *reinterpret_cast<wl_vf4*>(dst) = w;
            } else if (DM == 2) {
                wl_f2 w0, w1; w0.x = v0; w0.y = v1; w1.x = v2; w1.y = v3;
                *reinterpret_cast<wl_f2*>(dst) = w0; *reinterpret_cast<wl_f2*>(dst + 2) = w1;
            } else {                                          // odd: the middle two cells are 8-byte aligned
                wl_f2 w; w.x = v1; w.y = v2;
                dst[0] = v0; *reinterpret_cast<wl_f2*>(dst + 1) = w; dst[3] = v3;
            }
        }
    }
    // NGL = groups per lane and row (compile-time, so that every load of a half-batch is unconditional: the compiler
    // can then count them and wait for exactly the older register set; with predicated loads it waits for all of them,
    // i.e. for the loads it has just issued - measured 36 % slower)
    // PP = planes of this workgroup (run()): the wave takes row `sidx` of every half-batch of each of them, into that plane's ring
    template <int NGL, int PP>
    static WL_DEV void stager_direct(const Args& a, const Strip& s, const WlCtx& ctx, int64_t plane, int lane, int sidx) {
        static_assert(LROWS == 1 || PP == 1, "several planes per workgroup: four stager waves, one row each");
        const char* xp[PP];
#pragma unroll
        for (int p = 0; p < PP; ++p) {      // (a last workgroup with fewer planes stages its last plane again: nobody reads it)
            const int64_t pl = plane + p < a.NC ? plane + p : a.NC - 1;
            xp[p] = reinterpret_cast<const char*>(a.x + (size_t)pl * a.x_ps);
        }
        const int row_stride = a.x_rs * SZ;
        const bool wrap = wraps(a.ext);
        const int e_first = a.base + 2 * s.o_lo;
        const int e_last = a.base + 2 * (s.o_lo + s.nfeeds) - 1;
        // this lane's groups lane + 64 i: source byte inside a row (wrapped), valid range [imin, imax)
        int goff[MAXG];
        int imin, imax;
        {
            int g_lo = 0, g_hi = s.ng;
            if (!wrap) {
                if (s.c0a < 0) g_lo = (-s.c0a + 3) / 4;
                const int lim = (a.W - s.c0a) / 4;
                if (lim < g_hi) g_hi = lim;
            }
            imin = g_lo > lane ? (g_lo - lane + 63) / 64 : 0;
            imax = g_hi > lane ? (g_hi - lane + 63) / 64 : 0;
#pragma unroll
            for (int i = 0; i < MAXG; ++i) {
                int col = s.c0a + 4 * (lane + 64 * i);
                if (wrap) col = wl_pmod(col, a.W);
                goff[i] = (i >= imin && i < imax) ? col * SZ : 0;
            }
        }
        // single cells: the mirrored halo (symmetric / reflect, strips at either edge) and the last 1-3 columns of a row
        // whose width is not a multiple of four (the whole groups stop before them); a lane handles cells lane, lane + 64
        int hdst[2], hoff[2];
        {
            const int e_hi = 2 * (s.k1 - 1) + a.base + LT - 1 + 2;
            const int nl = s.e_lo < 0 ? -s.e_lo : 0, nr = e_hi > a.W - 1 ? e_hi - (a.W - 1) : 0;
            const int NH = (a.ext == WL_EXT_SYM || a.ext == WL_EXT_REFL) ? nl + nr : 0;
            const int t0 = s.c0a + (a.W - s.c0a) / 4 * 4;              // first column behind the whole groups
            const int NT = (!wrap && t0 <= e_hi && t0 < a.W) ? a.W - t0 : 0;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int c = lane + 64 * u;
                hdst[u] = -1; hoff[u] = 0;
                if (c < NH) {
                    const int e = c < nl ? s.e_lo + c : a.W + (c - nl);
                    hdst[u] = staged_of(s, e - s.c0a) * 4;
                    hoff[u] = wl_ext(e, a.W, a.ext) * SZ;
                } else if (c < NH + NT) {
                    const int e = t0 + (c - NH);
                    hdst[u] = staged_of(s, e - s.c0a) * 4;
                    hoff[u] = e * SZ;
                }
            }
        }
        // (register rows of a half-batch: q = p * LROWS + r - plane p of the workgroup, row sidx * LROWS + r of the half-batch)
        auto src_row = [&](int h, int r) {
            int e = e_first + 4 * h + sidx * LROWS + r;
            e = e < e_last ? e : e_last;
            return (unsigned)e < (unsigned)a.H ? e : wl_ext(e, a.H, a.ext);   // -1: a row of zeros
        };
        auto load = [&](int h, RowRegs (&rr)[PP * LROWS]) {
#pragma unroll
            for (int q = 0; q < PP * LROWS; ++q) {
                int r = src_row(h, q % LROWS);
                r = r < 0 ? 0 : r;
                const char* grow = xp[q / LROWS] + (size_t)r * row_stride;
#pragma unroll
                for (int i = 0; i < NGL; ++i) rr[q].g[i] = *reinterpret_cast<const Quad4*>(grow + goff[i]);   // (off lanes: the row's first group)
#pragma unroll
                for (int u = 0; u < 2; ++u) rr[q].h[u] = *reinterpret_cast<const T*>(grow + hoff[u]);
            }
        };
        auto stage = [&](int hb, const RowRegs (&rr)[PP * LROWS]) {
            if (!(WL_STRIP_ABLATE & 2)) {
#pragma unroll
                for (int q = 0; q < PP * LROWS; ++q) {
                    const bool zero = src_row(hb, q % LROWS) < 0;
                    char* srow0 = ctx.smem + a.st_off + (q / LROWS) * a.ring + ((hb & 1) * 4 + sidx * LROWS + q % LROWS) * a.st_pitch;
                    char* drow = srow0 + lane * 16 + (4 - s.dm) * 4;
                    if (s.dm == 0) stage_regs<0, NGL>(rr[q], drow, imin, imax, zero);
                    else if (s.dm == 2) stage_regs<2, NGL>(rr[q], drow, imin, imax, zero);
                    else stage_regs<1, NGL>(rr[q], drow, imin, imax, zero);
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        if (hdst[u] >= 0) *reinterpret_cast<float*>(srow0 + hdst[u]) = zero ? 0.f : (float)rr[q].h[u];
                }
            }
        };
        // PF register sets: the rows of the next PF - 1 half-batches are in flight while one is staged.  Measured on config 5
        // (round 4, same box): 2, 3 and 4 sets run the same 2.18 ms - the rows arrive in time with two; so did a stager with
        // every per-row branch hoisted out of its loop (2.20 ms) and one with aligned, conflict-free staging writes (2.13 ms,
        // a timing build): the level-1 kernel is bound by what the compute waves issue (VALU 0.67 of the cycles) next to a
        // memory system that is moving 4 TB/s, not by its stagers
        RowRegs rr[PF][PP * LROWS];
#pragma unroll
        for (int u = 0; u < PF - 1; ++u)
            if (u < s.nhb) load(u, rr[u]);
        for (int hb = 0; hb < s.nhb; hb += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                if (hb + u >= s.nhb) break;
                if (hb + u + PF - 1 < s.nhb) load(hb + u + PF - 1, rr[(u + PF - 1) % PF]);
                stage(hb + u, rr[u]);
                ctx.sync();
            }
        }
    }

    // ---- compute wave ---------------------------------------------------------------------------------------------
    // acc (+)= (lo[t], hi[t]) * s.x / s.y (XY = 0 / 1); t and XY are compile-time after unrolling: exactly one form survives
    static WL_DEV void tap_fma(wl_v2& acc, const wl_v2 (&bank)[NB], int t, int xy, wl_v2 s) {
        if (!QMF) { if (xy) wl_pk_fma_y(acc, bank[t], s); else wl_pk_fma_x(acc, bank[t], s); return; }
        const int u = t < LT / 2 ? t : LT - 1 - t;
        const bool sw = t >= LT / 2, neg = t & 1;
#if defined(__HIPCC__)
        if (!sw && !neg) { if (xy) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(bank[u]), "v"(s));
                           else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "s"(bank[u]), "v"(s)); }
        else if (!sw && neg) { if (xy) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_hi:[1,0,0]" : "+v"(acc) : "s"(bank[u]), "v"(s));
                               else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1] neg_hi:[1,0,0]" : "+v"(acc) : "s"(bank[u]), "v"(s)); }
        else if (sw && !neg) { if (xy) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(bank[u]), "v"(s));
                               else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,0,1]" : "+v"(acc) : "s"(bank[u]), "v"(s)); }
        else { if (xy) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "+v"(acc) : "s"(bank[u]), "v"(s));
               else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "+v"(acc) : "s"(bank[u]), "v"(s)); }
#else
        const float c = xy ? s.y : s.x;
        const float lo = sw ? bank[u].y : bank[u].x, hi = (sw ? bank[u].x : bank[u].y) * (neg ? -1.f : 1.f);
        acc.x = __builtin_fmaf(lo, c, acc.x); acc.y = __builtin_fmaf(hi, c, acc.y);
#endif
    }
    static WL_DEV wl_v2 tap_mul(const wl_v2 (&bank)[NB], int t, int xy, wl_v2 s) {
        if (!QMF) return xy ? wl_pk_mul_y(bank[t], s) : wl_pk_mul_x(bank[t], s);
        const int u = t < LT / 2 ? t : LT - 1 - t;
        const bool sw = t >= LT / 2, neg = t & 1;
        wl_v2 r;
#if defined(__HIPCC__)
        if (!sw && !neg) { if (xy) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "s"(bank[u]), "v"(s));
                           else asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "s"(bank[u]), "v"(s)); }
        else if (!sw && neg) { if (xy) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_hi:[1,0]" : "=v"(r) : "s"(bank[u]), "v"(s));
                               else asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0] neg_hi:[1,0]" : "=v"(r) : "s"(bank[u]), "v"(s)); }
        else if (sw && !neg) { if (xy) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(r) : "s"(bank[u]), "v"(s));
                               else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,0]" : "=v"(r) : "s"(bank[u]), "v"(s)); }
        else { if (xy) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1] neg_hi:[1,0]" : "=v"(r) : "s"(bank[u]), "v"(s));
               else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(r) : "s"(bank[u]), "v"(s)); }
#else
        const float c = xy ? s.y : s.x;
        const float lo = sw ? bank[u].y : bank[u].x, hi = (sw ? bank[u].x : bank[u].y) * (neg ? -1.f : 1.f);
        r.x = lo * c; r.y = hi * c;
#endif
        return r;
    }
    template <int HI, int NZ> static WL_DEV wl_v2 fma_s(wl_v2 w, wl_v2 pair, wl_v2 z) { return wl_fma_s<HI, NZ>(w, pair, z); }
    struct Wave {
        wl_v2 tw[NB], th[NB];             // (lo,hi) tap pairs along W / along H (wave-uniform: scalar registers); QMF: P[u] = (lo[u], lo[L-1-u])
        wl_v2 lt[LAT ? KL : 1];           // LAT: (T_k, -T_k) of the column lattice (the row taps carry its gain g)
        char* llp; char* hp0; char* hp1; char* hp2;
        unsigned rowb, llrowb;
        bool two, pair_ok;
    };
    // row filter of one staged row for both columns of the lane: s = its L+2 samples as (even, odd) pairs
    static WL_DEV void row_pass(const Wave& R, const wl_v2 (&s)[2 * NV4], wl_v2& ra, wl_v2& rb) {
        wl_v2 a0 = tap_mul(R.tw, 0, 0, s[0]), b0 = tap_mul(R.tw, 0, 0, s[1]);
        wl_v2 a1 = tap_mul(R.tw, 1, 1, s[0]), b1 = tap_mul(R.tw, 1, 1, s[1]);
#pragma unroll
        for (int u = 1; u < LT / 2; ++u) {
            tap_fma(a0, R.tw, 2 * u, 0, s[u]);
            tap_fma(b0, R.tw, 2 * u, 0, s[u + 1]);
            tap_fma(a1, R.tw, 2 * u + 1, 1, s[u]);
            tap_fma(b1, R.tw, 2 * u + 1, 1, s[u + 1]);
        }
        ra = a0 + a1;
        rb = b0 + b1;
    }
    // column filter over the circular window whose OLDEST row sits in slot `first` (compile-time after unrolling)
    static WL_DEV void col_pass(const Wave& R, const wl_v2 (&w)[LW], int first, wl_v2& cl, wl_v2& ch) {
        wl_v2 l0 = tap_mul(R.th, 0, 0, w[first % LW]), h0 = tap_mul(R.th, 0, 1, w[first % LW]);
        wl_v2 l1 = tap_mul(R.th, 1, 0, w[(first + 1) % LW]), h1 = tap_mul(R.th, 1, 1, w[(first + 1) % LW]);
#pragma unroll
        for (int t = 2; t < LT; t += 2) {
            tap_fma(l0, R.th, t, 0, w[(first + t) % LW]);
            tap_fma(h0, R.th, t, 1, w[(first + t) % LW]);
            tap_fma(l1, R.th, t + 1, 0, w[(first + t + 1) % LW]);
            tap_fma(h1, R.th, t + 1, 1, w[(first + t + 1) % LW]);
        }
        cl = l0 + l1;
        ch = h0 + h1;
    }
    // One feed of the column lattice for both columns of the lane (wl_lattice.h): (a, b) = the row-filtered pair of new rows;
    // delay slot of stage k at feed f: (k - f) mod K - stage k - 1 leaves its v in the slot it has just read, stage 0 in the
    // one free slot, so nothing is ever moved (`rot` = f mod K is compile-time after unrolling).  The two columns are
    // interleaved stage by stage: a packed FMA that depends on the one right before it costs a wait state.
    static WL_DEV void lat_feed(const Wave& R, wl_v2 (&SA)[KL], wl_v2 (&SB)[KL], int rot, wl_v2 aA, wl_v2 bA, wl_v2 aB, wl_v2 bB,
                                wl_v2& loA, wl_v2& hiA, wl_v2& loB, wl_v2& hiB) {
        wl_v2 uA = fma_s<0, 0>(bA, R.lt[0], aA), uB = fma_s<0, 0>(bB, R.lt[0], aB);         // a + T0 b
        wl_v2 vA = fma_s<1, 0>(aA, R.lt[0], bA), vB = fma_s<1, 0>(aB, R.lt[0], bB);         // b - T0 a
        if (KL == 1) { loA = uA; loB = uB; hiA = -vA; hiB = -vB; return; }
        SA[(KL - rot) % KL] = vA; SB[(KL - rot) % KL] = vB;
#pragma unroll
        for (int k = 1; k < KL; ++k) {
            const int slot = (k + KL - rot) % KL;
            const wl_v2 dA = SA[slot], dB = SB[slot];
            const wl_v2 nA = fma_s<0, 0>(dA, R.lt[k], uA), nB = fma_s<0, 0>(dB, R.lt[k], uB);   // u + Tk v'
            if (k < KL - 1) {
                SA[slot] = fma_s<1, 0>(uA, R.lt[k], dA); SB[slot] = fma_s<1, 0>(uB, R.lt[k], dB);   // v' - Tk u
            } else {
                hiA = fma_s<0, 1>(uA, R.lt[k], dA); hiB = fma_s<0, 1>(uB, R.lt[k], dB);             // -(v' - Tk u)
            }
            uA = nA; uB = nB;
        }
        loA = uA; loB = uB;
    }
    static WL_DEV void load_row(const char* p, wl_v2 (&s)[2 * NV4]) {
#pragma unroll
        for (int u = 0; u < NV4; ++u) {
            const wl_vf4 t = *reinterpret_cast<const wl_vf4*>(p + 16 * u);
            s[2 * u] = wl_v2{t.x, t.y};
            s[2 * u + 1] = wl_v2{t.z, t.w};
        }
    }
    // the two columns of a lane: one aligned 2-element store (wl_store2_s) when the geometry makes every pair aligned (even
    // band width), else element stores (an odd band width has rows on odd element offsets and a last lane with one column)
    static WL_DEV void store_single(const Wave& R, char* p, float va, float vb) {
        *reinterpret_cast<T*>(p) = (T)va;
        if (R.two) *reinterpret_cast<T*>(p + SZ) = (T)vb;
    }

    // cw = the wave's run of 64 column pairs inside the strip, sub = which of the workgroup's planes (its staged ring)
    static WL_DEV void compute(const Args& a, const Strip& s, const WlCtx& ctx, int64_t plane, int cw, int lane, int sub) {
        const int jp = 64 * cw + lane;                        // column pair inside the strip
        const int kA = s.k0 + 2 * jp;
        const bool active = kA < s.k1;
        Wave R;
        const float gsc = LAT ? a.lat[1] : 1.f;               // the lattice's gain rides on the row taps
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            R.tw[t] = wl_uniform_v2(LAT ? wl_v2{a.h_w_lo[t] * gsc, a.h_w_lo[LT - 1 - t] * gsc}
                                        : QMF ? wl_v2{a.h_w_lo[t], a.h_w_lo[LT - 1 - t]} : wl_v2{a.h_w_lo[t], a.h_w_hi[t]});
            if (!LAT) R.th[t] = wl_uniform_v2(QMF ? wl_v2{a.h_h_lo[t], a.h_h_lo[LT - 1 - t]} : wl_v2{a.h_h_lo[t], a.h_h_hi[t]});
        }
        if (LAT) {
#pragma unroll
            for (int k = 0; k < KL; ++k) R.lt[LAT ? k : 0] = wl_uniform_v2(wl_v2{a.lat[2 + k], -a.lat[2 + k]});
        }
        const unsigned bplane = (unsigned)a.Kh * (unsigned)a.Kw;
        R.hp0 = reinterpret_cast<char*>(a.highs + (size_t)plane * 3 * bplane);
        R.hp1 = R.hp0 + (size_t)bplane * SZ;
        R.hp2 = R.hp1 + (size_t)bplane * SZ;
        R.llp = reinterpret_cast<char*>(a.ll + (size_t)plane * a.ll_ps);
        R.rowb = (unsigned)a.Kw * SZ; R.llrowb = (unsigned)a.ll_rs * SZ;
        R.two = kA + 1 < s.k1;
        R.pair_ok = a.pair_ok != 0;
        const int soff = s.lane_off + 16 * (active ? jp : 0);
        // the next output row of the four planes as wave-uniform pointers (scalar registers, advanced by scalar adds) + ONE
        // 32-bit lane offset: the stores address as scalar base + vector offset, no 64-bit vector adds per store
        const unsigned voff = (unsigned)kA * SZ;
        char* pll = R.llp + (size_t)((unsigned)s.o_lo * R.llrowb);
        char* ph0 = R.hp0 + (size_t)((unsigned)s.o_lo * R.rowb);
        char* ph1 = R.hp1 + (size_t)((unsigned)s.o_lo * R.rowb);
        char* ph2 = R.hp2 + (size_t)((unsigned)s.o_lo * R.rowb);
        wl_v2 wa[LAT ? 2 : LW], wb[LAT ? 2 : LW];                             // circular windows of the two columns (LAT: the two new rows)
        wl_v2 SA[KL], SB[KL];                                                 // LAT: delay slots of the two columns' lattices
#pragma unroll
        for (int t = 0; t < (LAT ? 2 : LW); ++t) wa[t] = wb[t] = wl_v2{0.f, 0.f};
#pragma unroll
        for (int t = 0; t < KL; ++t) SA[t] = SB[t] = wl_v2{0.f, 0.f};
        char* const smem = ctx.smem;
        int fed = 0;                                                          // feeds done (wave-uniform)
        unsigned long long tbar = 0, tmath = 0;
        for (int hb0 = 0; hb0 < s.nhb; hb0 += PERIOD) {
#pragma unroll
            for (int ph = 0; ph < PERIOD; ++ph) {
                const int hb = hb0 + ph;
                if (hb >= s.nhb) break;
                const unsigned long long c0 = WL_STICK();
                ctx.sync();
                const unsigned long long c1 = WL_STICK();
                tbar += c1 - c0;
                const int left = s.nfeeds - fed;
                const int n = left > 2 ? 2 : left;
                const char* slot = smem + a.st_off + sub * a.ring + (hb & 1) * 4 * a.st_pitch + soff;
                if (active && !(WL_STRIP_ABLATE & 1)) {
                    // Short filters: all four rows of the half-batch are requested from LDS before the first FMA.  From 14
                    // taps on the registers do not allow that at four waves per SIMD: the rows of the second feed are
                    // requested behind the row filter of the first and land during its column filter.
                    // (The last half-batch of a segment may hold one feed only: the other two rows are read and ignored.)
                    wl_v2 sr[4][2 * NV4];
                    load_row(slot, sr[0]);
                    load_row(slot + a.st_pitch, sr[1]);
                    if (!kPipe) { load_row(slot + 2 * a.st_pitch, sr[2]); load_row(slot + 3 * a.st_pitch, sr[3]); }
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if (i < n) {
                            const int w0 = LAT ? 0 : (4 * ph + 2 * i) % LW;   // slots of the two new rows
                            row_pass(R, sr[2 * i], wa[w0], wb[w0]);
                            row_pass(R, sr[2 * i + 1], wa[(w0 + 1) % LW], wb[(w0 + 1) % LW]);
                            if (kPipe && i == 0) { load_row(slot + 2 * a.st_pitch, sr[2]); load_row(slot + 3 * a.st_pitch, sr[3]); }
                            wl_v2 cla, cha, clb, chb;
                            if constexpr (LAT != 0) {   // (every feed: the lattice's state; its first K - 1 outputs of a segment are the warm-up)
                                // the lattice filters the packed (row-lo, row-hi) pair: lo = (ll, hl), hi = (lh, hh); the direct
                                // form packs the other way round, cl = (ll, lh), ch = (hl, hh): a renaming of registers
                                wl_v2 loA, hiA, loB, hiB;
                                lat_feed(R, SA, SB, (2 * ph + i) % KL, wa[0], wa[1], wb[0], wb[1], loA, hiA, loB, hiB);
                                cla = wl_v2{loA.x, hiA.x}; cha = wl_v2{loA.y, hiA.y};
                                clb = wl_v2{loB.x, hiB.x}; chb = wl_v2{loB.y, hiB.y};
                            }
                            if (fed + i >= WARM) {
                                // the L rows of this output end with the two new ones: the oldest sits LT-1 slots back
                                const int first = (w0 + 1 + LW - (LT - 1)) % LW;
                                if constexpr (!LAT) {
                                    col_pass(R, wa, first, cla, cha);
                                    col_pass(R, wb, first, clb, chb);
                                }
                                if (!(WL_STRIP_ABLATE & 4) || cla.x + clb.y + cha.x + chb.y == 1.2345e30f) {
                                    // (the second feed's row is one further down when the first feed emitted one too)
                                    const unsigned k = (i == 1 && fed >= WARM) ? 1u : 0u;
                                    char* const q0 = pll + (size_t)(k * R.llrowb);
                                    char* const q1 = ph0 + (size_t)(k * R.rowb);
                                    char* const q2 = ph1 + (size_t)(k * R.rowb);
                                    char* const q3 = ph2 + (size_t)(k * R.rowb);
                                    if (R.pair_ok) {          // (one branch per row, not one per store)
                                        wl_store2_s(q0, voff, cla.x, clb.x, (T*)nullptr); wl_store2_s(q1, voff, cla.y, clb.y, (T*)nullptr);
                                        wl_store2_s(q2, voff, cha.x, chb.x, (T*)nullptr); wl_store2_s(q3, voff, cha.y, chb.y, (T*)nullptr);
                                    } else {
                                        store_single(R, q0 + voff, cla.x, clb.x); store_single(R, q1 + voff, cla.y, clb.y);
                                        store_single(R, q2 + voff, cha.x, chb.x); store_single(R, q3 + voff, cha.y, chb.y);
                                    }
                                }
                            }
                        }
                    }
                }
                {   // rows emitted in this half-batch: the row pointers advance outside the lanes' branch (they stay scalar)
                    int em = fed + n - WARM;
                    em = em < 0 ? 0 : em > n ? n : em;
                    pll += (size_t)((unsigned)em * R.llrowb); ph0 += (size_t)((unsigned)em * R.rowb);
                    ph1 += (size_t)((unsigned)em * R.rowb); ph2 += (size_t)((unsigned)em * R.rowb);
                }
                fed += n;
                tmath += WL_STICK() - c1;
            }
        }
        if ((WL_STRIP_ABLATE & 8) && lane == 0 && cw == 0 && sub == 0 && s.k0 == 0 && s.o_lo == 0) {
            T* o = a.ll + (size_t)plane * a.ll_ps;
            o[0] = (T)(float)(tbar >> 10); o[1] = (T)(float)(tmath >> 10);
        }
    }

    static WL_DEV void run(const Args& a, const WlCtx& ctx) {
        const int tid = ctx.tid;
        const int wave = wl_uniform(tid >> 6), lane = tid & 63;
        if (a.guard) {   // the relation the QMF / LAT variant relies on, checked against the taps as they are now (uniform: before any barrier)
            const bool holds = a.lat ? *reinterpret_cast<const unsigned*>(a.lat) == WL_LAT_OK   // (WlTapPrep's verdict, read by both launches)
                                     : wl_taps_qmf(a.h_w_lo, a.h_w_hi, LT) && wl_taps_qmf(a.h_h_lo, a.h_h_hi, LT);
            if (!wl_guard_pass(a.guard, holds)) return;
        }
        // workgroup -> (plane, segment, strip): strips of one plane and segment are neighbours on the same XCD (they
        // share halo columns), segments next
        // A NARROW strip (the deeper levels of a wide pyramid: a whole row of 256 / 128 output columns) keeps two / one of the four
        // compute waves busy, and the workgroup's cadence - one barrier per half-batch, ~3000 cycles whatever the width - does not
        // speed up for it (config 5, per-level counters: the 512- and 256-column levels ran at 0.34 / 0.23 of the VALU issue rate
        // against 0.60 for the wide ones).  Such a workgroup takes pp = 2 / 4 PLANES at once: compute waves [sub * 4 / pp, ..)
        // work on plane `sub` out of its own staged ring, every stager takes its row of each plane.
        const int64_t lbid = wl_xcd_remap(ctx.bid, a.nblocks);
        const int per_plane = a.nstrips * a.nseg;
        const int64_t pgroup = lbid / per_plane;
        const int64_t plane = pgroup * a.pp;
        const int rem = (int)(lbid - pgroup * per_plane);
        const int seg = rem / a.nstrips, strip = rem - seg * a.nstrips;
        const Strip s = geometry(a, strip, seg);
        // LDS starts as zeros: staged cells nobody writes are the zero padding
        for (int i = tid * 16; i < a.lds_bytes; i += kThreads * 16) {
            wl_f4 z; z.x = z.y = z.z = z.w = 0.f;
            *reinterpret_cast<wl_f4*>(ctx.smem + i) = z;
        }
        ctx.sync();
        if (wave >= WL_STRIP_CWAVES) {
#if defined(__HIPCC__)
            __builtin_amdgcn_s_setprio(2);   // every compute wave waits for the stagers at the barrier
#endif
            const int sidx = wave - WL_STRIP_CWAVES;
            if (WL_STRIP_DIRECT) {
                const int ngl = (s.ng + 63) >> 6;
                if constexpr (LROWS == 1) {   // (several planes: four stager waves, one row of each plane per wave)
                    if (a.pp == 4) {          // (the launcher: ngl <= 2 with four planes, <= 3 with two)
                        if (ngl == 1) stager_direct<1, 4>(a, s, ctx, plane, lane, sidx); else stager_direct<2, 4>(a, s, ctx, plane, lane, sidx);
                        return;
                    }
                    if (a.pp == 2) {
                        if (ngl == 1) stager_direct<1, 2>(a, s, ctx, plane, lane, sidx);
                        else if (ngl == 2) stager_direct<2, 2>(a, s, ctx, plane, lane, sidx);
                        else stager_direct<3, 2>(a, s, ctx, plane, lane, sidx);
                        return;
                    }
                }
                switch (ngl) {
                    case 1: stager_direct<1, 1>(a, s, ctx, plane, lane, sidx); break;
                    case 2: stager_direct<2, 1>(a, s, ctx, plane, lane, sidx); break;
                    case 3: stager_direct<3, 1>(a, s, ctx, plane, lane, sidx); break;
                    case 4: stager_direct<4, 1>(a, s, ctx, plane, lane, sidx); break;
                    case 5: stager_direct<5, 1>(a, s, ctx, plane, lane, sidx); break;
                    default: stager_direct<6, 1>(a, s, ctx, plane, lane, sidx); break;
                }
            } else stager(a, s, ctx, plane, lane, sidx);
            return;
        }
        const int nact = WL_STRIP_CWAVES / a.pp;              // compute waves per plane
        const int sub = wave / nact, cw = wave - sub * nact;
        if (plane + sub < a.NC && 64 * 2 * cw < s.k1 - s.k0) {
            compute(a, s, ctx, plane + sub, cw, lane, sub);
        } else {
            for (int hb = 0; hb < s.nhb; ++hb) ctx.sync();   // spare wave (narrow strip): keeps the barrier count
        }
    }
};
