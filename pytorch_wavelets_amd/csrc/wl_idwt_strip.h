// Streaming ONE-level 2-D DWT synthesis over column strips: the mirror image of wl_dwt_strip.h, for what the fused
// multi-level synthesis kernel (wl_idwt_rows.h) does not take - periodization (coefficient rows and columns wrap, the
// output is rolled by L/2-1), 14 to 20 taps, planes wider than one workgroup, float16 planes of any width.
//
// A workgroup owns one (plane, strip of output columns, segment of output rows) and marches down it once:
//   * four STAGER waves, one per source (ll, lh, hl, hh), bring the coefficient rows of the strip in by LDS-DMA (16-byte
//     pieces of the EXTENDED row: under periodization the pieces beyond either end come from the other end), three
//     half-batches ahead, and copy what they loaded as float32 (the float16 -> float32 conversion happens once per
//     coefficient) into a two-slot ring laid out so that every compute lane reads aligned 8-byte words;
//   * a compute lane owns FOUR adjacent output columns = two polyphase column pairs.  Per coefficient row ("feed") it
//     reads its L/2+1 (+1) coefficients of each source, runs the polyphase row synthesis of both pairs,
//         (a, b)[pair] = sum_j (ll | hl, lh | hh)[q - j] * (g[2j], g[2j+1]),
//     into a circular window of L/2 rows in registers (rotation resolved at compile time: the loop over half-batches is
//     unrolled by its period) and emits two output rows by the polyphase column synthesis: 17 packed FMAs per output
//     sample and nothing else;
//   * periodization with L % 4 == 0 rolls the output by an ODD number of samples: the lane's pairs then straddle two
//     polyphase pairs, which is the same sum with the tap pairs shifted by one, (g[2j-1], g[2j]) with g[-1] = g[L] = 0 -
//     one more tap pair per row, no shuffles (template parameter SODD);
//   * an output row leaves as 8 / 16 contiguous bytes per lane.
// HBM traffic = the strip's coefficients once (+ L/2 halo columns per strip, L/2-1 halo rows per segment) + the output.
//
// Restates SFB2D.forward (reference dwt/lowlevel.py:671-680 = sfb1d along H twice, then along W, :226-271), every mode;
// the crop to (OH, OW) is AFB2D.backward's (:356-364).
#pragma once
#include "wl_common.h"
#include "wl_dwt_rows.h"   // wl_pk_fma_x / _y, wl_pk_mul_x / _y, wl_uniform_v2
#include "wl_dwt_strip.h"  // WlStage, WL_STRIP_* geometry constants
#include "wl_lattice.h"

template <typename T>
struct WlIStripArgs {
    const T* ll;                   // (NC, Kh, Kw) through ll_ps / ll_rs
    const T* highs;                // (NC, 3, Kh, Kw) dense
    T* y;                          // (NC, OH, OW) dense
    const float* g_w_lo;
    const float* g_w_hi;
    const float* g_h_lo;
    const float* g_h_hi;
    int64_t NC, ll_ps, nblocks;
    int ll_rs;
    int Kh, Kw, OH, OW, per;       // per: periodization (coefficient rows / columns wrap, N = 2K outputs)
    int sw, sh;                    // z-column / z-row of output column / row 0 (z = the un-rolled, un-cropped synthesis)
    int nstrips, strip_units;      // lane units (4 output columns each) per strip; the last strip may be narrower
    int nseg, seg_pairs;           // z-row pairs per segment
    int m_first, m_end;            // z-row pairs [m_first, m_end) that hold output rows
    int dma_off, dma_pitch;        // DMA ring: WL_STRIP_D slots x 8 rows (4 sources x 2 coefficient rows) x dma_pitch bytes
    int st_off, st_pitch;          // staged ring: 2 slots x 8 rows x st_pitch bytes (float32)
    int lds_bytes;
    int pp, ring;                  // planes per workgroup (1, 2, 4: narrow strips, see WlAfbStrip::run) and the bytes of one plane's staged ring
    int quad_ok;                   // every lane's 4 output columns are one aligned store (OW % 4 == 0, aligned y)
    int guard;                     // tap-relation guard (wl_common.h): 1 = run only if both highpass banks are the quadrature mirrors of
                                   // their lowpass banks (the QMF variant), 2 = only if not (its armed two-bank fallback), 0 = no check
    const float* lat;              // LAT variant and its fallback: WlTapPrep's verdict + the column lattice in device scratch (wl_lattice.h);
                                   // the guard then reads the verdict word instead of the taps
};

// QMF = 1: the caller vouches that each highpass bank is the quadrature mirror of its lowpass bank, g1[t] = (-1)^t g0[L-1-t]
// (every orthogonal wavelet as pywt / the reference tabulates its RECONSTRUCTION pair): the highpass tap pairs are then the
// lowpass pairs read backwards with the halves swapped and one half negated - operand modifiers of the packed FMA - and never
// occupy scalar registers (68 of them at 16 taps: the scalar file overflowed, 32 v_readlane per half-batch).
// LAT = 1 (with QMF = 1): the COLUMN synthesis runs the orthogonal bank as a lattice of rotations (wl_lattice.h, the transposed
// recurrence): per coefficient row and column pair 2 K packed FMAs and K - 1 delayed values instead of 2 L FMAs and two L/2-row
// windows (16 taps: 64 of the 128 FMAs of a feed are column FMAs, 32 of them go).
template <typename T, int LT, int SODD, int QMF = 0, int LAT = 0>
struct WlSfbStrip {
    typedef WlIStripArgs<T> Args;
    static_assert(!LAT || QMF, "the lattice variant runs its row synthesis in the QMF form");
    static const int kWaves = WL_STRIP_CWAVES + 4;
    static const int kThreads = 64 * kWaves;
#ifndef WL_STRIP_LAT_MINW
#define WL_STRIP_LAT_MINW 4     // waves per SIMD the lattice variants are compiled for (6: three workgroups per CU, at most 80 registers)
#endif
    // two 8-wave workgroups per CU need four waves per SIMD: at most 128 registers (18, 20 taps: 168)
    static const int kMinWaves = (LAT && LT <= 16) ? WL_STRIP_LAT_MINW : LT >= 18 ? 3 : 4;
    static const int SZ = (int)sizeof(T);
    static const int A = 16 / SZ;
    static const int HL = LT / 2;
    static const int NT = HL + SODD;       // tap pairs of the row synthesis
    static const int NC2 = (NT + 1 + 1) / 2;   // 8-byte words a lane reads per source row: its NT + 1 coefficients
    static const int LW = (HL + 1) / 2 * 2;    // window slots: a multiple of the 2 rows of a half-batch
    static const int PERIOD = LAT ? (HL % 2 ? HL : HL / 2) : LW / 2;   // (lattice: HL delay slots, two feeds per half-batch)
    static const int D = WL_STRIP_D;

    struct Strip {
        int u0, u1;            // lane units [u0, u1): output columns [4 u0, 4 u1)
        int q_lo;              // first coefficient column a lane reads
        int c0a;               // coefficient column of DMA-ring cell 0 (a multiple of A, <= q_lo)
        int np, ppr, ng;       // pieces / DMA instructions / 4-cell groups per DMA-ring row
        int dm;                // (q_lo - c0a) & 1
        int lane_off;          // staged-ring byte offset of lane unit u0's first coefficient
        int m_lo, m_hi;        // z-row pairs of this segment
        int e_first, nfeeds, nhb;
    };
    static WL_HD Strip geometry(const Args& a, int strip, int seg) {
        Strip s;
        s.u0 = strip * a.strip_units;
        const int units = (a.OW + 3) / 4;
        s.u1 = s.u0 + a.strip_units < units ? s.u0 + a.strip_units : units;
        const int qoff = (a.sw + SODD) / 2;
        s.q_lo = 2 * s.u0 + qoff - NT + 1;
        const int q_hi = 2 * (s.u1 - 1) + qoff + 1;
        s.c0a = s.q_lo >= 0 ? s.q_lo / A * A : -((-s.q_lo + A - 1) / A * A);
        int last = q_hi;
        if (!a.per && last > a.Kw - 1) last = a.Kw - 1;
        s.np = (last - s.c0a) / A + 1;
        s.ppr = (s.np + 63) / 64;
        s.ng = s.np * A / 4;
        const int d = s.q_lo - s.c0a;
        s.dm = d & 1;
        s.lane_off = (d + s.dm) * 4;                 // staged cell of DMA cell c = c + dm: lanes read 8-byte aligned
        s.m_lo = a.m_first + seg * a.seg_pairs;
        s.m_hi = s.m_lo + a.seg_pairs < a.m_end ? s.m_lo + a.seg_pairs : a.m_end;
        s.e_first = s.m_lo - (HL - 1);
        s.nfeeds = s.m_hi - s.e_first;
        s.nhb = (s.nfeeds + 1) / 2;
        return s;
    }

    // ---- stager wave: source b (0 = ll, 1..3 = the high-pass bands) ---------------------------------------------------
    static WL_DEV void stager(const Args& a, const Strip& s, const WlCtx& ctx, int64_t plane, int lane, int b) {
        const char* bp = b == 0 ? reinterpret_cast<const char*>(a.ll + (size_t)plane * a.ll_ps)
                                : reinterpret_cast<const char*>(a.highs + ((size_t)plane * 3 + (b - 1)) * ((size_t)a.Kh * a.Kw));
        const int row_stride = (b == 0 ? a.ll_rs : a.Kw) * SZ;
        const int e_last = s.e_first + s.nfeeds - 1;
        int gbyte[WL_STRIP_MAXPPR];
#pragma unroll
        for (int q = 0; q < WL_STRIP_MAXPPR; ++q) {
            const int p = q * 64 + lane;
            int col = s.c0a + p * A;
            bool on = q < s.ppr && p < s.np;
            if (on && (unsigned)col >= (unsigned)a.Kw) {
                if (a.per) col = wl_pmod(col, a.Kw); else on = false;
            }
            gbyte[q] = on ? col * SZ : -1;
        }
        const int ngl = (s.ng + 63) >> 6;
        int imin, imax;
        {
            int g_lo = 0, g_hi = s.ng;
            if (!a.per) {
                if (s.c0a < 0) g_lo = (-s.c0a + 3) / 4;
                const int lim = (a.Kw - s.c0a) / 4;
                if (lim < g_hi) g_hi = lim;
            }
            imin = g_lo > lane ? (g_lo - lane + 63) / 64 : 0;
            imax = g_hi > lane ? (g_hi - lane + 63) / 64 : 0;
        }
        auto issue = [&](int h) {
            const int slot = a.dma_off + (h % D) * 8 * a.dma_pitch;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int e = s.e_first + 2 * h + i;
                e = e < e_last ? e : e_last;
                const int r = a.per ? wl_pmod(e, a.Kh) : e;      // (non-periodization: every feed is a real row)
                const char* grow = bp + (size_t)r * row_stride;
#pragma unroll
                for (int q = 0; q < WL_STRIP_MAXPPR; ++q)
                    if (q < s.ppr) wl_dma16(ctx, (unsigned)(slot + (2 * b + i) * a.dma_pitch + q * 1024), grow + gbyte[q], gbyte[q] >= 0);
            }
        };
        const int nl_inst = 2 * s.ppr;
        for (int h = 0; h < D && h < s.nhb; ++h) issue(h);
        int inflight = D < s.nhb ? D : s.nhb;
        for (int hb = 0; hb < s.nhb; ++hb) {
            wl_wait_vm_dyn((inflight - 1) * nl_inst);
            --inflight;
            const char* dslot = ctx.smem + a.dma_off + (hb % D) * 8 * a.dma_pitch + 2 * b * a.dma_pitch;
            char* sslot = ctx.smem + a.st_off + (hb & 1) * 8 * a.st_pitch + 2 * b * a.st_pitch;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const char* srow = dslot + i * a.dma_pitch + lane * 4 * SZ;
                char* drow = sslot + i * a.st_pitch + lane * 16 + s.dm * 4;
                if (s.dm == 0) WlStage<T>::template stage_row<0>(srow, drow, imin, imax, ngl);
                else WlStage<T>::template stage_row<1>(srow, drow, imin, imax, ngl);
            }
            ctx.sync();
            if (hb + D < s.nhb) { issue(hb + D); ++inflight; }
        }
        wl_wait_vm<0>();
    }

    // ---- stager wave, direct form (see wl_dwt_strip.h): source b, two coefficient rows per half-batch, loaded straight
    // into registers one half-batch ahead; NGL = 4-cell groups per lane and row (compile-time: static load counts)
    static const int MAXG = 6;
    typedef T Quad4 __attribute__((ext_vector_type(4), aligned(sizeof(T)), may_alias));   // element-aligned: any row width / pitch
    template <int NGL> struct RowRegs { Quad4 g[2][NGL]; T t[2]; };
    // PP = planes of this workgroup: the wave takes the two rows of its band for each of them, into that plane's ring
    template <int NGL, int PP>
    static WL_DEV void stager_direct(const Args& a, const Strip& s, const WlCtx& ctx, int64_t plane, int lane, int b) {
        const char* bp[PP];
#pragma unroll
        for (int p = 0; p < PP; ++p) {      // (a last workgroup with fewer planes stages its last plane again: nobody reads it)
            const int64_t pl = plane + p < a.NC ? plane + p : a.NC - 1;
            bp[p] = b == 0 ? reinterpret_cast<const char*>(a.ll + (size_t)pl * a.ll_ps)
                           : reinterpret_cast<const char*>(a.highs + ((size_t)pl * 3 + (b - 1)) * ((size_t)a.Kh * a.Kw));
        }
        const int row_stride = (b == 0 ? a.ll_rs : a.Kw) * SZ;
        const int e_last = s.e_first + s.nfeeds - 1;
        int goff[NGL];
        int imin, imax;
        {
            int g_lo = 0, g_hi = s.ng;
            if (!a.per) {
                if (s.c0a < 0) g_lo = (-s.c0a + 3) / 4;
                const int lim = (a.Kw - s.c0a) / 4;
                if (lim < g_hi) g_hi = lim;
            }
            imin = g_lo > lane ? (g_lo - lane + 63) / 64 : 0;
            imax = g_hi > lane ? (g_hi - lane + 63) / 64 : 0;
#pragma unroll
            for (int i = 0; i < NGL; ++i) {
                int col = s.c0a + 4 * (lane + 64 * i);
                if (a.per) col = wl_pmod(col, a.Kw);
                goff[i] = (i >= imin && i < imax) ? col * SZ : 0;
            }
        }
        // the last 1-3 coefficients of a row whose width is not a multiple of four (the whole groups stop before them):
        // lane c < NT takes cell t0 + c of both rows
        int tdst = -1, toff = 0;
        {
            const int q_hi = 2 * (s.u1 - 1) + (a.sw + SODD) / 2 + 1;
            const int t0 = s.c0a + (a.Kw - s.c0a) / 4 * 4;
            const int NTL = (!a.per && t0 <= q_hi && t0 < a.Kw) ? a.Kw - t0 : 0;
            if (lane < NTL) { tdst = (t0 + lane - s.c0a + s.dm) * 4; toff = (t0 + lane) * SZ; }
        }
        auto load = [&](int h, RowRegs<NGL> (&rrp)[PP]) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                int e = s.e_first + 2 * h + r;
                e = e < e_last ? e : e_last;
                const int row = a.per ? wl_pmod(e, a.Kh) : e;
#pragma unroll
                for (int p = 0; p < PP; ++p) {
                    const char* grow = bp[p] + (size_t)row * row_stride;
#pragma unroll
                    for (int i = 0; i < NGL; ++i) rrp[p].g[r][i] = *reinterpret_cast<const Quad4*>(grow + goff[i]);
                    rrp[p].t[r] = *reinterpret_cast<const T*>(grow + toff);
                }
            }
        };
        auto stage = [&](int hb, const RowRegs<NGL> (&rrp)[PP]) {
#pragma unroll
          for (int p = 0; p < PP; ++p) {
            const RowRegs<NGL>& rr = rrp[p];
            char* sslot = ctx.smem + a.st_off + p * a.ring + (hb & 1) * 8 * a.st_pitch + 2 * b * a.st_pitch;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                char* drow = sslot + r * a.st_pitch + lane * 16 + s.dm * 4;
#pragma unroll
                for (int i = 0; i < NGL; ++i) {
                    if (i < imin || i >= imax) continue;
                    const float v0 = (float)rr.g[r][i].x, v1 = (float)rr.g[r][i].y, v2 = (float)rr.g[r][i].z, v3 = (float)rr.g[r][i].w;
                    float* dst = reinterpret_cast<float*>(drow + i * 1024);
                    if (s.dm == 0) {
                        wl_vf4 w; w.x = v0; w.y = v1; w.z = v2; w.w = v3;
                        *reinterpret_cast<wl_vf4*>(dst) = w;
                    } else {
                        wl_f2 w; w.x = v1; w.y = v2;
                        dst[0] = v0; *reinterpret_cast<wl_f2*>(dst + 1) = w; dst[3] = v3;
                    }
                }
                if (tdst >= 0) *reinterpret_cast<float*>(sslot + r * a.st_pitch + tdst) = (float)rr.t[r];
            }
          }
        };
        // PF register sets (WL_STRIP_PF, see wl_dwt_strip.h): the coefficient rows of PF - 1 half-batches in flight
        static const int PF = WL_STRIP_PF;
        RowRegs<NGL> rr[PF][PP];
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
    struct Wave {
        wl_v2 twl[NT], twh[QMF ? 1 : NT];    // row-synthesis tap pairs of the W-low / W-high bank (shifted by one for SODD)
        wl_v2 ghl[HL], ghh[QMF ? 1 : HL];    // (g[2t], g[2t+1]) of the H-low / H-high bank
        wl_v2 lt[LAT ? HL : 1];              // LAT: (T_k, -T_k) of the column lattice (the row taps carry its gain g)
    };
    // One feed of the column lattice for both column pairs of the lane (wl_lattice.h, the synthesis recurrence): (a, b) = the
    // row-synthesised lowpass / highpass coefficient row -> (P, Q) = output rows (2m, 2m+1), each packed over the pair's two
    // columns.  Stage k leaves its q in delay slot (k + f) mod K at feed f - the slot stage k - 1 reads one feed later and
    // overwrites in place: nothing is ever moved (`rot` = f mod K is compile-time after unrolling).
    static WL_DEV void lat_feed(const Wave& R, wl_v2 (&SA)[HL], wl_v2 (&SB)[HL], int rot, wl_v2 aA, wl_v2 bA, wl_v2 aB, wl_v2 bB,
                                wl_v2& PA, wl_v2& QA, wl_v2& PB, wl_v2& QB) {
        wl_v2 qA = wl_fma_s<0, 1>(aA, R.lt[HL - 1], bA), qB = wl_fma_s<0, 1>(aB, R.lt[HL - 1], bB);   // T a - b
        wl_v2 pA = wl_fma_s<0, 0>(bA, R.lt[HL - 1], aA), pB = wl_fma_s<0, 0>(bB, R.lt[HL - 1], aB);   // a + T b
        if (HL == 1) { PA = pA; PB = pB; QA = qA; QB = qB; return; }
        SA[(HL - 1 + rot) % HL] = qA; SB[(HL - 1 + rot) % HL] = qB;
#pragma unroll
        for (int k = HL - 1; k >= 1; --k) {
            const int slot = (k - 1 + rot) % HL;
            const wl_v2 dA = SA[slot], dB = SB[slot];
            const wl_v2 nA = wl_fma_s<1, 0>(dA, R.lt[k - 1], pA), nB = wl_fma_s<1, 0>(dB, R.lt[k - 1], pB);   // p - T q'
            qA = wl_fma_s<0, 0>(pA, R.lt[k - 1], dA); qB = wl_fma_s<0, 0>(pB, R.lt[k - 1], dB);             // q' + T p
            if (k > 1) { SA[slot] = qA; SB[slot] = qB; }
            pA = nA; pB = nB;
        }
        PA = pA; PB = pB; QA = qA; QB = qB;
    }
    // acc += q(pair) * s.x / s.y, q(pair) = (pair.y, pair.x) with one half negated: NEGLO -> (-pair.y, pair.x), else (pair.y, -pair.x)
    template <int Y, int NEGLO> static WL_DEV void fma_q(wl_v2& acc, wl_v2 pair, wl_v2 sv) {
#if defined(__HIPCC__)
        if (Y) {
            if (NEGLO) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "+v"(acc) : "s"(pair), "v"(sv));
            else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "+v"(acc) : "s"(pair), "v"(sv));
        } else {
            if (NEGLO) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0]" : "+v"(acc) : "s"(pair), "v"(sv));
            else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "+v"(acc) : "s"(pair), "v"(sv));
        }
#else
        const float c = Y ? sv.y : sv.x;
        const float q0 = NEGLO ? -pair.y : pair.y, q1 = NEGLO ? pair.x : -pair.x;
        acc.x = __builtin_fmaf(q0, c, acc.x); acc.y = __builtin_fmaf(q1, c, acc.y);
#endif
    }
    // row synthesis: twh[j] = (-twl[NT-1-j].y, twl[NT-1-j].x) for the shifted pairs (SODD), (twl[HL-1-j].y, -twl[HL-1-j].x) else
    template <int r, int J> static WL_DEV void fma_cell_q(wl_v2& acc, const Wave& R, const wl_v2 (&v)[NC2]) {
        fma_q<(r & 1), SODD>(acc, R.twl[NT - 1 - J], v[r / 2]);
    }
    // the NT + 1 coefficients of one staged source row, as (even, odd) pairs
    static WL_DEV void load_row(const char* p, wl_v2 (&v)[NC2]) {
#pragma unroll
        for (int i = 0; i < NC2; ++i) {
            const wl_f2 t = *reinterpret_cast<const wl_f2*>(p + 8 * i);
            v[i] = wl_v2{t.x, t.y};
        }
    }
    // acc += tap * cell r of the lane's coefficients (r compile-time: the half is picked by op_sel)
    template <int r> static WL_DEV void fma_cell(wl_v2& acc, wl_v2 tap, const wl_v2 (&v)[NC2]) {
        if (r & 1) wl_pk_fma_y(acc, tap, v[r / 2]); else wl_pk_fma_x(acc, tap, v[r / 2]);
    }
    template <int r> static WL_DEV wl_v2 mul_cell(wl_v2 tap, const wl_v2 (&v)[NC2]) {
        return (r & 1) ? wl_pk_mul_y(tap, v[r / 2]) : wl_pk_mul_x(tap, v[r / 2]);
    }
    // polyphase row synthesis of one coefficient row for pair P (0: cells NT-1-j, 1: cells NT-j): lo / hi = the W-low and
    // the W-high source (ll, hl -> a;  lh, hh -> b)
    template <int P, int J> struct RowSyn {
        static WL_DEV void run(const Wave& R, const wl_v2 (&lo)[NC2], const wl_v2 (&hi)[NC2], wl_v2& acc) {
            fma_cell<NT - 1 - J + P>(acc, R.twl[J], lo);
            if (QMF) fma_cell_q<NT - 1 - J + P, J>(acc, R, hi); else fma_cell<NT - 1 - J + P>(acc, R.twh[QMF ? 0 : J], hi);
            RowSyn<P, J + 1>::run(R, lo, hi, acc);
        }
    };
    template <int P> struct RowSyn<P, NT> {
        static WL_DEV void run(const Wave&, const wl_v2 (&)[NC2], const wl_v2 (&)[NC2], wl_v2&) {}
    };
    template <int P>
    static WL_DEV wl_v2 row_syn(const Wave& R, const wl_v2 (&lo)[NC2], const wl_v2 (&hi)[NC2]) {
        wl_v2 acc = mul_cell<NT - 1 + P>(R.twl[0], lo);
        if (QMF) fma_cell_q<NT - 1 + P, 0>(acc, R, hi); else fma_cell<NT - 1 + P>(acc, R.twh[0], hi);
        RowSyn<P, 1>::run(R, lo, hi, acc);
        return acc;
    }
    // The four row syntheses of a feed - pairs A (P = 0) and B (P = 1), a from (ll, hl), b from (lh, hh) - with their accumulator
    // chains interleaved tap by tap: a packed FMA that reads the result of the one two instructions before it waits a cycle (the
    // compiler put an s_nop behind every other FMA of the two-chain form: 270 in the four half-batches of the 16-tap loop).
    template <int J, int U = 0> struct RowSyn4 {
        static WL_DEV void run(const Wave& R, const wl_v2 (&cll)[NC2], const wl_v2 (&clh)[NC2], const wl_v2 (&chl)[NC2], const wl_v2 (&chh)[NC2],
                               wl_v2& aA, wl_v2& bA, wl_v2& aB, wl_v2& bB) {
            fma_cell<NT - 1 - J>(aA, R.twl[J], cll); fma_cell<NT - 1 - J>(bA, R.twl[J], clh);
            fma_cell<NT - J>(aB, R.twl[J], cll); fma_cell<NT - J>(bB, R.twl[J], clh);
            if (QMF) {
                fma_cell_q<NT - 1 - J, J>(aA, R, chl); fma_cell_q<NT - 1 - J, J>(bA, R, chh);
                fma_cell_q<NT - J, J>(aB, R, chl); fma_cell_q<NT - J, J>(bB, R, chh);
            } else {
                fma_cell<NT - 1 - J>(aA, R.twh[QMF ? 0 : J], chl); fma_cell<NT - 1 - J>(bA, R.twh[QMF ? 0 : J], chh);
                fma_cell<NT - J>(aB, R.twh[QMF ? 0 : J], chl); fma_cell<NT - J>(bB, R.twh[QMF ? 0 : J], chh);
            }
            RowSyn4<J + 1, U>::run(R, cll, clh, chl, chh, aA, bA, aB, bB);
        }
    };
    template <int U> struct RowSyn4<NT, U> {
        static WL_DEV void run(const Wave&, const wl_v2 (&)[NC2], const wl_v2 (&)[NC2], const wl_v2 (&)[NC2], const wl_v2 (&)[NC2], wl_v2&, wl_v2&, wl_v2&, wl_v2&) {}
    };
    static WL_DEV void row_syn4(const Wave& R, const wl_v2 (&cll)[NC2], const wl_v2 (&clh)[NC2], const wl_v2 (&chl)[NC2], const wl_v2 (&chh)[NC2],
                                wl_v2& aA, wl_v2& bA, wl_v2& aB, wl_v2& bB) {
        aA = mul_cell<NT - 1>(R.twl[0], cll); bA = mul_cell<NT - 1>(R.twl[0], clh);
        aB = mul_cell<NT>(R.twl[0], cll); bB = mul_cell<NT>(R.twl[0], clh);
        if (QMF) {
            fma_cell_q<NT - 1, 0>(aA, R, chl); fma_cell_q<NT - 1, 0>(bA, R, chh);
            fma_cell_q<NT, 0>(aB, R, chl); fma_cell_q<NT, 0>(bB, R, chh);
        } else {
            fma_cell<NT - 1>(aA, R.twh[0], chl); fma_cell<NT - 1>(bA, R.twh[0], chh);
            fma_cell<NT>(aB, R.twh[0], chl); fma_cell<NT>(bB, R.twh[0], chh);
        }
        RowSyn4<1>::run(R, cll, clh, chl, chh, aA, bA, aB, bB);
    }
    // polyphase column synthesis from the circular window whose NEWEST row sits in slot `newest`:
    // y0 = (z-row 2m, 2m+1) of the pair's even column, y1 of its odd column
    static WL_DEV void col_syn(const Wave& R, const wl_v2 (&wa)[LW], const wl_v2 (&wb)[LW], int newest, wl_v2& y0, wl_v2& y1) {
        y0 = wl_pk_mul_x(R.ghl[0], wa[newest % LW]); y1 = wl_pk_mul_y(R.ghl[0], wa[newest % LW]);
        if (QMF) {   // ghh[t] = (ghl[HL-1-t].y, -ghl[HL-1-t].x)
            fma_q<0, 0>(y0, R.ghl[HL - 1], wb[newest % LW]);
            fma_q<1, 0>(y1, R.ghl[HL - 1], wb[newest % LW]);
        } else {
            wl_pk_fma_x(y0, R.ghh[0], wb[newest % LW]);
            wl_pk_fma_y(y1, R.ghh[0], wb[newest % LW]);
        }
#pragma unroll
        for (int t = 1; t < HL; ++t) {
            wl_pk_fma_x(y0, R.ghl[t], wa[(newest + LW - t) % LW]);
            wl_pk_fma_y(y1, R.ghl[t], wa[(newest + LW - t) % LW]);
            if (QMF) {
                fma_q<0, 0>(y0, R.ghl[HL - 1 - t], wb[(newest + LW - t) % LW]);
                fma_q<1, 0>(y1, R.ghl[HL - 1 - t], wb[(newest + LW - t) % LW]);
            } else {
                wl_pk_fma_x(y0, R.ghh[QMF ? 0 : t], wb[(newest + LW - t) % LW]);
                wl_pk_fma_y(y1, R.ghh[QMF ? 0 : t], wb[(newest + LW - t) % LW]);
            }
        }
    }

    // cw = the wave's run of 64 lane units inside the strip, sub = which of the workgroup's planes (its staged ring)
    static WL_DEV void compute(const Args& a, const Strip& s, const WlCtx& ctx, int64_t plane, int cw, int lane, int sub) {
        const int u = s.u0 + 64 * cw + lane;                  // lane unit: output columns 4u .. 4u+3
        const bool active = u < s.u1;
        Wave R;
        const float gsc = LAT ? a.lat[1] : 1.f;               // the lattice's gain rides on the row-synthesis taps
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            // SODD: pairs (g[2j-1], g[2j]) with g[-1] = g[L] = 0
            const int i0 = 2 * j - SODD, i1 = 2 * j + 1 - SODD;
            const float l0 = i0 >= 0 && i0 < LT ? a.g_w_lo[i0 >= 0 && i0 < LT ? i0 : 0] : 0.f;
            const float l1 = i1 >= 0 && i1 < LT ? a.g_w_lo[i1 >= 0 && i1 < LT ? i1 : 0] : 0.f;
            const float h0 = i0 >= 0 && i0 < LT ? a.g_w_hi[i0 >= 0 && i0 < LT ? i0 : 0] : 0.f;
            const float h1 = i1 >= 0 && i1 < LT ? a.g_w_hi[i1 >= 0 && i1 < LT ? i1 : 0] : 0.f;
            R.twl[j] = wl_uniform_v2(wl_v2{l0 * gsc, l1 * gsc});
            if (!QMF) R.twh[QMF ? 0 : j] = wl_uniform_v2(wl_v2{h0, h1});
        }
        if (LAT) {
#pragma unroll
            for (int k = 0; k < HL; ++k) R.lt[LAT ? k : 0] = wl_uniform_v2(wl_v2{a.lat[2 + k], -a.lat[2 + k]});
        }
#pragma unroll
        for (int t = 0; t < HL; ++t) {
            if (!LAT) R.ghl[t] = wl_uniform_v2(wl_v2{a.g_h_lo[2 * t], a.g_h_lo[2 * t + 1]});
            if (!QMF) R.ghh[QMF ? 0 : t] = wl_uniform_v2(wl_v2{a.g_h_hi[2 * t], a.g_h_hi[2 * t + 1]});
        }
        char* const yp = reinterpret_cast<char*>(a.y + (size_t)plane * a.OH * a.OW);
        const unsigned rowb = (unsigned)a.OW * SZ;
        const unsigned colb = (unsigned)(4 * u) * SZ;
        const int ncols = active ? (a.OW - 4 * u < 4 ? a.OW - 4 * u : 4) : 0;   // columns of this lane inside the output
        const int soff = s.lane_off + 8 * (active ? u - s.u0 : 0);
        const int N2 = 2 * a.Kh;
        static const int NW = LAT ? 1 : LW;
        wl_v2 waA[NW], wbA[NW], waB[NW], wbB[NW];              // circular windows: (a, b) of the two pairs (LAT: the new row only)
        wl_v2 SA[HL], SB[HL];                                  // LAT: delay slots of the two pairs' lattices
#pragma unroll
        for (int t = 0; t < NW; ++t) waA[t] = wbA[t] = waB[t] = wbB[t] = wl_v2{0.f, 0.f};
#pragma unroll
        for (int t = 0; t < HL; ++t) SA[t] = SB[t] = wl_v2{0.f, 0.f};
        char* const smem = ctx.smem;
        int fed = 0;
        for (int hb0 = 0; hb0 < s.nhb; hb0 += PERIOD) {
#pragma unroll
            for (int ph = 0; ph < PERIOD; ++ph) {
                const int hb = hb0 + ph;
                if (hb >= s.nhb) break;
                ctx.sync();
                const int left = s.nfeeds - fed;
                const int n = left > 2 ? 2 : left;
                const char* slot = smem + a.st_off + sub * a.ring + (hb & 1) * 8 * a.st_pitch + soff;
                if (active) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if (i < n) {
                            wl_v2 cll[NC2], clh[NC2], chl[NC2], chh[NC2];
                            load_row(slot + (0 + i) * a.st_pitch, cll);
                            load_row(slot + (2 + i) * a.st_pitch, clh);
                            load_row(slot + (4 + i) * a.st_pitch, chl);
                            load_row(slot + (6 + i) * a.st_pitch, chh);
                            const int w = LAT ? 0 : (2 * ph + i) % LW;
                            row_syn4(R, cll, clh, chl, chh, waA[w], wbA[w], waB[w], wbB[w]);
                            wl_v2 y0A, y1A, y0B, y1B;
                            if constexpr (LAT != 0) {   // (every feed: the lattice's state; the first K - 1 outputs of a segment are the warm-up)
                                // the lattice delivers (row 2m, row 2m+1), each packed over the pair's two columns; the direct form
                                // packs the other way round, y0 = (row 2m, row 2m+1) of the even column: a renaming of registers
                                wl_v2 PA, QA, PB, QB;
                                lat_feed(R, SA, SB, (2 * ph + i) % HL, waA[0], wbA[0], waB[0], wbB[0], PA, QA, PB, QB);
                                y0A = wl_v2{PA.x, QA.x}; y1A = wl_v2{PA.y, QA.y};
                                y0B = wl_v2{PB.x, QB.x}; y1B = wl_v2{PB.y, QB.y};
                            }
                            if (fed + i >= HL - 1) {
                                if constexpr (!LAT) {
                                    col_syn(R, waA, wbA, w, y0A, y1A);
                                    col_syn(R, waB, wbB, w, y0B, y1B);
                                }
                                // z-rows 2m, 2m+1 of feed m -> output rows 2m - sh (+1), rolled under periodization
                                const int m = s.e_first + fed + i;
                                int p0 = 2 * m - a.sh, p1 = p0 + 1;
                                if (a.per) { p0 = p0 < 0 ? p0 + N2 : p0; p1 = p1 < 0 ? p1 + N2 : (p1 >= N2 ? p1 - N2 : p1); }
                                store_row(a, yp + (unsigned)p0 * rowb + colb, p0, ncols, y0A.x, y1A.x, y0B.x, y1B.x);
                                store_row(a, yp + (unsigned)p1 * rowb + colb, p1, ncols, y0A.y, y1A.y, y0B.y, y1B.y);
                            }
                        }
                    }
                }
                fed += n;
            }
        }
    }
    static WL_DEV void store_row(const Args& a, char* p, int row, int ncols, float v0, float v1, float v2, float v3) {
        if ((unsigned)row >= (unsigned)a.OH) return;          // cropped away (AFB2D.backward's out_hw)
        if (a.quad_ok) {
            typedef T Vec4 __attribute__((ext_vector_type(4)));
            Vec4 v = {(T)v0, (T)v1, (T)v2, (T)v3};
            *reinterpret_cast<Vec4*>(p) = v;
        } else {
            T* o = reinterpret_cast<T*>(p);
            if (ncols > 0) o[0] = (T)v0;
            if (ncols > 1) o[1] = (T)v1;
            if (ncols > 2) o[2] = (T)v2;
            if (ncols > 3) o[3] = (T)v3;
        }
    }

    static WL_DEV void run(const Args& a, const WlCtx& ctx) {
        const int tid = ctx.tid;
        const int wave = wl_uniform(tid >> 6), lane = tid & 63;
        if (a.guard) {   // the relation the QMF / LAT variant relies on, checked against the taps as they are now (uniform: before any barrier)
            const bool holds = a.lat ? *reinterpret_cast<const unsigned*>(a.lat) == WL_LAT_OK   // (WlTapPrep's verdict, read by both launches)
                                     : wl_taps_qmf(a.g_w_lo, a.g_w_hi, LT) && wl_taps_qmf(a.g_h_lo, a.g_h_hi, LT);
            if (!wl_guard_pass(a.guard, holds)) return;
        }
        // (a narrow strip - one or two compute waves' worth - takes pp = 4 / 2 planes per workgroup: WlAfbStrip::run)
        const int64_t lbid = wl_xcd_remap(ctx.bid, a.nblocks);
        const int per_plane = a.nstrips * a.nseg;
        const int64_t pgroup = lbid / per_plane;
        const int64_t plane = pgroup * a.pp;
        const int rem = (int)(lbid - pgroup * per_plane);
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
            const int b = wave - WL_STRIP_CWAVES;
            if (WL_STRIP_DIRECT) {
                const int ngl = (s.ng + 63) >> 6;
                // (the launcher: ngl = 1 with four planes, <= 2 with two - what strips of one / two compute waves need; a wave holds
                // 2 rows x ngl x pp groups x WL_STRIP_PF sets in registers, 128 of them at <2, 4> in float32)
                if (a.pp == 4) stager_direct<1, 4>(a, s, ctx, plane, lane, b);
                else if (a.pp == 2) {
                    if (ngl == 1) stager_direct<1, 2>(a, s, ctx, plane, lane, b); else stager_direct<2, 2>(a, s, ctx, plane, lane, b);
                } else switch (ngl) {
                    case 1: stager_direct<1, 1>(a, s, ctx, plane, lane, b); break;
                    case 2: stager_direct<2, 1>(a, s, ctx, plane, lane, b); break;
                    case 3: stager_direct<3, 1>(a, s, ctx, plane, lane, b); break;
                    case 4: stager_direct<4, 1>(a, s, ctx, plane, lane, b); break;
                    case 5: stager_direct<5, 1>(a, s, ctx, plane, lane, b); break;
                    default: stager_direct<6, 1>(a, s, ctx, plane, lane, b); break;
                }
            } else stager(a, s, ctx, plane, lane, b);
            return;
        }
        const int nact = WL_STRIP_CWAVES / a.pp;              // compute waves per plane
        const int sub = wave / nact, cw = wave - sub * nact;
        if (plane + sub < a.NC && 64 * cw < s.u1 - s.u0) {
            compute(a, s, ctx, plane + sub, cw, lane, sub);
        } else {
            for (int hb = 0; hb < s.nhb; ++hb) ctx.sync();
        }
    }
};
