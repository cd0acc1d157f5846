// Streaming multi-level 2-D DWT analysis: one workgroup owns one whole (n,c) plane and marches down it.
//
// Why: a per-level tile kernel moves 1.32x the compulsory bytes of a 3-level transform (LL_1, LL_2 round trips), pays
// two small launches, and its 259-float band rows leave in 256-byte pieces that never line up with cache lines.  Here
//   * every input row is read from HBM exactly once, as whole 1 KiB wave-wide LDS-DMA loads (global_load_lds_dwordx4)
//     issued D half-batches ahead by a dedicated loader wave - no VGPRs, no LDS write instructions, no halo re-reads;
//   * the boundary extension along W is materialised once per row as a few halo cells next to the row in LDS (the
//     loader patches them after the row has landed; for LL_j rows the producing lane writes them), so EVERY lane
//     reads its L row-filter samples as one base address + constant offsets;
//   * a lane owns ONE output column of its level and keeps the L-row sliding window of that column's row-filtered
//     (lo,hi) pair in registers: per pair of new source rows it runs the row
//     filter (samples straight from the LDS ring) and the column filter (registers only) - one LDS crossing per
//     sample, no intermediate planes, one barrier per four input rows;
//   * LL_j rows go to a small LDS ring that the waves of level j+1 consume a few rows behind - LL_1 .. LL_{J-1}
//     never touch HBM; HBM traffic = x in + yl, yh[j] out = the algorithmic minimum of SURVEY.md 8(d);
//   * every band row of every level is written by consecutive lanes of consecutive waves within one half-batch, so
//     the partial cache lines at its ends are completed in the same L2 a few hundred cycles later.
// The kernel is bound by INSTRUCTION ISSUE, not by arithmetic: a SIMD issues about one instruction per four cycles
// for all of its waves together (rocprofv3: SQ_ACTIVE_INST_ANY ~ 90 % of the kernel's cycles in the first version).
// Hence: v_pk_fma_f32 on (lo,hi) pairs with the broadcast sample picked by op_sel (half the instructions of scalar
// FMAs), taps in scalar registers, the schedule as a host-built table instead of per-wave scalar arithmetic (that
// alone was 57 % of all instructions), and the roles dealt out so that every SIMD carries the same load.
// Roles are per WAVE (wave-uniform branches, each role has its own lean body): level-1 / level-2 / level-3 compute
// waves and the loaders.  All waves follow the same deterministic schedule: per half-batch level 1 consumes 4
// extended input rows (2 feeds), level j+1 consumes whatever LL_j rows were complete at the barrier.  The launcher
// simulates it on the host (WlRowsSched), refuses geometries whose rings would be overrun, and hands the result to
// the kernel as a table.
//
// Restates DWTForward.forward's level loop (reference dwt/transform2d.py:63-74) = J x AFB2D.forward
// (dwt/lowlevel.py:336-347) = 2J x afb1d (:91-172): filter along W, then along H, per level.
#pragma once
#include "wl_common.h"

#define WL_ROWS_MAXLEV 3
#ifndef WL_ROWS_WAVES
#define WL_ROWS_WAVES 12
#endif
#ifndef WL_ROWS_LOADERS
#define WL_ROWS_LOADERS 2       // loader waves: each issues the DMA of 4 / WL_ROWS_LOADERS rows of a half-batch (1, 2 or 4)
#endif
#define WL_ROWS_MAXHB 768       // half-batches one schedule table (kernel argument) can hold
// WL_ROWS_ABLATE (tools/build_ab.sh builds only, never defined in the product): 1 = no global stores, 2 = no DMA
// loads, 8 = in-kernel cycle counters written over a few LL samples
#ifndef WL_ROWS_ABLATE
#define WL_ROWS_ABLATE 0
#endif
#if (WL_ROWS_ABLATE & 8) && defined(__HIPCC__)
#define WL_TICK() __builtin_readcyclecounter()
#else
#define WL_TICK() 0ull
#endif
#ifndef WL_ROWS_DEPTH
#define WL_ROWS_DEPTH 3
#endif

// acc += tap * s.x  /  acc += tap * s.y  on (lo,hi) pairs: ONE v_pk_fma_f32, the broadcast half chosen by op_sel
#if defined(__HIPCC__)
WL_DEV void wl_pk_fma_x(wl_v2& acc, wl_v2 tap, wl_v2 s) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "s"(tap), "v"(s));
}
WL_DEV void wl_pk_fma_y(wl_v2& acc, wl_v2 tap, wl_v2 s) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(tap), "v"(s));
}
WL_DEV wl_v2 wl_pk_mul_x(wl_v2 tap, wl_v2 s) {
    wl_v2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "s"(tap), "v"(s));
    return r;
}
WL_DEV wl_v2 wl_pk_mul_y(wl_v2 tap, wl_v2 s) {
    wl_v2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "s"(tap), "v"(s));
    return r;
}
WL_DEV wl_v2 wl_uniform_v2(wl_v2 v) { return wl_v2{wl_uniform_f(v.x), wl_uniform_f(v.y)}; }
#else
inline void wl_pk_fma_x(wl_v2& acc, wl_v2 tap, wl_v2 s) { acc.x = __builtin_fmaf(tap.x, s.x, acc.x); acc.y = __builtin_fmaf(tap.y, s.x, acc.y); }
inline void wl_pk_fma_y(wl_v2& acc, wl_v2 tap, wl_v2 s) { acc.x = __builtin_fmaf(tap.x, s.y, acc.x); acc.y = __builtin_fmaf(tap.y, s.y, acc.y); }
inline wl_v2 wl_pk_mul_x(wl_v2 tap, wl_v2 s) { return wl_v2{tap.x * s.x, tap.y * s.x}; }
inline wl_v2 wl_pk_mul_y(wl_v2 tap, wl_v2 s) { return wl_v2{tap.x * s.y, tap.y * s.y}; }
inline wl_v2 wl_uniform_v2(wl_v2 v) { return v; }
#endif

#include "wl_lattice.h"   // the lattice form of the column pass (LAT variants), the QMF form of a tap pair

struct WlRowsLevel {
    int Hs, Ws;         // source rows / cols of this level
    int Kh, Kw;         // output rows / cols
    int ring_off;       // LDS byte offset of this level's SOURCE ring (level 0: the input ring)
    int ring_pitch;     // bytes per ring row: [left halo | Ws samples | right halo]
    int pad;            // byte offset of sample 0 inside a ring row (multiple of 16)
    int hl, hr;         // halo cells left / right of the samples (boundary extension along W)
    int nwaves;         // 64-column chunks of this level (each chunk is one compute wave)
    unsigned ring_magic;  // NP2 instantiations: floor(2^32 / ring_rows) - row r sits in slot r mod ring_rows, any ring_rows (wl_ring_slot)
    int ring_rows;      // levels >= 2: rows of this level's SOURCE ring (power of two, row r in r & (ring_rows - 1)); round 6: per level -
                        // the smallest the level's own schedule does not overrun (one size for all rings put 16-tap symmetric pyramids of
                        // three levels over the 80 KiB of two workgroups per CU)
};

// One SEGMENT of a plane: the feeds [f0, fend) of every level and the output rows [own_lo, own_hi) it stores.  A plane
// is either processed whole by one workgroup or cut once (at a row of the coarsest level) into a top and a bottom
// segment, so that the workgroups of a launch add up to whole rounds of the chip: with 384 planes on 256 CUs and two
// workgroups per CU, 256 whole planes + 128 planes in halves give every CU one whole and one half plane instead of
// leaving half the CUs idle for the second half of the launch.  The cut costs a re-computed halo (L-2 rows per level).
struct WlRowsSeg {
    int nhb;                        // half-batches (= barriers of the main loop) until every level has finished
    int f0[WL_ROWS_MAXLEV];         // first feed of each level (= first row it produces)
    int fend[WL_ROWS_MAXLEV];       // one past its last feed
    int own_lo[WL_ROWS_MAXLEV];     // rows of each level this segment stores to HBM
    int own_hi[WL_ROWS_MAXLEV];
    // the schedule, simulated on the host (WlRowsSched): one byte per half-batch, bits 2j+1:2j = feeds of level j.
    // Looked up by every wave instead of being recomputed: the scalar instructions of the schedule arithmetic were
    // 57 % of all instructions issued by the first version of this kernel (rocprofv3 SQ_INSTS_SALU).
    unsigned sched[WL_ROWS_MAXHB / 4];
};

template <typename T>
struct WlRowsArgs {
    const T* x;                    // (NC, H, W) through x_ps / x_rs
    T* ll;                         // LL of the last fused level, (NC, Kh, Kw) through ll_ps / ll_rs
    T* yh[WL_ROWS_MAXLEV];         // (NC, 3, Kh_j, Kw_j) dense
    const float* h_w_lo;
    const float* h_w_hi;
    const float* h_h_lo;
    const float* h_h_hi;
    int64_t NC, x_ps, ll_ps;
    int x_rs, ll_rs;
    int nlev, ext, base;
    int nwhole;         // workgroups [0, nwhole) take whole planes; pairs of the remaining ones the two halves of a plane
    int zero_off;       // LDS byte offset of an all-zero row (zero padding above / below the plane)
    int lds_bytes;
    // role of every wave: level (0..nlev-1) and first column of its 64-column chunk; -1 = loader (col0 = its index), -2 = spare.  Waves w
    // and w+4 share a SIMD, so the launcher deals the roles out for equal instruction load per SIMD.
    signed char role_level[WL_ROWS_WAVES];
    short role_col0[WL_ROWS_WAVES];
    // narrow planes: a workgroup owns pp consecutive planes, every one with its own compute waves, loaders and lds_plane bytes
    // of rings (role_sub = which of them a wave serves); the schedule is the same for all of them (see wl_idwt_rows.h)
    signed char role_sub[WL_ROWS_WAVES];
    int pp, lds_plane;
    WlRowsLevel g[WL_ROWS_MAXLEV];
    WlRowsSeg seg[3];   // 0: whole plane, 1: top half, 2: bottom half
    int guard;          // tap-relation guard (wl_common.h): 1 = run only if the row and the column banks hold the same taps (the SAME
                        // variant), 2 = only if they do not (its armed two-bank fallback), 0 = no check
    const float* lat;   // LAT variant and its fallback: WlTapPrep's verdict (+ the column lattice) in device scratch (wl_lattice.h): the
                        // guard reads the verdict word instead of comparing taps
};

// The schedule the LAUNCHER steps through to fill WlRowsSeg::sched: fed[j] = next feed of level j.  A feed is one pair
// of extended source rows (2f+base, 2f+base+1) and completes output row f - (L-2)/2 once the window is full, i.e.
// from the (L-2)/2-th feed of the segment on.
struct WlRowsSched {
    int fed[WL_ROWS_MAXLEV];
    WL_HD void init(const WlRowsSeg& sg) { for (int j = 0; j < WL_ROWS_MAXLEV; ++j) fed[j] = sg.f0[j]; }
    // the LL row of the level above that extended row e of a level >= 2 reads.  Periodization: row e itself - the level above
    // COMPUTES the rows above / below its plane (rows -hl .. -1 are rows Hs - hl .. Hs - 1 of its periodic output, produced a
    // second time at the start of the segment: its feeds start at a negative number, WlRowsSeg::f0)
    static WL_HD int src_row(int e, int Hs, int ext) { return ext == WL_EXT_PER ? e : wl_ext1(e, Hs, ext); }
    // rows [sg.f0[j], avail(j)) of level j are complete
    WL_HD int avail(const WlRowsSeg& sg, int j, int LT) const {
        const int warm = (LT - 2) / 2;
        return fed[j] - sg.f0[j] > warm ? fed[j] - warm : sg.f0[j];
    }
    // feeds level j runs in the half-batch that starts now; only looks at the state as of the barrier
    template <typename A> WL_HD int feeds_now(const A& a, const WlRowsSeg& sg, int j, int LT) const {
        const int left = sg.fend[j] - fed[j];
        if (j == 0) return left > 2 ? 2 : left;
        const int av = avail(sg, j - 1, LT);
        const int Hs = a.g[j].Hs;
        const int e = a.base + 2 * fed[j];
        const bool ok0 = left > 0 && src_row(e, Hs, a.ext) < av && src_row(e + 1, Hs, a.ext) < av;
        const bool ok1 = left > 1 && src_row(e + 2, Hs, a.ext) < av && src_row(e + 3, Hs, a.ext) < av;
        return ok0 ? (ok1 ? 2 : 1) : 0;
    }
};

// SAME = 1: the row and the column banks are the same taps (the launcher saw the same device buffers for both axes - what a
// transform built from ONE wavelet passes): one set of tap pairs in the scalar file instead of two.  (At 12 taps two sets are
// 96 scalar registers: the kernel spilled them, 31-35 SGPRs and 16-28 bytes of scratch per lane.)
#ifndef WL_ROWS_SAME_MIN
#define WL_ROWS_SAME_MIN 10            // tap counts from which the one-bank variant exists (below, two banks fit the scalar file)
#endif
// LAT = 1 (round 5, with SAME = 1): one ORTHOGONAL bank for both axes - the row pass in the QMF form (the lowpass bank only, L/2 tap
// pairs in scalar registers), the column pass as the lattice of wl_lattice.h: L packed FMAs and L/2 - 1 delayed values per output
// row where the direct form needs 2L FMAs and an L-row window that is moved along every feed.  The kernel is bound by the
// instructions its waves issue: 12 taps shed a third of them, and 14-20 taps - which the direct form cannot hold in 80
// registers - get a fused multi-level kernel at all.
// ODD = 1 (round 6): `base` is odd - periodization with L % 4 == 0, base = 1 - L/2 (db2, db4, db6, db8, db10) - so a lane's L row-filter
// samples start on an ODD cell of the ring row.  The lane then reads L/2 + 1 aligned (even, odd) pairs from one cell earlier and
// meets tap t with the OTHER half of a pair (even taps the .y of pair t/2, odd taps the .x of pair (t+1)/2): the same L packed
// FMAs, one more 8-byte LDS read per row, no unaligned access.  The unused first and last cells are halo cells like the others.
// NP2 = 1 (round 6): the LL rings have EXACTLY the rows the simulated schedule needs, not the next power of two - a symmetric / reflect pyramid
// needs L - 1 resident rows per ring for the mirrored rows above the plane plus what the producer runs ahead: 18-24 rows where the power of two is
// 32, and three levels of a 12- to 16-tap filter on 512 columns are 83-97 KiB with 32-row rings (two workgroups per CU: 80).  The slot of a row is
// r mod rows by a multiply-high (two scalar instructions more per row than the mask): instantiated for the long filters only.
WL_HD int wl_ring_slot(int r, int rows, unsigned magic) {
    const unsigned q = (unsigned)(((unsigned long long)(unsigned)r * magic) >> 32);     // floor(r / rows) or one less
    const int sl = r - (int)q * rows;
    return sl >= rows ? sl - rows : sl;
}
template <typename T, int LT, int PPR, int D = WL_ROWS_DEPTH, int SAME = 0, int LAT = 0, int ODD = 0, int NP2 = 0>
struct WlAfbRows {
    typedef WlRowsArgs<T> Args;
    static_assert(!LAT || SAME, "the lattice variant holds one bank");
    static const int NP = LT / 2 + ODD;    // (even, odd) sample pairs a lane reads per ring row
    static const int KL = LT / 2;          // rotations of the lattice
    static const int kThreads = 64 * WL_ROWS_WAVES;
    static const int kMinWaves = 6;        // two workgroups per CU: 24 waves on 4 SIMDs
    static const int LROWS = 4 / WL_ROWS_LOADERS;   // rows of a half-batch one loader wave is responsible for
    static const int NL = LROWS * PPR;     // DMA instructions per half-batch and loader wave (PPR pieces of 1 KiB per row)
    static const int WARM = (LT - 2) / 2;  // feeds that only fill the window
    static const int NSLOT = D + 1;        // half-batch slots of the input ring
    static const int SZ = (int)sizeof(T);
    static_assert((D - 1) * NL < 64, "prefetch distance exceeds the vmcnt range");

    // ---- loader wave ------------------------------------------------------------------------------------------
    static WL_DEV void loader(const Args& a, const WlRowsSeg& sg, const WlCtx& ctx, int64_t plane, int lane, int lidx) {
        const WlRowsLevel& g = a.g[0];
        const char* xp = reinterpret_cast<const char*>(a.x + (size_t)plane * a.x_ps);
        const int row_bytes = g.Ws * SZ;
        const int row_stride = a.x_rs * SZ;
        const int nhb0 = (sg.fend[0] - sg.f0[0] + 1) / 2;        // half-batches in which level 1 runs
        const int e_last = a.base + 2 * sg.fend[0] - 1;          // last extended row level 1 consumes
        const int ring = g.ring_off, pitch = g.ring_pitch, pad = g.pad, Hs = g.Hs, ext = a.ext;
        const int base = a.base + 2 * sg.f0[0];                  // extended row of the segment's first feed
        const int rfirst = lidx * LROWS;   // this loader's rows of every slot: rfirst .. rfirst + LROWS - 1
        // halo cells of those rows: item = (row r, cell c); a lane handles items lane and lane + 64
        const int NH = g.hl + g.hr;
        int hdst[2], hsrc[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int it = lane + 64 * u;
            hdst[u] = hsrc[u] = -1;
            // (zero mode: the halo cells stay zero - unless the row ends inside its last 16-byte piece, whose tail then lands in the
            // right halo cells: they are cleared like the mirrored cells of the other modes are written, wl_ext = -1)
            if ((ext != WL_EXT_ZERO || row_bytes % 16 != 0) && it < LROWS * NH) {
                const int r = rfirst + it / NH, c = it % NH;
                const int e = c < g.hl ? c - g.hl : g.Ws + (c - g.hl);
                const int s = wl_ext(e, g.Ws, ext);
                hdst[u] = r * pitch + pad + e * SZ;
                hsrc[u] = s < 0 ? -1 : r * pitch + pad + s * SZ;
            }
        }
        auto issue = [&](int h) {
            const int slot = ring + (h % NSLOT) * 4 * pitch + pad;
#pragma unroll
            for (int rr = 0; rr < LROWS; ++rr) {
                const int r = rfirst + rr;
                int e = base + 4 * h + r;
                e = e < e_last ? e : e_last;
                int src = e;
                if ((unsigned)e >= (unsigned)Hs) {               // above / below the plane (rare): full extension rule
                    src = wl_ext(e, Hs, ext);
                    src = src < 0 ? 0 : src;                     // zero rows: a dummy row keeps the DMA count exact
                }
                const char* grow = xp + (size_t)src * row_stride;
#pragma unroll
                for (int p = 0; p < PPR; ++p) {
                    const int byte = (p * 64 + lane) * 16;
                    if (!(WL_ROWS_ABLATE & 2)) wl_dma16(ctx, (unsigned)(slot + r * pitch + p * 1024), grow + byte, byte < row_bytes);
                }
            }
        };
        for (int h = 0; h < D; ++h) issue(h);
        unsigned long long tw = 0, tb = 0, ti = 0;
        for (int hb = 0; hb < sg.nhb; ++hb) {
            const unsigned long long c0 = WL_TICK();
            if (hb < nhb0) {
                if (!(WL_ROWS_ABLATE & 2)) wl_wait_vm<(D - 1) * NL>();   // the rows of this half-batch have landed
                char* slot = ctx.smem + ring + (hb % NSLOT) * 4 * pitch;
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (hdst[u] >= 0) *reinterpret_cast<T*>(slot + hdst[u]) = hsrc[u] < 0 ? (T)0 : *reinterpret_cast<const T*>(slot + hsrc[u]);
            }
            const unsigned long long c1 = WL_TICK();
            ctx.sync();
            const unsigned long long c2 = WL_TICK();
            if (hb < nhb0) issue(hb + D);                // its slot was consumed in half-batch hb-1
            const unsigned long long c3 = WL_TICK();
            tw += c1 - c0; tb += c2 - c1; ti += c3 - c2;
        }
        wl_wait_vm<0>();   // nothing may land after the workgroup has released its LDS
        if ((WL_ROWS_ABLATE & 8) && lane == 0) {
            T* o = a.ll + (size_t)plane * a.ll_ps;
            o[8] = (T)(float)(tw >> 6); o[9] = (T)(float)(tb >> 6); o[10] = (T)(float)(ti >> 6);
        }
    }

    // ---- compute waves of level j -----------------------------------------------------------------------------
    struct Lane {               // per lane
        wl_v2 win[LAT ? (KL > 1 ? KL - 1 : 1) : LT];   // row-filtered (lo,hi) of the last L extended rows of column k, oldest first (LAT: the K - 1 delayed values)
        unsigned ob;            // byte offset of this lane's next output sample inside a band plane
        int off;                // byte offset of its first row-filter sample inside a ring row
        int ndst, hx0, hx1;     // next level's ring row: its LL sample and the halo cells it is the source of
    };
    struct Role {               // per wave (wave-uniform)
        wl_v2 tw[LAT ? KL : LT], th[SAME ? 1 : LT];   // (lo,hi) tap pairs along W / along H (LAT: the QMF form, (lo[u], lo[L-1-u]) g)
        wl_v2 lt[LAT ? KL : 1];            // LAT: (T_k, -T_k) of the column lattice
        WL_DEV wl_v2 colt(int t) const { return SAME ? tw[LAT ? 0 : t] : th[SAME ? 0 : t]; }   // column-filter tap pair t (direct form)
        char* hp0; char* hp1; char* hp2; char* llp;   // band planes of this (plane, level); LL plane of the last level
        unsigned rowb, llrowb, kb;
        int nring, npitch, rmask;
        unsigned nmagic;                   // NP2: rmask holds the ROWS of the ring this level writes, nmagic its magic
        bool last, halo;
    };

    // the L samples of this lane in two ring rows (LDS byte offsets row0/row1, wave-uniform), as (even, odd) pairs
    static WL_DEV void load_rows(const Lane& L, const char* smem, int row0, int row1, wl_v2 (&s0)[NP], wl_v2 (&s1)[NP]) {
        const char* p0 = smem + (row0 + L.off);
        const char* p1 = smem + (row1 + L.off);
        if (SZ == 4) {
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                const wl_f2 t0 = *reinterpret_cast<const wl_f2*>(p0 + 8 * u);
                const wl_f2 t1 = *reinterpret_cast<const wl_f2*>(p1 + 8 * u);
                s0[u] = wl_v2{t0.x, t0.y}; s1[u] = wl_v2{t1.x, t1.y};
            }
        } else {
            typedef T Pair2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                const Pair2 t0 = *reinterpret_cast<const Pair2*>(p0 + 2 * SZ * u);
                const Pair2 t1 = *reinterpret_cast<const Pair2*>(p1 + 2 * SZ * u);
                s0[u] = wl_v2{(float)t0.x, (float)t0.y}; s1[u] = wl_v2{(float)t1.x, (float)t1.y};
            }
        }
    }
    // row filter of two rows: four independent chains (even / odd taps of either row; dependent v_pk_fma_f32 need a
    // wait state in between)
    static WL_DEV void row_pass(const Role& R, const wl_v2 (&s0)[NP], const wl_v2 (&s1)[NP], wl_v2& a0, wl_v2& a1) {
        if constexpr (ODD != 0) {    // sample of tap t = the OTHER half: even taps pair t/2 .y, odd taps pair (t+1)/2 .x
            wl_v2 b0, b1;
            if constexpr (LAT != 0) {
                a0 = wl_qmf_mul<LT>(R.tw, 0, 1, s0[0]); a1 = wl_qmf_mul<LT>(R.tw, 0, 1, s1[0]);
                b0 = wl_qmf_mul<LT>(R.tw, 1, 0, s0[1]); b1 = wl_qmf_mul<LT>(R.tw, 1, 0, s1[1]);
#pragma unroll
                for (int u = 1; u < LT / 2; ++u) {
                    wl_qmf_fma<LT>(a0, R.tw, 2 * u, 1, s0[u]);
                    wl_qmf_fma<LT>(a1, R.tw, 2 * u, 1, s1[u]);
                    wl_qmf_fma<LT>(b0, R.tw, 2 * u + 1, 0, s0[u + 1]);
                    wl_qmf_fma<LT>(b1, R.tw, 2 * u + 1, 0, s1[u + 1]);
                }
            } else {
                a0 = wl_pk_mul_y(R.tw[0], s0[0]); a1 = wl_pk_mul_y(R.tw[0], s1[0]);
                b0 = wl_pk_mul_x(R.tw[1], s0[1]); b1 = wl_pk_mul_x(R.tw[1], s1[1]);
#pragma unroll
                for (int u = 1; u < LT / 2; ++u) {
                    wl_pk_fma_y(a0, R.tw[2 * u], s0[u]);
                    wl_pk_fma_y(a1, R.tw[2 * u], s1[u]);
                    wl_pk_fma_x(b0, R.tw[2 * u + 1], s0[u + 1]);
                    wl_pk_fma_x(b1, R.tw[2 * u + 1], s1[u + 1]);
                }
            }
            a0 += b0;
            a1 += b1;
            return;
        }
        if constexpr (LAT != 0) {
            a0 = wl_qmf_mul<LT>(R.tw, 0, 0, s0[0]); a1 = wl_qmf_mul<LT>(R.tw, 0, 0, s1[0]);
            wl_v2 b0 = wl_qmf_mul<LT>(R.tw, 1, 1, s0[0]), b1 = wl_qmf_mul<LT>(R.tw, 1, 1, s1[0]);
#pragma unroll
            for (int u = 1; u < LT / 2; ++u) {
                wl_qmf_fma<LT>(a0, R.tw, 2 * u, 0, s0[u]);
                wl_qmf_fma<LT>(a1, R.tw, 2 * u, 0, s1[u]);
                wl_qmf_fma<LT>(b0, R.tw, 2 * u + 1, 1, s0[u]);
                wl_qmf_fma<LT>(b1, R.tw, 2 * u + 1, 1, s1[u]);
            }
            a0 += b0;
            a1 += b1;
            return;
        }
        a0 = wl_pk_mul_x(R.tw[0], s0[0]); a1 = wl_pk_mul_x(R.tw[0], s1[0]);
        wl_v2 b0 = wl_pk_mul_y(R.tw[1], s0[0]), b1 = wl_pk_mul_y(R.tw[1], s1[0]);
#pragma unroll
        for (int u = 1; u < LT / 2; ++u) {
            wl_pk_fma_x(a0, R.tw[2 * u], s0[u]);
            wl_pk_fma_x(a1, R.tw[2 * u], s1[u]);
            wl_pk_fma_y(b0, R.tw[2 * u + 1], s0[u]);
            wl_pk_fma_y(b1, R.tw[2 * u + 1], s1[u]);
        }
        a0 += b0;
        a1 += b1;
    }
    // column filter of the window w[0..L) (oldest first) into one sample of each sub-band, and their stores:
    // (LL, W-lo/H-hi) and (W-hi/H-lo, HH), two chains each (even / odd taps)
    template <bool LAST, bool HALO>
    static WL_DEV void col_pass(Lane& L, const Role& R, char* smem, const wl_v2* w, int orow, bool keep) {
        wl_v2 cl = wl_pk_mul_x(R.colt(0), w[0]), ch = wl_pk_mul_y(R.colt(0), w[0]);
        wl_v2 cl2 = wl_pk_mul_x(R.colt(1), w[1]), ch2 = wl_pk_mul_y(R.colt(1), w[1]);
#pragma unroll
        for (int t = 2; t < LT; t += 2) {
            wl_pk_fma_x(cl, R.colt(t), w[t]);
            wl_pk_fma_y(ch, R.colt(t), w[t]);
            wl_pk_fma_x(cl2, R.colt(t + 1), w[t + 1]);
            wl_pk_fma_y(ch2, R.colt(t + 1), w[t + 1]);
        }
        cl += cl2;
        ch += ch2;
        emit_row<LAST, HALO>(L, R, smem, cl, ch, orow, keep);
    }
    // one feed of the column lattice (wl_lattice.h): (a, b) = the row-filtered pair of new rows -> cl = (ll, lh), ch = (hl, hh); the
    // K - 1 delayed values move along by one stage (the direct form moves its L - 2 window rows)
    static WL_DEV void lat_feed(Lane& L, const Role& R, wl_v2 a, wl_v2 b, wl_v2& cl, wl_v2& ch) {
        wl_v2 u = wl_fma_s<0, 0>(b, R.lt[0], a);                  // a + T0 b
        wl_v2 v = wl_fma_s<1, 0>(a, R.lt[0], b);                  // b - T0 a
        wl_v2 hi = -v;
#pragma unroll
        for (int k = 1; k < KL; ++k) {
            const wl_v2 d = L.win[k - 1];
            L.win[k - 1] = v;
            const wl_v2 n = wl_fma_s<0, 0>(d, R.lt[k], u);         // u + Tk v'
            if (k < KL - 1) v = wl_fma_s<1, 0>(u, R.lt[k], d);     // v' - Tk u
            else hi = wl_fma_s<0, 1>(u, R.lt[k], d);               // -(v' - Tk u)
            u = n;
        }
        cl = wl_v2{u.x, hi.x}; ch = wl_v2{u.y, hi.y};             // the lattice filters the (row-lo, row-hi) pair: a renaming of registers
    }
    template <bool LAST, bool HALO>
    static WL_DEV void emit_row(Lane& L, const Role& R, char* smem, wl_v2 cl, wl_v2 ch, int orow, bool keep) {
        const unsigned ob = L.ob;
        L.ob = ob + R.rowb;
        // keep: the row belongs to this segment (wave-uniform); halo rows of a cut plane are computed, not stored
        const bool st = keep && (!(WL_ROWS_ABLATE & 1) || (cl.x + cl.y + ch.x + ch.y == 12345.f));
        if (st) {
            wl_store_stream(reinterpret_cast<T*>(R.hp0 + ob), (T)cl.y);    // W-lo / H-hi
            wl_store_stream(reinterpret_cast<T*>(R.hp1 + ob), (T)ch.x);    // W-hi / H-lo
            wl_store_stream(reinterpret_cast<T*>(R.hp2 + ob), (T)ch.y);    // W-hi / H-hi
        }
        if (LAST) {
            if (st) *reinterpret_cast<T*>(R.llp + ((unsigned)orow * R.llrowb + R.kb)) = (T)cl.x;
        } else {
            char* nrow = smem + (R.nring + (NP2 ? wl_uniform(wl_ring_slot(orow, R.rmask, R.nmagic)) : (orow & R.rmask)) * R.npitch);
            *reinterpret_cast<T*>(nrow + L.ndst) = (T)cl.x;
            if (HALO) {   // only waves that own a boundary column of the next level
                if (L.hx0 >= 0) *reinterpret_cast<T*>(nrow + L.hx0) = (T)cl.x;
                if (L.hx1 >= 0) *reinterpret_cast<T*>(nrow + L.hx1) = (T)cl.x;
            }
        }
    }

    // one feed: row-filter the two new source rows into the window and - once the window is full - emit one output row
    template <bool LAST, bool HALO>
    static WL_DEV void feed1(Lane& L, const Role& R, char* smem, int row0, int row1, bool emit, int orow, bool keep) {
        wl_v2 s0[NP], s1[NP], a0, a1;
        load_rows(L, smem, row0, row1, s0, s1);
        row_pass(R, s0, s1, a0, a1);
        if constexpr (LAT != 0) {   // (every feed: the lattice's state; the first K - 1 outputs of a segment are the warm-up)
            wl_v2 cl, ch;
            lat_feed(L, R, a0, a1, cl, ch);
            if (emit) emit_row<LAST, HALO>(L, R, smem, cl, ch, orow, keep);
        } else {
#pragma unroll
            for (int t = 0; t < LT - 2; ++t) L.win[t] = L.win[t + 2];
            L.win[LT - 2] = a0;
            L.win[LT - 1] = a1;
            if (emit) col_pass<LAST, HALO>(L, R, smem, L.win, orow, keep);
        }
    }
    // two feeds of the steady state (both emit): all four rows are requested from LDS before the first FMA, the
    // window moves once (by four rows) instead of twice
    template <bool LAST, bool HALO>
    static WL_DEV void feed2(Lane& L, const Role& R, char* smem, int row0, int row1, int row2, int row3, int orow,
                             bool keep0, bool keep1) {
        wl_v2 s0[NP], s1[NP], s2[NP], s3[NP];
        load_rows(L, smem, row0, row1, s0, s1);
        load_rows(L, smem, row2, row3, s2, s3);
        if constexpr (LAT != 0) {
            wl_v2 a0, a1, cl, ch;
            row_pass(R, s0, s1, a0, a1);
            lat_feed(L, R, a0, a1, cl, ch);
            emit_row<LAST, HALO>(L, R, smem, cl, ch, orow, keep0);
            row_pass(R, s2, s3, a0, a1);
            lat_feed(L, R, a0, a1, cl, ch);
            emit_row<LAST, HALO>(L, R, smem, cl, ch, orow + 1, keep1);
            return;
        }
        wl_v2 w[LT + 4];
#pragma unroll
        for (int t = 0; t < LT; ++t) w[t] = L.win[t];
        row_pass(R, s0, s1, w[LT], w[LT + 1]);
        col_pass<LAST, HALO>(L, R, smem, w + 2, orow, keep0);
        row_pass(R, s2, s3, w[LT + 2], w[LT + 3]);
        col_pass<LAST, HALO>(L, R, smem, w + 4, orow + 1, keep1);
#pragma unroll
        for (int t = 0; t < LT; ++t) L.win[t] = w[t + 4];
    }

    template <int j>
    static WL_DEV void compute(const Args& a, const WlRowsSeg& sg, const WlCtx& ctx, int64_t plane, int col0, int lane) {
        const WlRowsLevel& g = a.g[j];
        const int k = col0 + lane;
        const bool active = k < g.Kw;
        Role R;
        if constexpr (LAT != 0) {
            const float gsc = a.lat[1];                       // the lattice's gain rides on the row taps
#pragma unroll
            for (int u = 0; u < KL; ++u) {
                R.tw[u] = wl_uniform_v2(wl_v2{a.h_w_lo[u] * gsc, a.h_w_lo[LT - 1 - u] * gsc});
                R.lt[u] = wl_uniform_v2(wl_v2{a.lat[2 + u], -a.lat[2 + u]});
            }
        } else {
#pragma unroll
            for (int t = 0; t < LT; ++t) {
                R.tw[t] = wl_uniform_v2(wl_v2{a.h_w_lo[t], a.h_w_hi[t]});
                if (!SAME) R.th[SAME ? 0 : t] = wl_uniform_v2(wl_v2{a.h_h_lo[t], a.h_h_hi[t]});
            }
        }
        R.last = j == a.nlev - 1;
        const unsigned bplane = (unsigned)g.Kh * (unsigned)g.Kw;
        R.hp0 = reinterpret_cast<char*>(a.yh[j] + (size_t)plane * 3 * bplane);
        R.hp1 = R.hp0 + (size_t)bplane * SZ;
        R.hp2 = R.hp1 + (size_t)bplane * SZ;
        R.llp = reinterpret_cast<char*>(a.ll + (size_t)plane * a.ll_ps);
        R.rowb = (unsigned)g.Kw * SZ; R.llrowb = (unsigned)a.ll_rs * SZ; R.kb = (unsigned)k * SZ;
        const WlRowsLevel& gn = a.g[j + 1 < WL_ROWS_MAXLEV ? j + 1 : j];
        R.nring = gn.ring_off; R.npitch = gn.ring_pitch;
        R.rmask = NP2 ? gn.ring_rows : gn.ring_rows - 1;  // (the ring this level WRITES: the next level's source ring)
        R.nmagic = gn.ring_magic;
        Lane L;
#pragma unroll
        for (int t = 0; t < (LAT ? (KL > 1 ? KL - 1 : 1) : LT); ++t) L.win[t] = wl_v2{0.f, 0.f};
        L.ob = (unsigned)sg.f0[j] * R.rowb + (unsigned)k * SZ;   // the first row this segment produces is row f0
        L.off = g.pad + (2 * (active ? k : 0) + a.base - ODD) * SZ;   // halo cells included: always >= 0 (ODD: one cell earlier, an aligned pair)
        // the halo cells of the NEXT level's ring rows whose source column is k - at most one on either side
        L.ndst = gn.pad + k * SZ; L.hx0 = L.hx1 = -1;
        if (!R.last && active && a.ext != WL_EXT_ZERO) {
            for (int c = 0; c < gn.hl + gn.hr; ++c) {
                const int e = c < gn.hl ? c - gn.hl : gn.Ws + (c - gn.hl);
                if (wl_ext(e, gn.Ws, a.ext) == k) {
                    if (e < 0) L.hx0 = gn.pad + e * SZ; else L.hx1 = gn.pad + e * SZ;
                }
            }
        }
        // does any lane of this wave own a halo cell?  (columns near either edge of the next level's rows)
        R.halo = !R.last && a.ext != WL_EXT_ZERO && (col0 <= gn.hl + 1 || col0 + 63 >= gn.Ws - gn.hr - 2);
        if (R.last) main_loop<j, true, false>(a, sg, ctx, plane, L, R, active, k);
        else if (R.halo) main_loop<j, false, true>(a, sg, ctx, plane, L, R, active, k);
        else main_loop<j, false, false>(a, sg, ctx, plane, L, R, active, k);
    }

    template <int j, bool LAST, bool HALO>
    static WL_DEV void main_loop(const Args& a, const WlRowsSeg& sg, const WlCtx& ctx, int64_t plane, Lane& L, const Role& R,
                                 bool active, int k) {
        const WlRowsLevel& g = a.g[j];
        char* const smem = ctx.smem;
        const int rmask = NP2 ? g.ring_rows : g.ring_rows - 1, zrow = a.zero_off, ring = g.ring_off, pitch = g.ring_pitch, Hs = g.Hs;   // (the ring this level READS)
        const bool zmode = a.ext == WL_EXT_ZERO, per = a.ext == WL_EXT_PER;
        // LDS byte offsets (wave-uniform) of the two source rows of feed f = (2f+base, 2f+base+1); i = its index in
        // the half-batch hb
        auto rows_of = [&](int f, int hb, int i, int& r0, int& r1) {
            const int e = a.base + 2 * f;
            if (j == 0) {
                r0 = ring + ((hb % NSLOT) * 4 + 2 * i) * pitch;
                r1 = r0 + pitch;
                if (zmode) {
                    if ((unsigned)e >= (unsigned)Hs) r0 = zrow;
                    if ((unsigned)(e + 1) >= (unsigned)Hs) r1 = zrow;
                }
            } else {
                // periodization: the rows above / below the plane are COMPUTED (the level above produces rows -hl .. Hs + hr of its
                // periodic output, the wrapped ones twice): the ring is addressed by the unfolded row number, negative ones included
                int s0 = e, s1 = e + 1;
                if (!per && (e < 0 || e + 1 >= Hs)) { s0 = wl_ext1(e, Hs, a.ext); s1 = wl_ext1(e + 1, Hs, a.ext); }   // top / bottom rows only
                r0 = zmode && s0 < 0 ? zrow : ring + (NP2 ? wl_ring_slot(s0, rmask, g.ring_magic) : (s0 & rmask)) * pitch;
                r1 = zmode && s1 < 0 ? zrow : ring + (NP2 ? wl_ring_slot(s1, rmask, g.ring_magic) : (s1 & rmask)) * pitch;
            }
            r0 = wl_uniform(r0); r1 = wl_uniform(r1);
        };

        const int f0 = sg.f0[j], own_lo = sg.own_lo[j], own_hi = sg.own_hi[j];
        int fed = f0;           // next feed of this level (wave-uniform)
        unsigned long long tb = 0, tf = 0, ts = 0, c3 = WL_TICK();
        for (int hb = 0; hb < sg.nhb; ++hb) {
            const unsigned long long c0 = WL_TICK();
            ts += c0 - c3;
            const int n = wl_uniform((int)(sg.sched[hb >> 2] >> (8 * (hb & 3) + 2 * j)) & 3);   // read before the barrier
            ctx.sync();
            const unsigned long long c1 = WL_TICK();
            tb += c1 - c0;
            const int orow = fed - WARM;   // the output row the next feed completes (once the window is full)
#ifndef WL_ROWS_FEED2_MAXL
#define WL_ROWS_FEED2_MAXL 10    // the longest filter that takes the two-feed form: at 12 taps its four sample rows spilled 15 VGPRs
                                 // (28 B of scratch) at the 80-register budget; one feed at a time: 62 registers, db6 0.221 -> 0.2145 ms
#endif
            if (LT <= WL_ROWS_FEED2_MAXL && n == 2 && fed - f0 >= WARM) {       // the steady state of level 1
                int r0, r1, r2, r3;
                rows_of(fed, hb, 0, r0, r1);
                rows_of(fed + 1, hb, 1, r2, r3);
                if (active)
                    feed2<LAST, HALO>(L, R, smem, r0, r1, r2, r3, orow, orow >= own_lo && orow < own_hi,
                                      orow + 1 >= own_lo && orow + 1 < own_hi);
                fed += 2;
            } else {
                for (int i = 0; i < n; ++i) {
                    int r0, r1;
                    rows_of(fed, hb, i, r0, r1);
                    const int o = fed - WARM;
                    if (active) feed1<LAST, HALO>(L, R, smem, r0, r1, fed - f0 >= WARM, o, o >= own_lo && o < own_hi);
                    ++fed;
                }
            }
            c3 = WL_TICK();
            tf += c3 - c1;
        }
        if ((WL_ROWS_ABLATE & 8) && j == 0 && k == 64) {   // second level-1 wave
            T* o = a.ll + (size_t)plane * a.ll_ps;
            o[0] = (T)(float)(tb >> 6); o[1] = (T)(float)(tf >> 6); o[2] = (T)(float)(ts >> 6);
        }
    }

    static WL_DEV void run(const Args& a, const WlCtx& ctx) {
        const int tid = ctx.tid;
        const int wave = wl_uniform(tid >> 6), lane = tid & 63;
        if (a.guard) {   // "both axes filter with the same taps", checked against the taps as they are now (whatever tap count an A/B build arms)
            const bool holds = a.lat ? *reinterpret_cast<const unsigned*>(a.lat) == WL_LAT_OK   // (WlTapPrep's verdict: same banks, mirror pair, lattice)
                                     : wl_taps_same(a.h_w_lo, a.h_h_lo, LT) && wl_taps_same(a.h_w_hi, a.h_h_hi, LT);
            if (!wl_guard_pass(a.guard, holds)) return;
        }
        // workgroup -> (plane, segment)
        const int64_t bid = ctx.bid;
        const int sub = wl_uniform(a.role_sub[wave]);
        const int64_t plane = (bid < a.nwhole ? bid : a.nwhole + (bid - a.nwhole) / 2) * a.pp + sub;
        const WlRowsSeg& sg = a.seg[bid < a.nwhole ? 0 : 1 + (int)((bid - a.nwhole) & 1)];
        // all of LDS starts as zeros: the zero row, and halo cells that stay zero in zero-padding mode
        for (int i = tid * 16; i < a.lds_bytes; i += kThreads * 16) {
            wl_f4 z; z.x = z.y = z.z = z.w = 0.f;
            *reinterpret_cast<wl_f4*>(ctx.smem + i) = z;
        }
        ctx.sync();
        WlCtx cs = ctx;                                  // this wave's plane: its own part of the LDS
        cs.smem += sub * a.lds_plane; cs.lds_base += (unsigned)(sub * a.lds_plane);
        const int lev = plane < a.NC ? wl_uniform(a.role_level[wave]) : -2, col0 = wl_uniform(a.role_col0[wave]);   // (the last workgroup may hold fewer planes)
        if (lev == -1) loader(a, sg, cs, plane, lane, col0);
        else if (lev == 0) compute<0>(a, sg, cs, plane, col0, lane);
        else if (lev == 1) compute<1>(a, sg, cs, plane, col0, lane);
        else if (lev == 2) compute<2>(a, sg, cs, plane, col0, lane);
        else
            for (int hb = 0; hb < sg.nhb; ++hb) ctx.sync();   // spare wave: keeps the barrier count
    }
};
