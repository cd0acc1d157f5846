// Levels 1 AND 2 of the DTCWT forward in one streaming launch: fwd_j1 followed by fwd_j2plus (reference
// dtcwt/transform_funcs.py:98-121 and :226-249 = coldfilt / rowdfilt of dtcwt/lowlevel.py:99-151), the way
// DTCWTForward.forward chains them (dtcwt/transform2d.py:121-141) - without the level-1 lowpass ever leaving the chip.
// Level 1 is undecimated, so LL1 is as large as the image: written and read back it is 8 of the 28 bytes per pixel the
// two levels move.  Here it lives in a two-slot LDS ring of four rows.
//
// A workgroup owns one (plane, strip of quad columns, segment of 4-row groups) and streams down it, four rows per
// half-batch, with three kinds of waves:
//   * 4 stager waves: one input row each per half-batch, loaded as 4-cell groups straight into registers one half-batch
//     ahead (wl_dwt_strip.h), converted to float32 and staged with the mirrored cells of the plane's edges materialised;
//   * 4 level-1 waves: the lanes of WlDtFwd1Strip (a lane owns the two columns of a quad column, row filter pair into
//     circular register windows, column filter pair, q2c, the level-1 band-pass stores).  The lowpass goes into the LL1
//     ring instead of memory - including, at the plane's left / right edge, the mirrored copies level 2 will read;
//   * 2 level-2 waves, one half-batch behind: a lane owns FOUR LL1 columns = two half-resolution columns.  Per LL1 row
//     it reads the 2 LQ samples around them from the ring and runs the dual-tree row filters (even samples meet
//     (h0b, h1b), odd ones (h0a, h1a): one packed FMA per sample), keeps the last 2 LQ row-filtered rows in a lane-private
//     LDS window (no synchronisation: nobody else reads them), and every four rows runs the column filters over the
//     window: a 2 x 2 block of LL2 and, through q2c, one coefficient of each of the six level-2 orientations.
// Rows and columns outside a segment / strip that level 2 needs (LQ - 2 = 8 either side) are computed, not exchanged:
// the level-1 lanes run over them without storing their band-pass outputs.  Above / below the PLANE the same happens on
// the symmetrically extended input, with the column lowpass taps met in REVERSE order: LL1 row -1-r is then exactly LL1 row r
// - the symmetric extension of LL1 that coldfilt applies - for ANY taps (sum_t h[t] lo[-1-r + M - t] = sum_t h[t] lo[r + t - M]
// by the half-sample symmetry of the extended, row-filtered rows).  For the symmetric tables of the reference the reversed order
// changes nothing; a level-1 lowpass somebody loaded or edited into an asymmetric one is filtered correctly too, so no property
// of the taps is assumed (rounds 3-4 took the symmetry from a host-side check of the buffer, which writes through `.data` escape).
#pragma once
#include "wl_dtcwt_strip.h"

// acc += tap * s.x / s.y (both halves), the tap pair in vector registers (per-lane taps)
#if defined(__HIPCC__)
WL_DEV void wl_pk_fma_x_v(wl_v2& acc, wl_v2 tap, wl_v2 s) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(tap), "v"(s));
}
WL_DEV void wl_pk_fma_y_v(wl_v2& acc, wl_v2 tap, wl_v2 s) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(tap), "v"(s));
}
#else
inline void wl_pk_fma_x_v(wl_v2& acc, wl_v2 tap, wl_v2 s) { wl_pk_fma_x(acc, tap, s); }
inline void wl_pk_fma_y_v(wl_v2& acc, wl_v2 tap, wl_v2 s) { wl_pk_fma_y(acc, tap, s); }
#endif

#if defined(WL_DT12_TIME) && defined(__HIPCC__)
#define WL_DT12_TICK() __builtin_readcyclecounter()
#else
#define WL_DT12_TICK() 0ull
#endif
// (timing builds: workgroup 0 leaves (total, barrier) kilocycles of one wave per role in ll2[16..21])
#define WL_DT12_SYNC() { const unsigned long long t_ = WL_DT12_TICK(); ctx.sync(); tbar += WL_DT12_TICK() - t_; }
#ifndef WL_DT12_ROWLOADS
#define WL_DT12_ROWLOADS 0       // 1: the level-1 lanes read a row's samples when they filter it (fewer registers) instead of the four rows up front
#endif
#ifndef WL_DT12_ABLATE
#define WL_DT12_ABLATE 0        // A/B builds: 1 = level-2 waves idle, 2 = no level-1 band-pass stores, 4 = no level-2 stores, 8 = no level-1 arithmetic
#endif

template <typename T>
struct WlDtFusedArgs {
    WlDtFwd1Args<T> f;             // level 1: x, highs (ll = z = nullptr), taps, H = He, W = We (multiples of 4), ext = symmetric
    T* ll2;                        // (NC, H/2, W/2)
    T* highs2;                     // (NC, 6, H/4, W/4, 2)
    const float* h0a; const float* h0b; const float* h1a; const float* h1b;   // LQ taps each
    int64_t nblocks;
    int nstrips, strip_quads;      // own quad columns per strip (even); the last strip may be narrower
    int nseg, seg_groups;          // own 4-row groups per segment
    int st_off, st_pitch;          // staged input rows: 2 slots x 4 rows x st_pitch bytes (float32)
    int l1_off, l1_pitch;          // LL1 ring: 2 slots x 4 rows x l1_pitch bytes
    int w2_off;                    // level-2 windows: LQ rows x 256 lanes x 16 bytes
    int lds_bytes;
    const float* h2; int L2;       // MODE 6: the band-pass filter of the diagonal sub-band ('near_sym_b_bp'), L2 <= 2 M + 1 taps
};

// MODE 2: levels 1 + 2 (above).  The same stagers and level-1 lanes without the level-2 waves are the lean level-1 kernels:
// MODE 0: fwd_j1 alone (lowpass and band-pass coefficients to memory), MODE 1: ScatLayerj1_f.forward (scatternet/lowlevel.py:
// 76-111: the 2x2-averaged lowpass and the six smoothed magnitudes sqrt(re^2 + im^2 + b^2) - b, optionally the
// full-resolution lowpass for ScatLayerj2), MODE 3: MODE 1 + (re, im) / r saved for the backward pass (a compile-time
// variant: the inference kernel does not carry the pointers of the saved tensors through its scalar registers).
// MODE 4: fwd_j2plus alone - the stagers put the rows of the level's input (and their mirrored cells: exact, no symmetry
// assumed) straight into the ring the level-2 lanes read; no level-1 waves.  MODE 5: MODE 4 with ScatLayerj2's epilogue (the
// 2x2-averaged lowpass and the smoothed magnitudes instead of LL2 and the band-pass coefficients).
// MODE 6 (round 6): MODE 1 for the rotationally symmetric filters ('near_sym_b_bp': ScatLayerj1_rot_f.forward, scatternet/lowlevel.py:
// 140-182; fwd_j1_rot, dtcwt/transform_funcs.py:124-149) - a third row filter ba = R_h2 x (both columns of the quad in ONE packed FMA per
// tap: (ba0, ba1) += (h2[t], h2[t - 1]) x[t], the tap pairs in vector registers), a third window, hh = C_h2 ba instead of C_h1 hi.
// PP = 2 (lean level-1 kernels, planes of up to 256 columns): a workgroup owns TWO consecutive planes - level-1 waves 0, 1 the
// first, 2, 3 the second, every stager wave its row of both (the staged rows lie side by side) - so that all four level-1
// waves of the wide-plane kernel work (with one 256-column plane per workgroup of two level-1 waves + two stagers ScatLayer
// ran at 0.49-0.51 of its roofline against 0.55-0.57 on 512-column planes).  PP = 4: planes of up to 128 columns, one per
// level-1 wave (the second-order layer of ScatLayerj2 on 256 x 256 images: 18 planes of 128 x 128 per image).
template <typename T, int L0, int L1, int LQ, int MODE = 2, int CW_ = 4, int PP = 1>
struct WlDtFwd12Strip {
    typedef WlDtFusedArgs<T> Args;
    static const bool kL2 = MODE == 4 || MODE == 5;    // fwd_j2plus alone (5: with ScatLayerj2's epilogue)
    static_assert(PP == 1 || ((PP == 2 || PP == 4) && CW_ == 4 && MODE != 2 && MODE != 4 && MODE != 5), "several planes per workgroup: lean level-1 kernels only");
    static const bool kRot = MODE == 6;
#ifndef WL_DT12_SW
#define WL_DT12_SW 4
#endif
    // level-1, level-2 and stager waves (CW_ = 2: planes of up to 256 columns, two stagers of two rows each: one wave of the
    // workgroup per SIMD)
    static const int CW = kL2 ? 0 : CW_, QW = (MODE == 2 || kL2) ? CW_ : 0, SW = CW_ == 2 ? 2 : WL_DT12_SW;
    static const int LROWS = 4 / SW;                   // rows of a half-batch per stager wave
    static const int kWaves = CW + QW + SW;
    static const int kThreads = 64 * kWaves;
#ifndef WL_DT12_MINW1
#define WL_DT12_MINW1 6
#endif
    // two workgroups of 10 (12) waves per CU, or three of 8: at most 96 (80) registers
    static const bool kScat = MODE == 1 || MODE == 3 || MODE == 6;
    // (MODE 3, the training forward, keeps 19 output streams' addresses and values live: at the 80 registers of six waves per
    // SIMD it spilled 88 bytes per lane and ran 0.354 ms at config 4's shape; at 128 registers (four waves, two workgroups per
    // CU) nothing spills: 0.235 ms.  The inference kernel fits 78 registers and is faster at six.)
    static const int SZ = (int)sizeof(T);
    static const int M0 = L0 / 2, M1 = L1 / 2, M = kL2 ? 0 : (M0 > M1 ? M0 : M1);   // (MODE 4: no level-1 filters)
    // (the 13 / 19-tap pair of near_sym_b: two windows of 20 rows are 80 registers - room for 128, two workgroups of 8 waves per
    // CU - and a lane reads a row's samples when it filters it, not the four rows of a half-batch up front)
    static const bool kLong = M > 4;
    static const int kMinWaves = kLong ? 2 : (MODE == 3 ? 4 : (kScat ? WL_DT12_MINW1 : (SW == 2 && MODE == 2 ? 5 : 6)));
    static const bool kRowLoads = WL_DT12_ROWLOADS || kLong;
    static const int LW = (2 * M + 1 + 3) / 4 * 4;
    static const int PERIOD = LW / 4;
    static const int NS = 2 + 2 * M;
    static const int NC2 = NS / 2;
    static const int HQ = (MODE == 2 || kL2) ? LQ - 2 : 0;   // LL1 columns / rows level 2 reads beyond its own, either side
    static const int HG = HQ / 4;                      // the same in 4-row groups
    static const int NW2 = 2 * LQ;                     // rows of the level-2 window
    static const int NG2 = NW2 / 4;
    static const int WARM1 = kL2 ? 0 : (2 * M + 3) / 4;   // half-batches before the first LL1 row of a segment is complete
    static const int MAXG = 3;                         // 4-cell groups per stager lane and row
#ifndef WL_DT12_PF
#define WL_DT12_PF 2
#endif
    static const int PF = WL_DT12_PF;                  // register sets of a stager = half-batches a row is requested ahead
    static_assert(HQ % 4 == 0, "level-2 filters of 10, 14 or 18 taps");

    struct Strip {
        int q0, q1;            // own quad columns
        int qa, qb;            // quad columns the level-1 lanes run over (own + what level 2 needs of the neighbours)
        int e_lo;              // first extended pixel column a level-1 lane reads = staged cell 4 (whole groups start up to 3 cells earlier)
        int gc0, ng;           // first 4-column group loaded, number of groups
        int nl, nr;            // mirrored cells left / right of the row
        int g_lo, g_hi;        // own 4-row groups
        int o_base;            // LL1 row of (half-batch 0, row 0): a multiple of 4
        int nhb1, nhb;         // half-batches of level 1; barriers of the workgroup (level 2 runs one behind; a multiple of PF)
    };
    static WL_HD Strip geometry(const Args& a, int strip, int seg) {
        Strip s;
        const int Q = a.f.W / 2;
        s.q0 = strip * a.strip_quads;
        s.q1 = s.q0 + a.strip_quads < Q ? s.q0 + a.strip_quads : Q;
        s.qa = s.q0 > 0 ? s.q0 - HQ / 2 : 0;
        s.qb = s.q1 < Q ? s.q1 + HQ / 2 : Q;
        // columns the stagers provide: MODE 4 stages the ring level 2 reads (own columns + HQ either side), the others the
        // samples of the level-1 lanes
        s.e_lo = kL2 ? 2 * s.q0 - HQ : 2 * s.qa - M;
        const int e_hi = kL2 ? 2 * s.q1 - 1 + HQ : 2 * s.qb - 1 + M;
        s.nl = s.e_lo < 0 ? -s.e_lo : 0;
        s.nr = e_hi > a.f.W - 1 ? e_hi - (a.f.W - 1) : 0;
        s.gc0 = (s.e_lo < 0 ? 0 : s.e_lo) / 4;
        s.ng = (e_hi > a.f.W - 1 ? a.f.W - 1 : e_hi) / 4 - s.gc0 + 1;
        const int G = a.f.H / 4;
        s.g_lo = seg * a.seg_groups;
        s.g_hi = s.g_lo + a.seg_groups < G ? s.g_lo + a.seg_groups : G;
        s.o_base = 4 * (s.g_lo - HG - WARM1);
        s.nhb1 = WARM1 + (s.g_hi - s.g_lo) + 2 * HG;
        s.nhb = (s.nhb1 + (MODE == 2 ? 1 : 0) + PF - 1) / PF * PF;
        return s;
    }

    static WL_DEV void report(const Args& a, const WlCtx& ctx, int lane, bool first, int slot, unsigned long long t0, unsigned long long tbar) {
#if defined(WL_DT12_TIME) && defined(__HIPCC__)
        if (MODE == 2 && ctx.bid == 0 && first && lane == 0) {
            a.ll2[16 + slot] = (T)(float)((WL_DT12_TICK() - t0) >> 10);
            a.ll2[17 + slot] = (T)(float)(tbar >> 10);
        }
#endif
    }

    // ---- stager wave: row `sidx` of every half-batch --------------------------------------------------------------------
    typedef T Quad4 __attribute__((ext_vector_type(4), aligned(sizeof(T)), may_alias));
    struct RowRegs { Quad4 g[LROWS][MAXG]; T h[LROWS]; };
    static const int SLACK = kL2 ? 0 : 4;        // staged cell of column e_lo (MODE 4: e_lo is a multiple of 4, groups start on it)
    template <int NGL>
    static WL_DEV void stager(const Args& a, const Strip& s, const WlCtx& ctx, int64_t plane, int lane, int sidx) {
        const WlDtFwd1Args<T>& f = a.f;
        const char* xp = reinterpret_cast<const char*>(f.x + (size_t)plane * f.H * f.W);
        const int row_stride = f.W * SZ;
        // (PP = 2: groups [ng, 2 ng) are the second plane's - H W elements further on in memory, half a staged row further
        // on in LDS; an odd number of planes leaves the last workgroup's second half empty)
        const int np = f.NC - plane < PP ? (int)(f.NC - plane) : PP;      // planes of this workgroup that exist
        const int sub_src = f.H * f.W * SZ, sub_dst = a.st_pitch / PP;
        int goff[MAXG], gdst[MAXG];
#pragma unroll
        for (int i = 0; i < MAXG; ++i) {
            const int g0 = lane + 64 * i;
            int sub = 0, g = g0;
#pragma unroll
            for (int u = 1; u < PP; ++u) if (g >= s.ng) { g -= s.ng; ++sub; }
            const int col = 4 * (s.gc0 + g);
            const bool on = g < s.ng && sub < np;
            goff[i] = on ? col * SZ + sub * sub_src : 0;
            gdst[i] = on ? (col - s.e_lo + SLACK) * 4 + sub * sub_dst : -1;
        }
        int hdst = -1, hoff = 0;                               // one mirrored cell per lane (PP planes: 64 / PP lanes each)
        {
            const int sub = PP > 1 ? lane / (64 / PP) : 0;
            const int l = lane - (64 / PP) * sub;
            if (l < s.nl + s.nr && sub < np) {
                const int e = l < s.nl ? s.e_lo + l : f.W + (l - s.nl);
                hdst = (e - s.e_lo + SLACK) * 4 + sub * sub_dst;
                hoff = wl_ext(e, f.W, WL_EXT_SYM) * SZ + sub * sub_src;
            }
        }
        auto load = [&](int h, RowRegs& rr) {
#pragma unroll
            for (int r4 = 0; r4 < LROWS; ++r4) {
                const int r = wl_ext1(s.o_base + M + 4 * h + LROWS * sidx + r4, f.H, WL_EXT_SYM);  // input row e = o + M (one fold: H >= 32)
                const char* grow = xp + (size_t)r * row_stride;
#pragma unroll
                for (int i = 0; i < NGL; ++i) rr.g[r4][i] = *reinterpret_cast<const Quad4*>(grow + goff[i]);
                rr.h[r4] = *reinterpret_cast<const T*>(grow + hoff);
            }
        };
        auto stage = [&](int hb, const RowRegs& rr) {
#pragma unroll
            for (int r4 = 0; r4 < LROWS; ++r4) {
                char* srow = kL2 ? ctx.smem + a.l1_off + ((hb & 1) * 4 + LROWS * sidx + r4) * a.l1_pitch
                                       : ctx.smem + a.st_off + ((hb & 1) * 4 + LROWS * sidx + r4) * a.st_pitch;
#pragma unroll
                for (int i = 0; i < NGL; ++i) {
                    if (gdst[i] < 0) continue;
                    const float v0 = (float)rr.g[r4][i].x, v1 = (float)rr.g[r4][i].y, v2 = (float)rr.g[r4][i].z, v3 = (float)rr.g[r4][i].w;
                    float* dst = reinterpret_cast<float*>(srow + gdst[i]);
                    if ((M & 3) == 0) {                        // (the cell of a group's first column = 4 g + 4 + M - 2 qa, qa even)
                        wl_vf4 w; w.x = v0; w.y = v1; w.z = v2; w.w = v3;
                        *reinterpret_cast<wl_vf4*>(dst) = w;
                    } else if ((M & 1) == 0) {
                        wl_f2 w0, w1; w0.x = v0; w0.y = v1; w1.x = v2; w1.y = v3;
                        *reinterpret_cast<wl_f2*>(dst) = w0; *reinterpret_cast<wl_f2*>(dst + 2) = w1;
                    } else {
                        wl_f2 w; w.x = v1; w.y = v2;
                        dst[0] = v0; *reinterpret_cast<wl_f2*>(dst + 1) = w; dst[3] = v3;
                    }
                }
                if (hdst >= 0) *reinterpret_cast<float*>(srow + hdst) = (float)rr.h[r4];
            }
        };
        // PF register sets: a row is requested PF - 1 .. PF half-batches before it is staged (measured: 2 .. 5 sets make no
        // difference - the loads are not what this kernel waits for).  The loads are unconditional (behind the last row: that
        // row again) and the loop has no branch, so that the compiler can count them and wait for exactly the oldest set.
        unsigned long long tbar = 0;
        const unsigned long long tstart = WL_DT12_TICK();
        RowRegs rr[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) load(u < s.nhb1 ? u : s.nhb1 - 1, rr[u]);
        for (int hb = 0; hb < s.nhb; hb += PF) {              // (nhb is a multiple of PF: no branch inside the body)
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int h = hb + u;
                stage(h, rr[u]);                               // (behind level 1's last half-batch: into a slot nobody reads)
                WL_DT12_SYNC();
                load(h + PF < s.nhb1 ? h + PF : s.nhb1 - 1, rr[u]);
            }
        }
        report(a, ctx, lane, sidx == 0, 4, tstart, tbar);
    }

    // ---- level-1 wave ---------------------------------------------------------------------------------------------------
    // column filter with the taps packed two to a scalar pair: acc += w * (c, c), c = the pair's low / high half
    template <int HI> static WL_DEV void fma_cc(wl_v2& acc, wl_v2 w, wl_v2 pair) {
#if defined(__HIPCC__)
        if (HI) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w), "s"(pair));
        else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(w), "s"(pair));
#else
        const float c = HI ? pair.y : pair.x;
        acc.x = __builtin_fmaf(w.x, c, acc.x); acc.y = __builtin_fmaf(w.y, c, acc.y);
#endif
    }
    struct Taps1 {
        wl_v2 tr[2 * M + 1];               // row-filter tap pairs (h0[t], h1[t]), both centred in 2M+1 slots (zeros outside)
        wl_v2 c0[kLong ? 1 : (L0 + 1) / 2];   // column taps, two to a pair: (h0[2u], h0[2u+1])
        wl_v2 c1[kLong ? 1 : (L1 + 1) / 2];   // (kLong: 36 pairs overflow the scalar file - the column filters read the halves of tr)
        wl_v2 t2[kRot ? 2 * M + 2 : 1];       // MODE 6: (h2c[t], h2c[t - 1]) of the centred, zero-padded band-pass filter - vector registers
    };
    // MODE 6: ba = R_h2 x of both columns of the quad: (ba0, ba1) += (h2c[t], h2c[t - 1]) * x[t], t = 0 .. 2 M + 1 (two chains)
    static WL_DEV wl_v2 row_filter_ba(const Taps1& R, const wl_v2 (&s)[NC2]) {
        wl_v2 a0 = wl_v2{0.f, 0.f}, a1 = wl_v2{0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 2 * M + 2; ++t) {
            wl_v2& acc = (t & 2) ? a1 : a0;
            if (t & 1) wl_pk_fma_y_v(acc, R.t2[t], s[t / 2]); else wl_pk_fma_x_v(acc, R.t2[t], s[t / 2]);
        }
        return a0 + a1;
    }
    // ... and hh = C_h2 ba of the row whose window is centred on slot `c`, both columns
    static WL_DEV wl_v2 col_filter_ba(const Taps1& R, const wl_v2 (&w)[LW], int c) {
        wl_v2 a0 = wl_v2{0.f, 0.f}, a1 = wl_v2{0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 2 * M + 1; ++t) {
            wl_v2& acc = (t & 1) ? a1 : a0;
#if defined(__HIPCC__)
            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(w[(c + LW - M + t) % LW]), "v"(R.t2[t]));
#else
            acc.x = __builtin_fmaf(w[(c + LW - M + t) % LW].x, R.t2[t].x, acc.x); acc.y = __builtin_fmaf(w[(c + LW - M + t) % LW].y, R.t2[t].x, acc.y);
#endif
        }
        return a0 + a1;
    }
    // row filter pair of column COL of the quad: samples COL .. COL + 2M of the lane's NS -> (lo, hi)
    template <int COL> static WL_DEV wl_v2 row_filter(const Taps1& R, const wl_v2 (&s)[NC2]) {
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
    static WL_DEV void col_filter(const Taps1& R, const wl_v2 (&w)[LW], int c, wl_v2& aL, wl_v2& aH) {
        aL = wl_v2{0.f, 0.f}; aH = wl_v2{0.f, 0.f};
        if (kLong) {
#pragma unroll
            for (int t = 0; t < L0; ++t) fma_cc<0>(aL, w[(c + LW - M0 + t) % LW], R.tr[M - M0 + t]);
#pragma unroll
            for (int t = 0; t < L1; ++t) fma_cc<1>(aH, w[(c + LW - M1 + t) % LW], R.tr[M - M1 + t]);
            return;
        }
#pragma unroll
        for (int t = 0; t < L0; ++t) {
            if (t & 1) fma_cc<1>(aL, w[(c + LW - M0 + t) % LW], R.c0[t / 2]); else fma_cc<0>(aL, w[(c + LW - M0 + t) % LW], R.c0[t / 2]);
        }
#pragma unroll
        for (int t = 0; t < L1; ++t) {
            if (t & 1) fma_cc<1>(aH, w[(c + LW - M1 + t) % LW], R.c1[t / 2]); else fma_cc<0>(aH, w[(c + LW - M1 + t) % LW], R.c1[t / 2]);
        }
    }

    // the (ll, hl) column filter of an LL1 row above / below the plane (MODE 2; see the header): the window read backwards
    static WL_DEV void col_filter_rev(const Taps1& R, const wl_v2 (&w)[LW], int c, wl_v2& aL) {
        aL = wl_v2{0.f, 0.f};
        if (kLong) {
#pragma unroll
            for (int t = 0; t < L0; ++t) fma_cc<0>(aL, w[(c + LW + M0 - t) % LW], R.tr[M - M0 + t]);
            return;
        }
#pragma unroll
        for (int t = 0; t < L0; ++t) {
            if (t & 1) fma_cc<1>(aL, w[(c + LW + M0 - t) % LW], R.c0[t / 2]); else fma_cc<0>(aL, w[(c + LW + M0 - t) % LW], R.c0[t / 2]);
        }
    }

    // ---- level-1 wave ---------------------------------------------------------------------------------------------------
    // What bounds this kernel is each wave's own instruction stream (a wave issues at most one instruction per four cycles),
    // so the epilogue is written for few instructions: the addresses of the twelve band-pass stores of a quad row are one
    // scalar base per orientation plus ONE 32-bit lane offset, the column taps sit two to a scalar pair (the scalar file is
    // what overflows first: spilled taps come back through v_readlane + wait states).
    static WL_DEV void level1(const Args& a, const Strip& s, const WlCtx& ctx, int64_t plane0, int cw0, int lane) {
        const WlDtFwd1Args<T>& f = a.f;
        const int sub = PP > 1 ? cw0 / (CW / PP) : 0;          // PP = 2: waves 0, 1 the first plane, 2, 3 the second; PP = 4: a plane each
        const int cw = PP > 1 ? cw0 % (CW / PP) : cw0;
        const int64_t plane = plane0 + sub;
        const int q = s.qa + 64 * cw + lane;
        const bool active = q < s.qb && plane < f.NC;
        const bool own_q = q >= s.q0 && q < s.q1;
        Taps1 R;
#pragma unroll
        for (int t = 0; t < 2 * M + 1; ++t) {
            const int t0 = t - (M - M0), t1 = t - (M - M1);
            const float v0 = t0 >= 0 && t0 < L0 ? (float)f.h0[t0 >= 0 && t0 < L0 ? t0 : 0] : 0.f;
            const float v1 = t1 >= 0 && t1 < L1 ? (float)f.h1[t1 >= 0 && t1 < L1 ? t1 : 0] : 0.f;
            R.tr[t] = wl_uniform_v2(wl_v2{v0, v1});
        }
#pragma unroll
        for (int u = 0; u < (kLong ? 0 : (L0 + 1) / 2); ++u) R.c0[u] = wl_uniform_v2(wl_v2{(float)f.h0[2 * u], 2 * u + 1 < L0 ? (float)f.h0[2 * u + 1 < L0 ? 2 * u + 1 : 0] : 0.f});
#pragma unroll
        for (int u = 0; u < (kLong ? 0 : (L1 + 1) / 2); ++u) R.c1[u] = wl_uniform_v2(wl_v2{(float)f.h1[2 * u], 2 * u + 1 < L1 ? (float)f.h1[2 * u + 1 < L1 ? 2 * u + 1 : 0] : 0.f});
        if (kRot) {
            const int off2 = M - a.L2 / 2;
#pragma unroll
            for (int t = 0; t < 2 * M + 2; ++t) {
                const int u0 = t - off2, u1 = t - 1 - off2;
                wl_v2 p;
                p.x = (t <= 2 * M && u0 >= 0 && u0 < a.L2) ? a.h2[u0 >= 0 && u0 < a.L2 ? u0 : 0] : 0.f;
                p.y = (t >= 1 && u1 >= 0 && u1 < a.L2) ? a.h2[u1 >= 0 && u1 < a.L2 ? u1 : 0] : 0.f;
#if defined(__HIPCC__)
                asm volatile("" : "+v"(p));                    // (per-lane copies: 39 tap pairs do not fit the scalar file)
#endif
                R.t2[t] = p;
            }
        }
        const int soff = 16 + 8 * (active ? q - s.qa : 0) + (PP > 1 ? sub * (a.st_pitch / PP) : 0);
        // LL1 ring: cell 0 = pixel column 2 q0 - HQ.  At the plane's edges the mirrored copies go out with the pixel pair.
        const int Q = f.W / 2;
        const int l1c = 2 * q - (2 * s.q0 - HQ);
        int l1m = -1;                                           // cell of the mirrored pair (its columns swapped)
        if (q < HQ / 2 && s.q0 == 0) l1m = -2 - 2 * q + HQ;
        if (q >= Q - HQ / 2 && s.q1 == Q) l1m = 4 * Q - 2 - 2 * q - (2 * s.q0 - HQ);
        const int r_lo = 4 * s.g_lo, r_hi = 4 * s.g_hi;
        // band-pass stores: (re, im) of orientation o6 of quad (qr, q) at  hbase + ((o6 qplane + qr Q + q) 2) elements
        const unsigned qplane2 = (unsigned)(f.H / 2) * (unsigned)Q * 2u * SZ;      // bytes of one orientation plane (< 2^31: the launcher checks)
        char* const hbase = reinterpret_cast<char*>(f.highs + (size_t)plane * 12 * ((size_t)(f.H / 2) * Q));
        const unsigned voff = (unsigned)q * 2u * SZ, voff1 = (unsigned)q * SZ;
        typedef WlPair<T> Pair;
        char* const lbase = reinterpret_cast<char*>(f.ll + (size_t)plane * f.H * f.W);
        // ScatLayer output (N, 7, C, H/2, Q): my plane (n, c) of entry 0; entries 1 .. 6 follow C planes apart.  The saved
        // (re, im) / r are (N, 6, C, H/2, Q)
        const int64_t n_img = kScat ? plane / f.C : 0;
        const int c_img = kScat ? (int)(plane - n_img * f.C) : 0;
        const size_t zplane = (size_t)f.C * (f.H / 2) * Q * SZ;
        char* const zbase = reinterpret_cast<char*>(f.z) + ((size_t)n_img * f.z_bs + (size_t)c_img * ((size_t)(f.H / 2) * Q)) * SZ;
        const size_t zmag = (size_t)f.z_mag_off * SZ;
        const long zll = (long)f.z_ll_off * SZ;                                 // (negative: no lowpass entry)
        const size_t dbase = ((size_t)n_img * 6 * f.C + c_img) * ((size_t)(f.H / 2) * Q) * SZ;
        wl_v2 wa[LW], wb[LW], wc[kRot ? LW : 1];               // (MODE 6: wc = (ba of column 2q, ba of column 2q + 1))
#pragma unroll
        for (int t = 0; t < LW; ++t) wa[t] = wb[t] = wl_v2{0.f, 0.f};
#pragma unroll
        for (int t = 0; t < (kRot ? LW : 1); ++t) wc[t] = wl_v2{0.f, 0.f};
        char* const smem = ctx.smem;
        unsigned long long tbar = 0;
        const unsigned long long tstart = WL_DT12_TICK();
        for (int hb0 = 0; hb0 < s.nhb; hb0 += PERIOD) {
#pragma unroll
            for (int ph = 0; ph < PERIOD; ++ph) {
                const int hb = hb0 + ph;
                if (hb >= s.nhb) break;
                WL_DT12_SYNC();
                if (!active || hb >= s.nhb1 || (WL_DT12_ABLATE & 8)) continue;
                const char* slot = smem + a.st_off + (hb & 1) * 4 * a.st_pitch + soff;
                char* l1slot = smem + a.l1_off + (hb & 1) * 4 * a.l1_pitch;
                wl_v2 sr[kRowLoads ? 1 : 4][NC2];
                if (!kRowLoads) {
#pragma unroll
                    for (int i = 0; i < (kRowLoads ? 0 : 4); ++i)
#pragma unroll
                        for (int u = 0; u < NC2; ++u) {
                            const wl_f2 t = *reinterpret_cast<const wl_f2*>(slot + i * a.st_pitch + 8 * u);
                            sr[i][u] = wl_v2{t.x, t.y};
                        }
                }
                const int o0 = s.o_base + 4 * hb;              // LL1 rows o0 .. o0 + 3 are completed in this half-batch
                const bool outp = MODE == 2 && (o0 < 0 || o0 >= f.H);   // all four above / below the plane (o0, H: multiples of 4)
                wl_v2 pL[2], pH[2];                            // the quad's upper row: (ll, hl), (lh, hh) of its two columns
                wl_v2 pC = wl_v2{0.f, 0.f};                    // (MODE 6: its hh = C_h2 ba, both columns)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int w = (4 * ph + i) % LW;           // slot of the new input row e = o + M
                    if (kRowLoads) {
#pragma unroll
                        for (int u = 0; u < NC2; ++u) {
                            const wl_f2 t = *reinterpret_cast<const wl_f2*>(slot + i * a.st_pitch + 8 * u);
                            sr[0][u] = wl_v2{t.x, t.y};
                        }
                    }
                    wa[w] = row_filter<0>(R, sr[kRowLoads ? 0 : i]);
                    wb[w] = row_filter<1>(R, sr[kRowLoads ? 0 : i]);
                    if constexpr (kRot) wc[w] = row_filter_ba(R, sr[kRowLoads ? 0 : i]);
                    wl_v2 aL, aH, bL, bH;
                    if (outp) {                                // (wave-uniform; such rows own no band-pass output)
                        col_filter_rev(R, wa, (w + LW - M) % LW, aL);
                        col_filter_rev(R, wb, (w + LW - M) % LW, bL);
                        aH = bH = wl_v2{0.f, 0.f};
                    } else {
                        col_filter(R, wa, (w + LW - M) % LW, aL, aH);
                        col_filter(R, wb, (w + LW - M) % LW, bL, bH);
                    }
                    if (MODE == 2) {
                        wl_f2 p; p.x = aL.x; p.y = bL.x;
                        *reinterpret_cast<wl_f2*>(l1slot + i * a.l1_pitch + l1c * 4) = p;
                        if (l1m >= 0) {
                            wl_f2 m; m.x = bL.x; m.y = aL.x;
                            *reinterpret_cast<wl_f2*>(l1slot + i * a.l1_pitch + l1m * 4) = m;
                        }
                    }
                    const int o = o0 + i;
                    const bool own = own_q && o >= r_lo && o < r_hi;
                    if (MODE != 2 && f.ll && own) {            // full-resolution lowpass row o, columns 2q, 2q + 1
                        Pair z; z.a = (T)aL.x; z.b = (T)bL.x;
                        *reinterpret_cast<Pair*>(lbase + (size_t)((unsigned)o * (unsigned)f.W * SZ) + 2 * voff1) = z;
                    }
                    wl_v2 cH = wl_v2{0.f, 0.f};
                    if constexpr (kRot) cH = col_filter_ba(R, wc, (w + LW - M) % LW);
                    if (!(i & 1)) { pL[0] = aL; pL[1] = bL; pH[0] = aH; pH[1] = bH; pC = cH; continue; }
                    // the quad (rows o - 1, o; columns 2q, 2q + 1) is complete: q2c of lh, hh, hl (reference
                    // transform_funcs.py:61-72: p = (upper left, upper right, lower left, lower right))
                    if (!own || (WL_DT12_ABLATE & 2)) continue;
                    const float k = (float)WL_SQRT1_2;
                    const float lh0 = pH[0].x, lh1 = pH[1].x, lh2 = aH.x, lh3 = bH.x;
                    const float hh0 = kRot ? pC.x : pH[0].y, hh1 = kRot ? pC.y : pH[1].y, hh2 = kRot ? cH.x : aH.y, hh3 = kRot ? cH.y : bH.y;
                    const float hl0 = pL[0].y, hl1 = pL[1].y, hl2 = aL.y, hl3 = bL.y;
                    const unsigned qrow = (unsigned)((o - 1) / 2) * (unsigned)Q;
                    if (!kScat) {
                        if (!f.highs) continue;
                        char* const rowp = hbase + (size_t)(qrow * 2u * SZ);       // (uniform)
                        Pair z;
                        z.a = (T)((lh0 - lh3) * k); z.b = (T)((lh1 + lh2) * k); *reinterpret_cast<Pair*>(rowp + voff) = z;
                        z.a = (T)((hh0 - hh3) * k); z.b = (T)((hh1 + hh2) * k); *reinterpret_cast<Pair*>(rowp + (size_t)qplane2 + voff) = z;
                        z.a = (T)((hl0 - hl3) * k); z.b = (T)((hl1 + hl2) * k); *reinterpret_cast<Pair*>(rowp + 2 * (size_t)qplane2 + voff) = z;
                        z.a = (T)((hl0 + hl3) * k); z.b = (T)((hl1 - hl2) * k); *reinterpret_cast<Pair*>(rowp + 3 * (size_t)qplane2 + voff) = z;
                        z.a = (T)((hh0 + hh3) * k); z.b = (T)((hh1 - hh2) * k); *reinterpret_cast<Pair*>(rowp + 4 * (size_t)qplane2 + voff) = z;
                        z.a = (T)((lh0 + lh3) * k); z.b = (T)((lh1 - lh2) * k); *reinterpret_cast<Pair*>(rowp + 5 * (size_t)qplane2 + voff) = z;
                    } else {
                        // ScatLayer: |z_o| smoothed.  re^2 + im^2 = ((v0 -+ v3)^2 + (v1 +- v2)^2) / 2
                        const float b = (float)f.magbias, b2 = b * b;
                        char* const zp = zbase + (size_t)(qrow * SZ) + voff1;       // (n, 0, c, qr, q)
                        if (zll >= 0) *reinterpret_cast<T*>(zp + zll) = (T)((pL[0].x + pL[1].x + aL.x + bL.x) * 0.25f);
                        const float v0[3] = {lh0, hh0, hl0}, v1[3] = {lh1, hh1, hl1}, v2[3] = {lh2, hh2, hl2}, v3[3] = {lh3, hh3, hl3};
#pragma unroll
                        for (int u = 0; u < 3; ++u) {
#pragma unroll
                            for (int w2i = 0; w2i < 2; ++w2i) {
                                const int o6 = w2i ? 5 - u : u;                     // lh: 0 / 5, hh: 1 / 4, hl: 2 / 3
                                const float d = w2i ? v0[u] + v3[u] : v0[u] - v3[u];
                                const float e = w2i ? v1[u] - v2[u] : v1[u] + v2[u];
                                const float r = wl_sqrt(0.5f * (d * d + e * e) + b2);
                                *reinterpret_cast<T*>(zp + zmag + (size_t)o6 * zplane) = (T)(r - b);
                                if (MODE == 3) {
                                    const float ir = k * wl_rcp(r);
                                    char* const dp = reinterpret_cast<char*>(f.drdx) + dbase + (size_t)o6 * zplane + (size_t)(qrow * SZ) + voff1;
                                    char* const dq = reinterpret_cast<char*>(f.drdy) + dbase + (size_t)o6 * zplane + (size_t)(qrow * SZ) + voff1;
                                    *reinterpret_cast<T*>(dp) = (T)(d * ir);
                                    *reinterpret_cast<T*>(dq) = (T)(e * ir);
                                }
                            }
                        }
                    }
                }
            }
        }
        report(a, ctx, lane, cw == 0, 0, tstart, tbar);
    }

    // ---- level-2 wave ---------------------------------------------------------------------------------------------------
    // Lanes 0-31 of a wave own the EVEN rows of 32 column groups, lanes 32-63 the ODD rows of the same groups: a row of LL1
    // meets only the even-row taps (h0b, h1b) or only the odd-row taps (h0a, h1a) of the column filters, so the two halves
    // share nothing until the epilogue, where q2c mixes them (six values cross the wave by ds_bpermute).  Half the chain
    // per lane: the level-2 waves were what every barrier waited for (in-kernel counters: busy 89 % of the workgroup's time
    // with one lane per column group).
    static WL_DEV void level2(const Args& a, const Strip& s, const WlCtx& ctx, int64_t plane, int qw, int lane) {
        const WlDtFwd1Args<T>& f = a.f;
        const int half = lane >> 5;                            // 0: even rows, 1: odd rows
        const int j0 = 32 * qw + (lane & 31);                  // LL1 columns 2 q0 + 4 j .. + 3
        const bool active = 2 * j0 < s.q1 - s.q0;
        const int j = active ? j0 : 0;                         // (idle lanes run along on group 0: the shuffles need every lane)
        wl_v2 tE[LQ], tO[LQ];                                  // row filters: even samples meet (h0b, h1b), odd ones (h0a, h1a)
#pragma unroll
        for (int t = 0; t < LQ; ++t) {
            tE[t] = wl_uniform_v2(wl_v2{a.h0b[t], a.h1b[t]});
            tO[t] = wl_uniform_v2(wl_v2{a.h0a[t], a.h1a[t]});
        }
        wl_v2 tC[LQ];                                          // column filter of my rows' parity (per lane: vector registers)
#pragma unroll
        for (int t = 0; t < LQ; ++t) tC[t] = half ? wl_v2{a.h0a[t], a.h1a[t]} : wl_v2{a.h0b[t], a.h1b[t]};
        char* const smem = ctx.smem;
        // my window: LQ rows of one parity, slot r at win[r * 64 QW].  The rows of half-batch hb go into slots
        // 2 (hb mod P2) + {0, 1} (P2 = LQ / 2 half-batches fill the window once), and the loop is unrolled by P2 so that
        // every slot is a compile-time offset: what bounds the kernel is the number of instructions a SIMD has to issue,
        // scalar ones included.
        wl_vf4* const win = reinterpret_cast<wl_vf4*>(smem + a.w2_off) + 64 * qw + lane;
        const int Q = f.W / 2, Q2 = f.W / 4;
        // LL2 rows of my parity: (R + half) w2 + Cc elements; level-2 band-pass: ((o6 qplane + (R / 2) Q2 + Cc / 2) 2) elements
        char* const lbase = reinterpret_cast<char*>(a.ll2 + (size_t)plane * (f.H / 2) * Q + (size_t)half * Q);
        char* const hbase = reinterpret_cast<char*>(a.highs2 + (size_t)plane * 12 * ((size_t)(f.H / 4) * Q2));
        const unsigned qplane2 = (unsigned)(f.H / 4) * (unsigned)Q2 * 2u * SZ;    // bytes of one orientation plane
        const unsigned Cc = (unsigned)(s.q0 + 2 * j);
        const unsigned lvoff = Cc * SZ, hvoff = (Cc / 2) * 2u * SZ + (half ? 5u * qplane2 : 0u);
        const long hstep = half ? -(long)qplane2 : (long)qplane2;               // my three orientations: 0, 1, 2 or 5, 4, 3
        typedef WlPair<T> Pair;
        // MODE 5: image n, channel c of the (N, .., C, H/4, W/4) output: the averaged lowpass and magnitude 0 (the others C planes apart)
        const int64_t n_img = MODE == 5 ? plane / f.C : 0;
        const size_t zq = (size_t)(f.H / 4) * Q2;
        char* const zimg = reinterpret_cast<char*>(f.z) + ((size_t)n_img * f.z_bs + (size_t)(plane - n_img * f.C) * zq) * SZ;
        char* const zll = zimg + f.z_ll_off * SZ;                               // (z_ll_off < 0: no lowpass entry)
        char* const zmg = zimg + f.z_mag_off * SZ;
        const size_t zstep = (size_t)f.C * zq * SZ;
        const int l1lane = half * a.l1_pitch + 16 * j;
        // MODE 2: the level-1 lanes fill ring slot hb & 1 AFTER barrier hb, level 2 reads it one half-batch later; MODE 4:
        // the stagers fill it BEFORE barrier hb, level 2 reads it right after
        static const int LAG = MODE == 2 ? 1 : 0;
        const int G0 = s.o_base / 4 - LAG;                                     // LL1 group in the ring at half-batch hb: G0 + hb
        unsigned long long tbar = 0;
        const unsigned long long tstart = WL_DT12_TICK();
        static const int P2 = LQ / 2;
        for (int hb0 = 0; hb0 < s.nhb; hb0 += P2) {
#pragma unroll
            for (int ph = 0; ph < P2; ++ph) {
                const int hb = hb0 + ph;
                if (hb >= s.nhb) break;
                WL_DT12_SYNC();
                if (hb < LAG || hb - LAG >= s.nhb1 || (WL_DT12_ABLATE & 1)) continue;
                const int G = G0 + hb;
                if (G < s.g_lo - HG) continue;                                 // rows of the level-1 warm-up
                const char* l1slot = smem + a.l1_off + ((hb - LAG) & 1) * 4 * a.l1_pitch + l1lane;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    // dual-tree row filters of the 2 LQ samples X[4k + 2 - LQ ..]: k = my column group
                    wl_v2 aE = {0.f, 0.f}, aO = {0.f, 0.f};                    // (lo[2k], hi[2k+1]), (lo[2k+1], hi[2k])
#pragma unroll
                    for (int u = 0; u < LQ / 2; ++u) {
                        const wl_vf4 x4 = *reinterpret_cast<const wl_vf4*>(l1slot + 2 * i * a.l1_pitch + 16 * u);
                        const wl_v2 s0 = {x4.x, x4.y}, s1 = {x4.z, x4.w};
                        wl_pk_fma_x(aE, tE[2 * u], s0); wl_pk_fma_y(aO, tO[2 * u], s0);
                        wl_pk_fma_x(aE, tE[2 * u + 1], s1); wl_pk_fma_y(aO, tO[2 * u + 1], s1);
                    }
                    wl_vf4 w; w.x = aE.x; w.y = aE.y; w.z = aO.x; w.w = aO.y;
                    win[(2 * ph + i) * (64 * QW)] = w;
                }
                const int kr = G - HG;                                         // its window: rows 4 kr - HQ .. 4 kr + HQ + 3
                if (kr < s.g_lo || kr >= s.g_hi) continue;
                // column filters over my LQ rows, oldest first (slot 2 (ph + 1) mod LQ): xL? of the lowpass-H plane, xH? highpass-H
                wl_v2 xL0 = {0.f, 0.f}, xL1 = {0.f, 0.f}, xH0 = {0.f, 0.f}, xH1 = {0.f, 0.f};
#pragma unroll
                for (int t = 0; t < LQ; ++t) {
                    const wl_vf4 v = win[((2 * (ph + 1) + t) % LQ) * (64 * QW)];
                    const wl_v2 vE = {v.x, v.y}, vO = {v.z, v.w};              // (lo0, hi1), (lo1, hi0) of the row
                    wl_pk_fma_x_v(xL0, tC[t], vE); wl_pk_fma_y_v(xH1, tC[t], vE);
                    wl_pk_fma_x_v(xL1, tC[t], vO); wl_pk_fma_y_v(xH0, tC[t], vO);
                }
                if ((WL_DT12_ABLATE & 4) && xL0.x != 12345.f) continue;
                if (active && MODE != 5) {
                    Pair p0;
                    p0.a = (T)xL0.x; p0.b = (T)xL1.x;
                    *reinterpret_cast<Pair*>(lbase + (size_t)((unsigned)(2 * kr) * (unsigned)Q * SZ) + lvoff) = p0;
                }
                // q2c needs both parities: e? = the even-row lane's x?, o? = the odd-row lane's
                const int partner = lane ^ 32;
                const float pL0y = wl_shfl(xL0.y, partner), pL1y = wl_shfl(xL1.y, partner);
                const float pH0x = wl_shfl(xH0.x, partner), pH1x = wl_shfl(xH1.x, partner);
                const float pH0y = wl_shfl(xH0.y, partner), pH1y = wl_shfl(xH1.y, partner);
                // lh = {oL0y, oL1y, eL0y, eL1y}, hl = {eH0x, eH1x, oH0x, oH1x}, hh = {oH0y, oH1y, eH0y, eH1y} (e = even-row lane);
                // the even-row lane forms (v0 - v3, v1 + v2) = orientations 0, 1, 2, the odd-row lane (v0 + v3, v1 - v2) = 5, 4, 3
                const float sg = half ? 1.f : -1.f;
                const float k = (float)WL_SQRT1_2;
                float re[3], im[3];
                {   // lh: v0 = oL0y, v1 = oL1y, v2 = eL0y, v3 = eL1y
                    const float v0 = half ? xL0.y : pL0y, v1 = half ? xL1.y : pL1y, v2 = half ? pL0y : xL0.y, v3 = half ? pL1y : xL1.y;
                    re[0] = (v0 + sg * v3) * k; im[0] = (v1 - sg * v2) * k;
                }
                {   // hh: v0 = oH0y, v1 = oH1y, v2 = eH0y, v3 = eH1y
                    const float v0 = half ? xH0.y : pH0y, v1 = half ? xH1.y : pH1y, v2 = half ? pH0y : xH0.y, v3 = half ? pH1y : xH1.y;
                    re[1] = (v0 + sg * v3) * k; im[1] = (v1 - sg * v2) * k;
                }
                {   // hl: v0 = eH0x, v1 = eH1x, v2 = oH0x, v3 = oH1x
                    const float v0 = half ? pH0x : xH0.x, v1 = half ? pH1x : xH1.x, v2 = half ? xH0.x : pH0x, v3 = half ? xH1.x : pH1x;
                    re[2] = (v0 + sg * v3) * k; im[2] = (v1 - sg * v2) * k;
                }
                if (MODE == 5) {
                    // ScatLayerj2 (scatternet/lowlevel.py:237-262): the 2x2 average of LL2 (my row's two columns + the other
                    // parity's) and the smoothed magnitudes of my three orientations
                    const float mine = xL0.x + xL1.x;
                    const float avg = 0.25f * (mine + wl_shfl(mine, partner));
                    const float b = (float)f.magbias, b2 = b * b;
                    if (active) {
                        const unsigned zoff = ((unsigned)kr * (unsigned)Q2 + Cc / 2) * SZ;
                        if (!half && f.z_ll_off >= 0) *reinterpret_cast<T*>(zll + zoff) = (T)avg;
                        char* zp = zmg + zoff + (half ? 5 * zstep : 0);
#pragma unroll
                        for (int u = 0; u < 3; ++u) {
                            *reinterpret_cast<T*>(zp) = (T)(wl_sqrt(re[u] * re[u] + im[u] * im[u] + b2) - b);
                            zp += half ? -(long)zstep : (long)zstep;
                        }
                    }
                } else if (active) {
                    char* hp = hbase + (size_t)((unsigned)kr * (unsigned)Q2 * 2u * SZ) + hvoff;
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        Pair p; p.a = (T)re[u]; p.b = (T)im[u];
                        *reinterpret_cast<Pair*>(hp) = p;
                        hp += hstep;
                    }
                }
            }
        }
        report(a, ctx, lane, qw == 0, 2, tstart, tbar);
    }

    static WL_DEV void run(const Args& a, const WlCtx& ctx) {
        const int tid = ctx.tid;
        const int wave = wl_uniform(tid >> 6), lane = tid & 63;
        const int64_t lbid = wl_xcd_remap(ctx.bid, a.nblocks);
        const int per_plane = a.nstrips * a.nseg;
        const int64_t pidx = lbid / per_plane;
        const int64_t plane = pidx * PP;                       // (PP = 2: the first of the workgroup's two planes)
        const int rem = (int)(lbid - pidx * per_plane);
        const int seg = rem / a.nstrips, strip = rem - seg * a.nstrips;
        const Strip s = geometry(a, strip, seg);
        if (wave >= CW + QW) {
            const int ngl = (PP * s.ng + 63) >> 6;
            if (ngl <= 1) stager<1>(a, s, ctx, plane, lane, wave - CW - QW);
            else if (ngl == 2) stager<2>(a, s, ctx, plane, lane, wave - CW - QW);
            else stager<3>(a, s, ctx, plane, lane, wave - CW - QW);
        } else if (((MODE == 2 || kL2)) && wave >= CW) {
            level2(a, s, ctx, plane, wave - CW, lane);
        } else {
            if constexpr (!kL2) level1(a, s, ctx, plane, wave, lane);
        }
    }
};

// =================================================================================================================
// Streaming level >= 2 DTCWT inverse over column strips: inv_j2plus (reference dtcwt/transform_funcs.py:279-307 = c2q x 3,
// 4 colifilt, 4 rowifilt (dtcwt/lowlevel.py:154-239), 3 adds) for filters with an ODD half length m2 = L / 2 (10, 14,
// 18 taps).  With quads of the half-resolution inputs (pair of samples 2j, 2j + 1 along an axis):
//     Y[4q + s] = sum_{t < m2} h_s[e_s + 2t] X[2 (q - D2 + t) + p_s],   D2 = (m2 - 1) / 2,   e = (1, 1, 0, 0)[s]
//   lowpass streams (ha, hb) = (g0b, g0a): s = 0, 2 take ha and the EVEN sample (p = 0), s = 1, 3 take hb and the ODD one;
//   highpass streams (g1b, g1a): the parities are the other way round.
// The separable operators commute: y = C_lp (R_lp ll + R_hp hl) + C_hp (R_lp lh + R_hp hh), row interpolation first.
//   * half-batch = one input quad row = four output rows.  Every lane of the stager waves owns ONE input quad of the quad
//     row (the loads, the c2q and the flipped / mirrored copies of WlDtInv1Strip) and stages it as two 32-byte cells
//     (ll_e, ll_o, hl_e, hl_o | lh_e, lh_o, hh_e, hh_o), one per row;
//   * a compute lane owns TWO of the four output columns of an input quad column: waves 0, 1 the columns 4q, 4q + 1
//     (tap phase e = 1), waves 2, 3 the columns 4q + 2, 4q + 3 (e = 0) - so the taps are wave-uniform scalars.  The pair
//     (even sample, odd sample) of a band meets the pair (ha[e + 2t], hb[e + 2t]) elementwise: one packed FMA gives both
//     columns (the highpass streams read the pair swapped: op_sel).  (A, B) = (R ll + R hl, R lh + R hh) of its two
//     columns go into circular register windows of m2 quad rows whose rotation is compile-time (loop unrolled by m2);
//     the column interpolation of output rows 4 kr .. 4 kr + 3 runs over the window D2 quad rows later.
// =================================================================================================================
template <typename T>
struct WlDtI2StripArgs {
    WlDtInv2Args<T> f;             // tensors, taps, sizes (h, w: the half-resolution input)
    int64_t nblocks;
    int nstrips, strip_quads;      // input quad columns per strip
    int nseg, seg_groups;          // input quad rows (= groups of 4 output rows) per segment
    int st_off, st_pitch, lds_bytes;
};

template <typename T, int LQ>
struct WlDtInv2Strip {
    typedef WlDtI2StripArgs<T> Args;
    static const int CW = 4, SW = 3;                   // compute waves (2 per column phase), stager waves
    static const int kWaves = CW + SW;
    static const int kThreads = 64 * kWaves;
    static const int kMinWaves = LQ <= 10 ? 7 : 5;     // four (two) workgroups of 7 waves per CU: at most 72 (96) registers
    static const int SZ = (int)sizeof(T);
    static const int m2 = LQ / 2, D2 = (m2 - 1) / 2;
    static_assert(m2 & 1, "odd half length: 10, 14 or 18 taps");
    static const int QCAP = 32 * CW;                   // input quad columns per strip

    struct Strip {
        int q0, q1;            // input quad columns [q0, q1) -> output columns [4 q0, 4 q1)
        int Qa, nq;            // quad columns [Qa, Qa + nq) inside the plane are loaded, one per stager lane
        int c0;                // quad column of staged cell 0 = q0 - D2
        int g_lo, g_hi;        // output groups (= input quad rows) of this segment
        int nhb;
    };
    static WL_HD Strip geometry(const Args& a, int strip, int seg) {
        Strip s;
        const int W2 = a.f.w / 2, H2 = a.f.h / 2;
        s.q0 = strip * a.strip_quads;
        s.q1 = s.q0 + a.strip_quads < W2 ? s.q0 + a.strip_quads : W2;
        s.c0 = s.q0 - D2;
        int qa = s.c0 < 0 ? 0 : s.c0, qb = s.q1 - 1 + D2;
        if (qb > W2 - 1) qb = W2 - 1;
        s.Qa = qa; s.nq = qb - qa + 1;
        s.g_lo = seg * a.seg_groups;
        s.g_hi = s.g_lo + a.seg_groups < H2 ? s.g_lo + a.seg_groups : H2;
        s.nhb = (s.g_hi - s.g_lo + 2 * D2 + 1) / 2 * 2;        // (even: the stagers alternate two register sets)
        return s;
    }
    // extended quad row -> source quad row; flip: its two rows swap (symmetric extension of an even number of rows)
    static WL_HD int src_quad_row(int eq, int H2, bool& flip) {
        flip = false;
        if ((unsigned)eq < (unsigned)H2) return eq;
        flip = true;
        const int m = eq < 0 ? -1 - eq : 2 * H2 - 1 - eq;
        return m < 0 ? 0 : (m >= H2 ? H2 - 1 : m);             // (one fold: the launcher requires H2 > D2)
    }

    typedef T Pair2 __attribute__((ext_vector_type(2), may_alias));
    struct Quad { Pair2 l0, l1, b[6]; };

    static WL_DEV void stager(const Args& a, const Strip& s, const WlCtx& ctx, int64_t plane, int lane, int sidx) {
        const WlDtInv2Args<T>& f = a.f;
        const int H2 = f.h / 2, W2 = f.w / 2;
        const size_t qplane = (size_t)H2 * W2;
        const int j = 64 * sidx + lane;
        const int Q = s.Qa + j;
        const bool qon = j < s.nq;
        const T* llp = f.ll + (size_t)plane * f.ll_plane_stride + 2 * Q;
        const T* hp = f.highs + (size_t)plane * 6 * qplane * 2 + 2 * Q;
        const int cdst = (Q - s.c0) * 16;                      // (the two 16-byte halves of a cell live in the two halves of a row: lanes read consecutive words)
        const int hp2 = a.st_pitch / 2;
        int mdst = -1;                                         // the mirrored quad column (its samples swapped), if the strip reads it
        if (qon) {
            if (Q < D2 && -1 - Q >= s.c0) mdst = (-1 - Q - s.c0) * 16;
            if (Q >= W2 - D2 && 2 * W2 - 1 - Q <= s.q1 - 1 + D2) mdst = (2 * W2 - 1 - Q - s.c0) * 16;
        }
        const int eq0 = s.g_lo - D2;
        auto load = [&](int h, Quad& qd) {
            bool flip;
            const int sq = src_quad_row(eq0 + h, H2, flip);
            if (!qon) return;
            qd.l0 = *reinterpret_cast<const Pair2*>(llp + (size_t)(2 * sq) * f.ll_row_stride);
            qd.l1 = *reinterpret_cast<const Pair2*>(llp + (size_t)(2 * sq + 1) * f.ll_row_stride);
#pragma unroll
            for (int o = 0; o < 6; ++o) qd.b[o] = *reinterpret_cast<const Pair2*>(hp + ((size_t)o * qplane + (size_t)sq * W2) * 2);
        };
        const float k = (float)WL_SQRT1_2;
        auto stage = [&](int hb, const Quad& qd) {
            bool flip;
            src_quad_row(eq0 + hb, H2, flip);
            char* sslot = ctx.smem + a.st_off + (hb & 1) * 2 * a.st_pitch;
            if (!qon) return;
            float re[6], im[6];
#pragma unroll
            for (int o = 0; o < 6; ++o) { re[o] = (float)qd.b[o].x; im[o] = (float)qd.b[o].y; }
            // c2q (dtcwt/lowlevel.py:263-295): orientation pairs (0,5) -> lh, (2,3) -> hl, (1,4) -> hh;  v[row][col][ll, hl, lh, hh]
            float v[2][2][4];
#pragma unroll
            for (int ch = 1; ch < 4; ++ch) {
                const int o1 = ch == 2 ? 0 : (ch == 1 ? 2 : 1), o2 = ch == 2 ? 5 : (ch == 1 ? 3 : 4);
                v[0][0][ch] = (re[o1] + re[o2]) * k; v[0][1][ch] = (im[o1] + im[o2]) * k;
                v[1][0][ch] = (im[o1] - im[o2]) * k; v[1][1][ch] = (re[o2] - re[o1]) * k;
            }
            v[0][0][0] = (float)qd.l0.x; v[0][1][0] = (float)qd.l0.y; v[1][0][0] = (float)qd.l1.x; v[1][1][0] = (float)qd.l1.y;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                char* drow = sslot + (flip ? 1 - i : i) * a.st_pitch;
                wl_vf4 w0, w1;
                w0.x = v[i][0][0]; w0.y = v[i][1][0]; w0.z = v[i][0][1]; w0.w = v[i][1][1];      // ll_e ll_o hl_e hl_o
                w1.x = v[i][0][2]; w1.y = v[i][1][2]; w1.z = v[i][0][3]; w1.w = v[i][1][3];      // lh_e lh_o hh_e hh_o
                *reinterpret_cast<wl_vf4*>(drow + cdst) = w0;
                *reinterpret_cast<wl_vf4*>(drow + hp2 + cdst) = w1;
                if (mdst >= 0) {
                    wl_vf4 m0, m1;
                    m0.x = w0.y; m0.y = w0.x; m0.z = w0.w; m0.w = w0.z;
                    m1.x = w1.y; m1.y = w1.x; m1.z = w1.w; m1.w = w1.z;
                    *reinterpret_cast<wl_vf4*>(drow + mdst) = m0;
                    *reinterpret_cast<wl_vf4*>(drow + hp2 + mdst) = m1;
                }
            }
        };
        Quad qa, qb;
        load(0, qa);
        for (int hb = 0; hb < s.nhb; hb += 2) {                // (nhb is even)
            load(hb + 1, qb);
            stage(hb, qa);
            ctx.sync();
            load(hb + 2 < s.nhb ? hb + 2 : s.nhb - 1, qa);
            stage(hb + 1, qb);
            ctx.sync();
        }
    }

    // acc += taps (.) v  /  acc += taps (.) (v.y, v.x): elementwise, the tap pair in scalar registers
    static WL_DEV void fma_ee(wl_v2& acc, wl_v2 taps, wl_v2 v) {
#if defined(__HIPCC__)
        asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "s"(taps), "v"(v));
#else
        acc.x = __builtin_fmaf(taps.x, v.x, acc.x); acc.y = __builtin_fmaf(taps.y, v.y, acc.y);
#endif
    }
    static WL_DEV void fma_sw(wl_v2& acc, wl_v2 taps, wl_v2 v) {
#if defined(__HIPCC__)
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "s"(taps), "v"(v));
#else
        acc.x = __builtin_fmaf(taps.x, v.y, acc.x); acc.y = __builtin_fmaf(taps.y, v.x, acc.y);
#endif
    }
    // acc += w * (c, c), c = the scalar pair's low / high half
    template <int HI> static WL_DEV void fma_cc(wl_v2& acc, wl_v2 w, wl_v2 pair) {
#if defined(__HIPCC__)
        if (HI) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w), "s"(pair));
        else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(w), "s"(pair));
#else
        const float c = HI ? pair.y : pair.x;
        acc.x = __builtin_fmaf(w.x, c, acc.x); acc.y = __builtin_fmaf(w.y, c, acc.y);
#endif
    }

    static WL_DEV void compute(const Args& a, const Strip& s, const WlCtx& ctx, int64_t plane, int cw, int lane) {
        const WlDtInv2Args<T>& f = a.f;
        const int ph_c = cw >> 1;                              // column phase: 0 -> output columns 4q, 4q+1 (e = 1); 1 -> 4q+2, 4q+3 (e = 0)
        const int q = s.q0 + 64 * (cw & 1) + lane;
        const bool active = q < s.q1;
        // tap pairs (ha[e + 2t], hb[e + 2t]): lowpass (g0b, g0a), highpass (g1b, g1a); e = 1 and e = 0 sets (the column
        // interpolation needs both, the row interpolation the one of my phase)
        wl_v2 PL[2][m2], PH[2][m2];
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int t = 0; t < m2; ++t) {
                PL[e][t] = wl_uniform_v2(wl_v2{(float)f.g0b[e + 2 * t], (float)f.g0a[e + 2 * t]});
                PH[e][t] = wl_uniform_v2(wl_v2{(float)f.g1b[e + 2 * t], (float)f.g1a[e + 2 * t]});
            }
        const int er = 1 - ph_c;                               // my row-interpolation tap phase
        const int soff = (active ? q - s.q0 : 0) * 16;         // cell of quad column q - D2
        const int hp2 = a.st_pitch / 2;
        char* const yp = reinterpret_cast<char*>(f.y + (size_t)plane * (4 * (size_t)f.h * f.w));
        const unsigned rowb = (unsigned)(2 * f.w) * SZ, colb = (unsigned)(4 * q + 2 * ph_c) * SZ;
        wl_v2 wA[m2][2], wB[m2][2];                            // window: [quad row slot][row of the quad] = (col s0, col s1)
#pragma unroll
        for (int t = 0; t < m2; ++t) { wA[t][0] = wA[t][1] = wB[t][0] = wB[t][1] = wl_v2{0.f, 0.f}; }
        char* const smem = ctx.smem;
        for (int hb0 = 0; hb0 < s.nhb; hb0 += m2) {
#pragma unroll
            for (int ph = 0; ph < m2; ++ph) {
                const int hb = hb0 + ph;
                if (hb >= s.nhb) break;
                ctx.sync();
                if (!active) continue;
                const char* slot = smem + a.st_off + (hb & 1) * 2 * a.st_pitch + soff;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    wl_v2 A = {0.f, 0.f}, B = {0.f, 0.f};
#pragma unroll
                    for (int t = 0; t < m2; ++t) {
                        const wl_vf4 c0 = *reinterpret_cast<const wl_vf4*>(slot + i * a.st_pitch + 16 * t);
                        const wl_vf4 c1 = *reinterpret_cast<const wl_vf4*>(slot + i * a.st_pitch + hp2 + 16 * t);
                        if (er) {
                            fma_ee(A, PL[1][t], wl_v2{c0.x, c0.y}); fma_sw(A, PH[1][t], wl_v2{c0.z, c0.w});
                            fma_ee(B, PL[1][t], wl_v2{c1.x, c1.y}); fma_sw(B, PH[1][t], wl_v2{c1.z, c1.w});
                        } else {
                            fma_ee(A, PL[0][t], wl_v2{c0.x, c0.y}); fma_sw(A, PH[0][t], wl_v2{c0.z, c0.w});
                            fma_ee(B, PL[0][t], wl_v2{c1.x, c1.y}); fma_sw(B, PH[0][t], wl_v2{c1.z, c1.w});
                        }
                    }
                    wA[ph][i] = A; wB[ph][i] = B;
                }
                // the window now holds input quad rows eq - m2 + 1 .. eq (eq = g_lo - D2 + hb), oldest in slot ph + 1:
                // the output group kr = eq - D2
                const int kr = s.g_lo - 2 * D2 + hb;
                if (kr < s.g_lo || kr >= s.g_hi) continue;
                wl_v2 y0 = {0.f, 0.f}, y1 = {0.f, 0.f}, y2 = {0.f, 0.f}, y3 = {0.f, 0.f};
#pragma unroll
                for (int t = 0; t < m2; ++t) {
                    const int sl = (ph + 1 + t) % m2;
                    // rows 4kr, 4kr+2: lowpass on the EVEN row of the quad row, highpass on the ODD one; 4kr+1, 4kr+3: the other way round
                    fma_cc<0>(y0, wA[sl][0], PL[1][t]); fma_cc<0>(y0, wB[sl][1], PH[1][t]);
                    fma_cc<1>(y1, wA[sl][1], PL[1][t]); fma_cc<1>(y1, wB[sl][0], PH[1][t]);
                    fma_cc<0>(y2, wA[sl][0], PL[0][t]); fma_cc<0>(y2, wB[sl][1], PH[0][t]);
                    fma_cc<1>(y3, wA[sl][1], PL[0][t]); fma_cc<1>(y3, wB[sl][0], PH[0][t]);
                }
                typedef T Vec2 __attribute__((ext_vector_type(2)));
                char* const rp = yp + (size_t)((unsigned)(4 * kr) * rowb) + colb;
                *reinterpret_cast<Vec2*>(rp) = Vec2{(T)y0.x, (T)y0.y};
                *reinterpret_cast<Vec2*>(rp + rowb) = Vec2{(T)y1.x, (T)y1.y};
                *reinterpret_cast<Vec2*>(rp + 2 * (size_t)rowb) = Vec2{(T)y2.x, (T)y2.y};
                *reinterpret_cast<Vec2*>(rp + 3 * (size_t)rowb) = Vec2{(T)y3.x, (T)y3.y};
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
        if (wave >= CW) stager(a, s, ctx, plane, lane, wave - CW);
        else compute(a, s, ctx, plane, wave, lane);
    }
};
