// Streaming multi-level 2-D DWT analysis: one workgroup owns one whole (n,c) plane and marches down it.
//
// Why: a per-level tile kernel moves 1.32x the compulsory bytes of a 3-level transform (LL_1, LL_2 round trips), pays
// two small launches, and its 259-float band rows leave in 256-byte pieces that never line up with cache lines.  Here
//   * every input row is read from HBM exactly once, as whole 1 KiB wave-wide LDS-DMA loads (global_load_lds_dwordx4)
//     issued D half-batches ahead by a dedicated loader wave - no VGPRs, no LDS write instructions, no halo re-reads;
//   * a lane owns ONE output column of its level and keeps the L-row sliding window of that column's row-filtered
//     (lo,hi) pair in registers: per pair of new source rows it runs the row filter (samples straight from the LDS
//     ring) and the column filter (registers only) - one LDS crossing per sample, no intermediate planes, one
//     barrier per four input rows;
//   * LL_j rows go to a small LDS ring that the waves of level j+1 consume a few rows behind - LL_1 .. LL_{J-1}
//     never touch HBM; HBM traffic = x in + yl, yh[j] out = the algorithmic minimum of SURVEY.md 8(d);
//   * every band row of every level is written by consecutive lanes of consecutive waves within one half-batch, so
//     the partial cache lines at its ends are completed in the same L2 a few hundred cycles later.
// Roles are per WAVE (wave-uniform branches, each role has its own lean body): level-1 / level-2 / level-3 compute
// waves and the loader.  All waves follow the same deterministic schedule (WlRowsSched): per half-batch level 1
// consumes 4 extended input rows (2 feeds), level j+1 consumes whatever LL_j rows were complete at the barrier.
// The launcher simulates the same schedule on the host and refuses geometries whose rings would be overrun.
//
// Restates DWTForward.forward's level loop (reference dwt/transform2d.py:63-74) = J x AFB2D.forward
// (dwt/lowlevel.py:336-347) = 2J x afb1d (:91-172): filter along W, then along H, per level.
#pragma once
#include "wl_common.h"

#define WL_ROWS_MAXLEV 3
#define WL_ROWS_WAVES 11
// WL_ROWS_ABLATE (tools/ab builds only, never defined in the product): 1 = no global stores, 2 = no DMA loads
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

struct WlRowsLevel {
    int Hs, Ws;         // source rows / cols of this level
    int Kh, Kw;         // output rows / cols
    int ring_off;       // LDS byte offset of this level's SOURCE ring (level 0: the input ring)
    int ring_pitch;     // bytes per ring row; the element at column Ws of every ring row is a permanent zero
    int wave0, nwaves;  // compute waves of this level: wave0 .. wave0+nwaves-1, 64 columns each
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
    int nhb;            // half-batches (= barriers of the main loop) until every level has finished
    int ring_rows;      // rows of the LL rings (power of two)
    int zero_off;       // LDS byte offset of an all-zero row (zero padding above / below the plane)
    int zero_bytes;
    int loader_wave;
    WlRowsLevel g[WL_ROWS_MAXLEV];
};

// The schedule every wave (and the launcher) steps through: fed[j] = feeds level j has consumed.  A feed is one pair
// of extended source rows (2f+base, 2f+base+1); the first (L-2)/2 feeds of a level only fill its window.
struct WlRowsSched {
    int fed[WL_ROWS_MAXLEV];
    WL_HD void init() { for (int j = 0; j < WL_ROWS_MAXLEV; ++j) fed[j] = 0; }
    template <typename A> WL_HD int feeds_total(const A& a, int j, int LT) const { return a.g[j].Kh + (LT - 2) / 2; }
    WL_HD int emitted(int j, int LT) const { const int e = fed[j] - (LT - 2) / 2; return e > 0 ? e : 0; }
    // feeds level j runs in the half-batch that starts now; only looks at the state as of the barrier
    template <typename A> WL_HD int feeds_now(const A& a, int j, int LT) const {
        const int left = feeds_total(a, j, LT) - fed[j];
        if (j == 0) return left > 2 ? 2 : left;
        const int avail = emitted(j - 1, LT);
        const int Hs = a.g[j].Hs;
        const int e = a.base + 2 * fed[j];
        const bool ok0 = left > 0 && wl_ext1(e, Hs, a.ext) < avail && wl_ext1(e + 1, Hs, a.ext) < avail;
        const bool ok1 = left > 1 && wl_ext1(e + 2, Hs, a.ext) < avail && wl_ext1(e + 3, Hs, a.ext) < avail;
        const int n = ok0 ? (ok1 ? 2 : 1) : 0;
        return n;
    }
};

template <typename T, int LT, int PPR, int D = WL_ROWS_DEPTH>
struct WlAfbRows {
    typedef WlRowsArgs<T> Args;
    static const int kThreads = 64 * WL_ROWS_WAVES;
    static const int kMinWaves = 6;        // two workgroups per CU: 22 waves on 4 SIMDs
    static const int NL = 4 * PPR;         // DMA instructions per half-batch (4 rows x PPR pieces of 1 KiB)
    static const int WARM = (LT - 2) / 2;  // feeds that only fill the window
    static const int NSLOT = D + 1;        // half-batch slots of the input ring
    static_assert((D - 1) * NL < 64, "prefetch distance exceeds the vmcnt range");

    // ---- loader wave ------------------------------------------------------------------------------------------
    static WL_DEV void loader(const Args& a, const WlCtx& ctx, int64_t plane, int lane) {
        const WlRowsLevel& g = a.g[0];
        const char* xp = reinterpret_cast<const char*>(a.x + (size_t)plane * a.x_ps);
        const int row_bytes = g.Ws * (int)sizeof(T);
        const int nhb0 = (g.Kh + WARM + 1) / 2;                  // half-batches in which level 1 runs
        const int e_last = a.base + 2 * (g.Kh + WARM) - 1;       // last extended row level 1 consumes
        auto issue = [&](int h) {
            const int slot = h % NSLOT;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int e = a.base + 4 * h + r;
                e = e < e_last ? e : e_last;
                int src = wl_ext(e, g.Hs, a.ext);
                src = src < 0 ? 0 : src;                          // zero rows: a dummy row keeps the DMA count exact
                const char* grow = xp + (size_t)src * a.x_rs * sizeof(T);
                const unsigned dst = (unsigned)(g.ring_off + (slot * 4 + r) * g.ring_pitch);
#pragma unroll
                for (int p = 0; p < PPR; ++p) {
                    const int byte = (p * 64 + lane) * 16;
                    if (!(WL_ROWS_ABLATE & 2)) wl_dma16(ctx, dst + p * 1024, grow + byte, byte < row_bytes);
                }
            }
        };
        for (int h = 0; h < D; ++h) issue(h);
        unsigned long long tw = 0, tb = 0, ti = 0;
        for (int hb = 0; hb < a.nhb; ++hb) {
            const unsigned long long c0 = WL_TICK();
            if (hb < nhb0 && !(WL_ROWS_ABLATE & 2)) wl_wait_vm<(D - 1) * NL>();   // the rows of this half-batch have landed
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
    // FAST = 1: no lane of the wave touches the boundary extension and the sample origin 2k+base is even, so the L
    // samples of a row are one per-lane base address + constant offsets, read two elements at a time.
    // All arithmetic is plain scalar-tap FMAs (taps in SGPRs, four independent accumulator chains): v_pk_fma_f32 has
    // no throughput advantage on gfx950 and costs a register pair per broadcast sample.
    template <int FAST, int j>
    static WL_DEV void compute(const Args& a, const WlCtx& ctx, int64_t plane, int wave, int lane) {
        const WlRowsLevel& g = a.g[j];
        const int k = (wave - g.wave0) * 64 + lane;
        const bool active = k < g.Kw;
        const bool last = j == a.nlev - 1;
        char* const smem = ctx.smem;
        float twl[LT], twh[LT], thl[LT], thh[LT];
#pragma unroll
        for (int t = 0; t < LT; ++t) {
            twl[t] = wl_uniform_f(a.h_w_lo[t]); twh[t] = wl_uniform_f(a.h_w_hi[t]);
            thl[t] = wl_uniform_f(a.h_h_lo[t]); thh[t] = wl_uniform_f(a.h_h_hi[t]);
        }
        // where this lane's L row-filter samples live inside a ring row (bytes); zero padding -> the row's zero cell
        int off[FAST ? 1 : LT];
        if (FAST) {
            off[0] = (2 * (active ? k : 0) + a.base) * (int)sizeof(T);
        } else {
#pragma unroll
            for (int t = 0; t < LT; ++t) {
                const int c = active ? wl_ext(2 * k + a.base + t, g.Ws, a.ext) : 0;
                off[t] = (c < 0 ? g.Ws : c) * (int)sizeof(T);
            }
        }
        float wlo[LT], whi[LT];   // the window: row-filtered (lo, hi) of the last L extended rows of column k
#pragma unroll
        for (int t = 0; t < LT; ++t) wlo[t] = whi[t] = 0.f;
        const unsigned bplane = (unsigned)g.Kh * (unsigned)g.Kw;
        char* const hp = reinterpret_cast<char*>(a.yh[j] + (size_t)plane * 3 * bplane);
        char* const llp = last ? reinterpret_cast<char*>(a.ll + (size_t)plane * a.ll_ps) : nullptr;
        const int rmask = a.ring_rows - 1;
        const int zrow = a.zero_off;
        const int ring = g.ring_off, pitch = g.ring_pitch, Hs = g.Hs;
        const int nring = j + 1 < WL_ROWS_MAXLEV ? a.g[j + 1 < WL_ROWS_MAXLEV ? j + 1 : j].ring_off : 0;
        const int npitch = j + 1 < WL_ROWS_MAXLEV ? a.g[j + 1 < WL_ROWS_MAXLEV ? j + 1 : j].ring_pitch : 0;

        auto load_row = [&](int row_off, float (&v)[LT]) {   // row_off: wave-uniform LDS byte offset of the row
            if (FAST) {
                const char* p = smem + (row_off + off[0]);
                if (sizeof(T) == 4) {
#pragma unroll
                    for (int u = 0; u < LT / 2; ++u) {
                        const wl_f2 t2 = *reinterpret_cast<const wl_f2*>(p + 8 * u);
                        v[2 * u] = t2.x; v[2 * u + 1] = t2.y;
                    }
                } else {
                    typedef T Pair2 __attribute__((ext_vector_type(2)));
#pragma unroll
                    for (int u = 0; u < LT / 2; ++u) {
                        const Pair2 t2 = *reinterpret_cast<const Pair2*>(p + 2 * sizeof(T) * u);
                        v[2 * u] = (float)t2.x; v[2 * u + 1] = (float)t2.y;
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < LT; ++t) v[t] = (float)*reinterpret_cast<const T*>(smem + (row_off + off[t]));
            }
        };

        WlRowsSched sc;
        sc.init();
        unsigned long long tb = 0, tf = 0, ts = 0, c3 = WL_TICK();
        for (int hb = 0; hb < a.nhb; ++hb) {
            const unsigned long long c0 = WL_TICK();
            ts += c0 - c3;
            ctx.sync();
            const unsigned long long c1 = WL_TICK();
            tb += c1 - c0;
            const int n = wl_uniform(sc.feeds_now(a, j, LT));
            for (int i = 0; i < n; ++i) {
                const int f = sc.fed[j] + i;
                const int e = a.base + 2 * f;
                // the two source rows of this feed (wave-uniform LDS offsets)
                int r0, r1;
                if (j == 0) {
                    const int slot = ring + ((hb % NSLOT) * 4 + 2 * i) * pitch;
                    const bool zmode = a.ext == WL_EXT_ZERO;
                    r0 = (zmode && (unsigned)e >= (unsigned)Hs) ? zrow : slot;
                    r1 = (zmode && (unsigned)(e + 1) >= (unsigned)Hs) ? zrow : slot + pitch;
                } else {
                    const int s0 = wl_ext1(e, Hs, a.ext), s1 = wl_ext1(e + 1, Hs, a.ext);
                    r0 = s0 < 0 ? zrow : ring + (s0 & rmask) * pitch;
                    r1 = s1 < 0 ? zrow : ring + (s1 & rmask) * pitch;
                }
                r0 = wl_uniform(r0); r1 = wl_uniform(r1);
                if (active) {
                    float v0[LT], v1[LT];
                    load_row(r0, v0);
                    load_row(r1, v1);
                    float l0 = 0.f, h0 = 0.f, l1 = 0.f, h1 = 0.f;
#pragma unroll
                    for (int t = 0; t < LT; ++t) {
                        l0 = __builtin_fmaf(twl[t], v0[t], l0);
                        h0 = __builtin_fmaf(twh[t], v0[t], h0);
                        l1 = __builtin_fmaf(twl[t], v1[t], l1);
                        h1 = __builtin_fmaf(twh[t], v1[t], h1);
                    }
#pragma unroll
                    for (int t = 0; t < LT - 2; ++t) { wlo[t] = wlo[t + 2]; whi[t] = whi[t + 2]; }
                    wlo[LT - 2] = l0; whi[LT - 2] = h0;
                    wlo[LT - 1] = l1; whi[LT - 1] = h1;
                    if (f >= WARM) {
                        const int orow = f - WARM;
                        float ll = 0.f, lh = 0.f, hl = 0.f, hh = 0.f;
#pragma unroll
                        for (int t = 0; t < LT; ++t) {
                            ll = __builtin_fmaf(thl[t], wlo[t], ll);
                            lh = __builtin_fmaf(thh[t], wlo[t], lh);
                            hl = __builtin_fmaf(thl[t], whi[t], hl);
                            hh = __builtin_fmaf(thh[t], whi[t], hh);
                        }
                        const unsigned ob = ((unsigned)orow * (unsigned)g.Kw + (unsigned)k) * (unsigned)sizeof(T);
                        const unsigned bpb = bplane * (unsigned)sizeof(T);
                        const bool st = !(WL_ROWS_ABLATE & 1) || (lh + hl + hh + ll == 12345.f);   // product: always true
                        if (st) {
                        *reinterpret_cast<T*>(hp + ob) = (T)lh;                // W-lo / H-hi
                        *reinterpret_cast<T*>(hp + (bpb + ob)) = (T)hl;        // W-hi / H-lo
                        *reinterpret_cast<T*>(hp + (2 * bpb + ob)) = (T)hh;    // W-hi / H-hi
                        }
                        if (last) {
                            if (st)
                                *reinterpret_cast<T*>(llp + ((unsigned)orow * (unsigned)a.ll_rs + (unsigned)k) * (unsigned)sizeof(T)) = (T)ll;
                        } else {
                            *reinterpret_cast<T*>(smem + (nring + (orow & rmask) * npitch + k * (int)sizeof(T))) = (T)ll;
                        }
                    }
                }
            }
            c3 = WL_TICK();
            tf += c3 - c1;
            // advance the levels this wave depends on (same arithmetic in every wave: the schedule is shared)
            int nn[WL_ROWS_MAXLEV];
#pragma unroll
            for (int q = 0; q < WL_ROWS_MAXLEV; ++q) nn[q] = q < j ? sc.feeds_now(a, q, LT) : (q == j ? n : 0);
#pragma unroll
            for (int q = 0; q < WL_ROWS_MAXLEV; ++q) sc.fed[q] = wl_uniform(sc.fed[q] + nn[q]);
        }
        if ((WL_ROWS_ABLATE & 8) && j == 0 && k == 64) {   // second level-1 wave (an interior one)
            T* o = a.ll + (size_t)plane * a.ll_ps;
            o[0] = (T)(float)(tb >> 6); o[1] = (T)(float)(tf >> 6); o[2] = (T)(float)(ts >> 6);
        }
    }

    static WL_DEV void run(const Args& a, const WlCtx& ctx) {
        const int tid = ctx.tid;
        const int wave = wl_uniform(tid >> 6), lane = tid & 63;
        const int64_t plane = ctx.bid;
        // permanent zeros: the zero row and the zero cell (column Ws) of every ring row
        for (int i = tid * 4; i < a.zero_bytes; i += kThreads * 4) *reinterpret_cast<int*>(ctx.smem + a.zero_off + i) = 0;
        for (int j = 0; j < a.nlev; ++j) {
            const WlRowsLevel& g = a.g[j];
            const int rows = j == 0 ? 4 * NSLOT : a.ring_rows;
            for (int r = tid; r < rows; r += kThreads)
                *reinterpret_cast<T*>(ctx.smem + g.ring_off + r * g.ring_pitch + g.Ws * (int)sizeof(T)) = (T)0;
        }
        if (wave == a.loader_wave) {
            loader(a, ctx, plane, lane);
            return;
        }
        int lev = -1;
        for (int j = 0; j < a.nlev; ++j)
            if (wave >= a.g[j].wave0 && wave < a.g[j].wave0 + a.g[j].nwaves) lev = j;
        if (lev < 0) {   // spare wave: keeps the barrier count
            for (int hb = 0; hb < a.nhb; ++hb) ctx.sync();
            return;
        }
        // interior waves (no lane touches the boundary extension, even sample origin) read their samples as vectors
        const WlRowsLevel& g = a.g[lev];
        const int kmin = (wave - g.wave0) * 64;
        const int kmax = kmin + 63 < g.Kw - 1 ? kmin + 63 : g.Kw - 1;
        const bool fast = kmax >= kmin && !(a.base & 1) && 2 * kmin + a.base >= 0 && 2 * kmax + a.base + LT - 1 <= g.Ws - 1;
        if (fast) {
            if (lev == 0) compute<1, 0>(a, ctx, plane, wave, lane);
            else if (lev == 1) compute<1, 1>(a, ctx, plane, wave, lane);
            else compute<1, 2>(a, ctx, plane, wave, lane);
        } else {
            if (lev == 0) compute<0, 0>(a, ctx, plane, wave, lane);
            else if (lev == 1) compute<0, 1>(a, ctx, plane, wave, lane);
            else compute<0, 2>(a, ctx, plane, wave, lane);
        }
    }
};
