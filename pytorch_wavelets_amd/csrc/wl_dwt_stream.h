// Fused multi-level streaming DWT analysis kernel (the hot path of BASELINE configs[1]).
//
// One launch computes up to WL_MAXLEV decomposition levels.  A workgroup owns one horizontal
// STRIP of one (n,c) plane and marches down it; every input row is read from HBM exactly once
// (plus a small top halo per strip), the three detail bands of every level go straight to yh[j]
// and LL_1 .. LL_{J-1} are handed from level to level through LDS - they never touch HBM.
// HBM traffic = x in, yl + yh out: the algorithmic minimum of SURVEY.md 8(d).
//
// Level 1 (3/4 of all arithmetic and bytes) filters VERTICALLY FIRST, in registers:
//   * thread t owns CPT adjacent columns and keeps a sliding window of L input rows of them in VGPRs;
//     rows are fetched by EXTENDED row index (any boundary mode: row ext(e) is simply loaded again),
//     a whole batch (2*NB rows) ahead of use, so HBM latency is covered by the previous batch's work;
//   * per output row it produces the H-lowpass and H-highpass samples of its columns and writes just
//     those two rows to LDS (with their mirrored border columns);
//   * after one barrier the horizontal bank runs on the two rows: item = (row, band, q), the lane
//     reads V[4q .. 4q+L+1] as float4s (16-byte lane stride: conflict free), produces (lo,hi) for
//     k=2q,2q+1 and stores each band as 8 contiguous bytes per lane (512 B per wave).
//   There is no LDS ring at level 1: an input sample crosses LDS once instead of twice.
// Levels >= 2 (1/4 of the work, narrow rows) receive their source rows in natural order from the
// level above, filter along W into an LDS ring of (lo,hi) rows indexed by natural row, and filter the
// ring along H with the window rows taken from a small per-step offset table (extended rows resolved
// to source rows there; zero-extension rows point at a shared all-zero row).
// zero / symmetric / reflect for more than one fused level; any mode for a single level.
//
// Restates (fused, all levels): DWTForward.forward -> J x AFB2D.forward -> 2J x afb1d
// (reference dwt/transform2d.py:63-74, dwt/lowlevel.py:336-347, :91-172).  NB the reference filters
// along W then H; level 1 here filters along H then W - the same linear map, rounding differs by
// ~1e-7 relative.
#pragma once
#include "wl_common.h"

#define WL_MAXLEV 4

struct WlStreamLevel {
    int Hs, Ws;        // source rows / cols of this level
    int Kh, Kw;        // output rows / cols
    int nq;            // k-pairs per output row = (Kw+1)/2
    unsigned magic_q;  // floor(2^32/nq)+1 : flat item -> (row, q)
    int sr_pitch;      // floats per staged source row (multiple of 4)
    int sr_rows;       // staged rows capacity
    int in_pitch;      // floats per ring row (= 4*nq)
    int cap;           // ring rows
    int sr_off;        // LDS offsets in floats (multiples of 4)
    int in_off;
    int tab_off;       // window-row offset table: sr_rows(next) x LTP ints
};

template <typename T>
struct WlAfbStreamArgs {
    const T* x;
    T* yl;
    T* yh[WL_MAXLEV];
    const float* h_w_lo;
    const float* h_w_hi;
    const float* h_h_lo;
    const float* h_h_hi;
    int64_t NC;
    int nlev, S, ext, base, RS;
    int lds_floats;
    int taps_off;        // 4*LT floats
    int rows_off;        // 2*NBMAX ints: source rows of the batch being prefetched
    int zero_off;        // one all-zero ring row (zero padding at levels >= 2)
    int NB;              // level-1 output rows per batch (2*NB input rows are prefetched per thread)
    int col_vec;         // level-1 column groups can be loaded as one aligned vector
    int ablate;          // profiling only (WL_ABLATE): 1 = no band stores, 2 = no global loads, 4 = no horizontal bank
    WlStreamLevel g[WL_MAXLEV];
};

struct __attribute__((may_alias)) alignas(16) wl_f4 { float x, y, z, w; };
struct __attribute__((may_alias)) alignas(8) wl_f2 { float x, y; };
typedef float wl_v2 __attribute__((ext_vector_type(2)));   // (low-band, high-band) pair: one v_pk_fma_f32 per tap

WL_DEV unsigned wl_mulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * b) >> 32); }

// natural source rows needed to serve extended rows [eb, ee) of a length-n signal
WL_HD void wl_src_range(int eb, int ee, int n, int ext, int& rb, int& re) {
    rb = eb < 0 ? 0 : eb;
    re = ee > n ? n : ee;
    if (ext == WL_EXT_SYM) {
        if (eb < 0 && -eb > re) re = -eb;
        if (ee > n && 2 * n - ee < rb) rb = 2 * n - ee;
    } else if (ext == WL_EXT_REFL) {
        if (eb < 0 && 1 - eb > re) re = 1 - eb;
        if (ee > n && 2 * n - 1 - ee < rb) rb = 2 * n - 1 - ee;
    }
    if (rb < 0) rb = 0;
    if (re > n) re = n;
    if (re < rb) re = rb;
}

// Write sample `v` of source column `s` (row of width W staged at d[PADL + s]) to every border
// position it is the extension of.  Single reflection / single wrap only (W >= L).
WL_DEV void wl_mirror_cols(float* d, int s, float v, int W, int PADL, int pitch, int ext) {
    if (ext == WL_EXT_SYM) {
        if (s < PADL) d[PADL - 1 - s] = v;
        const int m = PADL + 2 * W - 1 - s;
        if (m < pitch) d[m] = v;
    } else if (ext == WL_EXT_REFL) {
        if (s >= 1 && s <= PADL) d[PADL - s] = v;
        const int m = PADL + 2 * W - 2 - s;
        if (s <= W - 2 && m < pitch) d[m] = v;
    } else if (ext == WL_EXT_PERIODIC || ext == WL_EXT_PER) {
        // periodization wraps with period Ne = W rounded up to even: the odd-length case repeats the last
        // sample at virtual position W (reference dwt/lowlevel.py:135-140)
        const int Ne = (ext == WL_EXT_PER) ? W + (W & 1) : W;
        if (s - Ne >= -PADL) d[PADL + s - Ne] = v;
        const int m = PADL + s + Ne;
        if (m < pitch) d[m] = v;
        if (Ne != W && s == W - 1) {
            if (PADL + W < pitch) d[PADL + W] = v;
            if (PADL >= 1) d[PADL - 1] = v;
        }
    }
}

// ML = number of levels this instantiation can fuse (1: the lean level-1-only kernel)
template <typename T, int LT, int NT, int CPT, int ML>
struct WlAfbStream {
    typedef WlAfbStreamArgs<T> Args;
    static const int kThreads = NT;
    static const int kMinWaves = 3;   // waves per SIMD the register allocation must allow
    static const int NV = (LT + 2 + 3) / 4;          // float4 reads per row-bank item
    static const int LTP = (LT + 3) / 4 * 4;         // table row length (ints)
    static const int NBMAX = 4;                      // level-1 output rows per batch (upper bound)
    typedef T ColVec __attribute__((ext_vector_type(CPT)));            // one aligned CPT-wide global load
    struct __attribute__((packed, aligned(sizeof(T)), may_alias)) Pair { T a, b; };   // two adjacent outputs, element-aligned
    struct __attribute__((may_alias)) alignas(16) wl_i4 { int x, y, z, w; };

    // taps as (lowpass, highpass) pairs so that both bands of a sample come out of ONE packed FMA with the
    // sample broadcast (no register shuffles).  They are re-read from LDS at the start of every phase: short
    // live ranges keep the register allocation of the level-1 loop free of spills.
    struct TapsW { wl_v2 t[LT]; };   // along W (row / horizontal banks)
    struct TapsH { wl_v2 t[LT]; };   // along H (column / vertical banks)
    static WL_DEV void load_taps_w(const float* tl, TapsW& t) {
#pragma unroll
        for (int j = 0; j < LT; ++j) { t.t[j].x = tl[2 * j]; t.t[j].y = tl[2 * j + 1]; }
    }
    static WL_DEV void load_taps_h(const float* tl, TapsH& t) {
#pragma unroll
        for (int j = 0; j < LT; ++j) { t.t[j].x = tl[2 * LT + 2 * j]; t.t[j].y = tl[2 * LT + 2 * j + 1]; }
    }

    struct State {
        int kb[WL_MAXLEV], ke[WL_MAXLEV];   // output rows computed by this strip, per level
        int ob[WL_MAXLEV], oe[WL_MAXLEV];   // output rows OWNED (written) by this strip
        int eb[WL_MAXLEV], ee[WL_MAXLEV];   // extended source rows needed, per level
        int next[WL_MAXLEV];                // next output row to produce
    };

    // ---- row bank: `nrows` staged source rows -> ring rows slot0, slot0+1, ... (mod cap) -----------------
    static WL_DEV void row_bank(const float* tl, const WlStreamLevel& g, const float* SR, float* IN, int nrows,
                                int slot0, int tid) {
        TapsW tp;
        load_taps_w(tl, tp);
        const int total = nrows * g.nq;
        _Pragma("nounroll") for (int f = tid; f < total; f += NT) {
            const int i = (int)wl_mulhi((unsigned)f, g.magic_q);
            const int q = f - i * g.nq;
            int slot = slot0 + i;
            if (slot >= g.cap) slot -= g.cap;
            float v[NV * 4];
            const wl_f4* s4 = reinterpret_cast<const wl_f4*>(SR + i * g.sr_pitch) + q;
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const wl_f4 t = s4[u];
                v[4 * u] = t.x; v[4 * u + 1] = t.y; v[4 * u + 2] = t.z; v[4 * u + 3] = t.w;
            }
            wl_v2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};   // (lo,hi) of k=2q and of k=2q+1
#pragma unroll
            for (int j = 0; j < LT; ++j) {
                a0 += tp.t[j] * v[j];
                a1 += tp.t[j] * v[j + 2];
            }
            wl_f4 o;
            o.x = a0.x; o.y = a0.y; o.z = a1.x; o.w = a1.y;
            reinterpret_cast<wl_f4*>(IN + slot * g.in_pitch)[q] = o;
        }
    }

    // ---- window table: for output rows k0..k0+nk-1 of level J, the LDS float offset of each tap's ring row ---
    // level 1 (J==0): the ring is indexed by extended row, slot(e) = (e - eb) mod cap, `slotA` = slot of the
    // first window row of k0.  levels >= 2: natural-row ring, each extended row is mapped to its source row;
    // rows of the zero extension point at the shared all-zero row.
    template <int J>
    static WL_DEV void fill_table(const Args& a, float* lds, int k0, int nk, int slotA, int tid) {
        const WlStreamLevel& g = a.g[J];
        int* tab = reinterpret_cast<int*>(lds + g.tab_off);
        _Pragma("nounroll") for (int t = tid; t < nk * LTP; t += NT) {
            const int kk = t / LTP, j = t - kk * LTP;
            int off = a.zero_off;
            if (j < LT) {
                if (J == 0) {
                    int slot = slotA + 2 * kk + j;
                    while (slot >= g.cap) slot -= g.cap;
                    off = g.in_off + slot * g.in_pitch;
                } else {
                    const int r = wl_ext(2 * (k0 + kk) + a.base + j, g.Hs, a.ext);
                    if (r >= 0) off = g.in_off + (r % g.cap) * g.in_pitch;
                }
            }
            tab[t] = off;
        }
    }

    // ---- column bank for `nk` output rows of level J starting at k0 (window rows from the table) ----------
    template <int J>
    static WL_DEV void col_bank(const Args& a, const State& st, float* lds, int64_t plane,
                                int k0, int nk, int tid) {
        TapsH tp;
        load_taps_h(lds + a.taps_off, tp);
        const WlStreamLevel& g = a.g[J];
        const int* tab = reinterpret_cast<const int*>(lds + g.tab_off);
        const bool last = (J + 1 == a.nlev);
        const unsigned bplane = (unsigned)g.Kh * (unsigned)g.Kw;
        T* hp = a.yh[J] + (size_t)plane * 3 * bplane;
        T* lp = last ? a.yl + (size_t)plane * bplane : nullptr;
        const WlStreamLevel& gn = a.g[J + 1 < ML ? J + 1 : J];
        float* NSR = lds + gn.sr_off;
        const int PADL = -a.base;
        const int total = nk * g.nq;
        _Pragma("nounroll") for (int f = tid; f < total; f += NT) {
            const int kk = (int)wl_mulhi((unsigned)f, g.magic_q);
            const int q = f - kk * g.nq;
            const int k = k0 + kk;
            int offs[LTP];
#pragma unroll
            for (int u = 0; u < LTP / 4; ++u) {
                const wl_i4 t = reinterpret_cast<const wl_i4*>(tab + kk * LTP)[u];
                offs[4 * u] = t.x; offs[4 * u + 1] = t.y; offs[4 * u + 2] = t.z; offs[4 * u + 3] = t.w;
            }
            wl_v2 cl0 = {0.f, 0.f}, ch0 = {0.f, 0.f}, cl1 = {0.f, 0.f}, ch1 = {0.f, 0.f};   // (H-lo,H-hi) of W-lo / W-hi
            const float* colp = lds + 4 * q;
#pragma unroll
            for (int j = 0; j < LT; ++j) {
                const wl_f4 p = *reinterpret_cast<const wl_f4*>(colp + offs[j]);
                cl0 += tp.t[j] * p.x; ch0 += tp.t[j] * p.y;
                cl1 += tp.t[j] * p.z; ch1 += tp.t[j] * p.w;
            }
            const float ll0 = cl0.x, lh0 = cl0.y, hl0 = ch0.x, hh0 = ch0.y;
            const float ll1 = cl1.x, lh1 = cl1.y, hl1 = ch1.x, hh1 = ch1.y;
            const int kw = 2 * q;
            const bool two = kw + 1 < g.Kw;
            if (k >= st.ob[J] && k < st.oe[J]) {
                const unsigned o = (unsigned)k * (unsigned)g.Kw + (unsigned)kw;
                hp[o] = (T)lh0;
                hp[bplane + o] = (T)hl0;
                hp[2 * bplane + o] = (T)hh0;
                if (last) lp[o] = (T)ll0;
                if (two) {
                    hp[o + 1] = (T)lh1;
                    hp[bplane + o + 1] = (T)hl1;
                    hp[2 * bplane + o + 1] = (T)hh1;
                    if (last) lp[o + 1] = (T)ll1;
                }
            }
            if (!last) {
                // hand the LL samples to the next level's staging row, with their mirrored border copies
                float* nrow = NSR + kk * gn.sr_pitch;
                nrow[PADL + kw] = ll0;
                if (two) nrow[PADL + kw + 1] = ll1;
                if (a.ext != WL_EXT_ZERO && (kw < PADL + 2 || kw + 2 + (gn.sr_pitch - PADL - g.Kw) >= g.Kw)) {
                    wl_mirror_cols(nrow, kw, ll0, g.Kw, PADL, gn.sr_pitch, a.ext);
                    if (two) wl_mirror_cols(nrow, kw + 1, ll1, g.Kw, PADL, gn.sr_pitch, a.ext);
                }
            }
        }
    }

    // largest natural source row the window of output k touches (levels >= 2)
    static WL_DEV int need_max(int k, int base, int n, int ext) {
        const int a0 = 2 * k + base, b0 = a0 + LT - 1;
        int m = b0;
        if (a0 < 0) {
            if (ext == WL_EXT_SYM && -a0 - 1 > m) m = -a0 - 1;
            if (ext == WL_EXT_REFL && -a0 > m) m = -a0;
        }
        return m > n - 1 ? n - 1 : m;
    }

    // ---- one step of level J >= 1 (0-based J): `nr` new source rows r0.. sit in its staging rows ------
    template <int J>
    static WL_DEV void level_step(const Args& a, State& st, float* lds, int64_t plane,
                                  int r0, int nr, const WlCtx& ctx) {
        const WlStreamLevel& g = a.g[J];
        const int tid = ctx.tid;
        const int arrived = r0 + nr;
        const bool deeper = (J + 1 < ML) && (J + 1 < a.nlev);
        const int maxnk = deeper ? a.g[J + 1 < ML ? J + 1 : J].sr_rows : g.sr_rows + LT;
        row_bank(lds + a.taps_off, g, lds + g.sr_off, lds + g.in_off, nr, r0 % g.cap, tid);
        for (;;) {
            const int k0 = st.next[J];
            int k = k0;
            while (k < st.ke[J] && k - k0 < maxnk && need_max(k, a.base, g.Hs, a.ext) < arrived) ++k;
            const int nk = k - k0;
            fill_table<J>(a, lds, k0, nk, 0, tid);
            ctx.sync();
            if (nk == 0) break;
            col_bank<J>(a, st, lds, plane, k0, nk, tid);
            st.next[J] = k;
            ctx.sync();
            if (deeper) level_step<(J + 1 < ML ? J + 1 : J)>(a, st, lds, plane, k0, nk, ctx);
        }
    }

    // ---- level-1 vertical bank: nb output rows from the register window + the prefetched rows -> V rows ----
    // window rows live in xw[(wb + j) % LT]; ROT: the batch advances the window by a multiple of LT rows, so the
    // rotation is a compile-time renaming (no moves); otherwise the window is shifted down two rows per output.
    template <bool ROT>
    static WL_DEV void vertical_bank(const Args& a, const float* tl, float (&xw)[LT][CPT],
                                     float (&pfr)[2 * NBMAX][CPT], float* V, int vpitch, int nb, int c0,
                                     bool has_cols, int W, int PADL) {
            TapsH tp;
            load_taps_h(tl, tp);
#pragma unroll
            for (int i = 0; i < NBMAX; ++i) {
                if (i < nb) {
                    if (!ROT) {
#pragma unroll
                        for (int j = 0; j < LT - 2; ++j)
#pragma unroll
                            for (int u = 0; u < CPT; ++u) xw[j][u] = xw[j + 2][u];
                    }
                    const int wb = ROT ? (2 * i + 2) % LT : 0;   // slot of the window's first row
#pragma unroll
                    for (int u = 0; u < CPT; ++u) {
                        xw[(wb + LT - 2) % LT][u] = pfr[2 * i][u];
                        xw[(wb + LT - 1) % LT][u] = pfr[2 * i + 1][u];
                    }
                    wl_v2 acc[CPT];
#pragma unroll
                    for (int u = 0; u < CPT; ++u) acc[u] = wl_v2{0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < LT; ++j)
#pragma unroll
                        for (int u = 0; u < CPT; ++u) acc[u] += tp.t[j] * xw[(wb + j) % LT][u];
                    float vl[CPT], vh[CPT];
#pragma unroll
                    for (int u = 0; u < CPT; ++u) { vl[u] = acc[u].x; vh[u] = acc[u].y; }
                    if (has_cols) {
                        float* r0 = V + i * vpitch;
                        float* r1 = V + (a.NB + i) * vpitch;
                        if (((PADL | CPT) & 1) == 0 && c0 + CPT <= W) {
#pragma unroll
                            for (int u = 0; u < CPT; u += 2) {
                                wl_f2 t0, t1;
                                t0.x = vl[u]; t0.y = vl[u + 1]; t1.x = vh[u]; t1.y = vh[u + 1];
                                *reinterpret_cast<wl_f2*>(r0 + PADL + c0 + u) = t0;
                                *reinterpret_cast<wl_f2*>(r1 + PADL + c0 + u) = t1;
                            }
                        } else {
#pragma unroll
                            for (int u = 0; u < CPT; ++u)
                                if (c0 + u < W) { r0[PADL + c0 + u] = vl[u]; r1[PADL + c0 + u] = vh[u]; }
                        }
                    }
                }
            }
    }

    static WL_DEV int floordiv2(int v) { return v >= 0 ? v / 2 : -((-v + 1) / 2); }

    static WL_DEV void run(const Args& a, const WlCtx& ctx) {
        const int tid = ctx.tid;
        const int64_t plane = ctx.bid / a.S;
        const int strip = (int)(ctx.bid - plane * a.S);
        float* lds = reinterpret_cast<float*>(ctx.smem);

        // ---- strip geometry: owned / computed output rows and needed extended rows per level ---------
        State st;
        const int Jl = a.nlev - 1;
#pragma unroll
        for (int j = WL_MAXLEV - 1; j >= 0; --j) {   // owned rows: split the last level evenly, map upwards
            st.ob[j] = st.oe[j] = 0;
            if (j == Jl) {
                st.ob[j] = (int)(((int64_t)a.g[j].Kh * strip) / a.S);
                st.oe[j] = (int)(((int64_t)a.g[j].Kh * (strip + 1)) / a.S);
            } else if (j < Jl) {
                const int n = a.g[j].Kh;
                const int b = strip == 0 ? 0 : 2 * st.ob[j + 1] + a.base + LT - 2;
                const int e = strip == a.S - 1 ? n : 2 * st.oe[j + 1] + a.base + LT - 2;
                st.ob[j] = b < 0 ? 0 : (b > n ? n : b);
                st.oe[j] = e < 0 ? 0 : (e > n ? n : e);
            }
        }
#pragma unroll
        for (int j = WL_MAXLEV - 1; j >= 0; --j) {   // computed rows / needed extended rows
            st.kb[j] = st.ke[j] = st.eb[j] = st.ee[j] = st.next[j] = 0;
            if (j > Jl) continue;
            if (j == Jl) {
                st.kb[j] = st.ob[j];
                st.ke[j] = st.oe[j];
            } else {
                wl_src_range(st.eb[j + 1], st.ee[j + 1], a.g[j + 1].Hs, a.ext, st.kb[j], st.ke[j]);
                if (st.ob[j] < st.kb[j]) st.kb[j] = st.ob[j];   // the strip also computes every row it owns
                if (st.oe[j] > st.ke[j]) st.ke[j] = st.oe[j];
            }
            st.eb[j] = 2 * st.kb[j] + a.base;
            st.ee[j] = 2 * (st.ke[j] - 1) + a.base + LT;
            st.next[j] = st.kb[j];
        }
        if (st.ke[0] <= st.kb[0]) return;   // empty strip (more strips than rows)

        // ---- LDS init: taps, the zero row; zero padding also keeps every staging border at zero ------
        {
            float* tl = lds + a.taps_off;
            for (int i = tid; i < LT; i += NT) {
                tl[2 * i] = a.h_w_lo[i]; tl[2 * i + 1] = a.h_w_hi[i];
                tl[2 * LT + 2 * i] = a.h_h_lo[i]; tl[2 * LT + 2 * i + 1] = a.h_h_hi[i];
            }
            const int zb = a.ext == WL_EXT_ZERO ? a.g[0].sr_off : a.zero_off;
            const int ze = a.ext == WL_EXT_ZERO ? a.lds_floats : a.zero_off + a.g[0].in_pitch;
            for (int i = zb + tid; i < ze; i += NT) lds[i] = 0.f;
        }
        ctx.sync();
        const float* tl = lds + a.taps_off;

        const WlStreamLevel& g0 = a.g[0];
        const int H = g0.Hs, W = g0.Ws;
        const int PADL = -a.base;
        const T* xp = a.x + (size_t)plane * H * W;
        float* V = lds + g0.sr_off;                  // V[band][row][pitch], band 0 = H-lowpass, 1 = H-highpass
        const int vpitch = g0.sr_pitch;
        const int c0 = tid * CPT;                    // this thread's first column
        const bool has_cols = c0 < W;
        const bool deeper = ML > 1 && a.nlev > 1;
        const WlStreamLevel& g1 = a.g[ML > 1 ? 1 : 0];
        const unsigned bplane = (unsigned)g0.Kh * (unsigned)g0.Kw;
        T* hp = a.yh[0] + (size_t)plane * 3 * bplane;
        T* lp = deeper ? nullptr : a.yl + (size_t)plane * bplane;

        // source rows of the next 2*NB extended rows, resolved once per batch by 2*NB lanes (LDS table) so that
        // the unrolled row loads below carry no boundary-extension code
        int* rowsrc = reinterpret_cast<int*>(lds + a.rows_off);
        auto fill_rows = [&](int efirst) {
            if (tid < 2 * NBMAX) rowsrc[tid] = wl_ext(efirst + tid, H, a.ext);
        };
        // one row of this thread's columns -> registers (zeros for rows of the zero extension)
        auto load_row = [&](int r, float (&dst)[CPT]) {
#pragma unroll
            for (int u = 0; u < CPT; ++u) dst[u] = 0.f;
            if (r < 0 || !has_cols || (a.ablate & 2)) return;
            const T* src = xp + (unsigned)(r * W + c0);
            if (a.col_vec) {
                const ColVec cv = *reinterpret_cast<const ColVec*>(src);
#pragma unroll
                for (int u = 0; u < CPT; ++u) dst[u] = (float)cv[u];
            } else {
#pragma unroll
                for (int u = 0; u < CPT; ++u) if (c0 + u < W) dst[u] = (float)src[u];
            }
        };

        float xw[LT][CPT];          // sliding window: rows e .. e+LT-1 of the current output
        float pfr[2 * NBMAX][CPT];  // the next batch's 2*NB rows, in flight while this batch is computed
        const int eb = st.eb[0];
        // prologue: first LT-2 window rows, then the first batch
        for (int j0 = 0; j0 < LT - 2; j0 += 2 * NBMAX) {
            fill_rows(eb + j0);
            ctx.sync();
#pragma unroll
            for (int j = 0; j < LT - 2; ++j)
                if (j >= j0 && j < j0 + 2 * NBMAX) load_row(rowsrc[j - j0], xw[j + 2]);
            ctx.sync();
        }
        int enext = eb + LT - 2;    // next extended row to fetch
        fill_rows(enext);
        ctx.sync();
#pragma unroll
        for (int i = 0; i < 2 * NBMAX; ++i)
            if (i < 2 * a.NB) load_row(rowsrc[i], pfr[i]);
        ctx.sync();
        fill_rows(enext + 2 * a.NB);   // rows of the batch after the first one
        ctx.sync();

        for (int k0 = st.kb[0]; k0 < st.ke[0]; k0 += a.NB) {
            const int nb = (st.ke[0] - k0) < a.NB ? (st.ke[0] - k0) : a.NB;
            // ---- vertical bank (registers) -> V rows ------------------------------------------------------
            if ((2 * NBMAX) % LT == 0 && a.NB == NBMAX) vertical_bank<true>(a, tl, xw, pfr, V, vpitch, nb, c0, has_cols, W, PADL);
            else vertical_bank<false>(a, tl, xw, pfr, V, vpitch, nb, c0, has_cols, W, PADL);
            // mirrored border columns of the rows just written (own LDS writes: program order suffices)
            if (has_cols && a.ext != WL_EXT_ZERO && (c0 < vpitch - W + 1 || c0 + CPT + (vpitch - W) >= W)) {
                _Pragma("nounroll") for (int i = 0; i < 2 * a.NB; ++i) {
                    float* rr = V + i * vpitch;
                    if ((i < a.NB ? i : i - a.NB) < nb) {
                        _Pragma("nounroll") for (int u = 0; u < CPT; ++u)
                            if (c0 + u < W) wl_mirror_cols(rr, c0 + u, rr[PADL + c0 + u], W, PADL, vpitch, a.ext);
                    }
                }
            }
            enext += 2 * nb;
            // ---- prefetch the next batch's rows (in flight during the horizontal bank and the deeper levels)
            if (k0 + nb < st.ke[0]) {
#pragma unroll
                for (int i = 0; i < 2 * NBMAX; ++i)
                    if (i < 2 * a.NB) load_row(rowsrc[i], pfr[i]);
            }
            ctx.sync();
            fill_rows(enext + 2 * a.NB);   // for the prefetch of the NEXT iteration (published by the barrier below)
            // ---- horizontal bank on the V rows: bands -> HBM, LL -> next level's staging (or yl) ------------
            {
                TapsW tp;
                load_taps_w(tl, tp);
                const int total = (a.ablate & 4) ? 0 : 2 * nb * g0.nq;
                _Pragma("nounroll") for (int f = tid; f < total; f += NT) {
                    const int gq = (int)wl_mulhi((unsigned)f, g0.magic_q);   // = band*nb + row
                    const int q = f - gq * g0.nq;
                    const int band = gq >= nb ? 1 : 0;
                    const int i = gq - band * nb;
                    float v[NV * 4];
                    const wl_f4* s4 = reinterpret_cast<const wl_f4*>(V + (band * a.NB + i) * vpitch) + q;
#pragma unroll
                    for (int u = 0; u < NV; ++u) {
                        const wl_f4 t = s4[u];
                        v[4 * u] = t.x; v[4 * u + 1] = t.y; v[4 * u + 2] = t.z; v[4 * u + 3] = t.w;
                    }
                    wl_v2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < LT; ++j) {
                        a0 += tp.t[j] * v[j];
                        a1 += tp.t[j] * v[j + 2];
                    }
                    const float lo0 = a0.x, hi0 = a0.y, lo1 = a1.x, hi1 = a1.y;
                    const int k = k0 + i;
                    const int kw = 2 * q;
                    const bool two = kw + 1 < g0.Kw;
                    const bool own = k >= st.ob[0] && k < st.oe[0];
                    const unsigned o = (unsigned)k * (unsigned)g0.Kw + (unsigned)kw;
                    // band 0 (H-low):  lo -> LL, hi -> highs[1] (W-hi/H-lo);  band 1 (H-high): lo -> highs[0], hi -> highs[2]
                    if (own && !((a.ablate & 1) && lo0 != 12345.f)) {
                        // each band row gets 8 contiguous bytes per lane (one dwordx2 store; rows are only
                        // element-aligned because Kw is odd in general)
                        T* dhi = hp + (band ? 2u : 1u) * bplane + o;
                        T* dlo = band ? hp + o : (deeper ? nullptr : lp + o);
                        if (two) {
                            Pair ph; ph.a = (T)hi0; ph.b = (T)hi1;
                            *reinterpret_cast<Pair*>(dhi) = ph;
                            if (dlo) { Pair pl; pl.a = (T)lo0; pl.b = (T)lo1; *reinterpret_cast<Pair*>(dlo) = pl; }
                        } else {
                            dhi[0] = (T)hi0;
                            if (dlo) dlo[0] = (T)lo0;
                        }
                    }
                    if (deeper && band == 0) {
                        float* nrow = lds + g1.sr_off + i * g1.sr_pitch;
                        nrow[PADL + kw] = lo0;
                        if (two) nrow[PADL + kw + 1] = lo1;
                        if (a.ext != WL_EXT_ZERO && (kw < PADL + 2 || kw + 2 + (g1.sr_pitch - PADL - g0.Kw) >= g0.Kw)) {
                            wl_mirror_cols(nrow, kw, lo0, g0.Kw, PADL, g1.sr_pitch, a.ext);
                            if (two) wl_mirror_cols(nrow, kw + 1, lo1, g0.Kw, PADL, g1.sr_pitch, a.ext);
                        }
                    }
                }
            }
            ctx.sync();
            if (ML > 1) { if (deeper) level_step<(ML > 1 ? 1 : 0)>(a, st, lds, plane, k0, nb, ctx); }
        }
    }
};
