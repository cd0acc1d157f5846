// Fused multi-level streaming DWT analysis kernel (the hot path of BASELINE configs[1]).
//
// One launch computes up to WL_MAXLEV decomposition levels.  A workgroup owns one horizontal
// STRIP of one (n,c) plane and marches down it: every input row is read from HBM exactly once
// (plus a small top halo per strip), filtered along W into an LDS ring of (lo,hi) rows, the ring is
// filtered along H, the three detail bands go straight to yh[j] and the LL row is handed to the
// NEXT level through LDS - LL_1 .. LL_{J-1} never touch HBM.  HBM traffic = x in, yl + yh out:
// the algorithmic minimum of SURVEY.md 8(d).
//
//   level 1 source rows are fetched by EXTENDED index (any boundary mode: the row ext(e) is simply
//   loaded again), levels >= 2 receive their source rows in natural order from the level above, keep
//   them in a ring indexed by natural row and resolve each tap's extended row to its source row
//   with scalar index math; mirrored COLUMNS are materialised when a sample is handed down
//   (zero / symmetric / reflect; periodic modes are only offered for a single fused level).
//
// Thread mapping (256 threads, wave64):
//   row bank   : item = (row, q): lane reads SR[4q .. 4q+L+1] as float4s (16-byte lane stride, no
//                bank conflicts), produces (lo,hi) for k=2q,2q+1, writes one float4 to the ring.
//   column bank: item = (out row, kw): L ds_read_b64 of (lo,hi) with consecutive lanes on consecutive
//                8-byte slots, 4 FMAs per tap, taps in SGPRs; 128-byte coalesced band stores.
//   items are dealt to lanes as a flat index modulo 256 so odd widths (259, 133, 70) cost a partial
//   pass instead of a whole one; the row index stays wave-uniform so ring-slot arithmetic is scalar.
//
// Restates (fused, all levels): DWTForward.forward -> J x AFB2D.forward -> 2J x afb1d
// (reference dwt/transform2d.py:63-74, dwt/lowlevel.py:336-347, :91-172).
#pragma once
#include "wl_common.h"

#define WL_MAXLEV 4
#define WL_NONE (-0x40000000)

struct WlStreamLevel {
    int Hs, Ws;     // source rows / cols of this level
    int Kh, Kw;     // output rows / cols
    int nq;         // k-pairs per output row = (Kw+1)/2
    int sr_pitch;   // floats per staged source row (multiple of 4)
    int sr_rows;    // staged rows capacity
    int in_pitch;   // floats per ring row (>= 2*Kw, multiple of 4)
    int cap;        // ring rows
    int sr_off;     // LDS offsets in floats (multiples of 4)
    int in_off;
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
    int vec_ok;          // level-1 rows are 16-byte aligned and W % (16/sizeof(T)) == 0
    unsigned magic_w;    // floor(2^32 / Wv) + 1, Wv = vectors per level-1 row
    int lds_floats;
    WlStreamLevel g[WL_MAXLEV];
};

struct alignas(16) wl_f4 { float x, y, z, w; };
struct alignas(8) wl_f2 { float x, y; };

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

template <typename T, int LT>
struct WlAfbStream {
    typedef WlAfbStreamArgs<T> Args;
    static const int kThreads = 256;
    static const int NV = (LT + 2 + 3) / 4;          // float4 reads per row-bank item
    static const int VEC = 16 / (int)sizeof(T);      // elements per 16-byte global load
    static const int PMAX = 8;                       // prefetch registers: PMAX 16-byte loads/thread
    struct alignas(16) Vec { T v[VEC]; };

    struct Taps { float wl[LT], wh[LT], hl[LT], hh[LT]; };

    struct State {
        int kb[WL_MAXLEV], ke[WL_MAXLEV];   // output rows computed by this strip, per level
        int ob[WL_MAXLEV], oe[WL_MAXLEV];   // output rows OWNED (written) by this strip
        int eb[WL_MAXLEV], ee[WL_MAXLEV];   // extended source rows needed, per level
        int next[WL_MAXLEV];                // next output row to produce
    };

    // ---- row bank: one staged source row -> (lo,hi) ring row(s) ------------------------------------
    static WL_DEV void row_bank(const Taps& tp, const float* srow, float* d0, int nq, int rot, int tid) {
        for (int q = (tid - rot) & (kThreads - 1); q < nq; q += kThreads) {
            float v[NV * 4];
            const wl_f4* s4 = reinterpret_cast<const wl_f4*>(srow) + q;
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const wl_f4 t = s4[u];
                v[4 * u] = t.x; v[4 * u + 1] = t.y; v[4 * u + 2] = t.z; v[4 * u + 3] = t.w;
            }
            float lo0 = 0.f, hi0 = 0.f, lo1 = 0.f, hi1 = 0.f;
#pragma unroll
            for (int j = 0; j < LT; ++j) {
                lo0 += tp.wl[j] * v[j];
                hi0 += tp.wh[j] * v[j];
                lo1 += tp.wl[j] * v[j + 2];
                hi1 += tp.wh[j] * v[j + 2];
            }
            wl_f4 o;
            o.x = lo0; o.y = hi0; o.z = lo1; o.w = hi1;
            reinterpret_cast<wl_f4*>(d0)[q] = o;
        }
    }

    // ---- column bank for `nk` output rows of level J starting at k0 --------------------------------
    template <int J>
    static WL_DEV void col_bank(const Args& a, const Taps& tp, const State& st, float* lds, int64_t plane,
                                int k0, int nk, int tid) {
        const WlStreamLevel& g = a.g[J];
        const float* IN = lds + g.in_off;
        const bool last = (J + 1 == a.nlev);
        const size_t bplane = (size_t)g.Kh * g.Kw;
        T* hp = a.yh[J] + (size_t)plane * 3 * bplane;
        T* lp = a.yl + (size_t)plane * bplane;   // only used when `last`
        const int PADL = -a.base;
        for (int kk = 0; kk < nk; ++kk) {
            const int k = k0 + kk;
            const int a0 = 2 * k + a.base;
            int rowoff[LT];
            bool use[LT];
            if (J == 0) {
                // level 1: the ring is indexed by EXTENDED row (rows were fetched as ext(e), zeros staged)
                int slot = wl_pmod(a0 - st.eb[0], g.cap);
#pragma unroll
                for (int j = 0; j < LT; ++j) {
                    rowoff[j] = slot * g.in_pitch;
                    use[j] = true;
                    slot = (slot + 1 == g.cap) ? 0 : slot + 1;
                }
            } else {
                // levels >= 2: the ring holds NATURAL rows; each tap's extended row is mapped to its source
                // row here (scalar arithmetic), rows of the zero extension are skipped
#pragma unroll
                for (int j = 0; j < LT; ++j) {
                    const int r = wl_ext(a0 + j, g.Hs, a.ext);
                    use[j] = r >= 0;
                    rowoff[j] = (r < 0 ? 0 : r % g.cap) * g.in_pitch;
                }
            }
            const bool own = (k >= st.ob[J] && k < st.oe[J]);
            float* nrow = nullptr;
            int nW = 0, npitch = 0;
            if (!last) {
                const WlStreamLevel& gn = a.g[J + 1 < WL_MAXLEV ? J + 1 : J];
                nrow = lds + gn.sr_off + kk * gn.sr_pitch;
                nW = g.Kw;
                npitch = gn.sr_pitch;
            }
            for (int kw = (tid - kk * g.Kw) & (kThreads - 1); kw < g.Kw; kw += kThreads) {
                float ll = 0.f, lh = 0.f, hl = 0.f, hh = 0.f;
#pragma unroll
                for (int j = 0; j < LT; ++j) {
                    if (use[j]) {
                        const wl_f2 p = *reinterpret_cast<const wl_f2*>(IN + rowoff[j] + 2 * kw);
                        ll += tp.hl[j] * p.x;
                        lh += tp.hh[j] * p.x;
                        hl += tp.hl[j] * p.y;
                        hh += tp.hh[j] * p.y;
                    }
                }
                if (own) {
                    const size_t o = (size_t)k * g.Kw + kw;
                    hp[o] = (T)lh;
                    hp[bplane + o] = (T)hl;
                    hp[2 * bplane + o] = (T)hh;
                    if (last) lp[o] = (T)ll;
                }
                if (!last) {
                    // hand the LL sample to the next level's staging row, with its mirrored copies
                    nrow[PADL + kw] = ll;
                    if (a.ext == WL_EXT_SYM) {
                        if (kw < PADL) nrow[PADL - 1 - kw] = ll;
                        const int m = PADL + 2 * nW - 1 - kw;
                        if (m < npitch) nrow[m] = ll;
                    } else if (a.ext == WL_EXT_REFL) {
                        if (kw >= 1 && kw <= PADL) nrow[PADL - kw] = ll;
                        const int m = PADL + 2 * nW - 2 - kw;
                        if (kw <= nW - 2 && m < npitch) nrow[m] = ll;
                    }
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
    static WL_DEV void level_step(const Args& a, const Taps& tp, State& st, float* lds, int64_t plane,
                                  int r0, int nr, const WlCtx& ctx) {
        const WlStreamLevel& g = a.g[J];
        float* SR = lds + g.sr_off;
        float* IN = lds + g.in_off;
        const int tid = ctx.tid;
        for (int i = 0; i < nr; ++i) {
            const int r = r0 + i;
            row_bank(tp, SR + i * g.sr_pitch, IN + (r % g.cap) * g.in_pitch, g.nq, i * g.nq, tid);
        }
        ctx.sync();
        const int arrived = r0 + nr;
        const bool deeper = (J + 1 < WL_MAXLEV) && (J + 1 < a.nlev);
        const int maxnk = deeper ? a.g[J + 1 < WL_MAXLEV ? J + 1 : J].sr_rows : 0x7fffffff;
        for (;;) {
            const int k0 = st.next[J];
            int k = k0;
            while (k < st.ke[J] && k - k0 < maxnk && need_max(k, a.base, g.Hs, a.ext) < arrived) ++k;
            const int nk = k - k0;
            if (nk == 0) break;
            col_bank<J>(a, tp, st, lds, plane, k0, nk, tid);
            st.next[J] = k;
            ctx.sync();
            if (deeper) level_step<(J + 1 < WL_MAXLEV ? J + 1 : J)>(a, tp, st, lds, plane, k0, nk, ctx);
        }
    }

    static WL_DEV void run(const Args& a, const WlCtx& ctx) {
        const int tid = ctx.tid;
        const int64_t plane = ctx.bid / a.S;
        const int strip = (int)(ctx.bid - plane * a.S);
        float* lds = reinterpret_cast<float*>(ctx.smem);

        Taps tp;
#pragma unroll
        for (int j = 0; j < LT; ++j) {
            tp.wl[j] = a.h_w_lo[j]; tp.wh[j] = a.h_w_hi[j];
            tp.hl[j] = a.h_h_lo[j]; tp.hh[j] = a.h_h_hi[j];
        }

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

        // zero padding: the staging rows of levels >= 2 keep zero borders for the whole kernel
        if (a.ext == WL_EXT_ZERO && a.nlev > 1) {
            for (int i = a.g[1].sr_off + tid; i < a.lds_floats; i += kThreads) lds[i] = 0.f;
        }

        const WlStreamLevel& g0 = a.g[0];
        const int H = g0.Hs, W = g0.Ws;
        const int PADL = -a.base;
        const T* xp = a.x + (size_t)plane * H * W;
        float* SR = lds + g0.sr_off;
        float* IN = lds + g0.in_off;
        const int Wv = W / VEC;
        const int nedge = g0.sr_pitch - W;   // PADL left + the rest on the right

        Vec pf[PMAX];
        // issue the 16-byte loads of chunk [e0, e0+n) into registers (no wait)
        auto prefetch = [&](int e0, int n) {
            if (!a.vec_ok) return;
            const int total = n * Wv;
#pragma unroll
            for (int p = 0; p < PMAX; ++p) {
                const int it = tid + p * kThreads;
                if (it < total) {
                    const int i = (int)wl_mulhi((unsigned)it, a.magic_w);
                    const int c = it - i * Wv;
                    const int r = wl_ext(e0 + i, H, a.ext);
                    if (r >= 0) pf[p] = *reinterpret_cast<const Vec*>(xp + (size_t)r * W + (size_t)c * VEC);
                }
            }
        };
        // write the prefetched chunk (or load it now, scalar path) + the extended borders into SR
        auto commit = [&](int e0, int n) {
            if (a.vec_ok) {
                const int total = n * Wv;
#pragma unroll
                for (int p = 0; p < PMAX; ++p) {
                    const int it = tid + p * kThreads;
                    if (it < total) {
                        const int i = (int)wl_mulhi((unsigned)it, a.magic_w);
                        const int c = it - i * Wv;
                        const int r = wl_ext(e0 + i, H, a.ext);
                        float* d = SR + i * g0.sr_pitch + PADL + c * VEC;
#pragma unroll
                        for (int u = 0; u < VEC; ++u) d[u] = r >= 0 ? (float)pf[p].v[u] : 0.f;
                    }
                }
            } else {
                for (int i = 0; i < n; ++i) {
                    const int r = wl_ext(e0 + i, H, a.ext);
                    float* d = SR + i * g0.sr_pitch + PADL;
                    const T* xr = xp + (size_t)(r < 0 ? 0 : r) * W;
                    for (int c = (tid - i * W) & (kThreads - 1); c < W; c += kThreads)
                        d[c] = r >= 0 ? (float)xr[c] : 0.f;
                }
            }
            for (int it = tid; it < n * nedge; it += kThreads) {
                const int i = it / nedge;
                const int c = it - i * nedge;
                const int pos = c < PADL ? c : c + W;
                const int r = wl_ext(e0 + i, H, a.ext);
                const int s = wl_ext(pos - PADL, W, a.ext);
                SR[i * g0.sr_pitch + pos] = (r >= 0 && s >= 0) ? (float)xp[(size_t)r * W + s] : 0.f;
            }
        };

        const int eb = st.eb[0], ee = st.ee[0];
        prefetch(eb, (ee - eb) < a.RS ? (ee - eb) : a.RS);
        ctx.sync();   // (zero-fill above)
        for (int e0 = eb; e0 < ee; e0 += a.RS) {
            const int n = (ee - e0) < a.RS ? (ee - e0) : a.RS;
            commit(e0, n);
            ctx.sync();
            if (e0 + a.RS < ee) {
                const int n2 = (ee - e0 - a.RS) < a.RS ? (ee - e0 - a.RS) : a.RS;
                prefetch(e0 + a.RS, n2);
            }
            // level-1 row bank: extended row e -> ring slot (e - eb) mod cap
            for (int i = 0; i < n; ++i) {
                float* d0 = IN + wl_pmod(e0 + i - eb, g0.cap) * g0.in_pitch;
                row_bank(tp, SR + i * g0.sr_pitch, d0, g0.nq, i * g0.nq, tid);
            }
            ctx.sync();
            // level-1 column bank for every output row whose window is complete
            const int avail = e0 + n;
            int klast = wl_floordiv2_(avail - a.base - LT) + 1;
            if (klast > st.ke[0]) klast = st.ke[0];
            const bool deeper = a.nlev > 1;
            const int maxnk = deeper ? a.g[1].sr_rows : 0x7fffffff;
            while (st.next[0] < klast) {
                const int k0 = st.next[0];
                const int nk = (klast - k0) < maxnk ? (klast - k0) : maxnk;
                col_bank<0>(a, tp, st, lds, plane, k0, nk, tid);
                st.next[0] = k0 + nk;
                ctx.sync();
                if (deeper) level_step<1>(a, tp, st, lds, plane, k0, nk, ctx);
            }
        }
    }

    static WL_DEV int wl_floordiv2_(int v) { return v >= 0 ? v / 2 : -((-v + 1) / 2); }
};
