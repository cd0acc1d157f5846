// Levels 2 AND 1 of the DTCWT inverse in ONE streaming launch: INV_J2PLUS.forward followed by INV_J1.forward the way
// DTCWTInverse.forward chains them (reference dtcwt/transform2d.py:240-254 -> transform_funcs.py:279-307 = inv_j2plus,
// :152-184 = inv_j1), the level-1 lowpass (the level-2 reconstruction, at the resolution of the image) staying on chip.
// Written and read back it is 8 of the 28 bytes per pixel the two per-level launches move (WlDtInv2Strip + WlDtInv1Strip:
// PMC traffic 1.57x the algorithmic bytes of a J = 3 inverse in round 3); here it lives in an LDS ring of 16 rows.
//
// A workgroup (15 waves, one per CU) owns one (plane, strip of level-1 quad columns, segment of output rows) and marches
// down it once in PHASES (one workgroup barrier each; a phase = one GROUP of four output rows = one level-2 quad row = two
// level-1 quad rows: the first version had one level-1 quad row per phase and spent twice the barriers and LDS round trips):
//   * 3 level-2 STAGER waves (WlDtInv2Strip's): every lane owns one input quad of the level-2 quad row (ll2 2 x 2, six
//     (re, im) pairs: 8-byte loads PF2 phases ahead in registers), c2q, band-pair cells (+ mirrored / flipped copies at
//     the plane's edges);
//   * 4 level-2 COMPUTE waves (WlDtInv2Strip's lanes: two of the four output columns of an input quad column, row
//     interpolation into register windows, column interpolation D2 quad rows later) in the phase after the staging: the four
//     rows x two columns of the level-1 lowpass go into the LL1 RING (NG groups of 4 rows) instead of memory;
//   * 4 level-1 STAGER waves (WlDtInv1Strip's): every lane owns one level-1 quad COLUMN: the six (re, im) pairs of its two
//     quads of the phase from memory PF1 phases ahead in registers, their 4 x 2 lowpass pixels from the LL1 ring (rows above /
//     below the plane are the ring's rows flipped, columns left / right of it mirrored cells: symmetric extension of the
//     reconstruction, exact for any taps), c2q, (ll, lh, hl, hh) 16-byte cells;
//   * 4 level-1 COMPUTE waves (WlDtInv1Strip's lanes): the two columns of a quad column, row filter into register
//     windows, column filter M rows later, 8-byte stores of y - four rows per phase.
// The level-1 stagers run P1 phases behind the level-2 stagers (the launcher simulates every segment: a ring group must have
// been written a phase before it is read and must not be overwritten before its last read); waves of one role sit on
// the four SIMDs of the CU (wave index mod 4), so every SIMD carries one wave of each role.
// HBM traffic = the level-2 inputs (4 B per output pixel) + the level-1 band-pass coefficients (12 B) + y (4 B) = 20 B per
// pixel (+ the halo rows a segment shares with its neighbours), against 28 B for the two launches.
#pragma once
#include "wl_common.h"
#include "wl_dtcwt_strip.h"     // WlDtInv1Strip: row_filter2 / col_filter2 / Wave of the level-1 lanes
#include "wl_dtcwt_fused.h"     // WlDtInv2Strip: packed-FMA forms of the level-2 lanes

template <typename T>
struct WlDtInv21Args {
    typedef typename WlAcc<T>::type A;
    const T* ll2;                  // (NC, h, w) through ll2_ps / ll2_rs, h = H / 2, w = W / 2
    const T* highs2;               // (NC, 6, h/2, w/2, 2)
    const T* highs1;               // (NC, 6, H/2, W/2, 2)
    T* y;                          // (NC, H, W)
    const A* g0o; const A* g1o;    // level-1 pair (L0, L1 taps, odd)
    const A* g0a; const A* g0b; const A* g1a; const A* g1b;   // level-2 filters (LQ taps)
    int64_t NC, ll2_ps, nblocks;
    int ll2_rs;
    int H, W;
    int nstrips, strip_quads;      // level-1 quad columns per strip (even); the last strip may be narrower
    int nseg, seg_rows;            // output rows per segment (a multiple of 4)
    int P1;                        // phase of the level-1 stagers' first group
    int st1_off, st1_pitch;        // level-1 staged ring: 2 slots x 4 rows
    int st2_off, st2_pitch;        // level-2 staged ring: 2 slots x 2 rows
    int l1_off, l1_pitch;          // LL1 ring: 4 * NG rows of float32
    int lds_bytes;
};

template <typename T, int L0, int L1, int LQ>
struct WlDtInv21Strip {
    typedef WlDtInv21Args<T> Args;
    typedef WlDtInv1Strip<T, L0, L1> K1;
    typedef WlDtInv2Strip<T, LQ> K2;
    static const int CW1 = 4, SW1 = 4, CW2 = 4, SW2 = 3;
    static const int kWaves = CW1 + SW1 + CW2 + SW2;
    static const int kThreads = 64 * kWaves;
    static const int kMinWaves = 4;                    // one workgroup per CU: up to 128 registers
    static const int SZ = (int)sizeof(T);
    static const int M = K1::M, ME = K1::ME, LW = K1::LW, NPX = K1::NPX;
    static_assert(ME == 4 && LW % 4 == 0, "level-1 pairs with M = 3 (7 / 5 taps): a segment starts one whole group above its rows");
    static const int PERIOD = LW / 4;                  // phases per rotation of the level-1 windows (4 rows per phase)
    static const int m2 = K2::m2, D2 = K2::D2;
#ifndef WL_DTI21_KO
#define WL_DTI21_KO 0      // knock-out experiments (tools/build_ab_dtinv.sh): bits switch parts of the work off - wrong results
#endif
    static const int KO = WL_DTI21_KO;
#ifndef WL_DTI21_NG
#define WL_DTI21_NG 4
#endif
#ifndef WL_DTI21_PF1
#define WL_DTI21_PF1 2
#endif
#ifndef WL_DTI21_PF2
#define WL_DTI21_PF2 3
#endif
    static const int NG = WL_DTI21_NG;                 // groups (of 4 rows) of the LL1 ring
    // register sets of the stagers = how many phases ahead their loads run
    static const int PF1 = WL_DTI21_PF1, PF2 = WL_DTI21_PF2;

    struct Geo {
        // level 1 (WlDtInv1Strip::Strip)
        int q0, q1;            // level-1 quad columns [q0, q1) -> output pixel columns [2 q0, 2 q1)
        int e_lo, px0;         // first extended pixel column a lane reads; pixel column of staged cell 0 (even)
        int Qa, nq;            // level-1 quads [Qa, Qa + nq) are staged, one per stager lane
        int r_lo, r_hi;        // output rows
        int ge_first, n1;      // first extended group (of four rows; -1 above the plane), groups = phases of level-1 work
        // level 2
        int ka, kb;            // level-2 quad columns [ka, kb] whose four output columns the strip reads (inside the plane)
        int c0;                // quad column of level-2 staged cell 0 = ka - D2
        int Qa2, nq2;          // level-2 input quads [Qa2, Qa2 + nq2) are loaded, one per stager lane
        int G_lo, G_hi;        // LL1 groups the segment reads
        int n2;                // level-2 half-batches (one per phase)
        int NP;                // phases (= barriers) of the workgroup
    };
    static WL_HD int src_quad_row(int eq, int nq, bool& flip) {
        flip = false;
        if ((unsigned)eq < (unsigned)nq) return eq;
        flip = true;
        const int m = eq < 0 ? -1 - eq : 2 * nq - 1 - eq;
        return m < 0 ? 0 : (m >= nq ? nq - 1 : m);
    }
    // LL1 group an extended group of four rows reads (both of its quad rows mirror into the same group)
    static WL_HD int group_of(int ge, int HG) { bool f; return src_quad_row(ge, HG, f); }
    // level-1 phases that produce rows of the segment: extended rows [r_lo - ME, r_hi + M) in groups of four
    static WL_HD int real_n1(int r_lo, int r_hi) { return (r_hi + M - (r_lo - ME) + 3) / 4; }
    static WL_HD Geo geometry(const Args& a, int strip, int seg) {
        Geo s;
        const int W1q = a.W / 2, HG = a.H / 4, W2q = a.W / 4;
        s.q0 = strip * a.strip_quads;
        s.q1 = s.q0 + a.strip_quads < W1q ? s.q0 + a.strip_quads : W1q;
        s.e_lo = 2 * s.q0 - M;
        const int e_hi = 2 * s.q1 - 1 + M;
        s.px0 = s.e_lo >= 0 ? s.e_lo & ~1 : -((-s.e_lo + 1) & ~1);
        int qa = s.px0 / 2, qb = e_hi / 2;
        if (qa < 0) qa = 0;
        if (qb > W1q - 1) qb = W1q - 1;
        s.Qa = qa; s.nq = qb - qa + 1;
        s.r_lo = seg * a.seg_rows;
        s.r_hi = s.r_lo + a.seg_rows < a.H ? s.r_lo + a.seg_rows : a.H;
        s.ge_first = s.r_lo / 4 - 1;                   // (r_lo is a multiple of 4, ME = 4)
        s.n1 = (real_n1(s.r_lo, s.r_hi) + PF1 - 1) / PF1 * PF1;   // (a multiple of the stagers' register sets)
        // level 2: the in-plane LL1 columns [2 Qa, 2 (Qa + nq)) in groups of four
        s.ka = (2 * s.Qa) / 4;
        s.kb = (2 * (s.Qa + s.nq) - 1) / 4;
        s.c0 = s.ka - D2;
        int qa2 = s.c0 < 0 ? 0 : s.c0, qb2 = s.kb + D2;
        if (qb2 > W2q - 1) qb2 = W2q - 1;
        s.Qa2 = qa2; s.nq2 = qb2 - qa2 + 1;
        // LL1 groups: the sources of extended groups [ge_first, ge_first + n1) (mirrored at the plane's edges)
        const int ge_last = s.ge_first + s.n1 - 1;
        s.G_lo = s.ge_first < 0 ? 0 : s.ge_first;
        s.G_hi = ge_last > HG - 1 ? HG - 1 : ge_last;
        s.n2 = s.G_hi - s.G_lo + 1 + 2 * D2;
        const int np1 = a.P1 + s.n1 + 1, np2 = s.n2 + 1;
        s.NP = np1 > np2 ? np1 : np2;
        return s;
    }

    typedef T Pair2 __attribute__((ext_vector_type(2), may_alias));

    // ---- level-2 stager wave (WlDtInv2Strip::stager on the phase schedule): phase j stages half-batch j ---------------------
    struct Quad2 { Pair2 l0, l1, b[6]; };
    static WL_DEV void stager2(const Args& a, const Geo& s, const WlCtx& ctx, int64_t plane, int lane, int sidx) {
        const int H2q = a.H / 4, W2q = a.W / 4;
        const size_t qplane = (size_t)H2q * W2q;
        const int j = 64 * sidx + lane;
        const int Q = s.Qa2 + j;
        const bool qon = j < s.nq2;
        const T* llp = a.ll2 + (size_t)plane * a.ll2_ps + 2 * Q;
        const T* hp = a.highs2 + (size_t)plane * 6 * qplane * 2 + 2 * Q;
        const int cdst = (Q - s.c0) * 16;
        const int hp2 = a.st2_pitch / 2;
        int mdst = -1;
        if (qon) {
            if (Q < D2 && -1 - Q >= s.c0) mdst = (-1 - Q - s.c0) * 16;
            if (Q >= W2q - D2 && 2 * W2q - 1 - Q <= s.kb + D2) mdst = (2 * W2q - 1 - Q - s.c0) * 16;
        }
        const int eq0 = s.G_lo - D2;
        auto load = [&](int h, Quad2& qd) {
            bool flip;
            const int sq = src_quad_row(eq0 + h, H2q, flip);
            if (!qon) return;
            qd.l0 = *reinterpret_cast<const Pair2*>(llp + (size_t)(2 * sq) * a.ll2_rs);
            qd.l1 = *reinterpret_cast<const Pair2*>(llp + (size_t)(2 * sq + 1) * a.ll2_rs);
#pragma unroll
            for (int o = 0; o < 6; ++o) qd.b[o] = *reinterpret_cast<const Pair2*>(hp + ((size_t)o * qplane + (size_t)sq * W2q) * 2);
        };
        const float k = (float)WL_SQRT1_2;
        auto stage = [&](int hb, const Quad2& qd) {
            bool flip;
            src_quad_row(eq0 + hb, H2q, flip);
            char* sslot = ctx.smem + a.st2_off + (hb & 1) * 2 * a.st2_pitch;
            if (!qon) return;
#if defined(__HIPCC__)
            if (KO & 8) { asm volatile("" :: "v"(qd.l0), "v"(qd.l1), "v"(qd.b[0]), "v"(qd.b[1]), "v"(qd.b[2]), "v"(qd.b[3]), "v"(qd.b[4]), "v"(qd.b[5])); return; }
#endif
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
                char* drow = sslot + (flip ? 1 - i : i) * a.st2_pitch;
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
        // PF2 register sets: the loads of half-batch j + PF2 go out right behind the staging of j (no load sits inside a
        // branch: the compiler counts them and waits for the oldest set)
        const int n2e = (s.n2 + PF2 - 1) / PF2 * PF2;
        Quad2 qq[PF2];
#pragma unroll
        for (int u = 0; u < PF2; ++u) load(u < s.n2 ? u : s.n2 - 1, qq[u]);
        int done = 0;
        for (int hb = 0; hb < n2e; hb += PF2) {
#pragma unroll
            for (int u = 0; u < PF2; ++u) {
                if (hb + u < s.n2) stage(hb + u, qq[u]);
                load(hb + u + PF2 < s.n2 ? hb + u + PF2 : s.n2 - 1, qq[u]);
                if (done < s.NP) { ctx.sync(); ++done; }
            }
        }
        for (; done < s.NP; ++done) ctx.sync();
    }

    // ---- level-2 compute wave: WlDtInv2Strip::compute, the output rows into the LL1 ring; phase j + 1 works on half-batch j --
    static WL_DEV void compute2(const Args& a, const Geo& s, const WlCtx& ctx, int cw, int lane) {
        const int ph_c = cw >> 1;                              // column phase: 0 -> columns 4q, 4q+1 (e = 1); 1 -> 4q+2, 4q+3 (e = 0)
        const int q = s.ka + 64 * (cw & 1) + lane;
        const bool active = q <= s.kb;
        wl_v2 PL[2][m2], PH[2][m2];
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int t = 0; t < m2; ++t) {
                PL[e][t] = wl_uniform_v2(wl_v2{(float)a.g0b[e + 2 * t], (float)a.g0a[e + 2 * t]});
                PH[e][t] = wl_uniform_v2(wl_v2{(float)a.g1b[e + 2 * t], (float)a.g1a[e + 2 * t]});
            }
        const int er = 1 - ph_c;
        const int soff = (active ? q - s.ka : 0) * 16;         // cell of quad column q - D2
        const int hp2 = a.st2_pitch / 2;
        const int loff = (4 * (active ? q - s.ka : 0) + 2 * ph_c) * 4;   // byte offset of my two columns in an LL1 ring row
        wl_v2 wA[m2][2], wB[m2][2];
#pragma unroll
        for (int t = 0; t < m2; ++t) { wA[t][0] = wA[t][1] = wB[t][0] = wB[t][1] = wl_v2{0.f, 0.f}; }
        char* const smem = ctx.smem;
        int done = 0;
        ctx.sync(); ++done;                                    // phase 0: the first level-2 quad row is being staged
        for (int hb0 = 0; hb0 < s.n2; hb0 += m2) {
#pragma unroll
            for (int ph = 0; ph < m2; ++ph) {
                const int hb = hb0 + ph;
                if (hb >= s.n2) break;
                if (active && !(KO & 4)) {
                    // row interpolation of both rows of the quad row
                    const char* slot = smem + a.st2_off + (hb & 1) * 2 * a.st2_pitch + soff;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        wl_v2 A = {0.f, 0.f}, B = {0.f, 0.f};
#pragma unroll
                        for (int t = 0; t < m2; ++t) {
                            const wl_vf4 c0 = *reinterpret_cast<const wl_vf4*>(slot + i * a.st2_pitch + 16 * t);
                            const wl_vf4 c1 = *reinterpret_cast<const wl_vf4*>(slot + i * a.st2_pitch + hp2 + 16 * t);
                            if (er) {
                                K2::fma_ee(A, PL[1][t], wl_v2{c0.x, c0.y}); K2::fma_sw(A, PH[1][t], wl_v2{c0.z, c0.w});
                                K2::fma_ee(B, PL[1][t], wl_v2{c1.x, c1.y}); K2::fma_sw(B, PH[1][t], wl_v2{c1.z, c1.w});
                            } else {
                                K2::fma_ee(A, PL[0][t], wl_v2{c0.x, c0.y}); K2::fma_sw(A, PH[0][t], wl_v2{c0.z, c0.w});
                                K2::fma_ee(B, PL[0][t], wl_v2{c1.x, c1.y}); K2::fma_sw(B, PH[0][t], wl_v2{c1.z, c1.w});
                            }
                        }
                        wA[ph][i] = A; wB[ph][i] = B;
                    }
                    // column interpolation of the group kr = G_lo - 2 D2 + hb -> LL1 ring
                    const int kr = s.G_lo - 2 * D2 + hb;
                    if (kr >= s.G_lo && kr <= s.G_hi) {
                        wl_v2 y0 = {0.f, 0.f}, y1 = {0.f, 0.f}, y2 = {0.f, 0.f}, y3 = {0.f, 0.f};
#pragma unroll
                        for (int t = 0; t < m2; ++t) {
                            const int sl = (ph + 1 + t) % m2;
                            K2::template fma_cc<0>(y0, wA[sl][0], PL[1][t]); K2::template fma_cc<0>(y0, wB[sl][1], PH[1][t]);
                            K2::template fma_cc<1>(y1, wA[sl][1], PL[1][t]); K2::template fma_cc<1>(y1, wB[sl][0], PH[1][t]);
                            K2::template fma_cc<0>(y2, wA[sl][0], PL[0][t]); K2::template fma_cc<0>(y2, wB[sl][1], PH[0][t]);
                            K2::template fma_cc<1>(y3, wA[sl][1], PL[0][t]); K2::template fma_cc<1>(y3, wB[sl][0], PH[0][t]);
                        }
                        char* const rp = smem + a.l1_off + (4 * (kr % NG)) * a.l1_pitch + loff;
                        *reinterpret_cast<wl_f2*>(rp) = wl_f2{y0.x, y0.y};
                        *reinterpret_cast<wl_f2*>(rp + a.l1_pitch) = wl_f2{y1.x, y1.y};
                        *reinterpret_cast<wl_f2*>(rp + 2 * a.l1_pitch) = wl_f2{y2.x, y2.y};
                        *reinterpret_cast<wl_f2*>(rp + 3 * a.l1_pitch) = wl_f2{y3.x, y3.y};
                    }
                }
                ctx.sync(); ++done;
            }
        }
        for (; done < s.NP; ++done) ctx.sync();
    }

    // ---- level-1 stager wave: WlDtInv1Strip::stager, two quad rows per phase, the lowpass pixels from the LL1 ring ----------
    struct Quad1 { Pair2 b[2][6]; };
    static WL_DEV void stager1(const Args& a, const Geo& s, const WlCtx& ctx, int64_t plane, int lane, int sidx) {
        const int H1q = a.H / 2, W1q = a.W / 2;
        const size_t qplane = (size_t)H1q * W1q;
        const int j = 64 * sidx + lane;
        const int Q = s.Qa + j;
        const bool qon = j < s.nq;
        const T* hp = a.highs1 + (size_t)plane * 6 * qplane * 2 + 2 * Q;
        const int e_hi = 2 * s.q1 - 1 + M;
        const int hp1 = a.st1_pitch / 2;
        int cdst[2], mdst[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int p = 2 * Q + c;
            const int cc = p - s.px0;
            cdst[c] = (cc & 1) * hp1 + (cc >> 1) * 16;
            int e = -1000000;
            if (qon) {
                if (p < M && -1 - p >= s.e_lo) e = -1 - p;
                if (p >= a.W - M && 2 * a.W - 1 - p <= e_hi) e = 2 * a.W - 1 - p;
            }
            const int mc = e - s.px0;
            mdst[c] = e == -1000000 ? -1 : (mc & 1) * hp1 + (mc >> 1) * 16;
        }
        const int lcol = (2 * Q - 4 * s.ka) * 4;                 // my two lowpass columns in an LL1 ring row
        const int eq_first = 2 * s.ge_first;
        auto load = [&](int h, Quad1& qd) {
            if (!qon) return;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                bool flip;
                const int sq = src_quad_row(eq_first + 2 * h + r, H1q, flip);
                if (KO & 32) {
#pragma unroll
                    for (int o = 0; o < 6; ++o) qd.b[r][o] = Pair2{(T)0.f, (T)0.f};
                } else {
#pragma unroll
                    for (int o = 0; o < 6; ++o) qd.b[r][o] = *reinterpret_cast<const Pair2*>(hp + ((size_t)o * qplane + (size_t)sq * W1q) * 2);
                }
            }
        };
        const float k = (float)WL_SQRT1_2;
        auto stage = [&](int hb, const Quad1& qd) {
            char* sslot = ctx.smem + a.st1_off + (hb & 1) * 4 * a.st1_pitch;
            if (!qon) return;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
#if defined(__HIPCC__)
                if (KO & 2) { asm volatile("" :: "v"(qd.b[r][0]), "v"(qd.b[r][1]), "v"(qd.b[r][2]), "v"(qd.b[r][3]), "v"(qd.b[r][4]), "v"(qd.b[r][5])); continue; }
#endif
                bool flip;
                const int sq = src_quad_row(eq_first + 2 * hb + r, H1q, flip);
                const char* lrow = ctx.smem + a.l1_off + ((2 * sq) % (4 * NG)) * a.l1_pitch + lcol;
                const wl_f2 l0 = *reinterpret_cast<const wl_f2*>(lrow);
                const wl_f2 l1 = *reinterpret_cast<const wl_f2*>(lrow + a.l1_pitch);
                float re[6], im[6];
#pragma unroll
                for (int o = 0; o < 6; ++o) { re[o] = (float)qd.b[r][o].x; im[o] = (float)qd.b[r][o].y; }
                // c2q: orientation pairs (0,5) -> lh, (2,3) -> hl, (1,4) -> hh;  v[row][col][ll, lh, hl, hh]
                float v[2][2][4];
#pragma unroll
                for (int ch = 1; ch < 4; ++ch) {
                    const int o1 = ch == 1 ? 0 : (ch == 2 ? 2 : 1), o2 = ch == 1 ? 5 : (ch == 2 ? 3 : 4);
                    v[0][0][ch] = (re[o1] + re[o2]) * k; v[0][1][ch] = (im[o1] + im[o2]) * k;
                    v[1][0][ch] = (im[o1] - im[o2]) * k; v[1][1][ch] = (re[o2] - re[o1]) * k;
                }
                v[0][0][0] = l0.x; v[0][1][0] = l0.y; v[1][0][0] = l1.x; v[1][1][0] = l1.y;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    char* drow = sslot + (2 * r + (flip ? 1 - i : i)) * a.st1_pitch;
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        wl_vf4 w;
                        w.x = v[i][c][0]; w.y = v[i][c][1]; w.z = v[i][c][2]; w.w = v[i][c][3];
                        *reinterpret_cast<wl_vf4*>(drow + cdst[c]) = w;
                        if (mdst[c] >= 0) *reinterpret_cast<wl_vf4*>(drow + mdst[c]) = w;
                    }
                }
            }
        };
        // phase P1 + h: stage group h (band-pass pairs loaded PF1 phases earlier: PF1 register sets)
        Quad1 qq[PF1];
#pragma unroll
        for (int u = 0; u < PF1; ++u) load(u < s.n1 ? u : s.n1 - 1, qq[u]);
        int done = 0;
        for (; done < a.P1; ++done) ctx.sync();
        for (int hb = 0; hb < s.n1; hb += PF1) {               // (n1 is a multiple of PF1)
#pragma unroll
            for (int u = 0; u < PF1; ++u) {
                stage(hb + u, qq[u]);
                load(hb + u + PF1 < s.n1 ? hb + u + PF1 : s.n1 - 1, qq[u]);
                ctx.sync(); ++done;
            }
        }
        for (; done < s.NP; ++done) ctx.sync();
    }

    // ---- level-1 compute wave: WlDtInv1Strip::compute, four rows per phase -----------------------------------------------
    static WL_DEV void compute1(const Args& a, const Geo& s, const WlCtx& ctx, int64_t plane, int cw, int lane) {
        const int q = s.q0 + 64 * cw + lane;
        const bool active = q < s.q1;
        typename K1::Wave R;
#pragma unroll
        for (int t = 0; t < L0; ++t) R.r0[t] = wl_uniform_v2(wl_v2{(float)a.g0o[t], (float)a.g0o[t]});
#pragma unroll
        for (int t = 0; t < L1; ++t) R.r1[t] = wl_uniform_v2(wl_v2{(float)a.g1o[t], (float)a.g1o[t]});
        const int M0 = L0 / 2, M1 = L1 / 2;
#pragma unroll
        for (int t = 0; t < 2 * M + 1; ++t) {
            const int t0 = t - (M - M0), t1 = t - (M - M1);
            const float v0 = t0 >= 0 && t0 < L0 ? (float)a.g0o[t0 >= 0 && t0 < L0 ? t0 : 0] : 0.f;
            const float v1 = t1 >= 0 && t1 < L1 ? (float)a.g1o[t1 >= 0 && t1 < L1 ? t1 : 0] : 0.f;
            R.cc[t] = wl_uniform_v2(wl_v2{v0, v1});
        }
        const int soff = (active ? q - s.q0 : 0) * 16;
        const int hp1 = a.st1_pitch / 2;
        char* const yp = reinterpret_cast<char*>(a.y + (size_t)plane * a.H * a.W);
        const unsigned rowb = (unsigned)a.W * SZ, colb = (unsigned)(2 * q) * SZ;
        wl_v2 wa[LW], wb[LW];
#pragma unroll
        for (int t = 0; t < LW; ++t) wa[t] = wb[t] = wl_v2{0.f, 0.f};
        char* const smem = ctx.smem;
        const int e_first = 4 * s.ge_first;
        int done = 0;
        for (; done <= a.P1; ++done) ctx.sync();               // phases 0 .. P1: the first group is staged in phase P1
        for (int hb0 = 0; hb0 < s.n1; hb0 += PERIOD) {
#pragma unroll
            for (int ph = 0; ph < PERIOD; ++ph) {
                const int hb = hb0 + ph;
                if (hb >= s.n1) break;
                if (active && !(KO & 1)) {
                    const char* slot = smem + a.st1_off + (hb & 1) * 4 * a.st1_pitch + soff;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        wl_vf4 px[NPX];
#pragma unroll
                        for (int u = 0; u < NPX; ++u)
                            px[u] = *reinterpret_cast<const wl_vf4*>(slot + i * a.st1_pitch + (((M & 1) + u) & 1) * hp1 + (((M & 1) + u) >> 1) * 16);
                        const int w = (4 * ph + i) % LW;
                        K1::row_filter2(R, px, wa[w], wb[w]);
                        const int o = e_first + 4 * hb + i - M;
                        float ya, yb;
                        K1::col_filter2(R, wa, wb, (w + LW - M) % LW, ya, yb);
                        if (o >= s.r_lo && o < s.r_hi && (!(KO & 16) || ya == 12345.678f)) {
                            typedef T Vec2 __attribute__((ext_vector_type(2)));
                            Vec2 v = {(T)ya, (T)yb};
                            *reinterpret_cast<Vec2*>(yp + (unsigned)o * rowb + colb) = v;
                        }
                    }
                }
                ctx.sync(); ++done;
            }
        }
        for (; done < s.NP; ++done) ctx.sync();
    }

    static WL_DEV void run(const Args& a, const WlCtx& ctx) {
        const int tid = ctx.tid;
        const int wave = wl_uniform(tid >> 6), lane = tid & 63;
        const int64_t lbid = wl_xcd_remap(ctx.bid, a.nblocks);
        const int per_plane = a.nstrips * a.nseg;
        const int64_t plane = lbid / per_plane;
        const int rem = (int)(lbid - plane * per_plane);
        const int seg = rem / a.nstrips, strip = rem - seg * a.nstrips;
        const Geo s = geometry(a, strip, seg);
        for (int i = tid * 16; i < a.lds_bytes; i += kThreads * 16) {
            wl_f4 z; z.x = z.y = z.z = z.w = 0.f;
            *reinterpret_cast<wl_f4*>(ctx.smem + i) = z;
        }
        ctx.sync();
        // roles by wave index so that every SIMD (wave index mod 4) carries one wave of each role
        if (wave < CW1) {
            if (64 * wave < s.q1 - s.q0) compute1(a, s, ctx, plane, wave, lane);
            else for (int p = 0; p < s.NP; ++p) ctx.sync();
        } else if (wave < CW1 + SW1) {
#if defined(__HIPCC__)
            __builtin_amdgcn_s_setprio(2);
#endif
            stager1(a, s, ctx, plane, lane, wave - CW1);
        } else if (wave < CW1 + SW1 + CW2) {
            compute2(a, s, ctx, wave - CW1 - SW1, lane);
        } else {
#if defined(__HIPCC__)
            __builtin_amdgcn_s_setprio(2);
#endif
            stager2(a, s, ctx, plane, lane, wave - CW1 - SW1 - CW2);
        }
    }
};
