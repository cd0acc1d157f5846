// Common definitions for the gfx950 wavelet filterbank kernels.
//
// The kernel BODIES in this directory are plain C++ templates over a small execution context
// (WlCtx: thread id, block id, LDS pointer, barrier).  They are compiled
//   * by hipcc for gfx950 (wl_hip.hip) - the product, and
//   * by a host compiler into tests/emu/libwl_emu.so, where a fibre scheduler runs one workgroup
//     at a time with exact barrier semantics.  The emulator exists so the index arithmetic of the
//     kernels can be debugged in the GPU-less authoring container; it is test infrastructure and
//     is never loaded by the pytorch_wavelets_amd package.
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define WL_DEV __device__ __forceinline__
#define WL_HD __host__ __device__ __forceinline__
#else
#define WL_DEV inline __attribute__((always_inline))
#define WL_HD inline __attribute__((always_inline))
#endif

// dtype codes of the C ABI (include/wavelets_hip.h)
#define WL_F32 0
#define WL_F16 1
#define WL_F64 2

// boundary-extension codes used INSIDE kernels (the ABI takes the reference's mode ints)
#define WL_EXT_ZERO 0
#define WL_EXT_SYM 1       // half-sample symmetric
#define WL_EXT_REFL 2      // whole-sample reflect
#define WL_EXT_PERIODIC 3  // wrap
#define WL_EXT_PER 4       // periodization: odd length repeats the last sample, then wraps

typedef _Float16 wl_half;

template <typename T> struct WlAcc { typedef float type; };
template <> struct WlAcc<double> { typedef double type; };

// value of `v` in the previous lane of the wave (lane 0 gets its own).  The host emulation runs one fibre per
// lane to the next barrier, so it emulates the shuffle through a per-workgroup exchange array + two barriers.
#if defined(__HIPCC__)
WL_DEV float wl_shfl_up1(float v) { return __shfl_up(v, 1); }
WL_DEV float wl_shfl(float v, int src_lane) { return __shfl(v, src_lane, 64); }   // value of `v` in lane src_lane & 63
#else
float wl_shfl_up1(float v);
float wl_shfl(float v, int src_lane);
#endif

struct __attribute__((may_alias)) alignas(16) wl_f4 { float x, y, z, w; };
struct __attribute__((may_alias)) alignas(8) wl_f2 { float x, y; };
typedef float wl_vf4 __attribute__((ext_vector_type(4), may_alias));   // one 16-byte LDS / global access, never split
typedef float wl_v2 __attribute__((ext_vector_type(2)));   // (low-band, high-band) pair: one v_pk_fma_f32 per tap

struct WlCtx {
    int tid;        // thread index in the workgroup
    int nthreads;   // workgroup size
    int64_t bid;    // linear workgroup index
    char* smem;     // dynamic LDS base (16-byte aligned)
    unsigned lds_base;   // LDS byte address of smem (M0 base of the LDS-DMA loads; unused by the emulator)
#if defined(__HIPCC__)
    WL_DEV void sync() const { __syncthreads(); }
#else
    void (*sync_fn)(void*);
    void* sync_arg;
    inline void sync() const { sync_fn(sync_arg); }
#endif
};

// ---------------------------------------------------------------------------------------------
// Wave-level helpers of the streaming kernels (wl_dwt_rows.h).
//   wl_uniform : a value the caller knows to be wave-uniform, pinned to a scalar register;
//   wl_dma16   : asynchronous global -> LDS copy of 16 bytes per lane (global_load_lds_dwordx4): the wave writes the
//                64 x 16 B of its lanes with lane_on to LDS bytes [lds_off + 16*lane, +16), lds_off wave-uniform; at
//                least one lane of the wave must be on (the wait counts are per wave-instruction).  The data
//                is visible to the issuing wave after wl_wait_vm<N>() (at most N younger DMA / vector-memory
//                operations of that wave still outstanding) and to other waves after a barrier behind that wait;
//   wl_dma4    : the same with 4 bytes per lane (global_load_lds_dword) at LDS bytes [lds_off + 4*lane, +4);
//   The instruction is emitted through inline assembly on purpose: the compiler would otherwise order every later
//   LDS read behind ALL outstanding DMA loads (s_waitcnt vmcnt(0)), i.e. no prefetch distance.
// The host emulation defers each copy until the lane's wl_wait_vm releases it, so a missing or too permissive wait
// reads NaN-poisoned LDS in the CPU tests.
// ---------------------------------------------------------------------------------------------
// WL_STREAM_NT (A/B builds, tools/build_ab.sh): bit 0 = the LDS-DMA loads carry the non-temporal hint (data read once),
// bit 1 = the band / output stores of the streaming kernels are non-temporal
#ifndef WL_STREAM_NT
#define WL_STREAM_NT 0
#endif
#if WL_STREAM_NT & 1
#define WL_DMA_NT " nt"
#else
#define WL_DMA_NT ""
#endif
#if defined(__HIPCC__)
template <typename T> WL_DEV void wl_store_stream(T* p, T v) {
#if WL_STREAM_NT & 2
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
#else
template <typename T> inline void wl_store_stream(T* p, T v) { *p = v; }
#endif
// Two adjacent elements (a, b) to sbase + voff: sbase is wave-uniform (a scalar register pair), voff this lane's 32-bit byte
// offset - the "scalar base + vector offset" form of global_store.  Written as a pointer sum the compiler widens voff to a
// 64-bit vector address per store (a v_lshl_add_u64 each: 12 per half-batch in the 16-tap strip kernel); float16 pairs are
// converted two at a time (v_cvt_pk_f16_f32, gfx950) instead of two conversions + v_perm.
#if defined(__HIPCC__)
WL_DEV void wl_store2_s(char* sbase, unsigned voff, float a, float b, float*) {
    wl_v2 v = {a, b};
    asm volatile("global_store_dwordx2 %0, %1, %2" :: "v"(voff), "v"(v), "s"(sbase) : "memory");
}
WL_DEV void wl_store2_s(char* sbase, unsigned voff, float a, float b, _Float16*) {
    typedef _Float16 H2 __attribute__((ext_vector_type(2)));
    const wl_v2 f = {a, b};
    const unsigned p = __builtin_bit_cast(unsigned, __builtin_convertvector(f, H2));     // v_cvt_pk_f16_f32 (gfx950)
    asm volatile("global_store_dword %0, %1, %2" :: "v"(voff), "v"(p), "s"(sbase) : "memory");
}
WL_DEV void wl_store2_s(char* sbase, unsigned voff, float a, float b, double*) {
    double* q = reinterpret_cast<double*>(sbase + voff); q[0] = a; q[1] = b;
}
#else
template <typename T> inline void wl_store2_s(char* sbase, unsigned voff, float a, float b, T*) {
    T* q = reinterpret_cast<T*>(sbase + voff); q[0] = (T)a; q[1] = (T)b;
}
#endif
#if defined(__HIPCC__)
WL_DEV int wl_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
WL_DEV float wl_uniform_f(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
WL_DEV void wl_dma16(const WlCtx& ctx, unsigned lds_off, const void* gsrc, bool lane_on) {
    const unsigned m0 = __builtin_amdgcn_readfirstlane(ctx.lds_base + lds_off);
    // lanes that are off copy nothing; the instruction still counts once per wave as long as one lane is on
    if (lane_on)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" WL_DMA_NT : : "s"(m0), "v"(gsrc) : "memory", "m0");
}
// the same with the global address split into a wave-uniform base (scalar registers) and a 32-bit per-lane byte offset
WL_DEV void wl_dma16_s(const WlCtx& ctx, unsigned lds_off, const void* sbase, unsigned voff, bool lane_on) {
    const unsigned m0 = __builtin_amdgcn_readfirstlane(ctx.lds_base + lds_off);
    if (lane_on)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" WL_DMA_NT : : "s"(m0), "v"(voff), "s"(sbase) : "memory", "m0");
}
WL_DEV void wl_dma4_s(const WlCtx& ctx, unsigned lds_off, const void* sbase, unsigned voff, bool lane_on) {
    const unsigned m0 = __builtin_amdgcn_readfirstlane(ctx.lds_base + lds_off);
    if (lane_on)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" WL_DMA_NT : : "s"(m0), "v"(voff), "s"(sbase) : "memory", "m0");
}
WL_DEV void wl_dma4(const WlCtx& ctx, unsigned lds_off, const void* gsrc, bool lane_on) {   // 4 bytes per lane
    const unsigned m0 = __builtin_amdgcn_readfirstlane(ctx.lds_base + lds_off);
    if (lane_on)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" WL_DMA_NT : : "s"(m0), "v"(gsrc) : "memory", "m0");
}
template <int N> WL_DEV void wl_wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));   // vmcnt(N), others untouched
}
#else
inline int wl_uniform(int v) { return v; }
inline float wl_uniform_f(float v) { return v; }
void wl_dma16(const WlCtx& ctx, unsigned lds_off, const void* gsrc, bool lane_on);
void wl_dma4(const WlCtx& ctx, unsigned lds_off, const void* gsrc, bool lane_on);
inline void wl_dma16_s(const WlCtx& ctx, unsigned lds_off, const void* sbase, unsigned voff, bool lane_on) { wl_dma16(ctx, lds_off, (const char*)sbase + voff, lane_on); }
inline void wl_dma4_s(const WlCtx& ctx, unsigned lds_off, const void* sbase, unsigned voff, bool lane_on) { wl_dma4(ctx, lds_off, (const char*)sbase + voff, lane_on); }
void wl_emu_wait_vm(int n);
template <int N> inline void wl_wait_vm() { wl_emu_wait_vm(N); }
#endif

#define WL_IROWS_MAX_VM 48      // DMA instructions a loader wave may have outstanding (range of wl_wait_vm_dyn)
#if defined(__HIPCC__)
WL_DEV void wl_fail() { __builtin_trap(); }
// s_waitcnt vmcnt(n) for a wave-uniform run-time n in [0, 48].  The count is an immediate, and both a switch and a
// hand-written decision tree come out of the compiler as a chain through every case (hundreds of cycles per
// half-batch in the loader waves, which every other wave then waits for at the barrier): a computed jump into a table
// of (s_waitcnt, s_branch) pairs instead - 8 bytes per entry, the table starts 20 bytes behind the s_getpc.
WL_DEV void wl_wait_vm_dyn(int n) {
    static_assert(WL_IROWS_MAX_VM == 48, "the table below has 49 entries");
    int t = n < WL_IROWS_MAX_VM ? n : WL_IROWS_MAX_VM;
    asm volatile(
        "s_getpc_b64 vcc\n\t"
        "s_lshl_b32 %0, %0, 3\n\t"
        "s_add_u32 %0, %0, 20\n\t"
        "s_add_u32 vcc_lo, vcc_lo, %0\n\t"
        "s_addc_u32 vcc_hi, vcc_hi, 0\n\t"
        "s_setpc_b64 vcc\n\t"
        "s_waitcnt vmcnt(0)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(1)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(2)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(3)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(4)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(5)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(6)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(7)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(8)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(9)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(10)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(11)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(12)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(13)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(14)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(15)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(16)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(17)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(18)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(19)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(20)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(21)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(22)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(23)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(24)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(25)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(26)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(27)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(28)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(29)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(30)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(31)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(32)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(33)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(34)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(35)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(36)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(37)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(38)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(39)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(40)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(41)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(42)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(43)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(44)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(45)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(46)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(47)\n\ts_branch .Lwl_vm_end_%=\n\t"
        "s_waitcnt vmcnt(48)\n\ts_branch .Lwl_vm_end_%=\n\t"
        ".Lwl_vm_end_%=:"
        : "+s"(t) : : "memory", "vcc", "scc");
}
#else
inline void wl_fail() { abort(); }
inline void wl_wait_vm_dyn(int n) { wl_emu_wait_vm(n < WL_IROWS_MAX_VM ? n : WL_IROWS_MAX_VM); }
#endif


// ---------------------------------------------------------------------------------------------
// Tap-relation guards.  Some kernel variants hold fewer taps in their scalar registers because of a RELATION between the
// filter banks (quadrature-mirror highpass banks, the same bank on both axes).  The host's knowledge of such a relation is a
// hint only (a cache keyed on tensor versions cannot see writes through `.data`): the relation is verified HERE, on the
// device, against the taps as they are when the kernel runs - every wave of the grid reads the same <= 4 x 20 taps and takes
// the same decision.  guard 0: no check (the relation is proven by construction, e.g. identical pointers, or not used);
// guard 1: run only if the relation holds (the hinted variant); guard 2: run only if it does NOT hold (the two-bank variant
// launched behind a hinted one as its armed fallback: exactly one of the two does the work, the other returns at once).
// The reference reads its filter buffers on every forward (dwt/transform2d.py:63-74, :131-148): so does this.
// ---------------------------------------------------------------------------------------------
WL_HD bool wl_guard_pass(int guard, bool holds) { return guard == 0 || (guard == 1) == holds; }
// hi[t] == (-1)^t lo[L-1-t] for every t (float compare: -0 == +0 is the same filter, a NaN tap never passes)
WL_HD bool wl_taps_qmf(const float* lo, const float* hi, int L) {
    bool ok = true;
    for (int t = 0; t < L; ++t) {
        const float m = lo[L - 1 - t];
        ok = ok & (hi[t] == ((t & 1) ? -m : m));       // (no short circuit: straight-line compares after unrolling)
    }
    return ok;
}
WL_HD bool wl_taps_same(const float* a, const float* b, int L) {
    bool ok = true;
    for (int t = 0; t < L; ++t) ok = ok & (a[t] == b[t]);
    return ok;
}

// ---------------------------------------------------------------------------------------------
// Boundary extension: extended position i of a length-n signal -> source position, or -1 for a
// zero sample.  Closed forms of SURVEY.md §8 (reference: dwt/lowlevel.py:28-88, utils.py:146-174).
// ---------------------------------------------------------------------------------------------
WL_HD int wl_pmod(int i, int p) {
    int r = i % p;
    return r < 0 ? r + p : r;
}

WL_HD int wl_ext(int i, int n, int ext) {
    if ((unsigned)i < (unsigned)n) return i;
    switch (ext) {
        case WL_EXT_ZERO:
            return -1;
        case WL_EXT_SYM: {
            int j = wl_pmod(i, 2 * n);
            return j < n ? j : 2 * n - 1 - j;
        }
        case WL_EXT_REFL: {
            if (n == 1) return 0;
            int p = 2 * n - 2;
            int j = wl_pmod(i, p);
            return j < n ? j : p - j;
        }
        case WL_EXT_PERIODIC:
            return wl_pmod(i, n);
        default: {  // WL_EXT_PER
            int ne = n + (n & 1);
            int j = wl_pmod(i, ne);
            return j == n ? n - 1 : j;
        }
    }
}

// Single-fold form of wl_ext for zero / symmetric / reflect (no division: used once per row by every wave of the
// streaming kernels); valid while -n <= i < 2n (the launcher checks that).
WL_HD int wl_ext1(int i, int n, int ext) {
    const int s = ext == WL_EXT_SYM ? 1 : 0;                        // branch-free on purpose (scalar selects)
    const int fold = (i < 0 ? -s : 2 * n - 2 + s) - i;
    const int out = ext == WL_EXT_ZERO ? -1 : fold;
    return (unsigned)i < (unsigned)n ? i : out;
}

// The reference's periodization analysis evaluated literally (dwt/lowlevel.py:134-150): the signal (its last sample
// repeated when n is odd: ne samples) is rolled by L2 = L/2 and zero padded by the convolution; Z(i) = source position of
// position i of that rolled signal, or -1 for a zero.  The reference's roll() (:9-25) is built from two slices and
// degenerates into the identity once the shift reaches twice the length (signals of 1-4 samples under 10-20 taps).
WL_HD int wl_per_rolled_src(int i, int n, int L2) {
    const int ne = n + (n & 1);
    if ((unsigned)i >= (unsigned)ne) return -1;
    const int t = (i + (L2 < 2 * ne ? L2 : 0)) % ne;
    return t < n ? t : n - 1;
}

// Replicate-padded view used by the DTCWT modules (dtcwt/transform2d.py:116-120, :131-135):
// a virtual signal of length n + pad_lo + pad_hi whose first pad_lo / last pad_hi samples
// replicate the edge.  Returns the source position of virtual position v (after extension).
WL_HD int wl_ext_padded(int v, int n, int pad_lo, int pad_hi, int ext) {
    if ((unsigned)(v - pad_lo) < (unsigned)n) return v - pad_lo;   // interior: no extension, no padding
    int nv = n + pad_lo + pad_hi;
    int j = wl_ext(v, nv, ext);
    if (j < 0) return -1;
    j -= pad_lo;
    return j < 0 ? 0 : (j >= n ? n - 1 : j);
}

WL_HD int wl_cdiv(int a, int b) { return (a + b - 1) / b; }
WL_HD int64_t wl_cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
WL_HD int wl_align_up(int a, int b) { return (a + b - 1) / b * b; }

// XCD-aware block remap (MI355X: workgroup b is dispatched to XCD b % 8, each XCD has its own L2).  Returns the
// LOGICAL block a hardware block should work on so that every XCD walks one contiguous range of logical blocks:
// neighbouring tiles (which share halo rows / cache lines) then meet in the same L2.  Placement only affects speed.
WL_HD int64_t wl_xcd_remap(int64_t bid, int64_t nblocks) {
    if (nblocks <= 0) return bid;   // remap disabled
    const int64_t q = nblocks / 8, r = nblocks % 8;
    const int64_t x = bid % 8, i = bid / 8;
    return x * q + (x < r ? x : r) + i;
}
