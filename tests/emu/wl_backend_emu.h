// Host emulation backend (TEST INFRASTRUCTURE, never loaded by the product package).
// Runs each workgroup as cooperative fibres on one OS thread: a fibre runs until its next barrier
// (WlCtx::sync), its next wave shuffle (wl_shfl_up1) or its end.  Barriers release when every live fibre of the
// workgroup waits at one; a shuffle resolves when every live lane of that 64-lane wave waits at it, so barrier and
// wave semantics are exact.  The visiting order of the fibres alternates forward / backward between barrier
// phases so that a MISSING barrier in a kernel shows up as a wrong result instead of passing by luck.  LDS is
// poisoned with NaNs.
// The library is built from several translation units side by side (tests/emu/build.sh, like the four units of the HIP
// build): everything at namespace scope here is `inline`, so the units share one kernel log and one current-block pointer.
#pragma once
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>
#include <vector>
#include "../../pytorch_wavelets_amd/csrc/wl_common.h"

#define WL_BACKEND_NAME "emu"

// a deliberately tiny "chip" so that persistent kernels walk several tiles per workgroup in the tests
inline int wl_emu_cus_v = 2;       // (wl_emu_set_cus of tests/emu/wl_emu_api.cpp: launcher policies that depend on the chip's size)
inline int wl_num_cus() { return wl_emu_cus_v; }
inline const char* wl_last_kernel_ptr = "";
inline const char* wl_last_kernel_name() { return wl_last_kernel_ptr; }
inline long long wl_last_grid_v = 0;
inline long long wl_last_grid_value() { return wl_last_grid_v; }
inline const char* wl_kernel_log_buf[32] = {};
inline long long wl_kernel_log_n = 0;
inline long long wl_launch_count_value() { return wl_kernel_log_n; }
inline const char* wl_kernel_history_name(int back) {
    if (back < 0 || back >= 32 || back >= wl_kernel_log_n) return "";
    return wl_kernel_log_buf[(wl_kernel_log_n - 1 - back) & 31];
}

struct WlEmuBlock {
    ucontext_t main;
    std::vector<ucontext_t> fib;
    std::vector<char*> stacks;
    std::vector<char> state;        // 0 ready, 1 at barrier, 2 at shuffle, 3 done
    std::vector<float> shfl_in, shfl_out;
    std::vector<int> shfl_src;      // source lane (0..63) of a pending shuffle, -1 = the lane below (wl_shfl_up1)
    struct Dma { char* dst; const char* src; int len; };
    std::vector<std::vector<Dma> > dma;   // per lane: asynchronous global->LDS copies not yet released by wl_wait_vm
    int cur;
    WlEmuBlock() : cur(0) {}
    ~WlEmuBlock() { for (size_t i = 0; i < stacks.size(); ++i) free(stacks[i]); }
};

inline thread_local WlEmuBlock* wl_emu_cur_block = nullptr;

inline void wl_emu_sync(void* arg) {
    WlEmuBlock* b = (WlEmuBlock*)arg;
    b->state[b->cur] = 1;
    swapcontext(&b->fib[b->cur], &b->main);
}

inline float wl_emu_shuffle(float v, int src) {
    WlEmuBlock* b = wl_emu_cur_block;
    const int me = b->cur;
    b->shfl_in[me] = v;
    b->shfl_src[me] = src;
    b->state[me] = 2;
    swapcontext(&b->fib[me], &b->main);
    return b->shfl_out[me];
}
inline float wl_shfl_up1(float v) { return wl_emu_shuffle(v, -1); }
inline float wl_shfl(float v, int src_lane) { return wl_emu_shuffle(v, src_lane & 63); }

// LDS-DMA: the copy lands only when the issuing lane's wl_wait_vm<N> releases it (oldest first)
inline void wl_emu_dma(const WlCtx& ctx, unsigned lds_off, const void* gsrc, bool lane_on, int len) {
    WlEmuBlock* b = wl_emu_cur_block;
    WlEmuBlock::Dma d;
    d.dst = lane_on ? ctx.smem + lds_off + len * (ctx.tid & 63) : nullptr;   // off lanes keep the per-wave count
    d.src = (const char*)gsrc;
    d.len = len;
    b->dma[b->cur].push_back(d);
}
inline void wl_dma16(const WlCtx& ctx, unsigned lds_off, const void* gsrc, bool lane_on) { wl_emu_dma(ctx, lds_off, gsrc, lane_on, 16); }
inline void wl_dma4(const WlCtx& ctx, unsigned lds_off, const void* gsrc, bool lane_on) { wl_emu_dma(ctx, lds_off, gsrc, lane_on, 4); }
// The counter is per WAVE: the wait is a wave-level rendezvous (all lanes have issued the same loads; all their
// copies have landed before any lane goes on), like the lockstep execution of the hardware.
inline void wl_emu_wait_vm(int n) {
    WlEmuBlock* b = wl_emu_cur_block;
    wl_emu_shuffle(0.f, b->cur & 63);
    std::vector<WlEmuBlock::Dma>& q = b->dma[b->cur];
    while ((int)q.size() > n) {
        if (q.front().dst) memcpy(q.front().dst, q.front().src, q.front().len);
        q.erase(q.begin());
    }
    wl_emu_shuffle(0.f, b->cur & 63);
}

template <typename K>
struct WlEmuJob {
    const typename K::Args* args;
    WlEmuBlock* blk;
    int64_t bid;
    char* smem;
};

template <typename K>
inline void wl_emu_entry(unsigned lo, unsigned hi) {
    WlEmuJob<K>* job = (WlEmuJob<K>*)(((uintptr_t)hi << 32) | (uintptr_t)lo);
    WlEmuBlock* b = job->blk;
    WlCtx ctx;
    ctx.tid = b->cur;
    ctx.nthreads = K::kThreads;
    ctx.bid = job->bid;
    ctx.smem = job->smem;
    ctx.sync_fn = wl_emu_sync;
    ctx.sync_arg = b;
    ctx.lds_base = 0;
    K::run(*job->args, ctx);
    if (!b->dma[ctx.tid].empty()) abort();   // a wave ended with LDS-DMA loads in flight (they would land in another workgroup's LDS)
    b->state[ctx.tid] = 3;
    // returning follows uc_link back to the scheduler
}

template <typename K>
static int wl_launch_named(const typename K::Args& a, int64_t nblocks, size_t lds, void* /*stream*/, const char* name, bool primary) {
    if (nblocks <= 0) return 0;
    if (lds > 160 * 1024) return -2;
    if (primary) { wl_last_kernel_ptr = name; wl_last_grid_v = nblocks; }
    wl_kernel_log_buf[wl_kernel_log_n++ & 31] = name;
    const int nt = K::kThreads;
    const size_t kStack = 256 * 1024;
#pragma omp parallel
    {
        WlEmuBlock blk;
        blk.fib.resize(nt);
        blk.state.resize(nt);
        blk.stacks.resize(nt);
        blk.shfl_in.resize(nt);
        blk.shfl_out.resize(nt);
        blk.shfl_src.resize(nt);
        blk.dma.resize(nt);
        for (int i = 0; i < nt; ++i) blk.stacks[i] = (char*)malloc(kStack);
        char* smem = (char*)aligned_alloc(64, ((lds + 63) / 64 + 1) * 64);
        WlEmuJob<K> job;
        job.args = &a;
        job.blk = &blk;
        job.smem = smem;
        wl_emu_cur_block = &blk;
#pragma omp for schedule(dynamic, 1)
        for (int64_t bid = 0; bid < nblocks; ++bid) {
            memset(smem, 0xFF, lds);   // NaN poison
            job.bid = bid;
            for (int i = 0; i < nt; ++i) {
                blk.state[i] = 0;
                getcontext(&blk.fib[i]);
                blk.fib[i].uc_stack.ss_sp = blk.stacks[i];
                blk.fib[i].uc_stack.ss_size = kStack;
                blk.fib[i].uc_link = &blk.main;
                uintptr_t p = (uintptr_t)&job;
                makecontext(&blk.fib[i], (void (*)())wl_emu_entry<K>, 2, (unsigned)(p & 0xffffffffu),
                            (unsigned)(p >> 32));
            }
            int phase = 0;
            for (;;) {
                // run every ready fibre to its next yield
                bool ran = false;
                for (int s = 0; s < nt; ++s) {
                    const int i = (phase & 1) ? nt - 1 - s : s;
                    if (blk.state[i] != 0) continue;
                    blk.cur = i;
                    swapcontext(&blk.main, &blk.fib[i]);
                    ran = true;
                }
                // resolve wave shuffles: all live lanes of the wave must have arrived
                bool resolved = false;
                for (int w0 = 0; w0 < nt; w0 += 64) {
                    bool any = false, all = true;
                    for (int i = w0; i < w0 + 64 && i < nt; ++i) {
                        if (blk.state[i] == 2) any = true;
                        else if (blk.state[i] != 3) all = false;
                    }
                    if (any && all) {
                        for (int i = w0; i < w0 + 64 && i < nt; ++i)
                            blk.shfl_out[i] = blk.shfl_src[i] < 0 ? ((i > w0) ? blk.shfl_in[i - 1] : blk.shfl_in[i])
                                                                  : blk.shfl_in[w0 + blk.shfl_src[i]];
                        for (int i = w0; i < w0 + 64 && i < nt; ++i)
                            if (blk.state[i] == 2) blk.state[i] = 0;
                        resolved = true;
                    }
                }
                if (resolved) continue;
                bool any_ready = false, any_barrier = false, any_shfl = false;
                for (int i = 0; i < nt; ++i) {
                    any_ready |= blk.state[i] == 0;
                    any_barrier |= blk.state[i] == 1;
                    any_shfl |= blk.state[i] == 2;
                }
                if (any_ready) continue;
                if (any_shfl) abort();   // a wave is stuck at a shuffle while others sit at a barrier: divergent shuffle
                if (!any_barrier) break; // everyone done
                for (int i = 0; i < nt; ++i) if (blk.state[i] == 1) blk.state[i] = 0;
                ++phase;
                (void)ran;
            }
        }
        wl_emu_cur_block = nullptr;
        free(smem);
    }
    return 0;
}
template <typename K>
static int wl_launch(const typename K::Args& a, int64_t nblocks, size_t lds, void* stream) {
    return wl_launch_named<K>(a, nblocks, lds, stream, __PRETTY_FUNCTION__, true);
}
// (see wl_backend_hip.h: the armed fallback of a hinted launch)
template <typename K>
static int wl_launch_armed(const typename K::Args& a, int64_t nblocks, size_t lds, void* stream) {
    return wl_launch_named<K>(a, nblocks, lds, stream, __PRETTY_FUNCTION__, false);
}
// A helper launch in front of the kernel the engine chose (WlTapPrep: one thread that examines the filter banks): in
// wl_kernel_history under this function's name, not in wl_last_kernel.
template <typename K>
static int wl_launch_aux(const typename K::Args& a, int64_t nblocks, size_t lds, void* stream) {
    return wl_launch_named<K>(a, nblocks, lds, stream, __PRETTY_FUNCTION__, false);
}
