// Host emulation backend (TEST INFRASTRUCTURE, never loaded by the product package).
// Runs each workgroup as 256 cooperative fibres on one OS thread: a fibre runs until its next
// barrier (WlCtx::sync) or its end, so barrier semantics are exact.  The visiting order of the
// fibres alternates forward / backward between barrier rounds so that a MISSING barrier in a
// kernel shows up as a wrong result instead of passing by luck.  LDS is poisoned with NaNs.
#pragma once
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>
#include <vector>
#include "../../pytorch_wavelets_amd/csrc/wl_common.h"

#define WL_BACKEND_NAME "emu"

struct WlEmuBlock {
    ucontext_t main;
    std::vector<ucontext_t> fib;
    std::vector<char*> stacks;
    std::vector<char> done;
    int cur;
    WlEmuBlock() : cur(0) {}
    ~WlEmuBlock() { for (size_t i = 0; i < stacks.size(); ++i) free(stacks[i]); }
};

static void wl_emu_sync(void* arg) {
    WlEmuBlock* b = (WlEmuBlock*)arg;
    swapcontext(&b->fib[b->cur], &b->main);
}

template <typename K>
struct WlEmuJob {
    const typename K::Args* args;
    WlEmuBlock* blk;
    int64_t bid;
    char* smem;
};

template <typename K>
static void wl_emu_entry(unsigned lo, unsigned hi) {
    WlEmuJob<K>* job = (WlEmuJob<K>*)(((uintptr_t)hi << 32) | (uintptr_t)lo);
    WlEmuBlock* b = job->blk;
    WlCtx ctx;
    ctx.tid = b->cur;
    ctx.nthreads = K::kThreads;
    ctx.bid = job->bid;
    ctx.smem = job->smem;
    ctx.sync_fn = wl_emu_sync;
    ctx.sync_arg = b;
    K::run(*job->args, ctx);
    b->done[ctx.tid] = 1;
    // returning follows uc_link back to the scheduler
}

template <typename K>
static int wl_launch(const typename K::Args& a, int64_t nblocks, size_t lds, void* /*stream*/) {
    if (nblocks <= 0) return 0;
    if (lds > 160 * 1024) return -2;
    const int nt = K::kThreads;
    const size_t kStack = 256 * 1024;
#pragma omp parallel
    {
        WlEmuBlock blk;
        blk.fib.resize(nt);
        blk.done.resize(nt);
        blk.stacks.resize(nt);
        for (int i = 0; i < nt; ++i) blk.stacks[i] = (char*)malloc(kStack);
        char* smem = (char*)aligned_alloc(64, ((lds + 63) / 64 + 1) * 64);
        WlEmuJob<K> job;
        job.args = &a;
        job.blk = &blk;
        job.smem = smem;
#pragma omp for schedule(dynamic, 1)
        for (int64_t bid = 0; bid < nblocks; ++bid) {
            memset(smem, 0xFF, lds);   // NaN poison
            job.bid = bid;
            for (int i = 0; i < nt; ++i) {
                blk.done[i] = 0;
                getcontext(&blk.fib[i]);
                blk.fib[i].uc_stack.ss_sp = blk.stacks[i];
                blk.fib[i].uc_stack.ss_size = kStack;
                blk.fib[i].uc_link = &blk.main;
                uintptr_t p = (uintptr_t)&job;
                makecontext(&blk.fib[i], (void (*)())wl_emu_entry<K>, 2, (unsigned)(p & 0xffffffffu),
                            (unsigned)(p >> 32));
            }
            bool alive = true;
            int round = 0;
            while (alive) {
                alive = false;
                for (int s = 0; s < nt; ++s) {
                    const int i = (round & 1) ? nt - 1 - s : s;
                    if (blk.done[i]) continue;
                    blk.cur = i;
                    swapcontext(&blk.main, &blk.fib[i]);
                    if (!blk.done[i]) alive = true;
                }
                ++round;
            }
        }
        free(smem);
    }
    return 0;
}
