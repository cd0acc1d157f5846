// HIP launch backend (the product).  One generic __global__ wrapper per kernel functor.
#pragma once
#include <hip/hip_runtime.h>
#include "wl_common.h"

#define WL_BACKEND_NAME "hip-gfx950"

template <typename K>
__global__ void __launch_bounds__(K::kThreads, K::kMinWaves) wl_kernel(const typename K::Args a) {
    extern __shared__ __attribute__((aligned(16))) char wl_smem[];
    WlCtx ctx;
    ctx.tid = threadIdx.x;
    ctx.nthreads = K::kThreads;
    ctx.bid = blockIdx.x;
    ctx.smem = wl_smem;
    ctx.lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)wl_smem;
    K::run(a, ctx);
}

// name of the kernel functor launched last by ANY thread of the process (autograd runs backward passes on its own
// threads; a relaxed atomic pointer to a string literal - a diagnostic, not a synchronisation point)
// (wl_last_kernel() of the C ABI: bench.py labels its
// roofline with the dispatch that was actually taken)
#include <atomic>
inline std::atomic<const char*> wl_last_kernel_ptr{""};
static const char* wl_last_kernel_name() { return wl_last_kernel_ptr.load(std::memory_order_relaxed); }
inline std::atomic<long long> wl_last_grid_v{0};   // workgroups of that launch (wl_last_grid() of the C ABI)
static long long wl_last_grid_value() { return wl_last_grid_v.load(std::memory_order_relaxed); }
// ... and the last 32, with a running count of launches: a benchmark names EVERY kernel of a multi-launch transform
inline std::atomic<const char*> wl_kernel_log_buf[32] = {};
inline std::atomic<long long> wl_kernel_log_n{0};
static void wl_kernel_log(const char* name) {
    const long long i = wl_kernel_log_n.fetch_add(1, std::memory_order_relaxed);
    wl_kernel_log_buf[i & 31].store(name, std::memory_order_relaxed);
}
static long long wl_launch_count_value() { return wl_kernel_log_n.load(std::memory_order_relaxed); }
static const char* wl_kernel_history_name(int back) {
    const long long n = wl_kernel_log_n.load(std::memory_order_relaxed);
    if (back < 0 || back >= 32 || back >= n) return "";
    const char* p = wl_kernel_log_buf[(n - 1 - back) & 31].load(std::memory_order_relaxed);
    return p ? p : "";
}

// compute units of the current device (persistent kernels size their grid from it)
static int wl_num_cus() {
    static thread_local int cached_dev = -1, cached = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev != cached_dev) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached = n; cached_dev = dev;
    }
    return cached;
}

template <typename K>
static int wl_launch_named(const typename K::Args& a, int64_t nblocks, size_t lds, void* stream, const char* name, bool primary) {
    if (nblocks <= 0) return 0;
    if (nblocks > 2147483647LL || lds > 160 * 1024) return -2;
    if (primary) { wl_last_kernel_ptr.store(name, std::memory_order_relaxed); wl_last_grid_v.store(nblocks, std::memory_order_relaxed); }
    wl_kernel_log(name);
    if (lds > 48 * 1024) {
        // opt in to large dynamic LDS once per kernel (idempotent, cheap)
        static thread_local unsigned granted = 0;   // bit d: done for device d (the attribute is per device)
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        if (dev >= 32 || !(granted >> dev & 1u)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wl_kernel<K>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return (int)e;
            if (dev < 32) granted |= 1u << dev;
        }
    }
    hipLaunchKernelGGL(wl_kernel<K>, dim3((unsigned)nblocks), dim3(K::kThreads), lds,
                       reinterpret_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}
template <typename K>
static int wl_launch(const typename K::Args& a, int64_t nblocks, size_t lds, void* stream) {
    return wl_launch_named<K>(a, nblocks, lds, stream, __PRETTY_FUNCTION__, true);
}
// The armed fallback of a hinted launch (wl_common.h, tap-relation guards): the two-bank variant queued behind a variant that
// relies on a relation between the filter banks; it returns at once unless the device finds the relation broken.  It is a
// launch like any other in wl_kernel_history (its name carries this function's), but wl_last_kernel keeps naming the variant
// the engine chose.
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
