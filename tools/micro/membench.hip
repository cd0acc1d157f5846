// Memory-system ceilings on the box the kernels run on, with the footprint of the benchmark's forward launch (384
// planes of 512x512 floats in, 384 x 4 x 259 x 259 floats out): copy / read / write kernels in the access forms the
// engine uses (16-byte vectors, LDS-DMA loads, non-temporal variants).  Prints ONE JSON object; bench.py runs the
// prebuilt binary (tools/micro/bin/membench, built by __graft_entry__.build()) and quotes it next to the roofline.
//   hipcc --offload-arch=gfx950 -O3 membench.hip -o bin/membench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("{\"error\": \"%s line %d\"}\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float vf4 __attribute__((ext_vector_type(4)));

template <int NT>
__global__ void __launch_bounds__(256) k_copy4(const vf4* __restrict__ in, vf4* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        vf4 v = (NT & 1) ? __builtin_nontemporal_load(in + i) : in[i];
        if (NT & 2) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}
template <int U, int NT>
__global__ void __launch_bounds__(256) k_copy4u(const vf4* __restrict__ in, vf4* __restrict__ out, size_t n) {
    size_t base = (size_t)blockIdx.x * blockDim.x * U + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x * U;
    for (; base < n; base += stride) {
        vf4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { size_t i = base + (size_t)u * blockDim.x; if (i < n) v[u] = (NT & 1) ? __builtin_nontemporal_load(in + i) : in[i]; }
#pragma unroll
        for (int u = 0; u < U; ++u) { size_t i = base + (size_t)u * blockDim.x; if (i < n) { if (NT & 2) __builtin_nontemporal_store(v[u], out + i); else out[i] = v[u]; } }
    }
}
__global__ void __launch_bounds__(256) k_read4(const vf4* __restrict__ in, float* sink, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (; i < n; i += stride) { vf4 v = in[i]; acc += v.x; }
    if (acc == 12345.678f) sink[0] = acc;
}
template <int NT>
__global__ void __launch_bounds__(256) k_write4(vf4* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    vf4 v = {1.f, 2.f, 3.f, 4.f};
    for (; i < n; i += stride) { if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v; }
}
// 4-byte stores to rows of 259 floats (the band rows of the benchmark: 1036 bytes, never cache-line aligned)
__global__ void __launch_bounds__(256) k_write1_rows(float* __restrict__ out, size_t rows, int rowlen) {
    for (size_t r = blockIdx.x; r < rows; r += gridDim.x)
        for (int c = threadIdx.x; c < rowlen; c += blockDim.x) out[r * rowlen + c] = 1.f;
}
// the engine's load path: global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction), LDS -> VGPR,
// 16-byte global stores.  Every wave owns an 8 KiB ring; 8 loads in flight per wave.
template <int NT>
__global__ void __launch_bounds__(256) k_copy_ldsdma(const char* __restrict__ in, vf4* __restrict__ out, size_t nchunks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned ring = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 8192;
    const size_t wid = (size_t)blockIdx.x * 4 + wave, nw = (size_t)gridDim.x * 4;
    for (size_t c0 = wid * 8; c0 < nchunks; c0 += nw * 8) {          // 8 chunks of 1 KiB per trip
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned m0 = __builtin_amdgcn_readfirstlane(ring + u * 1024);
            const char* src = in + (c0 + u) * 1024 + lane * 16;
            if (c0 + u < nchunks) {
                if (NT) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt" : : "s"(m0), "v"(src) : "memory", "m0");
                else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(m0), "v"(src) : "memory", "m0");
            }
        }
        __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));      // vmcnt(0)
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (c0 + u < nchunks) {
                vf4 v = *reinterpret_cast<vf4*>(smem + wave * 8192 + u * 1024 + lane * 16);
                if (NT) __builtin_nontemporal_store(v, out + (c0 + u) * 64 + lane); else out[(c0 + u) * 64 + lane] = v;
            }
        __builtin_amdgcn_s_waitcnt(0xc07f);                         // lgkmcnt(0): ring reads done before it is refilled
    }
}

template <typename F>
static float timeit(F f, int n = 30) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < n; ++i) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / n;
}

int main() {
    const size_t nin = (size_t)384 * 512 * 512, nout = (size_t)384 * 4 * 259 * 259;   // floats
    float *in, *out, *sink;
    CK(hipMalloc(&in, nin * 4)); CK(hipMalloc(&out, (nin > nout ? nin : nout) * 4)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(in, 0, nin * 4)); CK(hipMemset(out, 0, (nin > nout ? nin : nout) * 4));
    const size_t n4 = nin / 4;
    const int grid = 8192;
    // spin the clocks up
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_copy4<0>, dim3(grid), dim3(256), 0, 0, (const vf4*)in, (vf4*)out, n4);
    CK(hipDeviceSynchronize());
    printf("{\"footprint_mb_in\": %.1f", nin * 4 / 1e6);
    float t;
#define COPY(NAME, ...) t = timeit([&] { __VA_ARGS__; }); printf(", \"%s_gbs\": %.0f", NAME, 2.0 * nin * 4 / t / 1e6);
    COPY("copy_float4", hipLaunchKernelGGL(k_copy4<0>, dim3(grid), dim3(256), 0, 0, (const vf4*)in, (vf4*)out, n4))
    COPY("copy_float4_nt_store", hipLaunchKernelGGL(k_copy4<2>, dim3(grid), dim3(256), 0, 0, (const vf4*)in, (vf4*)out, n4))
    COPY("copy_float4_nt_both", hipLaunchKernelGGL(k_copy4<3>, dim3(grid), dim3(256), 0, 0, (const vf4*)in, (vf4*)out, n4))
    COPY("copy_float4x4", hipLaunchKernelGGL((k_copy4u<4, 0>), dim3(2048), dim3(256), 0, 0, (const vf4*)in, (vf4*)out, n4))
    COPY("copy_float4x4_nt_both", hipLaunchKernelGGL((k_copy4u<4, 3>), dim3(2048), dim3(256), 0, 0, (const vf4*)in, (vf4*)out, n4))
    COPY("copy_ldsdma", hipLaunchKernelGGL(k_copy_ldsdma<0>, dim3(1024), dim3(256), 32768, 0, (const char*)in, (vf4*)out, nin * 4 / 1024))
    COPY("copy_ldsdma_nt", hipLaunchKernelGGL(k_copy_ldsdma<1>, dim3(1024), dim3(256), 32768, 0, (const char*)in, (vf4*)out, nin * 4 / 1024))
    t = timeit([&] { hipLaunchKernelGGL(k_read4, dim3(grid), dim3(256), 0, 0, (const vf4*)in, sink, n4); });
    printf(", \"read_float4_gbs\": %.0f", 1.0 * nin * 4 / t / 1e6);
    t = timeit([&] { hipLaunchKernelGGL(k_write4<0>, dim3(grid), dim3(256), 0, 0, (vf4*)out, nout / 4); });
    printf(", \"write_float4_gbs\": %.0f", 1.0 * nout * 4 / t / 1e6);
    t = timeit([&] { hipLaunchKernelGGL(k_write4<1>, dim3(grid), dim3(256), 0, 0, (vf4*)out, nout / 4); });
    printf(", \"write_float4_nt_gbs\": %.0f", 1.0 * nout * 4 / t / 1e6);
    t = timeit([&] { hipLaunchKernelGGL(k_write1_rows, dim3(grid), dim3(256), 0, 0, out, (size_t)384 * 3 * 259, 259); });
    printf(", \"write_dword_rows_of_259_gbs\": %.0f", 384.0 * 3 * 259 * 259 * 4 / t / 1e6);
    printf("}\n");
    return 0;
}
