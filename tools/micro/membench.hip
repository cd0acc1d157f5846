// Memory-system ceilings on the box the kernels run on: plain copy / read / write kernels with the same footprint
// as the level-1 DWT launch (402 MB in, 412 MB out).  Build: hipcc --offload-arch=gfx950 -O3 membench.hip -o membench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <typename V>
__global__ void k_copy(const V* __restrict__ in, V* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}
template <typename V, int U>
__global__ void k_copy_unroll(const V* __restrict__ in, V* __restrict__ out, size_t n) {
    // each workgroup copies contiguous chunks of U*blockDim vectors, all loads issued before the stores
    size_t base = (size_t)blockIdx.x * blockDim.x * U + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x * U;
    for (; base < n; base += stride) {
        V v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { size_t i = base + (size_t)u * blockDim.x; if (i < n) v[u] = in[i]; }
#pragma unroll
        for (int u = 0; u < U; ++u) { size_t i = base + (size_t)u * blockDim.x; if (i < n) out[i] = v[u]; }
    }
}
template <typename V>
__global__ void k_read(const V* __restrict__ in, float* sink, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (; i < n; i += stride) { V v = in[i]; acc += v.x; }
    if (acc == 12345.678f) sink[0] = acc;
}
template <typename V>
__global__ void k_write(V* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    V v; v.x = 1.f; v.y = 2.f;
    for (; i < n; i += stride) out[i] = v;
}

__global__ void k_read1(const float* __restrict__ in, float* sink, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (; i < n; i += stride) acc += in[i];
    if (acc == 12345.678f) sink[0] = acc;
}
__global__ void k_copy1(const float* __restrict__ in, float* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}

template <typename F>
static float timeit(F f, int n = 20) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < n; ++i) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / n;
}

int main() {
    const size_t nin = (size_t)384 * 512 * 512, nout = (size_t)384 * 4 * 259 * 259;   // floats
    float *in, *out, *sink;
    CK(hipMalloc(&in, nin * 4)); CK(hipMalloc(&out, nout * 4)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(in, 0, nin * 4)); CK(hipMemset(out, 0, nout * 4));
    const size_t n2 = nin / 2, n4 = nin / 4;
    for (int grid : {8192}) {
        float t;
        t = timeit([&] { hipLaunchKernelGGL(k_read1, dim3(grid), dim3(256), 0, 0, (const float*)in, sink, nin); });
        printf("read float1 grid %6d: %.4f ms  %.0f GB/s\n", grid, t, 1.0 * nin * 4 / t / 1e6);
        t = timeit([&] { hipLaunchKernelGGL(k_copy1, dim3(grid), dim3(256), 0, 0, (const float*)in, out, nin); });
        printf("copy float1 grid %6d: %.4f ms  %.0f GB/s (r+w)\n", grid, t, 2.0 * nin * 4 / t / 1e6);
        t = timeit([&] { hipLaunchKernelGGL(k_copy<float2>, dim3(grid), dim3(256), 0, 0, (const float2*)in, (float2*)out, n2); });
        printf("copy float2 grid %6d: %.4f ms  %.0f GB/s (r+w)\n", grid, t, 2.0 * nin * 4 / t / 1e6);
        t = timeit([&] { hipLaunchKernelGGL(k_copy<float4>, dim3(grid), dim3(256), 0, 0, (const float4*)in, (float4*)out, n4); });
        printf("copy float4 grid %6d: %.4f ms  %.0f GB/s (r+w)\n", grid, t, 2.0 * nin * 4 / t / 1e6);
        t = timeit([&] { hipLaunchKernelGGL((k_copy_unroll<float2, 8>), dim3(grid), dim3(256), 0, 0, (const float2*)in, (float2*)out, n2); });
        printf("copy float2x8 grid %6d: %.4f ms  %.0f GB/s (r+w)\n", grid, t, 2.0 * nin * 4 / t / 1e6);
        t = timeit([&] { hipLaunchKernelGGL((k_copy_unroll<float4, 4>), dim3(grid), dim3(256), 0, 0, (const float4*)in, (float4*)out, n4); });
        printf("copy float4x4 grid %6d: %.4f ms  %.0f GB/s (r+w)\n", grid, t, 2.0 * nin * 4 / t / 1e6);
        t = timeit([&] { hipLaunchKernelGGL(k_read<float2>, dim3(grid), dim3(256), 0, 0, (const float2*)in, sink, n2); });
        printf("read float2 grid %6d: %.4f ms  %.0f GB/s\n", grid, t, 1.0 * nin * 4 / t / 1e6);
        t = timeit([&] { hipLaunchKernelGGL(k_read<float4>, dim3(grid), dim3(256), 0, 0, (const float4*)in, sink, n4); });
        printf("read float4 grid %6d: %.4f ms  %.0f GB/s\n", grid, t, 1.0 * nin * 4 / t / 1e6);
        t = timeit([&] { hipLaunchKernelGGL(k_write<float2>, dim3(grid), dim3(256), 0, 0, (float2*)out, nout / 2); });
        printf("write float2 grid %6d: %.4f ms  %.0f GB/s\n", grid, t, 1.0 * nout * 4 / t / 1e6);
    }
    return 0;
}
