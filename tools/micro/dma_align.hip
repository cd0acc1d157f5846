// Does global_load_lds_dwordx4 accept global addresses that are only 4-byte aligned?  (band planes of odd size start
// on 4-byte boundaries.)  Also times aligned vs misaligned LDS-DMA streaming.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void __launch_bounds__(64) k(const float* x, float* out, int shift) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const float* g = x + shift + threadIdx.x * 4;
    const unsigned m0 = __builtin_amdgcn_readfirstlane(base);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(m0), "v"(g) : "memory", "m0");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const float* S = (const float*)smem;
    for (int i = 0; i < 4; ++i) out[threadIdx.x * 4 + i] = S[threadIdx.x * 4 + i];
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int n = 1024;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *dx, *dy; hipMalloc(&dx, n * 4); hipMalloc(&dy, 256 * 4);
    hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int shift = 0; shift < 4; ++shift) {
        hipMemset(dy, 0, 256 * 4);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, dx, dy, shift);
        hipError_t e = hipDeviceSynchronize();
        std::vector<float> o(256);
        hipMemcpy(o.data(), dy, 256 * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 256; ++i) bad += o[i] != (float)(i + shift);
        printf("shift %d floats: err=%d mismatches=%d  first: %g %g %g %g %g\n", shift, (int)e, bad, o[0], o[1], o[2], o[3], o[4]);
    }
    return 0;
}
