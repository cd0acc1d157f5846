// What does ISSUING an LDS-DMA load cost a wave?  One wave per workgroup issues K global_load_lds instructions back to
// back (M0 rewritten before each, as the streaming kernels do), then waits for all of them; s_memtime around the issue
// phase and around the wait.  Kinds: 0 = dwordx4 all 64 lanes, 1 = dwordx4 33 lanes, 2 = dword 1 lane, 3 = dwordx4 64
// lanes with M0 left alone.   hipcc --offload-arch=gfx950 -O3 tools/micro/dma_issue.hip -o /tmp/dma_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
template <int KIND>
__global__ void __launch_bounds__(64) k(const char* x, long long* out, int K, int rounds, size_t wg_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int lane = threadIdx.x;
    const char* g = x + (size_t)blockIdx.x * wg_stride + lane * 16;
    long long t_issue = 0, t_wait = 0;
    for (int r = 0; r < rounds; ++r) {
        const long long t0 = __builtin_readcyclecounter();
        for (int i = 0; i < K; ++i) {
            const unsigned m0 = __builtin_amdgcn_readfirstlane(base + (i & 15) * 1024);
            const char* p = g + (size_t)(r * K + i) * 1024;
            if (KIND == 0) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(m0), "v"(p) : "memory", "m0");
            if (KIND == 1) { if (lane < 33) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(m0), "v"(p) : "memory", "m0"); }
            if (KIND == 2) { if (lane < 1) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" : : "s"(m0), "v"(p) : "memory", "m0"); }
            if (KIND == 3) asm volatile("global_load_lds_dwordx4 %0, off" : : "v"(p) : "memory");
        }
        const long long t1 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long t2 = __builtin_readcyclecounter();
        t_issue += t1 - t0; t_wait += t2 - t1;
    }
    if (lane == 0) { out[2 * blockIdx.x] = t_issue; out[2 * blockIdx.x + 1] = t_wait; }
}
int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int rounds = 64;
    const size_t wg_stride = (size_t)rounds * 64 * 1024 + 4096;
    const int maxwg = 1024;
    char* dx; long long* dout;
    hipMalloc(&dx, wg_stride * maxwg + (1 << 20)); hipMalloc(&dout, maxwg * 16);
    hipMemset(dx, 0, wg_stride * maxwg);
    for (int wgs : {1, 256, 1024})
        for (int kind = 0; kind < 4; ++kind)
            for (int K : {1, 4, 8, 16, 32}) {
                hipMemset(dout, 0, maxwg * 16);
                void (*f)(const char*, long long*, int, int, size_t) = kind == 0 ? k<0> : kind == 1 ? k<1> : kind == 2 ? k<2> : k<3>;
                hipLaunchKernelGGL(f, dim3(wgs), dim3(64), 16384, 0, dx, dout, K, rounds, wg_stride);
                hipError_t e = hipDeviceSynchronize();
                std::vector<long long> o(2 * wgs);
                hipMemcpy(o.data(), dout, 16 * wgs, hipMemcpyDeviceToHost);
                double ti = 0, tw = 0;
                for (int i = 0; i < wgs; ++i) { ti += o[2 * i]; tw += o[2 * i + 1]; }
                printf("wgs %4d kind %d K %2d err %d: issue %.0f cyc/instr, wait %.0f cyc/round (100 MHz counter ticks x?)\n", wgs, kind, K, (int)e,
                       ti / wgs / rounds / K, tw / wgs / rounds);
            }
    return 0;
}
