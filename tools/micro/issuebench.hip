// Instruction-issue microbenchmark for gfx950: cycles per instruction of one wave, and of 1/2/4 waves per SIMD,
// for the instruction kinds the streaming wavelet kernel is made of.  hipcc --offload-arch=gfx950 -O3 -o issuebench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float v2 __attribute__((ext_vector_type(2)));
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

template <int KIND>
__global__ void __launch_bounds__(1024) k(unsigned long long* out, int iters, float seed) {
    v2 a = {seed, seed}, b = {seed + 1, seed}, c = {seed + 2, seed}, d = {seed + 3, seed};
    v2 t = {1.0001f * seed, 0.9999f * seed};
    float f0 = seed, f1 = seed + 1, f2 = seed + 2, f3 = seed + 3;
    int s0 = __builtin_amdgcn_readfirstlane((int)seed), s1 = __builtin_amdgcn_readfirstlane((int)seed + 1);
    v2 ts = {__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, t.x))), __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, t.y)))};
    __shared__ float lds[4096];
    lds[threadIdx.x] = seed; lds[threadIdx.x + 1024] = seed;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {   // 4 independent v_pk_fma_f32 chains
            REP16(asm volatile("v_pk_fma_f32 %0, %4, %0, %0\n v_pk_fma_f32 %1, %4, %1, %1\n v_pk_fma_f32 %2, %4, %2, %2\n v_pk_fma_f32 %3, %4, %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(t));)
        } else if (KIND == 1) {   // 4 independent v_fma_f32 chains
            REP16(asm volatile("v_fma_f32 %0, %4, %0, %0\n v_fma_f32 %1, %4, %1, %1\n v_fma_f32 %2, %4, %2, %2\n v_fma_f32 %3, %4, %3, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(t.x));)
        } else if (KIND == 2) {   // 1 dependent v_pk_fma_f32 chain
            REP64(asm volatile("v_pk_fma_f32 %0, %1, %0, %0" : "+v"(a) : "v"(t));)
        } else if (KIND == 3) {   // SALU chain
            REP64(asm volatile("s_add_i32 s20, s20, s21" ::: "s20", "s21");)
        } else if (KIND == 4) {   // alternating VALU (pk, 4 chains) and SALU
            REP16(asm volatile("v_pk_fma_f32 %0, %4, %0, %0\n s_add_i32 s20, s20, s21\n v_pk_fma_f32 %1, %4, %1, %1\n s_add_i32 s20, s20, s21\n v_pk_fma_f32 %2, %4, %2, %2\n s_add_i32 s20, s20, s21\n v_pk_fma_f32 %3, %4, %3, %3\n s_add_i32 s20, s20, s21" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(t) : "s20", "s21");)
        } else if (KIND == 5) {   // ds_read_b64, 8-byte lane stride, results unused (throughput)
            unsigned addr = (threadIdx.x & 63) * 8;
            v2 r0, r1, r2, r3;
            REP16(asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:8\n ds_read_b64 %2, %4 offset:16\n ds_read_b64 %3, %4 offset:24\n s_waitcnt lgkmcnt(0)" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(addr));)
            a += r0 + r1 + r2 + r3;
        } else if (KIND == 6) {   // 4 independent v_pk_fma_f32 chains with an SGPR-pair operand (as in the kernel)
            REP16(asm volatile("v_pk_fma_f32 %0, %4, %0, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %4, %1, %1 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %2, %4, %2, %2 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n v_pk_fma_f32 %3, %4, %3, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(ts));)
        } else if (KIND == 7) {   // v_mov_b64
            REP16(asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
        } else if (KIND == 8) {   // taken scalar branches
            REP16(asm volatile("s_branch 1f\n s_nop 0\n1: s_branch 2f\n s_nop 0\n2: s_branch 3f\n s_nop 0\n3: s_branch 4f\n s_nop 0\n4:" ::: "memory");)
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float sink = a.x + b.x + c.x + d.x + f0 + f1 + f2 + f3 + (float)s0;
    if (sink == 1234.5f) out[1000000] = 1;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND> void run(const char* name, unsigned long long* d, int instr_per_iter) {
    for (int waves : {1, 4, 8, 16}) {
        const int iters = 200, blocks = 256;
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64 * waves), 0, 0, d, iters, 1.0f);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks * 16);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        double sum = 0; int n = 0;
        for (int b = 0; b < blocks; ++b) for (int w = 0; w < waves; ++w) { sum += (double)h[b * 16 + w]; ++n; }
        const double cyc_per_instr_wave = sum / n / (iters * (double)instr_per_iter);
        printf("%-34s waves/CU %2d (%.2f per SIMD): %6.2f cycles per instruction per wave -> %5.2f cycles per instruction per SIMD\n",
               name, waves, waves / 4.0, cyc_per_instr_wave, cyc_per_instr_wave / (waves < 4 ? 1 : waves / 4.0));
    }
}
int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int kind = argc > 1 ? atoi(argv[1]) : 0;
    unsigned long long* d; hipMalloc(&d, 8 * 1000008);
    switch (kind) {
        case 0: run<0>("v_pk_fma_f32 x4 chains (vgpr)", d, 64); break;
        case 6: run<6>("v_pk_fma_f32 x4 chains (sgpr,opsel)", d, 64); break;
        case 1: run<1>("v_fma_f32 x4 chains", d, 64); break;
        case 2: run<2>("v_pk_fma_f32 1 dependent chain", d, 64); break;
        case 3: run<3>("s_add_i32 dependent chain", d, 64); break;
        case 4: run<4>("pk_fma + s_add alternating", d, 128); break;
        case 5: run<5>("ds_read_b64 x4 + wait", d, 80); break;
        case 7: run<7>("v_mov_b64", d, 64); break;
        case 8: run<8>("taken s_branch (+1 skipped nop)", d, 64); break;
    }
    return 0;
}
