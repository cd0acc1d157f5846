// Access-pattern ceiling of the level-1 synthesis launch: per 32x64 output tile, 4 bands x 20 rows x 36 dwords loaded
// from K-pitch band planes, 32 rows x 32 pairs stored to a 512-pitch plane; no LDS / arithmetic.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int TH, int TW>
__global__ void __launch_bounds__(256, 4) k_syn(const float* __restrict__ ll, const float* __restrict__ hs, float* __restrict__ y,
                                                int K, int OH, int OW, int tiles_x, int tiles_y, int variant) {
    constexpr int NKR = TH / 2 + 4, NKC = TW / 2 + 4;
    constexpr int NIT = (NKR * NKC + 255) / 256;
    const int tid = threadIdx.x;
    const int tiles = tiles_x * tiles_y;
    const int plane = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const int ty = tile / tiles_x, tx = tile % tiles_x;
    const size_t bplane = (size_t)K * K;
    const float* llp = ll + plane * bplane;
    const float* hp = hs + plane * 3 * bplane;
    const int kr0 = ty * (TH / 2), kc0 = tx * (TW / 2);
    float acc = 0.f;
    float v[NIT][4];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int f = tid + it * 256;
        v[it][0] = v[it][1] = v[it][2] = v[it][3] = 0.f;
        if (f < NKR * NKC) {
            const int i = f / NKC, j = f % NKC;
            const int r = kr0 + i, c = kc0 + j;
            if (r < K && c < K) {
                const size_t o = (size_t)r * K + c;
                v[it][0] = llp[o];
                if (!(variant & 1)) { v[it][1] = hp[o]; v[it][2] = hp[bplane + o]; v[it][3] = hp[2 * bplane + o]; }
            }
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) acc += v[it][0] + v[it][1] + v[it][2] + v[it][3];
    if (variant & 2) { if (acc == 123.f) y[0] = acc; return; }
    float* yp = y + (size_t)plane * OH * OW;
    for (int f = tid; f < TH * (TW / 2); f += 256) {
        const int i = f / (TW / 2), q = f % (TW / 2);
        const int n = ty * TH + i, w0 = tx * TW + 2 * q;
        if (n < OH && w0 + 1 < OW) *reinterpret_cast<float2*>(yp + (size_t)n * OW + w0) = make_float2(acc, acc);
    }
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
    const int planes = 384, OH = 512, OW = 512;
    for (int K : {259, 256}) {
        const size_t nb = (size_t)planes * 4 * K * K, ny = (size_t)planes * OH * OW;
        float *bands, *y;
        CK(hipMalloc(&bands, nb * 4)); CK(hipMalloc(&y, ny * 4));
        CK(hipMemset(bands, 0, nb * 4));
        const float* ll = bands; const float* hs = bands + (size_t)planes * K * K;
        for (int variant : {0, 2, 1}) {
            float t = timeit([&] { hipLaunchKernelGGL((k_syn<32, 64>), dim3(planes * 16 * 8), dim3(256), 0, 0, ll, hs, y, K, OH, OW, 8, 16, variant); });
            printf("K=%d 32x64 variant=%d (1: ll only, 2: no stores): %.4f ms  %.0f GB/s\n", K, variant, t,
                   ((variant & 1 ? nb / 4 : nb) + (variant & 2 ? 0 : ny)) * 4.0 / t / 1e6);
            t = timeit([&] { hipLaunchKernelGGL((k_syn<16, 128>), dim3(planes * 32 * 4), dim3(256), 0, 0, ll, hs, y, K, OH, OW, 4, 32, variant); });
            printf("K=%d 16x128 variant=%d: %.4f ms  %.0f GB/s\n", K, variant, t, ((variant & 1 ? nb / 4 : nb) + (variant & 2 ? 0 : ny)) * 4.0 / t / 1e6);
        }
        CK(hipFree(bands)); CK(hipFree(y));
    }
    return 0;
}
