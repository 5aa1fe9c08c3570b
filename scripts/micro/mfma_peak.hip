// Micro-benchmark (GPU box): sustained v_mfma_f32_32x32x16_f16 rate with
// register-resident operands, random vs zero data. Establishes the power-capped
// MFMA ceiling the conv kernels are measured against.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_peak.hip -o promonet_amd/lib/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(
    const half8* __restrict__ src, float* __restrict__ sink, int iters) {
    half8 a[4], b[NACC];
    for (int i = 0; i < 4; ++i) a[i] = src[(threadIdx.x + 256 * i) & 4095];
    for (int i = 0; i < NACC; ++i) b[i] = src[(threadIdx.x + 256 * (i + 4)) & 4095];
    floatx16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                    a[k], b[i], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) sink[0] = s;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    half8* src; float* sink;
    hipMalloc(&src, 4096 * sizeof(half8));
    hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
        std::vector<_Float16> host(4096 * 8);
        srand(1);
        for (auto& v : host) {
            float r = (float)rand() / RAND_MAX * 2.f - 1.f;
            v = (_Float16)(mode == 0 ? 0.f : mode == 1 ? r : r * 0.05f);
        }
        hipMemcpy(src, host.data(), host.size() * 2, hipMemcpyHostToDevice);
        for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
            const int grid = 256 * wgs_per_cu;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                mfma_loop<4><<<grid, 256>>>(src, sink, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double flops = (double)grid * 4 * iters * 16 * 32768.0;
                if (rep == 2)
                    printf("data %s, %d waves/CU: %.2f ms, %.0f TFLOP/s\n",
                           mode == 0 ? "zero" : mode == 1 ? "uniform[-1,1)" : "uniform*0.05",
                           4 * wgs_per_cu, ms, flops / ms / 1e9);
            }
        }
    }
    return 0;
}
