// Micro-benchmark (GPU box): the conv kernels' MFMA loop (mma_taps: weights
// L2 -> VGPR, activations LDS -> VGPR, software pipelined) in isolation, at one
// and two waves per SIMD. Separates "the loop cannot feed the matrix pipe"
// from "the phases around it leave the pipe idle".
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-honor-nans -Iinclude \
//       scripts/micro/mma_loop.hip -o promonet_amd/lib/mma_loop
#include "../../promonet_amd/csrc/pm_conv.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int K, int WM, int WN, int NTW, int G = 4>
__global__ __launch_bounds__(WM * WN * 64) void loop_kernel(
    const half8* __restrict__ w, const _Float16* __restrict__ fill,
    float* sink, int reps, int dilation) {
    typedef ElemF16 ET;
    constexpr int C = 128, CH = 64, KC = CH / 16, NCH = C / CH, MTW = (C / 32) / WM;
    constexpr int S = CH * 2 + 16;
    constexpr int ROWS = WN * NTW * 32 + (K - 1) * 5;
    constexpr int W_CHUNK = K * KC * 64, W_MT_STRIDE = NCH * W_CHUNK;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < ROWS * S / 2; i += blockDim.x)
        reinterpret_cast<_Float16*>(smem)[i] = fill[i & 4095];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int ln = lane & 31, lh = lane >> 5;
    floatx16 acc[MTW][NTW];
    for (int mt = 0; mt < MTW; ++mt)
        for (int nt = 0; nt < NTW; ++nt)
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    const half8* wp = w + (size_t)wm * MTW * W_MT_STRIDE + lane;
    half8 afirst[G][MTW];
    load_a_group<ET, MTW, G>(afirst, wp, W_MT_STRIDE);
    const char* bptr = smem + (wn * NTW * 32 + ln) * S + lh * 16;
#pragma unroll 1
    for (int r = 0; r < reps; ++r)
#pragma unroll 1
        for (int c = 0; c < NCH; ++c)
            mma_taps<ET, K, KC, MTW, NTW, G, S>(
                acc, bptr, dilation * S, wp + (size_t)c * W_CHUNK, W_MT_STRIDE,
                afirst, wp + (size_t)((c + 1) % NCH) * W_CHUNK);
    float s = 0.f;
    for (int mt = 0; mt < MTW; ++mt)
        for (int nt = 0; nt < NTW; ++nt)
            for (int r = 0; r < 16; ++r) s += acc[mt][nt][r];
    if (s == 12345.678f) sink[0] = s;
}

template <int K, int WM, int WN, int NTW, int G = 4>
static void run(const char* name, const half8* w, const _Float16* fill,
                float* sink, int wgs_per_cu, int reps) {
    constexpr int S = 64 * 2 + 16;
    constexpr int smem = (WN * NTW * 32 + (K - 1) * 5) * S;
    auto kern = loop_kernel<K, WM, WN, NTW, G>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                        hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu;
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * WN * 64), smem, 0, w,
                           fill, sink, reps, 5);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double mfmas = (double)grid * WM * WN * reps * 2 * K * 4 * NTW * (4 / WM);
    printf("%-34s %d WG/CU (%2d waves/CU): %7.2f ms  %6.0f TFLOP/s\n", name,
           wgs_per_cu, wgs_per_cu * WM * WN, ms, mfmas * 32768.0 / ms / 1e9);
}

int main() {
    const size_t wn = (size_t)4 * 2 * 11 * 4 * 64;   // frags for C=128 k=11
    std::vector<_Float16> hw(wn * 8), hf(4096);
    srand(1);
    for (auto& v : hw) v = (_Float16)(((float)rand() / RAND_MAX * 2.f - 1.f) * 0.05f);
    for (auto& v : hf) v = (_Float16)((float)rand() / RAND_MAX * 2.f - 1.f);
    half8* w; _Float16* fill; float* sink;
    hipMalloc(&w, hw.size() * 2); hipMalloc(&fill, hf.size() * 2); hipMalloc(&sink, 4);
    hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(fill, hf.data(), hf.size() * 2, hipMemcpyHostToDevice);
    run<11, 4, 2, 4>("k11 4x2 waves, wave 32x128", w, fill, sink, 1, 2000);
    run<11, 2, 4, 2, 2>("k11 2x4 waves, wave 64x64 G2", w, fill, sink, 1, 2000);
    run<11, 2, 4, 2, 4>("k11 2x4 waves, wave 64x64 G4", w, fill, sink, 1, 2000);
    run<11, 2, 2, 4, 2>("k11 2x2 waves, wave 64x128 G2", w, fill, sink, 1, 2000);
    run<11, 2, 2, 4, 2>("k11 2x2 waves, wave 64x128 G2", w, fill, sink, 2, 2000);
    run<11, 2, 4, 4, 2>("k11 2x4 waves, wave 64x128 G2", w, fill, sink, 1, 1000);
    run<11, 1, 8, 2, 2>("k11 1x8 waves, wave 128x64 G2", w, fill, sink, 1, 1000);
    return 0;
}
