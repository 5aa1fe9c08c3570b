// Micro-benchmark (GPU box): does running the two waves of a SIMD in
// ANTI-PHASE (one in the MFMA loop while the other does epilogue-like VALU +
// LDS-write work) beat the lock-step phases of the pair kernel?
//   mode 0: lock-step  - every wave: mma, barrier, valu, barrier   (today)
//   mode 1: anti-phase - waves 4-7 start with the valu phase, no barriers
//   mode 2: mma only (no valu phase) - upper bound
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-honor-nans -Iinclude \
//       scripts/micro/phase_overlap.hip -o promonet_amd/lib/phase_overlap
#include "../../promonet_amd/csrc/pm_conv.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int K = 11, WM = 4, WN = 2, NTW = 4, CH = 64, KC = 4, NCH = 2, G = 4;
constexpr int S = CH * 2 + 16;
constexpr int ROWS = WN * NTW * 32 + (K - 1) * 5;
constexpr int W_CHUNK = K * KC * 64, W_MT_STRIDE = NCH * W_CHUNK;
constexpr int SCR = 256 * 272;     // epilogue scratch (like `inter`)

__device__ __forceinline__ void valu_phase(
    floatx16 (&acc)[1][NTW], char* scratch, int lane, int wave, int valu_iters) {
    // epilogue-like: lrelu + convert + LDS write of every accumulator, then a
    // staging-like stream of VALU work; leaves acc data-dependent but bounded
    const int ln = lane & 31, lh = lane >> 5;
    if (valu_iters < 0) {
        // pure VALU: independent fma chains on the accumulators, no LDS
#pragma unroll 1
        for (int it = 0; it < -valu_iters; ++it)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[0][nt][r] = fmaf(acc[0][nt][r], 0.999f, 0.001f);
        return;
    }
#pragma unroll 1
    for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float4 v = pm_lrelu4(acc_quad(acc[0][nt], g4));
                ElemF16::store4(
                    scratch + ((wave & 1) * 128 + nt * 32 + ln) * 272 +
                        ((wave >> 1) * 32 + 8 * g4 + 4 * lh) * 2, v);
                acc[0][nt][4 * g4] = v.x * 0.5f; acc[0][nt][4 * g4 + 1] = v.y * 0.5f;
                acc[0][nt][4 * g4 + 2] = v.z * 0.5f; acc[0][nt][4 * g4 + 3] = v.w * 0.5f;
            }
    }
}

__global__ __launch_bounds__(512) void kern(
    const half8* __restrict__ w, const _Float16* __restrict__ fill, float* sink,
    int reps, int mode, int valu_iters) {
    typedef ElemF16 ET;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < ROWS * S / 2; i += blockDim.x)
        reinterpret_cast<_Float16*>(smem)[i] = fill[i & 4095];
    char* scratch = smem + ROWS * S;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // wave w sits on SIMD w % 4: rows wm = w % 4, column half wn = w / 4, so
    // every SIMD holds one wave of each column half (= phase group)
    const int wm = wave & 3, wn = wave >> 2;
    const int ln = lane & 31, lh = lane >> 5;
    floatx16 acc[1][NTW];
    for (int nt = 0; nt < NTW; ++nt)
        for (int r = 0; r < 16; ++r) acc[0][nt][r] = 0.f;
    const half8* wp = w + (size_t)wm * W_MT_STRIDE + lane;
    half8 afirst[G][1];
    load_a_group<ET, 1, G>(afirst, wp, W_MT_STRIDE);
    const char* bptr = smem + (wn * NTW * 32 + ln) * S + lh * 16;
    // which waves start in the other phase: the SIMD of a wave is not
    // documented, so try every pairing (mode 1: w >= 4, 4: odd w, 5: w & 2)
    const bool shifted = mode == 1 ? wn == 1 : mode == 4 ? (wave & 1) : mode == 5 ? (wave & 2) != 0 : false;
    if (shifted) valu_phase(acc, scratch, lane, wave, valu_iters);
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
#pragma unroll 1
        for (int c = 0; c < NCH && mode != 3; ++c)
            mma_taps<ET, K, KC, 1, NTW, G, S>(
                acc, bptr, 5 * S, wp + (size_t)c * W_CHUNK, W_MT_STRIDE, afirst,
                wp + (size_t)((c + 1) % NCH) * W_CHUNK);
        if (mode == 0) __syncthreads();
        if (mode != 2) valu_phase(acc, scratch, lane, wave, valu_iters);
        if (mode == 0) __syncthreads();
    }
    float s = 0.f;
    for (int nt = 0; nt < NTW; ++nt)
        for (int r = 0; r < 16; ++r) s += acc[0][nt][r];
    if (s == 12345.678f) sink[0] = s;
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 1500;
    const bool zero = argc > 2 && atoi(argv[2]) == 0;   // zero operands: no power cap
    const size_t wn = (size_t)4 * 2 * 11 * 4 * 64;
    std::vector<_Float16> hw(wn * 8), hf(4096);
    srand(1);
    for (auto& v : hw) v = (_Float16)(zero ? 0.f : ((float)(rand() % 2001) / 1000.f - 1.f) * 0.05f);
    for (auto& v : hf) v = (_Float16)(zero ? 0.f : (float)(rand() % 2001) / 1000.f - 1.f);
    half8* w; _Float16* fill; float* sink;
    hipMalloc(&w, hw.size() * 2); hipMalloc(&fill, hf.size() * 2); hipMalloc(&sink, 4);
    hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(fill, hf.data(), hf.size() * 2, hipMemcpyHostToDevice);
    const int smem = ROWS * S + SCR;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                        hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[6] = {"lock-step phases", "anti-phase (waves 4-7)", "mma only", "valu only", "anti-phase (odd waves)", "anti-phase (waves w&2)"};
    for (int valu_iters : {-48}) {
        for (int mode = 0; mode < 6; ++mode) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(kern, dim3(256), dim3(512), smem, 0, w, fill,
                                   sink, reps, mode, valu_iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            const double mfmas = 256.0 * 8 * reps * NCH * K * KC * NTW;
            printf("valu x%d  %-26s %7.2f ms  %6.0f TFLOP/s\n", valu_iters,
                   names[mode], ms, mfmas * 32768.0 / ms / 1e9);
        }
    }
    return 0;
}
