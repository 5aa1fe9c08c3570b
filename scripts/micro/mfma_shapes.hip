// Micro-benchmark (GPU box): sustained rate under the power cap of the four
// 16-bit MFMA shapes with register-resident random operands - is one shape
// cheaper per FLOP than v_mfma_f32_32x32x16 (the conv kernels' instruction)?
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_shapes.hip -o promonet_amd/lib/mfma_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// KIND 0: f16 32x32x16, 1: bf16 32x32x16, 2: f16 16x16x32, 3: bf16 16x16x32
template <int KIND>
__global__ __launch_bounds__(256) void mfma_loop(
    const uint4* __restrict__ src, float* __restrict__ sink, int iters) {
    uint4 a[4], b[4];
    for (int i = 0; i < 4; ++i) a[i] = src[(threadIdx.x + 256 * i) & 4095];
    for (int i = 0; i < 4; ++i) b[i] = src[(threadIdx.x + 256 * (i + 4)) & 4095];
    float s = 0.f;
    if constexpr (KIND < 2) {
        floatx16 acc[4];
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if constexpr (KIND == 0)
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                            __builtin_bit_cast(half8, a[k]),
                            __builtin_bit_cast(half8, b[i]), acc[i], 0, 0, 0);
                    else
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, a[k]),
                            __builtin_bit_cast(bf16x8, b[i]), acc[i], 0, 0, 0);
                }
        }
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else {
        // the same FLOPs per iteration: 16 independent 16x16 accumulators,
        // two passes over them (a 32 x 128 tile = 2 x 8 tiles of 16 x 16)
        floatx4 acc[16];
        for (int i = 0; i < 16; ++i)
            for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if constexpr (KIND == 2)
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                            __builtin_bit_cast(half8, a[2 * k + (i & 1)]),
                            __builtin_bit_cast(half8, b[i >> 2]), acc[i], 0, 0, 0);
                    else
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, a[2 * k + (i & 1)]),
                            __builtin_bit_cast(bf16x8, b[i >> 2]), acc[i], 0, 0, 0);
                }
        }
        for (int i = 0; i < 16; ++i)
            for (int r = 0; r < 4; ++r) s += acc[i][r];
    }
    if (s == 12345.678f) sink[0] = s;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    uint4* src; float* sink;
    hipMalloc(&src, 4096 * sizeof(uint4));
    hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[4] = {"f16 32x32x16", "bf16 32x32x16", "f16 16x16x32",
                            "bf16 16x16x32"};
    for (int kind = 0; kind < 4; ++kind) {
        // random operands of the right type, |x| < 1
        std::vector<unsigned short> host(4096 * 8);
        srand(1);
        for (auto& v : host) {
            const float r = (float)rand() / RAND_MAX * 2.f - 1.f;
            if (kind & 1) {
                unsigned u; __builtin_memcpy(&u, &r, 4); v = (unsigned short)(u >> 16);
            } else {
                _Float16 h = (_Float16)r; __builtin_memcpy(&v, &h, 2);
            }
        }
        hipMemcpy(src, host.data(), host.size() * 2, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            const int grid = 512;
            switch (kind) {
                case 0: mfma_loop<0><<<grid, 256>>>(src, sink, iters); break;
                case 1: mfma_loop<1><<<grid, 256>>>(src, sink, iters); break;
                case 2: mfma_loop<2><<<grid, 256>>>(src, sink, iters); break;
                case 3: mfma_loop<3><<<grid, 256>>>(src, sink, iters); break;
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // per iteration and wave: 16 x 32768 FLOP (32x32x16) = 32 x 16384
            const double flops = (double)grid * 4 * iters * 16 * 32768.0;
            if (rep >= 2)
                printf("%s: %.2f ms, %.0f TFLOP/s\n", names[kind], ms,
                       flops / ms / 1e9);
        }
    }
    return 0;
}
