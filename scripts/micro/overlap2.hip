// Micro-benchmark (GPU box), round 3: re-opens the round-1 "a SIMD's time is
// the SUM of the issue costs of its waves" model. Round 1's probe let hipcc
// SLP-pack its VALU phase into v_pk_fma_f32 (which the hardware guide prices
// as an anti-lever beside MFMAs) and used no priorities. Here every phase is
// inline asm, so what runs is exactly what is written:
//
//   A. two waves per SIMD (8-wave workgroup, wave w and w + 4 share a SIMD):
//      an MFMA phase (NM v_mfma_f32_32x32x16_bf16 on 4 accumulators) and an
//      epilogue-like VALU phase (NV instructions) per repetition, as
//        lock   : every wave MFMA | barrier | VALU | barrier      (the kernels today)
//        anti   : waves 4-7 start with the VALU phase, no barriers
//        spec   : waves 0-3 only MFMA, waves 4-7 only VALU (same total work)
//        mfma / valu : one phase only (the two floors)
//      VALU flavours: scalar v_fma_f32 | scalar lrelu + cvt mix | packed
//      v_pk_fma_f32 | lrelu mix + ds_write_b128; priorities 0 / VALU wave 1 /
//      MFMA wave 1.
//   B. one stream: F filler instructions after every MFMA (F = 0..8), scalar or
//      packed, 1 or 2 waves per SIMD: are <= 5 fillers per MFMA gap free?
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/overlap2.hip \
//       -o promonet_amd/lib/overlap2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MFMA(acc) "v_mfma_f32_32x32x16_bf16 %[" #acc "], %[A], %[B], %[" #acc "]\n"
#define SFMA(x) "v_fma_f32 %[" #x "], %[" #x "], %[c], %[d]\n"
#define PFMA(p) "v_pk_fma_f32 %[" #p "], %[" #p "], %[cc], %[dd]\n"

struct State {
    floatx16 a0, a1, a2, a3;
    u32x4 A, B;
    float x0, x1, x2, x3, x4, x5, x6, x7;
    floatx2 p0, p1, p2, p3;
    float c, d;
    floatx2 cc, dd;
};

#define M_OPS(s) [a0] "+v"(s.a0), [a1] "+v"(s.a1), [a2] "+v"(s.a2), [a3] "+v"(s.a3)
#define X_OPS(s)                                                             \
    [x0] "+v"(s.x0), [x1] "+v"(s.x1), [x2] "+v"(s.x2), [x3] "+v"(s.x3),      \
    [x4] "+v"(s.x4), [x5] "+v"(s.x5), [x6] "+v"(s.x6), [x7] "+v"(s.x7)
#define P_OPS(s) [p0] "+v"(s.p0), [p1] "+v"(s.p1), [p2] "+v"(s.p2), [p3] "+v"(s.p3)
#define M_INS(s) [A] "v"(s.A), [B] "v"(s.B)
#define X_INS(s) [c] "v"(s.c), [d] "v"(s.d)
#define P_INS(s) [cc] "v"(s.cc), [dd] "v"(s.dd)
#define ST_OPS(s) M_OPS(s), X_OPS(s), P_OPS(s)
#define ST_INS(s) M_INS(s), X_INS(s), P_INS(s)

// ---- MFMA phase: 16 MFMAs per block ---------------------------------------
__device__ __forceinline__ void mfma16(State& s) {
    asm volatile(
        MFMA(a0) MFMA(a1) MFMA(a2) MFMA(a3) MFMA(a0) MFMA(a1) MFMA(a2) MFMA(a3)
        MFMA(a0) MFMA(a1) MFMA(a2) MFMA(a3) MFMA(a0) MFMA(a1) MFMA(a2) MFMA(a3)
        : M_OPS(s) : M_INS(s));
}

// ---- VALU phases: 64 instructions per block --------------------------------
// flavour 0: scalar v_fma_f32 on 8 independent chains
__device__ __forceinline__ void valu64_fma(State& s) {
#define R8 SFMA(x0) SFMA(x1) SFMA(x2) SFMA(x3) SFMA(x4) SFMA(x5) SFMA(x6) SFMA(x7)
    asm volatile(R8 R8 R8 R8 R8 R8 R8 R8 : X_OPS(s) : X_INS(s));
}
// flavour 1: the epilogue mix per 8 elements: 8 v_mul + 8 v_max (LeakyReLU),
// 4 v_cvt_pk_bf16_f32, 2 permlane32_swap-like v_mov = 22 -> x3 = 66 ~ 64
#define LRELU8                                                               \
    "v_mul_f32 %[t0], 0x3dcccccd, %[x0]\n v_mul_f32 %[t1], 0x3dcccccd, %[x1]\n" \
    "v_mul_f32 %[t2], 0x3dcccccd, %[x2]\n v_mul_f32 %[t3], 0x3dcccccd, %[x3]\n" \
    "v_max_f32 %[x0], %[x0], %[t0]\n v_max_f32 %[x1], %[x1], %[t1]\n"         \
    "v_max_f32 %[x2], %[x2], %[t2]\n v_max_f32 %[x3], %[x3], %[t3]\n"         \
    "v_mul_f32 %[t0], 0x3dcccccd, %[x4]\n v_mul_f32 %[t1], 0x3dcccccd, %[x5]\n" \
    "v_mul_f32 %[t2], 0x3dcccccd, %[x6]\n v_mul_f32 %[t3], 0x3dcccccd, %[x7]\n" \
    "v_max_f32 %[x4], %[x4], %[t0]\n v_max_f32 %[x5], %[x5], %[t1]\n"         \
    "v_max_f32 %[x6], %[x6], %[t2]\n v_max_f32 %[x7], %[x7], %[t3]\n"         \
    "v_cvt_pk_bf16_f32 %[t0], %[x0], %[x1]\n v_cvt_pk_bf16_f32 %[t1], %[x2], %[x3]\n" \
    "v_cvt_pk_bf16_f32 %[t2], %[x4], %[x5]\n v_cvt_pk_bf16_f32 %[t3], %[x6], %[x7]\n" \
    "v_sub_f32 %[x0], %[d], %[x0]\n v_sub_f32 %[x4], %[d], %[x4]\n"
__device__ __forceinline__ void valu64_mix(State& s, unsigned lds_addr, bool lds) {
    unsigned t0, t1, t2, t3;
    if (lds) {
        asm volatile(LRELU8 "ds_write_b128 %[la], %[tt]\n"
                     LRELU8 "ds_write_b128 %[la], %[tt] offset:4352\n"
                     LRELU8 "ds_write_b128 %[la], %[tt] offset:8704\n"
                     : X_OPS(s), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2),
                       [t3] "=&v"(t3)
                     : X_INS(s), [la] "v"(lds_addr), [tt] "v"(s.A));
    } else {
        asm volatile(LRELU8 LRELU8 LRELU8
                     : X_OPS(s), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2),
                       [t3] "=&v"(t3)
                     : X_INS(s));
    }
}
// flavour 2: packed f32: 32 v_pk_fma_f32 (= the arithmetic of 64 scalar fma)
__device__ __forceinline__ void valu64_pk(State& s) {
#define P4 PFMA(p0) PFMA(p1) PFMA(p2) PFMA(p3)
    asm volatile(P4 P4 P4 P4 P4 P4 P4 P4 : P_OPS(s) : P_INS(s));
}

template <int flavour>
__device__ __forceinline__ void valu_phase(State& s, int nv64,
                                           unsigned lds_addr) {
#pragma unroll 1
    for (int i = 0; i < nv64; ++i) {
        if constexpr (flavour == 0) valu64_fma(s);
        else if constexpr (flavour == 1) valu64_mix(s, lds_addr, false);
        else if constexpr (flavour == 2) valu64_pk(s);
        else valu64_mix(s, lds_addr, true);
    }
}
__device__ __forceinline__ void mfma_phase(State& s, int nm16) {
#pragma unroll 1
    for (int i = 0; i < nm16; ++i) mfma16(s);
}

__device__ __forceinline__ void init_state(State& s, const unsigned* fill) {
    const int t = threadIdx.x;
    for (int r = 0; r < 16; ++r) { s.a0[r] = 0.f; s.a1[r] = 0.f; s.a2[r] = 0.f; s.a3[r] = 0.f; }
    for (int r = 0; r < 4; ++r) { s.A[r] = fill[(t * 8 + r) & 4095]; s.B[r] = fill[(t * 8 + 4 + r) & 4095]; }
    s.x0 = 0.5f + t; s.x1 = -1.5f; s.x2 = 2.5f; s.x3 = -3.5f;
    s.x4 = 4.5f; s.x5 = -5.5f; s.x6 = 6.5f; s.x7 = -7.5f - t;
    s.p0 = floatx2{1.f, 2.f}; s.p1 = floatx2{3.f, 4.f}; s.p2 = floatx2{5.f, 6.f}; s.p3 = floatx2{7.f, 8.f};
    s.c = 0.999f; s.d = 0.001f; s.cc = floatx2{0.999f, 0.999f}; s.dd = floatx2{0.001f, 0.001f};
}
__device__ __forceinline__ void sink_state(State& s, float* sink) {
    asm volatile("s_nop 15\n s_nop 15\n" ::: "memory");
    float v = s.x0 + s.x1 + s.x2 + s.x3 + s.x4 + s.x5 + s.x6 + s.x7 + s.p0[0] + s.p1[1] + s.p2[0] + s.p3[1];
    for (int r = 0; r < 16; ++r) v += s.a0[r] + s.a1[r] + s.a2[r] + s.a3[r];
    if (v == 12345.678f) sink[0] = v;
}

// mode: 0 lock, 1 anti, 2 mfma only, 3 valu only, 4 spec (waves 0-3 MFMA x2,
// 4-7 VALU x2). prio: 0 none, 1 VALU-phase at priority 1, 2 MFMA-phase at 1,
// 3 waves 4-7 static priority 1
template <int mode, int flavour, int prio>
__global__ __launch_bounds__(512) void kernA(
    const unsigned* __restrict__ fill, float* sink, unsigned long long* cyc,
    int reps, int nm16, int nv64) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    State s;
    init_state(s, fill);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds_addr = (threadIdx.x & 63) * 16 + wave * 16384;
    const bool upper = wave >= 4;
    if (prio == 3 && upper) __builtin_amdgcn_s_setprio(1);
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    if (mode == 1 && upper) {
        if (prio == 1) __builtin_amdgcn_s_setprio(1);
        valu_phase<flavour>(s, nv64, lds_addr);
        if (prio == 1) __builtin_amdgcn_s_setprio(0);
    }
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
        if (mode == 4) {
            if (!upper) { mfma_phase(s, 2 * nm16); }
            else { valu_phase<flavour>(s, 2 * nv64, lds_addr); }
            continue;
        }
        if (mode != 3) {
            if (prio == 2) __builtin_amdgcn_s_setprio(1);
            mfma_phase(s, nm16);
            if (prio == 2) __builtin_amdgcn_s_setprio(0);
        }
        if (mode == 0) __builtin_amdgcn_s_barrier();
        if (mode != 2) {
            if (prio == 1) __builtin_amdgcn_s_setprio(1);
            valu_phase<flavour>(s, nv64, lds_addr);
            if (prio == 1) __builtin_amdgcn_s_setprio(0);
        }
        if (mode == 0) __builtin_amdgcn_s_barrier();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = c1 - c0;
    sink_state(s, sink);
}

// ---- B: F fillers behind every MFMA in ONE stream ---------------------------
template <int F, bool PK>
__device__ __forceinline__ void filled4(State& s) {
#define FS1(x) SFMA(x)
#define GAP_S(n) (n >= 1 ? SFMA(x0) : "")
    // (string selection has to happen at compile time: spelled out per F)
    if constexpr (!PK) {
        if constexpr (F == 0) asm volatile(MFMA(a0) MFMA(a1) MFMA(a2) MFMA(a3) : ST_OPS(s) : ST_INS(s));
        if constexpr (F == 2) asm volatile(MFMA(a0) SFMA(x0) SFMA(x1) MFMA(a1) SFMA(x2) SFMA(x3) MFMA(a2) SFMA(x4) SFMA(x5) MFMA(a3) SFMA(x6) SFMA(x7) : ST_OPS(s) : ST_INS(s));
        if constexpr (F == 4) asm volatile(MFMA(a0) SFMA(x0) SFMA(x1) SFMA(x2) SFMA(x3) MFMA(a1) SFMA(x4) SFMA(x5) SFMA(x6) SFMA(x7) MFMA(a2) SFMA(x0) SFMA(x1) SFMA(x2) SFMA(x3) MFMA(a3) SFMA(x4) SFMA(x5) SFMA(x6) SFMA(x7) : ST_OPS(s) : ST_INS(s));
        if constexpr (F == 5) asm volatile(MFMA(a0) SFMA(x0) SFMA(x1) SFMA(x2) SFMA(x3) SFMA(x4) MFMA(a1) SFMA(x5) SFMA(x6) SFMA(x7) SFMA(x0) SFMA(x1) MFMA(a2) SFMA(x2) SFMA(x3) SFMA(x4) SFMA(x5) SFMA(x6) MFMA(a3) SFMA(x7) SFMA(x0) SFMA(x1) SFMA(x2) SFMA(x3) : ST_OPS(s) : ST_INS(s));
        if constexpr (F == 6) asm volatile(MFMA(a0) SFMA(x0) SFMA(x1) SFMA(x2) SFMA(x3) SFMA(x4) SFMA(x5) MFMA(a1) SFMA(x6) SFMA(x7) SFMA(x0) SFMA(x1) SFMA(x2) SFMA(x3) MFMA(a2) SFMA(x4) SFMA(x5) SFMA(x6) SFMA(x7) SFMA(x0) SFMA(x1) MFMA(a3) SFMA(x2) SFMA(x3) SFMA(x4) SFMA(x5) SFMA(x6) SFMA(x7) : ST_OPS(s) : ST_INS(s));
        if constexpr (F == 8) asm volatile(MFMA(a0) R8 MFMA(a1) R8 MFMA(a2) R8 MFMA(a3) R8 : ST_OPS(s) : ST_INS(s));
        if constexpr (F == 12) asm volatile(MFMA(a0) R8 SFMA(x0) SFMA(x1) SFMA(x2) SFMA(x3) MFMA(a1) R8 SFMA(x4) SFMA(x5) SFMA(x6) SFMA(x7) MFMA(a2) R8 SFMA(x0) SFMA(x1) SFMA(x2) SFMA(x3) MFMA(a3) R8 SFMA(x4) SFMA(x5) SFMA(x6) SFMA(x7) : ST_OPS(s) : ST_INS(s));
    } else {
        // the same arithmetic as F scalar fma, packed: F / 2 v_pk_fma_f32
        if constexpr (F == 2) asm volatile(MFMA(a0) PFMA(p0) MFMA(a1) PFMA(p1) MFMA(a2) PFMA(p2) MFMA(a3) PFMA(p3) : ST_OPS(s) : ST_INS(s));
        if constexpr (F == 4) asm volatile(MFMA(a0) PFMA(p0) PFMA(p1) MFMA(a1) PFMA(p2) PFMA(p3) MFMA(a2) PFMA(p0) PFMA(p1) MFMA(a3) PFMA(p2) PFMA(p3) : ST_OPS(s) : ST_INS(s));
        if constexpr (F == 8) asm volatile(MFMA(a0) P4 MFMA(a1) P4 MFMA(a2) P4 MFMA(a3) P4 : ST_OPS(s) : ST_INS(s));
    }
}

template <int F, bool PK>
__global__ __launch_bounds__(512) void kernB(
    const unsigned* __restrict__ fill, float* sink, unsigned long long* cyc,
    int reps) {
    State s;
    init_state(s, fill);
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
        filled4<F, PK>(s); filled4<F, PK>(s); filled4<F, PK>(s); filled4<F, PK>(s);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = c1 - c0;
    sink_state(s, sink);
}


// ---- C: dependent chains ----------------------------------------------------
// NACC accumulators in rotation (1 = every MFMA accumulates onto the previous
// one's result, the N-tile-outer order of a conv with ONE M tile), F scalar
// fillers behind every MFMA.
#define CH1(F1) MFMA(a0) F1 MFMA(a0) F1 MFMA(a0) F1 MFMA(a0) F1
#define CH2(F1) MFMA(a0) F1 MFMA(a1) F1 MFMA(a0) F1 MFMA(a1) F1
#define FILL0
#define FILL1 SFMA(x0)
#define FILL2 SFMA(x0) SFMA(x1)
#define FILL4 SFMA(x0) SFMA(x1) SFMA(x2) SFMA(x3)
template <int NACC, int F>
__global__ __launch_bounds__(512) void kernC(
    const unsigned* __restrict__ fill, float* sink, unsigned long long* cyc,
    int reps) {
    State s;
    init_state(s, fill);
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
#define BODY(CH, FL) asm volatile(CH(FL) CH(FL) CH(FL) CH(FL) : ST_OPS(s) : ST_INS(s));
        if constexpr (NACC == 1 && F == 0) BODY(CH1, FILL0)
        if constexpr (NACC == 1 && F == 1) BODY(CH1, FILL1)
        if constexpr (NACC == 1 && F == 2) BODY(CH1, FILL2)
        if constexpr (NACC == 1 && F == 4) BODY(CH1, FILL4)
        if constexpr (NACC == 2 && F == 0) BODY(CH2, FILL0)
        if constexpr (NACC == 2 && F == 1) BODY(CH2, FILL1)
        if constexpr (NACC == 2 && F == 2) BODY(CH2, FILL2)
        if constexpr (NACC == 2 && F == 4) BODY(CH2, FILL4)
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = c1 - c0;
    sink_state(s, sink);
}

static unsigned* g_fill; static float* g_sink; static unsigned long long* g_cyc;
static hipEvent_t e0, e1;

template <class L>
static void timeit(const char* name, double mfmas, L launch) {
    float ms = 0; unsigned long long cyc = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    hipMemcpy(&cyc, g_cyc, 8, hipMemcpyDeviceToHost);
    printf("%-58s %8.2f ms  %9llu cyc  %5.2f GHz", name, ms, cyc, cyc / (ms * 1e6));
    if (mfmas > 0) printf("  %6.0f TFLOP/s", mfmas * 32768.0 / ms / 1e9);
    printf("\n");
    fflush(stdout);
}

int main(int argc, char** argv) {
    const bool zero = argc > 1 && atoi(argv[1]) == 0;
    std::vector<unsigned> hf(4096);
    srand(1);
    for (auto& v : hf) {
        // two bf16 in [-1, 1)
        auto bf = [&]() { float f = zero ? 0.f : (float)(rand() % 2001) / 1000.f - 1.f; unsigned u; std::memcpy(&u, &f, 4); return u >> 16; };
        v = bf() | (bf() << 16);
    }
    hipMalloc(&g_fill, hf.size() * 4); hipMalloc(&g_sink, 4); hipMalloc(&g_cyc, 8);
    hipMemcpy(g_fill, hf.data(), hf.size() * 4, hipMemcpyHostToDevice);
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("# overlap2: %s operands\n", zero ? "zero" : "random");
    const bool only_c = argc > 2 && argv[2][0] == 'C';
    // ---- A ----
    const int reps = 8000;
    // (NM, NV) in units of 16 MFMAs / 64 VALU: MFMA 32 cyc each = 512 per
    // unit; 64 VALU at ~4 cyc = 256 per unit
    struct Shape { int nm16, nv64; } shapes[] = {{8, 8}, {8, 16}, {8, 4}};
    for (auto sh : shapes) {
        if (only_c) break;
#define RUNA(MODE, FL, PRIO)                                                 \
        { const char* mname[5] = {"lock", "anti", "mfma only", "valu only", "spec"}; \
          const char* fname[4] = {"scalar fma", "scalar lrelu+cvt", "packed fma", "lrelu+cvt+ds_write"}; \
          char name[128];                                                    \
          snprintf(name, sizeof name, "A nm=%3d nv=%4d %-18s %-9s prio%d",   \
                   sh.nm16 * 16, sh.nv64 * 64, fname[FL], mname[MODE], PRIO); \
          const double mf = MODE == 3 ? 0 : 256.0 * 8 * reps * sh.nm16 * 16; \
          auto k = kernA<MODE, FL, PRIO>;                                    \
          hipFuncSetAttribute(reinterpret_cast<const void*>(k),              \
              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);       \
          timeit(name, mf, [&]() {                                           \
              hipLaunchKernelGGL(k, dim3(256), dim3(512), 160 * 1024, 0,     \
                                 g_fill, g_sink, g_cyc, reps, sh.nm16, sh.nv64); }); }
#define RUNA_FL(FL)                                                          \
        RUNA(2, FL, 0) RUNA(3, FL, 0) RUNA(0, FL, 0) RUNA(0, FL, 3)          \
        RUNA(1, FL, 0) RUNA(1, FL, 1) RUNA(1, FL, 2) RUNA(1, FL, 3)          \
        RUNA(4, FL, 0) RUNA(4, FL, 3)
        RUNA_FL(0) RUNA_FL(1) RUNA_FL(2) RUNA_FL(3)
    }
    // ---- B ----
    const int repsB = 40000;
    for (int threads : {256, 512}) {
        if (only_c) break;
        const double mf = 256.0 * (threads / 64) * repsB * 16;
#define RUNB(F, PK)                                                          \
        { char name[128];                                                    \
          snprintf(name, sizeof name, "B %d waves/SIMD  %2d %s fillers per MFMA", \
                   threads / 256, F, PK ? "packed-equivalent" : "scalar");   \
          timeit(name, mf, [&]() {                                           \
              hipLaunchKernelGGL((kernB<F, PK>), dim3(256), dim3(threads), 0, 0, \
                                 g_fill, g_sink, g_cyc, repsB); }); }
        RUNB(0, false) RUNB(2, false) RUNB(4, false) RUNB(5, false)
        RUNB(6, false) RUNB(8, false) RUNB(12, false)
        RUNB(2, true) RUNB(4, true) RUNB(8, true)
    }
    // ---- C ----
    {
        const int threads = 256;      // one wave per SIMD: cycles are exact
        const double mf = 256.0 * (threads / 64) * repsB * 16;
#define RUNC(NACC, F)                                                        \
        { char name[128];                                                    \
          snprintf(name, sizeof name, "C 1 wave/SIMD  %d accumulator(s) in rotation, %d scalar fillers per MFMA", NACC, F); \
          timeit(name, mf, [&]() {                                           \
              hipLaunchKernelGGL((kernC<NACC, F>), dim3(256), dim3(threads), 0, 0, \
                                 g_fill, g_sink, g_cyc, repsB); }); }
        RUNC(1, 0) RUNC(1, 1) RUNC(1, 2) RUNC(1, 4)
        RUNC(2, 0) RUNC(2, 1) RUNC(2, 2) RUNC(2, 4)
    }
    return 0;
}
