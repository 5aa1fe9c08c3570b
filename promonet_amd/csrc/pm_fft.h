// Framed STFT as a real FFT in LDS (round 3; the framed-DFT GEMM of pm_stft.h
// stays for the backward pass only).
// Reference: promonet/preprocess/spectrogram.py:15-60 (torch.stft of the
// reflect-padded audio, hann-1024, hop 256, sqrt(re^2 + im^2 + 1e-6)),
// promonet/preprocess/loudness.py:17-55 (librosa.stft, amplitude_to_db with
// top_db 80, A-weighting, band_average) and spectrogram.py:111-133 (log-mel).
//
// One wave = one frame at a time: the 1024 windowed real samples are packed
// into 512 complex points z[n] = x[2n] + i x[2n + 1], transformed by a
// 512-point complex FFT (three radix-8 stages, 8 points per lane in registers,
// two wave-private exchanges through LDS), and unpacked into the 513 bins of
// the real transform - bins k and 512 - k out of the same pair (Z[k],
// Z[512 - k]), so a lane reads 8 values and keeps 4 twiddles. A workgroup owns
// FR consecutive frames of one utterance: every wave loads its frame's samples
// straight into registers (coalesced 512-byte rows, the next frame requested
// before the current one is transformed, unconditionally - a frame past the
// end is a transform of the last one, never stored; the 4x overlap of
// neighbouring frames is served by L1 / L2, the reflect padding is resolved on
// the fly in the two edge frames), and the 513 x FR results are transposed
// through LDS so that the (B, 513, T) spectrogram is written in rows of FR
// consecutive frames, 16 bytes a store.
// HBM traffic = the algorithmic 4 B / sample in + 2052 B / frame out
// (the GEMM formulation did 40x the FLOPs: 57.9 GFLOP of dense DFT against
// ~1.4 GFLOP of FFT at batch 32 x 10 s).
// The kernel is INSTRUCTION-ISSUE bound (profiles/r05/stft_pmc.txt: the SIMDs'
// issue ports are busy 88 % of the time; a packed fp32 instruction holds a
// SIMD for 8 cycles, any other for 4), so the arithmetic is spelled as packed
// instructions with operand modifiers (pk_* below): 180 arithmetic
// instructions and no register moves a frame (round 4: 275 + 102).
//
// EPI 1: magnitude (B, 513, T)                         spectrogram.py:53
// EPI 2: utterance maximum of 10 log10(max(1e-10, |X|^2)) only (no output)
// EPI 3: A-weighted, floored dB, band means (B, bands, T)   loudness.py:46-55,84-111
// EPI 4: log-mel (B, mels, T)                           spectrogram.py:111-133
// EPI 5: EPI 3 for the default 8 bands: bin lane + 64 j of a frame lies in band
//        j (band_average's edges int(b 513 / 8) are 0, 64, ... 448, 513), so
//        the band sums are reduced across the wave out of the registers - no
//        513-row staging tile, 8 x FR floats of LDS instead
// EPI 6: the OPTIMISTIC first pass of the default loudness: EPI 5's band means
//        with no floor at all, plus what the floor needs (every group's
//        maximum, as EPI 2) and what tells whether it would have changed
//        anything (every group's MINIMUM dB). A bin only feels the floor when it
//        lies more than top_db under the utterance's maximum: for a group whose
//        minimum is not, max(v, floor) == v bit for bit and EPI 6's means are
//        final - the second pass (EPI 5) skips it without a transform.
#pragma once
#include "pm_common.h"

#define PM_FFT_N 1024
#define PM_FFT_HOP 256
#define PM_FFT_BINS 513
#define PM_FFT_PAD 384
// table layout (floats): hann[1024] | W512^j (re, im) j < 512 | W1024^k k <= 512
#define PM_FFT_TAB_W512 1024
#define PM_FFT_TAB_W1024 (1024 + 1024)
#define PM_FFT_TAB_FLOATS (1024 + 1024 + 1026)
// 10 log10(x) = (10 / log2(10)) log2(x): one v_log_f32 (1 ulp) instead of the
// ~25-instruction log10f, 8 times per lane and frame
#define PM_DB_PER_LOG2 3.01029995663981195f
// sqrt(|X|^2 + 1e-6): the argument is >= 1e-6, never subnormal, so the bare
// v_sqrt_f32 (1 ulp) is enough; sqrtf() wraps it in a subnormal-scaling
// sequence (7 more instructions, 8 times per lane and frame)
#ifdef PM_FFT_LIBM_SQRT
#define PM_FFT_SQRT(x) sqrtf(x)
#else
#define PM_FFT_SQRT(x) __builtin_amdgcn_sqrtf(x)
#endif

struct FftArgs {
    const float* audio;      // (B, N)
    float* out;
    const float* tables;     // PM_FFT_TAB_FLOATS
    float* group_max;        // (B, groups): every group's maximum dB (EPI 2 / 6
                             // write it, EPI 3 / 5 fold the utterance's row)
    float* group_min;        // (B, groups) or null: every group's MINIMUM dB
                             // (EPI 6 writes it; EPI 5 skips a group none of
                             // whose bins lies under the floor)
    const float* weights;    // EPI 3: (513) A-weights; EPI 4: mel basis (M, 513)
    const int* mel_span;     // EPI 4: pm_mel_csr_kernel's table (lo, hi, offset
                             // per filter, then the non-zero count)
    const float* mel_vals;   // EPI 4: the filters' non-zero spans, compacted
    int B, N, T;
    int groups;              // FR-frame groups per utterance
    int total;               // groups x B: a workgroup takes blockIdx.x,
                             // blockIdx.x + gridDim.x, ... (fft_launch)
    int rows;                // EPI 3: bands; EPI 4: mels
    int band_start[17];
    float min_db, top_db;    // EPI 3
    int use_thr; float thr;  // EPI 4
};

// Complex arithmetic on register PAIRS (re, im), one packed instruction per
// complex operation: the swizzles and signs of a multiplication by +-i, a
// conjugation or a complex product are operand modifiers of v_pk_add_f32 /
// v_pk_mul_f32 / v_pk_fma_f32 (op_sel picks the half of a source that feeds
// the low result, op_sel_hi the high one; neg_lo / neg_hi negate a source for
// one half). hipcc does not find these forms from scalar or vector C++ - it
// re-pairs registers with v_mov / v_pk_mov around every rotation (round 4's
// build: 102 moves and 275 arithmetic instructions a frame; this one: 182
// arithmetic, 0 moves) - so they are spelled out. Plain (non-volatile) asm:
// the compiler still schedules, CSEs and allocates around them.
typedef float pm_v2 __attribute__((ext_vector_type(2)));
#define PM_PK2(name, mods)                                                    \
    __device__ __forceinline__ pm_v2 name(pm_v2 a, pm_v2 b) {                 \
        pm_v2 d;                                                              \
        asm("v_pk_add_f32 %0, %1, %2" mods : "=v"(d) : "v"(a), "v"(b));       \
        return d;                                                             \
    }
PM_PK2(pk_add, "")                                                 // a + b
PM_PK2(pk_sub, " neg_lo:[0,1] neg_hi:[0,1]")                       // a - b
PM_PK2(pk_add_i, " op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")     // a + i b
PM_PK2(pk_sub_i, " op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")     // a - i b
PM_PK2(pk_add_conj, " neg_hi:[0,1]")                               // a + conj b
PM_PK2(pk_sub_conj, " neg_lo:[0,1]")                               // a - conj b
#undef PM_PK2
// a w = (a.x w.x - a.y w.y, a.x w.y + a.y w.x)
__device__ __forceinline__ pm_v2 pk_cmul(pm_v2 a, pm_v2 w) {
    pm_v2 t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] "
        "neg_lo:[0,1,0]" : "=v"(d) : "v"(a), "v"(w), "v"(t));
    return d;
}
// z[i] *= w[i] for i = FIRST .. N - 1: every product's first half, then every
// second half - a packed instruction that consumes the result of the one in
// front of it costs a wait state (an s_nop in the instruction stream)
template <int FIRST, int N>
__device__ __forceinline__ void pk_cmul_each(pm_v2 (&z)[N], const pm_v2 (&w)[N]) {
    pm_v2 t[N];
#pragma unroll
    for (int i = FIRST; i < N; ++i)
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]"
            : "=v"(t[i]) : "v"(z[i]), "v"(w[i]));
#pragma unroll
    for (int i = FIRST; i < N; ++i)
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] "
            "neg_lo:[0,1,0]" : "=v"(z[i]) : "v"(z[i]), "v"(w[i]), "v"(t[i]));
}
// (a.x + a.y, a.x - a.y)
__device__ __forceinline__ pm_v2 pk_sumdiff(pm_v2 a) {
    pm_v2 d;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]"
        : "=v"(d) : "v"(a));
    return d;
}
// Sum of v over the 64 lanes of the wave, returned wave-uniform: six DPP adds
// (quad swaps, the two row mirrors, then the row broadcasts that gfx9 keeps:
// row_bcast:15 / :31 leave the total in lane 63) and one v_readlane.
__device__ __forceinline__ float pm_wave_sum(float v) {
#define PM_DPP_ADD(ctrl, rows)                                                \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(               \
        0, __builtin_bit_cast(int, v), ctrl, rows, 0xf, true))
    PM_DPP_ADD(0xB1, 0xf);      // quad_perm:[1,0,3,2]
    PM_DPP_ADD(0x4E, 0xf);      // quad_perm:[2,3,0,1]
    PM_DPP_ADD(0x141, 0xf);     // row_half_mirror
    PM_DPP_ADD(0x140, 0xf);     // row_mirror
    PM_DPP_ADD(0x142, 0xa);     // row_bcast:15 into rows 1 and 3
    PM_DPP_ADD(0x143, 0xc);     // row_bcast:31 into rows 2 and 3
#undef PM_DPP_ADD
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(
        __builtin_bit_cast(int, v), 63));
}
// |a|^2
__device__ __forceinline__ float pk_norm(pm_v2 a) {
    const pm_v2 q = a * a;
    return q.x + q.y;
}

// In-place 8-point DFT, a[k] <- sum_n a[n] W8^(n k), W8 = exp(-2 pi i / 8):
// 28 packed instructions (the two 4-point halves with their +-i rotations
// folded into the additions, W8 and W8^3 as one scaling of a sum and a
// difference)
__device__ __forceinline__ void pm_radix8(pm_v2 (&a)[8]) {
    const float h = 0.70710678118654752440f;
    const pm_v2 b0 = pk_add(a[0], a[4]), b4 = pk_sub(a[0], a[4]);
    const pm_v2 b1 = pk_add(a[1], a[5]), e5 = pk_sub(a[1], a[5]);
    const pm_v2 b2 = pk_add(a[2], a[6]), e6 = pk_sub(a[2], a[6]);
    const pm_v2 b3 = pk_add(a[3], a[7]), e7 = pk_sub(a[3], a[7]);
    // even outputs: the 4-point DFT of b0 .. b3
    {
        const pm_v2 d0 = pk_add(b0, b2), d1 = pk_sub(b0, b2);
        const pm_v2 d2 = pk_add(b1, b3), e = pk_sub(b1, b3);
        a[0] = pk_add(d0, d2); a[2] = pk_sub_i(d1, e);
        a[4] = pk_sub(d0, d2); a[6] = pk_add_i(d1, e);
    }
    // odd outputs: the 4-point DFT of b4, W8 e5, -i e6, W8^3 e7 with
    // W8 e5 = h (e5 - i e5), W8^3 e7 = -h (e7 + i e7)
    {
        const pm_v2 u5 = pk_sub_i(e5, e5), u7 = pk_add_i(e7, e7);
        const pm_v2 d0 = pk_sub_i(b4, e6), d1 = pk_add_i(b4, e6);
        const pm_v2 d2 = pk_sub(u5, u7) * h, e = pk_add(u5, u7) * h;
        a[1] = pk_add(d0, d2); a[3] = pk_sub_i(d1, e);
        a[5] = pk_sub(d0, d2); a[7] = pk_add_i(d1, e);
    }
}

// Orders the lanes' LDS accesses of one wave: the exchanges below hand data
// from lane to lane through LDS without a workgroup barrier (a wave's DS
// instructions execute in order), so the compiler must not move this thread's
// reads above the other lanes' writes.
__device__ __forceinline__ void pm_wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

// EPI 4 keeps the mel filters' non-zero spans (2 x 513 + rounding values for
// librosa's 80 triangles) in LDS when they fit
#define PM_FFT_MEL_CAP 2048
template <int EPI, int NW, int FPW>
__host__ __device__ constexpr int pm_fft_smem_bytes() {
    constexpr int FR = NW * FPW;
    return NW * 576 * 8 +
           (EPI == 2 ? 64 : EPI == 5 ? 8 * (FR + 1) * 4 : EPI == 6 ? 8 * (FR + 1) * 4 + 128
                     : PM_FFT_BINS * (FR + 1) * 4) +
           (EPI == 4 ? PM_FFT_MEL_CAP * 4 : 0);
}

// The 1024 samples of frame t as 8 x (x[2 (64 i + lane)], x[.. + 1]): padded
// position p = 256 t + n maps to audio sample p - 384, reflected at both ends
// (torch.nn.functional.pad(mode='reflect'), spectrogram.py:36-37).
__device__ __forceinline__ void pm_fft_load_frame(
    pm_v2 (&raw)[8], const float* __restrict__ ab, int t, int N, int lane) {
    const int base = t * PM_FFT_HOP - PM_FFT_PAD;
    if (base >= 0 && base + PM_FFT_N <= N) {          // (wave-uniform)
        const float* p = ab + base + 2 * lane;
        if ((reinterpret_cast<uintptr_t>(p) & 7) == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                raw[i] = *reinterpret_cast<const pm_v2*>(p + 128 * i);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                raw[i] = pm_v2{p[128 * i], p[128 * i + 1]};
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float v[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                int j = base + 2 * (64 * i + lane) + e;
                j = j < 0 ? -j : j;
                j = j >= N ? 2 * (N - 1) - j : j;
                v[e] = ab[j];
            }
            raw[i] = pm_v2{v[0], v[1]};
        }
    }
}

// (second launch bound = waves per SIMD the register allocation must leave
// room for: four for the magnitude / log-mel shapes - two 8-wave workgroups a
// CU - three for the loudness passes' dB epilogues, two for the 32-frame
// groups whose staging tile leaves room for one workgroup a CU)
template <int EPI, int NW, int FPW>
__global__ __launch_bounds__(NW * 64, NW * FPW == 32 ? 2 : (EPI == 1 || EPI == 4) ? 4 : 3)
void pm_stft_fft_kernel(FftArgs a) {
    constexpr int FR = NW * FPW;
    constexpr int NT = NW * 64;
#ifdef PM_FFT_NO_DIRECT_DC           // (A/B builds)
    constexpr bool DIRECT_DC = false;
#else
    constexpr bool DIRECT_DC = EPI == 2 || EPI == 3 || EPI == 5 || EPI == 6;
#endif
    constexpr int WS = 576;          // complex slots per wave (8 x 72)
    constexpr int OS = FR + 1;       // staging row pitch (floats)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    pm_v2* work = reinterpret_cast<pm_v2*>(smem);
    [[maybe_unused]] float* ost = reinterpret_cast<float*>(work + NW * WS);
    [[maybe_unused]] float* melv = ost + PM_FFT_BINS * OS;   // EPI 4

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Persistent workgroups: the grid is what the chip holds at once and a
    // workgroup walks groups g = blockIdx.x, + gridDim.x, ... - the per-lane
    // tables below (32 twiddle / window pairs and their addresses, a quarter
    // of a two-frame wave's instructions) are set up once, not per group
    // XCD-aware walk: workgroup w runs on XCD w % 8, so each XCD takes one
    // contiguous eighth of the groups and its workgroups walk that eighth -
    // neighbouring groups of an utterance share every 128-byte line of the
    // (.., T) output rows (a group owns 64 bytes of each) and 3/4 of their
    // first frame's samples, and now meet in ONE L2 instead of two
    int g = blockIdx.x, stride = gridDim.x, end = a.total;
#ifndef PM_FFT_NO_XCD
    if ((gridDim.x & 7u) == 0u) {
        const long long x = blockIdx.x & 7u;
        stride = (int)(gridDim.x >> 3);
        g = (int)(x * a.total / 8) + (int)(blockIdx.x >> 3);
        end = (int)((x + 1) * a.total / 8);
        if (g >= end) return;               // (workgroup-uniform)
    }
#endif
    int b = g / a.groups;
    int t0 = (g - b * a.groups) * FR;
    const int T = a.T, N = a.N;
    const float* ab = a.audio + (size_t)b * N;

    // the first frame's samples are requested before anything else. Every
    // frame load of this kernel is UNCONDITIONAL (a frame index past the end
    // of the utterance is clamped to its last frame: computed, never stored) -
    // conditional reloads of a loop-carried register array cost a copy of it
    // in and out of every branch (24 v_mov_b64 a frame, round 4's build)
    pm_v2 raw[8];
    pm_fft_load_frame(raw, ab, min(t0 + wave, T - 1), N, lane);

    [[maybe_unused]] bool mel_in_lds = false;
    if constexpr (EPI == 4) {
        const int nnz = a.mel_span[3 * a.rows];
        mel_in_lds = nnz <= PM_FFT_MEL_CAP;
        if (mel_in_lds)
            for (int i = tid; i < nnz; i += NT) melv[i] = a.mel_vals[i];
    }

    // ---- per-lane constants -------------------------------------------------
    const float* __restrict__ tab = a.tables;
    const pm_v2* __restrict__ w512 =
        reinterpret_cast<const pm_v2*>(tab + PM_FFT_TAB_W512);
    const pm_v2* __restrict__ w1024 =
        reinterpret_cast<const pm_v2*>(tab + PM_FFT_TAB_W1024);
    const int k0p = lane >> 3, m0p = lane & 7;
    pm_v2 win[8], twa[8], twb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        win[i] = *reinterpret_cast<const pm_v2*>(tab + 2 * (64 * i + lane));
        twa[i] = w512[lane * i];            // W512^(m k0), m = lane
        twb[i] = w512[8 * m0p * i];         // W64^(m0 q0)
    }
    // Unpacking: bins k = lane + 64 j (j < 4) and 512 - k come out of the SAME
    // pair (Z[k], Z[512 - k]) - a lane takes both, so a frame reads 8 values
    // and 4 twiddles (kept here, as -i W1024^k) instead of 16 and 8
    pm_v2 wun[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const pm_v2 w = w1024[lane + 64 * j];
        wun[j] = pm_v2{w.y, -w.x};         // -i w
    }
    // EPI 3 / 5: the A-weights of this lane's 8 (+ 1) bins
    [[maybe_unused]] float wlo[4], whi[4], w256 = 0.f;
    if constexpr (EPI == 3 || EPI == 5 || EPI == 6) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            wlo[j] = a.weights[lane + 64 * j];
            whi[j] = a.weights[512 - lane - 64 * j];
        }
        w256 = a.weights[256];
    }
    [[maybe_unused]] float floor_db = EPI == 6 ? -INFINITY : 0.f;
    // (EPI 6: an OPAQUE -inf - with the constant the compiler drops the max and
    // contracts v + weight into the multiplication in front of it, one rounding
    // fewer than EPI 5's mul / max / add: the two passes must agree bit for bit)
    if constexpr (EPI == 6) asm volatile("" : "+v"(floor_db));
    [[maybe_unused]] int floor_of = -1;    // utterance floor_db belongs to
    [[maybe_unused]] int parity = 0;       // EPI 2 / 6: reduction slots alternate
    pm_v2* wk = work + wave * WS;
#pragma unroll 1
    for (;;) {
    // the group after this one (its first frame is requested under the last
    // frame's transform)
    const int gn = g + stride;
    const bool more = gn < end;             // (workgroup-uniform)
    const int bn = more ? gn / a.groups : b;
    const int t0n = more ? (gn - bn * a.groups) * FR : t0;
    const float* __restrict__ abn = a.audio + (size_t)bn * N;
    float local_max = (EPI == 2 || EPI == 6) ? 0.f : -INFINITY;   // (of |X|^2 >= 0)
    [[maybe_unused]] float local_min = INFINITY;     // EPI 6: of the bins' dB
    if constexpr (EPI == 3 || EPI == 5) {
        if (floor_of != b) {               // (workgroup-uniform)
        // the utterance maximum of pass 1 (librosa.amplitude_to_db's top_db
        // reference): fold this utterance's per-group maxima
        float m = -INFINITY;
        const float* gm = a.group_max + (size_t)b * a.groups;
        for (int i = tid; i < a.groups; i += NT) m = fmaxf(m, gm[i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        float* red = reinterpret_cast<float*>(work);   // (free between groups)
        if (lane == 0) red[wave] = m;
        __syncthreads();
        m = red[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) m = fmaxf(m, red[w]);
        __syncthreads();
        floor_db = m - a.top_db;
        floor_of = b;
        }
    }
    if constexpr (EPI == 5) {
        // behind the optimistic first pass (EPI 6): a group none of whose bins
        // lies under the floor already has its final means - no transform
        if (a.group_min &&
            a.group_min[(size_t)b * a.groups + t0 / FR] >= floor_db) {
            if (!more) break;
            pm_fft_load_frame(raw, abn, min(t0n + wave, T - 1), N, lane);
            g = gn; b = bn; t0 = t0n; ab = abn;
            continue;
        }
    }

#pragma unroll 1
    for (int i = 0; i < FPW; ++i) {
        const int fl = wave + NW * i;       // frame of this workgroup
        // (frames past the utterance's end are transforms of its last frame:
        // their columns of the staging tile are never written out)
        [[maybe_unused]] const bool valid = t0 + fl < T;   // (wave-uniform)
        pm_v2 z[8];
        // stage A: lane m holds z[64 n2 + m], n2 = 0..7
#pragma unroll
        for (int n2 = 0; n2 < 8; ++n2) z[n2] = raw[n2] * win[n2];
        // (the next frame's samples fly under this frame's transform: the
        // wave's next frame of this group, or the first one of the next group)
        {
            const bool same = i + 1 < FPW;
            const float* __restrict__ an = same ? ab : abn;
            const int tn = same ? t0 + fl + NW : t0n + wave;
            int glane = lane;              // (opaque per frame, see `wlane`)
            asm volatile("" : "+v"(glane));
            pm_fft_load_frame(raw, an, min(tn, T - 1), N, glane);
        }
        // Bins 0 and 512 directly (the dB epilogues): X[0] = sum w x, X[512] =
        // sum (-1)^n w x. Out of the transform they are Re Z[0] +- Im Z[0] - a
        // difference of two 512-term sums that carries the whole frame's
        // rounding noise (8e-8 of its amplitude), which a bin 70 dB under its
        // frame shows as millidecibels of loudness. Here the sum and the
        // difference of every (even, odd) sample pair are formed FIRST - the
        // cancellation happens between two neighbouring samples, where it is
        // all but exact - and only then added up: 8 + 7 packed instructions
        // and two 6-step wave sums a frame. (The magnitude / log-mel
        // epilogues add 1e-6 under a square root and keep the transform's.)
        [[maybe_unused]] float x_dc = 0.f, x_ny = 0.f;
        if constexpr (DIRECT_DC) {
            pm_v2 sd[8];
#pragma unroll
            for (int n2 = 0; n2 < 8; ++n2) sd[n2] = pk_sumdiff(z[n2]);
#pragma unroll
            for (int h = 4; h > 0; h >>= 1)
#pragma unroll
                for (int n2 = 0; n2 < h; ++n2) sd[n2] = pk_add(sd[n2], sd[n2 + h]);
            x_dc = pm_wave_sum(sd[0].x);
            x_ny = pm_wave_sum(sd[0].y);
        }
        pm_radix8(z);
        pk_cmul_each<1>(z, twa);
        pm_wave_lds_sync();                 // (previous frame's reads of wk)
#pragma unroll
        for (int k0 = 0; k0 < 8; ++k0) wk[k0 * 72 + lane] = z[k0];
        pm_wave_lds_sync();
        // stage B: lane (k0, m0) takes m1 = 0..7 of sequence k0
#pragma unroll
        for (int m1 = 0; m1 < 8; ++m1) z[m1] = wk[k0p * 72 + 8 * m1 + m0p];
        pm_radix8(z);
        pk_cmul_each<1>(z, twb);
        pm_wave_lds_sync();
#pragma unroll
        for (int q0 = 0; q0 < 8; ++q0) wk[k0p * 72 + q0 * 9 + m0p] = z[q0];
        pm_wave_lds_sync();
        // stage C: lane (k0, q0) takes m0 = 0..7
#pragma unroll
        for (int m0 = 0; m0 < 8; ++m0) z[m0] = wk[k0p * 72 + m0p * 9 + m0];
        pm_radix8(z);
        // z[q1] = Z[k0 + 8 q0 + 64 q1]: natural order for the unpacking
        pm_wave_lds_sync();
#pragma unroll
        for (int q1 = 0; q1 < 8; ++q1) wk[k0p + 8 * m0p + 64 * q1] = z[q1];
        pm_wave_lds_sync();
        // unpack: with E2 = Z[k] + conj Z[512 - k], D = Z[k] - conj Z[512 - k]
        //   2 X[k] = E2 - i W1024^k D,   2 conj X[512 - k] = E2 + i W1024^k D
        // (k = 0 pairs with itself: X[0] and X[512] = Re Z[0] +- Im Z[0])
        [[maybe_unused]] float bs[8];       // EPI 5: this lane's bin of band j
        auto bin_out = [&](float pw4, int k, [[maybe_unused]] float weight,
                           [[maybe_unused]] int band) {
            // pw4 = 4 |X[k]|^2
            if constexpr (EPI == 1 || EPI == 4) {
                ost[k * OS + fl] = PM_FFT_SQRT(fmaf(pw4, 0.25f, 1e-6f));
            } else if constexpr (EPI == 2) {
                // (the dB map is monotonic: the maximum is taken over |X|^2
                // and mapped once per group)
                local_max = fmaxf(local_max, valid ? pw4 : 0.f);
            } else {
                const float v = PM_DB_PER_LOG2 *
                                __log2f(fmaxf(1e-10f, 0.25f * pw4));
                if constexpr (EPI == 6) {
                    local_max = fmaxf(local_max, valid ? pw4 : 0.f);
                    local_min = fminf(local_min, valid ? v : INFINITY);
                }
                float u = fmaxf(v, floor_db) + weight;
                u = u < a.min_db ? a.min_db : u;
                if constexpr (EPI == 5 || EPI == 6) bs[band] = u;
                else ost[k * OS + fl] = u;
            }
        };
        {
            pm_v2 e2[4], dd[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = lane + 64 * j;
                const pm_v2 zk = wk[k];
                const pm_v2 zm = wk[(512 - k) & 511];
                e2[j] = pk_add_conj(zk, zm);
                dd[j] = pk_sub_conj(zk, zm);
            }
            pk_cmul_each<0>(dd, wun);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = lane + 64 * j;
                float plo = pk_norm(pk_add(e2[j], dd[j]));
                float phi = pk_norm(pk_sub(e2[j], dd[j]));
                if (DIRECT_DC && j == 0) {
                    // (lane 0: bins 0 and 512, from the direct sums)
                    plo = lane == 0 ? 4.f * x_dc * x_dc : plo;
                    phi = lane == 0 ? 4.f * x_ny * x_ny : phi;
                }
                bin_out(plo, k, (EPI == 3 || EPI >= 5) ? wlo[j] : 0.f, j);
                bin_out(phi, 512 - k, (EPI == 3 || EPI >= 5) ? whi[j] : 0.f,
                        7 - j);
            }
        }
        {
            // bin 256 pairs with itself: |X[256]| = |Z[256]|; lane 0 takes it
            const float pw4 = 4.f * pk_norm(wk[256]);
            if constexpr (EPI == 5 || EPI == 6) {
                // lane 0's partner bins 448, 384, 320 sit one band above where
                // the other lanes' do (512 - 64 j = 64 (8 - j)): shift them up
                // and put bin 256 into band 4
                const float v = PM_DB_PER_LOG2 *
                                __log2f(fmaxf(1e-10f, 0.25f * pw4));
                if constexpr (EPI == 6) {       // (the same value in every lane)
                    local_max = fmaxf(local_max, valid ? pw4 : 0.f);
                    local_min = fminf(local_min, valid ? v : INFINITY);
                }
                float u = fmaxf(v, floor_db) + w256;
                u = u < a.min_db ? a.min_db : u;
                if (lane == 0) {
                    bs[7] += bs[6]; bs[6] = bs[5]; bs[5] = bs[4]; bs[4] = u;
                }
            } else if (lane == 0) {
                bin_out(pw4, 256, w256, 4);
            }
        }
        if constexpr (EPI == 5 || EPI == 6) {
            // 8 sums over 64 lanes in 10 exchanges: three halving steps (a
            // lane keeps the bands of its own half and adds the partner's
            // values for them), then three plain ones; lane 8 band ends up
            // with the sum of band `band`
            float v4[4], v2[2], v1;
            {
                const bool hi = lane & 32;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float mine = hi ? bs[q + 4] : bs[q];
                    const float send = hi ? bs[q] : bs[q + 4];
                    v4[q] = mine + __shfl_xor(send, 32, 64);
                }
            }
            {
                const bool hi = lane & 16;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float mine = hi ? v4[q + 2] : v4[q];
                    const float send = hi ? v4[q] : v4[q + 2];
                    v2[q] = mine + __shfl_xor(send, 16, 64);
                }
            }
            {
                const bool hi = lane & 8;
                const float mine = hi ? v2[1] : v2[0];
                const float send = hi ? v2[0] : v2[1];
                v1 = mine + __shfl_xor(send, 8, 64);
            }
            v1 += __shfl_xor(v1, 4, 64);
            v1 += __shfl_xor(v1, 2, 64);
            v1 += __shfl_xor(v1, 1, 64);
            if ((lane & 7) == 0) {
                const int band = lane >> 3;
                ost[band * OS + fl] = v1 / (band == 7 ? 65.f : 64.f);
            }
        }
    }

    // (opaque copy of the thread index: the write-out's per-lane addresses are
    // recomputed per group instead of being kept in registers across the
    // frame loop, which sits at the 128-register bound of four waves per SIMD)
    int wtid = tid;
    asm volatile("" : "+v"(wtid));

    if constexpr (EPI == 2) {
        // one plain store per group (atomics on 32 addresses serialise); the
        // two sets of reduction slots alternate, so the next group's partial
        // maxima never land on ones thread 0 has yet to read
        static_assert(NW <= 8, "two sets of NW slots in the 64 bytes behind the work area");
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            local_max = fmaxf(local_max, __shfl_xor(local_max, o, 64));
        float* red = reinterpret_cast<float*>(work + NW * WS) + 8 * parity;
        parity ^= 1;
        if (lane == 0) red[wave] = local_max;
        __syncthreads();
        if (tid == 0) {
            float m = red[0];
#pragma unroll
            for (int w = 1; w < NW; ++w) m = fmaxf(m, red[w]);
            a.group_max[(size_t)b * a.groups + t0 / FR] =
                PM_DB_PER_LOG2 * __log2f(fmaxf(1e-10f, 0.25f * m));   // (m = 4 |X|^2)
        }
    } else {
    if constexpr (EPI == 6) {
        // the group's maximum |X|^2 and minimum dB: slots behind the staging
        // rows, two alternating sets (as EPI 2)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            local_max = fmaxf(local_max, __shfl_xor(local_max, o, 64));
            local_min = fminf(local_min, __shfl_xor(local_min, o, 64));
        }
        float* red = ost + 8 * OS + 16 * parity;
        if (lane == 0) { red[wave] = local_max; red[8 + wave] = local_min; }
    }
    __syncthreads();
    if constexpr (EPI == 6) {
        if (tid == 0) {
            const float* red = ost + 8 * OS + 16 * parity;
            float m = red[0], n = red[8];
#pragma unroll
            for (int w = 1; w < NW; ++w) {
                m = fmaxf(m, red[w]); n = fminf(n, red[8 + w]);
            }
            const size_t slot = (size_t)b * a.groups + t0 / FR;
            a.group_max[slot] =
                PM_DB_PER_LOG2 * __log2f(fmaxf(1e-10f, 0.25f * m));
            a.group_min[slot] = n;
        }
        parity ^= 1;
    }
    const int nf = min(FR, T - t0);         // frames this group owns
    if (EPI == 1 || (EPI == 3 && a.rows == PM_FFT_BINS)) {
        // a thread takes FOUR consecutive frames of one bin: one 16-byte store
        // (rows of the (.., T) output are only 4-byte aligned - T is odd -
        // which global_store_dwordx4 accepts) instead of four stores with
        // their index arithmetic: 4 (8 at 32 frames) passes over the tile
        // instead of 16 - the write-out was 80 of a frame's 440 instructions
        float* ob = a.out + (size_t)b * PM_FFT_BINS * T + t0;
        constexpr int QR = FR / 4;                  // quads per row
        typedef float pm_f4u __attribute__((ext_vector_type(4), aligned(4)));
        for (int idx = wtid; idx < PM_FFT_BINS * QR; idx += NT) {
            const int bin = idx / QR, c = (idx % QR) * 4;
            const float* src = ost + bin * OS + c;
            float* dst = ob + (size_t)bin * T + c;
            if (c + 3 < nf) {
                const pm_f4u v = {src[0], src[1], src[2], src[3]};
                *reinterpret_cast<pm_f4u*>(dst) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 3; ++e)
                    if (c + e < nf) dst[e] = src[e];
            }
        }
    } else if constexpr (EPI == 5 || EPI == 6) {
        float* ob = a.out + (size_t)b * 8 * T + t0;
        for (int idx = wtid; idx < 8 * FR; idx += NT) {
            const int band = idx / FR, c = idx % FR;
            if (c < nf) ob[(size_t)band * T + c] = ost[band * OS + c];
        }
    } else if constexpr (EPI == 3) {
        // band means, rows summed in ascending order (loudness.py:96-111)
        float* ob = a.out + (size_t)b * a.rows * T + t0;
        for (int idx = wtid; idx < a.rows * FR; idx += NT) {
            const int band = idx / FR, c = idx % FR;
            if (c >= nf) continue;
            const int r0 = a.band_start[band], r1 = a.band_start[band + 1];
            float s = 0.f;
            for (int f = r0; f < r1; ++f) s += ost[f * OS + c];
            ob[(size_t)band * T + c] = s / (float)(r1 - r0);
        }
    } else if constexpr (EPI == 4) {
        // log(basis @ magnitude) over each filter's non-zero span, ascending
        // (two loops, not one pointer select: a pointer that may be LDS or
        // global is a FLAT pointer, and flat loads cost this epilogue 60 us)
        float* ob = a.out + (size_t)b * a.rows * T + t0;
        for (int idx = wtid; idx < a.rows * FR; idx += NT) {
            const int m = idx / FR, c = idx % FR;
            if (c >= nf) continue;
            const int lo = a.mel_span[3 * m], n = a.mel_span[3 * m + 1] - lo;
            const int off = a.mel_span[3 * m + 2];
            const float* sp = ost + lo * OS + c;
            float acc = 0.f;
            if (mel_in_lds) {
                const float* bv = melv + off;
#pragma unroll 4
                for (int f = 0; f < n; ++f) acc = fmaf(bv[f], sp[f * OS], acc);
            } else {
                const float* __restrict__ bv = a.mel_vals + off;
#pragma unroll 4
                for (int f = 0; f < n; ++f) acc = fmaf(bv[f], sp[f * OS], acc);
            }
            float v = logf(acc);
            if (a.use_thr) v = fmaxf(v, a.thr);
            ob[(size_t)m * T + c] = v;
        }
    }
    }   // (EPI != 2)
    if (!more) break;
    if constexpr (EPI != 2) __syncthreads();   // staging tile read out
    g = gn; b = bn; t0 = t0n; ab = abn;
    }   // groups of this workgroup
}

// Compact form of a (M, F) filterbank: per row the first / one-past-last
// non-zero column and the offset of that span in `vals`, then the total count.
// One workgroup; M <= 1024.
__global__ __launch_bounds__(256) void pm_mel_csr_kernel(
    const float* __restrict__ basis, int* __restrict__ table,
    float* __restrict__ vals, int M, int F) {
    __shared__ int lo_s[1024], hi_s[1024], off_s[1025];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int m = wave; m < M; m += 4) {
        const float* br = basis + (size_t)m * F;
        int lo = F, hi = 0;
        for (int f = lane; f < F; f += 64)
            if (br[f] != 0.f) { lo = min(lo, f); hi = max(hi, f + 1); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo = min(lo, __shfl_xor(lo, o, 64));
            hi = max(hi, __shfl_xor(hi, o, 64));
        }
        if (lane == 0) { lo_s[m] = hi > lo ? lo : 0; hi_s[m] = hi > lo ? hi : 0; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
        for (int m = 0; m < M; ++m) { off_s[m] = total; total += hi_s[m] - lo_s[m]; }
        off_s[M] = total;
        table[3 * M] = total;
    }
    __syncthreads();
    for (int m = wave; m < M; m += 4) {
        if (lane == 0) {
            table[3 * m] = lo_s[m]; table[3 * m + 1] = hi_s[m];
            table[3 * m + 2] = off_s[m];
        }
        for (int f = lo_s[m] + lane; f < hi_s[m]; f += 64)
            vals[off_s[m] + f - lo_s[m]] = basis[(size_t)m * F + f];
    }
}
