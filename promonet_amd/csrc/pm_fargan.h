// FARGAN: frame-autoregressive GRU vocoder (reference: promonet/model/fargan.py,
// config/fargan.py). 4 dependent sub-frame steps per frame, ~11 dependent
// dense layers per step: latency-bound, not FLOP-bound (74 kFLOP / sample).
//
// Mapping (round 1): ONE PERSISTENT WORKGROUP PER UTTERANCE walks the whole
// sequence; every recurrent quantity (3 GRU states, the 260-wide framewise
// state, the last 512 output samples) stays in LDS for the entire utterance,
// HBM sees one feature row in and 256 samples out per frame. Each dense layer
// is a matrix-vector product with thread <-> output row, the weights streamed
// from L2 (2.76 M parameters do not fit a CU) in a [k / VEC][row][VEC] packing
// so a wave reads 1 KB of consecutive rows per instruction and needs no
// cross-lane reduction; short layers split K over thread groups and reduce
// through LDS. Utterances are independent -> B workgroups run concurrently.
#pragma once
#include "pm_common.h"
#include <cstddef>
#include <type_traits>

#ifndef FG_UNROLL
#define FG_UNROLL 16
#endif
#define FG_THREADS 768
#define FG_HOP 256
#define FG_SUB 64
#define FG_PREV 512
#define FG_SUBIN 260          // 128 features + 64 previous + 68 lookback
#define FG_SKIP 1152

// Dot products are written as explicit fma chains in a fixed order so that
// every instantiation (kernel variant, utterances in lockstep) rounds alike.
template <class WT> struct FgVec;
template <> struct FgVec<float> {
    static constexpr int VEC = 4;
    __device__ static __forceinline__ void unpack(uint4 wv, float (&f)[4]) {
        f[0] = __uint_as_float(wv.x); f[1] = __uint_as_float(wv.y);
        f[2] = __uint_as_float(wv.z); f[3] = __uint_as_float(wv.w);
    }
    __device__ static __forceinline__ float dotf(
        const float (&f)[4], const float* x) {
        const float4 b = *reinterpret_cast<const float4*>(x);
        return fmaf(f[3], b.w, fmaf(f[2], b.z, fmaf(f[1], b.y, f[0] * b.x)));
    }
    __device__ static __forceinline__ float dot(
        const float* __restrict__ w, const float* x) {
        float f[4];
        unpack(*reinterpret_cast<const uint4*>(w), f);
        return dotf(f, x);
    }
};
template <> struct FgVec<_Float16> {
    static constexpr int VEC = 8;
    // 16 weight bytes already in registers -> fp32 once, then one dot per
    // utterance
    __device__ static __forceinline__ void unpack(uint4 wv, float (&f)[8]) {
        const half8 a = __builtin_bit_cast(half8, wv);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = (float)a[i];
    }
    __device__ static __forceinline__ float dotf(
        const float (&f)[8], const float* x) {
        const float4 b0 = *reinterpret_cast<const float4*>(x);
        const float4 b1 = *reinterpret_cast<const float4*>(x + 4);
        float d = f[0] * b0.x;
        d = fmaf(f[1], b0.y, d); d = fmaf(f[2], b0.z, d); d = fmaf(f[3], b0.w, d);
        d = fmaf(f[4], b1.x, d); d = fmaf(f[5], b1.y, d); d = fmaf(f[6], b1.z, d);
        return fmaf(f[7], b1.w, d);
    }
    __device__ static __forceinline__ float dot(
        const _Float16* __restrict__ w, const float* x) {
        float f[8];
        unpack(*reinterpret_cast<const uint4*>(w), f);
        return dotf(f, x);
    }
};

// Partial matrix-vector product for thread `tid` of a (RPAD x PARTS) team:
// row = tid % RPAD, K-slice = tid / RPAD. x is an LDS vector; columns
// [0, split) come from xa, [split, K) from xb (split a multiple of VEC).
template <class WT, int RPAD, int PARTS>
__device__ __forceinline__ float fg_gemv(
    const WT* __restrict__ w, const float* xa, const float* xb, int split,
    int kpad, int tid) {
    constexpr int VEC = FgVec<WT>::VEC;
    const int row = tid % RPAD, part = tid / RPAD;
    const int blocks = kpad / VEC;
    const int b0 = part * blocks / PARTS, b1 = (part + 1) * blocks / PARTS;
    const int sb = split / VEC;
    const WT* wp = w + ((size_t)b0 * RPAD + row) * VEC;
    // FG_UNROLL independent 16-byte loads per thread in flight: the stream
    // is latency-bound (weights exceed the 4 MB L2 of an XCD and come from
    // the Infinity Cache), bytes in flight set the rate.
    float acc0 = 0.f, acc1 = 0.f;
#pragma unroll FG_UNROLL
    for (int b = b0; b < b1; ++b) {
        const float* x = b < sb ? xa + b * VEC : xb + (b - sb) * VEC;
        const float d = FgVec<WT>::dot(wp, x);
        if (b & 1) acc1 += d; else acc0 += d;
        wp += (size_t)RPAD * VEC;
    }
    return acc0 + acc1;
}

// torch-layout W (rows, cols) fp32 -> [kpad / VEC][rpad][VEC], zero padded
// GRU matrices (768 = 3 gates x 256 units) are stored gate-interleaved in
// blocks of 32 units: packed row r = (j / 32) * 96 + gate * 32 + j % 32, so
// that the 96 rows a cluster member needs for its 32 units are contiguous.
__host__ __device__ __forceinline__ int fg_gru_row(int gate, int j) {
    return (j >> 5) * 96 + gate * 32 + (j & 31);
}

template <class WT>
__global__ __launch_bounds__(256) void pm_fargan_pack_kernel(
    const float* __restrict__ w, WT* __restrict__ out, int rows, int cols,
    int rpad, int kpad, int gru, int col0) {
    constexpr int VEC = FgVec<WT>::VEC;
    const long long total = (long long)rpad * kpad;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int e = i % VEC;
    const int row = (i / VEC) % rpad;
    const int blk = i / ((long long)VEC * rpad);
    const int col = col0 + blk * VEC + e;      // col0: K-split sub-matrix
    int src = row;
    if (gru) src = ((row % 96) / 32) * 256 + (row / 96) * 32 + (row % 32);
    out[i] = (WT)((src < rows && col < cols) ? w[(size_t)src * cols + col]
                                             : 0.f);
}

// All packed layers live in ONE buffer at compile-time offsets (elements), so
// a kernel carries a single base pointer: with one pointer per layer the 23
// of them exhausted the scalar registers and spilled into vector registers.
// Order = the layer table of pm_api.hip (fargan_layers); the K-split copies
// used by the cluster kernel follow the row-packed layers.
// Mixed storage (round 4): the matrices whose rounding the audio does not feel
// - the three GRU cells' W_ih / W_hh (U(-1/16, 1/16) entries in front of a
// sigmoid / tanh) and every GLU gate (in front of a sigmoid) - are stored f16,
// the ones it does feel - conditioning network, framewise conv, skip dense,
// output layer (orthogonal, feeding tanh outputs directly) - stay fp32.
// 80 % of the per-step weight stream is of the first kind: 5.4 MB instead of
// 9.0 MB per sub-frame step at 1/10 of the all-f16 storage error
// (scripts/fargan_weight_sensitivity.py: 6e-6 against 6.7e-5 max-abs).
struct FgMixed {};
template <class WT> struct FgTypes { typedef WT S; typedef WT I; };
template <> struct FgTypes<FgMixed> { typedef float S; typedef _Float16 I; };

template <class WT>
struct FarganWeights {
    typedef typename FgTypes<WT>::S S;   // rounding-sensitive layers
    typedef typename FgTypes<WT>::I I;   // insensitive layers (GRU, gates)
    const S* base;
    const I* base_i;   // read only when I differs from S (same element offsets)
    __device__ __forceinline__ const I* ibase() const {
        if constexpr (std::is_same<S, I>::value)
            return reinterpret_cast<const I*>(base);
        else
            return base_i;
    }
    static constexpr size_t COND0 = 0;                         // 384 x 376
    static constexpr size_t COND1 = COND0 + 384 * 376;         // 384 x 376
    static constexpr size_t COND2 = COND1 + 384 * 376;         // 512 x 376
    static constexpr size_t FWCONV = COND2 + 512 * 376;        // 256 x 520
    static constexpr size_t FWGLU = FWCONV + 256 * 520;        // 256 x 256
    static constexpr size_t GRU_IH = FWGLU + 256 * 256;        // 3 x 768 x 384
    static constexpr size_t GRU_HH = GRU_IH + 3 * 768 * 384;   // 3 x 768 x 256
    static constexpr size_t GRU_GLU = GRU_HH + 3 * 768 * 256;  // 3 x 256 x 256
    static constexpr size_t SKIP = GRU_GLU + 3 * 256 * 256;    // 256 x 1152
    static constexpr size_t SKIP_GLU = SKIP + 256 * 1152;      // 256 x 256
    static constexpr size_t OUT = SKIP_GLU + 256 * 256;        // 64 x 256
    // K-split copies: member g's (R x K / 8) sub-matrix, 8 back to back
    static constexpr size_t K_COND1 = OUT + 64 * 256;          // 8 x 384 x 48
    static constexpr size_t K_FWGLU = K_COND1 + 8 * 384 * 48;  // 8 x 256 x 32
    static constexpr size_t K_GRU_GLU = K_FWGLU + 8 * 256 * 32;    // 3 x 8 x 256 x 32
    static constexpr size_t K_OUT = K_GRU_GLU + 3 * 8 * 256 * 32;  // 8 x 64 x 32
    static constexpr size_t TOTAL = K_OUT + 8 * 64 * 32;
    __device__ __forceinline__ const S* cond(int i) const {
        return base + (i == 0 ? COND0 : i == 1 ? COND1 : COND2);
    }
    __device__ __forceinline__ const S* fwconv() const { return base + FWCONV; }
    __device__ __forceinline__ const I* fwconv_glu() const { return ibase() + FWGLU; }
    __device__ __forceinline__ const I* gru_ih(int n) const {
        return ibase() + GRU_IH + (size_t)n * (768 * 384);
    }
    __device__ __forceinline__ const I* gru_hh(int n) const {
        return ibase() + GRU_HH + (size_t)n * (768 * 256);
    }
    __device__ __forceinline__ const I* gru_glu(int n) const {
        return ibase() + GRU_GLU + (size_t)n * (256 * 256);
    }
    __device__ __forceinline__ const S* skip() const { return base + SKIP; }
    __device__ __forceinline__ const I* skip_glu() const { return ibase() + SKIP_GLU; }
    __device__ __forceinline__ const S* out() const { return base + OUT; }
    // K-split sub-matrix of member g
    __device__ __forceinline__ const S* k_cond1(int g) const {
        return base + K_COND1 + (size_t)g * (384 * 48);
    }
    __device__ __forceinline__ const I* k_fwconv_glu(int g) const {
        return ibase() + K_FWGLU + (size_t)g * (256 * 32);
    }
    __device__ __forceinline__ const I* k_gru_glu(int n, int g) const {
        return ibase() + K_GRU_GLU + (size_t)(n * 8 + g) * (256 * 32);
    }
    __device__ __forceinline__ const S* k_out(int g) const {
        return base + K_OUT + (size_t)g * (64 * 32);
    }
};

struct FarganArgs {
    const float* features_cl;   // (B, T, cstride): 113 features, then period
    const float* global;        // (Bg, G)
    const float* previous;      // (Bp, 512) or null (zeros)
    float* out;                 // (B, 256 T)
    int B, T, cstride, nfeat, G;
    int global_batch, previous_batch;
    const int* lengths;         // (B) valid frames per utterance or null: the
                                // model is causal, so a zero-padded utterance
                                // simply stops early; its tail is zero-filled
};

// The activations sit on the dependency chain of every step (12 of them, one
// after the other): libm's expf + IEEE division are ~40 dependent instructions
// a sigmoid, tanhf ~80 - 250-500 cycles each at one wave's issue rate. Here the
// same formulas on the transcendental unit (v_exp_f32, v_rcp_f32: 1 ulp each,
// 4-6 instructions): absolute error ~1e-7, i.e. fp32 rounding of an O(1)
// value; what it does to the audio is measured (tests/test_gpu_fargan.py).
#ifndef FG_EXACT_ACTIVATIONS
__device__ __forceinline__ float fg_sigmoid(float v) {
    return __builtin_amdgcn_rcpf(
        1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
}
__device__ __forceinline__ float fg_tanh(float v) {
    // 1 - 2 / (1 + e^(2 v)); e^(2 v) = inf / 0 at the ends gives +1 / -1
    return 1.f - 2.f * __builtin_amdgcn_rcpf(
        1.f + __builtin_amdgcn_exp2f(2.8853900817779268f * v));
}
#else
__device__ __forceinline__ float fg_sigmoid(float v) {
    return 1.f / (1.f + expf(-v));
}
__device__ __forceinline__ float fg_tanh(float v) { return tanhf(v); }
#endif

template <class WT>
__global__ __launch_bounds__(FG_THREADS) void pm_fargan_kernel(
    FarganArgs a, FarganWeights<WT> w) {
    constexpr int NT = FG_THREADS;
    constexpr int CPAD = 376;   // 371 conditioning inputs, padded to x8
    typedef typename FgTypes<WT>::S WS;
    typedef typename FgTypes<WT>::I WI;
    __shared__ __attribute__((aligned(16))) float condin[CPAD];
    __shared__ __attribute__((aligned(16))) float c1[CPAD];
    __shared__ __attribute__((aligned(16))) float c2[CPAD];
    __shared__ __attribute__((aligned(16))) float cond[512];
    __shared__ __attribute__((aligned(16))) float subin[2 * FG_SUBIN + 8];
    __shared__ __attribute__((aligned(16))) float skipbuf[FG_SKIP];
    __shared__ __attribute__((aligned(16))) float hid[3][FG_HOP];
    __shared__ __attribute__((aligned(16))) float f1[FG_HOP];
    __shared__ __attribute__((aligned(16))) float part[NT];
    __shared__ __attribute__((aligned(16))) float part2[NT];
    __shared__ float prev[FG_PREV];
    __shared__ int s_period;

    int tid = threadIdx.x;   // re-laundered per step (see the cluster kernel)
    const int b = blockIdx.x;
    const int T = a.lengths ? min(max(a.lengths[b], 0), a.T) : a.T;
    for (int i = T * FG_HOP + tid; i < a.T * FG_HOP; i += NT)
        a.out[(size_t)b * a.T * FG_HOP + i] = 0.f;
    const float* feat = a.features_cl + (size_t)b * a.T * a.cstride;
    const float* glob =
        a.global + (size_t)(a.global_batch == 1 ? 0 : b) * a.G;
    float* out = a.out + (size_t)b * a.T * FG_HOP;
    const int nin = a.nfeat + a.G;   // 371

    // ---- initial recurrent state (fargan.py:406-415) ----
    for (int i = tid; i < 3 * FG_HOP; i += NT) (&hid[0][0])[i] = 0.f;
    for (int i = tid; i < 2 * FG_SUBIN + 8; i += NT) subin[i] = 0.f;
    for (int i = tid; i < CPAD; i += NT) { condin[i] = 0.f; c1[i] = 0.f; c2[i] = 0.f; }
    for (int i = tid; i < FG_PREV; i += NT)
        prev[i] = a.previous
            ? a.previous[(size_t)(a.previous_batch == 1 ? 0 : b) * FG_PREV + i]
            : 0.f;
    int base = 0;   // ring offset of `prev`: logical i -> (base + i) % 512
    __syncthreads();

#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        // ---- frame: conditioning network (fargan.py:139-160) ----
        const float* row = feat + (size_t)t * a.cstride;
        if (tid < a.nfeat) condin[tid] = row[tid];
        else if (tid < nin) condin[tid] = glob[tid - a.nfeat];
        if (tid == 0) s_period = (int)rintf(row[a.nfeat]);   // fargan.py:94
        __syncthreads();
        {
            part[tid] = fg_gemv<WS, 384, 2>(w.cond(0), condin, condin, CPAD, CPAD, tid);
            __syncthreads();
            if (tid < nin) c1[tid] = fg_tanh(part[tid] + part[tid + 384]);
            __syncthreads();
            part[tid] = fg_gemv<WS, 384, 2>(w.cond(1), c1, c1, CPAD, CPAD, tid);
            __syncthreads();
            if (tid < nin) c2[tid] = fg_tanh(part[tid] + part[tid + 384]);
            __syncthreads();
            if (tid < 512)
                cond[tid] = fg_tanh(fg_gemv<WS, 512, 1>(w.cond(2), c2, c2, CPAD, CPAD, tid));
            __syncthreads();
        }
        const int period = s_period;

#pragma unroll 1
        for (int s = 0; s < 4; ++s) {
            asm volatile("" : "+v"(tid));
            // ---- sub-frame inputs (fargan.py:233-256) ----
            if (tid < 128) {
                subin[tid] = cond[4 * tid + s];          // reshape/permute :109
            } else if (tid < 192) {
                const int i = tid - 128;
                const float v = prev[(base + FG_PREV - FG_SUB + i) & (FG_PREV - 1)];
                subin[128 + i] = v;
                skipbuf[1088 + i] = v;
            } else if (tid < 260) {
                const int i = tid - 192;
                int idx = FG_PREV - period + i - 2;
                if (idx >= FG_PREV) idx -= period;
                idx = idx < 0 ? 0 : (idx >= FG_PREV ? FG_PREV - 1 : idx);
                const float v = prev[(base + idx) & (FG_PREV - 1)];
                subin[192 + i] = v;
                if (i >= 2 && i < 66) skipbuf[1024 + i - 2] = v;
            }
            __syncthreads();

            // ---- framewise conv: Linear(520 -> 256), tanh, GLU (:349-372) ----
            part[tid] = fg_gemv<WS, 256, 3>(w.fwconv(), subin, subin, 520, 520, tid);
            __syncthreads();
            if (tid < 256) f1[tid] = fg_tanh(part[tid] + part[tid + 256] + part[tid + 512]);
            __syncthreads();
            part[tid] = fg_gemv<WI, 256, 3>(w.fwconv_glu(), f1, f1, 256, 256, tid);
            __syncthreads();
            if (tid < 256)
                skipbuf[768 + tid] = f1[tid] * fg_sigmoid(
                    part[tid] + part[tid + 256] + part[tid + 512]);
            __syncthreads();

            // ---- three GRU + GLU layers (:269-314) ----
#pragma unroll 1
            for (int n = 0; n < 3; ++n) {
                // input [x | lookback | previous subframe]; x = fwconv output
                // or the previous GLU output
                const float* xa = n == 0 ? skipbuf + 768 : skipbuf + (n - 1) * 256;
                part[tid] = fg_gemv<WI, 768, 1>(
                    w.gru_ih(n), xa, skipbuf + 1024, 256, 384, tid);
                part2[tid] = fg_gemv<WI, 768, 1>(
                    w.gru_hh(n), hid[n], hid[n], 256, 256, tid);
                __syncthreads();
                if (tid < 256) {
                    const int ir = fg_gru_row(0, tid), iz = fg_gru_row(1, tid);
                    const int in_ = fg_gru_row(2, tid);
                    const float r = fg_sigmoid(part[ir] + part2[ir]);
                    const float z = fg_sigmoid(part[iz] + part2[iz]);
                    const float nn = fg_tanh(part[in_] + r * part2[in_]);
                    hid[n][tid] = (1.f - z) * nn + z * hid[n][tid];
                }
                __syncthreads();
                part[tid] = fg_gemv<WI, 256, 3>(w.gru_glu(n), hid[n], hid[n], 256, 256, tid);
                __syncthreads();
                if (tid < 256)
                    skipbuf[n * 256 + tid] = hid[n][tid] * fg_sigmoid(
                        part[tid] + part[tid + 256] + part[tid + 512]);
                __syncthreads();
            }

            // ---- skip connection + output layer (:317-333) ----
            part[tid] = fg_gemv<WS, 256, 3>(w.skip(), skipbuf, skipbuf, FG_SKIP, FG_SKIP, tid);
            __syncthreads();
            if (tid < 256) f1[tid] = fg_tanh(part[tid] + part[tid + 256] + part[tid + 512]);
            __syncthreads();
            part[tid] = fg_gemv<WI, 256, 3>(w.skip_glu(), f1, f1, 256, 256, tid);
            __syncthreads();
            if (tid < 256)
                f1[tid] = f1[tid] * fg_sigmoid(
                    part[tid] + part[tid + 256] + part[tid + 512]);
            __syncthreads();
            part[tid] = fg_gemv<WS, 64, 12>(w.out(), f1, f1, 256, 256, tid);
            __syncthreads();
            if (tid < FG_SUB) {
                float v = 0.f;
#pragma unroll
                for (int p = 0; p < 12; ++p) v += part[tid + 64 * p];
                v = fg_tanh(v);
                out[(size_t)t * FG_HOP + s * FG_SUB + tid] = v;
                // the oldest 64 samples leave the window (fargan.py:122-129)
                prev[(base + tid) & (FG_PREV - 1)] = v;
            } else if (tid >= 256 && tid < 256 + FG_SUBIN) {
                // states[3] <- this sub-frame's input (fargan.py:334)
                subin[FG_SUBIN + tid - 256] = subin[tid - 256];
            }
            base = (base + FG_SUB) & (FG_PREV - 1);
            __syncthreads();
        }
    }
}


// ===========================================================================
// The conditioning network is NOT part of the recurrence (fargan.py:139-160: three
// bias-free Linear + tanh over a frame's 113 features + 258 global channels;
// nothing generated feeds back into it): for fp32-stored conditioning weights
// ('fp32' / 'mixed' storage) the cluster kernel takes it for every frame of the
// batch from ONE launch ahead of the utterance walk - three chained GEMMs over
// B x T independent frames on the exact-fp32 matrix unit (v_mfma_f32_32x32x2_f32)
// - instead of three weight slices and two inter-workgroup exchanges per frame.
// A workgroup owns 32 frames: input and hidden rows stay in LDS, the weights
// are read as they are already packed for the mat-vec kernels ([k / 4][row][4]
// fp32: a lane's float4 = one row, 4 consecutive k - the A operand of four
// MFMAs whose k pairs are (j, 4 + j) of an 8-wide K group). A frame's result
// depends on that frame alone, so it is the same bits whatever batch it is in.
// ===========================================================================
#define FG_CN 32            // frames per workgroup
#define FG_CPITCH 388       // LDS row pitch (floats) of a <= 384-wide operand
#define FG_OPITCH 516       // ... of the 512-wide result (16 B x odd: no conflicts)

struct FarganCondArgs {
    const float* features_cl;   // (B, T, cstride)
    const float* global;        // (Bg, G)
    float* cond;                // (B * T, 512)
    int B, T, cstride, nfeat, G, global_batch;
};

// out[f][row] = tanh(sum_k W[row][k] in[f][k]) for the workgroup's 32 frames;
// `in` has 376 valid (zero-padded) columns; 4 waves x MT M-tiles = ROWS rows
template <int ROWS, int OPITCH>
__device__ __forceinline__ void fg_cond_layer(
    const float* __restrict__ w, const float* in, float* out, int tid) {
    constexpr int MT = ROWS / 32 / 4;
    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    floatx16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const float* wp = w + ((size_t)h * ROWS + wave * MT * 32 + r) * 4;
    const float* bp = in + r * FG_CPITCH + 4 * h;
#pragma unroll 2
    for (int q = 0; q < 376 / 8; ++q) {
        const float4 b = *reinterpret_cast<const float4*>(bp + 8 * q);
        float4 a[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
            a[i] = *reinterpret_cast<const float4*>(
                wp + ((size_t)2 * q * ROWS + i * 32) * 4);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b.x, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b.y, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b.z, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b.w, acc[i], 0, 0, 0);
        }
    }
    // C/D layout: column (frame) = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 h
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            float4 v;
            v.x = fg_tanh(acc[i][4 * g4]);     v.y = fg_tanh(acc[i][4 * g4 + 1]);
            v.z = fg_tanh(acc[i][4 * g4 + 2]); v.w = fg_tanh(acc[i][4 * g4 + 3]);
            *reinterpret_cast<float4*>(
                out + r * OPITCH + (wave * MT + i) * 32 + 8 * g4 + 4 * h) = v;
        }
}

__global__ __launch_bounds__(256) void pm_fargan_cond_kernel(
    FarganCondArgs a, const float* __restrict__ w0,
    const float* __restrict__ w1, const float* __restrict__ w2) {
    extern __shared__ __attribute__((aligned(16))) float cl[];
    float* bufa = cl;                          // [32][FG_CPITCH]
    float* bufb = cl + FG_CN * FG_CPITCH;      // [32][FG_OPITCH]
    const int tid = threadIdx.x;
    const long long total = (long long)a.B * a.T;
    const long long f0 = (long long)blockIdx.x * FG_CN;
    const int nin = a.nfeat + a.G;
    // the frames' inputs [features | global | zeros] (a frame past the end of
    // the batch reads as zeros; its result is not stored)
    for (int i = tid; i < FG_CN * 376; i += 256) {
        const int f = i / 376, k = i % 376;
        const long long frame = f0 + f;
        float v = 0.f;
        if (frame < total && k < nin) {
            const int b = (int)(frame / a.T);
            v = k < a.nfeat
                ? a.features_cl[(size_t)frame * a.cstride + k]
                : a.global[(size_t)(a.global_batch == 1 ? 0 : b) * a.G +
                           (k - a.nfeat)];
        }
        bufa[f * FG_CPITCH + k] = v;
    }
    __syncthreads();
    fg_cond_layer<384, FG_CPITCH>(w0, bufa, bufb, tid);    // padded rows: tanh(0)
    __syncthreads();
    fg_cond_layer<384, FG_CPITCH>(w1, bufb, bufa, tid);
    __syncthreads();
    fg_cond_layer<512, FG_OPITCH>(w2, bufa, bufb, tid);
    __syncthreads();
    for (int i = tid; i < FG_CN * 128; i += 256) {
        const int f = i / 128, q = i % 128;
        if (f0 + f < total)
            reinterpret_cast<float4*>(a.cond + (size_t)(f0 + f) * 512)[q] =
                *reinterpret_cast<const float4*>(bufb + f * FG_OPITCH + 4 * q);
    }
}

// ===========================================================================
// Cluster variant: FG_G = 8 workgroups cooperate on one utterance.
//
// One CU streams the 9.2 MB of weights at ~45 GB/s: 213 us per sub-frame step.
// Here every layer is split over the 8 members of a cluster, so each streams
// 1/8 of the weights; the dispatcher places block b on XCD b % 8 and
// g = b % 8, so each XCD's 4 MB L2 keeps exactly its 1.15 MB slice resident
// for ALL clusters. What bounds a step is the number of inter-member
// exchanges on its dependency chain (an L2-and-beyond round trip each), so
// consecutive layers ALTERNATE the axis they are split on:
//   row split  (R):  member g computes rows [g R / 8, (g + 1) R / 8) of
//                    y = W x from the full (replicated) x -> owns a slice of y
//   K split    (K):  the next layer contracts exactly that slice: member g
//                    computes W[:, slice g] y_g for ALL rows -> partial sums
// and only the partial sums cross members: every member adds the 8 partials
// of every row in the same order (bit-identical replicas) and applies the
// nonlinearity. A (R, K) pair of layers costs ONE exchange instead of two:
//   framewise conv (R) + its GLU gate (K)            1 exchange
//   3 x [GRU cell (R) + GLU gate (K)]                 3
//   skip dense (R)                                    1  (vector exchange)
//   skip GLU (R) + output layer (K)                   1
// 6 per sub-frame step (was 11), and 2 instead of 3 for the per-frame
// conditioning network (R, K, R).
//
// Exchange = per-cluster global buffer of 8-byte {value, epoch} granules (the
// guide's tagged-granule hand-off):
//   publish: one relaxed agent-scope 64-bit atomic store per element
//            (write-through, `sc1`) - value and tag land together;
//   consume: a thread polls ITS granules (relaxed `sc1` 64-bit loads,
//            s_sleep between tries) until every tag equals the epoch.
// No arrival counter, no vmcnt drain, no barrier before the poll.
// Placement-independent (correct for any block -> XCD map), two payload
// buffers alternate by epoch parity (a member can only write epoch e after it
// has read every member's epoch e - 1, i.e. after every member finished
// reading epoch e - 2 out of the same buffer), state is zeroed by a memset
// node before every launch (tag 0 never matches: epochs start at 1), every
// spin is bounded and trips a global error word.
// All other state (GRU states, sample history) is replicated per workgroup.
// ===========================================================================
#define FG_G 8
// threads of a cluster member (the one-workgroup-per-utterance kernel keeps
// FG_THREADS)
#ifndef FG_CT
#define FG_CT 768
#endif

#define FG_UMAX 4              // utterances a cluster advances in lockstep
#ifndef FG_INFLIGHT
#define FG_INFLIGHT 8        // 16-byte weight loads per software-pipeline group
#endif
// What runs under the exchanges (see the step loop): level 0 nothing, 1 W_hh h
// in front of its cell, 2 the balanced schedule over all six exchanges.
// Batch 32 x 10 s (profiles/r03/fargan/ab_under_exchange.txt): fp32 weights
// 112-118 / 106-108 / 106-110 ms, f16-stored 91 / 81 / 75 ms at levels 0 / 1 / 2.
#ifndef FG_UNDER
#define FG_UNDER(WT) 2
#endif
#define FG_SLOTS 416           // granules one member publishes per exchange (max 384 + 32)

struct FgNoOverlap { __device__ __forceinline__ void operator()() const {} };

struct FgCluster {
    unsigned long long* buf;   // [2][FG_UMAX][FG_G][FG_SLOTS] granules
    unsigned* error;           // global: set when a bounded spin gave up
    unsigned epoch;            // epochs count from 1
    int g;                     // this member
};

#define FG_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define FG_CSTATE (2 * FG_UMAX * FG_G * FG_SLOTS * 2 + 16)   // uint32 words per cluster

// Per-utterance recurrent state + scratch of one cluster member, in LDS.
struct FgLds {
    float condin[376];
    float c1[384];
    float c2[384];
    float cond[512];
    float subin[2 * FG_SUBIN + 8];
    float skipbuf[FG_SKIP];
    float hid[3][FG_HOP];
    float f1[FG_HOP];
    float prev[FG_PREV];
    float own[64];             // this member's slice of a row-split output
    float part[FG_CT];    // reduction buffer P (and the exchanges')
    float part2[FG_CT];
    float part3[FG_CT];   // reduction buffer Q (see fg_slice)
    float part4[FG_CT];   // reduction buffers R, R': slices run under an
    float part5[FG_CT];   // exchange (see the step loop)
    float gh[3][96];           // W_hh h of the three GRU cells, computed ahead
    float gil[3][96];          // W_ih[:, 256:384] [lookback | previous subframe]
    float skpre[32];           // the skip dense layer without its last input
};
static_assert(sizeof(FgLds) % 16 == 0, "16-byte aligned per-utterance state");
#define FG_OFF(field) (offsetof(FgLds, field) / sizeof(float))
#define FG_LSTRIDE (sizeof(FgLds) / sizeof(float))

__device__ __forceinline__ unsigned long long* fg_granule(
    const FgCluster& c, unsigned epoch, int u, int member, int slot) {
    return c.buf + (((size_t)(epoch & 1u) * FG_UMAX + u) * FG_G + member) *
                       FG_SLOTS + slot;
}

// Poll N granules per utterance until every tag equals the epoch (bounded).
template <int U, int N>
__device__ __forceinline__ void fg_poll(
    FgCluster& c, unsigned epoch, unsigned long long* const (&src)[U][N],
    float (&val)[U][N]) {
    unsigned long long v[U][N];
    unsigned spins = 0;
    for (;;) {
        bool all = true;
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < N; ++i)
                v[u][i] = __hip_atomic_load(src[u][i], FG_RLX);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < N; ++i)
                all &= (unsigned)(v[u][i] >> 32) == epoch;
        if (all) break;
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 1023u) == 0u &&
            (spins > (1u << 22) || __hip_atomic_load(c.error, FG_RLX))) {
            __hip_atomic_store(c.error, 1u, FG_RLX);
            break;
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int i = 0; i < N; ++i)
            val[u][i] = __uint_as_float((unsigned)v[u][i]);
}

// Lane map of the slices reduced inside a wave (fg_slice_lanes below): LPR lanes
// a row, 64 / LPR rows a wave, rows fastest; lane j = 0 of a row leads
template <int LPR>
struct FgLanes {
    static constexpr int RPW = 64 / LPR;
    __device__ static __forceinline__ int row(int tid) {
        return (tid >> 6) * RPW + (tid & (RPW - 1));
    }
    __device__ static __forceinline__ bool lead(int tid, int rows) {
        return (tid & 63) < RPW && row(tid) < rows;
    }
};

// Vector exchange: `mine[u]` (valid for tid < N) = element g N + tid of
// utterance u's 8 N-long vector; on return field `dst` of every utterance's
// FgLds holds the whole vector in every member's LDS. Callers guarantee (a
// barrier since the last read) that nobody still reads dst's old content.
// `under` runs between the publish and the poll: work that does not depend on
// the exchanged data (a slice of the NEXT layers' products of the previous
// step's state) streams its weights while the granules travel.
// PUB > 0: `mine` comes out of fg_slice_lanes<..., PUB> (element r in the lead
// lane of row r) instead of thread r.
template <int U, int N, int PUB = 0, class Under = FgNoOverlap>
__device__ __forceinline__ void fg_exchange(
    FgCluster& c, const float (&mine)[U], float* lds, int dst, int tid,
    Under under = Under()) {
    c.epoch += 1u;
    const unsigned epoch = c.epoch;
    bool pub = tid < N;
    int slot = tid;
    if constexpr (PUB > 0) {
        pub = FgLanes<PUB>::lead(tid, N);
        slot = FgLanes<PUB>::row(tid);
    }
    if (pub) {
#pragma unroll
        for (int u = 0; u < U; ++u)
            __hip_atomic_store(
                fg_granule(c, epoch, u, c.g, slot),
                ((unsigned long long)epoch << 32) | __float_as_uint(mine[u]),
                FG_RLX);
    }
    under();
    if (tid < FG_G * N) {
        unsigned long long* src[U][1];
        float val[U][1];
#pragma unroll
        for (int u = 0; u < U; ++u)
            src[u][0] = fg_granule(c, epoch, u, tid / N, tid % N);
        fg_poll<U, 1>(c, epoch, src, val);
#pragma unroll
        for (int u = 0; u < U; ++u)
            lds[u * FG_LSTRIDE + dst + tid] = val[u][0];
    }
    __syncthreads();
}

// Partial-sum exchange after a K-split layer. `part[u]` (valid for tid < R) =
// this member's partial sum of row tid; `extra[u]` (valid for tid < E) = this
// member's element g E + tid of an 8 E = R long vector that travels with the
// sums (the gated activation itself). On return, for tid < R: total[u] = the
// sum over members 0..7, in that order, of row tid's partials (identical in
// every member), ext[u] = element tid of the gathered vector.
// Thread `row` polls all eight members' granules of its row itself.
// PAIRED: `part` comes from fg_slice_pair - row 32 w + l sits in lane l < 32 of
// wave w - instead of row tid in thread tid.
template <int U, int R, int E, bool PAIRED = false, class Under = FgNoOverlap>
__device__ __forceinline__ void fg_exchange_sum(
    FgCluster& c, const float (&part)[U], const float (&extra)[U], float* lds,
    int tid, float (&total)[U], float (&ext)[U], Under under = Under()) {
    static_assert(R + E <= FG_SLOTS && R <= FG_CT, "exchange geometry");
    c.epoch += 1u;
    const unsigned epoch = c.epoch;
    if (PAIRED ? ((tid & 32) == 0 && (tid >> 6) < R / 32) : tid < R) {
        const int prow = PAIRED ? (tid >> 6) * 32 + (tid & 31) : tid;
#pragma unroll
        for (int u = 0; u < U; ++u)
            __hip_atomic_store(
                fg_granule(c, epoch, u, c.g, prow),
                ((unsigned long long)epoch << 32) | __float_as_uint(part[u]),
                FG_RLX);
    }
    if (E > 0 && tid < E) {
#pragma unroll
        for (int u = 0; u < U; ++u)
            __hip_atomic_store(
                fg_granule(c, epoch, u, c.g, R + tid),
                ((unsigned long long)epoch << 32) | __float_as_uint(extra[u]),
                FG_RLX);
    }
    under();
    // Thread `row` polls all 8 members' granules of its row (and its element of
    // the vector that travels with them) in ONE loop and adds them in member
    // order: the same bits in every member, no partial totals through LDS and
    // no barrier of the exchange's own - the caller's next barrier (there is
    // at least one between two exchanges, which also keeps the two payload
    // buffers apart) orders whatever `under` left in LDS. What a caller writes to
    // LDS between this exchange and that barrier must not be anything `under`
    // READS (a wave may still be inside `under` when another has its totals):
    // the step keeps them apart - every under-slice reads state vectors /
    // skip columns other than the ones the exchange's consumer writes.
#pragma unroll
    for (int u = 0; u < U; ++u) { total[u] = 0.f; ext[u] = 0.f; }
    if (tid < R) {
        // (four utterances in lockstep: four rounds of two members - the
        // registers of 36 granules in flight are not there; same order)
        constexpr int CH = U >= 4 ? 2 : FG_G;
#pragma unroll
        for (int m0 = 0; m0 < FG_G; m0 += CH) {
            constexpr int XE = E > 0 ? 1 : 0;
            const bool last = m0 + CH == FG_G;
            unsigned long long* src[U][CH + XE];
            float val[U][CH + XE];
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int i = 0; i < CH; ++i)
                    src[u][i] = fg_granule(c, epoch, u, m0 + i, tid);
                if constexpr (E > 0)   // (polled in every round: one loop shape)
                    src[u][CH] = fg_granule(c, epoch, u, tid / (E > 0 ? E : 1),
                                            R + tid % (E > 0 ? E : 1));
            }
            fg_poll<U, CH + XE>(c, epoch, src, val);
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int i = 0; i < CH; ++i) total[u] += val[u][i];
                if constexpr (E > 0) { if (last) ext[u] = val[u][CH]; }
            }
        }
    }
}

// RW rows (r0 .. r0 + RW of a matrix packed with RPAD rows) of y_u = W x_u
// for the U utterances: every thread takes one (row, K-slice) pair, loads
// each 16-byte weight block ONCE and dots it with the U input vectors (LDS
// fields xa / xb of every FgLds); partial sums meet in LDS. sum[u] is valid
// in threads tid < RW. Row split: RW = R / 8 rows of the full-K matrix;
// K split: all RW = RPAD = R rows of this member's (R x K / 8) sub-matrix.
// PB: which of the two reduction buffers (P = `part`, Q = `part3`) the partial
// sums meet in. There is NO barrier behind the reduction: consecutive slices
// (and the partial-sum exchange, which uses P) alternate buffers, so the next
// writer of a buffer is always at least one barrier - the next slice's own, or
// the caller's before it touches the next input vector - behind its readers.
// DEFER: stop after the partial sums are in the reduction buffer - the caller
// passes a barrier of its own (an exchange's) and collects with fg_slice_sum.
// PB 2 / 3 = buffers R / R' (`part4`, `part5`), which only deferred slices use.
// A K sub-range [k0, k0 + KPAD) of a matrix packed with RPAD rows starts at
// w + k0 * RPAD (the packing is [k / VEC][row][VEC]).
template <int PB>
__host__ __device__ constexpr int fg_pbuf() {
    return PB == 0 ? FG_OFF(part) : PB == 1 ? FG_OFF(part3)
         : PB == 2 ? FG_OFF(part4) : FG_OFF(part5);
}

template <int RW, int U, int PB>
__device__ __forceinline__ void fg_slice_sum(
    const float* lds, int tid, float (&sum)[U]) {
    constexpr int PARTS = FG_CT / RW;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        float t = 0.f;
        if (tid < RW) {
#pragma unroll
            for (int q = 0; q < PARTS; ++q)
                t += lds[u * FG_LSTRIDE + fg_pbuf<PB>() + tid + q * RW];
        }
        sum[u] = t;
    }
}

template <class WT, int RW, int RPAD, int U, int KPAD, int PB,
          bool DEFER = false>
__device__ __forceinline__ void fg_slice(
    const WT* __restrict__ w, const float* lds, int xa, int xb, int split,
    int r0, float* ldsw, int tid, float (&sum)[U]) {
    // (with four utterances in lockstep the registers are tight: the
    // per-thread addresses of a slice are computed here, not earlier)
    if constexpr (U >= 2) asm volatile("" : "+v"(tid));
    constexpr int VEC = FgVec<WT>::VEC;
    constexpr int PARTS = FG_CT / RW;
    constexpr int BLOCKS = KPAD / VEC;
    constexpr int NB = (BLOCKS + PARTS - 1) / PARTS;   // blocks per thread
    const int row = tid % RW, p = tid / RW;
    float acc0[U], acc1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc0[u] = acc1[u] = 0.f;
    if (p < PARTS) {
        const int b0 = p * BLOCKS / PARTS, b1 = (p + 1) * BLOCKS / PARTS;
        const int sb = split / VEC;
        const WT* wp = w + ((size_t)b0 * RPAD + r0 + row) * VEC;
        // the stream is latency-bound (an L2 round trip per dependent batch
        // of loads): software-pipelined in groups of FG_INFLIGHT 16-byte
        // blocks - the next group is requested before the current one is
        // used, so only the first round trip of a slice is exposed
        // (four utterances in lockstep leave registers for groups of 2)
        constexpr int DEPTH = U >= 4 ? 2 : FG_INFLIGHT;
        constexpr int R = NB < DEPTH ? NB : DEPTH;
        constexpr int NGROUP = (NB + R - 1) / R;
        uint4 cur[R];
#pragma unroll
        for (int i = 0; i < R; ++i)
            cur[i] = b0 + i < b1
                ? *reinterpret_cast<const uint4*>(wp + (size_t)i * RPAD * VEC)
                : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll 1
        for (int gq = 0; gq < NGROUP; ++gq) {
            const int first = b0 + gq * R;
            uint4 nxt[R];
#pragma unroll
            for (int i = 0; i < R; ++i)
                nxt[i] = (gq + 1 < NGROUP && first + R + i < b1)
                    ? *reinterpret_cast<const uint4*>(
                          wp + (size_t)((gq + 1) * R + i) * RPAD * VEC)
                    : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int b = first + i;
                if (b < b1) {
                    float wf[VEC];
                    FgVec<WT>::unpack(cur[i], wf);
                    const int off =
                        b < sb ? xa + b * VEC : xb + (b - sb) * VEC;
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const float d =
                            FgVec<WT>::dotf(wf, lds + u * FG_LSTRIDE + off);
                        if (b & 1) acc1[u] += d; else acc0[u] += d;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < R; ++i) cur[i] = nxt[i];
        }
    }
    constexpr int PBUF = fg_pbuf<PB>();
#pragma unroll
    for (int u = 0; u < U; ++u)
        ldsw[u * FG_LSTRIDE + PBUF + tid] = acc0[u] + acc1[u];
    if constexpr (DEFER) return;
    __syncthreads();
    fg_slice_sum<RW, U, PB>(lds, tid, sum);
}

// ---------------------------------------------------------------------------
// Slices reduced INSIDE a wave (round 4). fg_slice hands every (row, K-part) to
// one thread and lets the parts of a row meet in LDS behind a barrier: ~500
// cycles a slice, 13 slices a step. Here the LPR lanes that share a row sit in
// ONE wave - lane = j * RPW + r with RPW = 64 / LPR rows a wave, rows fastest, so
// a load instruction still reads RPW consecutive rows (16 B each) of LPR
// different K blocks - and add up with row_ror DPP steps and the two gfx950
// half-wave / quarter-wave swaps: no LDS, no barrier. Lane j = 0 of a row
// ("lead") carries the result on.
// ---------------------------------------------------------------------------
// sum of `v` over the LPR lanes of a row; every lane of the row gets the same
// bits (each step adds the same two partial sums in either order)
template <int LPR>
__device__ __forceinline__ float fg_lane_sum(float v, int tid) {
    static_assert(LPR == 4 || LPR == 8 || LPR == 16, "lanes per row");
    // lane bits: LPR 4 -> j = bits 4-5; 8 -> bits 3-5; 16 -> bits 2-5
    if constexpr (LPR >= 8)
        v += __uint_as_float(__builtin_amdgcn_update_dpp(
            0u, __float_as_uint(v), 0x128, 0xf, 0xf, false));    // row_ror:8
    if constexpr (LPR == 16)
        v += __uint_as_float(__builtin_amdgcn_update_dpp(
            0u, __float_as_uint(v), 0x124, 0xf, 0xf, false));    // row_ror:4
    {
        const unsigned b = __float_as_uint(v);
        const auto sw = __builtin_amdgcn_permlane16_swap(b, b, false, false);
        v += __uint_as_float((tid & 16) ? sw[0] : sw[1]);
    }
    {
        const unsigned b = __float_as_uint(v);
        const auto sw = __builtin_amdgcn_permlane32_swap(b, b, false, false);
        v += __uint_as_float((tid & 32) ? sw[0] : sw[1]);
    }
    return v;
}

// RW rows of y_u = W x_u; `w` points at the first of them in K block 0 of a
// [k / VEC][...][VEC] packing whose K blocks are BSTRIDE elements apart (RPAD *
// VEC for the streamed matrices, RW * VEC for an LDS-resident copy). Lane j of a
// row takes K blocks j, j + LPR, ... (x: [0, split) from LDS field xa, the rest
// from xb). sum[u]: the row's product in every lane of the row.
template <class WT, int RW, int BSTRIDE, int U, int KPAD, int LPR>
__device__ __forceinline__ void fg_slice_lanes(
    const WT* __restrict__ w, const float* lds, int xa, int xb, int split,
    int tid, float (&sum)[U]) {
    if constexpr (U >= 2) asm volatile("" : "+v"(tid));   // (as fg_slice)
    constexpr int VEC = FgVec<WT>::VEC;
    constexpr int BLOCKS = KPAD / VEC;
    constexpr int NB = (BLOCKS + LPR - 1) / LPR;
    constexpr int RPW = 64 / LPR;
    static_assert(RW % RPW == 0 && RW * LPR <= FG_CT, "slice geometry");
    const int lane = tid & 63, j = lane / RPW;
    const int row = (tid >> 6) * RPW + lane % RPW;
    float acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = 0.f;
    if (row < RW) {                                   // (whole waves)
        const int sb = split / VEC;
        const WT* wp = w + (size_t)j * BSTRIDE + row * VEC;
        uint4 wv[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i)
            wv[i] = i * LPR + j < BLOCKS
                ? *reinterpret_cast<const uint4*>(wp + (size_t)i * LPR * BSTRIDE)
                : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int b = i * LPR + j;
            if (b < BLOCKS) {
                float wf[VEC];
                FgVec<WT>::unpack(wv[i], wf);
                const int off = b < sb ? xa + b * VEC : xb + (b - sb) * VEC;
#pragma unroll
                for (int u = 0; u < U; ++u)
                    acc[u] += FgVec<WT>::dotf(wf, lds + u * FG_LSTRIDE + off);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) sum[u] = fg_lane_sum<LPR>(acc[u], tid);
}

// A K-split layer on the member's own 32 inputs, all R rows (GLU gates, the
// output layer): 12 K-values a thread, three partial sums a row through LDS
// and a barrier is what fg_slice makes of it - here the two half-waves of
// wave w share rows 32 w .. 32 w + 31 (lanes 0-31 the first 16 inputs, lanes
// 32-63 the other 16) and ONE v_permlane32_swap adds the halves: no LDS
// reduction, no barrier. sum[u] is valid in lanes 0-31 of waves < R / 32
// (row 32 w + lane): fg_exchange_sum<..., PAIRED> publishes from there.
template <class WT, int R, int U>
__device__ __forceinline__ void fg_slice_pair(
    const WT* __restrict__ w, const float* lds, int x, int tid,
    float (&sum)[U]) {
    constexpr int VEC = FgVec<WT>::VEC;
    constexpr int HB = 16 / VEC;             // 16-byte blocks per half
    if constexpr (U >= 2) asm volatile("" : "+v"(tid));   // (as fg_slice)
    const int wave = tid >> 6, lane = tid & 63, half = lane >> 5;
    float acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = 0.f;
    if (wave < R / 32) {
        const WT* wp =
            w + ((size_t)half * HB * R + wave * 32 + (lane & 31)) * VEC;
        uint4 wv[HB];
#pragma unroll
        for (int i = 0; i < HB; ++i)
            wv[i] = *reinterpret_cast<const uint4*>(wp + (size_t)i * R * VEC);
#pragma unroll
        for (int i = 0; i < HB; ++i) {
            float wf[VEC];
            FgVec<WT>::unpack(wv[i], wf);
#pragma unroll
            for (int u = 0; u < U; ++u)
                acc[u] += FgVec<WT>::dotf(
                    wf, lds + u * FG_LSTRIDE + x + (half * HB + i) * VEC);
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const unsigned bits = __float_as_uint(acc[u]);
        // [1] in lanes 0-31 = the upper half-wave's value of the same row
        const auto sw = __builtin_amdgcn_permlane32_swap(bits, bits, false, false);
        sum[u] = acc[u] + __uint_as_float(sw[1]);
    }
}

// LDS-resident copies of a member's SHORT slices (one utterance per cluster,
// 2-byte insensitive type, i.e. 'mixed' and f16 storage): the five GLU-gate
// slices, the output layer's K slice and the last-GRU columns of the skip dense
// layer are 8 .. 32 KB each, so what they cost per step is not their bytes but
// one exposed L2 round trip apiece on the dependency chain (0.5 - 1.0 us of the
// 24 us step each, profiles/r03/fargan/timeline_fargan.txt). 120 KB ('mixed')
// beside the 37.5 KB of recurrent state: copied once per launch, read with
// ds_read_b128 ever after. Same packing, rows compacted to the member's own;
// the (row, K-slice) -> thread map and the summation order do not change, so
// the audio is bit-identical to the streamed form.
template <class WT, int U>
struct FgResident {
    typedef typename FgTypes<WT>::S S;
    typedef typename FgTypes<WT>::I I;
    static constexpr bool ON = U == 1;
    // 2-byte gate storage: all seven (120 KB 'mixed', 100 KB f16); all-fp32
    // storage (200 KB of them): the three GRU gates and the output layer,
    // 104 KB - what fits beside the state
    static constexpr bool ALL = ON && sizeof(I) == 2;
    static constexpr bool ON_FWGLU = ALL, ON_SKIPGLU = ALL, ON_SKIP3 = ALL;
    static constexpr bool ON_GRUGLU = ON, ON_OUT = ON;
    static constexpr size_t FWGLU = 0;                                    // (256 x 32) I
    static constexpr size_t GRUGLU =
        FWGLU + (ON_FWGLU ? 256 * 32 * sizeof(I) : 0);                    // 3 x (256 x 32) I
    static constexpr size_t SKIPGLU = GRUGLU + 3 * 256 * 32 * sizeof(I);  // (32 x 256) I
    static constexpr size_t OUT =
        SKIPGLU + (ON_SKIPGLU ? 32 * 256 * sizeof(I) : 0);                // (64 x 32) S
    static constexpr size_t SKIP3 = OUT + 64 * 32 * sizeof(S);            // (32 x 256) S
    static constexpr size_t BYTES =
        ON ? SKIP3 + (ON_SKIP3 ? 32 * 256 * sizeof(S) : 0) : 0;
    static_assert(BYTES + sizeof(FgLds) <= 160 * 1024, "LDS budget");
};

// rows r0 .. r0 + RW of a [KPAD / VEC][RPAD][VEC]-packed matrix -> LDS
// [KPAD / VEC][RW][VEC] (`src` points at row r0 of block 0)
template <class T, int RW, int RPAD, int KPAD>
__device__ __forceinline__ void fg_pin(char* dst, const T* src, int tid) {
    constexpr int VEC = FgVec<T>::VEC;
    constexpr int UNITS = KPAD / VEC * RW;      // 16-byte pieces
    for (int i = tid; i < UNITS; i += FG_CT) {
        const int b = i / RW, r = i % RW;
        reinterpret_cast<uint4*>(dst)[i] = *reinterpret_cast<const uint4*>(
            src + ((size_t)b * RPAD + r) * VEC);
    }
}

struct FarganClusterArgs {
    FarganArgs f;
    unsigned* state;      // per cluster: FG_CSTATE words of granules
    unsigned* error;
    int nclusters;
    const float* precond;   // (B * T, 512) from pm_fargan_cond_kernel (fp32-
                            // stored conditioning weights), or null
#ifdef PM_TUNING
    unsigned long long* timeline;   // debug: phase stamps of one sub-frame step
#endif
};

#ifdef PM_TUNING
#define FG_STAMP(i)                                                           \
    do {                                                                      \
        if (ca.timeline && blockIdx.x == 1 && tid == 0 && t == 7 && s == 1)   \
            ca.timeline[i] = __builtin_amdgcn_s_memtime();                    \
    } while (0)
#else
#define FG_STAMP(i) ((void)0)
#endif

// U utterances per cluster advance in lockstep: the weight slice is read once
// per layer for all of them and one exchange carries U vectors, so the
// latency of a layer (an L2-and-beyond round trip) is shared U ways. U = 1 is
// the batch <= 32 case (one utterance per cluster, 32 clusters = 256 CUs).
template <class WT, int U>
__global__ __launch_bounds__(FG_CT) void pm_fargan_cluster_kernel(
    FarganClusterArgs ca, FarganWeights<WT> w) {
    const FarganArgs& a = ca.f;
    typedef typename FgTypes<WT>::S WS;
    typedef typename FgTypes<WT>::I WI;
    constexpr int NT = FG_CT;
    constexpr int CPAD = 376;
    extern __shared__ __attribute__((aligned(16))) float lds[];   // FgLds[U]
    FgLds* L = reinterpret_cast<FgLds*>(lds);

    const int tid = threadIdx.x;
    const int g = blockIdx.x % FG_G;          // = XCD under the b % 8 dispatch
    const int cluster = blockIdx.x / FG_G;
    const int T = a.T;
    const int nin = a.nfeat + a.G;
    typedef FgResident<WT, U> RES;
    char* const res = reinterpret_cast<char*>(lds) + U * sizeof(FgLds);
    if constexpr (RES::ON) {
        if constexpr (RES::ON_FWGLU)
            fg_pin<WI, 256, 256, 32>(res + RES::FWGLU, w.k_fwconv_glu(g), tid);
#pragma unroll
        for (int n = 0; n < 3; ++n)
            fg_pin<WI, 256, 256, 32>(
                res + RES::GRUGLU + n * 256 * 32 * sizeof(WI), w.k_gru_glu(n, g),
                tid);
        if constexpr (RES::ON_SKIPGLU)
            fg_pin<WI, 32, 256, 256>(
                res + RES::SKIPGLU, w.skip_glu() + g * 32 * FgVec<WI>::VEC, tid);
        fg_pin<WS, 64, 64, 32>(res + RES::OUT, w.k_out(g), tid);
        if constexpr (RES::ON_SKIP3)
            fg_pin<WS, 32, 256, 256>(
                res + RES::SKIP3,
                w.skip() + 512 * 256 + g * 32 * FgVec<WS>::VEC, tid);
        // (the first barrier of the utterance loop orders these writes)
    }
    FgCluster c;
    c.buf = reinterpret_cast<unsigned long long*>(
        ca.state + (size_t)cluster * FG_CSTATE);
    c.error = ca.error;
    c.epoch = 0;
    c.g = g;

#pragma unroll 1
    for (int u0 = cluster * U; u0 < a.B; u0 += ca.nclusters * U) {
        // utterances u0 .. u0 + U - 1; slots past the batch replay the last
        // utterance (same arithmetic, stores masked) so every member of the
        // cluster runs the same number of exchanges
        int ut[U], len[U];
        bool live[U];
        int frames = 0;          // the longest of the U utterances
#pragma unroll
        for (int u = 0; u < U; ++u) {
            live[u] = u0 + u < a.B;
            ut[u] = live[u] ? u0 + u : a.B - 1;
            len[u] = a.lengths ? min(max(a.lengths[ut[u]], 0), T) : T;
            frames = len[u] > frames ? len[u] : frames;
            // zero the tail of a short utterance (each member its share)
            if (live[u])
                for (int i = len[u] * FG_HOP + g * NT + tid; i < T * FG_HOP;
                     i += FG_G * NT)
                    a.out[(size_t)ut[u] * T * FG_HOP + i] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            for (int i = tid; i < 3 * FG_HOP; i += NT) (&L[u].hid[0][0])[i] = 0.f;
            for (int i = tid; i < 3 * 96; i += NT) (&L[u].gh[0][0])[i] = 0.f;   // W_hh 0
            for (int i = tid; i < 2 * FG_SUBIN + 8; i += NT) L[u].subin[i] = 0.f;
            for (int i = tid; i < CPAD; i += NT) L[u].condin[i] = 0.f;
            for (int i = tid; i < 384; i += NT) { L[u].c1[i] = 0.f; L[u].c2[i] = 0.f; }
            for (int i = tid; i < FG_PREV; i += NT)
                L[u].prev[i] = a.previous
                    ? a.previous[(size_t)(a.previous_batch == 1 ? 0 : ut[u]) *
                                     FG_PREV + i]
                    : 0.f;
        }
        int base = 0;
        __syncthreads();

#pragma unroll 1
        for (int t = 0; t < frames; ++t) {
            // Opaque copy of the thread id, renewed every frame / step: the
            // per-thread weight addresses and K-slice bounds of the 13 matrix
            // slices below are loop invariants, and hoisted out of this
            // 3 444-step loop they occupy ~80 vector registers for the whole
            // kernel (spills); recomputing them costs a few VALU per slice.
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            int period[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float* row =
                    a.features_cl + ((size_t)ut[u] * T + t) * a.cstride;
                const float* glob = a.global +
                    (size_t)(a.global_batch == 1 ? 0 : ut[u]) * a.G;
                if (tid < a.nfeat) L[u].condin[tid] = row[tid];
                else if (tid < nin) L[u].condin[tid] = glob[tid - a.nfeat];
                period[u] = (int)rintf(row[a.nfeat]);
            }
            __syncthreads();
            float v[U], m[U], tot[U], ext[U];
            if constexpr (std::is_same<WS, float>::value) {
                // ---- conditioning network: computed for every frame by
                // pm_fargan_cond_kernel ahead of this launch ----
                if (tid < 512) {
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        L[u].cond[tid] =
                            ca.precond[((size_t)ut[u] * T + t) * 512 + tid];
                }
                __syncthreads();
            } else {
            // ---- conditioning network (fargan.py:139-160): R, K, R ----
            fg_slice<WS, 48, 384, U, CPAD, 0>(w.cond(0), lds, FG_OFF(condin),
                                           FG_OFF(condin), CPAD, g * 48, lds,
                                           tid, v);
            if (tid < 48) {
#pragma unroll
                for (int u = 0; u < U; ++u) L[u].c1[g * 48 + tid] = fg_tanh(v[u]);
            }
            __syncthreads();
            fg_slice<WS, 384, 384, U, 48, 1>(w.k_cond1(g), lds,
                                          FG_OFF(c1) + g * 48,
                                          FG_OFF(c1) + g * 48, 48, 0, lds, tid, v);
            fg_exchange_sum<U, 384, 0>(c, v, v, lds, tid, tot, ext);
            if (tid < 384) {
#pragma unroll
                for (int u = 0; u < U; ++u) L[u].c2[tid] = fg_tanh(tot[u]);
            }
            __syncthreads();
            fg_slice<WS, 64, 512, U, CPAD, 1>(w.cond(2), lds, FG_OFF(c2), FG_OFF(c2),
                                           CPAD, g * 64, lds, tid, v);
#pragma unroll
            for (int u = 0; u < U; ++u) m[u] = fg_tanh(v[u]);
            fg_exchange<U, 64>(c, m, lds, FG_OFF(cond), tid);
            }

#pragma unroll 1
            for (int s = 0; s < 4; ++s) {
                asm volatile("" : "+v"(tid));
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    FgLds& S = L[u];
                    if (tid < 128) {
                        S.subin[tid] = S.cond[4 * tid + s];
                    } else if (tid < 192) {
                        const int i = tid - 128;
                        const float x =
                            S.prev[(base + FG_PREV - FG_SUB + i) & (FG_PREV - 1)];
                        S.subin[128 + i] = x;
                        S.skipbuf[1088 + i] = x;
                    } else if (tid < 260) {
                        const int i = tid - 192;
                        int idx = FG_PREV - period[u] + i - 2;
                        if (idx >= FG_PREV) idx -= period[u];
                        idx = idx < 0 ? 0 : (idx >= FG_PREV ? FG_PREV - 1 : idx);
                        const float x = S.prev[(base + idx) & (FG_PREV - 1)];
                        S.subin[192 + i] = x;
                        if (i >= 2 && i < 66) S.skipbuf[1024 + i - 2] = x;
                    }
                }
                __syncthreads();
                FG_STAMP(0);

                // Half of a step's weight stream multiplies inputs that are
                // known BEFORE the layer in front of them has finished: W_hh h
                // (the previous step's state), the [lookback | previous
                // subframe] columns of W_ih, and all of the skip dense layer but
                // the last GRU's columns. Those products are streamed UNDER the
                // exchanges - between publishing this member's granules and
                // polling for the others', reduced inside their waves, the
                // rows' lead lanes storing gh / gil / skpre themselves -
                // so that a granule round trip carries ~1 us of weight stream
                // instead of a poll loop and the slices on the dependency chain
                // shrink to the columns that really wait. Schedule (level 2):
                //   E1 fwconv GLU   W_ih[0], W_ih[1] lookback columns
                //   E2 GRU 0        W_hh[2] h
                //   E3 GRU 1        W_ih[2] lookback columns, skip [fw|lb|prev]
                //   E4 GRU 2        skip [g0 | g1]
                //   E5 skip vector  W_hh[0] h   (for the NEXT step: h is final)
                //   E6 output       W_hh[1] h   (for the NEXT step)
                // (level 1: W_hh[n] h under the exchange in front of cell n)
                constexpr int LVL = FG_UNDER(WT);
                // lanes a row of the 96-row GRU slices: all FG_CT threads
                constexpr int GLPR = FG_CT >= 768 ? 8 : 4;
                auto under_hh = [&](int n) __attribute__((always_inline)) {
                    float unused[U];
                    // (opaque thread id again: without it this slice's weight
                    // addresses are computed ahead of the exchange and the
                    // kernel sits at the 168-register limit of 768 threads;
                    // with it 69 - and 3-4 % faster, profiles/r04/ab_fargan_tail.txt;
                    // all-f16 storage measured 1-2 % slower that way and keeps
                    // the hoisted form)
                    if constexpr (sizeof(WS) == 4 || U > 1)
                        asm volatile("" : "+v"(tid));
                    const int hoff = FG_OFF(hid) + n * FG_HOP;
                    // (reduced inside the wave too: the lead lane of a row
                    // writes gh[n] itself - nobody reads it before the
                    // exchange's barrier - so there is nothing to collect)
                    fg_slice_lanes<WI, 96, 768 * FgVec<WI>::VEC, U, 256, GLPR>(
                        w.gru_hh(n) + g * 96 * FgVec<WI>::VEC, lds, hoff, hoff,
                        256, tid, unused);
                    if (FgLanes<GLPR>::lead(tid, 96)) {
#pragma unroll
                        for (int u = 0; u < U; ++u)
                            L[u].gh[n][FgLanes<GLPR>::row(tid)] = unused[u];
                    }
                };
                auto under_ih = [&](int n) __attribute__((always_inline)) {
                    float unused[U];
                    fg_slice_lanes<WI, 96, 768 * FgVec<WI>::VEC, U, 128, GLPR>(
                        w.gru_ih(n) + 256 * 768 + g * 96 * FgVec<WI>::VEC, lds,
                        FG_OFF(skipbuf) + 1024, FG_OFF(skipbuf) + 1024, 128, tid,
                        unused);
                    if (FgLanes<GLPR>::lead(tid, 96)) {
#pragma unroll
                        for (int u = 0; u < U; ++u)
                            L[u].gil[n][FgLanes<GLPR>::row(tid)] = unused[u];
                    }
                };

                // ---- framewise conv (R) + its GLU gate (K): 1 exchange ----
                fg_slice_lanes<WS, 32, 256 * FgVec<WS>::VEC, U, 520, 16>(
                    w.fwconv() + g * 32 * FgVec<WS>::VEC, lds, FG_OFF(subin),
                    FG_OFF(subin), 520, tid, v);
                FG_STAMP(1);
                if (FgLanes<16>::lead(tid, 32)) {
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        L[u].own[FgLanes<16>::row(tid)] = fg_tanh(v[u]);
                }
                __syncthreads();
                // (the member's slice travels with the partial sums below)
#pragma unroll
                for (int u = 0; u < U; ++u)
                    m[u] = tid < 32 ? L[u].own[tid] : 0.f;
                fg_slice_pair<WI, 256, U>(
                    RES::ON_FWGLU ? reinterpret_cast<const WI*>(res + RES::FWGLU)
                            : w.k_fwconv_glu(g),
                    lds, FG_OFF(own), tid, v);
                FG_STAMP(2);
                if constexpr (LVL >= 2) {
                    fg_exchange_sum<U, 256, 32, true>(
                        c, v, m, lds, tid, tot, ext, [&]() {
                            under_ih(0); under_ih(1); });
                } else if constexpr (LVL == 1) {
                    fg_exchange_sum<U, 256, 32, true>(
                        c, v, m, lds, tid, tot, ext,
                        [&]() { under_hh(0); });
                } else {
                    fg_exchange_sum<U, 256, 32, true>(c, v, m, lds, tid, tot, ext);
                }
                FG_STAMP(3);
                if (tid < 256) {
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        L[u].skipbuf[768 + tid] = ext[u] * fg_sigmoid(tot[u]);
                }
                __syncthreads();

                // ---- three GRU cells (R) + GLU gates (K): 1 exchange each ----
#pragma unroll 1
                for (int n = 0; n < 3; ++n) {
                    const int xa = n == 0 ? FG_OFF(skipbuf) + 768
                                          : FG_OFF(skipbuf) + (n - 1) * 256;
                    // this member's 32 units x 3 gates = packed rows g*96 ..
                    float gi[U];
                    if constexpr (LVL >= 2)
                        fg_slice_lanes<WI, 96, 768 * FgVec<WI>::VEC, U, 256, GLPR>(
                            w.gru_ih(n) + g * 96 * FgVec<WI>::VEC, lds, xa, xa,
                            256, tid, gi);
                    else
                        fg_slice_lanes<WI, 96, 768 * FgVec<WI>::VEC, U, 384, GLPR>(
                            w.gru_ih(n) + g * 96 * FgVec<WI>::VEC, lds, xa,
                            FG_OFF(skipbuf) + 1024, 256, tid, gi);
                    if constexpr (LVL == 0) {
                        float ghv[U];
                        const int hoff = FG_OFF(hid) + n * FG_HOP;
                        fg_slice<WI, 96, 768, U, 256, 0>(
                            w.gru_hh(n), lds, hoff, hoff, 256, g * 96, lds, tid,
                            ghv);
                        if (tid < 96) {
#pragma unroll
                            for (int u = 0; u < U; ++u) L[u].gh[n][tid] = ghv[u];
                        }
                    }
                    FG_STAMP(4 + 4 * n);
                    if (FgLanes<GLPR>::lead(tid, 96)) {
                        const int row = FgLanes<GLPR>::row(tid);
#pragma unroll
                        for (int u = 0; u < U; ++u)
                            L[u].part2[row] = LVL >= 2
                                ? gi[u] + L[u].gil[n][row] : gi[u];
                    }
                    if (tid < 96) {
#pragma unroll
                        for (int u = 0; u < U; ++u)
                            L[u].part2[96 + tid] = L[u].gh[n][tid];
                    }
                    __syncthreads();
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        float hnew = 0.f;
                        if (tid < 32) {
                            const float* p2 = L[u].part2;
                            const float r = fg_sigmoid(p2[tid] + p2[96 + tid]);
                            const float z = fg_sigmoid(p2[32 + tid] + p2[128 + tid]);
                            const float nn = fg_tanh(p2[64 + tid] + r * p2[160 + tid]);
                            hnew = (1.f - z) * nn + z * L[u].hid[n][g * 32 + tid];
                            L[u].own[tid] = hnew;
                        }
                        m[u] = hnew;
                    }
                    __syncthreads();
                    FG_STAMP(5 + 4 * n);
                    fg_slice_pair<WI, 256, U>(
                        RES::ON_GRUGLU ? reinterpret_cast<const WI*>(
                                      res + RES::GRUGLU +
                                      n * 256 * 32 * sizeof(WI))
                                : w.k_gru_glu(n, g),
                        lds, FG_OFF(own), tid, v);
                    FG_STAMP(6 + 4 * n);
                    if (LVL >= 2 && n == 0) {
                        fg_exchange_sum<U, 256, 32, true>(
                            c, v, m, lds, tid, tot, ext,
                            [&]() { under_hh(2); });
                    } else if (LVL >= 2 && n == 1) {
                        // skip dense layer (fargan.py:317-322), columns
                        // [fwconv | lookback | previous]: known since E1
                        fg_exchange_sum<U, 256, 32, true>(
                            c, v, m, lds, tid, tot, ext, [&]() {
                                under_ih(2);
                                float sb[U];
                                fg_slice_lanes<WS, 32, 256 * FgVec<WS>::VEC, U,
                                               384, 16>(
                                    w.skip() + 768 * 256 +
                                        g * 32 * FgVec<WS>::VEC,
                                    lds, FG_OFF(skipbuf) + 768,
                                    FG_OFF(skipbuf) + 768, 384, tid, sb);
                                if (FgLanes<16>::lead(tid, 32)) {
#pragma unroll
                                    for (int u = 0; u < U; ++u)
                                        L[u].skpre[FgLanes<16>::row(tid)] = sb[u];
                                }
                            });
                    } else if (LVL >= 2) {
                        // ... and columns [g0 | g1]
                        fg_exchange_sum<U, 256, 32, true>(
                            c, v, m, lds, tid, tot, ext, [&]() {
                                // (the same lead lanes as under E3: a
                                // read-modify-write of their own element)
                                float sa[U];
                                fg_slice_lanes<WS, 32, 256 * FgVec<WS>::VEC, U,
                                               512, 16>(
                                    w.skip() + g * 32 * FgVec<WS>::VEC, lds,
                                    FG_OFF(skipbuf), FG_OFF(skipbuf), 512, tid,
                                    sa);
                                if (FgLanes<16>::lead(tid, 32)) {
#pragma unroll
                                    for (int u = 0; u < U; ++u)
                                        L[u].skpre[FgLanes<16>::row(tid)] += sa[u];
                                }
                            });
                    } else if (LVL == 1 && n < 2) {
                        fg_exchange_sum<U, 256, 32, true>(
                            c, v, m, lds, tid, tot, ext,
                            [&]() { under_hh(n + 1); });
                    } else {
                        fg_exchange_sum<U, 256, 32, true>(c, v, m, lds, tid, tot, ext);
                    }
                    FG_STAMP(7 + 4 * n);
                    if (tid < 256) {
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            L[u].hid[n][tid] = ext[u];
                            L[u].skipbuf[n * 256 + tid] =
                                ext[u] * fg_sigmoid(tot[u]);
                        }
                    }
                    __syncthreads();
                }

                // ---- skip dense (R): vector exchange ----
                FG_STAMP(16);
                if constexpr (LVL >= 2 && RES::ON_SKIP3)
                    fg_slice_lanes<WS, 32, 32 * FgVec<WS>::VEC, U, 256, 8>(
                        reinterpret_cast<const WS*>(res + RES::SKIP3), lds,
                        FG_OFF(skipbuf) + 512, FG_OFF(skipbuf) + 512, 256, tid,
                        v);
                else if constexpr (LVL >= 2)   // (only the last GRU's columns are left)
                    fg_slice_lanes<WS, 32, 256 * FgVec<WS>::VEC, U, 256, 8>(
                        w.skip() + 512 * 256 + g * 32 * FgVec<WS>::VEC, lds,
                        FG_OFF(skipbuf) + 512, FG_OFF(skipbuf) + 512, 256, tid,
                        v);
                else
                    fg_slice_lanes<WS, 32, 256 * FgVec<WS>::VEC, U, FG_SKIP, 8>(
                        w.skip() + g * 32 * FgVec<WS>::VEC, lds, FG_OFF(skipbuf),
                        FG_OFF(skipbuf), FG_SKIP, tid, v);
                FG_STAMP(17);
                {
                    const int row = FgLanes<8>::row(tid) & 31;
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        m[u] = fg_tanh(LVL >= 2 ? v[u] + L[u].skpre[row] : v[u]);
                }
                if constexpr (LVL >= 2) {
                    fg_exchange<U, 32, 8>(c, m, lds, FG_OFF(f1), tid,
                                          [&]() { under_hh(0); });
                } else {
                    fg_exchange<U, 32, 8>(c, m, lds, FG_OFF(f1), tid);
                }
                FG_STAMP(18);
                // ---- skip GLU (R) + output layer (K): 1 exchange ----
                if constexpr (RES::ON_SKIPGLU)
                    fg_slice_lanes<WI, 32, 32 * FgVec<WI>::VEC, U, 256, 8>(
                        reinterpret_cast<const WI*>(res + RES::SKIPGLU), lds,
                        FG_OFF(f1), FG_OFF(f1), 256, tid, v);
                else
                    fg_slice_lanes<WI, 32, 256 * FgVec<WI>::VEC, U, 256, 8>(
                        w.skip_glu() + g * 32 * FgVec<WI>::VEC, lds, FG_OFF(f1),
                        FG_OFF(f1), 256, tid, v);
                if (FgLanes<8>::lead(tid, 32)) {
                    const int row = FgLanes<8>::row(tid);
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        L[u].own[row] =
                            L[u].f1[g * 32 + row] * fg_sigmoid(v[u]);
                }
                __syncthreads();
                FG_STAMP(19);
                fg_slice_pair<WS, 64, U>(
                    RES::ON_OUT ? reinterpret_cast<const WS*>(res + RES::OUT)
                            : w.k_out(g),
                    lds, FG_OFF(own), tid, v);
                FG_STAMP(20);
                if constexpr (LVL >= 2) {
                    fg_exchange_sum<U, 64, 0, true>(c, v, v, lds, tid, tot, ext,
                                              [&]() { under_hh(1); });
                } else {
                    fg_exchange_sum<U, 64, 0, true>(c, v, v, lds, tid, tot, ext);
                }
                FG_STAMP(21);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    FgLds& S = L[u];
                    if (tid < FG_SUB) {
                        const float sample = fg_tanh(tot[u]);
                        if (tid / 8 == g && live[u] && t < len[u])
                            a.out[((size_t)ut[u] * T + t) * FG_HOP + s * FG_SUB +
                                  tid] = sample;
                        S.prev[(base + tid) & (FG_PREV - 1)] = sample;
                    } else if (tid < FG_SUB + FG_SUBIN) {
                        S.subin[FG_SUBIN + tid - FG_SUB] = S.subin[tid - FG_SUB];
                    }
                }
                base = (base + FG_SUB) & (FG_PREV - 1);
                __syncthreads();
                FG_STAMP(22);
            }
        }
    }
}
