// Shared device helpers for the promonet MI355X (gfx950 / CDNA4) kernels.
//
// Layout convention used by every kernel in this library: activations are
// CHANNELS-LAST, (B, L, C) with C contiguous and padded to a multiple of 32.
// A time tile of TL positions x C channels is one contiguous run in HBM
// (coalesced), a convolution tap / dilation is a pure ROW offset, and one
// lane's MFMA operand (8 consecutive input channels at one time position)
// is one 16-byte LDS read.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PM_LRELU_SLOPE 0.1f   // promonet/config/defaults.py:216

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned int pm_u4 __attribute__((ext_vector_type(4)));   // buffer_load/store_b128
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// max(v, 0.1 v) == (v > 0 ? v : 0.1 v) for every finite v (and keeps -0):
// one v_max instead of v_cmp + v_cndmask (the conv TUs are built with
// -fno-honor-nans so that no canonicalising v_max v,v is added)
__device__ __forceinline__ float pm_lrelu(float v) {
    return __builtin_fmaxf(v, v * PM_LRELU_SLOPE);
}
// four at once on vector types: two v_pk_mul_f32 + four v_max_f32
typedef float pm_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 pm_lrelu4(float4 v) {
    pm_f4 w = {v.x, v.y, v.z, v.w};
    w = __builtin_elementwise_max(w, w * PM_LRELU_SLOPE);
    return make_float4(w.x, w.y, w.z, w.w);
}

// ---------------------------------------------------------------------------
// Element traits: the MFMA operand type. One "k16 step" contracts 16 input
// channels for a 32(co) x 32(time) tile:
//   F16 / BF16 : one v_mfma_f32_32x32x16_{f16,bf16}      (fp32 accumulate)
//   F32        : eight v_mfma_f32_32x32x2_f32            (exact fp32)
// In all three the lane (l) holds, for row/col (l & 31), the 8 consecutive
// k-indices (l >> 5) * 8 + e, e = 0..7 - identical for A and B, so the sum
// over k is complete whatever order the hardware consumes it in.
// ---------------------------------------------------------------------------
// What every single-value element type shares: a row of an LDS operand tile is
// `channels * ESZ` bytes with channel c at byte c * ESZ, a packed weight stream
// an array of lds_t. (ElemF16X3 below interleaves hi and lo parts instead.)
#define PM_ELEM_COMMON(bias_split)                                            \
    static constexpr bool BIAS_SPLIT = bias_split;                            \
    /* four channels ch .. ch + 3 (ch % 4 == 0) of an LDS row */              \
    __device__ static __forceinline__ void store4_at(                         \
        char* rowp, int ch, float4 v) {                                       \
        store4(rowp + ch * ESZ, v);                                           \
    }                                                                         \
    /* element `index` of a packed weight stream */                           \
    __device__ static __forceinline__ void pack_store(                        \
        void* out, long long index, float v) {                                \
        reinterpret_cast<lds_t*>(out)[index] = cvt(v);                        \
    }

struct ElemF16 {
    typedef _Float16 lds_t;
    typedef half8 frag_t;
    typedef frag_t afrag_t;     // A operand (weights), as streamed from memory
    typedef frag_t bfrag_t;     // B operand (activations), as read from LDS
    static constexpr int ESZ = 2;       // bytes per element of an LDS row
    static constexpr int WSZ = 2;       // bytes per element of a weight stream
    static constexpr int ID = 1;
    static constexpr bool SPLIT = false;    // activations stored as hi + lo
    __device__ static __forceinline__ void mma(
        const frag_t& a, const frag_t& b, floatx16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    __device__ static __forceinline__ void store4(char* p, float4 v) {
        // vector conversion: two v_cvt_pk_f16_f32 (round to nearest even),
        // then saturation at the top of the f16 range (two v_pk_min_f16): an
        // activation beyond 65504 (a trained checkpoint may produce one where
        // the random-init weights of the tests do not) becomes 65504 instead
        // of +inf, so it cannot poison the MFMA sums. Every operand but the
        // conditioning features passes LeakyReLU first, which shrinks the
        // negative side tenfold (-inf only below -655 040): the lower clamp
        // is left out - both clamps cost 1.6 % of the step, this one half
        // (profiles/r02/ab_f16_saturate.txt).
        const pm_f4 w = {v.x, v.y, v.z, v.w};
        half4 h = __builtin_convertvector(w, half4);
#ifndef PM_NO_F16_SATURATE
        const _Float16 big = (_Float16)65504.f;
        const half4 hi = {big, big, big, big};
        h = __builtin_elementwise_min(h, hi);
#endif
        *reinterpret_cast<half4*>(p) = h;
    }
    // four fp32 -> four operand values in two dwords (saturating, as store4)
    __device__ static __forceinline__ uint2 pack4(float4 v) {
        uint2 r;
        store4(reinterpret_cast<char*>(&r), v);
        return r;
    }
    __device__ static __forceinline__ lds_t cvt(float v) { return (_Float16)v; }
    PM_ELEM_COMMON(true)
    // bias step (pm_pack_bias_step_kernel): c += b[co] for every column
    __device__ static __forceinline__ void mma_bias(
        const frag_t& a, floatx16& c) {
        // (an opaque constant, materialised at the use: as a loop invariant
        // of a long kernel the four registers of the fragment get spilled)
        unsigned pair = 0x3c003c00u;            // {1.0h, 1.0h}
        asm volatile("" : "+v"(pair));
        const pm_u4 bits = {pair, pair, pair, pair};
        const frag_t ones = __builtin_bit_cast(frag_t, bits);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, ones, c, 0, 0, 0);
    }
};

struct ElemBF16 {
    typedef __bf16 lds_t;
    typedef bf16x8 frag_t;
    typedef frag_t afrag_t;
    typedef frag_t bfrag_t;
    static constexpr int ESZ = 2;
    static constexpr int WSZ = 2;
    static constexpr int ID = 2;
    static constexpr bool SPLIT = false;
    __device__ static __forceinline__ void mma(
        const frag_t& a, const frag_t& b, floatx16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    __device__ static __forceinline__ void store4(char* p, float4 v) {
        const pm_f4 w = {v.x, v.y, v.z, v.w};
        *reinterpret_cast<bf16x4*>(p) = __builtin_convertvector(w, bf16x4);
    }
    __device__ static __forceinline__ uint2 pack4(float4 v) {
        const pm_f4 w = {v.x, v.y, v.z, v.w};
        return __builtin_bit_cast(uint2, __builtin_convertvector(w, bf16x4));
    }
    __device__ static __forceinline__ lds_t cvt(float v) { return (__bf16)v; }
    PM_ELEM_COMMON(true)
    __device__ static __forceinline__ void mma_bias(
        const frag_t& a, floatx16& c) {
        unsigned pair = 0x3f803f80u;            // {1.0bf16, 1.0bf16}
        asm volatile("" : "+v"(pair));
        const pm_u4 bits = {pair, pair, pair, pair};
        const frag_t ones = __builtin_bit_cast(frag_t, bits);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, ones, c, 0, 0, 0);
    }
};

struct ElemF32 {
    typedef float lds_t;
    struct frag_t { float4 lo, hi; };
    typedef frag_t afrag_t;
    typedef frag_t bfrag_t;
    static constexpr int ESZ = 4;
    static constexpr int WSZ = 4;
    static constexpr int ID = 0;
    static constexpr bool SPLIT = false;
    __device__ static __forceinline__ void mma(
        const frag_t& a, const frag_t& b, floatx16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo.x, b.lo.x, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo.y, b.lo.y, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo.z, b.lo.z, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo.w, b.lo.w, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi.x, b.hi.x, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi.y, b.hi.y, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi.z, b.hi.z, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi.w, b.hi.w, c, 0, 0, 0);
    }
    __device__ static __forceinline__ void store4(char* p, float4 v) {
        *reinterpret_cast<float4*>(p) = v;
    }
    __device__ static __forceinline__ lds_t cvt(float v) { return v; }
    PM_ELEM_COMMON(false)
    // k = 0 carries the bias (lanes 0-31), k = 1 (lanes 32-63) is zero
    __device__ static __forceinline__ void mma_bias(
        const frag_t& a, floatx16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo.x, 1.f, c, 0, 0, 0);
    }
};

// f16 operands SPLIT into hi + lo (v = hi + lo up to 2^-22 |v|): a k16 step is
// THREE v_mfma_f32_32x32x16_f16 - hi x hi, lo x hi, hi x lo (the lo x lo term,
// 2^-22 of the product, is dropped) - so a product carries ~21 bits of each
// factor at 3/16 of the cost of the exact fp32 MFMA path. It exists for the
// LAST upsampling stage of a trained checkpoint: there one f16 rounding of the
// activations alone is 3e-4 of an output that peaks near 1 (DESIGN.md
// section 3). Storage: the 8 consecutive k-elements of a fragment are 32
// bytes, [8 x hi | 8 x lo] - the same 4 bytes per element as ElemF32, so every
// tiling written for ESZ == 4 carries over.
struct ElemF16X3 {
    typedef _Float16 lds_t;
    struct frag_t { half8 hi, lo; };
    typedef frag_t afrag_t;
    typedef frag_t bfrag_t;
    static constexpr int ESZ = 4;
    static constexpr int WSZ = 4;
    static constexpr int ID = 3;
    static constexpr bool SPLIT = true;
    static constexpr bool BIAS_SPLIT = true;
    __device__ static __forceinline__ void mma(
        const frag_t& a, const frag_t& b, floatx16& c) {
#if !(defined(PM_TUNING) && defined(PM_X3_TWO_MFMA))   // (timing experiment: two of the three)
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.lo, b.hi, c, 0, 0, 0);
#endif
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, b.lo, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, b.hi, c, 0, 0, 0);
    }
    // hi = rn(v'), lo = rn(v' - hi) with v' = v clamped to the f16 range IN
    // FP32 (one v_med3_f32 a value): |v| beyond 65504 becomes +-65504 with a
    // zero lo part, never an infinity (-inf x a zero-padded weight would be NaN
    // in the MFMA sum), and neither part needs a clamp of its own (round 5
    // clamped both in f16: 8 packed min / max per four values; v' - hi is at
    // most half an ulp of hi). The remainder is ONE v_fma_mix_f32 a value -
    // the f16 hi part is an operand as it is, not converted back first.
    __device__ static __forceinline__ void split4(
        float4 v, half4& hi, half4& lo) {
#ifdef PM_SPLIT_R05         // (A/B builds: round 5's form)
        const pm_f4 w = {v.x, v.y, v.z, v.w};
        const _Float16 big = (_Float16)65504.f;
        const half4 top = {big, big, big, big};
        const half4 bottom = {-big, -big, -big, -big};
        hi = __builtin_elementwise_max(__builtin_elementwise_min(
            __builtin_convertvector(w, half4), top), bottom);
        const pm_f4 rest = w - __builtin_convertvector(hi, pm_f4);
        lo = __builtin_elementwise_max(__builtin_elementwise_min(
            __builtin_convertvector(rest, half4), top), bottom);
#else
        const float w[4] = {
            __builtin_fminf(__builtin_fmaxf(v.x, -65504.f), 65504.f),
            __builtin_fminf(__builtin_fmaxf(v.y, -65504.f), 65504.f),
            __builtin_fminf(__builtin_fmaxf(v.z, -65504.f), 65504.f),
            __builtin_fminf(__builtin_fmaxf(v.w, -65504.f), 65504.f)};
        const pm_f4 wv = {w[0], w[1], w[2], w[3]};
        hi = __builtin_convertvector(wv, half4);
        // (inline asm: written as fmaf((float)hi, -1, w) hipcc folds the
        // product away and converts hi back with v_cvt_f32_f16 + v_sub_f32)
        const uint2 hp = __builtin_bit_cast(uint2, hi);
        pm_f4 rest;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]"
            : "=v"(rest[0]) : "v"(hp.x), "v"(w[0]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
            : "=v"(rest[1]) : "v"(hp.x), "v"(w[1]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]"
            : "=v"(rest[2]) : "v"(hp.y), "v"(w[2]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
            : "=v"(rest[3]) : "v"(hp.y), "v"(w[3]));
        lo = __builtin_convertvector(rest, half4);
#endif
    }
    // channels ch .. ch + 3 of an LDS row: the 8-channel group g = ch / 8 is 32
    // bytes, hi parts first; ch % 8 selects the half of both parts
    __device__ static __forceinline__ void store4_at(
        char* rowp, int ch, float4 v) {
        half4 hi, lo;
        split4(v, hi, lo);
        char* p = rowp + (ch >> 3) * 32 + (ch & 4) * 2;
        *reinterpret_cast<half4*>(p) = hi;
        *reinterpret_cast<half4*>(p + 16) = lo;
    }
    __device__ static __forceinline__ void pack_store(
        void* out, long long index, float v) {
        const _Float16 hi = (_Float16)v;
        _Float16* p = reinterpret_cast<_Float16*>(out) +
                      (index >> 3) * 16 + (index & 7);
        p[0] = hi;
        p[8] = (_Float16)(v - (float)hi);
    }
    __device__ static __forceinline__ lds_t cvt(float v) { return (_Float16)v; }
    // the bias step's fragment carries hi(b), lo(b) in k = 0, 1 of its hi part
    // (pack_store of the two values; the lo part of those is zero or tiny)
    __device__ static __forceinline__ void mma_bias(
        const frag_t& a, floatx16& c) {
        unsigned pair = 0x3c003c00u;            // {1.0h, 1.0h}
        asm volatile("" : "+v"(pair));
        const pm_u4 bits = {pair, pair, pair, pair};
        const half8 ones = __builtin_bit_cast(half8, bits);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, ones, c, 0, 0, 0);
    }
};

// f16 operands with the ACTIVATIONS split into hi + lo and the weights rounded
// once: a k16 step is TWO v_mfma_f32_32x32x16_f16 - w x hi, w x lo - sharing one
// (16-byte) weight fragment, so the weight stream is ElemF16's, the LDS tiles
// are ElemF16X3's. What it keeps of the split's benefit is the part that
// matters in the Blocks of a trained checkpoint's last stage (the activation
// rounding: scripts/precision_emulate.py), for two thirds
// of the MFMAs and half the weight bytes.
struct ElemF16A2 {
    typedef _Float16 lds_t;
    typedef half8 afrag_t;
    typedef ElemF16X3::frag_t bfrag_t;
    typedef afrag_t frag_t;             // (the weight stream's unit)
    static constexpr int ESZ = 4;
    static constexpr int WSZ = 2;
    static constexpr int ID = 4;
    static constexpr bool SPLIT = true;
    static constexpr bool BIAS_SPLIT = true;
    __device__ static __forceinline__ void mma(
        const afrag_t& a, const bfrag_t& b, floatx16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b.lo, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b.hi, c, 0, 0, 0);
    }
    __device__ static __forceinline__ void store4_at(
        char* rowp, int ch, float4 v) {
        ElemF16X3::store4_at(rowp, ch, v);
    }
    __device__ static __forceinline__ void pack_store(
        void* out, long long index, float v) {
        reinterpret_cast<_Float16*>(out)[index] = (_Float16)v;
    }
    __device__ static __forceinline__ lds_t cvt(float v) { return (_Float16)v; }
    __device__ static __forceinline__ void mma_bias(
        const afrag_t& a, floatx16& c) {
        ElemF16::mma_bias(a, c);
    }
};

// Workgroup barrier for kernels whose threads talk through LDS only.
// __syncthreads() fences every address space: the compiler puts
// s_waitcnt vmcnt(0) in front of the s_barrier, i.e. every global load in
// flight (the next weight group, the next staged chunk, the residual) has to
// land before the wave may even ARRIVE. Fencing the LDS alone keeps those
// loads in flight across the barrier.
__device__ __forceinline__ void pm_block_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Bijective XCD-aware remap of a linear workgroup id: the dispatcher places
// block b on XCD b % 8; give each XCD a contiguous run of tiles so that
// neighbouring time tiles (which share halo rows) hit the same L2.
__device__ __forceinline__ int pm_xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7;
    const int q = nwg >> 3, r = nwg & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}
