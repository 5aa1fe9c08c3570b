// MFMA implicit-GEMM 1-D convolutions for the HiFi-GAN generator
// (reference: promonet/model/hifigan.py). Hand-written for gfx950.
//
//   out[co, t] = b[co] + sum_{ci, j} w[co, ci, j] * act(x[ci, t + (j - (k-1)/2) d])
//
// GEMM view per workgroup: M = output channels (A operand = weights, streamed
// from L2 straight into registers in a pre-packed fragment layout),
// N = time (B operand = activation rows staged once in LDS with the dilated
// halo, LeakyReLU + fp32->operand-type conversion fused into the staging
// write), K = (input channel, tap). A tap / dilation is a row offset into the
// LDS tile, so the tile is read k times from LDS and once from HBM.
//
// Kernels:
//   conv_pair_kernel     one HiFi-GAN `Block` iteration fused:
//                        y = x + conv2(lrelu(conv1(lrelu(x))))     hifigan.py:198-210
//                        (conv1 output never leaves LDS; the fp32 residual is
//                        loaded into the conv1 accumulators and conv2
//                        accumulates onto it), optional MRF accumulation
//                        epilogue (xs / 3, hifigan.py:141-145); C = 128, 256
//   conv_block3_kernel   a whole `Block` (all dilations) with the fp32 trunk in
//                        registers; C <= 64, and C = 128 at k 3
//   conv_mrf_kernel      a whole MRF stage (Blocks k 3, 7, 11) with the sum in
//                        registers; C = 32
//   conv_single_kernel   generic C_in -> M conv with per-M-tile tap windows:
//                        the input conv (hifigan.py:19-24), the narrow
//                        ConvTranspose1d upsamplers as r-phase polyphase GEMMs
//                        (hifigan.py:100-106), the framed DFT of the STFT
//   conv_upsample_kernel the wide (r = 8) upsamplers: x tile staged once, all M
//                        blocks walked by one workgroup
// Every access that can meet an utterance edge or a tile's halo goes through a
// buffer descriptor whose range is exactly the rows that exist (loads return
// 0, stores are dropped): no per-lane branches, exact vmcnt arithmetic.
// Workgroup barriers fence LDS only (pm_block_sync), so global loads stay in
// flight across them.
#pragma once
#include <type_traits>

#include "pm_common.h"

// ---------------------------------------------------------------------------
// Packed weight layout (built once at load by pm_pack_weights_kernel):
//   frag[(((mt * NCH + c) * KT + jj) * KC + kc) * 64 + lane] = 8 elements:
//     co = mt*32 + (lane & 31), ci = c*CH + kc*16 + (lane >> 5)*8 + e, tap jj
// so one wave's A operand for one k16 step is 64 contiguous fragments.
// MRF convolutions append one "bias step" (64 fragments) to every M tile's
// stream (pm_pack_bias_step_kernel): the bias is added by one MFMA against an
// all-ones B fragment, so an accumulator starts as `0 -> mfma(bias)` (no
// register fill, no add in the epilogue) and a residual accumulates in place
// (the fp32 trunk is the MFMA's C operand).
// ---------------------------------------------------------------------------

template <class ET, int CH, int NT, int XR_MAX>
struct ChunkStager {
    static constexpr int Q = CH / 4;  // float4 per row-chunk
    static constexpr int MAXIT = (XR_MAX * Q + NT - 1) / NT;
    static constexpr int S = CH * ET::ESZ + 16;
    float4 r[MAXIT];

    // Global (fp32, channels-last) -> registers. Rows outside [0, L) are the
    // convolution's zero padding (true sequence ends only): buffer loads
    // over a descriptor of exactly the rows that exist, so a row outside
    // reads as zero without a branch (its offset - negative rows wrap - is
    // out of range). Unconditional loads also keep the compiler's vmcnt
    // arithmetic exact: behind per-lane branches it cannot tell how many
    // loads were issued and waits for (almost) everything in flight.
    __device__ __forceinline__ void load(
        const float* __restrict__ xb, int cstride, int c0, int t_first,
        int XR, int L, int tid) {
        const int lo = max(t_first, 0);
        const int hi = min(L, t_first + XR);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(xb) + (size_t)lo * cstride, 0,
            max(hi - lo, 0) * cstride * 4, 0x00020000);
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int idx = tid + it * NT;
            const int row = idx / Q, q = idx % Q;
            const unsigned voff = (unsigned)(
                ((t_first - lo + row) * cstride + c0 + q * 4) * 4);
            const pm_u4 v =
                __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
            r[it] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y),
                                __uint_as_float(v.z), __uint_as_float(v.w));
        }
    }

    // The same tile from a tensor that already holds the operand values
    // (16-bit, activation applied by the kernel that produced it - the
    // skewed whole-Block walk's `act16` output): half the bytes, no VALU.
    static constexpr int Q16 = CH / 8;  // 16-byte pieces per row-chunk
    static constexpr int MAXIT16 = (XR_MAX * Q16 + NT - 1) / NT;
    static_assert(MAXIT16 <= MAXIT, "register budget of the 16-bit path");
    __device__ __forceinline__ void load16(
        const void* __restrict__ xb16, int cstride, int c0, int t_first,
        int XR, int L, int tid) {
        const int lo = max(t_first, 0);
        const int hi = min(L, t_first + XR);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(xb16)) +
                (size_t)lo * cstride * 2, 0,
            max(hi - lo, 0) * cstride * 2, 0x00020000);
#pragma unroll
        for (int it = 0; it < MAXIT16; ++it) {
            const int idx = tid + it * NT;
            const int row = idx / Q16, q = idx % Q16;
            const unsigned voff = (unsigned)(
                ((t_first - lo + row) * cstride + c0 + q * 8) * 2);
            const pm_u4 v =
                __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
            r[it] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y),
                                __uint_as_float(v.z), __uint_as_float(v.w));
        }
    }
    template <int STRIDE = S>
    __device__ __forceinline__ void store16(char* buf, int XR, int tid) {
#pragma unroll
        for (int it = 0; it < MAXIT16; ++it) {
            const int idx = tid + it * NT;
            const int row = idx / Q16, q = idx % Q16;
            if (row < XR)
                *reinterpret_cast<float4*>(buf + row * STRIDE + q * 16) = r[it];
        }
    }

    // Registers -> LDS with the input activation and operand conversion fused
    // (STRIDE: LDS row pitch, when the chunk is a slab of a wider tile)
    template <bool LRELU, int STRIDE = S>
    __device__ __forceinline__ void store(char* buf, int XR, int tid) {
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int idx = tid + it * NT;
            const int row = idx / Q, q = idx % Q;
            if (row < XR) {
                ET::store4_at(buf + row * STRIDE, q * 4,
                              LRELU ? pm_lrelu4(r[it]) : r[it]);
            }
        }
    }
};

// All taps x all k16-steps of one staged channel chunk for this wave's
// MTW x NTW grid of 32x32 tiles, software pipelined in registers: A
// fragments one group (G steps) ahead, B fragments one step ahead.
// Load the first A group of a weight stream (issued early, e.g. before a
// barrier, so its L2 latency is off the critical path of mma_taps()).
template <class ET, int MTW, int G>
__device__ __forceinline__ void load_a_group(
    typename ET::afrag_t (&dst)[G][MTW],
    const typename ET::afrag_t* __restrict__ wptr, const int w_mt_stride) {
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
            dst[g][mt] = wptr[mt * w_mt_stride + g * 64];
}

// `first` holds group 0 of this call's weight stream on entry (see
// load_a_group); when `wnext` is non-null, group 0 of the NEXT call's stream
// is fetched during the last group and returned in `first`.
struct NoHook { __device__ __forceinline__ void operator()() const {} };

// `mid` runs once, half way through the groups: the place for work that only
// needs loads issued before this call (they have landed by then: vmcnt retires
// in order and every group waits on younger A loads) and whose VALU / LDS
// writes should overlap the partner wave's MFMAs rather than sit between two
// barriers - the fp32 -> operand conversion + LDS write of the next chunk.
#ifndef PM_BDEPTH
#define PM_BDEPTH 2     // B fragments in flight: this many k16 steps' worth
#endif
template <class ET, int KT, int KC, int MTW, int NTW, int G, int S,
          class Hook = NoHook, int BDO = 0>
__device__ __forceinline__ void mma_taps(
    floatx16 (&acc)[MTW][NTW], const char* bptr, const int tap_bytes,
    const typename ET::afrag_t* __restrict__ wptr, const int w_mt_stride,
    typename ET::afrag_t (&first)[G][MTW],
    const typename ET::afrag_t* __restrict__ wnext, Hook mid = Hook()) {
    typedef typename ET::afrag_t frag_t;
    constexpr int NS = KT * KC;
#if defined(PM_TUNING) && defined(PM_ABLATE_MMA)   // timing experiment only: no MFMA loops
    return;
#endif
    // (split-f16: fragments are twice as wide and a step is three MFMAs per
    // tile; the second B buffer is what spilled in the whole-MRF kernel, and
    // the SIMD's other wave covers the LDS round trip)
    constexpr int BD = BDO ? BDO : (ET::SPLIT && NTW >= 2 ? 1 : PM_BDEPTH);
    static_assert(NS % G == 0, "group size must divide the step count");
    static_assert(NS >= BD, "fewer steps than B buffers");
    typedef typename ET::bfrag_t bfrag_t;
    frag_t abuf[2][G][MTW];   // A (weights): one GROUP ahead, from L2
    bfrag_t bbuf[BD][NTW];    // B (activations): BD - 1 STEPS ahead, from LDS
    auto load_b = [&](bfrag_t (&dst)[NTW], const int step) {
        const int j = step / KC, kc = step % KC;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
            dst[nt] = *reinterpret_cast<const bfrag_t*>(
                bptr + nt * 32 * S + j * tap_bytes + kc * 16 * ET::ESZ);
    };
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) abuf[0][g][mt] = first[g][mt];
#pragma unroll
    for (int s = 0; s < BD - 1; ++s) load_b(bbuf[s], s);
#pragma unroll
    for (int g0 = 0; g0 < NS; g0 += G) {
        const int cur = (g0 / G) & 1;
        if (g0 + G < NS) {
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
#if defined(PM_TUNING) && defined(PM_ABLATE_A)   // timing experiment only: no weight stream
                    abuf[cur ^ 1][g][mt] = abuf[cur][g][mt];
#else
                    abuf[cur ^ 1][g][mt] =
                        wptr[mt * w_mt_stride + (g0 + G + g) * 64];
#endif
        } else if (wnext) {
            load_a_group<ET, MTW, G>(abuf[cur ^ 1], wnext, w_mt_stride);
        }
        if (g0 == ((NS / G) / 2) * G) mid();
        // The fences pin the software pipeline. Left alone, the scheduler
        // sinks every load to just before its MFMA into ONE register set:
        // s_waitcnt vmcnt(0) (an L2 round trip) per A fragment and
        // ds_read -> lgkmcnt(0) -> MFMA (an LDS round trip) per MFMA.
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int step = g0 + g;
            const int cb = step % BD;
            if (step + BD - 1 < NS) {
#if defined(PM_TUNING) && defined(PM_ABLATE_B)   // timing experiment only: no LDS operand stream
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
                    bbuf[(step + BD - 1) % BD][nt] = bbuf[cb][nt];
#else
                load_b(bbuf[(step + BD - 1) % BD], step + BD - 1);
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
                    ET::mma(abuf[cur][g][mt], bbuf[cb][nt], acc[mt][nt]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    constexpr int LAST = ((NS / G) - 1) & 1;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) first[g][mt] = abuf[LAST ^ 1][g][mt];
}

// Debug instrumentation, -DPM_TUNING builds only (pm_debug_timeline): wave 0
// of every workgroup stamps the shader clock at phase boundaries. The shipped
// library carries none of it.
#ifdef PM_TUNING
#define PM_TIMELINE_FIELD unsigned long long* timeline;
#define PM_STAMP_AT(args, slot, i)                                           \
    do {                                                                     \
        if ((args).timeline && threadIdx.x == 0)                             \
            (args).timeline[(size_t)(slot) * 16 + (i)] =                      \
                __builtin_amdgcn_s_memtime();                                \
    } while (0)
// constant 100 MHz clock + where the workgroup ran: shader clock rate and
// the gap between consecutive tiles of a CU (scripts/timeline.py)
#define PM_STAMP_WALL(args, slot, i)                                         \
    do {                                                                     \
        if ((args).timeline && threadIdx.x == 0)                             \
            (args).timeline[(size_t)(slot) * 16 + (i)] = wall_clock64();      \
    } while (0)
#define PM_STAMP_PLACE(args, slot, i)                                        \
    do {                                                                     \
        if ((args).timeline && threadIdx.x == 0)                             \
            (args).timeline[(size_t)(slot) * 16 + (i)] =                      \
                ((unsigned long long)__builtin_amdgcn_s_getreg(              \
                     (31 << 11) | 20) << 32) |                               \
                (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);         \
    } while (0)
#else
#define PM_TIMELINE_FIELD
#define PM_STAMP_AT(args, slot, i) ((void)0)
#define PM_STAMP_WALL(args, slot, i) ((void)0)
#define PM_STAMP_PLACE(args, slot, i) ((void)0)
#endif
#define PM_STAMP(args, i) PM_STAMP_AT(args, blockIdx.x, i)

// Bias step of a packed weight stream (see the layout comment above)
template <class ET, int MTW>
__device__ __forceinline__ void load_bias_frags(
    typename ET::afrag_t (&bf)[MTW],
    const typename ET::afrag_t* __restrict__ wbias, const int w_mt_stride) {
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) bf[mt] = wbias[mt * w_mt_stride];
}

// acc = bias: the tile's first MFMA takes the constant 0 as its C operand
template <class ET, int MTW, int NTW>
__device__ __forceinline__ void bias_start(
    floatx16 (&acc)[MTW][NTW], const typename ET::afrag_t (&bf)[MTW]) {
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            floatx16 c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f,
                          0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            ET::mma_bias(bf[mt], c);
            acc[mt][nt] = c;
        }
}

// acc += bias (acc already holds the residual trunk)
template <class ET, int MTW, int NTW>
__device__ __forceinline__ void bias_add(
    floatx16 (&acc)[MTW][NTW], const typename ET::afrag_t (&bf)[MTW]) {
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) ET::mma_bias(bf[mt], acc[mt][nt]);
}

__device__ __forceinline__ float4 acc_quad(const floatx16& v, const int g4) {
    return make_float4(v[4 * g4], v[4 * g4 + 1], v[4 * g4 + 2], v[4 * g4 + 3]);
}

// One 32 x 32 accumulator tile -> LDS rows in the operand type, LeakyReLU
// fused, zeroed where `zero` (columns outside the utterance). In the C/D
// layout a lane holds 4 consecutive channels per register quad and lane
// l + 32 the next 4: two quads are exchanged across the half-waves
// (v_permlane32_swap) so that every lane owns 8 consecutive channels and
// stores 16 bytes. ds_write_b128 is served in 8-lane groups whose rows, at a
// row stride of 16 x odd bytes, cover the 32 banks exactly once; the 8-byte
// stores this replaces were 2-way conflicted (rows r and r + 8 on one bank:
// SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 18-20 % in the whole-Block kernels).
// `rowp` = the lane's row in the LDS tile, `ch` = the tile's first channel.
// Split (hi + lo) operand tiles: 1 = a lane stores its four channels' hi and lo
// parts as two 8-byte pieces (2-way bank conflicted at the 144-byte pitch); 0 =
// the half-waves exchange first and every lane stores ONE conflict-free 16-byte
// block. Measured: the exchange (8 more v_permlane32_swap a tile) costs more
// than the conflicts - A2 whole-MRF 4.07 vs 3.88-3.99 ms, x3 5.34 vs 5.26
// (profiles/r06/ab_split_store.txt) - so 1 ships.
#ifndef PM_SPLIT_STORE8
#define PM_SPLIT_STORE8 1
#endif
template <class ET, bool MASK>
__device__ __forceinline__ void store_tile_lrelu_impl(
    char* rowp, const int ch, const floatx16& v, const bool zero,
    const int lh) {
    if constexpr (ET::ESZ == 2) {
#pragma unroll
        for (int g4 = 0; g4 < 4; g4 += 2) {
            float4 lo = pm_lrelu4(acc_quad(v, g4));
            float4 hi = pm_lrelu4(acc_quad(v, g4 + 1));
            if (MASK && zero) {
                lo = make_float4(0.f, 0.f, 0.f, 0.f);
                hi = lo;
            }
            uint2 a = ET::pack4(lo), b = ET::pack4(hi);
            auto sx = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
            auto sy = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
            // lanes < 32: [own quad g4 | upper half's quad g4] = channels
            // 8 g4 .. 8 g4 + 7; lanes >= 32: [lower's g4 + 1 | own g4 + 1]
            *reinterpret_cast<uint4*>(rowp + (ch + 8 * (g4 + lh)) * 2) =
                make_uint4(sx[0], sy[0], sx[1], sy[1]);
        }
    } else if constexpr (ET::SPLIT && !PM_SPLIT_STORE8) {
        // (A/B form, see PM_SPLIT_STORE8) hi + lo layout - an 8-channel group =
        // [8 x hi | 8 x lo], 32 bytes: lanes < 32 take the group's 16-byte hi
        // block, lanes >= 32 its lo block, one ds_write_b128 each
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            float4 q = pm_lrelu4(acc_quad(v, g4));
            if (MASK && zero) q = make_float4(0.f, 0.f, 0.f, 0.f);
            half4 hi, lo;
            ElemF16X3::split4(q, hi, lo);
            const uint2 a = __builtin_bit_cast(uint2, hi);
            const uint2 b = __builtin_bit_cast(uint2, lo);
            auto sx = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
            auto sy = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
            // lanes < 32: [own hi | upper half's hi]; >= 32: [lower's lo | own lo]
            *reinterpret_cast<uint4*>(rowp + ((ch >> 3) + g4) * 32 + lh * 16) =
                make_uint4(sx[0], sy[0], sx[1], sy[1]);
        }
    } else {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            float4 q = pm_lrelu4(acc_quad(v, g4));
            if (MASK && zero) q = make_float4(0.f, 0.f, 0.f, 0.f);
            ET::store4_at(rowp, ch + 8 * g4 + 4 * lh, q);
        }
    }
}

// `t_tile` = time of the tile's first column: the zero-padding mask (columns
// outside [0, L)) only runs on tiles that straddle an utterance edge - a
// wave-uniform branch, both sides executed with all lanes active.
template <class ET>
__device__ __forceinline__ void store_tile_lrelu(
    char* rowp, const int ch, const floatx16& v, const int t_tile,
    const int L, const int ln, const int lh) {
#if defined(PM_TUNING) && defined(PM_ABLATE_EPI)   // timing experiment only: no epilogue stores
    return;
#endif
    if (t_tile >= 0 && t_tile + 32 <= L) {
        store_tile_lrelu_impl<ET, false>(rowp, ch, v, false, lh);
    } else {
        const int t = t_tile + ln;
        store_tile_lrelu_impl<ET, true>(rowp, ch, v, !(t >= 0 && t < L), lh);
    }
}

// ---------------------------------------------------------------------------
// Fused Block iteration
// ---------------------------------------------------------------------------

struct PairArgs {
    const float* x;      // (B, L, C) fp32 trunk (pre-activation)
    float* out;          // (B, L, C)
    const void* w1;      // packed conv1 weights (dilated) + bias step
    const void* w2;      // packed conv2 weights (dilation 1) + bias step
    int B, L;
    int dilation;
    int mode;            // 0: out = y; 1: out = y * scale; 2: out += y * scale
    float scale;
    int ntiles;          // tiles per utterance
    const int* lengths;  // (B) valid frames per utterance or null (ragged batch)
    int len_scale;       // samples of this tensor per frame
    PM_TIMELINE_FIELD    // debug stamps (tuning builds)
};

template <int C, int K, int WN, int NTW>
struct PairGeom {
    static constexpr int N1 = WN * NTW * 32;   // conv1 output columns
    static constexpr int TL = N1 - (K - 1);    // valid outputs per tile
};

// ALIAS: the conv1 -> conv2 intermediate tile overlays the staged x chunks
// (dead once every wave has finished conv1; costs one barrier).
template <class ET, int C, int K, int WM, int WN, int NTW, int CH, int ALIAS>
__host__ __device__ constexpr int pair_smem_bytes(int d) {
    constexpr int NCH = C / CH;
    constexpr int N1 = WN * NTW * 32;
    const int x = (NCH > 1 ? 2 : 1) * (N1 + (K - 1) * d) * (CH * ET::ESZ + 16);
    const int inter = (N1 + K - 1) * (C * ET::ESZ + 16);
    return ALIAS ? (x > inter ? x : inter) : x + inter;
}

template <class ET, int C, int K, int WM, int WN, int NTW, int CH, int ALIAS>
__global__ __launch_bounds__(WM * WN * 64) void conv_pair_kernel(
    PairArgs a) {
    typedef typename ET::afrag_t frag_t;
    constexpr int NCH = C / CH;
    constexpr int KC = CH / 16;
    constexpr int MT = C / 32;
    constexpr int MTW = MT / WM;
    static_assert(MT % WM == 0, "WM must divide the M tile count");
    constexpr int N1 = WN * NTW * 32;
    constexpr int NT = WM * WN * 64;
    constexpr int H2 = (K - 1) / 2;
    constexpr int TL = N1 - (K - 1);
    constexpr int SX = CH * ET::ESZ + 16;
    constexpr int SI = C * ET::ESZ + 16;
    constexpr int XR_MAX = N1 + (K - 1) * 5;
    // weight-fragment prefetch depth (k16 steps). The 6-tile-wide C = 256
    // variant at k 11 is at the 256-register limit: depth 4 spilled 4-8
    // registers into its MFMA loop, depth 2 fits
    constexpr int G = (ET::ESZ == 4) ? (KC >= 2 ? 2 : 1)
                    : (KC < 4 ? KC : (NTW >= 6 && K == 11 ? 2 : 4));

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int ln = lane & 31, lh = lane >> 5;

    // The second-dispatched half of an 8-wave workgroup loses every issue
    // arbitration to its older SIMD partner; a static priority evens the
    // pair out (-0.3 % of the step, profiles/r02/ab_block_buffer_ops_prio.txt)
    if (WM * WN == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);
    const int wg = pm_xcd_remap(blockIdx.x, gridDim.x);
    const int tile = wg % a.ntiles;
    const int b = wg / a.ntiles;
    const int t0 = tile * TL;
    // Ragged batches: utterance b is L rows long inside a (B, a.L, C) buffer;
    // rows >= L are never read (they count as the conv's zero padding) nor
    // written, so every utterance equals its stand-alone synthesis.
    const int L = a.lengths ? min(a.lengths[b] * a.len_scale, a.L) : a.L;
    if (t0 >= L) return;
    const int d = a.dilation;
    const int hd = H2 * d;
    const int XR = N1 + (K - 1) * d;

    char* xbuf = smem;
    char* inter = ALIAS ? smem : smem + (NCH > 1 ? 2 : 1) * XR * SX;

    const float* __restrict__ xb = a.x + (size_t)b * a.L * C;
    const int t_first = t0 - H2 - hd;

    floatx16 acc[MTW][NTW];

    // ---------------- conv1: K-loop over staged channel chunks -------------
    constexpr int W_BIAS = NCH * K * KC * 64;       // bias step of a stream
    constexpr int W_MT_STRIDE = W_BIAS + 64;
    constexpr int W_CHUNK = K * KC * 64;
    const frag_t* w1 = reinterpret_cast<const frag_t*>(a.w1) +
                       (size_t)(wm * MTW) * W_MT_STRIDE + lane;
    const frag_t* w2 = reinterpret_cast<const frag_t*>(a.w2) +
                       (size_t)(wm * MTW) * W_MT_STRIDE + lane;
    frag_t bf[MTW];
    load_bias_frags<ET, MTW>(bf, w1 + W_BIAS, W_MT_STRIDE);
    PM_STAMP(a, 0);
    PM_STAMP_WALL(a, blockIdx.x, 12);
    PM_STAMP_PLACE(a, blockIdx.x, 14);
    const int lane_off_x =
        ((wn * NTW * 32) + ln) * SX + lh * 8 * ET::ESZ;
    frag_t afirst[G][MTW];

    load_a_group<ET, MTW, G>(afirst, w1, W_MT_STRIDE);

    ChunkStager<ET, CH, NT, XR_MAX> stager;
    stager.load(xb, C, 0, t_first, XR, L, tid);
    stager.template store<true>(xbuf, XR, tid);
    // Chunk 1 is requested before the barrier, not behind it: vmcnt retires
    // in order, so the weight groups the MFMA loop fetches queue up behind
    // these loads - the earlier they go out, the shorter that first stall
    // (timeline: chunk 0's MFMA phase took 13.9 k cycles, chunk 1's 9.0 k).
    if (NCH > 1) stager.load(xb, C, CH, t_first, XR, L, tid);
    pm_block_sync();
    PM_STAMP(a, 1);
    bias_start<ET, MTW, NTW>(acc, bf);

#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
        char* cur = xbuf + (NCH > 1 ? (c & 1) * XR * SX : 0);
        const bool more = c + 1 < NCH;
        if (more && c > 0)
            stager.load(xb, C, (c + 1) * CH, t_first, XR, L, tid);
        char* nxt = xbuf + ((c + 1) & 1) * XR * SX;
        auto stage_next = [&]() {
            if (more) stager.template store<true>(nxt, XR, tid);
        };
        mma_taps<ET, K, KC, MTW, NTW, G, SX>(
            acc, cur + lane_off_x, d * SX, w1 + (size_t)c * W_CHUNK,
            W_MT_STRIDE, afirst, more ? w1 + (size_t)(c + 1) * W_CHUNK : w2,
            stage_next);
        if (c < 2) PM_STAMP(a, 6 + 3 * c);
        if (more) {
            pm_block_sync();
            if (c < 2) PM_STAMP(a, 8 + 3 * c);
        }
    }

    PM_STAMP(a, 2);
    if (ALIAS) pm_block_sync();     // every wave is done with the x chunks
    // ---------------- epilogue 1: lrelu, zero-pad mask -> LDS --------------
    // (bias already in the accumulator; the mask only on tiles that straddle
    // an utterance edge - a wave-uniform branch)
    // The residual rides in the accumulator: as soon as a tile's conv1
    // result is in LDS its registers take the fp32 trunk x of the conv2
    // output tile (same C/D layout), conv2 accumulates on top of it and
    // epilogue 2 only stores. The loads fly across the barrier
    // (pm_block_sync fences LDS only) instead of costing one HBM round trip
    // per 32 x 32 tile between conv2 and the stores.
    load_bias_frags<ET, MTW>(bf, w2 + W_BIAS, W_MT_STRIDE);
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int n_first = (wn * NTW + nt) * 32;
            const int n = n_first + ln;
            const int t_tile = t0 - H2 + n_first;
            store_tile_lrelu<ET>(inter + n * SI, (wm * MTW + mt) * 32,
                                 acc[mt][nt], t_tile, L, ln, lh);
            // (columns beyond the tile / the utterance are never stored:
            // they load a valid row and their sums are don't-cares)
            const int t = min(t0 + n, L - 1);
            const float* xr = xb + (size_t)t * C +
                              (wm * MTW + mt) * 32 + 4 * lh;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 r = *reinterpret_cast<const float4*>(xr + 8 * g4);
                acc[mt][nt][4 * g4 + 0] = r.x;
                acc[mt][nt][4 * g4 + 1] = r.y;
                acc[mt][nt][4 * g4 + 2] = r.z;
                acc[mt][nt][4 * g4 + 3] = r.w;
            }
        }
    }
    pm_block_sync();
    __builtin_amdgcn_sched_barrier(0);   // the bias MFMAs wait for the loads:
    bias_add<ET, MTW, NTW>(acc, bf);     // keep them behind the barrier
    PM_STAMP(a, 3);

    // ---------------- conv2 (dilation 1) straight out of LDS ---------------
    const int lane_off_i = ((wn * NTW * 32) + ln) * SI + lh * 8 * ET::ESZ;
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
        mma_taps<ET, K, KC, MTW, NTW, G, SI>(
            acc, inter + lane_off_i + c * CH * ET::ESZ, SI,
            w2 + (size_t)c * W_CHUNK, W_MT_STRIDE, afirst,
            c + 1 < NCH ? w2 + (size_t)(c + 1) * W_CHUNK : nullptr);
    }

    PM_STAMP(a, 4);
    // ---------------- epilogue 2: bias + residual (+ MRF accumulate) -------
    // All global loads of a 32x32 tile are issued before its first store:
    // interleaved, every load queued behind the previous store's address
    // dependence and the epilogue became 16 serial HBM round trips.
    // Buffer accesses: rows beyond the tile / the utterance fall outside
    // the descriptor's range - stores are dropped, loads return 0 - so the
    // epilogue has no per-lane branches.
    const int mode = a.mode;
    const float scale = a.scale;
    const int rows = min(TL, L - t0);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
        a.out + ((size_t)b * a.L + t0) * C, 0, rows * C * 4, 0x00020000);
    const unsigned voff0 =
        (unsigned)(((wn * NTW * 32 + ln) * C + wm * MTW * 32 + 4 * lh) * 4);
    if (mode == 2) {
        // MRF accumulation: the reads of `out` are issued in batches before
        // the first store of a batch (one round trip per batch, not per
        // tile; the operand fragment registers are dead by now)
        constexpr int NB = NTW > 4 ? (NTW + 1) / 2 : NTW;
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int n0 = 0; n0 < NTW; n0 += NB) {
                pm_u4 old[NB][4];
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    if (n0 + i >= NTW) break;
                    const unsigned voff =
                        voff0 + (unsigned)((mt * 32 + (n0 + i) * 32 * C) * 4);
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4)
                        old[i][g4] = __builtin_amdgcn_raw_buffer_load_b128(
                            orsrc, voff + g4 * 32, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    if (n0 + i >= NTW) break;
                    const int nt = n0 + i;
                    const unsigned voff =
                        voff0 + (unsigned)((mt * 32 + nt * 32 * C) * 4);
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const pm_u4 o = old[i][g4];
                        pm_u4 r;
                        r.x = __float_as_uint(__uint_as_float(o.x) +
                                              acc[mt][nt][4 * g4 + 0] * scale);
                        r.y = __float_as_uint(__uint_as_float(o.y) +
                                              acc[mt][nt][4 * g4 + 1] * scale);
                        r.z = __float_as_uint(__uint_as_float(o.z) +
                                              acc[mt][nt][4 * g4 + 2] * scale);
                        r.w = __float_as_uint(__uint_as_float(o.w) +
                                              acc[mt][nt][4 * g4 + 3] * scale);
                        __builtin_amdgcn_raw_buffer_store_b128(
                            r, orsrc, voff + g4 * 32, 0, 0);
                    }
                }
            }
    } else {
        const float s = mode == 1 ? scale : 1.f;
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const unsigned voff =
                    voff0 + (unsigned)((mt * 32 + nt * 32 * C) * 4);
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    pm_u4 r;
                    r.x = __float_as_uint(acc[mt][nt][4 * g4 + 0] * s);
                    r.y = __float_as_uint(acc[mt][nt][4 * g4 + 1] * s);
                    r.z = __float_as_uint(acc[mt][nt][4 * g4 + 2] * s);
                    r.w = __float_as_uint(acc[mt][nt][4 * g4 + 3] * s);
                    __builtin_amdgcn_raw_buffer_store_b128(
                        r, orsrc, voff + g4 * 32, 0, 0);
                }
            }
    }
    PM_STAMP(a, 5);
    PM_STAMP_WALL(a, blockIdx.x, 13);
}

// ---------------------------------------------------------------------------
// Generic single convolution / polyphase transposed convolution
// ---------------------------------------------------------------------------

struct SingleArgs {
    const void* x16;      // (B, L, Cin) operand-type copy of act(x), or null:
                          // staged as it is instead of x (16-bit types only)
    const float* x;       // (B, L, Cin) fp32
    float* out;           // (B, L, M) fp32  (for ConvTranspose: (B, L*r, Cout))
    const void* w;        // packed weights, KT taps per M tile
    const float* bias;    // (M)
    const float* gbias;   // (B|1, M) per-utterance bias (speaker conv) or null
    int gbias_batch;      // 1 -> broadcast row 0
    int B, L, Cin, M;
    int Lout;             // output rows per utterance (== L except STFT)
    int bins;             // EPI 1/2/3: DFT bins (output (B, bins, Lout))
    unsigned* maxbits;    // EPI 2: per-utterance max (order-preserving bits)
    const float* grad;    // EPI 3: d loss / d magnitude, (B, bins, Lout)
    int lrelu;            // apply LeakyReLU to the input while staging
    int pad;              // rows of left padding of the staged tile
    // tap window start for an M row block (ConvTranspose phases): window
    // starts at tap 1 when ((m / phase_c) + phase_p >= phase_r), else 0;
    // phase_r == 0 disables (plain conv, window = all KT taps from 0)
    int phase_c, phase_p, phase_r;
    int ntiles, nmblocks;
    const int* lengths;   // (B) valid frames per utterance or null
    int len_scale;        // input rows per frame
};

// KT: taps contracted per M tile; KSPAN: taps spanned by the staged halo
// (KSPAN == KT for a plain conv, 3 for the 2-tap polyphase ConvTranspose).
// EPI 0: out (B, Lout, M) = acc + bias (+ per-utterance bias)
// EPI 1: framed-DFT magnitude: M rows are (re, im) pairs of one bin,
//        out (B, bins, Lout) = sqrt(re^2 + im^2 + 1e-6)   spectrogram.py:53
// EPI 2: out (B, bins, Lout) = 10 log10(max(1e-10, re^2 + im^2)) and the
//        per-utterance maximum (librosa.amplitude_to_db, loudness.py:46)
// EPI 3: backward of EPI 1, first half: out (B, Lout, M) channels-last =
//        grad[bin] / magnitude * (re, im) - the cotangent of the framed DFT,
//        which a second conv against the transposed basis overlap-adds
__device__ __forceinline__ unsigned pm_float_order_bits(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

template <class ET, int KT, int KSPAN, int CH, int WM, int WN, int MTW, int NTW,
          int EPI>
__global__ __launch_bounds__(WM * WN * 64) void conv_single_kernel(
    SingleArgs a) {
    typedef typename ET::afrag_t frag_t;
    constexpr int KC = CH / 16;
    constexpr int N1 = WN * NTW * 32;
    constexpr int NT = WM * WN * 64;
    constexpr int SX = CH * ET::ESZ + 16;
    constexpr int XR = N1 + KSPAN - 1;
    constexpr int MB = WM * MTW * 32;
    constexpr int G = (ET::ESZ == 4) ? (KC >= 2 ? 2 : 1) : KC;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int ln = lane & 31, lh = lane >> 5;

    const int wg = pm_xcd_remap(blockIdx.x, gridDim.x);
    const int mb = wg % a.nmblocks;
    const int tile = (wg / a.nmblocks) % a.ntiles;
    const int b = wg / (a.nmblocks * a.ntiles);
    const int t0 = tile * N1;
    const int Cin = a.Cin, M = a.M;
    const int L = a.lengths ? min(a.lengths[b] * a.len_scale, a.L) : a.L;
    const int Lout = a.lengths ? L : a.Lout;   // ragged: conv / convT only
    if (t0 >= Lout) return;
    const int NCH = Cin / CH;
    const int m0 = mb * MB + wm * MTW * 32;   // this wave's first M row
    int js = 0;
    if (a.phase_r > 0 && (m0 / a.phase_c) + a.phase_p >= a.phase_r) js = 1;

    const float* xb = a.x + (size_t)b * a.L * Cin;
    const int t_first = t0 - a.pad;

    floatx16 acc[MTW][NTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    ChunkStager<ET, CH, NT, XR> stager;
    // (x16: the producer already wrote the operand values - see SingleArgs)
    bool pre = false;
    const char* xb16 = nullptr;
    if constexpr (ET::ESZ == 2) {
        pre = a.x16 != nullptr;
        xb16 = reinterpret_cast<const char*>(a.x16) + (size_t)b * a.L * Cin * 2;
    }
    if (pre) {
        stager.load16(xb16, Cin, 0, t_first, XR, L, tid);
        stager.store16(smem, XR, tid);
    } else {
        stager.load(xb, Cin, 0, t_first, XR, L, tid);
        if (a.lrelu) stager.template store<true>(smem, XR, tid);
        else stager.template store<false>(smem, XR, tid);
    }
    pm_block_sync();

    const int w_mt_stride = NCH * KT * KC * 64;
    const frag_t* w = reinterpret_cast<const frag_t*>(a.w) +
                      (size_t)(m0 / 32) * w_mt_stride + lane;
    frag_t afirst[G][MTW];
    load_a_group<ET, MTW, G>(afirst, w, w_mt_stride);
    const int lane_off_x =
        ((wn * NTW * 32) + ln + js) * SX + lh * 8 * ET::ESZ;

#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
        char* cur = smem + (c & 1) * XR * SX;
        if (c + 1 < NCH) {
            if (pre) stager.load16(xb16, Cin, (c + 1) * CH, t_first, XR, L, tid);
            else stager.load(xb, Cin, (c + 1) * CH, t_first, XR, L, tid);
        }
        mma_taps<ET, KT, KC, MTW, NTW, G, SX>(
            acc, cur + lane_off_x, SX, w + (size_t)c * (KT * KC * 64),
            w_mt_stride, afirst,
            c + 1 < NCH ? w + (size_t)(c + 1) * (KT * KC * 64) : nullptr);
        if (c + 1 < NCH) {
            char* nxt = smem + ((c + 1) & 1) * XR * SX;
            if (pre) stager.store16(nxt, XR, tid);
            else if (a.lrelu) stager.template store<true>(nxt, XR, tid);
            else stager.template store<false>(nxt, XR, tid);
            pm_block_sync();
        }
    }

    if constexpr (EPI == 0) {
    // buffer stores over the rows of this utterance the tile owns: columns
    // beyond Lout are out of range and dropped (no per-lane branch); the
    // bias of an M tile is read once, not once per column tile
    const float* gb = a.gbias
        ? a.gbias + (size_t)(a.gbias_batch == 1 ? 0 : b) * M : nullptr;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
        a.out + ((size_t)b * a.Lout + t0) * M, 0,
        min(Lout - t0, N1) * M * 4, 0x00020000);
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        const int co_base = m0 + mt * 32 + 4 * lh;
        float4 bias[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int co = co_base + 8 * g4;
            bias[g4] = *reinterpret_cast<const float4*>(a.bias + co);
            if (gb) {
                const float4 g = *reinterpret_cast<const float4*>(gb + co);
                bias[g4].x += g.x; bias[g4].y += g.y;
                bias[g4].z += g.z; bias[g4].w += g.w;
            }
        }
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int n = (wn * NTW + nt) * 32 + ln;
            const unsigned voff = (unsigned)((n * M + co_base) * 4);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                pm_u4 r;
                r.x = __float_as_uint(acc[mt][nt][4 * g4 + 0] + bias[g4].x);
                r.y = __float_as_uint(acc[mt][nt][4 * g4 + 1] + bias[g4].y);
                r.z = __float_as_uint(acc[mt][nt][4 * g4 + 2] + bias[g4].z);
                r.w = __float_as_uint(acc[mt][nt][4 * g4 + 3] + bias[g4].w);
                __builtin_amdgcn_raw_buffer_store_b128(
                    r, orsrc, voff + g4 * 32, 0, 0);
            }
        }
    }
    } else if constexpr (EPI == 3) {
    float* ob = a.out + (size_t)b * a.Lout * M;
    const float* gb = a.grad + (size_t)b * a.bins * a.Lout;
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        const int row_base = m0 + mt * 32 + 4 * lh;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int n = (wn * NTW + nt) * 32 + ln;
            const int t = t0 + n;
            if (t < a.Lout) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    float v[4];
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        const int bin = (row_base + 8 * g4) / 2 + pr;
                        const float re = acc[mt][nt][4 * g4 + 2 * pr];
                        const float im = acc[mt][nt][4 * g4 + 2 * pr + 1];
                        float scale = 0.f;
                        if (bin < a.bins)
                            scale = gb[(size_t)bin * a.Lout + t] /
                                    sqrtf(re * re + im * im + 1e-6f);
                        v[2 * pr] = scale * re;
                        v[2 * pr + 1] = scale * im;
                    }
                    *reinterpret_cast<float4*>(
                        ob + (size_t)t * M + row_base + 8 * g4) =
                        make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    }
    } else {
    float* ob = a.out + (size_t)b * a.bins * a.Lout;
    float local_max = -INFINITY;
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        const int row_base = m0 + mt * 32 + 4 * lh;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int n = (wn * NTW + nt) * 32 + ln;
            const int t = t0 + n;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const int bin = (row_base + 8 * g4) / 2 + pr;
                    const float re = acc[mt][nt][4 * g4 + 2 * pr];
                    const float im = acc[mt][nt][4 * g4 + 2 * pr + 1];
                    const float pw = re * re + im * im;
                    if (t < a.Lout && bin < a.bins) {
                        float v;
                        if constexpr (EPI == 1) {
                            v = sqrtf(pw + 1e-6f);
                        } else {
                            v = 10.f * log10f(fmaxf(1e-10f, pw));
                            local_max = fmaxf(local_max, v);
                        }
                        ob[(size_t)bin * a.Lout + t] = v;
                    }
                }
            }
        }
    }
    if constexpr (EPI == 2) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            local_max = fmaxf(local_max, __shfl_xor(local_max, o, 64));
        if (lane == 0 && local_max > -INFINITY)
            atomicMax(a.maxbits + b, pm_float_order_bits(local_max));
    }
    }
}

// ---------------------------------------------------------------------------
// Wide polyphase upsampler (C_in = 256 / 512, r = 8: M = r C_out = 1024 / 2048
// GEMM rows, K = 2 C_in): conv_single_kernel walks K in 64-channel chunks and
// covers M with one workgroup per 256 rows, so the same x tile is staged
// M / 256 times and an MFMA phase between two barriers is 8 k16-steps long.
// Here the whole x tile ((128 + 2) rows x C_in, 16-bit) is staged ONCE and the
// workgroup walks its M blocks with no barrier in between - one continuous
// weight stream (packed as a single chunk, CH = C_in).
// ---------------------------------------------------------------------------
template <class ET, int CIN, int WM, int WN, int MTW, int NTW>
__global__ __launch_bounds__(WM * WN * 64) void conv_upsample_kernel(
    SingleArgs a) {
    typedef typename ET::afrag_t frag_t;
    constexpr int KT = 2, KSPAN = 3;
    constexpr int KC = CIN / 16;
    constexpr int N1 = WN * NTW * 32;
    constexpr int NT = WM * WN * 64;
    constexpr int S = CIN * ET::ESZ + 16;
    constexpr int XR = N1 + KSPAN - 1;
    constexpr int MB = WM * MTW * 32;
    // (depth 2 under a 128-register cap, so that two workgroups share a CU
    // at C_in = 256, is 12 % slower: profiles/r02/ab_upsample_whole_k.txt)
    constexpr int G = 4;
    constexpr int SLAB = 128;          // channels staged per register pass

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int ln = lane & 31, lh = lane >> 5;

    // nmblocks = M groups per column tile (adjacent workgroup ids: the groups
    // of a tile share its x rows in one XCD's L2)
    const int wg = pm_xcd_remap(blockIdx.x, gridDim.x);
    const int mg = wg % a.nmblocks;
    const int tile = (wg / a.nmblocks) % a.ntiles;
    const int b = wg / (a.nmblocks * a.ntiles);
    const int t0 = tile * N1;
    const int M = a.M;
    const int L = a.lengths ? min(a.lengths[b] * a.len_scale, a.L) : a.L;
    const int Lout = a.lengths ? L : a.Lout;
    if (t0 >= Lout) return;

    const float* xb = a.x + (size_t)b * a.L * CIN;
    const int t_first = t0 - a.pad;
    {
        ChunkStager<ET, SLAB, NT, XR> stager;
#pragma unroll 1
        for (int c0 = 0; c0 < CIN; c0 += SLAB) {
            stager.load(xb, CIN, c0, t_first, XR, L, tid);
            if (a.lrelu)
                stager.template store<true, S>(smem + c0 * ET::ESZ, XR, tid);
            else
                stager.template store<false, S>(smem + c0 * ET::ESZ, XR, tid);
        }
    }
    pm_block_sync();

    constexpr int w_mt_stride = KT * KC * 64;
    const int per_group = (M / MB) / a.nmblocks;
    const frag_t* wbase = reinterpret_cast<const frag_t*>(a.w) + lane;
    const float* gb = a.gbias
        ? a.gbias + (size_t)(a.gbias_batch == 1 ? 0 : b) * M : nullptr;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
        a.out + ((size_t)b * a.Lout + t0) * M, 0,
        min(Lout - t0, N1) * M * 4, 0x00020000);

    frag_t afirst[G][MTW];
    load_a_group<ET, MTW, G>(
        afirst,
        wbase + (size_t)((mg * per_group * MB + wm * MTW * 32) / 32) * w_mt_stride,
        w_mt_stride);
#pragma unroll 1
    for (int mi = 0; mi < per_group; ++mi) {
        const int m0 = (mg * per_group + mi) * MB + wm * MTW * 32;
        const int js =
            (a.phase_r > 0 && (m0 / a.phase_c) + a.phase_p >= a.phase_r) ? 1 : 0;
        const frag_t* w = wbase + (size_t)(m0 / 32) * w_mt_stride;
        floatx16 acc[MTW][NTW];
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
        const int lane_off =
            ((wn * NTW * 32) + ln + js) * S + lh * 8 * ET::ESZ;
        mma_taps<ET, KT, KC, MTW, NTW, G, S>(
            acc, smem + lane_off, S, w, w_mt_stride, afirst,
            mi + 1 < per_group ? w + (size_t)(MB / 32) * w_mt_stride : nullptr);
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            const int co_base = m0 + mt * 32 + 4 * lh;
            float4 bias[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int co = co_base + 8 * g4;
                bias[g4] = *reinterpret_cast<const float4*>(a.bias + co);
                if (gb) {
                    const float4 g = *reinterpret_cast<const float4*>(gb + co);
                    bias[g4].x += g.x; bias[g4].y += g.y;
                    bias[g4].z += g.z; bias[g4].w += g.w;
                }
            }
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int n = (wn * NTW + nt) * 32 + ln;
                const unsigned voff = (unsigned)((n * M + co_base) * 4);
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    pm_u4 r;
                    r.x = __float_as_uint(acc[mt][nt][4 * g4 + 0] + bias[g4].x);
                    r.y = __float_as_uint(acc[mt][nt][4 * g4 + 1] + bias[g4].y);
                    r.z = __float_as_uint(acc[mt][nt][4 * g4 + 2] + bias[g4].z);
                    r.w = __float_as_uint(acc[mt][nt][4 * g4 + 3] + bias[g4].w);
                    __builtin_amdgcn_raw_buffer_store_b128(
                        r, orsrc, voff + g4 * 32, 0, 0);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Whole `Block` fused (where the +-6 (k-1) halo is affordable): all 3 iterations of
//   x <- x + conv2(lrelu(conv1(lrelu(x))))            hifigan.py:198-210
// in one kernel. The fp32 trunk x lives in REGISTERS (MFMA C/D layout, each
// wave owns a run of time columns x a 32-channel slab), LDS only holds the two
// 16-bit MFMA B operands: `a` = lrelu(x) and `t` = lrelu(conv1). HBM traffic
// per Block drops from 3 x (read + write) to one read (with a +-6 (k-1)
// halo, recomputed) and one write.
// ---------------------------------------------------------------------------

struct Block3Args {
    const float* x;
    float* out;
    const void* w1[3];   // packed weights + bias step per iteration
    const void* w2[3];
    int dil[3];
    int niter;
    int B, L;
    int mode;            // 0: out = y; 1: out = y * scale; 2: out += y * scale
    float scale;
    int ntiles, halo, TL;
    const int* lengths;  // (B) valid frames per utterance or null
    int len_scale;
    char* scratch;       // device scratch for the skewed walk, or null
    size_t scratch_bytes;
    // Optional (skewed walk only): instead of the fp32 tensor `out`, store
    // cvt(lrelu(result)) as the 16-bit MFMA operand type `act16_type` (PM_F16 /
    // PM_BF16 numbering) - what the next stage's upsampler, the only reader
    // of a stage's output, stages anyway. *act16_done (host) is set to 1 by
    // the launcher when the kernel it took honours the request.
    void* act16;
    int act16_type;
    int* act16_done;
    // pm_launch_mrf: take the skewed whole-MRF walk or nothing
    // (hipErrorNotSupported) - the caller's fallback is Block by Block
    int skew_only;
    PM_TIMELINE_FIELD    // debug stamps (tuning builds)
};

template <class ET, int C, int K, int WM, int WN, int NTW>
__host__ __device__ constexpr int block3_smem_bytes() {
    constexpr int NC = WN * NTW * 32;
    constexpr int S = C * ET::ESZ + 16;
    return (NC + 10 * ((K - 1) / 2)) * S + (NC + (K - 1)) * S;
}

// SUM: 0 = stand-alone Block (store per a.mode); inside a whole-MRF launch
// with the sum of the Blocks held in registers: 1 = first Block (sum = y),
// 2 = middle (sum += y), 3 = last (store (sum + y) * scale).
// XMODE (whole-MRF launches, where the three Blocks start from the same x
// tile): bit 0 = start from `xnext` instead of reading x from memory; bit 1 =
// request x again for the NEXT Block during the last iteration's epilogue 1,
// into the conv1 accumulators (dead from there on), and hand it over in
// `xnext`: the round trip runs under the last conv2 instead of in front of
// the next Block's first conv. Walked kernels: bit 2 = the request is for the
// NEXT tile (B3Walk::next_c_first), and bit 0 applies when B3Walk::have_x.
// WALK (conv_mrf_walk_kernel): the workgroup walks consecutive tiles of one
// utterance segment, left to right, and every layer's input keeps its last
// columns in a small LDS "carry" area, from where the next tile fills its LEFT
// margins. The left halo is then exact context instead of recomputed garbage:
// a tile is valid from its first column, only the right halo (a.halo columns)
// is recomputed by the next tile - half the redundant work. Same arithmetic per
// column, so results are bit-identical to the two-sided tiling.
struct B3Walk {
    int b;            // utterance
    int L;            // its length in columns
    int c_first;      // time of the tile's column 0
    int store_lo;     // first column this tile stores
    int store_n;      // columns it stores
    int left;         // 1: left margins come from the carry area
    int have_x;       // XMODE bit 0: `xnext` really holds this tile's x
    int next_c_first; // XMODE bit 2: the tile whose x is requested, or < 0
    char* carry;      // this Block's carry area (halo rows)
    // the Block's weight streams and dilations, read straight from the
    // kernel-argument segment (scalar loads with a run-time iteration index;
    // a by-value copy indexed at run time would live in scratch)
    const void* const __attribute__((address_space(4)))* w1;
    const void* const __attribute__((address_space(4)))* w2;
    const int __attribute__((address_space(4)))* dil;
};

template <class ET, int C, int K, int WM, int WN, int NTW, int SUM = 0,
          int XMODE = 0, int WALK = 0>
__device__ __forceinline__ void block3_body(
    const Block3Args& a, char* smem,
    floatx16 (&sum)[(C / 32) / WM][NTW],
    floatx16 (&xnext)[(C / 32) / WM][NTW],
    const B3Walk* wk = nullptr) {
    typedef typename ET::afrag_t frag_t;
    constexpr int CH = C < 64 ? C : 64;    // weight-stream chunk (as packed)
    constexpr int NCH = C / CH;
    constexpr int KC = CH / 16;
    constexpr int MTW = (C / 32) / WM;     // M tiles per wave
    constexpr int NC = WN * NTW * 32;      // columns held by the workgroup
    constexpr int NT = WM * WN * 64;
    constexpr int H2 = (K - 1) / 2;
    constexpr int MA = 5 * H2;             // margin of `a` (max dilation 5)
    constexpr int S = C * ET::ESZ + 16;
    // (walked: no right margin behind `a` - the dilated taps of the last
    // columns then read the first rows of `t`, finite values that only ever
    // reach columns of the right halo, which the next tile recomputes)
    constexpr int ROWS_A = NC + (WALK ? 1 : 2) * MA;
    // (weight-fragment prefetch depth in k16 steps; a split-f16 step is three
    // MFMAs per tile - 192+ cycles at two tiles per wave -, so one step ahead
    // covers the L2 round trip and depth 2 spilled at C = 32 k 11)
#ifndef PM_K3_G
#define PM_K3_G 0       // (A/B builds: weight prefetch depth of the k 3 whole-Block kernels at C >= 128)
#endif
    constexpr int G = (ET::ESZ == 4) ? (ET::SPLIT && NTW >= 2 ? 1 : 2)
                    : (PM_K3_G && K == 3 && C >= 128 ? PM_K3_G : KC);
    constexpr int W_CHUNK = K * KC * 64;
    constexpr int W_BIAS = NCH * W_CHUNK;          // bias step of a stream
    constexpr int W_MT_STRIDE = W_BIAS + 64;

    char* abuf = smem;
    char* tbuf = smem + ROWS_A * S;

    int tid = threadIdx.x;
    // (walked: everything derived from the thread id is recomputed per tile -
    // hoisted out of the walk it is carried, i.e. spilled, across it)
    if constexpr (WALK) asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int m_first = wm * MTW * 32;     // this wave's first channel
    const int ln = lane & 31, lh = lane >> 5;
    if (WM * WN == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);   // see the pair kernel

    int b, c_first, store_lo, store_n;
    if constexpr (WALK) {
        b = wk->b; c_first = wk->c_first;
        store_lo = wk->store_lo; store_n = wk->store_n;
    } else {
        const int wg = pm_xcd_remap(blockIdx.x, gridDim.x);
        const int tile = wg % a.ntiles;
        b = wg / a.ntiles;
        c_first = tile * a.TL - a.halo;   // time of column 0
        store_lo = a.halo; store_n = a.TL;
    }
    int L;
    if constexpr (WALK) {
        L = wk->L;
    } else {
        L = a.lengths ? min(a.lengths[b] * a.len_scale, a.L) : a.L;
        if (c_first + a.halo >= L) return;
    }
    const float* __restrict__ xb = a.x + (size_t)b * a.L * C;
    PM_STAMP(a, 0);
    // carry strips: rows [NC - halo - m, NC - halo) of a layer's input, which
    // the next tile (c_first + NC - halo) needs as columns [-m, 0)
    constexpr int QS16 = S / 16;
    [[maybe_unused]] const int keep = NC - a.halo;
    // (two strips in one pass over disjoint threads: the copies are a few
    // hundred bytes each and latency-bound, so they run side by side)
    [[maybe_unused]] auto strip_copy2 = [&](
        char* dst_a, const char* src_a, int rows_a, char* dst_b,
        const char* src_b, int rows_b) {
#if defined(PM_TUNING) && defined(PM_ABLATE_STRIP)   // timing experiment only: no carries
        return;
#endif
        const int na = rows_a * QS16, nb = rows_b * QS16;
        for (int i = tid; i < na + nb; i += NT) {
            const bool second = i >= na;
            const int k = second ? i - na : i;
            const float4 v = reinterpret_cast<const float4*>(
                second ? src_b : src_a)[k];
            reinterpret_cast<float4*>(second ? dst_b : dst_a)[k] = v;
        }
    };

    // ---- zero the margins (they stand for neighbours' columns: only ever
    // feed the recomputed halo, but must be finite) ------------------------
    {
        constexpr int QS = S / 16;
        // (walked: an opaque zero, materialised here - as a loop invariant of
        // the walk the four registers were spilled)
        float zero = 0.f;
        if constexpr (WALK) asm volatile("" : "+v"(zero));
        const float4 z = make_float4(zero, zero, zero, zero);
        // (walked: the last H2 d_0 rows of the left margin are the previous
        // tile's last columns of a_0 = lrelu(x), from the carry area - written
        // by the thread that would have zeroed them: two loops over the margin
        // put the zero and the carried value of one address in different waves
        // with nothing but their arrival order between them)
        int first_carried = MA * QS;
        if constexpr (WALK)
            if (wk->left) first_carried = (MA - H2 * wk->dil[0]) * QS;
        for (int i = tid; i < (WALK ? 1 : 2) * MA * QS; i += NT) {
            const int r = i / QS, q = i % QS;
            const int row = r < MA ? r : NC + r;
            float4 v = z;
            if constexpr (WALK)
                if (i >= first_carried)
                    v = reinterpret_cast<const float4*>(wk->carry)[i - first_carried];
            *reinterpret_cast<float4*>(abuf + row * S + q * 16) = v;
        }
        // (walked: no right margin behind `t` either - what lies behind it,
        // the carry area, is finite and only reaches right-halo columns)
        for (int i = tid; i < (WALK ? 1 : 2) * H2 * QS; i += NT) {
            const int r = i / QS, q = i % QS;
            const int row = r < H2 ? r : NC + r;
            *reinterpret_cast<float4*>(tbuf + row * S + q * 16) = z;
        }
    }
    // ---- trunk registers <- x in the MFMA C/D layout (zero outside the
    // utterance), and a = lrelu(x) -> LDS out of the same registers: the
    // workgroup's NC columns are exactly its waves' tiles, so x is read from
    // HBM once. All loads are issued before the first LDS write.
    // (buffer loads over a descriptor of exactly the rows of the utterance
    // this tile covers: a column outside reads as zero without a branch)
    floatx16 trunk[MTW][NTW];
    const int x_lo = max(c_first, 0);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(xb) + (size_t)x_lo * C, 0,
        max(min(L, c_first + NC) - x_lo, 0) * C * 4, 0x00020000);
    const unsigned xvoff0 = (unsigned)(
        ((c_first - x_lo + wn * NTW * 32 + ln) * C + m_first + 4 * lh) * 4);
    bool from_xnext = (XMODE & 1) != 0;
    if constexpr (WALK && (XMODE & 1)) from_xnext = wk->have_x != 0;
    if (from_xnext) {
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) trunk[mt][nt] = xnext[mt][nt];
    } else {
        const unsigned voff0 = xvoff0;
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const unsigned voff =
                    voff0 + (unsigned)((mt * 32 + nt * 32 * C) * 4);
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const pm_u4 v = __builtin_amdgcn_raw_buffer_load_b128(
                        xrsrc, voff + g4 * 32, 0, 0);
                    trunk[mt][nt][4 * g4 + 0] = __uint_as_float(v.x);
                    trunk[mt][nt][4 * g4 + 1] = __uint_as_float(v.y);
                    trunk[mt][nt][4 * g4 + 2] = __uint_as_float(v.z);
                    trunk[mt][nt][4 * g4 + 3] = __uint_as_float(v.w);
                }
            }
    }
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int col = (wn * NTW + nt) * 32 + ln;
            store_tile_lrelu_impl<ET, false>(
                abuf + (MA + col) * S, m_first + mt * 32, trunk[mt][nt], false,
                lh);
        }
    pm_block_sync();
    PM_STAMP(a, 1);

    const int col_off = (wn * NTW * 32 + ln) * S + lh * 8 * ET::ESZ;
    floatx16 acc[MTW][NTW];
    frag_t bf[MTW];
    frag_t afirst[G][MTW];
    {
        const void* first_stream;
        if constexpr (WALK) first_stream = wk->w1[0];
        else first_stream = a.w1[0];
        const frag_t* w1 = reinterpret_cast<const frag_t*>(first_stream) +
                           (size_t)(wm * MTW) * W_MT_STRIDE + lane;
        load_bias_frags<ET, MTW>(bf, w1 + W_BIAS, W_MT_STRIDE);
        load_a_group<ET, MTW, G>(afirst, w1, W_MT_STRIDE);
    }
    bias_start<ET, MTW, NTW>(acc, bf);

    // carry area: per iteration the last H2 d columns of `a`, then the last H2
    // columns of `t` - a.halo rows in all
    [[maybe_unused]] int coff = 0;
#pragma unroll 1
    for (int it = 0; it < a.niter; ++it) {
        auto stream1 = [&](int i) -> const void* {
            if constexpr (WALK) return wk->w1[i]; else return a.w1[i];
        };
        auto stream2 = [&](int i) -> const void* {
            if constexpr (WALK) return wk->w2[i]; else return a.w2[i];
        };
        int d;
        if constexpr (WALK) d = wk->dil[it]; else d = a.dil[it];
        const frag_t* w1 = reinterpret_cast<const frag_t*>(stream1(it)) +
                           (size_t)(wm * MTW) * W_MT_STRIDE + lane;
        const frag_t* w2 = reinterpret_cast<const frag_t*>(stream2(it)) +
                           (size_t)(wm * MTW) * W_MT_STRIDE + lane;

        // ---- conv1 (dilation d) out of `a` ----
#pragma unroll 1
        for (int c = 0; c < NCH; ++c)
            mma_taps<ET, K, KC, MTW, NTW, G, S>(
                acc, abuf + (MA - H2 * d) * S + col_off + c * CH * ET::ESZ,
                d * S, w1 + (size_t)c * W_CHUNK, W_MT_STRIDE, afirst,
                c + 1 < NCH ? w1 + (size_t)(c + 1) * W_CHUNK : w2);
        PM_STAMP(a, 2 + 4 * it);
        load_bias_frags<ET, MTW>(bf, w2 + W_BIAS, W_MT_STRIDE);
        if constexpr (WALK) {
            // `a` is stable until this iteration's epilogue 2 (behind the
            // next barrier): keep its last exact columns for the next tile;
            // and the left margin of `t` (last read by the previous conv2,
            // two barriers ago) takes the previous tile's columns
            char* cit = wk->carry + coff;
            const int ma = H2 * d;
            strip_copy2(cit, abuf + (keep + MA - ma) * S, ma, tbuf,
                        cit + ma * S, wk->left ? H2 : 0);
        }
        // (walked: epilogue addresses from a fresh copy of the thread id, so
        // that they are not carried across the MFMA loops)
        int te = tid;
        if constexpr (WALK) asm volatile("" : "+v"(te));
        const int ln1 = te & 31, lh1 = (te >> 5) & 1;
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int col_first = (wn * NTW + nt) * 32;
                const int col = col_first + ln1;
                const int t_tile = c_first + col_first;
                store_tile_lrelu<ET>(tbuf + (H2 + col) * S, m_first + mt * 32,
                                     acc[mt][nt], t_tile, L, ln1, lh1);
                if constexpr (XMODE & 2) {
                    bool request = it + 1 == a.niter;
                    if constexpr (WALK && (XMODE & 4))
                        request = request && wk->next_c_first >= 0;
                    if (request) {
                        // the tile whose x is wanted: this one (the next
                        // Block of an MRF launch) or the next of the walk
                        int cf = c_first;
                        if constexpr (WALK && (XMODE & 4)) cf = wk->next_c_first;
                        const int lo = max(cf, 0);
                        const __amdgpu_buffer_rsrc_t nrsrc =
                            __builtin_amdgcn_make_buffer_rsrc(
                                const_cast<float*>(xb) + (size_t)lo * C, 0,
                                max(min(L, cf + NC) - lo, 0) * C * 4,
                                0x00020000);
                        const unsigned xv = WALK
                            ? (unsigned)(((cf - lo + wn * NTW * 32 + ln1) *
                                          C + m_first + 4 * lh1) * 4)
                            : xvoff0;
                        const unsigned voff =
                            xv + (unsigned)((mt * 32 + nt * 32 * C) * 4);
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            const pm_u4 v =
                                __builtin_amdgcn_raw_buffer_load_b128(
                                    nrsrc, voff + g4 * 32, 0, 0);
                            acc[mt][nt][4 * g4 + 0] = __uint_as_float(v.x);
                            acc[mt][nt][4 * g4 + 1] = __uint_as_float(v.y);
                            acc[mt][nt][4 * g4 + 2] = __uint_as_float(v.z);
                            acc[mt][nt][4 * g4 + 3] = __uint_as_float(v.w);
                        }
                    }
                }
            }
        // ---- conv2 (dilation 1) out of `t`, accumulated IN PLACE onto the
        // fp32 trunk (the residual add is the MFMA's C operand) ----
        bias_add<ET, MTW, NTW>(trunk, bf);
        pm_block_sync();
        PM_STAMP(a, 3 + 4 * it);

        const bool last = it + 1 == a.niter;
#pragma unroll 1
        for (int c = 0; c < NCH; ++c)
            mma_taps<ET, K, KC, MTW, NTW, G, S>(
                trunk, tbuf + col_off + c * CH * ET::ESZ, S,
                w2 + (size_t)c * W_CHUNK, W_MT_STRIDE, afirst,
                c + 1 < NCH
                    ? w2 + (size_t)(c + 1) * W_CHUNK
                    : (last ? nullptr
                            : reinterpret_cast<const frag_t*>(stream1(it + 1)) +
                                  (size_t)(wm * MTW) * W_MT_STRIDE + lane));
        PM_STAMP(a, 4 + 4 * it);
        if (!last)
            load_bias_frags<ET, MTW>(
                bf, reinterpret_cast<const frag_t*>(stream1(it + 1)) +
                        (size_t)(wm * MTW) * W_MT_STRIDE + lane + W_BIAS,
                W_MT_STRIDE);
        if constexpr (WALK) {
            // `t` is stable until the next iteration's epilogue 1: keep its
            // last exact columns; the left margin of `a` (last read by this
            // iteration's conv1, behind a barrier) takes the previous tile's
            // columns of the NEXT iteration's input
            const int ma = H2 * d;
            char* cit = wk->carry + coff;
            coff += (ma + H2) * S;
            const int ma_next = !last && wk->left ? H2 * wk->dil[it + 1] : 0;
            strip_copy2(cit + ma * S, tbuf + keep * S, H2,
                        abuf + (MA - ma_next) * S, wk->carry + coff, ma_next);
        }
        int tf = tid;
        if constexpr (WALK) asm volatile("" : "+v"(tf));
        const int ln2 = tf & 31, lh2 = (tf >> 5) & 1;
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                if (!last) {
                    const int col_first = (wn * NTW + nt) * 32;
                    const int col = col_first + ln2;
                    const int t_tile = c_first + col_first;
                    store_tile_lrelu<ET>(abuf + (MA + col) * S,
                                         m_first + mt * 32, trunk[mt][nt],
                                         t_tile, L, ln2, lh2);
                }
            }
        if (!last) bias_start<ET, MTW, NTW>(acc, bf);
        if (!last) pm_block_sync();
        PM_STAMP(a, 5 + 4 * it);
    }

    if constexpr (XMODE & 2) {
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) xnext[mt][nt] = acc[mt][nt];
    }
    if constexpr (SUM == 1 || SUM == 2) {
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                if (SUM == 1) sum[mt][nt] = trunk[mt][nt];
                else sum[mt][nt] += trunk[mt][nt];
            }
        return;
    }
    if constexpr (SUM == 3) {
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) trunk[mt][nt] += sum[mt][nt];
    }
    // ---- store the valid interior (+ MRF accumulate) ----------------------
    // Buffer accesses over a descriptor of exactly the rows this tile owns:
    // halo columns and columns beyond the utterance are out of range (stores
    // dropped, loads 0) - no per-lane branches.
#if defined(PM_TUNING) && defined(PM_ABLATE_OUT)   // timing experiment only: nothing stored
    if (trunk[0][0][0] != 12345.f) return;
#endif
    const int mode = SUM == 3 ? 1 : a.mode;
    const float scale = a.scale;
    const int own_first = c_first + store_lo;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
        a.out + ((size_t)b * a.L + own_first) * C, 0,
        max(min(store_n, L - own_first), 0) * C * 4, 0x00020000);
    // (walked: the store offsets are computed here from a fresh copy of the
    // thread id - derived at the top of the Block they would be carried, and
    // spilled, across all of it)
    int tid_s = tid;
    if constexpr (WALK) asm volatile("" : "+v"(tid_s));
    const int lane_s = tid_s & 63;
    const int wave_s = __builtin_amdgcn_readfirstlane(tid_s >> 6);
    const unsigned ovoff0 = (unsigned)(
        (((wave_s % WN) * NTW * 32 + (lane_s & 31) - store_lo) * C +
         (wave_s / WN) * MTW * 32 + 4 * (lane_s >> 5)) * 4);
    if (mode == 2) {
        // the reads of `out` go out in batches before the first store of a
        // batch: one round trip per batch, not per 32 x 32 tile
        constexpr int NB = NTW > 4 ? (NTW + 1) / 2 : NTW;
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int n0 = 0; n0 < NTW; n0 += NB) {
                pm_u4 old[NB][4];
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    if (n0 + i >= NTW) break;
                    const unsigned voff = ovoff0 +
                        (unsigned)((mt * 32 + (n0 + i) * 32 * C) * 4);
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4)
                        old[i][g4] = __builtin_amdgcn_raw_buffer_load_b128(
                            orsrc, voff + g4 * 32, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    if (n0 + i >= NTW) break;
                    const int nt = n0 + i;
                    const unsigned voff =
                        ovoff0 + (unsigned)((mt * 32 + nt * 32 * C) * 4);
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const pm_u4 o = old[i][g4];
                        pm_u4 r;
                        r.x = __float_as_uint(__uint_as_float(o.x) +
                                              trunk[mt][nt][4 * g4 + 0] * scale);
                        r.y = __float_as_uint(__uint_as_float(o.y) +
                                              trunk[mt][nt][4 * g4 + 1] * scale);
                        r.z = __float_as_uint(__uint_as_float(o.z) +
                                              trunk[mt][nt][4 * g4 + 2] * scale);
                        r.w = __float_as_uint(__uint_as_float(o.w) +
                                              trunk[mt][nt][4 * g4 + 3] * scale);
                        __builtin_amdgcn_raw_buffer_store_b128(
                            r, orsrc, voff + g4 * 32, 0, 0);
                    }
                }
            }
    } else {
        const float sc = mode == 1 ? scale : 1.f;
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const unsigned voff =
                    ovoff0 + (unsigned)((mt * 32 + nt * 32 * C) * 4);
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    pm_u4 r;
                    r.x = __float_as_uint(trunk[mt][nt][4 * g4 + 0] * sc);
                    r.y = __float_as_uint(trunk[mt][nt][4 * g4 + 1] * sc);
                    r.z = __float_as_uint(trunk[mt][nt][4 * g4 + 2] * sc);
                    r.w = __float_as_uint(trunk[mt][nt][4 * g4 + 3] * sc);
                    __builtin_amdgcn_raw_buffer_store_b128(
                        r, orsrc, voff + g4 * 32, 0, 0);
                }
            }
    }
    PM_STAMP(a, 14);
}

// A 4-wave whole-Block workgroup (16-bit operands) must leave room for a
// second one on its CU: min 2 waves per SIMD caps it at 256 registers (left to
// itself the compiler took 320 and one workgroup owned the CU: block_c64_k3
// 1.18 ms instead of 0.81 ms, profiles/r02/ab_tile_variants.txt).
template <class ET, int C, int K, int WM, int WN, int NTW>
__global__ __launch_bounds__(WM * WN * 64,
                             WM * WN == 4 && ET::ESZ == 2 ? 2 : 1)
void conv_block3_kernel(Block3Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    floatx16 unused[(C / 32) / WM][NTW];
    block3_body<ET, C, K, WM, WN, NTW>(a, smem, unused, unused);
}

// Whole MRF stage (the three Blocks k = 3, 7, 11 of one upsampling stage,
// generator.py MRF: out = (B3(x) + B7(x) + B11(x)) / 3) in one launch: a
// workgroup runs the three Blocks back to back on ITS tile with the k = 11
// tiling, so the re-reads of x and the read-modify-write of `out` by the
// second and third Block find the lines this same workgroup just touched in
// L2 instead of sweeping HBM three times per Block. Same lanes touch the same
// addresses in every phase: program order is the only ordering needed.
struct MrfArgs { Block3Args k[3]; };

template <class ET, int C, int WM, int WN, int NTW, bool INREG>
__global__ __launch_bounds__(WM * WN * 64) void conv_mrf_kernel(MrfArgs m) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    floatx16 sum[(C / 32) / WM][NTW];
    if constexpr (INREG) {
        // the sum of the Blocks stays in registers (no read-modify-write of
        // `out`); k = 11 first: its MFMA loop has the highest register
        // pressure and runs before the sum is live
        // (the three Blocks start from the same x tile - the launcher checks
        // - and each requests it for the next one under its last conv2)
        // (split-f16 operands: fragments are twice as wide and the kernel is
        // MFMA-bound - no registers for the hand-over, every Block reads its
        // x tile itself, out of L2)
        constexpr int XN = ET::SPLIT ? 0 : 1;
        floatx16 xnext[(C / 32) / WM][NTW];
        block3_body<ET, C, 11, WM, WN, NTW, 1, 2 * XN>(m.k[2], smem, sum, xnext);
        pm_block_sync();
        block3_body<ET, C, 7, WM, WN, NTW, 2, 3 * XN>(m.k[1], smem, sum, xnext);
        pm_block_sync();
        block3_body<ET, C, 3, WM, WN, NTW, 3, 1 * XN>(m.k[0], smem, sum, xnext);
    } else {
        block3_body<ET, C, 3, WM, WN, NTW>(m.k[0], smem, sum, sum);
        pm_block_sync();
        block3_body<ET, C, 7, WM, WN, NTW>(m.k[1], smem, sum, sum);
        pm_block_sync();
        block3_body<ET, C, 11, WM, WN, NTW>(m.k[2], smem, sum, sum);
    }
}

// Whole MRF stage, walked (B3Walk): one workgroup per (utterance, segment)
// processes its tiles left to right; only a segment's first tile pays the
// two-sided halo. Carry areas of the three Blocks sit behind the k = 11 LDS
// layout.
// (a Block's carry area: halo rows; LDS of the walked layout: no right
// margin behind `a`)
template <class ET, int C>
__host__ __device__ constexpr int block3_carry_bytes(int halo) {
    return halo * (C * ET::ESZ + 16);
}
template <class ET, int C, int K, int WM, int WN, int NTW>
__host__ __device__ constexpr int block3_walk_smem_bytes() {
    constexpr int NC = WN * NTW * 32;
    constexpr int S = C * ET::ESZ + 16;
    return (NC + 5 * ((K - 1) / 2)) * S + (NC + (K - 1) / 2) * S;
}

// (compact arguments: the three Blocks share everything but their weights
// and dilations, and all of it stays live across the walk)
struct MrfWalkArgs {
    const float* x;
    float* out;
    const void* w1[3][3];   // [Block k 3 / 7 / 11][iteration]
    const void* w2[3][3];
    int dil[3][3];
    int B, L, halo;
    float scale;
    const int* lengths;
    int len_scale;
    int nseg;               // segments per utterance
};

template <class ET, int C, int WM, int WN, int NTW>
__global__ __launch_bounds__(WM * WN * 64) void conv_mrf_walk_kernel(
    MrfWalkArgs m) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NC = WN * NTW * 32;
    const int H = m.halo;
    const int b = blockIdx.x / m.nseg, seg = blockIdx.x % m.nseg;
    const int L = m.lengths ? min(m.lengths[b] * m.len_scale, m.L) : m.L;
    // segment boundaries: multiples of 32 columns
    const int per = (((L + m.nseg - 1) / m.nseg) + 31) & ~31;
    const int s0 = seg * per;
    const int e0 = min(L, s0 + per);
    if (s0 >= e0) return;

    // (all three Blocks are tiled with the k = 11 halo; their own carry
    // areas need 5 / 3 / 1 twelfths of it... bounded by H rows each)
    char* carry11 = smem + block3_walk_smem_bytes<ET, C, 11, WM, WN, NTW>();
    char* carry7 = carry11 + block3_carry_bytes<ET, C>(H);
    char* carry3 = carry7 + block3_carry_bytes<ET, C>(H);

    Block3Args common = {};     // (its weight / dilation arrays stay unused)
    common.x = m.x; common.out = m.out; common.niter = 3;
    common.B = m.B; common.L = m.L; common.mode = 1; common.scale = m.scale;
    common.halo = H; common.lengths = m.lengths; common.len_scale = m.len_scale;
    const MrfWalkArgs __attribute__((address_space(4)))* karg =
        (const MrfWalkArgs __attribute__((address_space(4)))*)
            __builtin_amdgcn_kernarg_segment_ptr();
    auto block = [&](B3Walk& w, int j, char* carry) {
        w.carry = carry;
        w.w1 = karg->w1[j]; w.w2 = karg->w2[j]; w.dil = karg->dil[j];
    };

    for (int i = threadIdx.x; i < 3 * H * (C * ET::ESZ + 16) / 16;
         i += WM * WN * 64)
        reinterpret_cast<float4*>(carry11)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    floatx16 sum[(C / 32) / WM][NTW];
    floatx16 xnext[(C / 32) / WM][NTW];
    int own = s0;          // next column to store
    int left = 0;
#pragma unroll 1
    while (own < e0) {
        B3Walk w;
        w.b = b;
        w.L = L;
        // (readfirstlane: the walk is wave-uniform - scalar registers)
        w.c_first = __builtin_amdgcn_readfirstlane(
            left ? own : (s0 == 0 ? 0 : own - H));
        w.store_lo = own - w.c_first;
        w.store_n = __builtin_amdgcn_readfirstlane(
            min(NC - H - w.store_lo, e0 - own));
        w.left = left;
        // x: requested by the previous tile's last Block for this tile's
        // first, by every Block for the next one of the same tile
        const int next_own = own + w.store_n;
        w.next_c_first = next_own < e0 ? next_own : -1;
        w.have_x = left;
        block(w, 2, carry11);
        block3_body<ET, C, 11, WM, WN, NTW, 1, 3, 1>(
            common, smem, sum, xnext, &w);
        pm_block_sync();
        w.have_x = 1;
        block(w, 1, carry7);
        block3_body<ET, C, 7, WM, WN, NTW, 2, 3, 1>(
            common, smem, sum, xnext, &w);
        pm_block_sync();
        block(w, 0, carry3);
        block3_body<ET, C, 3, WM, WN, NTW, 3, 7, 1>(
            common, smem, sum, xnext, &w);
        pm_block_sync();
        own += w.store_n;
        left = 1;
    }
}

// A whole Block, walked (see B3Walk): where the carry area still fits the LDS
// beside the two operand tiles (C = 64 at k 7, C = 128 at k 3).
struct Block3WalkArgs {
    Block3Args a;
    int nseg;               // segments per utterance
};

template <class ET, int C, int K, int WM, int WN, int NTW>
__global__ __launch_bounds__(WM * WN * 64) void conv_block3_walk_kernel(
    Block3WalkArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NC = WN * NTW * 32;
    const Block3Args& a = p.a;
    const int H = a.halo;
    const int b = blockIdx.x / p.nseg, seg = blockIdx.x % p.nseg;
    const int L = a.lengths ? min(a.lengths[b] * a.len_scale, a.L) : a.L;
    const int per = (((L + p.nseg - 1) / p.nseg) + 31) & ~31;
    const int s0 = seg * per;
    const int e0 = min(L, s0 + per);
    if (s0 >= e0) return;
    const Block3WalkArgs __attribute__((address_space(4)))* karg =
        (const Block3WalkArgs __attribute__((address_space(4)))*)
            __builtin_amdgcn_kernarg_segment_ptr();
    B3Walk w;
    w.b = b;
    w.L = L;
    w.carry = smem + block3_walk_smem_bytes<ET, C, K, WM, WN, NTW>();
    w.w1 = karg->a.w1; w.w2 = karg->a.w2; w.dil = karg->a.dil;
    // (the carry area doubles as what the last taps of `t` read behind its
    // last row: finite from the start)
    for (int i = threadIdx.x; i < H * (C * ET::ESZ + 16) / 16; i += WM * WN * 64)
        reinterpret_cast<float4*>(w.carry)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    floatx16 unused[(C / 32) / WM][NTW];
    floatx16 xnext[(C / 32) / WM][NTW];
    int own = s0, left = 0;
#pragma unroll 1
    while (own < e0) {
        // (readfirstlane: the walk is wave-uniform - scalar registers)
        w.c_first = __builtin_amdgcn_readfirstlane(
            left ? own : (s0 == 0 ? 0 : own - H));
        w.store_lo = own - w.c_first;
        w.store_n = __builtin_amdgcn_readfirstlane(
            min(NC - H - w.store_lo, e0 - own));
        w.left = left;
        // (x of the next tile is requested under this tile's last conv2)
        const int next_own = own + w.store_n;
        w.next_c_first = next_own < e0 ? next_own : -1;
        w.have_x = left;
        block3_body<ET, C, K, WM, WN, NTW, 0, 7, 1>(a, smem, unused, xnext,
                                                    &w);
        pm_block_sync();
        own += w.store_n;
        left = 1;
    }
}

// ---------------------------------------------------------------------------
// A whole Block, walked with SKEWED windows: no column is computed twice.
//
// The walked kernels above keep the windows of all six layers aligned, so the
// columns that lack right context (a.halo of them per tile) are recomputed by
// the next tile: 12 ... 31 % of the MFMA work. Here iteration i works on the
// window [c0 - 32 i, c0 - 32 i + NC): its input ends 32 columns further right
// than its output, which covers the right context of both of its convolutions
// (H2 (d + 1) <= 30). A step advances every window by exactly NC columns.
//   * the fp32 trunk stays in the accumulator layout; between two iterations
//     it moves one 32-column tile to the right in the register file (the
//     window moves 32 columns left): within a wave by register moves, from
//     wave to wave through the part of `t` the receiving wave's next
//     epilogue 1 overwrites anyway (4 KB, one more barrier per iteration);
//     the workgroup's first tile comes from a slot of scratch its last column
//     wave wrote one step earlier (slots alternate with the step's parity);
//   * the left context of a layer's input - the last 32 + H2 (d - 1) columns
//     of `a`, the last 2 H2 of `t` - is what the previous step left behind;
//     it travels through scratch too, each 16 bytes written and read back by
//     the same thread (the LDS holds the two operand tiles and nothing else:
//     C = 128 k 11 fits in 156 128 bytes);
//   * conv1 of iteration 0 needs lrelu(x) H2 (d_0 + 1) columns beyond the
//     trunk's window: a 32-column strip is staged next to it;
//   * everything a step needs from memory is requested under the MFMA loop in
//     front of it: x and the strip of the next step under the last conv2
//     (into the dead conv1 accumulators), an iteration's `a` carry under the
//     previous conv2, its `t` carry under its conv1.
// A segment's first step starts halo columns early (or one tile early at the
// utterance start) and stores nothing it cannot vouch for; its last step runs
// 32 (niter - 1) columns past the segment. Same arithmetic per column as
// every other tiling of the Block (tests: bit-identical).
// ---------------------------------------------------------------------------
// Weight-fragment prefetch depth (k16 steps) of the skewed kernels' MFMA loops:
// a conv's whole chunk for the 16-bit types; exact fp32 two steps; split f16 one
// (depth 2 spills at k 11 and measures the same) - except ElemF16A2, whose
// weight fragments are the 16-byte ones of f16: two steps
// (-3.7 % at k 11 over depth 1: profiles/r06/ab_a2g2.txt)
#ifndef PM_A2_SKEW_G
#define PM_A2_SKEW_G 2
#endif
template <class ET>
__host__ __device__ constexpr int pm_skew_prefetch_depth(int kc) {
    return ET::ESZ == 4
        ? (ET::SPLIT ? (ET::WSZ == 2 ? PM_A2_SKEW_G : 1) : 2) : kc;
}

struct Block3SkewArgs {
    Block3Args a;
    int nseg;               // segments per utterance
    int wg_scratch;         // bytes of scratch per workgroup
    char* scratch;
};

template <class ET, int C, int K, int WM, int WN, int NTW>
struct SkewGeom {
    static constexpr int NC = WN * NTW * 32;
    static constexpr int H2 = (K - 1) / 2;
    static constexpr int AL = 4 * H2;            // left margin of `a` (d <= 5)
    static constexpr int S = C * ET::ESZ + 16;
    static constexpr int RA = AL + 32 + NC;      // rows of `a`
    static constexpr int RT = NC + 2 * H2;       // rows of `t`
    static constexpr int MTW = (C / 32) / WM;
    static constexpr int NW = WM * WN;
    static constexpr int TB = MTW * 4096;        // a wave's boundary tiles, fp32
    static constexpr int CA = 32 + 4 * H2;       // rows of an `a` carry, at most
    static constexpr int OFF_AC = 2 * 2 * WM * TB;   // wrap slots [parity][iteration][wm]
    static constexpr int OFF_TC = OFF_AC + 3 * CA * S;
    static constexpr int SCRATCH = (OFF_TC + 3 * 2 * H2 * S + 255) & ~255;
    static constexpr int SMEM = (RA + RT) * S;
};

template <class ET, int C, int K, int WM, int WN, int NTW>
__global__ __launch_bounds__(WM * WN * 64) void conv_block3_skew_kernel(
    Block3SkewArgs p) {
    typedef SkewGeom<ET, C, K, WM, WN, NTW> GE;
    typedef typename ET::afrag_t frag_t;
    constexpr int CH = C < 64 ? C : 64;
    constexpr int NCH = C / CH;
    constexpr int KC = CH / 16;
    constexpr int MTW = GE::MTW;
    constexpr int NC = GE::NC;
    constexpr int NT = WM * WN * 64;
    constexpr int H2 = GE::H2;
    constexpr int AL = GE::AL;
    constexpr int S = GE::S;
    constexpr int QS = S / 16;
    // (split f16: weight prefetch depth 1 - depth 2 spills at k 11 and measures
    // the same; its B fragments two steps deep: -3 % at k 11, profiles/r04/ab_x3_skew.txt)
    constexpr int G = pm_skew_prefetch_depth<ET>(KC);
    constexpr int W_CHUNK = K * KC * 64;
    constexpr int W_BIAS = NCH * W_CHUNK;
    constexpr int W_MT_STRIDE = W_BIAS + 64;
    constexpr int AUX = 16;        // sc1: served by the L2, never by this CU's L1
    // (B-fragment depth of the MFMA loops: the default rule, or - split f16,
    // where this kernel has the registers the fused whole-MRF kernel lacks - 2)
    constexpr int PM_SKEW_BD = ET::SPLIT ? 2 : 0;
    static_assert(NTW >= 2, "the trunk shift needs two tiles per wave");
    // (the exchange slot is 64 B x 64 rows of the receiving wave's part of t: a
    // row of its MTW x 32 channels is 64 B with 16-bit operands, 128 B with the
    // 4-byte layouts)


    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* abuf = smem;
    char* tbuf = smem + GE::RA * S;

    const Block3Args& a = p.a;
    const int H = a.halo;
    const int b = blockIdx.x / p.nseg, seg = blockIdx.x % p.nseg;
    const int L = a.lengths ? min(a.lengths[b] * a.len_scale, a.L) : a.L;
    const int per = (((L + p.nseg - 1) / p.nseg) + 31) & ~31;
    const int s0 = seg * per;
    const int e0 = min(L, s0 + per);
    if (s0 >= e0) return;
    const Block3SkewArgs __attribute__((address_space(4)))* karg =
        (const Block3SkewArgs __attribute__((address_space(4)))*)
            __builtin_amdgcn_kernarg_segment_ptr();
    const int niter = a.niter;
    const int skew = 32 * (niter - 1);

    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(
        p.scratch + (size_t)blockIdx.x * p.wg_scratch, 0, GE::SCRATCH,
        0x00020000);
    const float* __restrict__ xb = a.x + (size_t)b * a.L * C;

    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int m_first = wm * MTW * 32;
    if (WM * WN == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);
    // Everything derived from the thread id is recomputed where it is used,
    // from an opaque copy: hoisted out of the walk it would be carried - i.e.
    // spilled - across the MFMA loops (trunk, the requested x and the operand
    // pipeline take 210 of the 256 registers)
    auto fresh_tid = []() {
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));
        return t;
    };
    auto col_offset = [&](int t) {
        return (wn * NTW * 32 + (t & 31)) * S + ((t >> 5) & 1) * 8 * ET::ESZ;
    };
    auto stream_of = [&](const void* w, int t) {
        return reinterpret_cast<const frag_t*>(w) +
               (size_t)(wm * MTW) * W_MT_STRIDE + (t & 63);
    };

    floatx16 trunk[MTW][NTW];
    floatx16 acc[MTW][NTW];
    frag_t bf[MTW];
    frag_t afirst[G][MTW];

    // LDS rows -> scratch (a layer's last columns, for the next step)
    auto rows_out = [&](int tid, const char* src, int rows, unsigned soff) {
        const int n = rows * QS;
        for (int i = tid; i < n; i += NT) {
            const float4 v = reinterpret_cast<const float4*>(src)[i];
            const pm_u4 u = {__float_as_uint(v.x), __float_as_uint(v.y),
                             __float_as_uint(v.z), __float_as_uint(v.w)};
            __builtin_amdgcn_raw_buffer_store_b128(
                u, srsrc, soff + (unsigned)i * 16, 0, 0);
        }
    };
    // scratch -> registers (issued early) -> LDS rows (a layer's left margin);
    // `dead` pushes the offsets out of range - zeros - on a segment's first step
    constexpr int Q = C / 4;
    constexpr int CIN = (GE::CA * QS + NT - 1) / NT;     // an `a` carry
    constexpr int TIN = (2 * H2 * QS + NT - 1) / NT;     // a `t` carry
    constexpr int SIT = (32 * Q + NT - 1) / NT;          // the x strip (fp32)
    constexpr int PRE = CIN > SIT ? CIN : SIT;
    auto rows_request = [&](int tid, auto& r, int rows, unsigned soff,
                            unsigned dead) {
        constexpr int N = sizeof(r) / sizeof(r[0]);
#pragma unroll
        for (int it = 0; it < N; ++it) {
            const int i = tid + it * NT;
            r[it] = __builtin_amdgcn_raw_buffer_load_b128(
                srsrc, (i < rows * QS ? soff + (unsigned)i * 16 : 0x40000000u) + dead,
                0, AUX);
        }
    };
    auto rows_in = [&](int tid, char* dst, const auto& r, int rows) {
        constexpr int N = sizeof(r) / sizeof(r[0]);
#pragma unroll
        for (int it = 0; it < N; ++it) {
            const int i = tid + it * NT;
            if (i < rows * QS)
                reinterpret_cast<pm_u4*>(dst)[i] = r[it];
        }
    };

#ifdef PM_TUNING
    // phase totals of wave 0 (shader clocks): 0 stage, 1 conv1, 2 epilogue 1,
    // 3 its barrier, 4 conv2, 5 epilogue 2 (hand-over included), 6 unused,
    // 7 its barrier, 8 output store, 9 steps
    unsigned long long ph[10] = {};
    unsigned long long last = __builtin_amdgcn_s_memtime();
#define PM_SKEW_MARK(k)                                                       \
    do {                                                                      \
        const unsigned long long now = __builtin_amdgcn_s_memtime();          \
        ph[k] += now - last; last = now;                                      \
    } while (0)
#else
#define PM_SKEW_MARK(k) ((void)0)
#endif
    // the 32 columns of x behind a step's window (fp32, all channels)
    auto strip_request = [&](int tid, pm_u4 (&r)[PRE], int c_first) {
        const int lo = max(c_first, 0);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(xb) + (size_t)lo * C, 0,
            max(min(L, c_first + 32) - lo, 0) * C * 4, 0x00020000);
#pragma unroll
        for (int it = 0; it < SIT; ++it) {
            const int idx = tid + it * NT;
            r[it] = __builtin_amdgcn_raw_buffer_load_b128(
                rsrc, (unsigned)(((c_first - lo) * C + idx * 4) * 4), 0, 0);
        }
    };
    // requested under a conv2: the next iteration's `a` carry, or (last
    // iteration) the next step's x strip
    pm_u4 pre[PRE];
    int left = 0, par = 0;
#pragma unroll 1
    // (a segment's first step has no t on the first H2 columns of iteration
    // 0's window: garbage inside the warm-up columns of a later segment; at the
    // utterance start the walk begins one tile early, where they are padding)
    for (int c0 = s0 == 0 ? -32 : s0 - H; c0 - skew < e0;
         c0 += NC, left = 1, par ^= 1) {
        const unsigned dead = left ? 0u : 0x40000000u;
        {
            const frag_t* w1 = stream_of(karg->a.w1[0], fresh_tid());
            load_bias_frags<ET, MTW>(bf, w1 + W_BIAS, W_MT_STRIDE);
            load_a_group<ET, MTW, G>(afirst, w1, W_MT_STRIDE);
        }
        // ---- trunk <- x on [c0, c0 + NC), a_0 = lrelu(x) on [c0, c0 + NC + 32)
        {
            const int tid = fresh_tid();
            const int ln = tid & 31, lh = (tid >> 5) & 1;
            const int x_lo = max(c0, 0);
            const __amdgpu_buffer_rsrc_t xrsrc =
                __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(xb) + (size_t)x_lo * C, 0,
                    max(min(L, c0 + NC + 32) - x_lo, 0) * C * 4, 0x00020000);
            // (a segment's first step: the strip first, its round trip runs
            // under the tile stores; later steps requested it under conv2)
            if (!left) strip_request(tid, pre, c0 + NC);
            if (left) {
                // requested by the previous step under its last conv2
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) trunk[mt][nt] = acc[mt][nt];
            } else {
                const unsigned voff0 = (unsigned)(
                    ((c0 - x_lo + wn * NTW * 32 + ln) * C + m_first + 4 * lh) * 4);
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) {
                        const unsigned voff =
                            voff0 + (unsigned)((mt * 32 + nt * 32 * C) * 4);
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            const pm_u4 v = __builtin_amdgcn_raw_buffer_load_b128(
                                xrsrc, voff + g4 * 32, 0, 0);
                            trunk[mt][nt][4 * g4 + 0] = __uint_as_float(v.x);
                            trunk[mt][nt][4 * g4 + 1] = __uint_as_float(v.y);
                            trunk[mt][nt][4 * g4 + 2] = __uint_as_float(v.z);
                            trunk[mt][nt][4 * g4 + 3] = __uint_as_float(v.w);
                        }
                    }
            }
            const int d0 = karg->a.dil[0];
            pm_u4 cin[CIN] = {};
            if (d0 > 1) rows_request(tid, cin, H2 * (d0 - 1), GE::OFF_AC, dead);
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    const int col = (wn * NTW + nt) * 32 + ln;
                    store_tile_lrelu_impl<ET, false>(
                        abuf + (AL + col) * S, m_first + mt * 32, trunk[mt][nt],
                        false, lh);
                }
#pragma unroll
            for (int it = 0; it < SIT; ++it) {
                const int idx = tid + it * NT;
                const int r = idx / Q, q = idx % Q;
                if (idx < 32 * Q)
                    ET::store4_at(abuf + (AL + NC + r) * S, q * 4,
                                  pm_lrelu4(make_float4(
                                   __uint_as_float(pre[it].x),
                                   __uint_as_float(pre[it].y),
                                   __uint_as_float(pre[it].z),
                                   __uint_as_float(pre[it].w))));
            }
            if (d0 > 1)
                rows_in(tid, abuf + (AL + H2 - H2 * d0) * S, cin, H2 * (d0 - 1));
        }
        // (once per step: what the previous step stored into scratch - the
        // wrap slots are read by ANOTHER wave - has left this wave before the
        // barriers those reads sit behind)
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        pm_block_sync();
        PM_SKEW_MARK(0);

#pragma unroll 1
        for (int it = 0; it < niter; ++it) {
            const int d = karg->a.dil[it];
            const int o = c0 - 32 * it;           // time of the window's column 0
            const bool more = it + 1 < niter;

            // ---- conv1 (dilation d): t on [o + H2, o + H2 + NC) ----
            bias_start<ET, MTW, NTW>(acc, bf);
            // the previous step's last 2 H2 columns of t: requested here,
            // written behind the tiles of epilogue 1
            pm_u4 tcin[TIN];
            {
                const int t1 = fresh_tid();
                rows_request(t1, tcin, 2 * H2, GE::OFF_TC + it * 2 * H2 * S, dead);
                const frag_t* w1 = stream_of(karg->a.w1[it], t1);
                const frag_t* w2 = stream_of(karg->a.w2[it], t1);
                const int col_off = col_offset(t1);
#pragma unroll 1
                for (int c = 0; c < NCH; ++c)
                    mma_taps<ET, K, KC, MTW, NTW, G, S, NoHook, PM_SKEW_BD>(
                        acc, abuf + (AL + H2 - H2 * d) * S + col_off + c * CH * ET::ESZ,
                        d * S, w1 + (size_t)c * W_CHUNK, W_MT_STRIDE, afirst,
                        c + 1 < NCH ? w1 + (size_t)(c + 1) * W_CHUNK : w2);
            }
            PM_SKEW_MARK(1);
            const int tid = fresh_tid();
            const int ln = tid & 31, lh = (tid >> 5) & 1;
            load_bias_frags<ET, MTW>(
                bf, stream_of(karg->a.w2[it], tid) + W_BIAS, W_MT_STRIDE);
            // `a` is stable until epilogue 2: what the next step needs of it
            {
                const int rows = (it ? 32 : 0) + H2 * (d - 1);
                rows_out(tid, abuf + (AL + NC + H2 - H2 * d) * S, rows,
                         GE::OFF_AC + it * GE::CA * S);
            }
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    const int col_first = (wn * NTW + nt) * 32;
                    store_tile_lrelu<ET>(
                        tbuf + (2 * H2 + col_first + ln) * S, m_first + mt * 32,
                        acc[mt][nt], o + H2 + col_first, L, ln, lh);
                    // x of the next step, into the (now dead) conv1
                    // accumulators: the round trip runs under the last conv2
                    if (!more) {
                        const int cn = c0 + NC;           // (>= 0)
                        const __amdgpu_buffer_rsrc_t nrsrc =
                            __builtin_amdgcn_make_buffer_rsrc(
                                const_cast<float*>(xb) + (size_t)cn * C, 0,
                                (cn - skew < e0 ? max(min(L, cn + NC) - cn, 0) : 0) *
                                    C * 4, 0x00020000);
                        const unsigned voff = (unsigned)(
                            ((col_first + ln) * C + m_first + mt * 32 + 4 * lh) * 4);
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            const pm_u4 v = __builtin_amdgcn_raw_buffer_load_b128(
                                nrsrc, voff + g4 * 32, 0, 0);
                            acc[mt][nt][4 * g4 + 0] = __uint_as_float(v.x);
                            acc[mt][nt][4 * g4 + 1] = __uint_as_float(v.y);
                            acc[mt][nt][4 * g4 + 2] = __uint_as_float(v.z);
                            acc[mt][nt][4 * g4 + 3] = __uint_as_float(v.w);
                        }
                    }
                }
            rows_in(tid, tbuf, tcin, 2 * H2);
            // ---- conv2 (dilation 1) onto the trunk, in place ----
            bias_add<ET, MTW, NTW>(trunk, bf);
            PM_SKEW_MARK(2);
            pm_block_sync();
            PM_SKEW_MARK(3);
            {
                const int t2 = fresh_tid();
                if (more)
                    rows_request(t2, pre, 32 + H2 * (karg->a.dil[it + 1] - 1),
                                 GE::OFF_AC + (it + 1) * GE::CA * S, dead);
                else if (c0 + NC - skew < e0)
                    strip_request(t2, pre, c0 + 2 * NC);
                const frag_t* w2 = stream_of(karg->a.w2[it], t2);
                const frag_t* w1n =
                    stream_of(karg->a.w1[more ? it + 1 : it], t2);
                const int col_off = col_offset(t2);
#pragma unroll 1
                for (int c = 0; c < NCH; ++c)
                    mma_taps<ET, K, KC, MTW, NTW, G, S, NoHook, PM_SKEW_BD>(
                        trunk, tbuf + col_off + c * CH * ET::ESZ, S,
                        w2 + (size_t)c * W_CHUNK, W_MT_STRIDE, afirst,
                        c + 1 < NCH ? w2 + (size_t)(c + 1) * W_CHUNK
                                    : (more ? w1n : nullptr));
            }
            PM_SKEW_MARK(4);
            const int te = fresh_tid();
            const int ln2 = te & 31, lh2 = (te >> 5) & 1;
            const unsigned lane_slot = (unsigned)((te & 63) * 16);
            if (more)
                load_bias_frags<ET, MTW>(
                    bf, stream_of(karg->a.w1[it + 1], te) + W_BIAS, W_MT_STRIDE);
            // `t` is stable until the next epilogue 1
            rows_out(te, tbuf + NC * S, 2 * H2, GE::OFF_TC + it * 2 * H2 * S);
            if (more) {
                // The wave's last tile leaves for its right-hand neighbour:
                // through the part of `t` the NEIGHBOUR's epilogue 1 will
                // overwrite (64 rows x its own 32 MTW channels = the tile's
                // 4 KB; nobody else writes there, and the neighbour reads it
                // before it does) once every wave is done with conv2. The
                // workgroup's last column wave hands its tile to the FIRST one
                // of the next step through scratch.
                pm_block_sync();
                if (wn + 1 < WN) {
                    char* slot = tbuf +
                        (2 * H2 + (wn + 1) * NTW * 32 + (te & 63)) * S +
                        m_first * ET::ESZ;
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4)
                            *reinterpret_cast<float4*>(
                                slot + mt * 32 * ET::ESZ + g4 * 16) =
                                acc_quad(trunk[mt][NTW - 1], g4);
                } else {
                    const unsigned xo =
                        (unsigned)(((par * 2 + it) * WM + wm) * GE::TB) +
                        lane_slot;
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            const floatx16& v = trunk[mt][NTW - 1];
                            const pm_u4 u = {__float_as_uint(v[4 * g4 + 0]),
                                             __float_as_uint(v[4 * g4 + 1]),
                                             __float_as_uint(v[4 * g4 + 2]),
                                             __float_as_uint(v[4 * g4 + 3])};
                            __builtin_amdgcn_raw_buffer_store_b128(
                                u, srsrc, xo + mt * 4096 + g4 * 1024, 0, 0);
                        }
                }
                const int dn = karg->a.dil[it + 1];
                // a_{it+1} = lrelu(trunk) on [o, o + NC): rows AL + 32 ... of
                // the next iteration's frame
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) {
                        const int col_first = (wn * NTW + nt) * 32;
                        store_tile_lrelu<ET>(
                            abuf + (AL + 32 + col_first + ln2) * S,
                            m_first + mt * 32, trunk[mt][nt], o + col_first, L,
                            ln2, lh2);
                    }
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                    for (int nt = NTW - 1; nt > 0; --nt)
                        trunk[mt][nt] = trunk[mt][nt - 1];
                rows_in(te, abuf + (AL + H2 - H2 * dn) * S, pre,
                        32 + H2 * (dn - 1));
                PM_SKEW_MARK(5);
                pm_block_sync();
                PM_SKEW_MARK(7);
                if (wn > 0) {
                    const char* slot = tbuf +
                        (2 * H2 + wn * NTW * 32 + (te & 63)) * S +
                        m_first * ET::ESZ;
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            const float4 v = *reinterpret_cast<const float4*>(
                                slot + mt * 32 * ET::ESZ + g4 * 16);
                            trunk[mt][0][4 * g4 + 0] = v.x;
                            trunk[mt][0][4 * g4 + 1] = v.y;
                            trunk[mt][0][4 * g4 + 2] = v.z;
                            trunk[mt][0][4 * g4 + 3] = v.w;
                        }
                } else {
                    // (written one step ago, many barriers and drained loads
                    // back; `dead`: zeros on a segment's first step)
                    const unsigned xi =
                        (unsigned)((((par ^ 1) * 2 + it) * WM + wm) * GE::TB) +
                        dead;
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            const pm_u4 v = __builtin_amdgcn_raw_buffer_load_b128(
                                srsrc, xi + lane_slot + mt * 4096 + g4 * 1024, 0,
                                AUX);
                            trunk[mt][0][4 * g4 + 0] = __uint_as_float(v.x);
                            trunk[mt][0][4 * g4 + 1] = __uint_as_float(v.y);
                            trunk[mt][0][4 * g4 + 2] = __uint_as_float(v.z);
                            trunk[mt][0][4 * g4 + 3] = __uint_as_float(v.w);
                        }
                }
            }
        }

        // ---- store [o, o + NC) of the last iteration, clipped to the
        // segment: out-of-range rows are dropped by the descriptor ----
        const int o = c0 - skew;
        const int ts = fresh_tid();
        const int ln = ts & 31, lh = (ts >> 5) & 1;
        const int own_first = max(s0, o);
        const int own_n = min(e0, o + NC) - own_first;
        const float scale = a.scale;
        const int mode = a.mode;
        const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
            a.out + ((size_t)b * a.L + own_first) * C, 0,
            max(own_n, 0) * C * 4, 0x00020000);
        const unsigned ovoff0 = (unsigned)(
            ((o - own_first + wn * NTW * 32 + ln) * C + m_first + 4 * lh) * 4);
        const bool a16 = a.act16 != nullptr;
        const bool a16f = a.act16_type == 1;       // PM_F16, else PM_BF16
        const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<char*>(a.act16) +
                ((size_t)b * a.L + own_first) * C * 2, 0,
            a16 ? max(own_n, 0) * C * 2 : 0, 0x00020000);
        // (MRF accumulation: the reads of `out` of up to four tiles go out
        // before the first store - the operand pipeline's registers are free
        // here -, one HBM round trip per step instead of one per tile)
        const float sc = mode == 0 ? 1.f : scale;
        constexpr int NB = MTW * NTW < 4 ? MTW * NTW : 4;
#pragma unroll
        for (int t0 = 0; t0 < MTW * NTW; t0 += NB) {
            pm_u4 old[NB][4] = {};
            if (mode == 2) {
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    if (t0 + i >= MTW * NTW) break;
                    const int mt = (t0 + i) / NTW, nt = (t0 + i) % NTW;
                    const unsigned voff =
                        ovoff0 + (unsigned)((mt * 32 + nt * 32 * C) * 4);
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4)
                        old[i][g4] = __builtin_amdgcn_raw_buffer_load_b128(
                            orsrc, voff + g4 * 32, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                if (t0 + i >= MTW * NTW) break;
                const int mt = (t0 + i) / NTW, nt = (t0 + i) % NTW;
                const unsigned voff =
                    ovoff0 + (unsigned)((mt * 32 + nt * 32 * C) * 4);
                float4 rq[4];
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
                    rq[g4] = make_float4(
                        __uint_as_float(old[i][g4].x) +
                            trunk[mt][nt][4 * g4 + 0] * sc,
                        __uint_as_float(old[i][g4].y) +
                            trunk[mt][nt][4 * g4 + 1] * sc,
                        __uint_as_float(old[i][g4].z) +
                            trunk[mt][nt][4 * g4 + 2] * sc,
                        __uint_as_float(old[i][g4].w) +
                            trunk[mt][nt][4 * g4 + 3] * sc);
                if (a16) {
                    // the stage's output as the next upsampler's operand:
                    // lrelu, 16-bit, 8 consecutive channels per lane (the
                    // half-wave exchange of store_tile_lrelu_impl)
                    const unsigned v16 = (voff - 16u * lh) / 2;
#pragma unroll
                    for (int g4 = 0; g4 < 4; g4 += 2) {
                        const float4 lo = pm_lrelu4(rq[g4]);
                        const float4 hi = pm_lrelu4(rq[g4 + 1]);
                        const uint2 pa = a16f ? ElemF16::pack4(lo)
                                              : ElemBF16::pack4(lo);
                        const uint2 pb = a16f ? ElemF16::pack4(hi)
                                              : ElemBF16::pack4(hi);
                        auto sx = __builtin_amdgcn_permlane32_swap(
                            pa.x, pb.x, false, false);
                        auto sy = __builtin_amdgcn_permlane32_swap(
                            pa.y, pb.y, false, false);
                        const pm_u4 u = {sx[0], sy[0], sx[1], sy[1]};
                        __builtin_amdgcn_raw_buffer_store_b128(
                            u, arsrc, v16 + 16 * (g4 + lh), 0, 0);
                    }
                } else {
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const pm_u4 r = {
                            __float_as_uint(rq[g4].x), __float_as_uint(rq[g4].y),
                            __float_as_uint(rq[g4].z), __float_as_uint(rq[g4].w)};
                        __builtin_amdgcn_raw_buffer_store_b128(
                            r, orsrc, voff + g4 * 32, 0, 0);
                    }
                }
            }
        }
        PM_SKEW_MARK(8);
#ifdef PM_TUNING
        ph[9] += 1;
#endif
    }
#ifdef PM_TUNING
    if (a.timeline && threadIdx.x == 0)
        for (int i = 0; i < 10; ++i) a.timeline[(size_t)blockIdx.x * 16 + i] = ph[i];
#endif
}

// ---------------------------------------------------------------------------
// The whole MRF of a stage (Blocks k 11, 7, 3 back to back) on the skewed walk:
// conv_block3_skew_kernel's step, three times per window, with the sum of the
// Blocks in registers. Against three skewed Block launches (what the 4-byte
// operand layouts - split f16, exact fp32 - run at C = 32) a step reads x from
// HBM once instead of three times (the second and third Block re-request the
// tile the first one just pulled through the L2, under the previous Block's
// last conv2), and `out` is written once instead of being written, then read
// and written twice more: 1.8 GB of traffic per launch instead of 7.2, two
// HBM round trips and two store phases a step less. Every Block keeps its own
// carries and wrap slots in its own third of the workgroup's scratch; the LDS
// holds one Block's two operand tiles at a time (sized for k 11). All three
// Blocks run iteration i on the window [c0 - 32 i, ...), so their outputs
// cover the same columns and add up in the accumulator layout.
// Same arithmetic per column as the Block-by-Block launches except the order
// of the final sum ((B11 + B7 + B3) / 3 with ONE scaling, as the whole-MRF
// walk of the 16-bit types does).
// ---------------------------------------------------------------------------
struct MrfSkewArgs {
    const float* x;
    float* out;
    const void* w1[3][3];   // [Block k 3 / 7 / 11][iteration]
    const void* w2[3][3];
    int dil[3][3];
    int B, L, halo;         // halo: the k 11 Block's (the segments' warm-up)
    float scale;
    const int* lengths;
    int len_scale;
    int nseg;               // segments per utterance
    int wg_scratch;         // bytes of scratch per workgroup
    char* scratch;
    PM_TIMELINE_FIELD       // debug phase totals (tuning builds)
};

template <class ET, int C, int WM, int WN, int NTW>
struct MrfSkewGeom {
    typedef SkewGeom<ET, C, 11, WM, WN, NTW> G11;
    static constexpr int BLOCK_SCRATCH = G11::SCRATCH;    // (the largest)
    static constexpr int SCRATCH = 3 * BLOCK_SCRATCH;
    static constexpr int SMEM = G11::SMEM;
};

template <class ET, int C, int WM, int WN, int NTW>
__global__ __launch_bounds__(WM * WN * 64) void conv_mrf_skew_kernel(
    MrfSkewArgs p) {
    typedef MrfSkewGeom<ET, C, WM, WN, NTW> MG;
    typedef typename ET::afrag_t frag_t;
    constexpr int CH = C < 64 ? C : 64;
    constexpr int NCH = C / CH;
    constexpr int KC = CH / 16;
    constexpr int MTW = (C / 32) / WM;
    constexpr int NC = WN * NTW * 32;
    constexpr int NT = WM * WN * 64;
    constexpr int S = C * ET::ESZ + 16;
    constexpr int QS = S / 16;
    constexpr int G = pm_skew_prefetch_depth<ET>(KC);
    constexpr int AUX = 16;        // sc1: served by the L2, never by this CU's L1
    constexpr int PM_SKEW_BD = ET::SPLIT ? 2 : 0;
    constexpr int TB = MTW * 4096;
    constexpr int Q = C / 4;
    constexpr int SIT = (32 * Q + NT - 1) / NT;          // the x strip (fp32)
    constexpr int CIN11 = ((32 + 4 * 5) * QS + NT - 1) / NT;
    constexpr int PRE = CIN11 > SIT ? CIN11 : SIT;
    static_assert(NTW >= 2, "the trunk shift needs two tiles per wave");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* abuf = smem;

    const int H = p.halo;
    const int b = blockIdx.x / p.nseg, seg = blockIdx.x % p.nseg;
    const int L = p.lengths ? min(p.lengths[b] * p.len_scale, p.L) : p.L;
    const int per = (((L + p.nseg - 1) / p.nseg) + 31) & ~31;
    const int s0 = seg * per;
    const int e0 = min(L, s0 + per);
    if (s0 >= e0) return;
    const MrfSkewArgs __attribute__((address_space(4)))* karg =
        (const MrfSkewArgs __attribute__((address_space(4)))*)
            __builtin_amdgcn_kernarg_segment_ptr();
    constexpr int skew = 64;            // 32 (niter - 1), niter == 3

    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(
        p.scratch + (size_t)blockIdx.x * p.wg_scratch, 0, MG::SCRATCH,
        0x00020000);
    const float* __restrict__ xb = p.x + (size_t)b * p.L * C;

    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int m_first = wm * MTW * 32;
    if (WM * WN == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);
    auto fresh_tid = []() {
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));
        return t;
    };
    auto col_offset = [&](int t) {
        return (wn * NTW * 32 + (t & 31)) * S + ((t >> 5) & 1) * 8 * ET::ESZ;
    };

    floatx16 trunk[MTW][NTW];
    floatx16 acc[MTW][NTW];
    floatx16 sum[MTW][NTW];
    frag_t bf[MTW];
    frag_t afirst[G][MTW];

    auto rows_out = [&](int tid, const char* src, int rows, unsigned soff) {
        const int n = rows * QS;
        for (int i = tid; i < n; i += NT) {
            const float4 v = reinterpret_cast<const float4*>(src)[i];
            const pm_u4 u = {__float_as_uint(v.x), __float_as_uint(v.y),
                             __float_as_uint(v.z), __float_as_uint(v.w)};
            __builtin_amdgcn_raw_buffer_store_b128(
                u, srsrc, soff + (unsigned)i * 16, 0, 0);
        }
    };
    auto rows_request = [&](int tid, auto& r, int rows, unsigned soff,
                            unsigned dead) {
        constexpr int N = sizeof(r) / sizeof(r[0]);
#pragma unroll
        for (int it = 0; it < N; ++it) {
            const int i = tid + it * NT;
            r[it] = __builtin_amdgcn_raw_buffer_load_b128(
                srsrc, (i < rows * QS ? soff + (unsigned)i * 16 : 0x40000000u) + dead,
                0, AUX);
        }
    };
    auto rows_in = [&](int tid, char* dst, const auto& r, int rows) {
        constexpr int N = sizeof(r) / sizeof(r[0]);
#pragma unroll
        for (int it = 0; it < N; ++it) {
            const int i = tid + it * NT;
            if (i < rows * QS)
                reinterpret_cast<pm_u4*>(dst)[i] = r[it];
        }
    };
    auto strip_request = [&](int tid, pm_u4 (&r)[PRE], int c_first) {
        const int lo = max(c_first, 0);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(xb) + (size_t)lo * C, 0,
            max(min(L, c_first + 32) - lo, 0) * C * 4, 0x00020000);
#pragma unroll
        for (int it = 0; it < SIT; ++it) {
            const int idx = tid + it * NT;
            r[it] = __builtin_amdgcn_raw_buffer_load_b128(
                rsrc, (unsigned)(((c_first - lo) * C + idx * 4) * 4), 0, 0);
        }
    };
    pm_u4 pre[PRE];
    int left = 0, par = 0;
#ifdef PM_TUNING
    // phase totals of wave 0 per Block (shader clocks): 0 stage, 1 conv1,
    // 2 epilogue 1, 3 its barrier, 4 conv2, 5 epilogue 2 + hand-over, 7 its
    // barrier, 8 sum / store, 9 steps
    unsigned long long ph[3][10] = {};
    unsigned long long last_mark = __builtin_amdgcn_s_memtime();
#define PM_MRF_MARK(k)                                                        \
    do {                                                                      \
        const unsigned long long now = __builtin_amdgcn_s_memtime();          \
        ph[POS][k] += now - last_mark; last_mark = now;                       \
    } while (0)
#else
#define PM_MRF_MARK(k) ((void)0)
#endif

    // One Block (kernel size KV::value) on the window of this step. POS: 0 the
    // first Block of the step (x and the strip come from the previous STEP's
    // requests, or are loaded here on a segment's first step), 1 the middle
    // one, 2 the last (requests the next step's x; adds up and stores).
    auto block_step = [&](auto kv, auto posv, const int c0) {
        constexpr int K = decltype(kv)::value;
        constexpr int POS = decltype(posv)::value;
        constexpr int J = K == 3 ? 0 : K == 7 ? 1 : 2;     // row of the weight arrays
        typedef SkewGeom<ET, C, K, WM, WN, NTW> GE;
        constexpr int H2 = GE::H2;
        constexpr int AL = GE::AL;
        constexpr int CIN = (GE::CA * QS + NT - 1) / NT;
        constexpr int TIN = (2 * H2 * QS + NT - 1) / NT;
        constexpr int W_CHUNK = K * KC * 64;
        constexpr int W_BIAS = NCH * W_CHUNK;
        constexpr int W_MT_STRIDE = W_BIAS + 64;
        constexpr unsigned SB = (unsigned)(POS * MG::BLOCK_SCRATCH);
        char* tbuf = smem + GE::RA * S;
        auto stream_of = [&](const void* w, int t) {
            return reinterpret_cast<const frag_t*>(w) +
                   (size_t)(wm * MTW) * W_MT_STRIDE + (t & 63);
        };
        const unsigned dead = left ? 0u : 0x40000000u;
        const bool have_x = POS > 0 || left;
        // (the LDS is laid out per kernel size: the previous Block's last
        // conv2 may still be reading rows this one's staging writes)
        pm_block_sync();
        {
            const frag_t* w1 = stream_of(karg->w1[J][0], fresh_tid());
            load_bias_frags<ET, MTW>(bf, w1 + W_BIAS, W_MT_STRIDE);
            load_a_group<ET, MTW, G>(afirst, w1, W_MT_STRIDE);
        }
        // ---- trunk <- x on [c0, c0 + NC), a_0 = lrelu(x) on [c0, c0 + NC + 32)
        {
            const int tid = fresh_tid();
            const int ln = tid & 31, lh = (tid >> 5) & 1;
            const int x_lo = max(c0, 0);
            const __amdgpu_buffer_rsrc_t xrsrc =
                __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(xb) + (size_t)x_lo * C, 0,
                    max(min(L, c0 + NC + 32) - x_lo, 0) * C * 4, 0x00020000);
            if (!have_x) strip_request(tid, pre, c0 + NC);
            if (have_x) {
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) trunk[mt][nt] = acc[mt][nt];
            } else {
                const unsigned voff0 = (unsigned)(
                    ((c0 - x_lo + wn * NTW * 32 + ln) * C + m_first + 4 * lh) * 4);
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) {
                        const unsigned voff =
                            voff0 + (unsigned)((mt * 32 + nt * 32 * C) * 4);
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            const pm_u4 v = __builtin_amdgcn_raw_buffer_load_b128(
                                xrsrc, voff + g4 * 32, 0, 0);
                            trunk[mt][nt][4 * g4 + 0] = __uint_as_float(v.x);
                            trunk[mt][nt][4 * g4 + 1] = __uint_as_float(v.y);
                            trunk[mt][nt][4 * g4 + 2] = __uint_as_float(v.z);
                            trunk[mt][nt][4 * g4 + 3] = __uint_as_float(v.w);
                        }
                    }
            }
            const int d0 = karg->dil[J][0];
            pm_u4 cin[CIN] = {};
            if (d0 > 1)
                rows_request(tid, cin, H2 * (d0 - 1), SB + GE::OFF_AC, dead);
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    const int col = (wn * NTW + nt) * 32 + ln;
                    store_tile_lrelu_impl<ET, false>(
                        abuf + (AL + col) * S, m_first + mt * 32, trunk[mt][nt],
                        false, lh);
                }
#pragma unroll
            for (int it = 0; it < SIT; ++it) {
                const int idx = tid + it * NT;
                const int r = idx / Q, q = idx % Q;
                if (idx < 32 * Q)
                    ET::store4_at(abuf + (AL + NC + r) * S, q * 4,
                                  pm_lrelu4(make_float4(
                                   __uint_as_float(pre[it].x),
                                   __uint_as_float(pre[it].y),
                                   __uint_as_float(pre[it].z),
                                   __uint_as_float(pre[it].w))));
            }
            if (d0 > 1)
                rows_in(tid, abuf + (AL + H2 - H2 * d0) * S, cin, H2 * (d0 - 1));
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): see the Block kernel
        pm_block_sync();
        PM_MRF_MARK(0);

#pragma unroll 1
        for (int it = 0; it < 3; ++it) {
            const int d = karg->dil[J][it];
            const int o = c0 - 32 * it;
            const bool more = it + 1 < 3;

            bias_start<ET, MTW, NTW>(acc, bf);
            pm_u4 tcin[TIN];
            {
                const int t1 = fresh_tid();
                rows_request(t1, tcin, 2 * H2,
                             SB + GE::OFF_TC + it * 2 * H2 * S, dead);
                const frag_t* w1 = stream_of(karg->w1[J][it], t1);
                const frag_t* w2 = stream_of(karg->w2[J][it], t1);
                const int col_off = col_offset(t1);
#pragma unroll 1
                for (int c = 0; c < NCH; ++c)
                    mma_taps<ET, K, KC, MTW, NTW, G, S, NoHook, PM_SKEW_BD>(
                        acc, abuf + (AL + H2 - H2 * d) * S + col_off + c * CH * ET::ESZ,
                        d * S, w1 + (size_t)c * W_CHUNK, W_MT_STRIDE, afirst,
                        c + 1 < NCH ? w1 + (size_t)(c + 1) * W_CHUNK : w2);
            }
            PM_MRF_MARK(1);
            const int tid = fresh_tid();
            const int ln = tid & 31, lh = (tid >> 5) & 1;
            load_bias_frags<ET, MTW>(
                bf, stream_of(karg->w2[J][it], tid) + W_BIAS, W_MT_STRIDE);
            {
                const int rows = (it ? 32 : 0) + H2 * (d - 1);
                rows_out(tid, abuf + (AL + NC + H2 - H2 * d) * S, rows,
                         SB + GE::OFF_AC + it * GE::CA * S);
            }
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    const int col_first = (wn * NTW + nt) * 32;
                    store_tile_lrelu<ET>(
                        tbuf + (2 * H2 + col_first + ln) * S, m_first + mt * 32,
                        acc[mt][nt], o + H2 + col_first, L, ln, lh);
                    // x for the NEXT Block of this step (the same window) or,
                    // from the last Block, for the next step - into the dead
                    // conv1 accumulators, under the last conv2
                    if (!more) {
                        const int cn = POS == 2 ? c0 + NC : c0;
                        const int lo = max(cn, 0);
                        const bool wanted = POS < 2 || cn - skew < e0;
                        const __amdgpu_buffer_rsrc_t nrsrc =
                            __builtin_amdgcn_make_buffer_rsrc(
                                const_cast<float*>(xb) + (size_t)lo * C, 0,
                                (wanted ? max(min(L, cn + NC) - lo, 0) : 0) *
                                    C * 4, 0x00020000);
                        const unsigned voff = (unsigned)(
                            ((cn - lo + col_first + ln) * C + m_first +
                             mt * 32 + 4 * lh) * 4);
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            const pm_u4 v = __builtin_amdgcn_raw_buffer_load_b128(
                                nrsrc, voff + g4 * 32, 0, 0);
                            acc[mt][nt][4 * g4 + 0] = __uint_as_float(v.x);
                            acc[mt][nt][4 * g4 + 1] = __uint_as_float(v.y);
                            acc[mt][nt][4 * g4 + 2] = __uint_as_float(v.z);
                            acc[mt][nt][4 * g4 + 3] = __uint_as_float(v.w);
                        }
                    }
                }
            rows_in(tid, tbuf, tcin, 2 * H2);
            bias_add<ET, MTW, NTW>(trunk, bf);
            PM_MRF_MARK(2);
            pm_block_sync();
            PM_MRF_MARK(3);
            {
                const int t2 = fresh_tid();
                if (more)
                    rows_request(t2, pre, 32 + H2 * (karg->dil[J][it + 1] - 1),
                                 SB + GE::OFF_AC + (it + 1) * GE::CA * S, dead);
                else if (POS < 2)
                    strip_request(t2, pre, c0 + NC);
                else if (c0 + NC - skew < e0)
                    strip_request(t2, pre, c0 + 2 * NC);
                const frag_t* w2 = stream_of(karg->w2[J][it], t2);
                const frag_t* w1n =
                    stream_of(karg->w1[J][more ? it + 1 : it], t2);
                const int col_off = col_offset(t2);
#pragma unroll 1
                for (int c = 0; c < NCH; ++c)
                    mma_taps<ET, K, KC, MTW, NTW, G, S, NoHook, PM_SKEW_BD>(
                        trunk, tbuf + col_off + c * CH * ET::ESZ, S,
                        w2 + (size_t)c * W_CHUNK, W_MT_STRIDE, afirst,
                        c + 1 < NCH ? w2 + (size_t)(c + 1) * W_CHUNK
                                    : (more ? w1n : nullptr));
            }
            PM_MRF_MARK(4);
            const int te = fresh_tid();
            const int ln2 = te & 31, lh2 = (te >> 5) & 1;
            const unsigned lane_slot = (unsigned)((te & 63) * 16);
            if (more)
                load_bias_frags<ET, MTW>(
                    bf, stream_of(karg->w1[J][it + 1], te) + W_BIAS, W_MT_STRIDE);
            rows_out(te, tbuf + NC * S, 2 * H2, SB + GE::OFF_TC + it * 2 * H2 * S);
            if (more) {
                pm_block_sync();
                if (wn + 1 < WN) {
                    char* slot = tbuf +
                        (2 * H2 + (wn + 1) * NTW * 32 + (te & 63)) * S +
                        m_first * ET::ESZ;
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4)
                            *reinterpret_cast<float4*>(
                                slot + mt * 32 * ET::ESZ + g4 * 16) =
                                acc_quad(trunk[mt][NTW - 1], g4);
                } else {
                    const unsigned xo = SB +
                        (unsigned)(((par * 2 + it) * WM + wm) * TB) + lane_slot;
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            const floatx16& v = trunk[mt][NTW - 1];
                            const pm_u4 u = {__float_as_uint(v[4 * g4 + 0]),
                                             __float_as_uint(v[4 * g4 + 1]),
                                             __float_as_uint(v[4 * g4 + 2]),
                                             __float_as_uint(v[4 * g4 + 3])};
                            __builtin_amdgcn_raw_buffer_store_b128(
                                u, srsrc, xo + mt * 4096 + g4 * 1024, 0, 0);
                        }
                }
                const int dn = karg->dil[J][it + 1];
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) {
                        const int col_first = (wn * NTW + nt) * 32;
                        store_tile_lrelu<ET>(
                            abuf + (AL + 32 + col_first + ln2) * S,
                            m_first + mt * 32, trunk[mt][nt], o + col_first, L,
                            ln2, lh2);
                    }
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                    for (int nt = NTW - 1; nt > 0; --nt)
                        trunk[mt][nt] = trunk[mt][nt - 1];
                rows_in(te, abuf + (AL + H2 - H2 * dn) * S, pre,
                        32 + H2 * (dn - 1));
                PM_MRF_MARK(5);
                pm_block_sync();
                PM_MRF_MARK(7);
                if (wn > 0) {
                    const char* slot = tbuf +
                        (2 * H2 + wn * NTW * 32 + (te & 63)) * S +
                        m_first * ET::ESZ;
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            const float4 v = *reinterpret_cast<const float4*>(
                                slot + mt * 32 * ET::ESZ + g4 * 16);
                            trunk[mt][0][4 * g4 + 0] = v.x;
                            trunk[mt][0][4 * g4 + 1] = v.y;
                            trunk[mt][0][4 * g4 + 2] = v.z;
                            trunk[mt][0][4 * g4 + 3] = v.w;
                        }
                } else {
                    const unsigned xi = SB +
                        (unsigned)((((par ^ 1) * 2 + it) * WM + wm) * TB) + dead;
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            const pm_u4 v = __builtin_amdgcn_raw_buffer_load_b128(
                                srsrc, xi + lane_slot + mt * 4096 + g4 * 1024, 0,
                                AUX);
                            trunk[mt][0][4 * g4 + 0] = __uint_as_float(v.x);
                            trunk[mt][0][4 * g4 + 1] = __uint_as_float(v.y);
                            trunk[mt][0][4 * g4 + 2] = __uint_as_float(v.z);
                            trunk[mt][0][4 * g4 + 3] = __uint_as_float(v.w);
                        }
                }
            }
        }

        // ---- the Block's result on [c0 - 64, c0 - 64 + NC): into the sum ----
        if constexpr (POS == 0) {
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) sum[mt][nt] = trunk[mt][nt];
        } else if constexpr (POS == 1) {
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) sum[mt][nt] += trunk[mt][nt];
        } else {
            // store (sum of the three) * scale, clipped to the segment:
            // out-of-range rows are dropped by the descriptor
            const int o = c0 - skew;
            const int ts = fresh_tid();
            const int ln = ts & 31, lh = (ts >> 5) & 1;
            const int own_first = max(s0, o);
            const int own_n = min(e0, o + NC) - own_first;
            const float scale = p.scale;
            const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
                p.out + ((size_t)b * p.L + own_first) * C, 0,
                max(own_n, 0) * C * 4, 0x00020000);
            const unsigned ovoff0 = (unsigned)(
                ((o - own_first + wn * NTW * 32 + ln) * C + m_first + 4 * lh) * 4);
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    const unsigned voff =
                        ovoff0 + (unsigned)((mt * 32 + nt * 32 * C) * 4);
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const pm_u4 r = {
                            __float_as_uint((sum[mt][nt][4 * g4 + 0] +
                                             trunk[mt][nt][4 * g4 + 0]) * scale),
                            __float_as_uint((sum[mt][nt][4 * g4 + 1] +
                                             trunk[mt][nt][4 * g4 + 1]) * scale),
                            __float_as_uint((sum[mt][nt][4 * g4 + 2] +
                                             trunk[mt][nt][4 * g4 + 2]) * scale),
                            __float_as_uint((sum[mt][nt][4 * g4 + 3] +
                                             trunk[mt][nt][4 * g4 + 3]) * scale)};
                        __builtin_amdgcn_raw_buffer_store_b128(
                            r, orsrc, voff + g4 * 32, 0, 0);
                    }
                }
        }
        PM_MRF_MARK(8);
#ifdef PM_TUNING
        ph[POS][9] += 1;
#endif
    };

    typedef std::integral_constant<int, 0> P0;
    typedef std::integral_constant<int, 1> P1;
    typedef std::integral_constant<int, 2> P2;
#pragma unroll 1
    for (int c0 = s0 == 0 ? -32 : s0 - H; c0 - skew < e0;
         c0 += NC, left = 1, par ^= 1) {
        block_step(std::integral_constant<int, 11>(), P0(), c0);
        block_step(std::integral_constant<int, 7>(), P1(), c0);
        block_step(std::integral_constant<int, 3>(), P2(), c0);
    }
#ifdef PM_TUNING
    if (p.timeline && threadIdx.x == 0)
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 10; ++i)
                p.timeline[((size_t)blockIdx.x * 3 + j) * 16 + i] = ph[j][i];
#endif
}
