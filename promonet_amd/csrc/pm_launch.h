// Host-side launchers: pick the template instantiation (tile geometry) for a
// layer shape. Explicitly instantiated once per operand type in
// pm_conv_{f16,bf16,f32}.hip so the three compile in parallel.
#pragma once
#include <map>
#include <mutex>
#include <utility>

#include "pm_conv.h"

// Operand types (ElemXX::ID = the PM_* dtype code) whose C = 32 / 64 stages
// run Block by Block on the skewed walk instead of the fused whole-MRF launch:
// split f16 (ID 3; three MFMAs per step, so the whole-MRF tiling's 23 % halo
// recompute shows: profiles/r04/ab_x3_skew.txt) and - measured separately,
// profiles/r05/ab_x3skew_f32.txt - exact fp32 (ID 0; 16 times the MFMA time per
// step, the same argument). ONE predicate for the launcher's template choice
// (pm_launch_block3) and the engine's fusion decision (forward_impl).
#ifndef PM_X3SKEW_F32
#define PM_X3SKEW_F32 1
#endif
// (A/B builds: -DPM_SKEW16_C32=1 sends the 16-bit C = 32 stage the same way)
#ifndef PM_SKEW16_C32
#define PM_SKEW16_C32 0
#endif
#ifndef PM_A2_SKEW
#define PM_A2_SKEW 1
#endif
// the skewed whole-MRF walk for those types (-DPM_MRF_SKEW=0: A/B builds)
#ifndef PM_MRF_SKEW
#define PM_MRF_SKEW 1
#endif
constexpr bool pm_x3skew_id(int id) {
    return id == 3 || (PM_A2_SKEW && id == 4) || (PM_X3SKEW_F32 && id == 0) ||
           (PM_SKEW16_C32 && (id == 1 || id == 2));
}

// Opt a kernel into `bytes` of dynamic LDS (> 48 KB needs the attribute).
// The grant is a property of (kernel, DEVICE): cached per pair, so one process
// driving several GPUs sets it on each, and guarded so that concurrent
// launches from several host threads / streams are safe.
inline hipError_t pm_ensure_dynamic_lds(const void* kern, int bytes) {
    static std::mutex guard;
    static std::map<std::pair<const void*, int>, int> granted;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(guard);
    int& have = granted[std::make_pair(kern, dev)];
    if (bytes > have) {
        e = hipFuncSetAttribute(
            kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
        have = bytes;
    }
    return hipSuccess;
}

// Compute units of the current device, queried once per device (the walked
// launchers size their grids from it on every forward).
inline int pm_device_cus() {
    static std::mutex guard;
    static std::map<int, int> cus_of;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    std::lock_guard<std::mutex> lock(guard);
    auto it = cus_of.find(dev);
    if (it != cus_of.end()) return it->second;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount,
                              dev) != hipSuccess)
        cus = 0;
    cus_of[dev] = cus;
    return cus;
}

// Test hooks (pm_debug_force): the walked kernels and the multi-block path of
// the wide upsampler are chosen from the grid size, which unit-sized inputs
// never reach. walk_nseg > 0: take the walked variant wherever it exists, with
// that many segments per utterance; upsample_groups > 0: that many M groups
// per column tile. 0 = the production heuristics.
// skew: -1 = never the skewed walk, 0 = where it measured faster, 1 = wherever
// it fits.
#ifndef PM_SKEW_DEFAULT
#define PM_SKEW_DEFAULT 0    // (A/B builds: -DPM_SKEW_DEFAULT=-1)
#endif
struct PmForce { int walk_nseg = 0; int upsample_groups = 0; int skew = PM_SKEW_DEFAULT; };
// scratch the skewed walk takes per workgroup, at most
#define PM_SKEW_WG_SCRATCH (256 << 10)
// (per host thread: see pm_debug_force in pm_api.hip)
inline PmForce& pm_force() { static thread_local PmForce f; return f; }

// Latency (narrow-tile) variants can be switched off for A/B runs in a
// -DPM_TUNING build only; the shipped library reads no environment.
inline bool pm_narrow_allowed() {
#ifdef PM_TUNING
    static const bool allowed = !getenv("PM_NO_NARROW");
    return allowed;
#else
    return true;
#endif
}

template <class ET> hipError_t pm_launch_pair(
    int C, int K, const PairArgs& args, hipStream_t stream);
template <class ET> int pm_pair_tile_len(int C, int K);
// input-channel chunk size the pair kernel's weight stream is packed with
template <class ET> int pm_pair_chunk(int C);
// whether pm_launch_block3 has an instantiation for (C, K)
template <class ET> bool pm_block3_supported(int C, int K);

// Whole-Block fusion for C <= 64; returns hipErrorNotSupported when the shape
// has no instantiation (caller falls back to the pair kernel).
template <class ET> hipError_t pm_launch_block3(
    int C, int K, const Block3Args& args, hipStream_t stream);

// Whole MRF stage (Blocks k = 3, 7, 11 in that order) in one launch for
// C <= 64; hipErrorNotSupported otherwise (caller launches Block by Block).
template <class ET> hipError_t pm_launch_mrf(
    int C, const Block3Args (&blocks)[3], hipStream_t stream);

// kind 0: plain conv with KT = KSPAN = 7 (input conv); kind 1: polyphase
// ConvTranspose (KT = 2, KSPAN = 3). cfg: 0 = 256 x 128 tile, 1 = 64 x 128,
// 3 = 128 x 128 (measured: a 512 x 128 tile is 20 % slower than cfg 0),
// 2 = 32 x 128.
template <class ET> hipError_t pm_launch_single(
    int kind, int ch, int cfg, const SingleArgs& args, hipStream_t stream);
hipError_t pm_launch_stft(int epi, const SingleArgs& args, hipStream_t stream);

#ifdef PM_INSTANTIATE

// ---- fused pair geometry per (operand type, C) ---------------------------
template <class ET, int C> struct PairCfg;
// 16-bit operands: every wave owns a 32 (co) x 128 (time) register tile, so
// one A fragment (weights, streamed L2 -> VGPR) feeds 4 MFMAs: the vector
// memory path, not the matrix pipe, was the limiter at 2 loads per 4 MFMAs
// (rocprof r01: MFMA busy 43 %). C = 128 takes 256-column tiles
// (LDS 160,480 B at k 11, d 5).
// CH: input channels staged per LDS chunk (one barrier per chunk). Measured:
// CH = 128 (half the barriers, LDS tiles aliased) is 4 % slower than 64.
// 192 columns per weight fetch (LDS tiles aliased to fit): the pair kernels
// run at the power wall with the L2 94 % busy streaming weights, so fewer L2
// bytes per MFMA is what buys clock (measured -7 % on k 7 / k 11).
// (r02: 64 x 128 wave tiles on 256-column workgroup tiles with CH = 32 are
// 3-5 % slower at every k - profiles/r02/ab_tile_variants.txt)
template <> struct PairCfg<ElemF16, 256> { enum { WM = 8, WN = 1, NTW = 6, CH = 64, ALIAS = 1 }; };
// C = 128 alternatives measured and dropped: 384-column tiles with CH = 32
// (neutral), two 4-wave 128-column workgroups per CU (neutral: L2 weight
// traffic doubles), one fat wave per SIMD with 64 x 128 tiles (+1...+4 %).
// r02: 4-wave workgroups of 64 x 128 wave tiles, two per CU (CH = 32, tiles
// aliased) spill the staging registers and run 8-15 % slower; the same tiling
// fed by LDS-DMA from a 16-bit copy of lrelu(x) (no staging registers, no
// spills) is still 3-6 % slower, 8 waves on 512 columns 7-13 % slower
// (profiles/r02/ab_pair_dma_*.txt).
template <> struct PairCfg<ElemF16, 128> { enum { WM = 4, WN = 2, NTW = 4, CH = 64, ALIAS = 0 }; };
template <> struct PairCfg<ElemF16, 64>  { enum { WM = 2, WN = 2, NTW = 2, CH = 64, ALIAS = 0 }; };
template <> struct PairCfg<ElemF16, 32>  { enum { WM = 1, WN = 4, NTW = 1, CH = 32, ALIAS = 0 }; };
template <int C> struct PairCfg<ElemBF16, C> : PairCfg<ElemF16, C> {};
// exact fp32 operands: LDS rows are twice as wide -> 64-column tiles
// (round 4: 128-column tiles with the conv1 -> conv2 tile overlaying the x
// chunks instead of 64-column ones: 8 % of the columns recomputed instead of
// 16 % at k 11 and half the weight bytes per column - config 2 22.1 -> 20.4 ms;
// 192 columns at C = 128, where they fit: 19.7 ms - profiles/r04/ab_x3_skew.txt)
template <> struct PairCfg<ElemF32, 256> { enum { WM = 4, WN = 2, NTW = 2, CH = 64, ALIAS = 1 }; };
template <> struct PairCfg<ElemF32, 128> { enum { WM = 4, WN = 2, NTW = 3, CH = 64, ALIAS = 1 }; };
template <> struct PairCfg<ElemF32, 64>  { enum { WM = 2, WN = 2, NTW = 1, CH = 64, ALIAS = 0 }; };
template <> struct PairCfg<ElemF32, 32>  { enum { WM = 1, WN = 4, NTW = 1, CH = 32, ALIAS = 0 }; };
// split f16 (hi + lo, three MFMAs per step): 4 bytes per element like fp32,
// the same tiles
template <int C> struct PairCfg<ElemF16X3, C> : PairCfg<ElemF32, C> {};
// activations split, weights single f16: the same LDS tiles
template <int C> struct PairCfg<ElemF16A2, C> : PairCfg<ElemF32, C> {};

// Latency variant: when the wide tiling yields fewer workgroups than the chip
// has CUs (single utterances: 8 tiles at C = 256 for 2 s of audio), 64-column
// (C = 256) / 128-column (C = 128) tiles put 3x / 2x as many CUs to work on a
// third / half of the per-tile critical path. More halo recompute and weight
// traffic per column, so only below PM_NARROW_BELOW workgroups.
template <class ET, int C> struct PairCfgNarrow : PairCfg<ET, C> {};
// (CH must equal the wide variant's: both read the same packed weights)
template <> struct PairCfgNarrow<ElemF16, 256> { enum { WM = 8, WN = 1, NTW = 2, CH = PairCfg<ElemF16, 256>::CH, ALIAS = 1 }; };
template <> struct PairCfgNarrow<ElemF16, 128> { enum { WM = 4, WN = 2, NTW = 2, CH = PairCfg<ElemF16, 128>::CH, ALIAS = 0 }; };
// (fp32: the 64-column tiles of rounds 1-3 as the latency variant)
template <> struct PairCfgNarrow<ElemF32, 256> { enum { WM = 4, WN = 2, NTW = 1, CH = 64, ALIAS = 0 }; };
template <> struct PairCfgNarrow<ElemF32, 128> { enum { WM = 4, WN = 2, NTW = 1, CH = 64, ALIAS = 0 }; };
template <> struct PairCfgNarrow<ElemF16X3, 256> : PairCfgNarrow<ElemF32, 256> {};
template <> struct PairCfgNarrow<ElemF16X3, 128> : PairCfgNarrow<ElemF32, 128> {};
template <> struct PairCfgNarrow<ElemF16A2, 256> : PairCfgNarrow<ElemF32, 256> {};
template <> struct PairCfgNarrow<ElemF16A2, 128> : PairCfgNarrow<ElemF32, 128> {};
template <> struct PairCfgNarrow<ElemBF16, 256> : PairCfgNarrow<ElemF16, 256> {};
template <> struct PairCfgNarrow<ElemBF16, 128> : PairCfgNarrow<ElemF16, 128> {};
#define PM_NARROW_BELOW 150   // 3x the tiles must still fit ~2 rounds of 256 CUs

template <class ET, int C, int K, class G>
static hipError_t launch_pair_cfg(const PairArgs& a0, hipStream_t stream) {
    constexpr int WM = G::WM, WN = G::WN, NTW = G::NTW, CH = G::CH;
    constexpr int ALIAS = G::ALIAS;
    constexpr int TL = WN * NTW * 32 - (K - 1);
    PairArgs a = a0;
    a.ntiles = (a.L + TL - 1) / TL;
    auto kern = conv_pair_kernel<ET, C, K, WM, WN, NTW, CH, ALIAS>;
    const int smem =
        pair_smem_bytes<ET, C, K, WM, WN, NTW, CH, ALIAS>(a.dilation);
    hipError_t e = pm_ensure_dynamic_lds(
        reinterpret_cast<const void*>(kern), smem);
    if (e != hipSuccess) return e;
    const int grid = a.ntiles * a.B;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * WN * 64), smem,
                       stream, a);
    return hipGetLastError();
}

template <class ET, int C, int K>
static hipError_t launch_pair_ck(const PairArgs& a, hipStream_t stream) {
    typedef PairCfg<ET, C> W;
    typedef PairCfgNarrow<ET, C> N;
    if constexpr ((int)N::NTW != (int)W::NTW || (int)N::WN != (int)W::WN) {
        constexpr int TL = W::WN * W::NTW * 32 - (K - 1);
        const bool allowed = pm_narrow_allowed();
        if (allowed && (long long)((a.L + TL - 1) / TL) * a.B < PM_NARROW_BELOW)
            return launch_pair_cfg<ET, C, K, N>(a, stream);
    }
    return launch_pair_cfg<ET, C, K, W>(a, stream);
}

template <class ET, int C>
static hipError_t launch_pair_c(int K, const PairArgs& a, hipStream_t s) {
    switch (K) {
        case 3: return launch_pair_ck<ET, C, 3>(a, s);
        case 7: return launch_pair_ck<ET, C, 7>(a, s);
        case 11: return launch_pair_ck<ET, C, 11>(a, s);
    }
    return hipErrorInvalidValue;
}

template <class ET>
hipError_t pm_launch_pair(int C, int K, const PairArgs& a, hipStream_t s) {
    switch (C) {
        case 256: return launch_pair_c<ET, 256>(K, a, s);
        case 128: return launch_pair_c<ET, 128>(K, a, s);
        case 64: return launch_pair_c<ET, 64>(K, a, s);
        case 32: return launch_pair_c<ET, 32>(K, a, s);
    }
    return hipErrorInvalidValue;
}

template <class ET>
int pm_pair_tile_len(int C, int K) {
    switch (C) {
        case 256: return PairCfg<ET, 256>::WN * PairCfg<ET, 256>::NTW * 32 - (K - 1);
        case 128: return PairCfg<ET, 128>::WN * PairCfg<ET, 128>::NTW * 32 - (K - 1);
        case 64: return PairCfg<ET, 64>::WN * PairCfg<ET, 64>::NTW * 32 - (K - 1);
        case 32: return PairCfg<ET, 32>::WN * PairCfg<ET, 32>::NTW * 32 - (K - 1);
    }
    return 0;
}

template <class ET>
int pm_pair_chunk(int C) {
    switch (C) {
        case 256: return PairCfg<ET, 256>::CH;
        case 128: return PairCfg<ET, 128>::CH;
        case 64: return PairCfg<ET, 64>::CH;
        case 32: return PairCfg<ET, 32>::CH;
    }
    return 0;
}

// ---- whole-Block fusion ---------------------------------------------------
// Geometry per (operand type, C, K). WM == 0: no whole-Block instantiation
// (the caller runs one pair kernel per iteration instead). Measured: halving
// the columns PER WAVE so that two workgroups share a CU is slower (C = 64
// k 3: 1.65 vs 1.22 ms - the fixed cost per barrier phase dominates), so a
// wave always keeps the widest register tile.
template <class ET, int C, int K> struct Block3Cfg { enum { WM = 0, WN = 1, NTW = 1 }; };
// Small k (halo <= 36 columns): two 4-wave workgroups per CU with the same
// per-wave work beat one 8-wave workgroup (their phases interleave): -14 %
// at C = 32 k 3, -5 % at C = 32 k 7, -13 % at C = 64 k 3. With a larger halo
// (k 11, or k 7 at C = 64) the extra recompute eats the gain; at C = 128 it is
// 28 % slower.
template <> struct Block3Cfg<ElemF16, 32, 3>   { enum { WM = 1, WN = PM_SKEW16_C32 ? 8 : 4, NTW = 3 }; };
template <> struct Block3Cfg<ElemF16, 32, 7>   { enum { WM = 1, WN = PM_SKEW16_C32 ? 8 : 4, NTW = 3 }; };
template <> struct Block3Cfg<ElemF16, 32, 11>  { enum { WM = 1, WN = 8, NTW = 3 }; };
template <> struct Block3Cfg<ElemF16, 64, 3>   { enum { WM = 2, WN = 2, NTW = 4 }; };
template <> struct Block3Cfg<ElemF16, 64, 7>   { enum { WM = 2, WN = 4, NTW = 4 }; };
template <> struct Block3Cfg<ElemF16, 64, 11>  { enum { WM = 2, WN = 4, NTW = 4 }; };
#ifndef PM_K3_NTW
#define PM_K3_NTW 4     // (A/B builds: 3 tiles per wave at C = 128 k 3)
#endif
template <> struct Block3Cfg<ElemF16, 128, 3>  { enum { WM = 4, WN = 2, NTW = PM_K3_NTW }; };
// C = 128, k 7: walked only (conv_block3_walk_kernel; stand-alone, the 36-column
// halo on both sides of a 256-column tile made it 28 % slower than three pair
// launches - walked it is 10.6 % faster, profiles/r02/ab_block128_k7_walk.txt)
template <> struct Block3Cfg<ElemF16, 128, 7>  { enum { WM = 4, WN = 2, NTW = 4 }; };
// C = 128, k 11: skewed walk only (conv_block3_skew_kernel)
template <> struct Block3Cfg<ElemF16, 128, 11> { enum { WM = 4, WN = 2, NTW = 4 }; };
// C = 256, k 3: walked only, 128-column tiles (12 of them halo): 0.67 -> 0.53 ms
// against three pair launches (profiles/r02/ab_block256_walk.txt)
template <> struct Block3Cfg<ElemF16, 256, 3>  { enum { WM = 8, WN = 1, NTW = 4 }; };
// C = 256, k 7: skewed walk only
template <> struct Block3Cfg<ElemF16, 256, 7>  { enum { WM = 8, WN = 1, NTW = 4 }; };
// (k 7 the same way: 36 of 128 columns halo, 1.45 vs 1.19 ms - stays on the pair kernel)
// (k 11 the same way - it fits 160 KB once `t` loses its right margin too - is
// 7 % slower than three pair launches: 31 % more MFMA work)
template <int C, int K> struct Block3Cfg<ElemBF16, C, K> : Block3Cfg<ElemF16, C, K> {};
template <> struct Block3Cfg<ElemF32, 32, 3>   { enum { WM = 1, WN = 8, NTW = 2 }; };
template <> struct Block3Cfg<ElemF32, 32, 7>   { enum { WM = 1, WN = 8, NTW = 2 }; };
template <> struct Block3Cfg<ElemF32, 32, 11>  { enum { WM = 1, WN = 8, NTW = 2 }; };
template <> struct Block3Cfg<ElemF32, 64, 3>   { enum { WM = 2, WN = 4, NTW = 2 }; };
template <> struct Block3Cfg<ElemF32, 64, 7>   { enum { WM = 2, WN = 4, NTW = 2 }; };
template <> struct Block3Cfg<ElemF32, 64, 11>  { enum { WM = 2, WN = 4, NTW = 2 }; };
template <int C, int K> struct Block3Cfg<ElemF16X3, C, K> : Block3Cfg<ElemF32, C, K> {};
template <int C, int K> struct Block3Cfg<ElemF16A2, C, K> : Block3Cfg<ElemF32, C, K> {};

// Latency variants (see PairCfgNarrow): half the waves, same per-wave tile.
template <class ET, int C, int K> struct Block3CfgNarrow : Block3Cfg<ET, C, K> {};
template <> struct Block3CfgNarrow<ElemF16, 32, 11> { enum { WM = 1, WN = 4, NTW = 3 }; };
template <> struct Block3CfgNarrow<ElemF16, 64, 7>  { enum { WM = 2, WN = 2, NTW = 4 }; };
template <> struct Block3CfgNarrow<ElemF16, 64, 11> { enum { WM = 2, WN = 2, NTW = 4 }; };
template <> struct Block3CfgNarrow<ElemF16, 128, 3> { enum { WM = 4, WN = 1, NTW = 4 }; };
template <int C, int K> struct Block3CfgNarrow<ElemBF16, C, K> : Block3CfgNarrow<ElemF16, C, K> {};

template <class ET, int C, int K, class G>
static hipError_t launch_block3_cfg(const Block3Args& a0, hipStream_t stream) {
    if constexpr (G::WM == 0) {
        return hipErrorNotSupported;
    } else {
    constexpr int WM = G::WM, WN = G::WN, NTW = G::NTW;
    constexpr int NC = WN * NTW * 32;
    Block3Args a = a0;
    a.halo = 0;
    for (int i = 0; i < a.niter; ++i) a.halo += (a.dil[i] + 1) * ((K - 1) / 2);
    a.TL = NC - 2 * a.halo;
    if (a.TL < 32) return hipErrorNotSupported;
    a.ntiles = (a.L + a.TL - 1) / a.TL;
    constexpr int smem = block3_smem_bytes<ET, C, K, WM, WN, NTW>();
    // Skewed walk (no recompute at all) for grids that fill the chip several
    // times over: needs scratch, dilations <= 5 and H2 (d + 1) <= 30
    // (split f16, C = 32: the last stage of the 'checkpoint' operand mode is
    // MFMA-bound at three MFMAs per step, so what the skew removes - the 23 %
    // halo of the stand-alone whole-MRF tiling - shows: three skewed Block
    // launches 6.07 ms against 6.99 ms fused, profiles/r04/ab_x3_skew.txt)
    constexpr bool X3SKEW = pm_x3skew_id(ET::ID) &&
                            (C == 32 || (C == 64 && ET::ESZ == 4));
    if constexpr ((ET::ESZ == 2 || X3SKEW) && WM * WN == 8 && NTW >= 2) {
        typedef SkewGeom<ET, C, K, WM, WN, NTW> GE;
        static_assert(GE::SCRATCH <= PM_SKEW_WG_SCRATCH, "scratch bound");
        const int cus = pm_device_cus();
        const int forced = pm_force().walk_nseg;
        // Measured at batch 32 x 10 s (profiles/r03/ab_skew.txt): -14 % at C = 128
        // k 11 (against three pair launches), -10 % at C = 128 k 7, -7 % at
        // C = 64 k 11, -1.3 % at C = 64 k 7 (against the walked kernels); where
        // the walked halo is small (k 3: 12 columns) its carries and the
        // hand-over cost 5 % more than the recompute they save, on the
        // 128-column tiles of C = 256 k 7 it is even with three pair launches.
        constexpr bool WINS = ((C == 128 || C == 64) && K >= 7) || X3SKEW;
        bool fits = GE::SMEM <= 160 * 1024 && a.niter >= 1 && a.niter <= 3 &&
                    a.scratch && (cus > 0 || forced) &&
                    (pm_force().skew > 0 || (pm_force().skew == 0 && WINS));
        for (int i = 0; i < a.niter; ++i)
            fits = fits && a.dil[i] >= 1 && a.dil[i] <= 5 &&
                   ((K - 1) / 2) * (a.dil[i] + 1) <= 30;
        if (fits) {
            int nseg = forced ? forced : cus / a.B;
            if (nseg < 1) nseg = 1;
            const size_t need = (size_t)a.B * nseg * GE::SCRATCH;
            if ((forced || (a.L / NC) / nseg >= 4) && need <= a.scratch_bytes) {
                Block3SkewArgs p;
                if (a.act16 && a.act16_done) *a.act16_done = 1;
                a.act16_done = nullptr;     // (a host pointer)
                p.a = a; p.nseg = nseg; p.wg_scratch = GE::SCRATCH;
                p.scratch = a.scratch;
                auto skew = conv_block3_skew_kernel<ET, C, K, WM, WN, NTW>;
                hipError_t e = pm_ensure_dynamic_lds(
                    reinterpret_cast<const void*>(skew), GE::SMEM);
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL(skew, dim3(a.B * nseg), dim3(WM * WN * 64),
                                   GE::SMEM, stream, p);
                return hipGetLastError();
            }
        }
    }
    a.act16 = nullptr; a.act16_done = nullptr;   // (the skewed walk only)
    // Walked variant (no left-halo recompute) for grids that fill the chip
    // several times over, where its carry area (halo rows) fits the LDS
    if constexpr (ET::ESZ == 2 && WM * WN == 8 && !(C == 128 && K == 11) &&
                  !(C == 256 && K == 7)) {
        const int smem_walk = block3_walk_smem_bytes<ET, C, K, WM, WN, NTW>() +
                              block3_carry_bytes<ET, C>(a.halo);
        const int cus = pm_device_cus();
        const int forced = pm_force().walk_nseg;
        if (a.niter <= 3 && smem_walk <= 160 * 1024 && (cus > 0 || forced)) {
            int nseg = forced ? forced : cus / a.B;
            if (nseg < 1) nseg = 1;
            if (forced || (a.L / (NC - a.halo)) / nseg >= 6) {
                Block3WalkArgs p;
                p.a = a; p.nseg = nseg;
                auto walk = conv_block3_walk_kernel<ET, C, K, WM, WN, NTW>;
                hipError_t e = pm_ensure_dynamic_lds(
                    reinterpret_cast<const void*>(walk), smem_walk);
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL(walk, dim3(a.B * nseg), dim3(WM * WN * 64),
                                   smem_walk, stream, p);
                return hipGetLastError();
            }
        }
    }
    if constexpr ((C == 128 && K >= 7) || C == 256)
        return hipErrorNotSupported;  // walked only
    auto kern = conv_block3_kernel<ET, C, K, WM, WN, NTW>;
    hipError_t e = pm_ensure_dynamic_lds(
        reinterpret_cast<const void*>(kern), smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(a.ntiles * a.B), dim3(WM * WN * 64), smem,
                       stream, a);
    return hipGetLastError();
    }
}

template <class ET, int C, int K>
static hipError_t launch_block3_ck(const Block3Args& a, hipStream_t stream) {
    typedef Block3Cfg<ET, C, K> W;
    typedef Block3CfgNarrow<ET, C, K> N;
    if constexpr ((int)W::WM != 0 && (int)N::WN != (int)W::WN) {
        int halo = 0;
        for (int i = 0; i < a.niter; ++i) halo += (a.dil[i] + 1) * ((K - 1) / 2);
        const int TL = W::WN * W::NTW * 32 - 2 * halo;
        const int TLn = N::WN * N::NTW * 32 - 2 * halo;
        // (a forced walk - tests - keeps the 8-wave geometry the walks exist for)
        const bool allowed = pm_narrow_allowed() && !pm_force().walk_nseg;
        if (allowed && TL > 0 && TLn >= 32 &&
            (long long)((a.L + TL - 1) / TL) * a.B < PM_NARROW_BELOW)
            return launch_block3_cfg<ET, C, K, N>(a, stream);
    }
    return launch_block3_cfg<ET, C, K, W>(a, stream);
}

template <class ET, int C>
static hipError_t launch_block3_c(int K, const Block3Args& a, hipStream_t s) {
    switch (K) {
        case 3: return launch_block3_ck<ET, C, 3>(a, s);
        case 7: return launch_block3_ck<ET, C, 7>(a, s);
        case 11: return launch_block3_ck<ET, C, 11>(a, s);
    }
    return hipErrorNotSupported;
}

// Whole-MRF launch: kernel sizes (3, 7, 11), the k = 11 tiling for all three.
template <class ET, int C, class G>
static hipError_t launch_mrf_cfg(const Block3Args (&blocks)[3], hipStream_t stream) {
    if constexpr (G::WM == 0 || Block3Cfg<ET, C, 3>::WM == 0 ||
                  Block3Cfg<ET, C, 7>::WM == 0) {
        return hipErrorNotSupported;
    } else {
    constexpr int WM = G::WM, WN = G::WN, NTW = G::NTW;
    constexpr int NC = WN * NTW * 32;
    MrfArgs m;
    int halo = 0;
    constexpr int KS[3] = {3, 7, 11};
    for (int j = 0; j < 3; ++j) {
        m.k[j] = blocks[j];
        int h = 0;
        for (int i = 0; i < m.k[j].niter; ++i)
            h += (m.k[j].dil[i] + 1) * ((KS[j] - 1) / 2);
        halo = h > halo ? h : halo;
    }
    const int TL = NC - 2 * halo;
    if (TL < 32) return hipErrorNotSupported;
    // one x tile serves the three Blocks (generator.py MRF: every Block of a
    // stage reads the stage input)
    if (blocks[0].x != blocks[1].x || blocks[0].x != blocks[2].x ||
        blocks[0].L != blocks[1].L || blocks[0].L != blocks[2].L)
        return hipErrorNotSupported;
    for (int j = 0; j < 3; ++j) {
        m.k[j].halo = halo; m.k[j].TL = TL;
        m.k[j].ntiles = (m.k[j].L + TL - 1) / TL;
#ifdef PM_TUNING
        m.k[j].timeline = nullptr;
#endif
    }
    // Skewed whole-MRF walk (conv_mrf_skew_kernel): the operand layouts that
    // run this stage Block by Block on the skewed walk (pm_x3skew_id), when
    // the caller hands scratch over and the grid fills the chip several times
    if constexpr (PM_MRF_SKEW && pm_x3skew_id(ET::ID) && C == 32 &&
                  WM * WN == 8 && NTW >= 2) {
        typedef MrfSkewGeom<ET, C, WM, WN, NTW> MG;
        static_assert(MG::SCRATCH <= PM_SKEW_WG_SCRATCH, "scratch bound");
        const int cus = pm_device_cus();
        const int forced = pm_force().walk_nseg;
        const Block3Args& a = blocks[0];
        bool fits = MG::SMEM <= 160 * 1024 && a.scratch &&
                    (cus > 0 || forced) && pm_force().skew >= 0;
        for (int j = 0; j < 3; ++j) {
            fits = fits && blocks[j].niter == 3;
            for (int i = 0; i < 3 && fits; ++i)
                fits = blocks[j].dil[i] >= 1 && blocks[j].dil[i] <= 5 &&
                       ((KS[j] - 1) / 2) * (blocks[j].dil[i] + 1) <= 30;
        }
        if (fits) {
            int nseg = forced ? forced : cus / a.B;
            if (nseg < 1) nseg = 1;
            const size_t need = (size_t)a.B * nseg * MG::SCRATCH;
            if ((forced || (a.L / NC) / nseg >= 4) && need <= a.scratch_bytes) {
                MrfSkewArgs p = {};
                p.x = a.x; p.out = a.out;
                for (int j = 0; j < 3; ++j)
                    for (int n = 0; n < 3; ++n) {
                        p.w1[j][n] = blocks[j].w1[n];
                        p.w2[j][n] = blocks[j].w2[n];
                        p.dil[j][n] = blocks[j].dil[n];
                    }
                p.B = a.B; p.L = a.L; p.halo = halo; p.scale = a.scale;
                p.lengths = a.lengths; p.len_scale = a.len_scale;
                p.nseg = nseg; p.wg_scratch = MG::SCRATCH;
                p.scratch = a.scratch;
#ifdef PM_TUNING
                p.timeline = blocks[0].timeline;
#endif
                auto kern = conv_mrf_skew_kernel<ET, C, WM, WN, NTW>;
                hipError_t e = pm_ensure_dynamic_lds(
                    reinterpret_cast<const void*>(kern), MG::SMEM);
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL(kern, dim3(a.B * nseg), dim3(WM * WN * 64),
                                   MG::SMEM, stream, p);
                return hipGetLastError();
            }
        }
    }
    // (skew_only: the caller prefers Block-by-Block skewed launches to the
    // two-sided tiling below - except when the test hook has switched the
    // skewed walk off altogether: then this launch, whose sum order is the
    // skewed whole-MRF walk's, is the stand-alone counterpart the tests compare
    // bit for bit)
    if (blocks[0].skew_only && pm_force().skew >= 0)
        return hipErrorNotSupported;
    // Walked variant (no left-halo recompute): one workgroup per (utterance,
    // segment) with enough tiles per segment to amortise its two-sided first
    // tile; the sum-in-registers geometry (C = 32) only. (Split-f16 operands,
    // 4 bytes per element in LDS: the carry areas only fit beside 256-column
    // tiles of one tile per wave, and that walked variant measured 8.4 ms
    // against 7.0 ms for the stand-alone 512-column tiling - round 4, not kept.)
    if constexpr (C == 32 && ET::ESZ == 2 && WM * WN == 8) {
        const int cus = pm_device_cus();
        const int forced = pm_force().walk_nseg;
        if ((cus > 0 || forced) &&
            m.k[0].niter == 3 && m.k[1].niter == 3 && m.k[2].niter == 3) {
            const int B = m.k[0].B, L = m.k[0].L;
            const int step = NC - halo;
            int nseg = forced ? forced : cus / B;
            if (nseg < 1) nseg = 1;
            const int tiles_per_seg = (L / step) / nseg;
            if (forced || tiles_per_seg >= 8) {
                MrfWalkArgs wa = {};
                wa.x = m.k[0].x; wa.out = m.k[0].out;
                for (int j = 0; j < 3; ++j)
                    for (int n = 0; n < 3; ++n) {
                        wa.w1[j][n] = m.k[j].w1[n];
                        wa.w2[j][n] = m.k[j].w2[n];
                        wa.dil[j][n] = m.k[j].dil[n];
                    }
                wa.B = B; wa.L = L; wa.halo = halo; wa.scale = m.k[0].scale;
                wa.lengths = m.k[0].lengths; wa.len_scale = m.k[0].len_scale;
                wa.nseg = nseg;
                auto walk = conv_mrf_walk_kernel<ET, C, WM, WN, NTW>;
                const int smem_walk =
                    block3_walk_smem_bytes<ET, C, 11, WM, WN, NTW>() +
                    3 * block3_carry_bytes<ET, C>(halo);
                hipError_t e = pm_ensure_dynamic_lds(
                    reinterpret_cast<const void*>(walk), smem_walk);
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL(walk, dim3(B * nseg), dim3(WM * WN * 64),
                                   smem_walk, stream, wa);
                return hipGetLastError();
            }
        }
    }
    auto kern = conv_mrf_kernel<ET, C, WM, WN, NTW, C == 32>;
    constexpr int smem = block3_smem_bytes<ET, C, 11, WM, WN, NTW>();
    hipError_t e = pm_ensure_dynamic_lds(
        reinterpret_cast<const void*>(kern), smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(m.k[0].ntiles * m.k[0].B),
                       dim3(WM * WN * 64), smem, stream, m);
    return hipGetLastError();
    }
}

template <class ET, int C>
static hipError_t launch_mrf_c(const Block3Args (&blocks)[3], hipStream_t stream) {
    typedef Block3Cfg<ET, C, 11> W;
    typedef Block3CfgNarrow<ET, C, 11> N;
    if constexpr ((int)W::WM != 0 && (int)N::WN != (int)W::WN) {
        int halo = 0;
        for (int i = 0; i < blocks[2].niter; ++i) halo += (blocks[2].dil[i] + 1) * 5;
        const int TL = W::WN * W::NTW * 32 - 2 * halo;
        const int TLn = N::WN * N::NTW * 32 - 2 * halo;
        const bool allowed = pm_narrow_allowed() && !pm_force().walk_nseg;
        if (allowed && TL > 0 && TLn >= 32 &&
            (long long)((blocks[0].L + TL - 1) / TL) * blocks[0].B < PM_NARROW_BELOW)
            return launch_mrf_cfg<ET, C, N>(blocks, stream);
    }
    return launch_mrf_cfg<ET, C, W>(blocks, stream);
}

template <class ET>
hipError_t pm_launch_mrf(int C, const Block3Args (&blocks)[3], hipStream_t s) {
    for (int j = 0; j < 3; ++j) {
        if (blocks[j].niter < 1 || blocks[j].niter > 3)
            return hipErrorNotSupported;
        for (int i = 0; i < blocks[j].niter; ++i)
            if (blocks[j].dil[i] < 1 || blocks[j].dil[i] > 5)
                return hipErrorNotSupported;
    }
    // C = 64 has no registers left for the sum (trunk + accumulator are 128
    // VGPRs): its whole-MRF variant read-modify-writes `out` through L2, the
    // lines do not survive there (rocprof: 2.6 GB written per launch instead
    // of 0.9) and it only ties with three Block launches - not used.
    if (C == 32) return launch_mrf_c<ET, 32>(blocks, s);
    return hipErrorNotSupported;
}

template <class ET, int C>
static bool block3_supported_c(int K) {
    switch (K) {
        case 3: return Block3Cfg<ET, C, 3>::WM != 0;
        case 7: return Block3Cfg<ET, C, 7>::WM != 0;
        case 11: return Block3Cfg<ET, C, 11>::WM != 0;
    }
    return false;
}

template <class ET>
bool pm_block3_supported(int C, int K) {
    switch (C) {
        case 256: return block3_supported_c<ET, 256>(K);
        case 128: return block3_supported_c<ET, 128>(K);
        case 64: return block3_supported_c<ET, 64>(K);
        case 32: return block3_supported_c<ET, 32>(K);
    }
    return false;
}

template <class ET>
hipError_t pm_launch_block3(int C, int K, const Block3Args& a, hipStream_t s) {
    for (int i = 0; i < a.niter; ++i)
        if (a.dil[i] < 1 || a.dil[i] > 5) return hipErrorNotSupported;
    if (a.niter < 1 || a.niter > 3) return hipErrorNotSupported;
    switch (C) {
        case 256: return launch_block3_c<ET, 256>(K, a, s);
        case 128: return launch_block3_c<ET, 128>(K, a, s);
        case 64: return launch_block3_c<ET, 64>(K, a, s);
        case 32: return launch_block3_c<ET, 32>(K, a, s);
    }
    return hipErrorNotSupported;
}

// ---- single conv ----------------------------------------------------------
template <class ET, int KT, int KSPAN, int CH, int WM, int WN, int MTW, int NTW,
          int EPI = 0>
static hipError_t launch_single_cfg(const SingleArgs& a0, hipStream_t stream) {
    constexpr int N1 = WN * NTW * 32;
    constexpr int MB = WM * MTW * 32;
    SingleArgs a = a0;
    a.ntiles = (a.Lout + N1 - 1) / N1;
    a.nmblocks = a.M / MB;
    auto kern = conv_single_kernel<ET, KT, KSPAN, CH, WM, WN, MTW, NTW, EPI>;
    constexpr int smem = 2 * (N1 + KSPAN - 1) * (CH * ET::ESZ + 16);
    if (smem > 48 * 1024) {
        hipError_t e = pm_ensure_dynamic_lds(
            reinterpret_cast<const void*>(kern), smem);
        if (e != hipSuccess) return e;
    }
    const int grid = a.ntiles * a.nmblocks * a.B;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * WN * 64), smem, stream, a);
    return hipGetLastError();
}

template <class ET, int KT, int KSPAN>
static hipError_t launch_single_k(
    int ch, int cfg, const SingleArgs& a, hipStream_t s) {
    if (ch == 64) {
        switch (cfg) {
            case 0: return launch_single_cfg<ET, KT, KSPAN, 64, 4, 2, 2, 2>(a, s);
            case 1: return launch_single_cfg<ET, KT, KSPAN, 64, 2, 2, 1, 2>(a, s);
            case 2: return launch_single_cfg<ET, KT, KSPAN, 64, 1, 4, 1, 1>(a, s);
            // all 128 rows of a 2x upsampler in one workgroup: x is read once
            case 3: return launch_single_cfg<ET, KT, KSPAN, 64, 4, 2, 1, 2>(a, s);
            // the C_in = 128 r = 2 upsampler on long batches: 256 input columns
            // a workgroup instead of 128 - half as many workgroups, a weight
            // fragment feeds four MFMAs instead of two: 0.364 -> 0.342 ms (the
            // same widening of cfg 1 for C_in = 64 measured neutral and is not
            // kept: profiles/r05/ab_wide_upsampler.txt)
            case 5:
                if constexpr (KT == 2)
                    return launch_single_cfg<ET, KT, KSPAN, 64, 4, 2, 1, 4>(a, s);
                break;
        }
    } else if (ch == 32) {
        switch (cfg) {
            case 1: return launch_single_cfg<ET, KT, KSPAN, 32, 2, 2, 1, 2>(a, s);
            case 2: return launch_single_cfg<ET, KT, KSPAN, 32, 1, 4, 1, 1>(a, s);
        }
    }
    return hipErrorInvalidValue;
}

// cfg 4: the wide upsampler with the whole K staged once (ch = C_in = 256 /
// 512, 16-bit operands, weights packed as one chunk)
template <class ET, int CIN>
static hipError_t launch_upsample_cfg(const SingleArgs& a0, hipStream_t stream) {
    if constexpr (ET::ESZ != 2) {
        return hipErrorNotSupported;
    } else {
    constexpr int WM = 4, WN = 2, MTW = 2, NTW = 2;
    constexpr int N1 = WN * NTW * 32, MB = WM * MTW * 32;
    SingleArgs a = a0;
    if (a.Cin != CIN || a.M % MB) return hipErrorInvalidValue;
    a.ntiles = (a.Lout + N1 - 1) / N1;
    // M groups per column tile: as few as still give the chip ~2 workgroups
    // per CU (every group stages the x tile again)
    const int mblocks = a.M / MB;
    int groups = 1;
    while (groups < mblocks && (long long)a.ntiles * a.B * groups < 512)
        groups *= 2;
    if (pm_force().upsample_groups > 0) groups = pm_force().upsample_groups;
    if (groups > mblocks) groups = mblocks;
    while (mblocks % groups) groups /= 2;
    a.nmblocks = groups;
    auto kern = conv_upsample_kernel<ET, CIN, WM, WN, MTW, NTW>;
    constexpr int smem = (N1 + 2) * (CIN * ET::ESZ + 16);
    hipError_t e = pm_ensure_dynamic_lds(
        reinterpret_cast<const void*>(kern), smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(a.ntiles * a.nmblocks * a.B),
                       dim3(WM * WN * 64), smem, stream, a);
    return hipGetLastError();
    }
}

template <class ET>
hipError_t pm_launch_single(
    int kind, int ch, int cfg, const SingleArgs& a, hipStream_t s) {
    if (kind == 1 && cfg == 4) {
        if (ch == 256) return launch_upsample_cfg<ET, 256>(a, s);
        if (ch == 512) return launch_upsample_cfg<ET, 512>(a, s);
        return hipErrorInvalidValue;
    }
    if (kind == 0) return launch_single_k<ET, 7, 7>(ch, cfg, a, s);
    if (kind == 1) return launch_single_k<ET, 2, 3>(ch, cfg, a, s);
    return hipErrorInvalidValue;
}

// Framed DFT (STFT) as a 4-tap conv over the hop-reshaped padded audio:
// exact-fp32 MFMA only (the forward transforms run as FFTs, pm_fft.h; this is
// the backward and the brute-force cross-check). epi 1: magnitude, 3: the DFT
// cotangent grad / magnitude * (re, im) (backward, first half), 0: the
// overlap-add conv of that cotangent against the transposed basis (backward,
// second half: C_in = 1088 -> 256 samples of one hop, 4 taps).
#ifdef PM_INSTANTIATE_STFT
hipError_t pm_launch_stft(int epi, const SingleArgs& a, hipStream_t s) {
    if (epi == 1)
        return launch_single_cfg<ElemF32, 4, 4, 64, 2, 2, 1, 2, 1>(a, s);
    if (epi == 3)
        return launch_single_cfg<ElemF32, 4, 4, 64, 2, 2, 1, 2, 3>(a, s);
    if (epi == 0)
        return launch_single_cfg<ElemF32, 4, 4, 64, 2, 2, 1, 2, 0>(a, s);
    return hipErrorInvalidValue;
}
#endif

#endif  // PM_INSTANTIATE
