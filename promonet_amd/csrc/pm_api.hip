// C ABI of libpromonet_hip.so (see include/promonet_hip.h) and the HiFi-GAN
// engine behind it: weight folding / packing at load, workspace planning and
// the per-forward launch sequence. Host C++; every kernel it launches is
// hand-written HIP for gfx950 (pm_conv.h, pm_misc.h, pm_stft.h, pm_fft.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/promonet_hip.h"
#include "pm_launch.h"
#include "pm_misc.h"
#include "pm_stft.h"
#include "pm_fft.h"
#include "pm_fargan.h"

#define PM_VERSION 100

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local char g_error[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                        \
    do {                                                                     \
        hipError_t e_ = (expr);                                              \
        if (e_ != hipSuccess)                                                \
            return fail(PM_EHIP, "%s failed: %s (%s:%d)", #expr,             \
                        hipGetErrorString(e_), __FILE__, __LINE__);          \
    } while (0)

#ifdef PM_TUNING
static unsigned long long* g_timeline = nullptr;   // pm_debug_timeline
#endif

static inline int pad32(int c) { return (c + 31) / 32 * 32; }
static inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }
// bytes per element of an LDS operand row / of a packed weight stream
static inline int esz(int dtype) {
    return dtype == PM_F32 || dtype == PM_F16X3 || dtype == PM_F16A2 ? 4 : 2;
}
static inline int wsz(int dtype) {
    return dtype == PM_F32 || dtype == PM_F16X3 ? 4 : 2;
}
// A stage's (or the engine's) operand type code -> the type of its Blocks and
// of its upsampler (promonet_hip.h: PM_F16A2, PM_F16UX)
static inline int block_dtype(int code) { return code == PM_F16UX ? PM_F16 : code; }
// (the PM_F16UX upsampler as PM_F16A2 instead: same step, 1 % more error)
static inline int up_dtype(int code) {
    return code == PM_F16A2 || code == PM_F16UX ? PM_F16X3 : code;
}

// ---------------------------------------------------------------------------
// dtype dispatch
// ---------------------------------------------------------------------------
static hipError_t launch_pair(
    int dtype, int C, int K, const PairArgs& a0, hipStream_t s) {
    PairArgs a = a0;
#ifdef PM_TUNING
    a.timeline = g_timeline;
#endif
    switch (dtype) {
        case PM_F32: return pm_launch_pair<ElemF32>(C, K, a, s);
        case PM_F16: return pm_launch_pair<ElemF16>(C, K, a, s);
        case PM_BF16: return pm_launch_pair<ElemBF16>(C, K, a, s);
        case PM_F16X3: return pm_launch_pair<ElemF16X3>(C, K, a, s);
        case PM_F16A2: return pm_launch_pair<ElemF16A2>(C, K, a, s);
    }
    return hipErrorInvalidValue;
}

static hipError_t launch_block3(
    int dtype, int C, int K, const Block3Args& a0, hipStream_t s) {
    Block3Args a = a0;
#ifdef PM_TUNING
    a.timeline = g_timeline;
#endif
    switch (dtype) {
        case PM_F32: return pm_launch_block3<ElemF32>(C, K, a, s);
        case PM_F16: return pm_launch_block3<ElemF16>(C, K, a, s);
        case PM_BF16: return pm_launch_block3<ElemBF16>(C, K, a, s);
        case PM_F16X3: return pm_launch_block3<ElemF16X3>(C, K, a, s);
        case PM_F16A2: return pm_launch_block3<ElemF16A2>(C, K, a, s);
    }
    return hipErrorInvalidValue;
}

static hipError_t launch_mrf(
    int dtype, int C, const Block3Args (&a0)[3], hipStream_t s) {
    Block3Args a[3] = {a0[0], a0[1], a0[2]};
#ifdef PM_TUNING
    a[0].timeline = g_timeline;      // (the skewed whole-MRF walk's phase totals)
#endif
    switch (dtype) {
        case PM_F32: return pm_launch_mrf<ElemF32>(C, a, s);
        case PM_F16: return pm_launch_mrf<ElemF16>(C, a, s);
        case PM_BF16: return pm_launch_mrf<ElemBF16>(C, a, s);
        case PM_F16X3: return pm_launch_mrf<ElemF16X3>(C, a, s);
        case PM_F16A2: return pm_launch_mrf<ElemF16A2>(C, a, s);
    }
    return hipErrorInvalidValue;
}

// Fusion level: 2 = whole-MRF launches where they exist, 1 = one kernel per
// Block, 0 = one kernel per Block iteration. The shipped library always runs
// level 2; a -DPM_TUNING build reads PM_FUSION=pair|block for A/B runs.
static int fusion_level() {
#ifdef PM_TUNING
    static const int level = [] {
        const char* e = getenv("PM_FUSION");
        return (e && !strcmp(e, "pair")) ? 0 : (e && !strcmp(e, "block")) ? 1 : 2;
    }();
    return level;
#else
    return 2;
#endif
}
static bool block3_enabled() {
    return fusion_level() >= 1;
}

static int pair_chunk(int dtype, int C) {
    switch (dtype) {
        case PM_F32: return pm_pair_chunk<ElemF32>(C);
        case PM_F16: return pm_pair_chunk<ElemF16>(C);
        case PM_BF16: return pm_pair_chunk<ElemBF16>(C);
        case PM_F16X3: return pm_pair_chunk<ElemF16X3>(C);
        case PM_F16A2: return pm_pair_chunk<ElemF16A2>(C);
    }
    return 0;
}

static bool block3_supported(int dtype, int C, int K) {
    if (!block3_enabled()) return false;
    switch (dtype) {
        case PM_F32: return pm_block3_supported<ElemF32>(C, K);
        case PM_F16: return pm_block3_supported<ElemF16>(C, K);
        case PM_BF16: return pm_block3_supported<ElemBF16>(C, K);
        case PM_F16X3: return pm_block3_supported<ElemF16X3>(C, K);
        case PM_F16A2: return pm_block3_supported<ElemF16A2>(C, K);
    }
    return false;
}

// Chunk size the MRF conv weights of a (C, K) layer are packed with: the
// whole-Block kernel streams 64-channel chunks, the pair kernel its own CH.
static int mrf_chunk(int dtype, int C, int K, int ndil) {
    if (ndil <= 3 && block3_supported(dtype, C, K)) return C < 64 ? C : 64;
    return pair_chunk(dtype, C);
}

static hipError_t launch_single(
    int dtype, int kind, int ch, int cfg, const SingleArgs& a, hipStream_t s) {
    switch (dtype) {
        case PM_F32: return pm_launch_single<ElemF32>(kind, ch, cfg, a, s);
        case PM_F16: return pm_launch_single<ElemF16>(kind, ch, cfg, a, s);
        case PM_BF16: return pm_launch_single<ElemBF16>(kind, ch, cfg, a, s);
        case PM_F16X3: return pm_launch_single<ElemF16X3>(kind, ch, cfg, a, s);
        case PM_F16A2: return pm_launch_single<ElemF16A2>(kind, ch, cfg, a, s);
    }
    return hipErrorInvalidValue;
}

static hipError_t launch_pack(int dtype, const PackArgs& a, hipStream_t s) {
    const unsigned grid = (unsigned)((a.total + 255) / 256);
    switch (dtype) {
        case PM_F32:
            hipLaunchKernelGGL(pm_pack_kernel<ElemF32>, dim3(grid), dim3(256), 0, s, a);
            break;
        case PM_F16:
            hipLaunchKernelGGL(pm_pack_kernel<ElemF16>, dim3(grid), dim3(256), 0, s, a);
            break;
        case PM_BF16:
            hipLaunchKernelGGL(pm_pack_kernel<ElemBF16>, dim3(grid), dim3(256), 0, s, a);
            break;
        case PM_F16X3:
            hipLaunchKernelGGL(pm_pack_kernel<ElemF16X3>, dim3(grid), dim3(256), 0, s, a);
            break;
        case PM_F16A2:
            hipLaunchKernelGGL(pm_pack_kernel<ElemF16A2>, dim3(grid), dim3(256), 0, s, a);
            break;
        default:
            return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// Geometry of one packed convolution
struct ConvGeom {
    int mode = 0;            // 0 conv, 1 conv-transpose
    int cout = 0, cin = 0, k = 0;
    int cout_pad = 0, cin_pad = 0;
    int M = 0;               // packed rows (cout_pad, or r * cout_pad)
    int ch = 64, kt = 0;
    int r = 0, p = 0;
    bool bias_step = false;  // MRF convs: the stream of every M tile ends
                             // with one bias step (pm_pack_bias_step_kernel)
    size_t weight_elems() const { return (size_t)M * cin_pad * kt; }
    size_t packed_elems() const {
        return weight_elems() + (bias_step ? (size_t)(M / 32) * 512 : 0);
    }
};

static hipError_t pack_weights(
    int dtype, const ConvGeom& g, const float* w, void* out, hipStream_t s) {
    PackArgs a;
    a.w = w; a.out = out; a.mode = g.mode;
    a.cout = g.cout; a.cin = g.cin; a.k = g.k;
    a.cout_pad = g.cout_pad; a.cin_pad = g.cin_pad;
    a.mtiles = g.M / 32; a.nch = g.cin_pad / g.ch; a.ch = g.ch; a.kt = g.kt;
    a.r = g.r; a.p = g.p;
    a.bias_step = g.bias_step ? 1 : 0;
    a.total = (long long)g.weight_elems();
    return launch_pack(dtype, a, s);
}

// Write the bias step of every M tile of a packed MRF conv stream
static hipError_t pack_bias_step(
    int dtype, const ConvGeom& g, const float* bias, void* out, hipStream_t s) {
    const int mtiles = g.M / 32;
    const long long per_mt = (long long)g.weight_elems() / mtiles;
    const dim3 grid((mtiles * 512 + 255) / 256), block(256);
    switch (dtype) {
        case PM_F32:
            hipLaunchKernelGGL(pm_pack_bias_step_kernel<ElemF32>, grid, block,
                               0, s, bias, out, g.cout, mtiles, per_mt);
            break;
        case PM_F16:
            hipLaunchKernelGGL(pm_pack_bias_step_kernel<ElemF16>, grid, block,
                               0, s, bias, out, g.cout, mtiles, per_mt);
            break;
        case PM_BF16:
            hipLaunchKernelGGL(pm_pack_bias_step_kernel<ElemBF16>, grid, block,
                               0, s, bias, out, g.cout, mtiles, per_mt);
            break;
        case PM_F16X3:
            hipLaunchKernelGGL(pm_pack_bias_step_kernel<ElemF16X3>, grid, block,
                               0, s, bias, out, g.cout, mtiles, per_mt);
            break;
        case PM_F16A2:
            hipLaunchKernelGGL(pm_pack_bias_step_kernel<ElemF16A2>, grid, block,
                               0, s, bias, out, g.cout, mtiles, per_mt);
            break;
        default:
            return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

static hipError_t pad_bias(
    const float* src, float* dst, int n, int n_pad, int rep, hipStream_t s) {
    const int total = n_pad * rep;
    hipLaunchKernelGGL(pm_pad_bias_kernel, dim3((total + 255) / 256),
                       dim3(256), 0, s, src, dst, n, n_pad, rep);
    return hipGetLastError();
}

// Wide upsampler (conv_upsample_kernel): 16-bit operands, C_in 256 / 512,
// 256-row M blocks whose tap window is uniform. Sets the single-chunk packing.
static bool upsample_whole_k(int dtype, ConvGeom& g) {
    if (esz(dtype) == 4 || g.mode != 1) return false;
    if (g.cin_pad != 256 && g.cin_pad != 512) return false;
    if (g.M % 256 || ((g.r / 2) * g.cout_pad) % 64) return false;
    g.ch = g.cin_pad;
    return true;
}

static int single_cfg(int M, int ch, int wave64_ok) {
    if (M % 256 == 0 && ch == 64 && wave64_ok) return 0;
    if (M == 128 && ch == 64) return 3;   // whole M in one workgroup
    if (M % 64 == 0) return 1;
    return 2;
}

// ---------------------------------------------------------------------------
// engine
// ---------------------------------------------------------------------------
struct Layer {
    ConvGeom geom;
    void* w = nullptr;        // packed (+ bias steps for MRF convs)
    float* bias = nullptr;    // padded (tiled per phase for conv-transpose)
    float* tmp_g = nullptr;   // weight-norm pair awaiting its partner
    float* tmp_v = nullptr;
    bool has_w = false, has_b = false;
    int cfg = 0;
    int dtype = PM_F16;       // MFMA operand type this layer is packed for
};

struct Stage {
    int cin = 0, cout = 0, cin_pad = 0, cout_pad = 0, r = 0, k = 0;
    int dtype = PM_F16;       // operand type of the stage (upsampler + MRF)
    Layer up;
    Layer c1[PM_MAX_RESBLOCKS][PM_MAX_DILATIONS];
    Layer c2[PM_MAX_RESBLOCKS][PM_MAX_DILATIONS];
};

struct pm_hifigan_s {
    pm_hifigan_config cfg;
    int dtype = PM_F16;
    int cfp = 0, c0 = 0, c0p = 0, hop = 1;
    Layer in_conv;
    float* spk_w = nullptr; float* spk_b = nullptr;
    bool has_spk_w = false, has_spk_b = false;
    float* out_w = nullptr; bool has_out_w = false;
    int c_last = 0, c_last_pad = 0;
    std::vector<Stage> stages;
    bool finalized = false;
    // ---- optional per-launch timing with HIP events (bench.py roofline) ----
    bool profile = false;
    std::string profile_only;            // bracket this label's launches only
    std::vector<hipEvent_t> events;      // pool, grown on demand
    size_t events_used = 0;
    struct Mark { std::string label; double flops, bytes; size_t e0, e1; };
    std::vector<Mark> marks;             // launches since the last collect
    struct Acc { long long count = 0; double ms = 0, flops = 0, bytes = 0; };
    std::map<std::string, Acc> totals;
    std::string report;
};

static int prof_event(pm_hifigan_t h, hipStream_t s, size_t* index) {
    if (h->events_used == h->events.size()) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        h->events.push_back(e);
    }
    *index = h->events_used++;
    HIP_TRY(hipEventRecord(h->events[*index], s));
    return PM_OK;
}

// Bracket the launch(es) issued by `body` with two events on the same stream
#define PROF(h, s, label_, flops_, bytes_, body)                             \
    do {                                                                     \
        size_t e0_ = 0, e1_ = 0;                                             \
        const bool on_ = (h)->profile &&                                     \
            ((h)->profile_only.empty() || (h)->profile_only == (label_));    \
        if (on_) { int r_ = prof_event(h, s, &e0_); if (r_) return r_; }     \
        body;                                                                \
        if (on_) {                                                           \
            int r_ = prof_event(h, s, &e1_); if (r_) return r_;              \
            (h)->marks.push_back({label_, (double)(flops_), (double)(bytes_), e0_, e1_}); \
        }                                                                    \
    } while (0)

static void free_layer(Layer& l) {
    if (l.w) hipFree(l.w);
    if (l.bias) hipFree(l.bias);
    if (l.tmp_g) hipFree(l.tmp_g);
    if (l.tmp_v) hipFree(l.tmp_v);
    l = Layer();
}

extern "C" int pm_version(void) { return PM_VERSION; }
extern "C" const char* pm_last_error(void) { return g_error; }

extern "C" int pm_hifigan_create(
    const pm_hifigan_config* c, pm_hifigan_t* out) {
    if (!c || !out) return fail(PM_EINVAL, "null argument");
    if (c->compute_dtype < 0 || c->compute_dtype > PM_F16UX)
        return fail(PM_EINVAL, "compute_dtype %d unknown", c->compute_dtype);
    if (c->num_stages < 1 || c->num_stages > PM_MAX_STAGES ||
        c->num_resblocks < 1 || c->num_resblocks > PM_MAX_RESBLOCKS ||
        c->num_dilations < 1 || c->num_dilations > PM_MAX_DILATIONS)
        return fail(PM_EINVAL, "stage / resblock / dilation count out of range");
    if (c->num_features < 1 || c->global_channels < 1 ||
        c->initial_channels < (1 << c->num_stages))
        return fail(PM_EINVAL, "bad channel configuration");
    for (int j = 0; j < c->num_resblocks; ++j) {
        const int k = c->resblock_kernel_sizes[j];
        if (k != 3 && k != 7 && k != 11)
            return fail(PM_EINVAL,
                        "resblock kernel size %d unsupported (3, 7, 11)", k);
        for (int n = 0; n < c->num_dilations; ++n) {
            const int d = c->resblock_dilations[j][n];
            if (d < 1 || d > 5)
                return fail(PM_EINVAL, "dilation %d unsupported (1..5)", d);
        }
    }
    for (int i = 0; i < c->num_stages; ++i)
        if (c->stage_compute_dtype[i] < 0 ||
            c->stage_compute_dtype[i] > 1 + PM_F16UX)
            return fail(PM_EINVAL, "stage_compute_dtype[%d] = %d unknown", i,
                        c->stage_compute_dtype[i]);
    auto* h = new pm_hifigan_s();
    h->cfg = *c;
    h->dtype = c->compute_dtype;
    h->in_conv.dtype = block_dtype(h->dtype);
    h->cfp = pad32(c->num_features);
    h->c0 = c->initial_channels;
    h->c0p = pad32(h->c0);
    h->hop = 1;
    h->stages.resize(c->num_stages);
    for (int i = 0; i < c->num_stages; ++i) {
        Stage& s = h->stages[i];
        s.r = c->upsample_rates[i];
        s.k = c->upsample_kernel_sizes[i];
        const int code = c->stage_compute_dtype[i]
            ? c->stage_compute_dtype[i] - 1 : h->dtype;
        s.dtype = block_dtype(code);
        s.up.dtype = up_dtype(code);
        if (s.r < 2 || (s.r & 1) || s.k != 2 * s.r) {
            delete h;
            return fail(PM_EINVAL,
                        "upsample stage %d: rate %d kernel %d unsupported "
                        "(need even rate and kernel == 2 * rate)", i,
                        c->upsample_rates[i], c->upsample_kernel_sizes[i]);
        }
        s.cin = h->c0 >> i;
        s.cout = h->c0 >> (i + 1);
        s.cin_pad = pad32(s.cin);
        s.cout_pad = pad32(s.cout);
        if (s.cout_pad != 32 && s.cout_pad != 64 && s.cout_pad != 128 &&
            s.cout_pad != 256) {
            delete h;
            return fail(PM_EINVAL,
                        "stage %d: %d channels unsupported (padded channel "
                        "count must be 32, 64, 128 or 256)", i, s.cout);
        }
        h->hop *= s.r;
        // conv-transpose as a polyphase GEMM
        ConvGeom& g = s.up.geom;
        g.mode = 1; g.cout = s.cout; g.cin = s.cin; g.k = s.k;
        g.cout_pad = s.cout_pad; g.cin_pad = s.cin_pad;
        g.M = s.r * s.cout_pad; g.kt = 2; g.r = s.r; g.p = s.r / 2;
        g.ch = (s.cin_pad % 64 == 0) ? 64 : 32;
        s.up.cfg = upsample_whole_k(s.up.dtype, g)
            ? 4
            : single_cfg(g.M, g.ch, ((s.r / 2) * s.cout_pad) % 64 == 0);
        for (int j = 0; j < c->num_resblocks; ++j)
            for (int n = 0; n < c->num_dilations; ++n)
                for (int which = 0; which < 2; ++which) {
                    (which ? s.c2 : s.c1)[j][n].dtype = s.dtype;
                    ConvGeom& q = (which ? s.c2 : s.c1)[j][n].geom;
                    q.mode = 0; q.cout = q.cin = s.cout;
                    q.k = c->resblock_kernel_sizes[j];
                    q.cout_pad = q.cin_pad = q.M = s.cout_pad;
                    q.kt = q.k;
                    q.ch = mrf_chunk(s.dtype, s.cout_pad, q.k, c->num_dilations);
                    q.bias_step = true;
                }
    }
    {
        ConvGeom& g = h->in_conv.geom;
        g.mode = 0; g.cout = h->c0; g.cin = c->num_features; g.k = 7;
        g.cout_pad = g.M = h->c0p; g.cin_pad = h->cfp; g.kt = 7;
        g.ch = (h->cfp % 64 == 0) ? 64 : 32;
        h->in_conv.cfg = single_cfg(g.M, g.ch, 1);
    }
    h->c_last = h->stages.back().cout;
    h->c_last_pad = h->stages.back().cout_pad;
    if (h->c_last_pad > 64) {
        delete h;
        return fail(PM_EINVAL, "output conv supports <= 64 input channels");
    }
    *out = h;
    return PM_OK;
}

extern "C" int pm_hifigan_destroy(pm_hifigan_t h) {
    if (!h) return PM_OK;
    free_layer(h->in_conv);
    if (h->spk_w) hipFree(h->spk_w);
    if (h->spk_b) hipFree(h->spk_b);
    if (h->out_w) hipFree(h->out_w);
    for (auto e : h->events) hipEventDestroy(e);
    for (auto& s : h->stages) {
        free_layer(s.up);
        for (int j = 0; j < PM_MAX_RESBLOCKS; ++j)
            for (int n = 0; n < PM_MAX_DILATIONS; ++n) {
                free_layer(s.c1[j][n]);
                free_layer(s.c2[j][n]);
            }
    }
    delete h;
    return PM_OK;
}

static size_t numel(const int64_t* shape, int ndim) {
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
    return n;
}

static int copy_dev(float** dst, const float* src, size_t n, hipStream_t s) {
    if (!*dst) HIP_TRY(hipMalloc((void**)dst, n * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(*dst, src, n * sizeof(float),
                           hipMemcpyDeviceToDevice, s));
    return PM_OK;
}

// Pack a folded torch-layout weight into the layer
static int set_weight(
    pm_hifigan_t h, Layer& l, const float* w, const int64_t* shape, int ndim,
    hipStream_t s, const char* name) {
    const ConvGeom& g = l.geom;
    const int64_t d0 = g.mode == 0 ? g.cout : g.cin;
    const int64_t d1 = g.mode == 0 ? g.cin : g.cout;
    if (ndim != 3 || shape[0] != d0 || shape[1] != d1 || shape[2] != g.k)
        return fail(PM_EINVAL, "%s: expected shape (%lld, %lld, %d)", name,
                    (long long)d0, (long long)d1, g.k);
    const size_t bytes = g.packed_elems() * wsz(l.dtype);
    if (!l.w) HIP_TRY(hipMalloc(&l.w, bytes));
    HIP_TRY(pack_weights(l.dtype, g, w, l.w, s));
    l.has_w = true;
    if (g.bias_step && l.has_b)
        HIP_TRY(pack_bias_step(l.dtype, g, l.bias, l.w, s));
    return PM_OK;
}

static int set_bias(
    pm_hifigan_t h, Layer& l, const float* b, const int64_t* shape, int ndim,
    hipStream_t s, const char* name) {
    const ConvGeom& g = l.geom;
    if (ndim != 1 || shape[0] != g.cout)
        return fail(PM_EINVAL, "%s: expected shape (%d)", name, g.cout);
    const int rep = g.mode == 1 ? g.r : 1;
    if (!l.bias)
        HIP_TRY(hipMalloc((void**)&l.bias,
                          (size_t)g.cout_pad * rep * sizeof(float)));
    HIP_TRY(pad_bias(b, l.bias, g.cout, g.cout_pad, rep, s));
    l.has_b = true;
    if (g.bias_step && l.has_w)
        HIP_TRY(pack_bias_step(l.dtype, g, l.bias, l.w, s));
    return PM_OK;
}

static int set_norm_part(
    pm_hifigan_t h, Layer& l, bool is_g, const float* t, const int64_t* shape,
    int ndim, hipStream_t s, const char* name) {
    const ConvGeom& g = l.geom;
    const int64_t rows = g.mode == 0 ? g.cout : g.cin;
    const int64_t cols = (g.mode == 0 ? g.cin : g.cout) * (int64_t)g.k;
    if (is_g) {
        if (ndim != 3 || shape[0] != rows || shape[1] != 1 || shape[2] != 1)
            return fail(PM_EINVAL, "%s: expected shape (%lld, 1, 1)", name,
                        (long long)rows);
        int rc = copy_dev(&l.tmp_g, t, rows, s);
        if (rc) return rc;
    } else {
        if (ndim != 3 || (int64_t)numel(shape, ndim) != rows * cols ||
            shape[0] != rows)
            return fail(PM_EINVAL, "%s: unexpected weight_v shape", name);
        int rc = copy_dev(&l.tmp_v, t, rows * cols, s);
        if (rc) return rc;
    }
    if (l.tmp_g && l.tmp_v) {
        float* folded = nullptr;
        HIP_TRY(hipMalloc((void**)&folded, rows * cols * sizeof(float)));
        hipLaunchKernelGGL(pm_fold_kernel, dim3((unsigned)rows), dim3(256), 0,
                           s, l.tmp_g, l.tmp_v, folded, (int)cols);
        HIP_TRY(hipGetLastError());
        const int64_t wshape[3] = {
            rows, g.mode == 0 ? (int64_t)g.cin : (int64_t)g.cout, g.k};
        int rc = set_weight(h, l, folded, wshape, 3, s, name);
        HIP_TRY(hipStreamSynchronize(s));
        hipFree(folded);
        hipFree(l.tmp_g); hipFree(l.tmp_v);
        l.tmp_g = l.tmp_v = nullptr;
        if (rc) return rc;
    }
    return PM_OK;
}

static int set_layer_tensor(
    pm_hifigan_t h, Layer& l, const char* leaf, const float* t,
    const int64_t* shape, int ndim, hipStream_t s, const char* name) {
    if (!strcmp(leaf, "weight")) return set_weight(h, l, t, shape, ndim, s, name);
    if (!strcmp(leaf, "bias")) return set_bias(h, l, t, shape, ndim, s, name);
    if (!strcmp(leaf, "weight_g"))
        return set_norm_part(h, l, true, t, shape, ndim, s, name);
    if (!strcmp(leaf, "weight_v"))
        return set_norm_part(h, l, false, t, shape, ndim, s, name);
    return fail(PM_EINVAL, "%s: unknown tensor", name);
}

extern "C" int pm_hifigan_load_tensor(
    pm_hifigan_t h, const char* name, const float* dev, const int64_t* shape,
    int ndim, void* stream) {
    if (!h || !name || !dev || !shape) return fail(PM_EINVAL, "null argument");
    hipStream_t s = (hipStream_t)stream;
    const int ns = (int)h->stages.size();
    int rc = PM_EINVAL;
    int i = 0, j = 0, n = 0, which = 0;
    char leaf[32] = "";
    if (!strncmp(name, "input_feature_conv.", 19)) {
        rc = set_layer_tensor(h, h->in_conv, name + 19, dev, shape, ndim, s, name);
    } else if (!strcmp(name, "input_speaker_conv.weight")) {
        if (ndim != 3 || shape[0] != h->c0 ||
            shape[1] != h->cfg.global_channels || shape[2] != 1)
            return fail(PM_EINVAL, "%s: unexpected shape", name);
        rc = copy_dev(&h->spk_w, dev, numel(shape, ndim), s);
        h->has_spk_w = rc == PM_OK;
    } else if (!strcmp(name, "input_speaker_conv.bias")) {
        if (ndim != 1 || shape[0] != h->c0)
            return fail(PM_EINVAL, "%s: unexpected shape", name);
        rc = copy_dev(&h->spk_b, dev, numel(shape, ndim), s);
        h->has_spk_b = rc == PM_OK;
    } else if (sscanf(name, "model.%d.model.2.model.%d.convs%d.%d.%31s", &i,
                      &j, &which, &n, leaf) == 5) {
        if (i < 0 || i >= ns || j < 0 || j >= h->cfg.num_resblocks || n < 0 ||
            n >= h->cfg.num_dilations || which < 1 || which > 2)
            return fail(PM_EINVAL, "%s: index out of range", name);
        Layer& l = (which == 1 ? h->stages[i].c1 : h->stages[i].c2)[j][n];
        rc = set_layer_tensor(h, l, leaf, dev, shape, ndim, s, name);
    } else if (sscanf(name, "model.%d.model.1.%31s", &i, leaf) == 2) {
        if (i < 0 || i >= ns)
            return fail(PM_EINVAL, "%s: stage out of range", name);
        rc = set_layer_tensor(h, h->stages[i].up, leaf, dev, shape, ndim, s, name);
    } else if (sscanf(name, "model.%d.%31s", &i, leaf) == 2 && i == ns + 1 &&
               !strcmp(leaf, "weight")) {
        if (ndim != 3 || shape[0] != 1 || shape[1] != h->c_last ||
            shape[2] != 7)
            return fail(PM_EINVAL, "%s: expected shape (1, %d, 7)", name,
                        h->c_last);
        rc = copy_dev(&h->out_w, dev, numel(shape, ndim), s);
        h->has_out_w = rc == PM_OK;
    } else {
        return fail(PM_EINVAL, "%s: not a HiFiGAN state-dict key", name);
    }
    if (rc != PM_OK) return rc;
    HIP_TRY(hipStreamSynchronize(s));
    h->finalized = false;
    return PM_OK;
}

extern "C" int pm_hifigan_finalize(pm_hifigan_t h, void* stream) {
    if (!h) return fail(PM_EINVAL, "null handle");
    auto check = [](const Layer& l, const std::string& what) -> int {
        if (!l.has_w) return fail(PM_ESTATE, "missing tensor: %s weight", what.c_str());
        if (!l.has_b) return fail(PM_ESTATE, "missing tensor: %s bias", what.c_str());
        return PM_OK;
    };
    int rc;
    if ((rc = check(h->in_conv, "input_feature_conv"))) return rc;
    if (!h->has_spk_w || !h->has_spk_b)
        return fail(PM_ESTATE, "missing tensor: input_speaker_conv");
    if (!h->has_out_w) return fail(PM_ESTATE, "missing tensor: output conv");
    for (size_t i = 0; i < h->stages.size(); ++i) {
        Stage& s = h->stages[i];
        if ((rc = check(s.up, "model." + std::to_string(i) + ".model.1")))
            return rc;
        for (int j = 0; j < h->cfg.num_resblocks; ++j)
            for (int n = 0; n < h->cfg.num_dilations; ++n) {
                const std::string base = "model." + std::to_string(i) +
                    ".model.2.model." + std::to_string(j);
                if ((rc = check(s.c1[j][n], base + ".convs1." + std::to_string(n))))
                    return rc;
                if ((rc = check(s.c2[j][n], base + ".convs2." + std::to_string(n))))
                    return rc;
            }
    }
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    h->finalized = true;
    return PM_OK;
}

extern "C" int pm_hifigan_hopsize(pm_hifigan_t h) { return h ? h->hop : 0; }
extern "C" int pm_hifigan_features_cl_channels(pm_hifigan_t h) {
    return h ? h->cfp : 0;
}

struct Plan {
    size_t off_feat, off_gbias, off_buf, buf_elems, off_scratch, scratch, total;
};

// Scratch of the skewed whole-Block walk: one workgroup per (utterance,
// segment), at most max(B, CUs) of them (pm_launch.h)
static size_t walk_scratch_bytes(int B) {
    const int cus = pm_device_cus();
    return (size_t)std::max(B, cus > 0 ? cus : 256) * PM_SKEW_WG_SCRATCH;
}

static Plan make_plan(pm_hifigan_t h, int B, int T) {
    Plan p;
    size_t mx = (size_t)T * h->c0p;
    size_t L = T;
    for (auto& s : h->stages) {
        L *= s.r;
        mx = std::max(mx, L * (size_t)s.cout_pad);
    }
    p.buf_elems = align256((size_t)B * mx * sizeof(float)) / sizeof(float);
    p.off_feat = 0;
    p.off_gbias = align256((size_t)B * T * h->cfp * sizeof(float));
    p.off_buf = p.off_gbias + align256((size_t)B * h->c0p * sizeof(float));
    p.off_scratch = p.off_buf + 4 * p.buf_elems * sizeof(float);
    // (the skewed walk is only taken with >= 4 steps of >= 256 columns per
    // segment, max(1, CUs / B) segments per utterance: short calls - a
    // streaming frame, a single 2 s utterance - carry no scratch)
    {
        const int cus = pm_device_cus();
        const size_t nseg = std::max(1, (cus > 0 ? cus : 256) / B);
        p.scratch = pm_force().walk_nseg || pm_force().skew > 0 ||
                    L / 256 / nseg >= 4
            ? walk_scratch_bytes(B) : 0;   // (a forced walk / skew: tests)
    }
    p.total = p.off_scratch + p.scratch;
    return p;
}

extern "C" size_t pm_hifigan_workspace_bytes(pm_hifigan_t h, int B, int T) {
    if (!h || B < 1 || T < 1) return 0;
    return make_plan(h, B, T).total;
}

static int forward_impl(
    pm_hifigan_t h, const float* features, bool features_cl, const float* g,
    int gbatch, float* out, int B, int T, void* ws, size_t ws_bytes,
    hipStream_t s, const int* lengths = nullptr) {
    if (!h || !features || !g || !out || !ws)
        return fail(PM_EINVAL, "null argument");
    if (!h->finalized) return fail(PM_ESTATE, "pm_hifigan_finalize not called");
    if (B < 1 || T < 1) return fail(PM_EINVAL, "empty batch or sequence");
    if (gbatch != 1 && gbatch != B)
        return fail(PM_EINVAL, "global_batch must be 1 or batch");
    const Plan p = make_plan(h, B, T);
    if (ws_bytes < p.total)
        return fail(PM_ENOMEM, "workspace %zu < required %zu", ws_bytes, p.total);
    char* base = (char*)ws;
    float* feat = (float*)(base + p.off_feat);
    float* gbias = (float*)(base + p.off_gbias);
    float* buf[4];
    for (int i = 0; i < 4; ++i)
        buf[i] = (float*)(base + p.off_buf) + (size_t)i * p.buf_elems;

    const float* feat_cl = features;
    if (!features_cl) {
        dim3 grid((T + 31) / 32, h->cfp / 32, B);
        PROF(h, s, "to_channels_last", 0,
             (double)B * T * (h->cfg.num_features + h->cfp) * 4, {
            hipLaunchKernelGGL(pm_to_channels_last_kernel, grid, dim3(256), 0,
                               s, features, feat, h->cfg.num_features, T,
                               h->cfp);
            HIP_TRY(hipGetLastError());
        });
        feat_cl = feat;
    }
    // speaker conditioning as a per-utterance bias of the input conv
    hipLaunchKernelGGL(pm_speaker_bias_kernel, dim3((h->c0p + 3) / 4, gbatch),
                       dim3(256), 0, s, g, h->spk_w, h->spk_b, gbias,
                       h->cfg.global_channels, h->c0, h->c0p);
    HIP_TRY(hipGetLastError());
    {
        SingleArgs a = {};
        a.x = feat_cl; a.out = buf[0]; a.w = h->in_conv.w;
        a.bias = h->in_conv.bias; a.gbias = gbias; a.gbias_batch = gbatch;
        a.B = B; a.L = T; a.Lout = T; a.Cin = h->cfp; a.M = h->c0p;
        a.lrelu = 0; a.pad = 3; a.phase_r = 0; a.phase_c = 1;
        a.lengths = lengths; a.len_scale = 1;
        PROF(h, s, "input_conv",
             2.0 * h->c0 * h->cfg.num_features * 7 * B * T,
             (double)B * T * (h->cfg.num_features + h->c0) * 4, {
            HIP_TRY(launch_single(h->in_conv.dtype, 0, h->in_conv.geom.ch,
                                  h->in_conv.cfg, a, s));
        });
    }
    int xi = 0;        // index of the buffer holding the stage input
    // the stage input once more, as the 16-bit operand values of the next
    // upsampler (written by the previous stage's last Block launch when that
    // is a skewed walk - Block3Args::act16), or null
    const void* x16 = nullptr;
    int L = T;
    int rate = 1;      // samples per frame at the current stage
    const float scale = 1.f / (float)h->cfg.num_resblocks;
    for (auto& st : h->stages) {
        const int ui = (xi + 1) & 3, ai = (xi + 2) & 3, bi = (xi + 3) & 3;
        {
            SingleArgs a = {};
            a.x = buf[xi]; a.out = buf[ui]; a.w = st.up.w; a.bias = st.up.bias;
            a.gbias = nullptr; a.gbias_batch = 1;
            a.B = B; a.L = L; a.Lout = L; a.Cin = st.cin_pad;
            a.M = st.up.geom.M; a.lrelu = 1; a.pad = 1;
            a.phase_c = st.cout_pad; a.phase_p = st.r / 2; a.phase_r = st.r;
            a.lengths = lengths; a.len_scale = rate;
            a.x16 = x16;
            x16 = nullptr;
            char label[64];
            snprintf(label, sizeof(label), "convT_c%d_r%d", st.cin, st.r);
            PROF(h, s, label, 2.0 * st.cin * st.cout * st.k * B * L,
                 (double)B * L * (st.cin + (double)st.r * st.cout) * 4, {
                // (long batches: the 256-column variant of cfg 3)
                int cfg = st.up.cfg;
                if (cfg == 3 && esz(st.up.dtype) == 2 && st.up.geom.ch == 64 &&
                    (long long)B * ((L + 255) / 256) >= 4 * 256)
                    cfg = 5;
                HIP_TRY(launch_single(st.up.dtype, 1, st.up.geom.ch, cfg, a, s));
            });
        }
        L *= st.r;
        rate *= st.r;
        const int si = xi;   // stage input is dead after the upsampler
        // whole MRF (Blocks k = 3, 7, 11) in one launch: U -> S
        bool mrf_done = false;
        // (split-f16 operands on a batch long enough for the skewed walk: Block
        // by Block - three launches that recompute nothing beat the fused
        // whole-MRF launch and its 23 % halo there, pm_launch.h)
        const bool x3_skew = pm_x3skew_id(st.dtype) && st.cout_pad == 32 &&
            p.scratch && pm_device_cus() > 0 &&
            (L / 512) / std::max(1, pm_device_cus() / B) >= 4;
        if (fusion_level() >= 2 && h->cfg.num_resblocks == 3 &&
            h->cfg.num_dilations <= 3 && st.cout_pad <= 64 &&
            h->cfg.resblock_kernel_sizes[0] == 3 &&
            h->cfg.resblock_kernel_sizes[1] == 7 &&
            h->cfg.resblock_kernel_sizes[2] == 11) {
            Block3Args blocks[3] = {};
            double flops = 0;
            for (int j = 0; j < 3; ++j) {
                Block3Args& a = blocks[j];
                a.x = buf[ui]; a.out = buf[si];
                a.niter = h->cfg.num_dilations;
                for (int n = 0; n < a.niter; ++n) {
                    a.w1[n] = st.c1[j][n].w;
                    a.w2[n] = st.c2[j][n].w;
                    a.dil[n] = h->cfg.resblock_dilations[j][n];
                    flops += 4.0 * st.cout * st.cout *
                             h->cfg.resblock_kernel_sizes[j] * B * L;
                }
                a.B = B; a.L = L; a.mode = j == 0 ? 1 : 2; a.scale = scale;
                a.lengths = lengths; a.len_scale = rate;
                // (4-byte operand layouts on a long batch: the skewed
                // whole-MRF walk, or Block by Block on the skewed walk - never
                // the two-sided tiling of the fused launch, pm_launch.h)
                a.scratch = p.scratch ? base + p.off_scratch : nullptr;
                a.scratch_bytes = p.scratch;
                a.skew_only = x3_skew ? 1 : 0;
            }
            char label[64];
            snprintf(label, sizeof(label), "mrf_c%d", st.cout);
            hipError_t e = hipSuccess;
            const size_t marks_before = h->marks.size();
            PROF(h, s, label, flops, (double)B * L * st.cout * 4 * 2, {
                e = launch_mrf(st.dtype, st.cout_pad, blocks, s);
                if (e != hipSuccess && e != hipErrorNotSupported) HIP_TRY(e);
            });
            mrf_done = e == hipSuccess;
            if (!mrf_done && h->marks.size() > marks_before)
                h->marks.pop_back();
        }
        for (int j = 0; !mrf_done && j < h->cfg.num_resblocks; ++j) {
            const int K = h->cfg.resblock_kernel_sizes[j];
            bool fused = false;
            if (h->cfg.num_dilations <= 3 &&
                block3_supported(st.dtype, st.cout_pad, K)) {
                // whole Block (all dilations) in one kernel: U -> S
                Block3Args a = {};
                a.x = buf[ui]; a.out = buf[si];
                a.niter = h->cfg.num_dilations;
                double flops = 0;
                for (int n = 0; n < a.niter; ++n) {
                    a.w1[n] = st.c1[j][n].w;
                    a.w2[n] = st.c2[j][n].w;
                    a.dil[n] = h->cfg.resblock_dilations[j][n];
                    flops += 4.0 * st.cout * st.cout * K * B * L;
                }
                a.B = B; a.L = L; a.mode = j == 0 ? 1 : 2; a.scale = scale;
                a.lengths = lengths; a.len_scale = rate;
                a.scratch = p.scratch ? base + p.off_scratch : nullptr;
                a.scratch_bytes = p.scratch;
                // The stage's last Block launch completes `out`, whose only
                // reader is the next stage's upsampler - which stages
                // cvt(lrelu(out)): let the kernel write exactly that (half the
                // bytes out, half the bytes in, no staging VALU) when the
                // upsampler is the conv_single_kernel of a 16-bit stage.
                int act16_done = 0;
                const size_t si_index = &st - &h->stages[0];
                if (j == h->cfg.num_resblocks - 1 && a.mode == 2 &&
                    si_index + 1 < h->stages.size()) {
                    const Stage& next = h->stages[si_index + 1];
                    if (esz(next.up.dtype) == 2 && next.up.cfg != 4) {
                        a.act16 = buf[ai];      // (free: no pair iterations)
                        a.act16_type = next.up.dtype;
                        a.act16_done = &act16_done;
                    }
                }
                char label[64];
                snprintf(label, sizeof(label), "block_c%d_k%d", st.cout, K);
                hipError_t e = hipSuccess;
                const size_t marks_before = h->marks.size();
                PROF(h, s, label, flops,
                     (double)B * L * st.cout * 4 * (a.mode == 2 ? 3 : 2), {
                    e = launch_block3(st.dtype, st.cout_pad, K, a, s);
                    if (e != hipSuccess && e != hipErrorNotSupported) HIP_TRY(e);
                });
                fused = e == hipSuccess;
                if (!fused && h->marks.size() > marks_before)
                    h->marks.pop_back();
                if (fused && act16_done) x16 = buf[ai];
            }
            if (fused) continue;
            const float* src = buf[ui];
            for (int n = 0; n < h->cfg.num_dilations; ++n) {
                const bool last = n == h->cfg.num_dilations - 1;
                float* dst = last ? buf[si] : ((n & 1) ? buf[bi] : buf[ai]);
                PairArgs a = {};
                a.x = src; a.out = dst;
                a.w1 = st.c1[j][n].w;
                a.w2 = st.c2[j][n].w;
                a.B = B; a.L = L;
                a.dilation = h->cfg.resblock_dilations[j][n];
                a.mode = last ? (j == 0 ? 1 : 2) : 0;
                a.scale = scale;
                a.lengths = lengths; a.len_scale = rate;
                char label[64];
                snprintf(label, sizeof(label), "pair_c%d_k%d", st.cout, K);
                PROF(h, s, label, 4.0 * st.cout * st.cout * K * B * L,
                     (double)B * L * st.cout * 4 * (a.mode == 2 ? 3 : 2), {
                    HIP_TRY(launch_pair(st.dtype, st.cout_pad, K, a, s));
                });
                src = dst;
            }
        }
        xi = si;
    }
    {
        constexpr int TH = 256;
        const int C = h->c_last_pad;
        PROF(h, s, "out_conv_tanh", 2.0 * h->c_last * 7 * B * L,
             (double)B * L * (h->c_last + 1) * 4, {
            if (C == 32 && h->c_last == 32) {
                constexpr int T32 = 128;   // 38 KB of LDS: four per CU
                const size_t smem = (size_t)32 * PM_OUT32_RL(T32) * sizeof(float);
                hipLaunchKernelGGL(pm_out_conv32_kernel<T32>,
                                   dim3((L + T32 * 2 - 1) / (T32 * 2), B),
                                   dim3(T32), smem, s, buf[xi], h->out_w, out,
                                   L, lengths, rate);
            } else {
                const size_t smem =
                    ((size_t)(TH + 6) * (C + 1) + 7 * C) * sizeof(float);
                hipLaunchKernelGGL(pm_out_conv_kernel<TH>,
                                   dim3((L + TH - 1) / TH, B), dim3(TH), smem,
                                   s, buf[xi], h->out_w, out, L, C, h->c_last,
                                   lengths, rate);
            }
            HIP_TRY(hipGetLastError());
        });
    }
    return PM_OK;
}

// ---- profiling API ---------------------------------------------------------
extern "C" int pm_hifigan_profile_enable(pm_hifigan_t h, int enable) {
    if (!h) return fail(PM_EINVAL, "null handle");
    h->profile = enable != 0;
    return PM_OK;
}

// Restrict the event pairs to the launches reported under `label` (null or
// "": all of them): a timed region then carries two events per step instead
// of two per launch.
extern "C" int pm_hifigan_profile_only(pm_hifigan_t h, const char* label) {
    if (!h) return fail(PM_EINVAL, "null handle");
    h->profile_only = label ? label : "";
    return PM_OK;
}

// Synchronise, fold the event pairs recorded since the last call into the
// per-label totals and recycle the events. Call between forwards at will.
extern "C" int pm_hifigan_profile_collect(pm_hifigan_t h) {
    if (!h) return fail(PM_EINVAL, "null handle");
    for (auto& m : h->marks) {
        HIP_TRY(hipEventSynchronize(h->events[m.e1]));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, h->events[m.e0], h->events[m.e1]));
        auto& acc = h->totals[m.label];
        acc.count += 1; acc.ms += ms; acc.flops += m.flops; acc.bytes += m.bytes;
    }
    h->marks.clear();
    h->events_used = 0;
    return PM_OK;
}

extern "C" int pm_hifigan_profile_reset(pm_hifigan_t h) {
    if (!h) return fail(PM_EINVAL, "null handle");
    int rc = pm_hifigan_profile_collect(h);
    h->totals.clear();
    return rc;
}

// Text report, one line per kernel label:
//   label launches total_ms algorithmic_flops algorithmic_bytes
extern "C" const char* pm_hifigan_profile_report(pm_hifigan_t h) {
    if (!h) return "";
    h->report.clear();
    char line[256];
    for (auto& kv : h->totals) {
        snprintf(line, sizeof(line), "%s %lld %.6f %.6e %.6e\n",
                 kv.first.c_str(), kv.second.count, kv.second.ms,
                 kv.second.flops, kv.second.bytes);
        h->report += line;
    }
    return h->report.c_str();
}

extern "C" int pm_hifigan_forward(
    pm_hifigan_t h, const float* features, const float* g, int gbatch,
    float* out, int B, int T, void* ws, size_t ws_bytes, void* stream) {
    return forward_impl(h, features, false, g, gbatch, out, B, T, ws, ws_bytes,
                        (hipStream_t)stream);
}

extern "C" int pm_hifigan_forward_cl(
    pm_hifigan_t h, const float* features_cl, const float* g, int gbatch,
    float* out, int B, int T, void* ws, size_t ws_bytes, void* stream) {
    return forward_impl(h, features_cl, true, g, gbatch, out, B, T, ws,
                        ws_bytes, (hipStream_t)stream);
}

// Ragged batch: utterance b is lengths[b] <= T frames long inside the padded
// (B, T, ...) tensors. Every kernel treats frames >= lengths[b] as outside
// the sequence (the convolutions' zero padding), so out[b, :, :256 lengths[b]]
// is bit-identical to synthesising utterance b alone; the tail is zeros.
extern "C" int pm_hifigan_forward_ragged(
    pm_hifigan_t h, const float* features, int features_cl, const float* g,
    int gbatch, const int* lengths, float* out, int B, int T, void* ws,
    size_t ws_bytes, void* stream) {
    if (!lengths) return fail(PM_EINVAL, "null lengths");
    return forward_impl(h, features, features_cl != 0, g, gbatch, out, B, T,
                        ws, ws_bytes, (hipStream_t)stream, lengths);
}

// ---------------------------------------------------------------------------
// conditioning
// ---------------------------------------------------------------------------
extern "C" int pm_prepare_features(
    const float* loudness, const float* pitch, const float* periodicity,
    const float* ppg, const float* pitch_edges, const float* pitch_table,
    float* out_ref, float* out_cl, int B, int T, int F, int P, int NB, int E,
    int bands, int cl_channels, int sparse_method, float ppg_threshold,
    float fmin, float fmax, float min_db, float ref_db, float period_rate,
    void* stream) {
    if (!loudness || !pitch || !periodicity || !ppg || !pitch_edges ||
        !pitch_table || (!out_ref && !out_cl))
        return fail(PM_EINVAL, "null argument");
    if (B < 1 || T < 1 || P < 2 || bands < 1 || bands > 16 || F < bands)
        return fail(PM_EINVAL, "bad feature dimensions");
    if (sparse_method < PM_SPARSE_NONE || sparse_method > PM_SPARSE_TOPK)
        return fail(PM_EINVAL, "sparse_method %d unknown", sparse_method);
    const int C = P + E + bands + 1 + (period_rate > 0.f ? 1 : 0);
    if (out_cl && cl_channels < C)
        return fail(PM_EINVAL, "cl_channels %d < %d", cl_channels, C);
    FeatureArgs a;
    a.loudness = loudness; a.pitch = pitch; a.periodicity = periodicity;
    a.ppg = ppg; a.pitch_edges = pitch_edges; a.pitch_table = pitch_table;
    a.out_cl = out_cl; a.out_ref = out_ref;
    a.B = B; a.T = T; a.F = F; a.P = P; a.NB = NB; a.E = E; a.bands = bands;
    a.Cpad = cl_channels;
    const double step = (double)F / (double)bands;   // generator.py:174
    for (int b = 0; b <= bands; ++b) a.band_start[b] = (int)(b * step);
    a.sparse_method = sparse_method;
    a.rank_below = a.rank_above = 0; a.rank_weight = 0.f;
    a.threshold = ppg_threshold; a.topk = 0;
    if (sparse_method == PM_SPARSE_PERCENTILE) {
        if (!(ppg_threshold >= 0.f && ppg_threshold <= 1.f))
            return fail(PM_EINVAL, "percentile threshold must be in [0, 1]");
        // torch.quantile(..., interpolation='linear') in the tensor's dtype
        const float rank = ppg_threshold * (float)(P - 1);
        a.rank_below = (int)floorf(rank);
        a.rank_above = (int)ceilf(rank);
        a.rank_weight = rank - floorf(rank);
    } else if (sparse_method == PM_SPARSE_TOPK) {
        a.topk = (int)ppg_threshold;
        if (a.topk < 1 || a.topk > P)
            return fail(PM_EINVAL, "topk must be in [1, %d]", P);
    }
    a.fmin = fmin; a.fmax = fmax; a.min_db = min_db;
    a.db_range = ref_db - min_db;
    a.period_rate = period_rate;
    constexpr int FR = 64;     // frames per workgroup
    const int width = a.Cpad > C ? a.Cpad : C;
    const size_t smem = ((size_t)P * FR + (size_t)FR * (width + 1) + 6 * FR) *
                        sizeof(float);
    hipLaunchKernelGGL(pm_prepare_features_kernel<FR>,
                       dim3((T + FR - 1) / FR, B), dim3(256), smem,
                       (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return PM_OK;
}

extern "C" int pm_prepare_global_features(
    const int64_t* speakers, const float* sbr, const float* lr,
    const float* table, float* out, int B, int S, int num_speakers,
    void* stream) {
    if (!speakers || !table || !out) return fail(PM_EINVAL, "null argument");
    if (B < 1 || S < 1 || num_speakers < 1)
        return fail(PM_EINVAL, "bad dimensions");
    hipLaunchKernelGGL(pm_global_features_kernel, dim3(B), dim3(256), 0,
                       (hipStream_t)stream, (const long long*)speakers, sbr,
                       lr, table, out, B, S, num_speakers);
    HIP_TRY(hipGetLastError());
    return PM_OK;
}

extern "C" int pm_prepare_global_features_linear(
    const float* emb, const float* weight, const float* bias,
    const float* sbr, const float* lr, float* out, int B, int E, int S,
    void* stream) {
    if (!emb || !weight || !bias || !out) return fail(PM_EINVAL, "null argument");
    if (B < 1 || E < 1 || S < 1) return fail(PM_EINVAL, "bad dimensions");
    hipLaunchKernelGGL(pm_global_features_linear_kernel,
                       dim3((S + 3) / 4, B), dim3(256), 0,
                       (hipStream_t)stream, emb, weight, bias, sbr, lr, out, B,
                       E, S);
    HIP_TRY(hipGetLastError());
    return PM_OK;
}

// ---------------------------------------------------------------------------
// per-kernel entry points
// ---------------------------------------------------------------------------
extern "C" size_t pm_op_workspace_bytes(int c_in, int c_out, int k) {
    const size_t ci = pad32(c_in), co = pad32(c_out);
    // two packed weight streams (fp32-sized, + bias steps) and two padded biases
    return 2 * align256((ci * co * (size_t)k + co * 16) * 4) +
           2 * align256(co * 64 * 4);
}

extern "C" int pm_block_iteration_cl(
    int dtype, const float* x, float* out, const float* w1, const float* b1,
    const float* w2, const float* b2, int B, int L, int C, int K, int d,
    int mode, float scale, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !out || !w1 || !b1 || !w2 || !b2 || !ws)
        return fail(PM_EINVAL, "null argument");
    const int Cp = pad32(C);
    if (Cp != 32 && Cp != 64 && Cp != 128 && Cp != 256)
        return fail(PM_EINVAL, "channels %d unsupported", C);
    if ((K != 3 && K != 7 && K != 11) || d < 1 || d > 5)
        return fail(PM_EINVAL, "kernel %d / dilation %d unsupported", K, d);
    if (ws_bytes < pm_op_workspace_bytes(C, C, K))
        return fail(PM_ENOMEM, "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    ConvGeom g;
    g.mode = 0; g.cout = g.cin = C; g.k = K; g.cout_pad = g.cin_pad = g.M = Cp;
    g.kt = K; g.ch = pair_chunk(dtype, Cp); g.bias_step = true;
    char* base = (char*)ws;
    const size_t wsz = align256(((size_t)Cp * Cp * K + Cp * 16) * 4);
    void* p1 = base; void* p2 = base + wsz;
    float* pb1 = (float*)(base + 2 * wsz);
    float* pb2 = pb1 + align256(Cp * 64 * 4) / 4;
    HIP_TRY(pack_weights(dtype, g, w1, p1, s));
    HIP_TRY(pack_weights(dtype, g, w2, p2, s));
    HIP_TRY(pad_bias(b1, pb1, C, Cp, 1, s));
    HIP_TRY(pad_bias(b2, pb2, C, Cp, 1, s));
    HIP_TRY(pack_bias_step(dtype, g, pb1, p1, s));
    HIP_TRY(pack_bias_step(dtype, g, pb2, p2, s));
    PairArgs a = {};
    a.x = x; a.out = out; a.w1 = p1; a.w2 = p2;
    a.B = B; a.L = L; a.dilation = d; a.mode = mode; a.scale = scale;
    HIP_TRY(launch_pair(dtype, Cp, K, a, s));
    return PM_OK;
}

// Debug (-DPM_TUNING builds): subsequent pair / whole-Block launches make
// wave 0 of workgroup i write 16 shader-clock stamps to timeline[16 i ..]
// (NULL switches it off). The shipped library has no such instrumentation
// and returns PM_ESTATE.
extern "C" int pm_debug_timeline(void* dev_buffer) {
#ifdef PM_TUNING
    g_timeline = (unsigned long long*)dev_buffer;
    return PM_OK;
#else
    (void)dev_buffer;
    return fail(PM_ESTATE, "pm_debug_timeline needs a -DPM_TUNING build "
                           "(make TUNING=1)");
#endif
}

static int block_cl_impl(
    int dtype, const float* x, float* out, const float* const* w1,
    const float* const* b1, const float* const* w2, const float* const* b2,
    const int* dilations, int niter, int B, int L, int C, int K, int mode,
    float scale, void* ws, size_t ws_bytes, void* stream, void* act16,
    int act_dtype) {
    if (!x || !out || !w1 || !b1 || !w2 || !b2 || !dilations || !ws)
        return fail(PM_EINVAL, "null argument");
    const int Cp = pad32(C);
    if (Cp > 256) return fail(PM_EINVAL, "channels %d unsupported (<= 256)", C);
    if ((K != 3 && K != 7 && K != 11) || niter < 1 || niter > 3)
        return fail(PM_EINVAL, "kernel %d / %d iterations unsupported", K, niter);
    if (ws_bytes < 3 * pm_op_workspace_bytes(C, C, K))
        return fail(PM_ENOMEM, "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    ConvGeom g;
    g.mode = 0; g.cout = g.cin = C; g.k = K; g.cout_pad = g.cin_pad = g.M = Cp;
    g.kt = K; g.ch = Cp < 64 ? Cp : 64; g.bias_step = true;
    Block3Args a = {};
    a.x = x; a.out = out; a.niter = niter; a.B = B; a.L = L; a.mode = mode;
    a.scale = scale;
    const size_t per = pm_op_workspace_bytes(C, C, K);
    const size_t wsz = align256(((size_t)Cp * Cp * K + Cp * 16) * 4);
    for (int n = 0; n < niter; ++n) {
        char* base = (char*)ws + n * per;
        void* p1 = base; void* p2 = base + wsz;
        float* pb1 = (float*)(base + 2 * wsz);
        float* pb2 = pb1 + align256(Cp * 64 * 4) / 4;
        HIP_TRY(pack_weights(dtype, g, w1[n], p1, s));
        HIP_TRY(pack_weights(dtype, g, w2[n], p2, s));
        HIP_TRY(pad_bias(b1[n], pb1, C, Cp, 1, s));
        HIP_TRY(pad_bias(b2[n], pb2, C, Cp, 1, s));
        HIP_TRY(pack_bias_step(dtype, g, pb1, p1, s));
        HIP_TRY(pack_bias_step(dtype, g, pb2, p2, s));
        a.w1[n] = p1; a.w2[n] = p2;
        a.dil[n] = dilations[n];
    }
    // (what the caller hands over beyond the packed weights serves the skewed
    // walk: pm_walk_scratch_bytes)
    if (ws_bytes > 3 * per) {
        a.scratch = (char*)ws + 3 * per;
        a.scratch_bytes = ws_bytes - 3 * per;
    }
    int act16_done = 0;
    if (act16) {
        a.act16 = act16; a.act16_type = act_dtype; a.act16_done = &act16_done;
    }
    hipError_t e = launch_block3(dtype, Cp, K, a, s);
    if (e == hipErrorNotSupported)
        return fail(PM_EINVAL, "no whole-Block kernel for this shape");
    HIP_TRY(e);
    if (act16 && !act16_done)
        return fail(PM_ESTATE, "the launch did not take the skewed walk: `out` "
                    "holds the fp32 result, the 16-bit operand copy was not "
                    "written");
    return PM_OK;
}

extern "C" int pm_block_cl(
    int dtype, const float* x, float* out, const float* const* w1,
    const float* const* b1, const float* const* w2, const float* const* b2,
    const int* dilations, int niter, int B, int L, int C, int K, int mode,
    float scale, void* ws, size_t ws_bytes, void* stream) {
    return block_cl_impl(dtype, x, out, w1, b1, w2, b2, dilations, niter, B, L,
                         C, K, mode, scale, ws, ws_bytes, stream, nullptr, 0);
}

// pm_block_cl whose result leaves as the NEXT upsampler's MFMA operand instead
// of fp32 (Block3Args::act16): act16 (B, L, c_pad) 16-bit = cvt(lrelu(result)),
// act_dtype PM_F16 or PM_BF16; `out` is read (mode 2) and not written. Only the
// skewed walk does this: PM_ESTATE (and the plain fp32 result in `out`) when the
// launcher took another kernel.
extern "C" int pm_block_act16_cl(
    int dtype, int act_dtype, const float* x, float* out, void* act16,
    const float* const* w1, const float* const* b1, const float* const* w2,
    const float* const* b2, const int* dilations, int niter, int B, int L,
    int C, int K, int mode, float scale, void* ws, size_t ws_bytes,
    void* stream) {
    if (!act16) return fail(PM_EINVAL, "null argument");
    if (act_dtype != PM_F16 && act_dtype != PM_BF16)
        return fail(PM_EINVAL, "act_dtype must be PM_F16 or PM_BF16");
    return block_cl_impl(dtype, x, out, w1, b1, w2, b2, dilations, niter, B, L,
                         C, K, mode, scale, ws, ws_bytes, stream, act16,
                         act_dtype);
}

// Whole MRF ResidualBlock (hifigan.py:141-145): out = (B_3(x) + B_7(x) +
// B_11(x)) / 3, the three Blocks k = 3, 7, 11 (niter dilations each) in ONE
// launch with their sum held in registers. Only where the whole-MRF kernel
// exists (32 channels); w1 / b1 / w2 / b2 are HOST arrays of 3 * niter device
// pointers, Block-major. workspace >= 3 * niter * pm_op_workspace_bytes(c, c, 11).
extern "C" int pm_mrf_cl(
    int dtype, const float* x, float* out, const float* const* w1,
    const float* const* b1, const float* const* w2, const float* const* b2,
    const int* dilations, int niter, int B, int L, int C, void* ws,
    size_t ws_bytes, void* stream) {
    if (!x || !out || !w1 || !b1 || !w2 || !b2 || !dilations || !ws)
        return fail(PM_EINVAL, "null argument");
    const int Cp = pad32(C);
    if (Cp != 32)
        return fail(PM_EINVAL, "whole-MRF kernel exists for <= 32 channels");
    if (niter < 1 || niter > 3) return fail(PM_EINVAL, "1..3 dilations");
    const size_t per = pm_op_workspace_bytes(C, C, 11);
    if (ws_bytes < 3 * (size_t)niter * per)
        return fail(PM_ENOMEM, "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    static const int KS[3] = {3, 7, 11};
    Block3Args blocks[3] = {};
    for (int j = 0; j < 3; ++j) {
        ConvGeom g;
        g.mode = 0; g.cout = g.cin = C; g.k = KS[j];
        g.cout_pad = g.cin_pad = g.M = Cp; g.kt = KS[j]; g.ch = Cp;
        g.bias_step = true;
        Block3Args& a = blocks[j];
        a.x = x; a.out = out; a.niter = niter; a.B = B; a.L = L;
        a.mode = j == 0 ? 1 : 2; a.scale = 1.f / 3.f;
        const size_t wsz = align256(((size_t)Cp * Cp * KS[j] + Cp * 16) * 4);
        for (int n = 0; n < niter; ++n) {
            char* base = (char*)ws + (size_t)(j * niter + n) * per;
            void* p1 = base; void* p2 = base + wsz;
            float* pb1 = (float*)(base + 2 * wsz);
            float* pb2 = pb1 + align256(Cp * 64 * 4) / 4;
            const int i = j * niter + n;
            HIP_TRY(pack_weights(dtype, g, w1[i], p1, s));
            HIP_TRY(pack_weights(dtype, g, w2[i], p2, s));
            HIP_TRY(pad_bias(b1[i], pb1, C, Cp, 1, s));
            HIP_TRY(pad_bias(b2[i], pb2, C, Cp, 1, s));
            HIP_TRY(pack_bias_step(dtype, g, pb1, p1, s));
            HIP_TRY(pack_bias_step(dtype, g, pb2, p2, s));
            a.w1[n] = p1; a.w2[n] = p2; a.dil[n] = dilations[n];
        }
    }
    // (what the caller hands over beyond the packed weights serves the skewed
    // whole-MRF walk of the 4-byte operand layouts: pm_walk_scratch_bytes)
    if (ws_bytes > 3 * (size_t)niter * per)
        for (int j = 0; j < 3; ++j) {
            blocks[j].scratch = (char*)ws + 3 * (size_t)niter * per;
            blocks[j].scratch_bytes = ws_bytes - 3 * (size_t)niter * per;
        }
    hipError_t e = launch_mrf(dtype, Cp, blocks, s);
    if (e == hipErrorNotSupported)
        return fail(PM_EINVAL, "no whole-MRF kernel for this shape");
    HIP_TRY(e);
    return PM_OK;
}

// Input layers (hifigan.py:19-30, 67-68): Conv1d(c_in -> c_out, k 7, pad 3) on
// channels-last features plus the speaker conditioning Conv1d(G -> c_out, k 1)
// of the (B|1, G) global features, added as a per-utterance bias.
//   x_cl (B, L, pad32(c_in)), w (c_out, c_in, 7), bias (c_out),
//   global (gbatch, G), ws_w (c_out, G), ws_b (c_out)  ->  (B, L, pad32(c_out))
extern "C" int pm_input_conv_cl(
    int dtype, const float* x, float* out, const float* w, const float* bias,
    const float* global, const float* ws_w, const float* ws_b, int gbatch,
    int G, int B, int L, int c_in, int c_out, void* ws, size_t ws_bytes,
    void* stream) {
    if (!x || !out || !w || !bias || !global || !ws_w || !ws_b || !ws)
        return fail(PM_EINVAL, "null argument");
    if (gbatch != 1 && gbatch != B)
        return fail(PM_EINVAL, "global batch must be 1 or batch");
    const int cip = pad32(c_in), cop = pad32(c_out);
    if (cop % 64 && cop != 32)
        return fail(PM_EINVAL, "output channels %d unsupported", c_out);
    if (ws_bytes < pm_op_workspace_bytes(c_in, c_out, 7) +
                       align256((size_t)B * cop * 4))
        return fail(PM_ENOMEM, "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    ConvGeom g;
    g.mode = 0; g.cout = c_out; g.cin = c_in; g.k = 7;
    g.cout_pad = g.M = cop; g.cin_pad = cip; g.kt = 7;
    g.ch = (cip % 64 == 0) ? 64 : 32;
    const int cfg = single_cfg(g.M, g.ch, 1);
    char* base = (char*)ws;
    const size_t wsz = 2 * align256(((size_t)cip * cop * 7 + cop * 16) * 4);
    float* pb = (float*)(base + wsz);
    float* gbias = (float*)(base + pm_op_workspace_bytes(c_in, c_out, 7));
    HIP_TRY(pack_weights(dtype, g, w, base, s));
    HIP_TRY(pad_bias(bias, pb, c_out, cop, 1, s));
    hipLaunchKernelGGL(pm_speaker_bias_kernel, dim3((cop + 3) / 4, gbatch),
                       dim3(256), 0, s, global, ws_w, ws_b, gbias, G, c_out,
                       cop);
    HIP_TRY(hipGetLastError());
    SingleArgs a = {};
    a.x = x; a.out = out; a.w = base; a.bias = pb; a.gbias = gbias;
    a.gbias_batch = gbatch; a.B = B; a.L = L; a.Lout = L; a.Cin = cip;
    a.M = cop; a.lrelu = 0; a.pad = 3; a.phase_r = 0; a.phase_c = 1;
    HIP_TRY(launch_single(dtype, 0, g.ch, cfg, a, s));
    return PM_OK;
}

static int conv_transpose_cl_impl(
    int dtype, const float* x, const void* x16, float* out, const float* w,
    const float* bias, int B, int L, int c_in, int c_out, int r, int lrelu,
    void* ws, size_t ws_bytes, void* stream) {
    if ((!x && !x16) || !out || !w || !bias || !ws)
        return fail(PM_EINVAL, "null argument");
    if (r < 2 || (r & 1)) return fail(PM_EINVAL, "rate %d unsupported", r);
    if (ws_bytes < pm_op_workspace_bytes(c_in, c_out, 2 * r))
        return fail(PM_ENOMEM, "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    ConvGeom g;
    g.mode = 1; g.cout = c_out; g.cin = c_in; g.k = 2 * r;
    g.cout_pad = pad32(c_out); g.cin_pad = pad32(c_in);
    g.M = r * g.cout_pad; g.kt = 2; g.r = r; g.p = r / 2;
    g.ch = (g.cin_pad % 64 == 0) ? 64 : 32;
    const int cfg = upsample_whole_k(dtype, g)
        ? 4
        : single_cfg(g.M, g.ch, ((r / 2) * g.cout_pad) % 64 == 0);
    char* base = (char*)ws;
    const size_t wsz = 2 * align256((size_t)g.cin_pad * g.cout_pad * g.k * 4);
    float* pb = (float*)(base + wsz);
    HIP_TRY(pack_weights(dtype, g, w, base, s));
    HIP_TRY(pad_bias(bias, pb, c_out, g.cout_pad, r, s));
    SingleArgs a = {};
    a.x = x; a.out = out; a.w = base; a.bias = pb; a.gbias = nullptr;
    a.gbias_batch = 1; a.B = B; a.L = L; a.Lout = L; a.Cin = g.cin_pad;
    a.M = g.M; a.lrelu = lrelu; a.pad = 1;
    a.phase_c = g.cout_pad; a.phase_p = r / 2; a.phase_r = r;
    if (x16) {
        if (esz(dtype) != 2 || cfg == 4)
            return fail(PM_EINVAL, "a 16-bit operand input needs a 16-bit "
                        "operand type and the narrow upsampler kernel");
        a.x16 = x16;
    }
    HIP_TRY(launch_single(dtype, 1, g.ch, cfg, a, s));
    return PM_OK;
}

extern "C" int pm_conv_transpose_cl(
    int dtype, const float* x, float* out, const float* w, const float* bias,
    int B, int L, int c_in, int c_out, int r, int lrelu, void* ws,
    size_t ws_bytes, void* stream) {
    if (!x) return fail(PM_EINVAL, "null argument");
    return conv_transpose_cl_impl(dtype, x, nullptr, out, w, bias, B, L, c_in,
                                  c_out, r, lrelu, ws, ws_bytes, stream);
}

// pm_conv_transpose_cl on an input that already holds the operand values:
// x16_cl (B, L, c_in_pad) in the operand type `dtype` = cvt(lrelu(x)), as
// pm_block_act16_cl writes it - staged as it is (SingleArgs::x16).
extern "C" int pm_conv_transpose_x16_cl(
    int dtype, const void* x16, float* out, const float* w, const float* bias,
    int B, int L, int c_in, int c_out, int r, void* ws, size_t ws_bytes,
    void* stream) {
    if (!x16) return fail(PM_EINVAL, "null argument");
    return conv_transpose_cl_impl(dtype, nullptr, x16, out, w, bias, B, L, c_in,
                                  c_out, r, 1, ws, ws_bytes, stream);
}

extern "C" int pm_out_conv_tanh(
    const float* x, const float* w, float* out, int B, int L, int C,
    void* stream) {
    if (!x || !w || !out) return fail(PM_EINVAL, "null argument");
    const int Cp = pad32(C);
    if (Cp > 64) return fail(PM_EINVAL, "channels %d unsupported", C);
    constexpr int TH = 256;
    const size_t smem = ((size_t)(TH + 6) * (Cp + 1) + 7 * Cp) * sizeof(float);
    hipLaunchKernelGGL(pm_out_conv_kernel<TH>, dim3((L + TH - 1) / TH, B),
                       dim3(TH), smem, (hipStream_t)stream, x, w, out, L, Cp, C,
                       (const int*)nullptr, 1);
    HIP_TRY(hipGetLastError());
    return PM_OK;
}

// Test hook: force the walked whole-Block / whole-MRF kernels (with
// `walk_nseg` segments per utterance) and the number of M groups of the wide
// upsampler, which the launchers otherwise pick from the grid size - so that
// unit-sized inputs reach those code paths. 0 restores the heuristics.
//
// The hooks are OFF unless the process was started with PROMONET_HIP_DEBUG=1
// (tests/conftest.py and the scripts under scripts/ set it): a production
// process cannot have its launch geometry changed under it. Their state is
// per host thread (pm_launch.h), so a test thread's override never reaches a
// forward running on another thread, and every API call reads it on the
// thread that launches.
static bool debug_hooks_enabled() {
    static const bool enabled = [] {
        const char* e = getenv("PROMONET_HIP_DEBUG");
        return e && e[0] == '1';
    }();
    return enabled;
}

extern "C" int pm_debug_force(int walk_nseg, int upsample_groups) {
    if (!debug_hooks_enabled())
        return fail(PM_ESTATE, "debug hooks are disabled "
                    "(start the process with PROMONET_HIP_DEBUG=1)");
    if (walk_nseg < 0 || upsample_groups < 0)
        return fail(PM_EINVAL, "negative value");
    pm_force().walk_nseg = walk_nseg;
    pm_force().upsample_groups = upsample_groups;
    return PM_OK;
}

// Sustained-rate probe of the matrix pipe (bench.py times it with HIP events
// on `stream`): `workgroups` x 4 waves x `iterations` x 16 MFMAs of 32 768 FLOP.
// operands: >= 32 768 bytes of finite values of the operand type (2 048 uint4
// are read). A measurement aid, not a test hook: always available, it changes
// no state of the library.
extern "C" int pm_mfma_probe(int dtype, int iterations, const void* operands,
                             float* sink, int workgroups, void* stream) {
    if (!operands || !sink) return fail(PM_EINVAL, "null argument");
    if (iterations < 1 || workgroups < 1)
        return fail(PM_EINVAL, "bad probe arguments");
    hipStream_t s = (hipStream_t)stream;
    const uint4* src = (const uint4*)operands;
    if (dtype == PM_F16)
        hipLaunchKernelGGL(pm_mfma_probe_kernel<1>, dim3(workgroups), dim3(256),
                           0, s, src, sink, iterations);
    else if (dtype == PM_BF16)
        hipLaunchKernelGGL(pm_mfma_probe_kernel<2>, dim3(workgroups), dim3(256),
                           0, s, src, sink, iterations);
    else
        return fail(PM_EINVAL, "probe operand type must be PM_F16 or PM_BF16");
    HIP_TRY(hipGetLastError());
    return PM_OK;
}

// Test hook: -1 keeps the launchers off the skewed whole-Block walk, 1 takes
// it wherever it fits, 0 restores the default (the shapes it measured faster
// on, when scratch was handed over).
extern "C" int pm_debug_skew(int mode) {
    if (!debug_hooks_enabled())
        return fail(PM_ESTATE, "debug hooks are disabled "
                    "(start the process with PROMONET_HIP_DEBUG=1)");
    if (mode < -1 || mode > 1) return fail(PM_EINVAL, "mode is -1, 0 or 1");
    pm_force().skew = mode;
    return PM_OK;
}

// Scratch bytes the skewed whole-Block walk wants behind the workspace of
// pm_block_cl for a batch of B utterances (optional: without it the walked
// or stand-alone kernels run).
extern "C" size_t pm_walk_scratch_bytes(int B) {
    return B < 1 ? 0 : walk_scratch_bytes(B);
}

extern "C" int pm_fold_weight_norm(
    const float* g, const float* v, float* w, int rows, int cols,
    void* stream) {
    if (!g || !v || !w || rows < 1 || cols < 1)
        return fail(PM_EINVAL, "bad argument");
    hipLaunchKernelGGL(pm_fold_kernel, dim3(rows), dim3(256), 0,
                       (hipStream_t)stream, g, v, w, cols);
    HIP_TRY(hipGetLastError());
    return PM_OK;
}

extern "C" int pm_to_channels_last(
    const float* src, float* dst, int B, int C, int T, int c_pad,
    void* stream) {
    if (!src || !dst || c_pad % 32) return fail(PM_EINVAL, "bad argument");
    dim3 grid((T + 31) / 32, c_pad / 32, B);
    hipLaunchKernelGGL(pm_to_channels_last_kernel, grid, dim3(256), 0,
                       (hipStream_t)stream, src, dst, C, T, c_pad);
    HIP_TRY(hipGetLastError());
    return PM_OK;
}

// ---------------------------------------------------------------------------
// preprocessing
// ---------------------------------------------------------------------------
static const int NFFT = 1024, HOP = 256, BINS = 513, DFT_M = 1088;

static std::mutex g_basis_mutex;
struct DftBasis {
    void* forward = nullptr;     // packed windowed DFT basis (1088 x 256 x 4)
    void* backward = nullptr;    // packed transposed basis (256 x 1088 x 4)
    float* zeros = nullptr;      // 256 zero biases for the backward conv
};
static std::map<int, DftBasis> g_basis;   // per device

static int get_dft_basis(DftBasis* out, hipStream_t s) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_basis_mutex);
    auto it = g_basis.find(dev);
    if (it != g_basis.end()) { *out = it->second; return PM_OK; }
    float* raw = nullptr;
    float* rawt = nullptr;
    DftBasis basis;
    const size_t raw_elems = 2ull * BINS * NFFT;
    const size_t packed_bytes = (size_t)DFT_M * NFFT * sizeof(float);
    HIP_TRY(hipMalloc((void**)&raw, raw_elems * sizeof(float)));
    HIP_TRY(hipMalloc((void**)&rawt, packed_bytes));
    HIP_TRY(hipMalloc(&basis.forward, packed_bytes));
    HIP_TRY(hipMalloc(&basis.backward, packed_bytes));
    HIP_TRY(hipMalloc((void**)&basis.zeros, HOP * sizeof(float)));
    HIP_TRY(hipMemsetAsync(basis.zeros, 0, HOP * sizeof(float), s));
    hipLaunchKernelGGL(pm_dft_basis_kernel,
                       dim3((unsigned)((raw_elems + 255) / 256)), dim3(256), 0,
                       s, raw, BINS, NFFT, HOP);
    HIP_TRY(hipGetLastError());
    ConvGeom g;
    g.mode = 0; g.cout = 2 * BINS; g.cin = HOP; g.k = NFFT / HOP;
    g.cout_pad = g.M = DFT_M; g.cin_pad = HOP; g.kt = g.k; g.ch = 64;
    HIP_TRY(pack_weights(PM_F32, g, raw, basis.forward, s));
    // backward: out[q][c] = sum_j sum_m G[q - 3 + j][m] wt[c][m][j]
    const size_t t_elems = (size_t)HOP * DFT_M * (NFFT / HOP);
    hipLaunchKernelGGL(pm_dft_basis_transpose_kernel,
                       dim3((unsigned)((t_elems + 255) / 256)), dim3(256), 0,
                       s, raw, rawt, 2 * BINS, DFT_M, HOP, NFFT / HOP);
    HIP_TRY(hipGetLastError());
    ConvGeom gt;
    gt.mode = 0; gt.cout = HOP; gt.cin = DFT_M; gt.k = NFFT / HOP;
    gt.cout_pad = gt.M = HOP; gt.cin_pad = DFT_M; gt.kt = gt.k; gt.ch = 64;
    HIP_TRY(pack_weights(PM_F32, gt, rawt, basis.backward, s));
    HIP_TRY(hipStreamSynchronize(s));
    hipFree(raw);
    hipFree(rawt);
    g_basis[dev] = basis;
    *out = basis;
    return PM_OK;
}

extern "C" size_t pm_stft_scratch_bytes(int B, int N) {
    if (B < 1 || N < HOP) return 0;
    const size_t T = N / HOP;
    return align256((size_t)B * (T + 3) * HOP * sizeof(float));
}

static int stft_launch(
    int epi, const float* audio, float* out, unsigned* maxbits, int B, int N,
    void* scratch, size_t scratch_bytes, hipStream_t s,
    const float* grad = nullptr) {
    if (!audio || !out || !scratch) return fail(PM_EINVAL, "null argument");
    const int pad = (NFFT - HOP) / 2;
    if (B < 1 || N <= pad)
        return fail(PM_EINVAL, "need more than %d samples (reflect pad)", pad);
    const int T = N / HOP;
    if (T < 1) return fail(PM_EINVAL, "fewer samples than one hop");
    if (scratch_bytes < pm_stft_scratch_bytes(B, N))
        return fail(PM_ENOMEM, "scratch too small");
    DftBasis basis;
    int rc = get_dft_basis(&basis, s);
    if (rc) return rc;
    float* padded = (float*)scratch;
    // only the first (T + 3) * HOP padded samples are ever framed
    const int Np = (T + 3) * HOP;
    hipLaunchKernelGGL(pm_reflect_pad_kernel, dim3((Np + 255) / 256, B),
                       dim3(256), 0, s, audio, padded, N, pad, Np);
    HIP_TRY(hipGetLastError());
    SingleArgs a = {};
    a.x = padded; a.out = out; a.w = basis.forward; a.bias = nullptr;
    a.gbias = nullptr; a.gbias_batch = 1;
    a.B = B; a.L = T + 3; a.Lout = T; a.Cin = HOP; a.M = DFT_M;
    a.bins = BINS; a.maxbits = maxbits; a.grad = grad; a.lrelu = 0; a.pad = 0;
    a.phase_r = 0; a.phase_c = 1;
    HIP_TRY(pm_launch_stft(epi, a, s));
    return PM_OK;
}

// Brute-force cross-check of pm_stft_magnitude: the same spectrogram by the
// framed-DFT GEMM (exact-fp32 MFMA), independent of the FFT code path.
extern "C" int pm_stft_magnitude_dft(
    const float* audio, float* out, int B, int N, void* scratch,
    size_t scratch_bytes, void* stream) {
    return stft_launch(1, audio, out, nullptr, B, N, scratch, scratch_bytes,
                       (hipStream_t)stream);
}

// ---- FFT path (forward transforms; pm_fft.h) --------------------------------
static std::map<int, float*> g_fft_tables;   // per device (g_basis_mutex)
// (per host thread, like the other test / tuning hooks: a setter on one thread
// cannot change the launch geometry of a call in flight on another)
static thread_local int g_fft_frames_per_group = 16;

static int get_fft_tables(const float** out, hipStream_t s) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_basis_mutex);
    auto it = g_fft_tables.find(dev);
    if (it != g_fft_tables.end()) { *out = it->second; return PM_OK; }
    std::vector<float> h(PM_FFT_TAB_FLOATS);
    const double pi2 = 6.283185307179586476925286766559;
    // periodic hann (torch.hann_window(1024), spectrogram.py:29)
    for (int n = 0; n < NFFT; ++n)
        h[n] = (float)(0.5 - 0.5 * cos(pi2 * n / NFFT));
    for (int j = 0; j < 512; ++j) {
        h[PM_FFT_TAB_W512 + 2 * j] = (float)cos(pi2 * j / 512.0);
        h[PM_FFT_TAB_W512 + 2 * j + 1] = (float)-sin(pi2 * j / 512.0);
    }
    for (int k = 0; k <= 512; ++k) {
        h[PM_FFT_TAB_W1024 + 2 * k] = (float)cos(pi2 * k / 1024.0);
        h[PM_FFT_TAB_W1024 + 2 * k + 1] = (float)-sin(pi2 * k / 1024.0);
    }
    float* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, h.size() * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(d, h.data(), h.size() * sizeof(float),
                           hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    g_fft_tables[dev] = d;
    *out = d;
    return PM_OK;
}

// 2 (default): both transforms for every frame; 1: the 8-band loudness runs the
// OPTIMISTIC first pass and transforms a group a second time only where the
// floor bites. Opt-in because it depends on the material: noise at a steady
// level 56 -> 47 us (batch 32 x 10 s), but a group with ANY bin more than 80 dB
// under its utterance's maximum is transformed twice by a first pass that costs
// what the second does (36 + 34 us against 27 + 34 when every group is) - and
// 16 frames x 513 bins of recorded speech usually hold such a bin.
static thread_local int g_loudness_passes = 2;
extern "C" int pm_stft_set_loudness_passes(int passes) {
    if (passes != 1 && passes != 2)
        return fail(PM_EINVAL, "loudness passes: 1 (optimistic) or 2");
    g_loudness_passes = passes;
    return PM_OK;
}

extern "C" int pm_stft_set_frames_per_group(int frames) {
    if (frames != 16 && frames != 32)
        return fail(PM_EINVAL, "frames per workgroup must be 16 or 32");
    g_fft_frames_per_group = frames;
    return PM_OK;
}

// One geometry of the FFT kernel as PERSISTENT workgroups: the grid is what
// the device holds at once (occupancy x CUs, asked once per geometry) and a
// workgroup walks the (utterance, group of NW x FPW frames) pairs with that
// stride.
// `dry` (pm_stft_launch_info): fill a.groups / a.total / a.grid, launch nothing
template <int EPI, int NW, int FPW>
static int fft_launch_shape(FftArgs& a, hipStream_t s, int* dry_grid = nullptr) {
    auto kern = pm_stft_fft_kernel<EPI, NW, FPW>;
    constexpr int smem = pm_fft_smem_bytes<EPI, NW, FPW>();
    HIP_TRY(pm_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem));
    static std::atomic<int> per_cu{0};
    int resident = per_cu.load(std::memory_order_relaxed);
    if (resident == 0) {
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(
            &resident, kern, NW * 64, smem));
        if (resident < 1) resident = 1;
        per_cu.store(resident, std::memory_order_relaxed);
    }
    a.groups = (a.T + NW * FPW - 1) / (NW * FPW);
    const long long total = (long long)a.groups * a.B;
    if (total > 0x7fffffffLL) return fail(PM_EINVAL, "batch too large");
    a.total = (int)total;
    const int cus = pm_device_cus() > 0 ? pm_device_cus() : 256;
    const int grid = (int)std::min<long long>(total, (long long)resident * cus);
    if (dry_grid) { *dry_grid = grid; return PM_OK; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), smem, s, a);
    HIP_TRY(hipGetLastError());
    return PM_OK;
}

// `frames_per_group`: read ONCE per API call by the caller (pm_loudness runs
// two passes whose per-group maxima must be indexed the same way)
template <int EPI>
static int fft_launch(FftArgs& a, hipStream_t s,
                      int frames_per_group = g_fft_frames_per_group,
                      int* dry_grid = nullptr) {
    const int pad = (NFFT - HOP) / 2;
    if (!a.audio && !dry_grid) return fail(PM_EINVAL, "null argument");
    if (a.B < 1 || a.N <= pad)
        return fail(PM_EINVAL, "need more than %d samples (reflect pad)", pad);
    a.T = a.N / HOP;
    if (a.T < 1) return fail(PM_EINVAL, "fewer samples than one hop");
    if (a.B > 65535) return fail(PM_EINVAL, "batch too large (max 65535)");
    if (!dry_grid) {
        int rc = get_fft_tables(&a.tables, s);
        if (rc) return rc;
    }
    // 16 frames by EIGHT waves of two frames for the magnitude / log-mel
    // launches (two 512-thread workgroups per CU: their 513 x 17 staging tiles
    // fill the LDS), four waves of four frames for the loudness passes, which
    // stage 8 rows or nothing and keep four workgroups per CU resident (round 5:
    // every FFT kernel fits 128 registers; measured shapes and their history:
    // profiles/r04/stft_pmc.txt, profiles/r05/ab_fft_packed.txt)
    if (frames_per_group == 32)
        return fft_launch_shape<EPI, 8, 4>(a, s, dry_grid);
#ifndef PM_FFT_LOUD_8X2
#define PM_FFT_LOUD_8X2 0
#endif
    if constexpr (EPI == 1 || EPI == 4 || PM_FFT_LOUD_8X2)
        return fft_launch_shape<EPI, 8, 2>(a, s, dry_grid);
    else
        return fft_launch_shape<EPI, 4, 4>(a, s, dry_grid);
}

// The geometry the next launch of one FFT transform would take on this device
// and host thread (tests assert that the persistent multi-group walk - more
// groups than resident workgroups - is what they exercise). transform: 1
// magnitude, 4 log-mel, 2 / 3 / 5 / 6 the loudness passes (maximum, generic
// bands, the 8 default bands, their optimistic first pass).
extern "C" int pm_stft_launch_info(
    int transform, int B, int N, int* total_groups, int* workgroups) {
    if (!total_groups || !workgroups) return fail(PM_EINVAL, "null argument");
    FftArgs a = {};
    a.B = B; a.N = N;
    int grid = 0, rc;
    switch (transform) {
        case 1: rc = fft_launch<1>(a, nullptr, g_fft_frames_per_group, &grid); break;
        case 2: rc = fft_launch<2>(a, nullptr, g_fft_frames_per_group, &grid); break;
        case 3: rc = fft_launch<3>(a, nullptr, g_fft_frames_per_group, &grid); break;
        case 4: rc = fft_launch<4>(a, nullptr, g_fft_frames_per_group, &grid); break;
        case 5: rc = fft_launch<5>(a, nullptr, g_fft_frames_per_group, &grid); break;
        case 6: rc = fft_launch<6>(a, nullptr, g_fft_frames_per_group, &grid); break;
        default: return fail(PM_EINVAL, "transform must be 1..6");
    }
    if (rc) return rc;
    *total_groups = a.total;
    *workgroups = grid;
    return PM_OK;
}

extern "C" int pm_stft_magnitude(
    const float* audio, float* out, int B, int N, void* scratch,
    size_t scratch_bytes, void* stream) {
    (void)scratch; (void)scratch_bytes;   // (the FFT path needs none)
    if (!out) return fail(PM_EINVAL, "null argument");
    FftArgs a = {};
    a.audio = audio; a.out = out; a.B = B; a.N = N;
    return fft_launch<1>(a, (hipStream_t)stream);
}

// spectrogram.from_audio(audio, mels=True): the log-mel spectrogram straight
// from the FFT workgroup's LDS tile (the (B, 513, T) magnitudes never reach
// HBM). basis (mels, 513) -> pm_stft_mel_prepare -> `prepared`
// (pm_stft_mel_scratch_bytes(mels) bytes, reusable)
extern "C" size_t pm_stft_mel_scratch_bytes(int mels) {
    return mels < 1 ? 0 : align256((size_t)(3 * mels + 1) * sizeof(int)) +
                              align256((size_t)mels * BINS * sizeof(float));
}

// Compact the (mels, 513) filterbank once (per basis): `prepared` then feeds
// any number of pm_stft_mel calls.
extern "C" int pm_stft_mel_prepare(
    const float* basis, int mels, void* prepared, size_t prepared_bytes,
    void* stream) {
    if (!basis || !prepared) return fail(PM_EINVAL, "null argument");
    if (mels < 1 || mels > 1024)
        return fail(PM_EINVAL, "1..1024 mel filters");
    if (prepared_bytes < pm_stft_mel_scratch_bytes(mels))
        return fail(PM_ENOMEM, "buffer too small");
    int* table = (int*)prepared;
    float* vals = (float*)((char*)prepared +
                           align256((size_t)(3 * mels + 1) * sizeof(int)));
    hipLaunchKernelGGL(pm_mel_csr_kernel, dim3(1), dim3(256), 0,
                       (hipStream_t)stream, basis, table, vals, mels, BINS);
    HIP_TRY(hipGetLastError());
    return PM_OK;
}

extern "C" int pm_stft_mel(
    const float* audio, const void* prepared, float* out, int B, int N,
    int mels, int use_threshold, float log_threshold, void* stream) {
    if (!prepared || !out) return fail(PM_EINVAL, "null argument");
    if (mels < 1 || mels > 1024)
        return fail(PM_EINVAL, "1..1024 mel filters");
    hipStream_t s = (hipStream_t)stream;
    const int* table = (const int*)prepared;
    const float* vals = (const float*)((const char*)prepared +
                           align256((size_t)(3 * mels + 1) * sizeof(int)));
    FftArgs a = {};
    a.audio = audio; a.out = out; a.B = B; a.N = N;
    a.mel_span = table; a.mel_vals = vals; a.rows = mels;
    a.use_thr = use_threshold; a.thr = log_threshold;
    return fft_launch<4>(a, s);
}

// Backward of pm_stft_magnitude (the training mel loss differentiates through
// spectrogram.from_audio: promonet/train/core.py:277-305). Two exact-fp32 MFMA
// convs: the framed DFT again, its epilogue turning the incoming gradient into
// the DFT cotangent grad / |X| * (re, im); then the overlap-add of that
// cotangent against the transposed basis; then the adjoint of the reflect pad.
extern "C" size_t pm_stft_backward_scratch_bytes(int B, int N) {
    if (B < 1 || N < HOP) return 0;
    const size_t T = N / HOP;
    return pm_stft_scratch_bytes(B, N) +
           align256((size_t)B * T * DFT_M * sizeof(float)) +
           align256((size_t)B * (T + 3) * HOP * sizeof(float));
}

extern "C" int pm_stft_magnitude_backward(
    const float* audio, const float* grad_out, float* grad_audio, int B, int N,
    void* scratch, size_t scratch_bytes, void* stream) {
    if (!audio || !grad_out || !grad_audio || !scratch)
        return fail(PM_EINVAL, "null argument");
    if (B < 1 || N < HOP || scratch_bytes < pm_stft_backward_scratch_bytes(B, N))
        return fail(PM_ENOMEM, "scratch too small");
    hipStream_t s = (hipStream_t)stream;
    const int T = N / HOP;
    const int pad = (NFFT - HOP) / 2;
    char* base = (char*)scratch;
    const size_t stft_bytes = pm_stft_scratch_bytes(B, N);
    float* cot = (float*)(base + stft_bytes);               // (B, T, 1088)
    float* gpad = (float*)(base + stft_bytes +
                           align256((size_t)B * T * DFT_M * sizeof(float)));
    int rc = stft_launch(3, audio, cot, nullptr, B, N, base, stft_bytes, s,
                         grad_out);
    if (rc) return rc;
    DftBasis basis;
    rc = get_dft_basis(&basis, s);
    if (rc) return rc;
    SingleArgs a = {};
    a.x = cot; a.out = gpad; a.w = basis.backward; a.bias = basis.zeros;
    a.gbias = nullptr; a.gbias_batch = 1;
    a.B = B; a.L = T; a.Lout = T + 3; a.Cin = DFT_M; a.M = HOP;
    a.lrelu = 0; a.pad = 3; a.phase_r = 0; a.phase_c = 1;
    HIP_TRY(pm_launch_stft(0, a, s));
    const int Np = (T + 3) * HOP;
    hipLaunchKernelGGL(pm_reflect_pad_adjoint_kernel,
                       dim3((N + 255) / 256, B), dim3(256), 0, s, gpad,
                       grad_audio, N, pad, Np);
    HIP_TRY(hipGetLastError());
    return PM_OK;
}

// Backward of pm_linear_to_mel: grad_mel (B, M, T) -> grad_spec (B, F, T);
// scratch holds B * M * T floats.
extern "C" int pm_linear_to_mel_backward(
    const float* spec, const float* basis, const float* grad_mel,
    float* grad_spec, float* scratch, int B, int F, int M, int T,
    int use_threshold, float log_threshold, void* stream) {
    if (!spec || !basis || !grad_mel || !grad_spec || !scratch)
        return fail(PM_EINVAL, "null argument");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(pm_mel_backward_rows_kernel,
                       dim3((T + 255) / 256, M, B), dim3(256), 0, s, spec,
                       basis, grad_mel, scratch, F, M, T, use_threshold,
                       log_threshold);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(pm_mel_backward_cols_kernel,
                       dim3((T + 255) / 256, F, B), dim3(256), 0, s, basis,
                       scratch, grad_spec, F, M, T);
    HIP_TRY(hipGetLastError());
    return PM_OK;
}

extern "C" int pm_linear_to_mel(
    const float* spec, const float* basis, float* out, int B, int F, int M,
    int T, int use_threshold, float log_threshold, void* stream) {
    if (!spec || !basis || !out) return fail(PM_EINVAL, "null argument");
    hipLaunchKernelGGL(pm_mel_kernel, dim3((T + 255) / 256, M, B), dim3(256),
                       0, (hipStream_t)stream, spec, basis, out, F, M, T,
                       use_threshold, log_threshold);
    HIP_TRY(hipGetLastError());
    return PM_OK;
}

extern "C" size_t pm_loudness_scratch_bytes(int B, int N) {
    if (B < 1 || N < HOP) return 0;
    // one maximum and one minimum per FFT workgroup (>= 16 frames each) and
    // utterance
    const size_t groups = ((size_t)(N / HOP) + 15) / 16;
    return 2 * align256((size_t)B * groups * sizeof(float));
}

// Two passes over the audio (4 B / sample each) instead of a (B, 513, T) dB
// tensor written and re-read: pass 1 finds every utterance's maximum dB
// (librosa.amplitude_to_db's top_db reference, loudness.py:46), pass 2 repeats
// the FFT and writes the floored, A-weighted band means.
// With pm_stft_set_loudness_passes(1) the default 8 bands run OPTIMISTICALLY:
// pass 1 (EPI 6) already writes the band means, without a floor, and records
// every 16-frame group's minimum dB next to its maximum; pass 2 (EPI 5)
// transforms only the groups that have a bin under their utterance's floor -
// for the others max(v, floor) == v and pass 1's means are final, bit for bit
// (tests/test_gpu_preprocess_full.py). Off by default: see g_loudness_passes.
extern "C" int pm_loudness(
    const float* audio, const float* a_weights, float* out, int B, int N,
    int bands, float min_db, void* scratch, size_t scratch_bytes,
    void* stream) {
    if (!audio || !a_weights || !out || !scratch)
        return fail(PM_EINVAL, "null argument");
    if (bands < 1 || (bands > 16 && bands != BINS))
        return fail(PM_EINVAL, "bands must be 1..16 or 513 (no averaging)");
    if (B < 1 || N < HOP || scratch_bytes < pm_loudness_scratch_bytes(B, N))
        return fail(PM_ENOMEM, "scratch too small");
    hipStream_t s = (hipStream_t)stream;
    FftArgs a = {};
    a.audio = audio; a.out = out; a.B = B; a.N = N;
    a.group_max = (float*)scratch;
    const int frames_per_group = g_fft_frames_per_group;   // both passes
    a.weights = a_weights; a.rows = bands;
    const double step = (double)BINS / (double)bands;   // loudness.py:96
    for (int b = 0; b <= bands && b <= 16; ++b)
        a.band_start[b] = (int)(b * step);
    if (bands == 1) { a.band_start[0] = 0; a.band_start[1] = BINS; }
    a.min_db = min_db; a.top_db = 80.f;
    // the default 8 bands: band j = bins 64 j .. 64 j + 63 (+ bin 512 in the
    // last), reduced across the wave out of the registers (EPI 5)
    bool aligned8 = bands == 8;
    for (int b = 0; aligned8 && b < 8; ++b) aligned8 = a.band_start[b] == 64 * b;
#ifdef PM_LOUD_NO_EPI5
    aligned8 = false;
#endif
    aligned8 = aligned8 && a.band_start[8] == BINS;
    if (aligned8 && g_loudness_passes == 1) {
        a.group_min = (float*)((char*)scratch +
                               pm_loudness_scratch_bytes(B, N) / 2);
        int rc = fft_launch<6>(a, s, frames_per_group);
        if (rc) return rc;
        return fft_launch<5>(a, s, frames_per_group);
    }
    int rc = fft_launch<2>(a, s, frames_per_group);
    if (rc) return rc;
    if (aligned8) return fft_launch<5>(a, s, frames_per_group);
    return fft_launch<3>(a, s, frames_per_group);
}

// promonet.edit feature editing (edit/core.py:17-132, edit/grid.py:12-45)
extern "C" int pm_grid_sample(
    const float* seq, const float* grid, float* out, int rows, int n_in,
    int n_out, int mode, float scale, float offset, float lo, float hi,
    void* stream) {
    if (!seq || !out) return fail(PM_EINVAL, "null argument");
    if (rows < 1 || n_in < 1 || n_out < 1 || mode < 0 || mode > 2)
        return fail(PM_EINVAL, "bad grid-sample arguments");
    if (rows > 65535) return fail(PM_EINVAL, "too many rows (max 65535)");
    hipLaunchKernelGGL(pm_grid_sample_kernel, dim3((n_out + 255) / 256, rows),
                       dim3(256), 0, (hipStream_t)stream, seq, grid, out, rows,
                       n_in, n_out, mode, scale, offset, lo, hi);
    HIP_TRY(hipGetLastError());
    return PM_OK;
}

// Selective time-stretch grid (edit/core.py:57-110)
extern "C" int pm_stretch_grid(
    const float* ppg, int ppg_rows, const int* indices, int n_indices,
    float* selected, float* grid, int frames, int target_frames,
    void* stream) {
    if (!ppg || !indices || !selected || !grid)
        return fail(PM_EINVAL, "null argument");
    if (ppg_rows < 1 || n_indices < 1 || frames < 1 || target_frames < 1)
        return fail(PM_EINVAL, "bad stretch-grid arguments");
    StretchArgs a;
    a.ppg = ppg; a.indices = indices; a.selected = selected; a.grid = grid;
    a.n = n_indices; a.T = frames; a.target = target_frames; a.P = ppg_rows;
    const size_t bytes = (size_t)frames * sizeof(float);
    const size_t smem = bytes <= 64 * 1024 ? bytes : 0;
    auto kern = pm_stretch_grid_kernel;
    if (smem > 48 * 1024)
        HIP_TRY(pm_ensure_dynamic_lds(reinterpret_cast<const void*>(kern),
                                      (int)smem));
    hipLaunchKernelGGL(kern, dim3(1), dim3(256), smem, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return PM_OK;
}

// ---------------------------------------------------------------------------
// FARGAN engine (config/fargan.py): replaces promonet.model.FARGAN
// ---------------------------------------------------------------------------
struct FLayer {
    const char* key;     // state-dict prefix (without .weight / .weight_g ...)
    const char* leaf;    // leaf of a plain tensor ("weight", "weight_ih", ...)
    bool normed;         // weight-normed Linear: accepts weight_g + weight_v
    int rows, cols, rpad, kpad;
    int kw = 0;              // > 0: also packed K-split, 8 x (rpad x kw)
    bool insensitive = false;  // stored f16 under PM_FARGAN_MIXED (GRU, gates)
    size_t offset = 0;       // element offsets into the one weight buffer
    size_t offset_k = 0;
    float* tmp_g = nullptr;
    float* tmp_v = nullptr;
    bool has = false;
};

struct pm_fargan_s {
    void* weights = nullptr;   // every packed layer (FarganWeights layout)
    void* weights_i = nullptr; // PM_FARGAN_MIXED: the f16-stored layers (same
                               // element offsets; only their regions are used)
    int nfeat, G, dtype;
    int mode = 0;        // 0 auto, 1 one workgroup per utterance, 2 clusters
    std::vector<FLayer> layers;
    bool finalized = false;
};

#define FG_P "subframe_network."
static std::vector<FLayer> fargan_layers(int nin) {
    const int cpad = 376;
    std::vector<FLayer> l = {
        {"conditioning_network.0", "weight", false, nin, nin, 384, cpad},
        {"conditioning_network.2", "weight", false, nin, nin, 384, cpad},
        {"conditioning_network.4", "weight", false, 512, nin, 512, cpad},
        {FG_P "framewise_convolution.model.0", "weight", true, 256, 520, 256, 520},
        {FG_P "framewise_convolution.model.2.gate", "weight", true, 256, 256, 256, 256},
        {FG_P "gru1", "weight_ih", false, 768, 384, 768, 384},
        {FG_P "gru2", "weight_ih", false, 768, 384, 768, 384},
        {FG_P "gru3", "weight_ih", false, 768, 384, 768, 384},
        {FG_P "gru1", "weight_hh", false, 768, 256, 768, 256},
        {FG_P "gru2", "weight_hh", false, 768, 256, 768, 256},
        {FG_P "gru3", "weight_hh", false, 768, 256, 768, 256},
        {FG_P "gru1_glu.gate", "weight", true, 256, 256, 256, 256},
        {FG_P "gru2_glu.gate", "weight", true, 256, 256, 256, 256},
        {FG_P "gru3_glu.gate", "weight", true, 256, 256, 256, 256},
        {FG_P "skip_dense", "weight", false, 256, 1152, 256, 1152},
        {FG_P "skip_glu.gate", "weight", true, 256, 256, 256, 256},
        {FG_P "output_layer", "weight", false, 64, 256, 64, 256},
    };
    // layers that contract a member-owned slice in the cluster kernel
    l[1].kw = 48;                                   // conditioning_network.2
    l[4].kw = 32;                                   // framewise conv GLU gate
    l[11].kw = l[12].kw = l[13].kw = 32;            // GRU GLU gates
    l[16].kw = 32;                                  // output layer
    // FgTypes<FgMixed>::I (pm_fargan.h): the GRU cells and the GLU gates
    for (int i : {4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15}) l[i].insensitive = true;
    // offsets = FarganWeights<>: row-packed layers in table order, then the
    // K-split copies
    size_t at = 0;
    for (auto& layer : l) {
        layer.offset = at;
        at += (size_t)layer.rpad * layer.kpad;
    }
    for (auto& layer : l)
        if (layer.kw) {
            layer.offset_k = at;
            at += (size_t)FG_G * layer.rpad * layer.kw;
        }
    return l;
}

extern "C" int pm_fargan_create(
    int num_features, int global_channels, int weight_dtype,
    pm_fargan_t* out) {
    if (!out) return fail(PM_EINVAL, "null argument");
    if (num_features + global_channels != 371 || num_features < 1)
        return fail(PM_EINVAL,
                    "FARGAN kernel is built for 113 + 258 conditioning "
                    "channels (config/fargan.py)");
    if (weight_dtype != PM_F32 && weight_dtype != PM_F16 &&
        weight_dtype != PM_FARGAN_MIXED)
        return fail(PM_EINVAL,
                    "weight dtype must be PM_F32, PM_F16 or PM_FARGAN_MIXED");
    auto* h = new pm_fargan_s();
    h->nfeat = num_features; h->G = global_channels; h->dtype = weight_dtype;
    h->layers = fargan_layers(num_features + global_channels);
    typedef FarganWeights<float> W;
    const auto& l = h->layers;
    if (l[0].offset != W::COND0 || l[1].offset != W::COND1 ||
        l[2].offset != W::COND2 || l[3].offset != W::FWCONV ||
        l[4].offset != W::FWGLU || l[5].offset != W::GRU_IH ||
        l[8].offset != W::GRU_HH || l[11].offset != W::GRU_GLU ||
        l[14].offset != W::SKIP || l[15].offset != W::SKIP_GLU ||
        l[16].offset != W::OUT || l[1].offset_k != W::K_COND1 ||
        l[4].offset_k != W::K_FWGLU || l[11].offset_k != W::K_GRU_GLU ||
        l[16].offset_k != W::K_OUT) {
        delete h;
        return fail(PM_ESTATE, "FARGAN layer table and FarganWeights disagree");
    }
    *out = h;      // the weight buffer is allocated with the first tensor
    return PM_OK;
}

extern "C" int pm_fargan_destroy(pm_fargan_t h) {
    if (!h) return PM_OK;
    if (h->weights) hipFree(h->weights);
    if (h->weights_i) hipFree(h->weights_i);
    for (auto& l : h->layers) {
        if (l.tmp_g) hipFree(l.tmp_g);
        if (l.tmp_v) hipFree(l.tmp_v);
    }
    delete h;
    return PM_OK;
}

template <class WT>
static hipError_t fargan_pack_t(
    void* buffer, FLayer& l, const float* w, hipStream_t s) {
    const int gru = l.rows == 768 ? 1 : 0;   // gate-interleaved GRU rows
    const size_t elems = (size_t)l.rpad * l.kpad;
    WT* base = (WT*)buffer;
    hipLaunchKernelGGL(pm_fargan_pack_kernel<WT>,
                       dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, s,
                       w, base + l.offset, l.rows, l.cols, l.rpad, l.kpad, gru,
                       0);
    if (l.kw) {
        // K-split copy: member g's sub-matrix W[:, g kw : (g + 1) kw]
        const size_t sub = (size_t)l.rpad * l.kw;
        for (int g = 0; g < FG_G; ++g)
            hipLaunchKernelGGL(pm_fargan_pack_kernel<WT>,
                               dim3((unsigned)((sub + 255) / 256)), dim3(256),
                               0, s, w, base + l.offset_k + g * sub, l.rows,
                               l.cols, l.rpad, l.kw, 0, g * l.kw);
    }
    return hipGetLastError();
}

static int fargan_pack(pm_fargan_t h, FLayer& l, const float* w, hipStream_t s) {
    if (!h->weights)
        HIP_TRY(hipMalloc(&h->weights, FarganWeights<float>::TOTAL *
                                           (h->dtype == PM_F16 ? 2 : 4)));
    if (h->dtype == PM_FARGAN_MIXED && !h->weights_i)
        HIP_TRY(hipMalloc(&h->weights_i, FarganWeights<float>::TOTAL * 2));
    if (h->dtype == PM_FARGAN_MIXED && l.insensitive)
        HIP_TRY(fargan_pack_t<_Float16>(h->weights_i, l, w, s));
    else if (h->dtype == PM_F16)
        HIP_TRY(fargan_pack_t<_Float16>(h->weights, l, w, s));
    else
        HIP_TRY(fargan_pack_t<float>(h->weights, l, w, s));
    l.has = true;
    return PM_OK;
}

extern "C" int pm_fargan_load_tensor(
    pm_fargan_t h, const char* name, const float* dev, const int64_t* shape,
    int ndim, void* stream) {
    if (!h || !name || !dev || !shape) return fail(PM_EINVAL, "null argument");
    hipStream_t s = (hipStream_t)stream;
    for (auto& l : h->layers) {
        const size_t n = strlen(l.key);
        if (strncmp(name, l.key, n) || name[n] != '.') continue;
        const char* leaf = name + n + 1;
        if (!strcmp(leaf, l.leaf)) {
            if (ndim != 2 || shape[0] != l.rows || shape[1] != l.cols)
                return fail(PM_EINVAL, "%s: expected shape (%d, %d)", name,
                            l.rows, l.cols);
            int rc = fargan_pack(h, l, dev, s);
            if (rc) return rc;
        } else if (l.normed && (!strcmp(leaf, "weight_g") ||
                                !strcmp(leaf, "weight_v"))) {
            const bool is_g = leaf[7] == 'g';
            if (is_g) {
                if (ndim != 2 || shape[0] != l.rows || shape[1] != 1)
                    return fail(PM_EINVAL, "%s: expected (%d, 1)", name, l.rows);
                int rc = copy_dev(&l.tmp_g, dev, l.rows, s);
                if (rc) return rc;
            } else {
                if (ndim != 2 || shape[0] != l.rows || shape[1] != l.cols)
                    return fail(PM_EINVAL, "%s: expected (%d, %d)", name,
                                l.rows, l.cols);
                int rc = copy_dev(&l.tmp_v, dev, (size_t)l.rows * l.cols, s);
                if (rc) return rc;
            }
            if (l.tmp_g && l.tmp_v) {
                float* folded = nullptr;
                HIP_TRY(hipMalloc((void**)&folded,
                                  (size_t)l.rows * l.cols * sizeof(float)));
                hipLaunchKernelGGL(pm_fold_kernel, dim3(l.rows), dim3(256), 0,
                                   s, l.tmp_g, l.tmp_v, folded, l.cols);
                HIP_TRY(hipGetLastError());
                int rc = fargan_pack(h, l, folded, s);
                HIP_TRY(hipStreamSynchronize(s));
                hipFree(folded); hipFree(l.tmp_g); hipFree(l.tmp_v);
                l.tmp_g = l.tmp_v = nullptr;
                if (rc) return rc;
            }
        } else {
            continue;   // e.g. gru1.weight_hh is a different table row
        }
        HIP_TRY(hipStreamSynchronize(s));
        h->finalized = false;
        return PM_OK;
    }
    return fail(PM_EINVAL, "%s: not a FARGAN state-dict key", name);
}

extern "C" int pm_fargan_finalize(pm_fargan_t h, void* stream) {
    if (!h) return fail(PM_EINVAL, "null handle");
    for (auto& l : h->layers)
        if (!l.has)
            return fail(PM_ESTATE, "missing tensor: %s.%s", l.key, l.leaf);
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    h->finalized = true;
    return PM_OK;
}

// Kernel choice. Measured (MI355X, 10 s utterances, fp32 weights): clusters of
// 8 workgroups take 112 ms for 32 utterances (one per cluster), 162 ms for 64
// (two in lockstep per cluster), 242 ms for 128 and 485 ms for 256 (four in
// lockstep, two waves); one workgroup per utterance takes 721 ms per wave of
// 256 -> clusters at every batch size, PROVIDED every workgroup of the grid is
// resident at once: the members of a cluster wait for each other's granules.
// The grid is therefore sized from the device: one 768-thread workgroup per CU
// (its LDS / wave budget admits at least that on any gfx950 partition), i.e.
// at most multiProcessorCount / 8 clusters; a device with fewer than 8 CUs
// visible gets the one-workgroup-per-utterance kernel. What the query cannot
// see (another process sharing the GPU, a CU mask) is caught by the bounded
// spins: pm_fargan_check reports the timeout and the caller re-runs with
// pm_fargan_set_mode(h, 1). pm_fargan_set_mode() overrides the choice
// (a -DPM_TUNING build also reads PM_FARGAN=single|cluster).
static const int FG_MAX_CLUSTERS = 32;   // 32 x 8 workgroups = one per CU

static int fargan_resident_clusters() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount,
                              dev) != hipSuccess)
        return 0;
    const int n = cus / FG_G;
    return n < FG_MAX_CLUSTERS ? n : FG_MAX_CLUSTERS;
}

static bool fargan_use_cluster(pm_fargan_t h, int B) {
    int mode = h->mode;
#ifdef PM_TUNING
    static const int forced = [] {
        const char* e = getenv("PM_FARGAN");
        return !e ? 0 : (!strcmp(e, "single") ? 1 : (!strcmp(e, "cluster") ? 2 : 0));
    }();
    if (!mode) mode = forced;
#endif
    (void)B;
    if (mode == 1) return false;
    return fargan_resident_clusters() >= 1;
}

extern "C" int pm_fargan_set_mode(pm_fargan_t h, int mode) {
    if (!h || mode < 0 || mode > 2) return fail(PM_EINVAL, "bad mode");
    h->mode = mode;
    return PM_OK;
}

static size_t fargan_state_bytes() {
    return align256((size_t)FG_MAX_CLUSTERS * FG_CSTATE * 4 + 256);
}

// (B * T, 512) conditioning vectors of pm_fargan_cond_kernel: fp32-stored
// conditioning weights only (f16 storage keeps them inside the cluster kernel)
static size_t fargan_precond_bytes(pm_fargan_t h, int B, int T) {
    return h->dtype == PM_F16 ? 0 : align256((size_t)B * T * 512 * sizeof(float));
}

extern "C" size_t pm_fargan_workspace_bytes(pm_fargan_t h, int B, int T) {
    if (!h || B < 1 || T < 1) return 0;
    return align256((size_t)B * T * pad32(h->nfeat + 1) * sizeof(float)) +
           fargan_state_bytes() + fargan_precond_bytes(h, B, T);
}

template <class WT>
static int fargan_launch(
    pm_fargan_t h, const FarganArgs& a, hipStream_t s, void* cluster_state) {
    FarganWeights<WT> w;
    w.base = (const typename FarganWeights<WT>::S*)h->weights;
    w.base_i = (const typename FarganWeights<WT>::I*)(
        h->weights_i ? h->weights_i : h->weights);
    if (cluster_state) {
        // counters / payload / error word are re-initialised on every call
        HIP_TRY(hipMemsetAsync(cluster_state, 0, fargan_state_bytes(), s));
        FarganClusterArgs ca;
        ca.f = a;
        ca.state = (unsigned*)cluster_state;
        ca.error = ca.state + (size_t)FG_MAX_CLUSTERS * FG_CSTATE;
        ca.precond = nullptr;
        if constexpr (std::is_same<typename FarganWeights<WT>::S, float>::value) {
            // the conditioning network of every frame, ahead of the walk
            FarganCondArgs cn;
            cn.features_cl = a.features_cl; cn.global = a.global;
            cn.cond = (float*)((char*)cluster_state + fargan_state_bytes());
            cn.B = a.B; cn.T = a.T; cn.cstride = a.cstride; cn.nfeat = a.nfeat;
            cn.G = a.G; cn.global_batch = a.global_batch;
            const size_t cond_lds =
                (size_t)FG_CN * (FG_CPITCH + FG_OPITCH) * sizeof(float);
            hipError_t ce = pm_ensure_dynamic_lds(
                reinterpret_cast<const void*>(pm_fargan_cond_kernel),
                (int)cond_lds);
            HIP_TRY(ce);
            const long long frames = (long long)a.B * a.T;
            hipLaunchKernelGGL(
                pm_fargan_cond_kernel,
                dim3((unsigned)((frames + FG_CN - 1) / FG_CN)), dim3(256),
                cond_lds, s, cn, w.base + FarganWeights<WT>::COND0,
                w.base + FarganWeights<WT>::COND1,
                w.base + FarganWeights<WT>::COND2);
            HIP_TRY(hipGetLastError());
            ca.precond = cn.cond;
        }
#ifdef PM_TUNING
        ca.timeline = g_timeline;
#endif
        // U utterances per cluster in lockstep: 1 while one cluster per
        // utterance fits the resident grid (32 clusters = 256 CUs), then 2,
        // then 4; beyond that the clusters walk the batch in waves
        const int resident = fargan_resident_clusters();
        if (resident < 1)
            return fail(PM_ESTATE, "FARGAN cluster kernel needs >= %d CUs", FG_G);
        const int U = a.B <= resident ? 1 : a.B <= 2 * resident ? 2 : FG_UMAX;
        const int groups = (a.B + U - 1) / U;
        ca.nclusters = groups < resident ? groups : resident;
        const dim3 grid(ca.nclusters * FG_G), block(FG_CT);
        // (+ the LDS-resident short slices of a one-utterance cluster)
        const size_t smem = (size_t)U * sizeof(FgLds) +
                            (U == 1 ? FgResident<WT, 1>::BYTES : 0);
        auto launch = [&](auto kern) -> hipError_t {
            hipError_t e = pm_ensure_dynamic_lds(
                reinterpret_cast<const void*>(kern), (int)smem);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, grid, block, smem, s, ca, w);
            return hipGetLastError();
        };
        hipError_t e = U == 1 ? launch(pm_fargan_cluster_kernel<WT, 1>)
                     : U == 2 ? launch(pm_fargan_cluster_kernel<WT, 2>)
                              : launch(pm_fargan_cluster_kernel<WT, FG_UMAX>);
        HIP_TRY(e);
        return PM_OK;
    }
    hipLaunchKernelGGL(pm_fargan_kernel<WT>, dim3(a.B), dim3(FG_THREADS), 0, s,
                       a, w);
    HIP_TRY(hipGetLastError());
    return PM_OK;
}

// Synchronises `stream` and reports whether a cluster exchange of the last
// forward on `workspace` gave up (bounded spin): PM_OK or PM_EHIP.
extern "C" int pm_fargan_check(
    pm_fargan_t h, int B, int T, void* ws, void* stream) {
    if (!h || !ws) return fail(PM_EINVAL, "null argument");
    if (!fargan_use_cluster(h, B)) return PM_OK;
    unsigned flag = 0;
    const char* state = (const char*)ws +
        align256((size_t)B * T * pad32(h->nfeat + 1) * sizeof(float));
    HIP_TRY(hipMemcpyAsync(&flag, state + (size_t)FG_MAX_CLUSTERS * FG_CSTATE * 4,
                           4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if (flag) return fail(PM_ETIMEOUT, "FARGAN cluster exchange timed out");
    return PM_OK;
}

// FARGAN.forward (model/fargan.py:21-59): features (B, nfeat + 1, T) with the
// pitch period as last channel (or channels-last (B, T, pad32(nfeat + 1)) when
// features_cl != 0), global (Bg, G), previous (Bp, 512) or NULL -> (B, 1, 256 T)
static int fargan_forward_impl(
    pm_fargan_t h, const float* features, int features_cl, const float* g,
    int gbatch, const float* previous, int pbatch, const int* lengths,
    float* out, int B, int T, void* ws, size_t ws_bytes, void* stream) {
    if (!h || !features || !g || !out) return fail(PM_EINVAL, "null argument");
    if (!h->finalized) return fail(PM_ESTATE, "pm_fargan_finalize not called");
    if (B < 1 || T < 1) return fail(PM_EINVAL, "empty batch or sequence");
    if ((gbatch != 1 && gbatch != B) || (previous && pbatch != 1 && pbatch != B))
        return fail(PM_EINVAL, "broadcast batch must be 1 or batch");
    hipStream_t s = (hipStream_t)stream;
    const int cpad = pad32(h->nfeat + 1);
    const float* fcl = features;
    if (!ws || ws_bytes < pm_fargan_workspace_bytes(h, B, T))
        return fail(PM_ENOMEM, "workspace too small");
    void* cluster_state = fargan_use_cluster(h, B)
        ? (char*)ws + align256((size_t)B * T * pad32(h->nfeat + 1) * sizeof(float))
        : nullptr;
    if (!features_cl) {
        dim3 grid((T + 31) / 32, cpad / 32, B);
        hipLaunchKernelGGL(pm_to_channels_last_kernel, grid, dim3(256), 0, s,
                           features, (float*)ws, h->nfeat + 1, T, cpad);
        HIP_TRY(hipGetLastError());
        fcl = (const float*)ws;
    }
    FarganArgs a;
    a.features_cl = fcl; a.global = g; a.previous = previous; a.out = out;
    a.B = B; a.T = T; a.cstride = cpad; a.nfeat = h->nfeat; a.G = h->G;
    a.global_batch = gbatch; a.previous_batch = pbatch;
    a.lengths = lengths;
    return h->dtype == PM_F32 ? fargan_launch<float>(h, a, s, cluster_state)
         : h->dtype == PM_F16 ? fargan_launch<_Float16>(h, a, s, cluster_state)
                              : fargan_launch<FgMixed>(h, a, s, cluster_state);
}

extern "C" int pm_fargan_forward(
    pm_fargan_t h, const float* features, int features_cl, const float* g,
    int gbatch, const float* previous, int pbatch, float* out, int B, int T,
    void* ws, size_t ws_bytes, void* stream) {
    return fargan_forward_impl(h, features, features_cl, g, gbatch, previous,
                               pbatch, nullptr, out, B, T, ws, ws_bytes, stream);
}

// Ragged batch: utterance b is lengths[b] <= T frames long inside the padded
// tensors. FARGAN is causal (frame t reads features <= t only), so the valid
// prefix equals the stand-alone synthesis bit for bit; the tail is zeros.
extern "C" int pm_fargan_forward_ragged(
    pm_fargan_t h, const float* features, int features_cl, const float* g,
    int gbatch, const float* previous, int pbatch, const int* lengths,
    float* out, int B, int T, void* ws, size_t ws_bytes, void* stream) {
    if (!lengths) return fail(PM_EINVAL, "null lengths");
    return fargan_forward_impl(h, features, features_cl, g, gbatch, previous,
                               pbatch, lengths, out, B, T, ws, ws_bytes, stream);
}
