// Streaming (HBM-bound) kernels around the MFMA convolutions: weight
// folding / packing (load time), conditioning-feature preparation, speaker
// bias, output conv + tanh. Reference: promonet/model/generator.py,
// promonet/model/hifigan.py.
#pragma once
#include "pm_common.h"

// ---------------------------------------------------------------------------
// weight_norm fold: w[r, :] = v[r, :] * g[r] / ||v[r, :]||   (dim = 0)
// torch.nn.utils.weight_norm as used at model/core.py:43-45 and
// model/hifigan.py:100-106. One workgroup per row.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pm_fold_kernel(
    const float* __restrict__ g, const float* __restrict__ v,
    float* __restrict__ w, int cols) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    const float* vr = v + (size_t)row * cols;
    float s = 0.f;
    for (int i = threadIdx.x; i < cols; i += 256) s += vr[i] * vr[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float norm = sqrtf(red[0] + red[1] + red[2] + red[3]);
    const float scale = g[row] / norm;
    for (int i = threadIdx.x; i < cols; i += 256)
        w[(size_t)row * cols + i] = vr[i] * scale;
}

// ---------------------------------------------------------------------------
// Pack torch-layout fp32 weights into the MFMA A-fragment stream of
// pm_conv.h. One thread per packed element.
//   mode 0: Conv1d           w[co][ci][k]   -> taps jj = 0..KT-1
//   mode 1: ConvTranspose1d  w[ci][co][k], stride r, pad p = (k - r) / 2,
//           M row m = ph * cout_pad + co; two taps per phase:
//           s = ph + p; window start js = (s >= r); tap jt = js + jj reads
//           input q - 1 + jt with kernel index (1 - jt) * r + s
// ---------------------------------------------------------------------------
struct PackArgs {
    const float* w;
    void* out;
    int mode;
    int cout, cin, k;          // actual (unpadded) torch dims
    int cout_pad, cin_pad;     // padded
    int mtiles, nch, ch, kt;   // packed geometry (kc = ch / 16)
    int r, p;                  // ConvTranspose stride / padding
    int bias_step;             // 1: each M tile's stream ends with a bias step
    long long total;           // packed elements (weights, without bias steps)
};

template <class ET>
__global__ __launch_bounds__(256) void pm_pack_kernel(PackArgs a) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.total) return;
    const int kc_n = a.ch / 16;
    long long rem = idx;
    const int e = rem % 8; rem /= 8;
    const int lane = rem % 64; rem /= 64;
    const int kc = rem % kc_n; rem /= kc_n;
    const int jj = rem % a.kt; rem /= a.kt;
    const int c = rem % a.nch; rem /= a.nch;
    const int mt = (int)rem;
    const int m = mt * 32 + (lane & 31);
    const int ci = c * a.ch + kc * 16 + (lane >> 5) * 8 + e;
    float v = 0.f;
    if (a.mode == 0) {
        if (m < a.cout && ci < a.cin)
            v = a.w[((size_t)m * a.cin + ci) * a.k + jj];
    } else {
        const int ph = m / a.cout_pad, co = m % a.cout_pad;
        const int s = ph + a.p;
        const int js = s >= a.r ? 1 : 0;
        const int jt = js + jj;
        const int jw = (1 - jt) * a.r + s;
        if (co < a.cout && ci < a.cin && jw >= 0 && jw < a.k)
            v = a.w[((size_t)ci * a.cout + co) * a.k + jw];
    }
    // with bias steps the stream of M tile mt is 512 elements longer
    const long long dst = idx + (a.bias_step ? (long long)mt * 512 : 0);
    ET::pack_store(a.out, dst, v);
}

// The conv bias as one more k16 step of the weight stream ("bias step", the
// last 64 fragments of every M tile's stream): A[co][k] = {hi(b), lo(b), 0...}
// in the lanes holding k = 0..7 and zeros elsewhere; multiplied with an
// all-ones B fragment the MFMA adds b[co] to every column of the tile, exact
// to the operand type's DOUBLE precision (hi + lo split; fp32: b itself).
// The accumulator then needs neither a bias fill nor an add in the epilogue.
template <class ET>
__global__ __launch_bounds__(256) void pm_pack_bias_step_kernel(
    const float* __restrict__ bias, void* out, int cout, int mtiles,
    long long weights_per_mt) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= mtiles * 512) return;
    const int mt = idx / 512, r = idx % 512;
    const int lane = r / 8, e = r % 8;
    const int co = mt * 32 + (lane & 31);
    // (k = 0: the bias rounded to the operand type; k = 1, 16-bit types: what
    // the rounding left - the MFMA against ones adds both)
    float v = 0.f;
    if (lane < 32 && co < cout) {
        const float b = bias[co];
        const float hi = (float)ET::cvt(b);
        if (e == 0) v = ET::BIAS_SPLIT ? hi : b;
        if (e == 1 && ET::BIAS_SPLIT) v = b - hi;
    }
    ET::pack_store(out, (long long)mt * (weights_per_mt + 512) +
                            weights_per_mt + r, v);
}

// dst[i] = i < n ? src[i] : 0  for i < n_pad; optionally tiled `rep` times
__global__ void pm_pad_bias_kernel(
    const float* __restrict__ src, float* __restrict__ dst, int n, int n_pad,
    int rep) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad * rep) return;
    const int c = i % n_pad;
    dst[i] = (src && c < n) ? src[c] : 0.f;
}

// ---------------------------------------------------------------------------
// (B, C, T) fp32 -> (B, T, Cpad) fp32 channels-last, zero padded.
// The module-seam entry HiFiGAN.forward(x (B,113,T), ...) hifigan.py:63.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pm_to_channels_last_kernel(
    const float* __restrict__ src, float* __restrict__ dst, int C, int T,
    int Cpad) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, t = t0 + tx;
        tile[i][tx] = (c < C && t < T) ? src[((size_t)b * C + c) * T + t] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + i, c = c0 + tx;
        if (t < T && c < Cpad) dst[((size_t)b * T + t) * Cpad + c] = tile[tx][i];
    }
}

// ---------------------------------------------------------------------------
// Generator.prepare_features  (model/generator.py:137-197, default config)
//   channels [ppg 0:40 | pitch embedding 40:104 | loudness 104:112 | per 112]
// ---------------------------------------------------------------------------
struct FeatureArgs {
    const float* loudness;     // (B, F, T) dB, F = 8 or 513 (any >= bands)
    const float* pitch;        // (B, T) Hz
    const float* periodicity;  // (B, T)
    const float* ppg;          // (B, P, T)
    const float* pitch_edges;  // (NB) ascending bin edges (load.py:54-74)
    const float* pitch_table;  // (NB, E) embedding
    float* out_cl;             // (B, T, Cpad) or null
    float* out_ref;            // (B, P + E + bands + 1, T) or null
    int B, T, F, P, NB, E, bands, Cpad;
    int band_start[17];        // int(b * F / bands), b = 0..bands
    int sparse_method;         // PM_SPARSE_* (generator.py:140-147)
    int rank_below, rank_above;  // percentile: torch.quantile(linear) gather
    float rank_weight;           //   indices and lerp weight
    float threshold;           // constant: the threshold itself
    int topk;                  // topk: entries kept per frame
    float fmin, fmax, min_db, db_range;
    float period_rate;         // > 0: append SAMPLE_RATE / hz (FARGAN)
};

// A workgroup of 256 threads builds the rows of FRAMES = 64 consecutive frames
// in LDS - work items are (frame, channel) pairs, so every global read is
// coalesced along time and the 40 x 40 rank comparisons of a frame are spread
// over 40 threads - and writes them out as one contiguous (64, Cpad) block of
// the channels-last tensor (and / or channel rows of the reference layout).
// Arithmetic per value is unchanged: rank by counting with index tie-break,
// torch's lerp, log / max / exp / sequential sum in channel order.
template <int FRAMES>
__global__ __launch_bounds__(256) void pm_prepare_features_kernel(
    FeatureArgs a) {
    constexpr int NT = 256;
    extern __shared__ float sm[];
    const int P = a.P, T = a.T;
    const int Cb = P + a.E + a.bands + 1;
    const int C = Cb + (a.period_rate > 0.f ? 1 : 0);
    const int W = a.Cpad > C ? a.Cpad : C;     // row width (Cpad is 0 without out_cl)
    const int RS = W + 1;                      // row stride (odd: no conflicts)
    float* col = sm;                           // [P][FRAMES] raw ppg
    float* row = col + P * FRAMES;             // [FRAMES][RS] assembled rows
    float* quant = row + FRAMES * RS;          // [2][FRAMES] below / above
    float* stat = quant + 2 * FRAMES;          // [2][FRAMES] max, sum
    float* hzs = stat + 2 * FRAMES;            // [FRAMES] clipped pitch
    int* bins = reinterpret_cast<int*>(hzs + FRAMES);   // [FRAMES]
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * FRAMES;
    const int nf = min(FRAMES, T - t0);        // live frames of this tile
    const int tid = threadIdx.x;

    // ppg tile, pitch, zero the padded tail of every row
    for (int i = tid; i < P * FRAMES; i += NT) {
        const int c = i / FRAMES, f = i % FRAMES;
        col[i] = f < nf ? a.ppg[((size_t)b * P + c) * T + t0 + f] : 0.f;
    }
    for (int i = tid; i < FRAMES * (W - C); i += NT) {
        const int f = i / (W - C), c = C + i % (W - C);
        row[f * RS + c] = 0.f;
    }
    if (tid < FRAMES) {
        // --- pitch: clip, searchsorted(right=False), clip (:152-164) ---
        float hz = tid < nf ? a.pitch[(size_t)b * T + t0 + tid] : a.fmin;
        hz = fminf(fmaxf(hz, a.fmin), a.fmax);
        int lo = 0, hi = a.NB;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (a.pitch_edges[mid] < hz) lo = mid + 1; else hi = mid;
        }
        hzs[tid] = hz;
        bins[tid] = lo > a.NB - 1 ? a.NB - 1 : lo;
    }
    __syncthreads();

    // --- ppgs.sparsify(ppg, method, threshold)  (generator.py:140-147) ---
    // percentile (default): q = per-frame quantile, keep v > q
    // constant:             keep v > threshold
    // topk:                 keep the k largest (ties: lower channel first)
    // then softmax(log(p + 1e-8)) over channels; method None passes ppg on
    const int method = a.sparse_method;
    if (method == 1 || method == 3) {
        for (int i = tid; i < P * FRAMES; i += NT) {
            const int c = i / FRAMES, f = i % FRAMES;
            const float vi = col[i];
            int rank = 0;
            for (int j = 0; j < P; ++j) {
                const float vj = col[j * FRAMES + f];
                rank += (vj < vi) || (vj == vi && j < c);
            }
            if (method == 1) {
                if (rank == a.rank_below) quant[f] = vi;
                if (rank == a.rank_above) quant[FRAMES + f] = vi;
            } else {
                // descending order with index tie-break = ascending rank
                // counted with the opposite tie-break; keep the top k
                int above = 0;
                for (int j = 0; j < P; ++j) {
                    const float vj = col[j * FRAMES + f];
                    above += (vj > vi) || (vj == vi && j < c);
                }
                row[f * RS + c] = above < a.topk ? vi : 0.f;
            }
        }
        __syncthreads();
    }
    if (tid < FRAMES) {
        const int f = tid;
        if (method != 0) {
            float q = a.threshold;
            if (method == 1) {
                const float below = quant[f], above = quant[FRAMES + f];
                // torch lerp (weight < 0.5 branch is the one 0.85 * 39 takes)
                const float wq = a.rank_weight;
                q = wq < 0.5f ? below + wq * (above - below)
                              : above - (above - below) * (1.f - wq);
            }
            float mx = -INFINITY;
            for (int c = 0; c < P; ++c) {
                float v = col[c * FRAMES + f];
                if (method == 3) v = row[f * RS + c];
                else v = v > q ? v : 0.f;
                v = logf(v + 1e-8f);
                col[c * FRAMES + f] = v;
                mx = fmaxf(mx, v);
            }
            float sum = 0.f;
            for (int c = 0; c < P; ++c) {
                const float ev = expf(col[c * FRAMES + f] - mx);
                col[c * FRAMES + f] = ev;
                sum += ev;
            }
            stat[f] = sum;
        } else {
            stat[f] = 1.f;
        }
        // periodicity (:187-188), FARGAN pitch period (:191-195)
        row[f * RS + Cb - 1] =
            f < nf ? a.periodicity[(size_t)b * T + t0 + f] : 0.f;
        if (C > Cb) row[f * RS + Cb] = a.period_rate / hzs[f];
    }
    __syncthreads();
    for (int i = tid; i < P * FRAMES; i += NT) {
        const int c = i / FRAMES, f = i % FRAMES;
        row[f * RS + c] = method != 0 ? col[i] / stat[f] : col[i];
    }
    // pitch embedding gather (:160-164)
    for (int i = tid; i < a.E * FRAMES; i += NT) {
        const int f = i / a.E, e = i % a.E;
        row[f * RS + P + e] = a.pitch_table[(size_t)bins[f] * a.E + e];
    }
    // --- loudness band means + normalize (:172-184, loudness.py:144-146) ---
    for (int i = tid; i < a.bands * FRAMES; i += NT) {
        const int band = i / FRAMES, f = i % FRAMES;
        float v = 0.f;
        if (f < nf) {
            const float* lp = a.loudness + (size_t)b * a.F * T + t0 + f;
            float s = 0.f;
            const int r0 = a.band_start[band], r1 = a.band_start[band + 1];
            for (int r = r0; r < r1; ++r) s += lp[(size_t)r * T];
            v = (s / (float)(r1 - r0) - a.min_db) / a.db_range;
        }
        row[f * RS + P + a.E + band] = v;
    }
    __syncthreads();

    if (a.out_cl) {
        float* o = a.out_cl + ((size_t)b * T + t0) * a.Cpad;
        for (int i = tid; i < nf * a.Cpad; i += NT)
            o[i] = row[(i / a.Cpad) * RS + i % a.Cpad];
    }
    if (a.out_ref) {
        float* o = a.out_ref + (size_t)b * C * T + t0;
        for (int i = tid; i < C * FRAMES; i += NT) {
            const int c = i / FRAMES, f = i % FRAMES;
            if (f < nf) o[(size_t)c * T + f] = row[f * RS + c];
        }
    }
}

// ---------------------------------------------------------------------------
// prepare_global_features (generator.py:49-70): embedding row + two ratios
// ---------------------------------------------------------------------------
// sbr / lr may be null (AUGMENT_PITCH / AUGMENT_LOUDNESS off): the row is
// then S (+1) wide. A speaker id outside [0, num_speakers) reads nothing and
// yields a NaN row (the reference's Embedding raises; NaN audio is the
// device-side equivalent - the Python layer validates host-side ids first).
__global__ void pm_global_features_kernel(
    const long long* __restrict__ speakers, const float* __restrict__ sbr,
    const float* __restrict__ lr, const float* __restrict__ table,
    float* __restrict__ out, int B, int S, int num_speakers) {
    const int b = blockIdx.x;
    const long long spk = speakers[b];
    const bool valid = spk >= 0 && spk < num_speakers;
    const int W = S + (sbr ? 1 : 0) + (lr ? 1 : 0);
    for (int i = threadIdx.x; i < S; i += blockDim.x)
        out[(size_t)b * W + i] =
            valid ? table[(size_t)spk * S + i] : __uint_as_float(0x7fc00000u);
    if (threadIdx.x == 0) {
        int c = S;
        if (sbr) out[(size_t)b * W + c++] = sbr[b];
        if (lr) out[(size_t)b * W + c] = lr[b];
    }
}

// ZERO_SHOT speaker conditioning (generator.py:35-38): the speaker "id" is a
// WavLM x-vector (B, E) and the embedding a Linear(E -> S). One wave per
// (utterance, output channel), fp32, sequential-in-lane + shuffle reduction.
__global__ __launch_bounds__(256) void pm_global_features_linear_kernel(
    const float* __restrict__ emb, const float* __restrict__ w,
    const float* __restrict__ bias, const float* __restrict__ sbr,
    const float* __restrict__ lr, float* __restrict__ out, int B, int E,
    int S) {
    const int b = blockIdx.y;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int W = S + (sbr ? 1 : 0) + (lr ? 1 : 0);
    if (m < S) {
        float s = 0.f;
        for (int c = lane; c < E; c += 64)
            s += w[(size_t)m * E + c] * emb[(size_t)b * E + c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if (lane == 0) out[(size_t)b * W + m] = s + bias[m];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int c = S;
        if (sbr) out[(size_t)b * W + c++] = sbr[b];
        if (lr) out[(size_t)b * W + c] = lr[b];
    }
}

// ---------------------------------------------------------------------------
// input_speaker_conv: 1x1 conv on (B|1, G, 1)  (hifigan.py:26-30, 68)
//   gbias[b][m] = bs[m] + sum_c Ws[m][c] g[b][c]      one wave per (b, m)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pm_speaker_bias_kernel(
    const float* __restrict__ g, const float* __restrict__ ws,
    const float* __restrict__ bs, float* __restrict__ out, int G, int M,
    int Mpad) {
    const int b = blockIdx.y;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (m >= Mpad) return;
    float s = 0.f;
    if (m < M)
        for (int c = lane; c < G; c += 64)
            s += ws[(size_t)m * G + c] * g[(size_t)b * G + c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (lane == 0) out[(size_t)b * Mpad + m] = m < M ? s + bs[m] : 0.f;
}

// ---------------------------------------------------------------------------
// Output layer: LeakyReLU -> Conv1d(C -> 1, k 7, pad 3, no bias) -> tanh
// (hifigan.py:55-60). HBM-bound: reads C floats, writes 1 per sample.
// ---------------------------------------------------------------------------
template <int THREADS>
__global__ __launch_bounds__(THREADS) void pm_out_conv_kernel(
    const float* __restrict__ x, const float* __restrict__ w,
    float* __restrict__ y, int Lmax, int C, int Cw,
    const int* __restrict__ lengths, int len_scale) {
    extern __shared__ float sm[];
    constexpr int KW = 7, HALO = 3;
    const int S = C + 1;
    float* xs = sm;                       // [(THREADS + 6)][C + 1]
    float* wsm = sm + (THREADS + 2 * HALO) * S;   // [KW][C]
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * THREADS;
    // ragged batch: samples past the utterance's end are written as zeros
    const int L = lengths ? min(lengths[b] * len_scale, Lmax) : Lmax;
    if (t0 >= L) {
        if (t0 + (int)threadIdx.x < Lmax)
            y[(size_t)b * Lmax + t0 + threadIdx.x] = 0.f;
        return;
    }
    const float* xb = x + (size_t)b * Lmax * C;
    for (int i = threadIdx.x; i < KW * C; i += THREADS) {
        const int j = i / C, c = i % C;
        wsm[i] = c < Cw ? w[c * KW + j] : 0.f;
    }
    const int Q = C / 4;
    for (int i = threadIdx.x; i < (THREADS + 2 * HALO) * Q; i += THREADS) {
        const int row = i / Q, q = i % Q;
        const int t = t0 - HALO + row;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t >= 0 && t < L)
            v = *reinterpret_cast<const float4*>(xb + (size_t)t * C + q * 4);
        float* d = xs + row * S + q * 4;
        d[0] = pm_lrelu(v.x); d[1] = pm_lrelu(v.y);
        d[2] = pm_lrelu(v.z); d[3] = pm_lrelu(v.w);
    }
    __syncthreads();
    const int t = t0 + threadIdx.x;
    if (t >= L) {
        if (t < Lmax) y[(size_t)b * Lmax + t] = 0.f;
        return;
    }
    float acc = 0.f;
    for (int j = 0; j < KW; ++j) {
        const float* xr = xs + (threadIdx.x + j) * S;
        const float* wr = wsm + j * C;
        for (int c = 0; c < C; ++c) acc = fmaf(wr[c], xr[c], acc);
    }
    y[(size_t)b * Lmax + t] = tanhf(acc);
}

// channel-row length of pm_out_conv32_kernel: >= tile + halo + window slack,
// even (8-byte window reads), = 10 mod 32 (staging writes 2-way at worst)
#define PM_OUT32_RL(threads) ((((threads) * 2 + 8 + 31) / 32) * 32 + 10)

// Same layer for the default 32-channel last stage, restructured around what
// bounds it: the generic kernel reads LDS twice per FMA (activation + weight).
// Here the tile sits channel-major in LDS and every thread produces two
// consecutive samples from an 8-sample register window per channel: four
// conflict-free ds_read_b64 of activations and the channel's 7 taps from the
// scalar cache feed 14 FMAs.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void pm_out_conv32_kernel(
    const float* __restrict__ x, const float* __restrict__ w,
    float* __restrict__ y, int Lmax, const int* __restrict__ lengths,
    int len_scale) {
    constexpr int C = 32, KW = 7, HALO = 3, PT = 2;
    constexpr int TILE = THREADS * PT;
    constexpr int ROWS = TILE + 2 * HALO;
    constexpr int RL = PM_OUT32_RL(THREADS);  // floats per channel row
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [C][RL]
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * TILE;
    const int L = lengths ? min(lengths[b] * len_scale, Lmax) : Lmax;
    float* yb = y + (size_t)b * Lmax;
    if (t0 >= L) {
        for (int i = threadIdx.x; i < TILE; i += THREADS)
            if (t0 + i < Lmax) yb[t0 + i] = 0.f;
        return;
    }
    const float* xb = x + (size_t)b * Lmax * C;
    constexpr int Q = C / 4;
    // every load of the tile is in flight before the first LDS write (a
    // load-use-per-iteration loop is one HBM round trip per iteration)
    constexpr int ITER = (ROWS * Q + THREADS - 1) / THREADS;
    float4 v[ITER];
#pragma unroll
    for (int u = 0; u < ITER; ++u) {
        const int i = threadIdx.x + u * THREADS;
        const int row = i / Q, q = i % Q;
        const int t = t0 - HALO + row;
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < ROWS && t >= 0 && t < L)
            v[u] = *reinterpret_cast<const float4*>(xb + (size_t)t * C + q * 4);
    }
#pragma unroll
    for (int u = 0; u < ITER; ++u) {
        const int i = threadIdx.x + u * THREADS;
        const int row = i / Q, q = i % Q;
        if (row < ROWS) {
            const float4 r = pm_lrelu4(v[u]);
            sm[(q * 4 + 0) * RL + row] = r.x;
            sm[(q * 4 + 1) * RL + row] = r.y;
            sm[(q * 4 + 2) * RL + row] = r.z;
            sm[(q * 4 + 3) * RL + row] = r.w;
        }
    }
    __syncthreads();
    const int first = threadIdx.x * PT;      // window rows first .. first + 7
    float acc[PT] = {0.f, 0.f};
    // (fully unrolled, the scheduler hoists every LDS read of all 32
    // channels: 512 VGPRs and spills)
#pragma unroll 2
    for (int c = 0; c < C; ++c) {
        float win[PT + KW - 1];
#pragma unroll
        for (int r = 0; r < PT + KW - 1; r += 2) {
            const float2 v =
                *reinterpret_cast<const float2*>(sm + c * RL + first + r);
            win[r] = v.x; win[r + 1] = v.y;
        }
        // the channel's 7 taps as scalar loads (uniform address, scalar
        // cache): no LDS bandwidth for weights. The opaque channel index keeps
        // the loads in the loop - hoisted, the 224 of them spill the SGPR file
        // through v_readlane (2x slower); two broadcast ds_read_b128 of an
        // LDS copy instead: 0.222 against 0.212 ms (profiles/r04/ab_tail_kernels.txt)
        int cc = c;
        asm volatile("" : "+s"(cc));
        const float* __restrict__ wc = w + cc * KW;
        float wj[KW];
#pragma unroll
        for (int j = 0; j < KW; ++j) wj[j] = wc[j];
#pragma unroll
        for (int j = 0; j < KW; ++j)
#pragma unroll
            for (int p = 0; p < PT; ++p)
                acc[p] = fmaf(wj[j], win[p + j], acc[p]);
    }
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const int t = t0 + first + p;
        if (t < Lmax) yb[t] = t < L ? tanhf(acc[p]) : 0.f;
    }
}

// ---------------------------------------------------------------------------
// promonet.edit: 1-D grid sampling of a frame
// sequence (edit/grid.py:12-45) with the per-feature post-ops of
// edit/core.py:114-127 fused: pitch is interpolated in log2 and re-exponent-
// iated, shifted by a ratio and clipped; loudness gets a dB offset.
//   mode 0: linear   y = a (fl + 1 - x) + b (x - fl), fl = floor(x),
//                    b = seq[min(fl + 1, n - 1)]  (replicate pad, grid.py:27-35)
//   mode 1: linear in log2, then 2 ** y            (core.py:114)
//   mode 2: nearest  seq[round(x)]                 (grid.py:41-42)
// then y = clip(y * scale + offset, lo, hi).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pm_grid_sample_kernel(
    const float* __restrict__ seq, const float* __restrict__ grid,
    float* __restrict__ out, int rows, int n_in, int n_out, int mode,
    float scale, float offset, float lo, float hi) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y;
    if (j >= n_out) return;
    const float* s = seq + (size_t)r * n_in;
    const float x = grid ? grid[j] : (float)j;
    float y;
    if (mode == 2) {
        int i = (int)rintf(x);
        i = i < 0 ? 0 : (i >= n_in ? n_in - 1 : i);
        y = s[i];
    } else {
        int fl = (int)floorf(x);
        fl = fl < 0 ? 0 : (fl >= n_in ? n_in - 1 : fl);
        const int up = fl + 1 < n_in ? fl + 1 : n_in - 1;
        float a = s[fl], b = s[up];
        if (mode == 1) { a = log2f(a); b = log2f(b); }
        y = a * ((float)(fl + 1) - x) + b * (x - (float)fl);
        if (mode == 1) y = exp2f(y);
    }
    y = y * scale + offset;
    out[(size_t)r * n_out + j] = fminf(fmaxf(y, lo), hi);
}

// ---------------------------------------------------------------------------
// promonet.edit.from_features with stretch_unvoiced / stretch_silence off
// (edit/core.py:57-110): a time-stretch grid whose step size follows the
// probability mass of the SELECTED phonemes, so that only those are stretched.
//   selected[t] = sum_k ppg[indices[k]][t]
//   effective   = (target - (T - sum selected)) / sum selected
//   grid[j]     = grid[j - 1] + 1 / (p effective + 1 - p),  p = selected
//                 interpolated at grid[j - 1]
// Phase 1 is parallel; phase 2 is the reference's sequential fp32 recurrence
// (each step depends on the previous position), walked by one thread out of
// LDS - a few hundred dependent steps, microseconds.
// ---------------------------------------------------------------------------
struct StretchArgs {
    const float* ppg;      // (P, T)
    const int* indices;    // (n) selected phoneme rows
    float* selected;       // (T) scratch, also returned for inspection
    float* grid;           // (target)
    int n, T, target, P;
};

// (the recurrence is the reference's fp32 step sequence, edit/core.py:94-110:
// no FMA contraction, so that hundreds of dependent steps round as torch's
// unfused multiply / add do)
__global__ __launch_bounds__(256) void pm_stretch_grid_kernel(StretchArgs a) {
#pragma clang fp contract(off)
    extern __shared__ float sel_lds[];         // T floats when they fit
    __shared__ float red[4];
    __shared__ int bad_row;
    const int tid = threadIdx.x;
    const bool in_lds = (size_t)a.T * sizeof(float) <= 64 * 1024;
    if (tid == 0) bad_row = 0;
    __syncthreads();
    float partial = 0.f;
    bool bad = false;
    for (int t = tid; t < a.T; t += 256) {
        float s = 0.f;
        for (int k = 0; k < a.n; ++k) {
            // a row outside the PPG reads nothing (and poisons the grid, below)
            const int row = a.indices[k];
            if (row >= 0 && row < a.P) s += a.ppg[(size_t)row * a.T + t];
            else bad = true;
        }
        a.selected[t] = s;
        if (in_lds) sel_lds[t] = s;
        partial += s;
    }
    if (bad) bad_row = 1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) partial += __shfl_down(partial, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = partial;
    __syncthreads();
    if (bad_row) {
        // NaN as a BIT PATTERN through integer stores: this translation unit
        // is built -fno-honor-nans, under which floating-point arithmetic on
        // a NaN (and __builtin_nanf itself) is undefined to the compiler
        for (int j = tid; j < a.target; j += 256)
            reinterpret_cast<unsigned*>(a.grid)[j] = 0x7fc00000u;
        for (int t = tid; t < a.T; t += 256)
            reinterpret_cast<unsigned*>(a.selected)[t] = 0x7fc00000u;
        return;
    }
    if (tid != 0) return;
    const float* sel = in_lds ? sel_lds : a.selected;
    const float total = red[0] + red[1] + red[2] + red[3];
    const float unselected = (float)a.T - total;
    const float effective = ((float)a.target - unselected) / total;
    float position = 0.f;
    a.grid[0] = 0.f;
    for (int j = 1; j < a.target; ++j) {
        // (a selection without probability mass makes the reference's
        // arithmetic produce inf / NaN / negative positions: clamp the index -
        // a NaN compares false - so that nothing is read out of bounds; the
        // values stay what the reference's formula yields)
        int left = position >= 0.f
            ? (int)fminf(floorf(position), (float)(a.T - 1)) : 0;
        float probability;
        if (left + 1 < a.T) {
            const float offset = position - (float)left;
            probability = offset * sel[left + 1] + (1.f - offset) * sel[left];
        } else {
            probability = sel[left];
        }
        const float ratio = probability * effective + (1.f - probability);
        position += 1.f / ratio;
        a.grid[j] = position;
    }
}

// Sustained MFMA rate probe (pm_mfma_probe): 4 x 4 independent 32x32x16
// MFMAs per iteration and wave on register-resident operands - what the matrix
// pipe holds on THIS device under ITS power cap, measured beside the bench's
// kernels instead of quoted from an earlier round. KIND 1: f16, 2: bf16.
template <int KIND>
__global__ __launch_bounds__(256) void pm_mfma_probe_kernel(
    const uint4* __restrict__ src, float* __restrict__ sink, int iters) {
    uint4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = src[(threadIdx.x + 256 * i) & 4095];
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = src[(threadIdx.x + 256 * (i + 4)) & 4095];
    floatx16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (KIND == 1)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                        __builtin_bit_cast(half8, a[k]),
                        __builtin_bit_cast(half8, b[i]), acc[i], 0, 0, 0);
                else
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(bf16x8, a[k]),
                        __builtin_bit_cast(bf16x8, b[i]), acc[i], 0, 0, 0);
            }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) sink[0] = s;
}

