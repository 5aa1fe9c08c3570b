// Framed-DFT formulation of the STFT (the BACKWARD pass of
// spectrogram.from_audio, and a brute-force cross-check of the forward FFT of
// pm_fft.h) and the stand-alone mel kernels.
// Reference: promonet/preprocess/spectrogram.py, promonet/preprocess/loudness.py
//
// The 1024-point hann-windowed DFT at hop 256 is a 4-tap convolution over the
// padded audio viewed as (frames + 3, 256) "channels-last" rows, so it runs on
// the exact-fp32 MFMA conv kernel (pm_conv.h, EPI 1/3) against a precomputed
// (re, im)-interleaved windowed DFT basis. The adjoint of a linear map is the
// same GEMM transposed, which is why the backward keeps this form.
#pragma once
#include "pm_common.h"

// torch.nn.functional.pad(audio, (p, p), mode='reflect') (spectrogram.py:36-37,
// loudness.py:20-25): (B, N) -> (B, N + 2 p)
__global__ __launch_bounds__(256) void pm_reflect_pad_kernel(
    const float* __restrict__ src, float* __restrict__ dst, int N, int pad,
    int Np) {
    // Np <= N + 2 pad: number of padded samples kept per utterance
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Np) return;
    int j = i - pad;
    if (j < 0) j = -j;
    if (j >= N) j = 2 * (N - 1) - j;
    dst[(size_t)b * Np + i] = src[(size_t)b * N + j];
}

// Windowed DFT basis in torch Conv1d layout w[m][c][j], m = 2 bin + part,
// sample n = j * hop + c: hann(n) cos(2 pi bin n / nfft) | -hann(n) sin(...)
// hann = periodic (torch.hann_window(1024), spectrogram.py:29).
__global__ __launch_bounds__(256) void pm_dft_basis_kernel(
    float* __restrict__ w, int bins, int nfft, int hop) {
    const int taps = nfft / hop;
    const long long total = 2LL * bins * nfft;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int j = idx % taps;
    const int c = (idx / taps) % hop;
    const int m = idx / ((long long)taps * hop);
    const int bin = m >> 1, part = m & 1;
    const int n = j * hop + c;
    const double pi2 = 6.283185307179586476925286766559;
    const double win = 0.5 - 0.5 * cos(pi2 * n / nfft);
    // reduce the phase exactly before the trig call
    const int ph = (int)(((long long)bin * n) % nfft);
    const double ang = pi2 * ph / nfft;
    w[idx] = (float)(part == 0 ? win * cos(ang) : -win * sin(ang));
}

// Adjoint of pm_reflect_pad_kernel: grad (B, Np) of the (truncated) padded
// signal -> grad (B, N) of the audio. Sample i receives its own padded
// position and, near the ends, the mirrored ones.
__global__ __launch_bounds__(256) void pm_reflect_pad_adjoint_kernel(
    const float* __restrict__ gp, float* __restrict__ ga, int N, int pad,
    int Np) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float* g = gp + (size_t)b * Np;
    float v = 0.f;
    if (i + pad < Np) v += g[i + pad];
    if (i >= 1 && i <= pad && pad - i < Np) v += g[pad - i];      // left mirror
    const int m = 2 * (N - 1) - i + pad;                          // right mirror
    if (i <= N - 2 && m >= N + pad && m < N + 2 * pad && m < Np) v += g[m];
    ga[(size_t)b * N + i] = v;
}

// Transposed DFT basis for the backward overlap-add conv, torch Conv1d layout
// wt[c][m][j] = w[m][c][taps - 1 - j] (w from pm_dft_basis_kernel), rows
// m >= 2 bins are zero.
__global__ __launch_bounds__(256) void pm_dft_basis_transpose_kernel(
    const float* __restrict__ w, float* __restrict__ wt, int rows, int rows_pad,
    int hop, int taps) {
    const long long total = (long long)hop * rows_pad * taps;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int j = idx % taps;
    const int m = (idx / taps) % rows_pad;
    const int c = idx / ((long long)taps * rows_pad);
    wt[idx] = m < rows ? w[((size_t)m * hop + c) * taps + (taps - 1 - j)] : 0.f;
}

// Backward of pm_mel_kernel, two passes. Pass 1: gl[m][t] = grad[m][t] /
// (basis[m] . spec[:, t]), zero where the forward clamp was active.
__global__ __launch_bounds__(256) void pm_mel_backward_rows_kernel(
    const float* __restrict__ spec, const float* __restrict__ basis,
    const float* __restrict__ grad, float* __restrict__ gl, int F, int M,
    int T, int use_thr, float thr) {
    __shared__ int span[2];
    const int b = blockIdx.z, m = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const float* br = basis + (size_t)m * F;
    if (threadIdx.x == 0) { span[0] = F; span[1] = 0; }
    __syncthreads();
    for (int f = threadIdx.x; f < F; f += 256)
        if (br[f] != 0.f) {
            atomicMin(&span[0], f);
            atomicMax(&span[1], f + 1);
        }
    __syncthreads();
    if (t >= T) return;
    const float* sp = spec + (size_t)b * F * T + t;
    float acc = 0.f;
    for (int f = span[0]; f < span[1]; ++f)
        acc = fmaf(br[f], sp[(size_t)f * T], acc);
    const size_t at = ((size_t)b * M + m) * T + t;
    const bool pass = !use_thr || logf(acc) >= thr;   // torch.clamp backward
    gl[at] = pass ? grad[at] / acc : 0.f;
}

// Pass 2: grad_spec[f][t] = sum_m basis[m][f] gl[m][t] (a bin sits under at
// most a few triangular filters: zero weights are skipped)
__global__ __launch_bounds__(256) void pm_mel_backward_cols_kernel(
    const float* __restrict__ basis, const float* __restrict__ gl,
    float* __restrict__ gspec, int F, int M, int T) {
    const int b = blockIdx.z, f = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    float acc = 0.f;
    for (int m = 0; m < M; ++m) {
        const float wgt = basis[(size_t)m * F + f];
        if (wgt != 0.f) acc = fmaf(wgt, gl[((size_t)b * M + m) * T + t], acc);
    }
    gspec[((size_t)b * F + f) * T + t] = acc;
}

// linear_to_mel (spectrogram.py:111-133): out[b][m][t] = log(sum_f
// basis[m][f] spec[b][f][t]) with optional clamp. One thread per (m, t). A mel
// filter is a triangle over a few dozen of the F bins: the workgroup first
// finds its row's non-zero span and sums only that (same order, the skipped
// terms are exact zeros), so the spectrogram is read ~2x instead of M times.
__global__ __launch_bounds__(256) void pm_mel_kernel(
    const float* __restrict__ spec, const float* __restrict__ basis,
    float* __restrict__ out, int F, int M, int T, int use_thr, float thr) {
    __shared__ int span[2];
    const int b = blockIdx.z, m = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const float* br = basis + (size_t)m * F;
    if (threadIdx.x == 0) { span[0] = F; span[1] = 0; }
    __syncthreads();
    for (int f = threadIdx.x; f < F; f += 256)
        if (br[f] != 0.f) {
            atomicMin(&span[0], f);
            atomicMax(&span[1], f + 1);
        }
    __syncthreads();
    if (t >= T) return;
    const float* sp = spec + (size_t)b * F * T + t;
    float acc = 0.f;
    for (int f = span[0]; f < span[1]; ++f)
        acc = fmaf(br[f], sp[(size_t)f * T], acc);
    float v = logf(acc);
    if (use_thr) v = fmaxf(v, thr);
    out[((size_t)b * M + m) * T + t] = v;
}

__device__ __forceinline__ float pm_float_from_order_bits(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
