/* promonet_hip.h - C ABI of libpromonet_hip.so
 *
 * MI355X (gfx950 / CDNA4) implementation of promonet's synthesis hot path.
 * The reference (maxrmorrison/promonet @ 2024_08_07) is 100 % Python on
 * PyTorch and has NO FFI / plugin interface of its own; the seams this
 * library attaches to are Python module methods. Each entry point cites the
 * reference code it replaces (paths relative to the reference repo).
 *
 * Conventions
 *   - C linkage, plain pointers and sizes, no torch types.
 *   - Every pointer named *dev* / every tensor argument is a DEVICE pointer
 *     owned by the caller (e.g. torch.Tensor.data_ptr()); never mutated
 *     unless documented as an output.
 *   - `stream` is a hipStream_t passed as void* (0 = default stream). All
 *     compute entry points are asynchronous on that stream.
 *   - Return value: PM_OK (0) or a negative PM_E* code; pm_last_error()
 *     returns a thread-local human-readable message for the last failure.
 *   - No allocation inside the forward path: the caller passes a workspace
 *     whose size comes from pm_hifigan_workspace_bytes(). Weight storage is
 *     allocated at load time (pm_hifigan_load_tensor / pm_hifigan_finalize).
 *   - fp32 tensors are dense row-major in the reference's (PyTorch) layout:
 *     activations (B, C, T), conv weights (C_out, C_in, k), transposed-conv
 *     weights (C_in, C_out, k). Arguments suffixed `_cl` are the library's
 *     internal channels-last layout (B, T, C_pad), C_pad = round_up(C, 32).
 */
#ifndef PROMONET_HIP_H
#define PROMONET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PM_OK 0
#define PM_EINVAL (-1)      /* bad argument / unsupported configuration */
#define PM_ESTATE (-2)      /* call order (e.g. forward before finalize)  */
#define PM_EHIP (-3)        /* a HIP runtime call failed                   */
#define PM_ENOMEM (-4)      /* workspace too small / allocation failed     */
#define PM_ETIMEOUT (-5)    /* a bounded inter-workgroup wait gave up      */

/* MFMA operand type (accumulation is always fp32; activations between
 * kernels are always fp32 in HBM) */
#define PM_F32 0            /* v_mfma_f32_32x32x2_f32   - exact fp32       */
#define PM_F16 1            /* v_mfma_f32_32x32x16_f16                     */
#define PM_BF16 2           /* v_mfma_f32_32x32x16_bf16                    */
#define PM_F16X3 3          /* f16 operands split hi + lo, three MFMAs per
                             * step: ~21 bits per factor (trained-scale
                             * accuracy for the last stage)  */
#define PM_F16A2 4          /* f16 operands, the ACTIVATIONS split hi + lo
                             * (two MFMAs per step, f16 weight stream). As
                             * the type of a STAGE: its Blocks; the stage's
                             * upsampler - one rounding of a wide sum - runs
                             * PM_F16X3                                    */
#define PM_F16UX 5          /* stage / engine type only: f16 Blocks behind a
                             * PM_F16X3 upsampler                          */

#define PM_MAX_STAGES 8
#define PM_MAX_RESBLOCKS 4
#define PM_MAX_DILATIONS 4

typedef struct pm_hifigan_s* pm_hifigan_t;

/* Mirrors the constants the reference reads at import time:
 * promonet/config/defaults.py:216,250-262 and config/static.py:42-53.   */
typedef struct pm_hifigan_config {
    int num_features;                 /* NUM_FEATURES (113)                */
    int global_channels;              /* GLOBAL_CHANNELS (258)             */
    int initial_channels;             /* HIFIGAN_UPSAMPLE_INITIAL_SIZE     */
    int num_stages;                   /* len(HIFIGAN_UPSAMPLE_RATES)       */
    int upsample_rates[PM_MAX_STAGES];
    int upsample_kernel_sizes[PM_MAX_STAGES];
    int num_resblocks;                /* len(HIFIGAN_RESBLOCK_KERNEL_SIZES)*/
    int resblock_kernel_sizes[PM_MAX_RESBLOCKS];
    int num_dilations;
    int resblock_dilations[PM_MAX_RESBLOCKS][PM_MAX_DILATIONS];
    int compute_dtype;                /* PM_F32 ... PM_F16UX                 */
    /* Per-stage override of the MFMA operand type (upsampler + MRF of stage
     * i): 0 = compute_dtype, else 1 + PM_F32 ... PM_F16UX. E.g. bf16
     * for the wide stages and f16 for the last two, whose rounding reaches
     * the output most directly.                                           */
    int stage_compute_dtype[PM_MAX_STAGES];
} pm_hifigan_config;

int pm_version(void);
const char* pm_last_error(void);

/* ---- HiFi-GAN vocoder engine: replaces promonet.model.HiFiGAN ----------
 * (promonet/model/hifigan.py:13-77; constructed at generator.py:22-25)   */
int pm_hifigan_create(const pm_hifigan_config* config, pm_hifigan_t* out);
int pm_hifigan_destroy(pm_hifigan_t h);

/* Load one tensor of HiFiGAN.state_dict() by its reference key, e.g.
 *   input_feature_conv.weight (512,113,7)   input_speaker_conv.bias (512)
 *   model.0.model.1.weight_g (512,1,1)      model.0.model.1.weight_v (512,256,16)
 *   model.0.model.2.model.1.convs2.0.bias   model.5.weight (1,32,7)
 * Weight-normed layers accept either the (weight_g, weight_v) pair of the
 * checkpoint (folded here: w = g v / ||v||, replacing the per-forward
 * torch.nn.utils.weight_norm hook of model/core.py:43-45) or an already
 * folded `.weight`. Replaces torchutil.checkpoint.load -> load_state_dict
 * at promonet/synthesize/core.py:245. Synchronous w.r.t. `stream`.       */
int pm_hifigan_load_tensor(pm_hifigan_t h, const char* name,
                           const float* dev, const int64_t* shape, int ndim,
                           void* stream);
/* Verify every tensor is present; must precede forward.                  */
int pm_hifigan_finalize(pm_hifigan_t h, void* stream);

size_t pm_hifigan_workspace_bytes(pm_hifigan_t h, int batch, int frames);
int pm_hifigan_hopsize(pm_hifigan_t h);
int pm_hifigan_features_cl_channels(pm_hifigan_t h);

/* HiFiGAN.forward(x, g, p) (hifigan.py:63-70):
 *   features (B, num_features, T) fp32, global_features (Bg, global_channels)
 *   with Bg == 1 (broadcast) or B; out (B, 1, T * hop) fp32.              */
int pm_hifigan_forward(pm_hifigan_t h, const float* features,
                       const float* global_features, int global_batch,
                       float* out, int batch, int frames, void* workspace,
                       size_t workspace_bytes, void* stream);
/* Same with the features already channels-last (B, T, C_pad) - what
 * pm_prepare_features writes.                                             */
int pm_hifigan_forward_cl(pm_hifigan_t h, const float* features_cl,
                          const float* global_features, int global_batch,
                          float* out, int batch, int frames, void* workspace,
                          size_t workspace_bytes, void* stream);

/* Ragged batch (no reference counterpart: the reference synthesises one
 * utterance per call, synthesize/core.py:158-201): lengths (B) int32 device
 * array of valid frames <= frames; frames past an utterance's end are never
 * read, so each utterance equals its stand-alone synthesis; the output tail
 * is zero. features_cl != 0: features are channels-last (B, T, C_pad).     */
int pm_hifigan_forward_ragged(pm_hifigan_t h, const float* features,
                              int features_cl, const float* global_features,
                              int global_batch, const int* lengths,
                              float* out, int batch, int frames,
                              void* workspace, size_t workspace_bytes,
                              void* stream);

/* Optional per-launch timing (HIP events recorded on `stream` around every
 * kernel the forward launches; no reference counterpart - the reference only
 * has wall-clock torchutil timers, synthesize/core.py:222,250). collect()
 * synchronises and folds the recorded pairs into per-label totals; report()
 * returns "label launches total_ms algorithmic_flops algorithmic_bytes" lines.
 * profile_only(label) restricts the events to that label's launches (NULL or
 * "": every launch), so that a timed region pays for two events per step.  */
int pm_hifigan_profile_enable(pm_hifigan_t h, int enable);
int pm_hifigan_profile_only(pm_hifigan_t h, const char* label);
int pm_hifigan_profile_collect(pm_hifigan_t h);
int pm_hifigan_profile_reset(pm_hifigan_t h);
const char* pm_hifigan_profile_report(pm_hifigan_t h);

/* ---- conditioning: replaces Generator.prepare_features ------------------
 * (promonet/model/generator.py:137-197) incl. the third-party ppgs.sparsify
 * it calls (:140-147).
 *   loudness (B, F, T) dB with F == bands or any F >= bands (513)
 *   pitch (B, T) Hz, periodicity (B, T), ppg (B, P, T)
 *   pitch_edges (NB) = Generator.pitch_distribution, pitch_table (NB, E)
 *   out_ref (B, P+E+bands+1, T) and/or out_cl (B, T, C_pad); either may be
 *   NULL. period_rate > 0 appends the channel period_rate / clip(hz), the
 *   pitch period in samples FARGAN reads (generator.py:191-195).
 *   sparse_method = promonet.SPARSE_PPG_METHOD: PM_SPARSE_NONE (ppg passed
 *   through), _PERCENTILE (ppg_threshold = quantile in [0, 1], the default
 *   0.85), _CONSTANT (ppg_threshold = the cut itself), _TOPK (ppg_threshold
 *   = k, the number of entries kept per frame).                            */
#define PM_SPARSE_NONE 0
#define PM_SPARSE_PERCENTILE 1
#define PM_SPARSE_CONSTANT 2
#define PM_SPARSE_TOPK 3
int pm_prepare_features(const float* loudness, const float* pitch,
                        const float* periodicity, const float* ppg,
                        const float* pitch_edges, const float* pitch_table,
                        float* out_ref, float* out_cl, int batch, int frames,
                        int loudness_rows, int ppg_channels, int pitch_bins,
                        int embedding_size, int bands, int cl_channels,
                        int sparse_method, float ppg_threshold, float fmin,
                        float fmax, float min_db, float ref_db,
                        float period_rate, void* stream);

/* BaseGenerator.prepare_global_features (generator.py:49-70): speaker
 * embedding lookup + the augmentation ratios -> (B, S + 2). Either ratio
 * pointer may be NULL (AUGMENT_PITCH / AUGMENT_LOUDNESS off: the row is
 * narrower). speaker_table is (num_speakers, S); an id outside
 * [0, num_speakers) reads nothing and produces a NaN row.                 */
int pm_prepare_global_features(const int64_t* speakers,
                               const float* spectral_balance_ratios,
                               const float* loudness_ratios,
                               const float* speaker_table, float* out,
                               int batch, int speaker_channels,
                               int num_speakers, void* stream);
/* ZERO_SHOT variant (generator.py:35-38, synthesize/core.py:253-254): the
 * speaker is a (B, E) x-vector, the embedding a Linear(E -> S) with weight
 * (S, E) and bias (S).                                                     */
int pm_prepare_global_features_linear(const float* speaker_embeddings,
                                      const float* weight, const float* bias,
                                      const float* spectral_balance_ratios,
                                      const float* loudness_ratios,
                                      float* out, int batch,
                                      int embedding_channels,
                                      int speaker_channels, void* stream);

/* ---- per-kernel entry points (unit parity tests; weights in torch layout,
 * packed into `workspace` on every call) --------------------------------- */
size_t pm_op_workspace_bytes(int c_in, int c_out, int k);
/* One Block iteration (hifigan.py:204-210), channels-last fp32 in/out:
 *   y = x + conv2(lrelu(conv1(lrelu(x)))); mode 0: out = y,
 *   1: out = y * scale, 2: out += y * scale                               */
int pm_block_iteration_cl(int dtype, const float* x_cl, float* out_cl,
                          const float* w1, const float* b1, const float* w2,
                          const float* b2, int batch, int length,
                          int channels, int kernel_size, int dilation,
                          int mode, float scale, void* workspace,
                          size_t workspace_bytes, void* stream);
/* A whole Block (hifigan.py:198-210), all `niter` <= 3 dilations fused in
 * one kernel: 16-bit operands for channels <= 64 at every k, 128 at k 3 and -
 * walked / skewed variants, taken for long inputs or through pm_debug_force -
 * 128 at k 7 / 11 and 256 at k 3 / 7; fp32 operands for channels <= 64
 * (PM_EINVAL "no whole-Block kernel" otherwise: use pm_block_iteration_cl);
 * w1/b1/w2/b2 are HOST arrays of `niter` device pointers;
 * workspace >= 3 * pm_op_workspace_bytes(c, c, k) [+ pm_walk_scratch_bytes] */
int pm_block_cl(int dtype, const float* x_cl, float* out_cl,
                const float* const* w1, const float* const* b1,
                const float* const* w2, const float* const* b2,
                const int* dilations, int niter, int batch, int length,
                int channels, int kernel_size, int mode, float scale,
                void* workspace, size_t workspace_bytes, void* stream);
/* pm_block_cl whose result leaves as the next upsampler's MFMA operand
 * (hifigan.py:100-106 stages lrelu(stage output) for its ConvTranspose1d):
 * act16_cl (B, L, c_pad) 16-bit = cvt(lrelu(result)) in `act_dtype` (PM_F16 |
 * PM_BF16) instead of the fp32 tensor; out_cl is read (mode 2) and left as it
 * is. Only the skewed walk does it (long inputs, scratch behind the workspace,
 * or pm_debug_skew(1)): PM_ESTATE, and the fp32 result in out_cl, otherwise. */
int pm_block_act16_cl(int dtype, int act_dtype, const float* x_cl,
                      float* out_cl, void* act16_cl, const float* const* w1,
                      const float* const* b1, const float* const* w2,
                      const float* const* b2, const int* dilations, int niter,
                      int batch, int length, int channels, int kernel_size,
                      int mode, float scale, void* workspace,
                      size_t workspace_bytes, void* stream);
/* The whole MRF ResidualBlock of one stage (hifigan.py:141-145):
 * out = (Block_3(x) + Block_7(x) + Block_11(x)) / 3 in one launch, the sum
 * held in registers (32 channels). w1/b1/w2/b2: HOST arrays of 3 * niter
 * device pointers, Block-major (k = 3 first);
 * workspace >= 3 * niter * pm_op_workspace_bytes(c, c, 11); with
 * pm_walk_scratch_bytes(batch) more behind it the 4-byte operand layouts
 * (PM_F32, PM_F16X3, PM_F16A2) take the skewed whole-MRF walk on long batches
 * (three skewed Blocks per window, nothing recomputed)                      */
int pm_mrf_cl(int dtype, const float* x_cl, float* out_cl,
              const float* const* w1, const float* const* b1,
              const float* const* w2, const float* const* b2,
              const int* dilations, int niter, int batch, int length,
              int channels, void* workspace, size_t workspace_bytes,
              void* stream);
/* Input layers (hifigan.py:19-30, 67-68): Conv1d(c_in, c_out, 7, padding 3)
 * of the channels-last features + the k = 1 speaker conv of the (B|1, G)
 * global features as a per-utterance bias. workspace >=
 * pm_op_workspace_bytes(c_in, c_out, 7) + 256-aligned batch * pad32(c_out) * 4 */
int pm_input_conv_cl(int dtype, const float* x_cl, float* out_cl,
                     const float* weight, const float* bias,
                     const float* global_features,
                     const float* speaker_weight, const float* speaker_bias,
                     int global_batch, int global_channels, int batch,
                     int length, int c_in, int c_out, void* workspace,
                     size_t workspace_bytes, void* stream);
/* lrelu(optional) -> ConvTranspose1d(c_in, c_out, k = 2 r, stride r,
 * padding r / 2) (hifigan.py:97-106), channels-last: (B, L, c_in_pad) ->
 * (B, L * r, c_out_pad)                                                   */
int pm_conv_transpose_cl(int dtype, const float* x_cl, float* out_cl,
                         const float* w, const float* bias, int batch,
                         int length, int c_in, int c_out, int rate, int lrelu,
                         void* workspace, size_t workspace_bytes,
                         void* stream);
/* pm_conv_transpose_cl on an input that already holds the operand values:
 * x16_cl (B, L, c_in_pad) in the operand type `dtype` (PM_F16 | PM_BF16) =
 * cvt(lrelu(x)) as pm_block_act16_cl writes it; narrow (r = 2 ...) upsamplers */
int pm_conv_transpose_x16_cl(int dtype, const void* x16_cl, float* out_cl,
                             const float* w, const float* bias, int batch,
                             int length, int c_in, int c_out, int rate,
                             void* workspace, size_t workspace_bytes,
                             void* stream);
/* LeakyReLU -> Conv1d(C, 1, 7, pad 3, no bias) -> tanh (hifigan.py:55-60):
 * (B, L, c_pad) channels-last -> (B, L)                                   */
int pm_out_conv_tanh(const float* x_cl, const float* w, float* out,
                     int batch, int length, int channels, void* stream);
/* Debug instrumentation: per-workgroup phase timestamps (s_memtime) of the
 * following fused-conv launches; buffer of 8 uint64 per workgroup or NULL  */
int pm_debug_timeline(void* dev_buffer);
/* Test hook: the walked (carry-through-LDS) variants of pm_block_cl /
 * pm_mrf_cl / the engine and the multi-block path of the wide upsampler are
 * chosen from the grid size; walk_nseg > 0 forces the walked variant with
 * that many segments per utterance, upsample_groups > 0 that many M groups
 * per column tile; 0, 0 restores the heuristics. State per HOST THREAD;
 * PM_ESTATE unless the process runs with PROMONET_HIP_DEBUG=1.              */
int pm_debug_force(int walk_nseg, int upsample_groups);
/* Test hook: -1 keeps every launcher off the skewed whole-Block walk
 * (conv_block3_skew_kernel), 1 takes it wherever it fits, 0 restores the
 * default (the shapes it measured faster on). Per host thread, and only
 * with PROMONET_HIP_DEBUG=1, like pm_debug_force.                           */
int pm_debug_skew(int mode);
/* Measurement aid (bench.py): a register-resident loop of v_mfma_f32_32x32x16
 * (dtype PM_F16 | PM_BF16) - `workgroups` x 4 waves x `iterations` x 16 MFMAs
 * of 32 768 FLOP each - whose HIP-event time gives the matrix pipe's SUSTAINED
 * rate on this device under its power cap. operands: >= 32 768 bytes of finite
 * values of that type; sink: 4 bytes (never written). Always available (no
 * PROMONET_HIP_DEBUG needed): it changes no state of the library.           */
int pm_mfma_probe(int dtype, int iterations, const void* operands,
                  float* sink, int workgroups, void* stream);
/* Scratch the skewed whole-Block walk wants BEHIND the 3 x
 * pm_op_workspace_bytes() of pm_block_cl's workspace for `batch` utterances
 * (optional: without it the walked / stand-alone kernels run). The engine's
 * own scratch is part of pm_hifigan_workspace_bytes().                      */
size_t pm_walk_scratch_bytes(int batch);
/* torch.nn.utils.weight_norm fold: w = g * v / ||v||, rows x cols        */
int pm_fold_weight_norm(const float* g, const float* v, float* w, int rows,
                        int cols, void* stream);
/* (B, C, T) -> (B, T, c_pad) zero padded, and back                        */
int pm_to_channels_last(const float* src, float* dst, int batch, int channels,
                        int frames, int c_pad, void* stream);

/* ---- feature editing: promonet.edit (edit/core.py:17-132) -----------------
 * 1-D grid sampling (edit/grid.py:12-45) of `rows` sequences (rows, n_in) at
 * the fractional frame positions grid (n_out) (NULL = identity), with the
 * per-feature post-op fused: mode 0 linear, 1 linear in log2 then 2 ** y
 * (pitch, core.py:114), 2 nearest; then y = clip(y * scale + offset, lo, hi)
 * (pitch shift + clip core.py:121-125, loudness offset :128-129).          */
int pm_grid_sample(const float* seq, const float* grid, float* out, int rows,
                   int n_in, int n_out, int mode, float scale, float offset,
                   float lo, float hi, void* stream);
/* Selective time-stretch grid (edit/core.py:57-110, stretch_unvoiced /
 * stretch_silence off): ppg (ppg_rows, frames), `indices` (n) = rows of the
 * phonemes that ARE stretched (device int32; a row outside [0, ppg_rows) is
 * not read and turns the grid into NaN); writes selected (frames) = their
 * summed probability and grid (target_frames), the reference's sequential
 * fp32 recurrence whose step follows that probability. No selected mass (or
 * more unselected mass than target frames) yields a non-finite / decreasing
 * grid exactly as the reference's arithmetic does: callers check.           */
int pm_stretch_grid(const float* ppg, int ppg_rows, const int* indices,
                    int n_indices, float* selected, float* grid, int frames,
                    int target_frames, void* stream);

/* ---- FARGAN vocoder engine: replaces promonet.model.FARGAN ---------------
 * (promonet/model/fargan.py, selected by config/fargan.py MODEL = 'fargan').
 * One persistent workgroup per utterance; weight_dtype PM_F32, PM_F16 or
 * PM_FARGAN_MIXED is the STORAGE type of the streamed weights (math is fp32).
 * PM_FARGAN_MIXED keeps the conditioning network, the framewise conv, the
 * skip dense and the output layer (fargan.py:139-160, :349-372, :317-333) in
 * fp32 and stores the three GRU cells and every GLU gate (:212-223, :375-388)
 * - 80 % of the per-step weight stream - as f16.                            */
#define PM_FARGAN_MIXED 16
typedef struct pm_fargan_s* pm_fargan_t;
int pm_fargan_create(int num_features, int global_channels, int weight_dtype,
                     pm_fargan_t* out);
int pm_fargan_destroy(pm_fargan_t h);
/* FARGAN.state_dict() keys, e.g. conditioning_network.0.weight (371,371),
 * subframe_network.gru1.weight_ih (768,384),
 * subframe_network.skip_glu.gate.weight_g (256,1) / weight_v (256,256)     */
int pm_fargan_load_tensor(pm_fargan_t h, const char* name, const float* dev,
                          const int64_t* shape, int ndim, void* stream);
int pm_fargan_finalize(pm_fargan_t h, void* stream);
size_t pm_fargan_workspace_bytes(pm_fargan_t h, int batch, int frames);
/* FARGAN.forward(features, global_features, previous_samples)
 * (fargan.py:21-59): features (B, 114, T) - last channel = pitch period in
 * samples - or channels-last (B, T, 128) when features_cl != 0; global
 * (Bg, 258); previous (Bp, 512) or NULL (zeros); out (B, 1, 256 T).         */
int pm_fargan_forward(pm_fargan_t h, const float* features, int features_cl,
                      const float* global_features, int global_batch,
                      const float* previous_samples, int previous_batch,
                      float* out, int batch, int frames, void* workspace,
                      size_t workspace_bytes, void* stream);
/* Ragged batch (lengths (B) int32 frames on the device): the model is causal,
 * so out[b, :256 lengths[b]] equals utterance b synthesised alone, bit for
 * bit; the tail is zeros.                                                  */
int pm_fargan_forward_ragged(pm_fargan_t h, const float* features,
                             int features_channels_last,
                             const float* global_features, int global_batch,
                             const float* previous_samples, int previous_batch,
                             const int* lengths, float* out, int batch,
                             int frames, void* workspace,
                             size_t workspace_bytes, void* stream);

/* Kernel selection: 0 auto (clusters of 8 workgroups per utterance up to 160
 * utterances per launch, else one persistent workgroup per utterance),
 * 1 force the latter, 2 force clusters.                                     */
int pm_fargan_set_mode(pm_fargan_t h, int mode);
/* Debug / safety: synchronise and report whether an inter-workgroup exchange
 * of the last forward on `workspace` timed out: PM_ETIMEOUT (never expected
 * on a GPU this process has to itself; the audio of that forward is invalid
 * and the caller re-runs, e.g. after pm_fargan_set_mode(h, 1)).             */
int pm_fargan_check(pm_fargan_t h, int batch, int frames, void* workspace,
                    void* stream);

/* ---- preprocessing: promonet/preprocess/spectrogram.py, loudness.py ---- */
/* spectrogram.from_audio (spectrogram.py:15-60): reflect-pad 384, hann-1024
 * hop-256 framed DFT, sqrt(re^2 + im^2 + 1e-6): audio (B, N) ->
 * (B, 513, N / 256)                                                       */
/* (a 1024-point real FFT per frame in LDS; `scratch` is unused by it and may
 * be NULL - the argument stays for the ABI of rounds 1-2)                  */
size_t pm_stft_scratch_bytes(int batch, int samples);
int pm_stft_magnitude(const float* audio, float* out, int batch, int samples,
                      void* scratch, size_t scratch_bytes, void* stream);
/* The same spectrogram by the brute-force framed-DFT GEMM (exact-fp32 MFMA):
 * an independent cross-check of the FFT; scratch: pm_stft_scratch_bytes()  */
int pm_stft_magnitude_dft(const float* audio, float* out, int batch,
                          int samples, void* scratch, size_t scratch_bytes,
                          void* stream);
/* spectrogram.from_audio(audio, mels=True) (spectrogram.py:56-58 ->
 * linear_to_mel :111-133) in one kernel: log(basis @ magnitude), optional
 * clamp, straight from the FFT workgroup's LDS tile: audio (B, N) ->
 * (B, mels, N / 256). pm_stft_mel_prepare compacts the (mels, 513) basis
 * (librosa.filters.mel at spectrogram.py:118-121) into `prepared`
 * (pm_stft_mel_scratch_bytes(mels) bytes) once; the reference rebuilds its
 * basis on every call.                                                      */
size_t pm_stft_mel_scratch_bytes(int mels);
int pm_stft_mel_prepare(const float* basis, int mels, void* prepared,
                        size_t prepared_bytes, void* stream);
int pm_stft_mel(const float* audio, const void* prepared, float* out,
                int batch, int samples, int mels, int use_threshold,
                float log_threshold, void* stream);
/* Tuning knob: frames per FFT workgroup, 16 (default; two workgroups per CU)
 * or 32 (one 8-wave workgroup, 128-byte output rows). Per host thread; an API
 * call reads it once (both passes of pm_loudness use the same value).      */
int pm_stft_set_frames_per_group(int frames);
/* Schedule of pm_loudness with the default 8 bands: 2 (default) = a maximum
 * pass, then the band pass, two transforms per frame; 1 = OPTIMISTIC: the
 * first pass already writes the band means without a floor and the second
 * transforms only the 16-frame groups that have a bin under their utterance's
 * floor (the others are final, bit for bit). Pays on material at a steady
 * level (56 -> 47 us at batch 32 x 10 s of noise), costs up to + 25 % where
 * most groups hold a bin 80 dB under the maximum. Per host thread.           */
int pm_stft_set_loudness_passes(int passes);
/* The launch geometry one FFT transform would take for (batch, samples) on the
 * current device and host thread: `total_groups` (utterance, frame-group)
 * pairs walked by `workgroups` persistent workgroups (= min(total, occupancy x
 * CUs)); transform 1 = magnitude, 4 = log-mel, 2 / 3 / 5 = the loudness passes
 * (maximum, generic bands, the default 8 bands). Launches nothing.           */
int pm_stft_launch_info(int transform, int batch, int samples,
                        int* total_groups, int* workgroups);
/* spectrogram.linear_to_mel (spectrogram.py:111-133): log(basis @ spec),
 * optional clamp: spec (B, F, T), basis (Mel, F) -> (B, Mel, T)           */
int pm_linear_to_mel(const float* spec, const float* basis, float* out,
                     int batch, int bins, int mels, int frames,
                     int use_threshold, float log_threshold, void* stream);
/* Backward passes of the two functions above, for the training-side mel loss
 * that differentiates through spectrogram.from_audio(generated, True)
 * (promonet/train/core.py:277-305): grad_out (B, 513, N / 256) ->
 * grad_audio (B, N); grad_mel (B, Mel, T) -> grad_spec (B, F, T), `scratch`
 * = batch * mels * frames floats.                                          */
size_t pm_stft_backward_scratch_bytes(int batch, int samples);
int pm_stft_magnitude_backward(const float* audio, const float* grad_out,
                               float* grad_audio, int batch, int samples,
                               void* scratch, size_t scratch_bytes,
                               void* stream);
int pm_linear_to_mel_backward(const float* spec, const float* basis,
                              const float* grad_mel, float* grad_spec,
                              float* scratch, int batch, int bins, int mels,
                              int frames, int use_threshold,
                              float log_threshold, void* stream);
/* loudness.from_audio (loudness.py:17-55) per utterance: A-weighted dB with
 * the utterance-global (max - 80 dB) floor of librosa.amplitude_to_db, then
 * band_average (loudness.py:84-111): audio (B, N) -> (B, bands, N / 256);
 * two FFT passes over the audio (maximum, then bands), no dB tensor in HBM;
 * a_weights (513) = perceptual_weights() (loudness.py:149-160);
 * scratch: pm_loudness_scratch_bytes()                                     */
size_t pm_loudness_scratch_bytes(int batch, int samples);
int pm_loudness(const float* audio, const float* a_weights, float* out,
                int batch, int samples, int bands, float min_db,
                void* scratch, size_t scratch_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PROMONET_HIP_H */
