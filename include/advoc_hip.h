/*
 * advoc_hip.h -- C ABI of libadvoc_hip.so, the MI355X (gfx950 / CDNA4) hot path of the
 * adversarial vocoder (AdVoc) train / inference pipeline.
 *
 * The reference (paarthneekhara/advoc) has no FFI of its own: its "operator API" for this
 * path is Python + TF1 graph ops.  Every entry point below therefore cites the reference
 * call site (path:line under /root/reference) whose TF1 / lws / librosa kernel it replaces.
 *
 * Conventions
 *   - all pointers are DEVICE pointers (HBM) unless the name ends in _host
 *   - the caller owns every buffer; nothing here allocates, frees or synchronises
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*)
 *   - tensors are NHWC, H = time frames, W = frequency bins, float32, like the reference
 *   - return value: ADVOC_OK (0) or a negative ADVOC_ERR_* code; never throws
 *   - re-entrant: no mutable global state
 */
#ifndef ADVOC_HIP_H_
#define ADVOC_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADVOC_OK 0
#define ADVOC_ERR_BAD_SHAPE (-1)   /* inconsistent or non-positive dimensions            */
#define ADVOC_ERR_UNSUPPORTED (-2) /* valid request outside what the kernels implement   */
#define ADVOC_ERR_HIP (-3)         /* a HIP runtime call / kernel launch failed          */
#define ADVOC_ERR_NULL (-4)        /* required pointer is NULL                           */

#define ADVOC_ABI_VERSION 1

typedef void* advoc_stream_t; /* hipStream_t */

/* activation applied to a conv INPUT on load (the reference rectifies the previous layer's
 * raw output right before each conv: advoc_model.py:109,138,155,187,196) */
#define ADVOC_ACT_NONE 0
#define ADVOC_ACT_LRELU02 1 /* max(0.2x, x)   advoc_model.py:86-87 */
#define ADVOC_ACT_RELU 2    /* tf.nn.relu     advoc_model.py:138   */

int advoc_abi_version(void);
const char* advoc_error_string(int code);
/* text of the last HIP runtime error this host thread hit inside the library (diagnostics for
 * ADVOC_ERR_HIP; thread-local, never reset) */
const char* advoc_last_hip_error(void);
/* name of the GPU arch the kernels in this library were compiled for ("gfx950") */
const char* advoc_target_arch(void);
/* The library reads its diagnostic environment switches (ADVOC_IGEMM_*, ADVOC_WGRAD_X6, ADVOC_X6D*: kernel
 * selection overrides for A/B measurements and tests) once, on first use; this re-reads them. */
void advoc_tuning_reload(void);

/* (r6) Diagnostic: the shader clock the dominant kernel of the train step (wgrad_h3_256_kernel) actually ran at.  Its first
 * workgroup adds {shader cycles, ticks of 10 ns, 1} of its life to three device counters on every launch; this call copies
 * them to out3_host (synchronously -- a measurement call, not part of the data path) and, with reset != 0, zeroes them.
 * clock [GHz] = 0.1 * out3[0] / out3[1].  (No counterpart in the reference: bench.py's `roofline.clock_ghz`.) */
int advoc_clock_probe_read(uint64_t* out3_host, int32_t reset);

/* ------------------------------------------------------------------------------------------
 * Feature extractor
 * ---------------------------------------------------------------------------------------- */

/* |STFT| of a batch of mono waveforms.
 * Replaces tf.abs(tf.contrib.signal.stft(...)) reached from advoc/spectral.py:60-83 (stft_tf)
 * via advoc/loader.py:116-128 (magspec branch).
 *   wav     [batch, nsamps]                 float32
 *   window  [nfft]                          float32  (lws sqrt-Hann, advoc/spectral.py:44-57)
 *   twiddle [nfft, 2]                       float32  {cos, sin}(2 pi e / nfft), from
 *                                           advoc_stft_twiddle_host (device copy owned by the caller)
 *   mag     [batch, nframes, nfft/2+1]      float32
 * Frame t covers samples [t*nhop, t*nhop+nfft); samples >= nsamps read as zero (pad_end).
 * nfft must be 1024 (the only size the reference's tensor path is used with); wav 16-byte aligned. */
int advoc_stft_mag_f32(const float* wav, int64_t batch, int64_t nsamps, const float* window,
                       const float* twiddle, int32_t nfft, int32_t nhop, int64_t nframes, float* mag,
                       advoc_stream_t stream);

/* Inverse STFT, lws(nfft, nhop, perfectrec=False).istft reached from advoc/spectral.py:300-309,
 * 320-321: irfft of every frame, synthesis window, overlap-add.
 *   spec        [batch, nframes, nfft/2+1] complex64 (interleaved re, im)
 *   window      [nfft] synthesis window (equal to the analysis window for the lws default)
 *   frames_work [batch, nframes, nfft] float32 scratch, caller-owned
 *   wav         [batch, (nframes-1)*nhop + nfft] float32, fully overwritten
 * nfft must be 1024, nhop a multiple of 4 and <= nfft. */
int advoc_istft_f32(const float* spec, int64_t batch, int64_t nframes, const float* window,
                    const float* twiddle, int32_t nfft, int32_t nhop, float* frames_work, float* wav,
                    advoc_stream_t stream);

/* One Griffin-Lim half step (advoc/spectral.py:306-307): the inverse STFT of |mag| * spec / |spec|
 * (phase 0 where spec == 0), the projection applied while the bins are loaded.  mag: [batch, nframes,
 * nfft/2+1] float32; everything else as advoc_istft_f32. */
int advoc_istft_project_f32(const float* spec, const float* mag, int64_t batch, int64_t nframes,
                            const float* window, const float* twiddle, int32_t nfft, int32_t nhop,
                            float* frames_work, float* wav, advoc_stream_t stream);

/* Griffin-Lim projection step (advoc/spectral.py:306-307): spec[i] <- |mag[i]| * spec[i] / |spec[i]|,
 * phase 0 where spec[i] == 0 (numpy: angle(0) == 0).  n complex elements, in place. */
int advoc_phase_project_c64(float* spec, const float* mag, int64_t n, advoc_stream_t stream);

/* out[i] = |spec[i]| for n complex64 elements (tf.abs, advoc/loader.py:128).  Used by the generic-nfft
 * STFT path (DFT as a matmul through advoc_matmul_nt_f32); the 1024-point kernel fuses it. */
int advoc_cabs_f32(const float* spec, float* out, int64_t n, advoc_stream_t stream);

/* Griffin-Lim initialisation (advoc/spectral.py:301-304): spec[i] = |mag[i]| * exp(2 pi i * unit_phase[i]),
 * unit_phase in [0, 1). */
int advoc_polar_c64(const float* mag, const float* unit_phase, float* spec, int64_t n, advoc_stream_t stream);

/* Local Weighted Sums phase reconstruction: lws.lws(nfft, nhop, mode='speech', perfectrec=False).run_lws(X) reached from
 * advoc/spectral.py:314-326 and models/advoc/spectral_util.py:45-50.  lws 1.2 is a third-party C++ library outside
 * /root/reference: these kernels restate the published algorithm (advoc_amd/csrc/lws.hip), parity unpinned.
 *   weights   [2Q-1][2L-1][period] complex64 (caller-computed, read-only), Q = nfft/nhop, period = nfft / gcd(nfft, nhop):
 *             alpha_q(p) * exp(-2 pi i r q nhop / nfft) with alpha_q(p) = 1/nfft sum_n awin[n] swin[n - q nhop]
 *             exp(2 pi i p n / nfft) -- the STFT o iSTFT projection kernel; r = (f + p) mod period
 *             THE TABLE MUST HAVE THIS FORM: for the reference geometry (nfft 1024, nhop 256: Q = 4, L = 5, period 4) the
 *             time-ordered pass and the batch sweeps read weights[q][p][0] only and apply the rotation (-i)^(q r) themselves
 *             (advoc_amd/csrc/lws.hip); a table of another form is honoured only with ADVOC_LWS_GENERIC=1 in the
 *             environment, which keeps every geometry on the kernels that read the whole table
 *   mean_mag  [clips]: mean magnitude of each clip (advoc_lws_mean_mag_f32); thresholds are multiples of it
 * advoc_lws_causal_c64: the time-ordered pass (no-future initialisation look_ahead frames ahead, with the host array
 *   nofuture_thresholds_host of nofuture_steps descending multiples -- the last must be 0 --, then online_iterations
 *   refinements per frame with thresholds online_alpha * exp(-online_beta * i)); writes spec [clips][nframes][nfft/2+1]
 *   complex64.  use_init != 0: spec holds starting phases (complex input of run_lws) and frames are not re-initialised.
 * advoc_lws_batch_c64: ONE sweep of the whole spectrogram from spec_in into spec_out (must differ): bins with
 *   mag > threshold * mean_mag get mag * phase(local weighted sum), the others are copied. */
int advoc_lws_mean_mag_f32(const float* mag, int64_t clips, int64_t per_clip, float* mean_mag, advoc_stream_t stream);
int advoc_lws_causal_c64(float* spec, const float* mag, const float* mean_mag, int64_t clips, int64_t nframes,
                         int32_t nfft, int32_t nhop, const float* weights, int32_t period, int32_t L,
                         int32_t look_ahead, const float* nofuture_thresholds_host, int32_t nofuture_steps,
                         int32_t online_iterations, float online_alpha, float online_beta, int32_t use_init,
                         advoc_stream_t stream);
int advoc_lws_batch_c64(const float* spec_in, float* spec_out, const float* mag, const float* mean_mag, int64_t clips,
                        int64_t nframes, int32_t nfft, int32_t nhop, const float* weights, int32_t period,
                        int32_t L, float threshold, advoc_stream_t stream);
/* All batch sweeps of run_lws in one call (thresholds_host[n_sweeps], multiples of mean_mag): spec_a = output of the
 * time-ordered pass on entry, the final spectrogram on return; spec_b = work buffer of the same size; tile_work = clips *
 * ceil(nframes / 8) floats of scratch (may be NULL: dense sweeps).  Non-increasing thresholds on the reference geometry
 * (nfft 1024, hop 256, L 5) run SPARSE: tiles of 8 frames none of whose bins exceeds the sweep's threshold are skipped. */
int advoc_lws_batch_sweeps_c64(float* spec_a, float* spec_b, const float* mag, const float* mean_mag, int64_t clips,
                               int64_t nframes, int32_t nfft, int32_t nhop, const float* weights, int32_t period,
                               int32_t L, const float* thresholds_host, int32_t n_sweeps, float* tile_work,
                               advoc_stream_t stream);

/* Host helper: fills tw_host[2 * nfft] with the double-precision-evaluated twiddle table the
 * STFT kernels expect (upload it once; it is read-only). */
int advoc_stft_twiddle_host(float* tw_host, int32_t nfft);

/* Complex STFT, interleaved (re, im) float32 pairs: out [batch, nframes, nfft/2+1, 2].
 * Replaces tf.contrib.signal.stft at advoc/spectral.py:79. */
int advoc_stft_c64(const float* wav, int64_t batch, int64_t nsamps, const float* window,
                   const float* twiddle, int32_t nfft, int32_t nhop, int64_t nframes, float* out,
                   advoc_stream_t stream);

/* out[r, n] = sum_k x[r, k] * w[n, k]      (x @ w^T over the last dimension)
 * Replaces tf.tensordot / tf.matmul at models/advoc/spectral_util.py:29-32 (mag->mel,
 * w = mel filterbank [80,513]), :34-43 (mel->mag, w = pinv [513,80]) and
 * advoc/spectral.py:204-208.
 *   x [rows, k]   w [n, k]   out [rows, n]      all float32, dense row-major */
int advoc_matmul_nt_f32(const float* x, const float* w, float* out, int64_t rows, int32_t k,
                        int32_t n, advoc_stream_t stream);

/* mel[r, m] = sum_k mag[r, k] * W[m, k]  and  inv[r, k] = sum_m mel[r, m] * inv_wt[m, k]  in one pass over mag:
 * the mag -> mel -> pseudo-inverted mag chain models/advoc/train_evaluate.py:55-56 builds for every training batch out
 * of models/advoc/spectral_util.py:29-43 (two tf.tensordot / tf.matmul launches there, two advoc_matmul_nt_f32 here
 * before).  The mel filterbank W [n_mels, bins] is triangular and is passed as its RUNS: band_lo_hi[m] = {first bin,
 * one past the last bin} with a non-zero weight in row m, mel_wp = the runs' weights W[m, lo .. hi) back to back, every
 * run padded with zeros to a multiple of 4 floats (zeros inside a run are multiplied, bins outside it skipped).
 *   mag [rows, bins]  mel_wp [packed_weights]  band_lo_hi [n_mels, 2] int32  inv_wt [n_mels, bins] (the pseudo-inverse
 *   [bins, n_mels] of spectral_util.py:26-27 TRANSPOSED: the kernel reads it along the bins)
 *   mel_out [rows, n_mels]  inv_out [rows, bins]        float32, dense row-major, device memory
 * packed_weights = floats in mel_wp (a multiple of 4).
 * bins = 513, n_mels = 80 (the reference's extractor) and packed_weights <= 2048; anything else:
 * ADVOC_ERR_UNSUPPORTED, use advoc_matmul_nt_f32 twice. */
int advoc_mel_pinv_f32(const float* mag, const float* mel_wp, const int32_t* band_lo_hi, const float* inv_wt,
                       float* mel_out, float* inv_out, int64_t rows, int32_t bins, int32_t n_mels,
                       int32_t packed_weights, advoc_stream_t stream);
/* The feature extractor of a training batch in ONE launch (advoc/loader.py:116-128 + models/advoc/spectral_util.py:29-43
 * as models/advoc/train_evaluate.py:55-56 chains them): wav [batch][nsamps] -> mag [batch][nframes][513] = |STFT| (as
 * advoc_stft_mag_f32), mel [batch][nframes][80] = mag W^T, inv [batch][nframes][513] = mel P^T (as advoc_mel_pinv_f32),
 * the magnitudes never re-read from HBM.  mel_wp / band_lo_hi / packed_weights as advoc_mel_pinv_f32.  pinv_pairs: the
 * pseudo-inverse pre-split for the f16 matrix cores, [17][5][2][64] blocks of 16 bytes (256-byte aligned): entry
 * [nb][st][plane][lane = 32 half + l32] = the 8 fp16 values `plane` (0: rounded, 1: remainder) of
 * P[n = 32 nb + l32][k = 16 st + 8 half + 0..7] * 2^s(n), zeros for n >= 513; pinv_unscale[544] = 2^-s(n), s(n) placing
 * the largest |P[n][:]| in [2^13, 2^14) (advoc_amd.spectral.pack_inverse_pairs builds both).  nfft 1024, 80 mel bands
 * only: ADVOC_ERR_UNSUPPORTED otherwise (callers fall back to the two launches). */
int advoc_stft_mel_pinv_f32(const float* wav, int64_t batch, int64_t nsamps, const float* window, const float* twiddle,
                            int32_t nfft, int32_t nhop, int64_t nframes, const float* mel_wp, const int32_t* band_lo_hi,
                            int32_t packed_weights, int32_t bins, int32_t n_mels, const void* pinv_pairs,
                            const float* pinv_unscale, float* mag, float* mel, float* inv, advoc_stream_t stream);

/* y[i] = tanh(x[i]) * scale + shift (x == y allowed).  Replaces tf.nn.tanh + feats_denorm at the end of
 * the MelspecGAN generator, models/melspecgan/conv2d.py:139 and util.py:11-12 (scale = shift = 0.5). */
int advoc_tanh_affine_f32(const float* x, float* y, int64_t n, float scale, float shift,
                          advoc_stream_t stream);

/* In-place r9y9 dB normalisation of a linear mel spectrogram:
 *   v = clip((20*log10(max(min_level, v)) - ref_db - min_db) / -min_db, 0, 1)
 * Replaces advoc/spectral.py:210-225. */
int advoc_mel_dbnorm_f32(float* v, int64_t count, float min_level, float ref_db, float min_db,
                         advoc_stream_t stream);


/* ------------------------------------------------------------------------------------------
 * Convolution stack (pix2pix generator + PatchGAN discriminator)
 * ---------------------------------------------------------------------------------------- */

/* NHWC float32 tensor view.  `w` is the LOGICAL width; rows are `w_pitch` pixels apart
 * (w_pitch >= w), images h * w_pitch pixels apart.  The reference's `[:, :, :-1, :]` trims
 * (advoc_model.py:137,154,156) are views with w = w_pitch - 1: never materialised. */
typedef struct advoc_tensor4 {
  float* p;
  int32_t n, h, w, c;
  int32_t w_pitch;
} advoc_tensor4;

#define ADVOC_CONV 0   /* y[oy,ox] = sum in[oy*sh - pad_t + ky, ox*sw - pad_l + kx] * w[ky,kx,ci,co] */
#define ADVOC_DECONV 1 /* y[iy*sh - pad_t + ky, ix*sw - pad_l + kx] += in[iy,ix] * w[ky,kx,co,ci]   */

/* One conv / transposed-conv layer of the reference graph, with everything the reference
 * applies around it folded in:
 *   input  = act( in_mask * in_mask_scale * (scale * concat_c(x0, x1) + shift) )
 *            (x1.p may be NULL; scale/shift NULL = none; in_mask covers the channels of x0 only)
 *   y      = (conv(input, w) + b) * drop_mask * drop_scale   (drop_mask NULL = no dropout)
 * scale/shift fold a batch normalisation of the PRODUCING layers into this layer's loads
 * (advoc_model.py:78-84,118,142); in_mask is the producer's dropout when it has to act AFTER that
 * normalisation (advoc_model.py:142-149) -- without BN the producer applies drop_mask itself.
 * ADVOC_CONV   : tf.layers.conv2d, kernel [kh,kw,cin,cout]  (advoc_model.py:25-32, 34-51);
 *                SAME padding is passed explicitly as pad_t/pad_l (bottom/right implied by y.h/y.w).
 * ADVOC_DECONV : tf.layers.conv2d_transpose, kernel [kh,kw,cout,cin], 4x4 stride 2 "same"
 *                == pad_t = pad_l = 1, y.h = 2*x.h; y.w <= 2*x.w clips the output (advoc_model.py:53-69,156).
 * x0 and x1 share n, h and the logical w; cin = x0.c + x1.c (skip concat, advoc_model.py:137,154;
 * discriminator input concat, :184).  drop_mask is uint8 {0,1} indexed exactly like y (tf.nn.dropout,
 * advoc_model.py:144-149; drop_scale = 1/keep_prob). */
typedef struct advoc_conv_layer {
  int32_t kind;
  int32_t kh, kw, sh, sw, pad_t, pad_l;
  int32_t in_act;
  advoc_tensor4 x0, x1;
  const float* in_scale;
  const float* in_shift;
  advoc_tensor4 y;
  const float* w;
  const float* b;
  const uint8_t* drop_mask;
  float drop_scale;
  const uint8_t* in_mask; /* uint8 {0,1}, indexed exactly like x0 */
  float in_mask_scale;
  /* optional scratch (caller-owned, 16-byte aligned).  Layers with <= 2 output channels (forward)
   * or <= 2 input channels (backward-data) run as a pointwise MFMA GEMM into this buffer followed
   * by a tap gather-sum when it holds at least advoc_conv_workspace_bytes(); without it they
   * use the slower direct kernel.  The gather-GEMM layers use it to balance a launch whose
   * workgroup count is not a whole number of rounds of the chip (the last tiles are cut into K
   * slices whose partial sums are parked here) and, for launches of >= ~450 row tiles, to hold
   * the weights split into three bf16 planes for the split-bf16 matrix path (every fp32 product
   * as six bf16 MFMA products with fp32 accumulation: fp32-level error, ~1.4x faster); without it
   * they launch unbalanced on the fp32 MFMA kernels.  Contents are scratch: nothing is kept
   * between calls, one buffer can serve every layer on a stream. */
  float* workspace;
  int64_t workspace_bytes;
  /* optional persistent operand images (caller-owned device memory, 256-byte aligned; NULL = none).  The image-based
   * kernels read fp16 pair images of the layer's input (after its fused transform) and of the output gradient; with
   * these buffers the forward call leaves the input image (and its 8-byte header {largest magnitude, 2^-s}) in x_img /
   * x_hdr and the backward-data call the image of dy in dy_img / dy_hdr, and the backward-weight call of the same step
   * reads them instead of making its own: img_flags bit 0 = x_img is current (the inputs have not changed since the
   * call that filled it), bit 1 = dy_img is current (filled by a call with the same dy).  A forward / backward-data call
   * that finds its bit set skips its image passes too (advoc_conv_make_image fills a buffer on its own).  Sizes:
   * advoc_conv_image_bytes(); headers 32 bytes each (8 words, zeroed by the caller once: {magnitude accumulator (0 between
   * calls), 2^-s, largest magnitude of the image in the buffer, values outside the one-pass head room, [3] before the last
   * pass, exact refits, arrival counter, reserved}).
   * Without the buffers the images live in `workspace` per call. */
  uint16_t* x_img;
  uint32_t* x_hdr;
  uint16_t* dy_img;
  uint32_t* dy_hdr;
  int32_t img_flags;
  /* optional: the bias gradient rides in the pass that builds the output-gradient image.  When non-null, a
   * backward-data call (or advoc_conv_make_image(which = 1)) that BUILDS dy_img also ADDS sum over the logical pixels
   * of dy (times the forward dropout mask, as advoc_conv_backward_bias) to db_fused[cout] -- the caller zeroes it
   * first if it does not accumulate, and skips advoc_conv_backward_bias.  Only where advoc_conv_bias_fusable() says
   * so and the backward-data call runs on the image kernels (advoc_conv_kernel_name); ignored otherwise. */
  float* db_fused;
  /* optional: float bits of max |w| over the whole kernel tensor, already on the device (advoc_segmented_amax_f32 after
   * every weight update).  The image kernels scale their weight image with it and skip their own magnitude pass over
   * the weights (one per forward / backward-data call otherwise).  MUST be current: a stale, smaller value overflows
   * the fp16 image. */
  const uint32_t* w_amax;
  /* optional: persistent fp16 pair images of the kernel for the forward ([0]) and the backward-data ([1]) call, and their
   * 16-byte headers, built for ALL layers of a network by ONE advoc_weight_images_f32 launch after a weight update
   * (sizes and shapes: advoc_conv_weight_image_desc).  When non-null the call builds no weight image of its own (54
   * small launches per AdVoc train step otherwise) and trusts the image to be CURRENT. */
  const uint16_t* w_img[2];
  const uint32_t* w_img_hdr[2];
  /* optional: ADVOC_WGRAD_TABLE_BYTES of device scratch OWNED BY THIS LAYER for the backward-weight call of a layer with
   * <= 2 input channels: when non-null and `db` is passed, the bias gradient rides in the weight-gradient kernel (which
   * reads every dy element exactly once) through a replica table kept here.  It is deliberately not part of
   * `workspace`: weight gradients may run on a stream of their own next to backward-data calls of other layers that
   * use the shared workspace. */
  float* wgrad_table;
  /* optional (forward calls): up to two CONSUMERS' operand images of this layer's output y, written by the forward
   * kernel's epilogue next to y itself (fp32 y is still written) -- the image the consuming layer's next forward call
   * would otherwise build with a pass over y: `img` = start of y's region in the consumer's x_img (source 0: x_img; source
   * 1: x_img + the 256-byte-rounded size of source 0's image), `hdr` = the consumer's x_hdr, `act` = the consumer's in_act.
   * Only under the one-pass (delayed) scale: the consumer's header must hold a previous image's magnitude, and the
   * consumer's next forward call must be made with ADVOC_IMG_X_CURRENT | ADVOC_IMG_X_EMITTED (it then runs the refit
   * check instead of an image pass).  Honoured by the image kernels only: advoc_conv_emits_images() says whether this
   * layer's forward call will write them; img == NULL: none.
   * (r5) `mode`: ADVOC_Y_BOUNDED -- the scale comes from a bound of |y| known before the launch (max|x| max|w| taps K +
   * max|b|) instead of a previous magnitude: the image is final, the consumer's next forward call is made with
   * ADVOC_IMG_X_CURRENT | ADVOC_IMG_X_BOUNDED (no refit check); with ADVOC_Y_IMAGE_ONLY the fp32 tensor y is NOT written --
   * for a tensor whose only reader is that one consumer: its backward-data call must then gate on the image
   * (ADVOC_IMG_X_GATES).  One consumer only; where advoc_conv_emits_images() == 2 (the <= 2-input-channel matrix kernel with
   * max |w| on the device). */
  struct { uint16_t* img; uint32_t* hdr; int32_t act; int32_t mode; } y_img[2];
  /* optional (backward-weight calls of the image kernels): device scratch for the partial tiles of the K slices the
   * pixel grid is cut into, advoc_conv_wgrad_ws_bytes() of it.  With it the slices are stored plainly and summed in
   * slice order by a second launch -- no zero fill of dw, no fp32 atomics (a quarter of the kernel's time on the large
   * layers), and a sum that does not depend on the order workgroups finish in.  May be SHARED by all layers whose
   * backward-weight calls run on one stream; like wgrad_table it is not part of `workspace`.  NULL: atomics. */
  float* wgrad_ws;
  int64_t wgrad_ws_bytes;
  /* optional (backward-data calls, r4): dx0 of this call IS the output gradient of the layer below, and its operand image
   * -- the one that layer's backward-data / backward-weight calls would otherwise build with a pass over dx0 -- is written
   * by this call's epilogue next to dx0 itself: `img` / `hdr` = the lower layer's dy_img / dy_hdr (header of the caller
   * role in use), under the one-pass scale (that header must hold a previous image's magnitude).  The lower layer's
   * backward-data call must then be made with ADVOC_IMG_DY_CURRENT | ADVOC_IMG_DY_EMITTED (refit check instead of an
   * image pass).  Since that pass also carried the lower layer's bias gradient (db_fused), the sums move here as well:
   * `colsum` non-null = sum over the logical pixels of dx0 is ADDED to colsum[c0] (the caller zeroes it first if it does
   * not accumulate), through `table`, ADVOC_WGRAD_TABLE_BYTES of device scratch owned by this layer.  Honoured where
   * advoc_conv_emits_dx_image() says so (the <= 2-output-channel layers' matrix kernel, one source, no accumulation);
   * img == NULL: none.
   * (r5) `mode`: ADVOC_DX_BOUNDED -- the scale of the image comes from a bound of |dx0| known BEFORE the launch
   *     |dx0| <= max |dy| * max |w| * (taps per output x output channels of this layer)  [+ *bound_add]
   * (placed at [2^14, 2^15): no value can leave the fp16 range), so the image is final when the call returns: no previous
   * magnitude is needed in `hdr`, no refit check follows, and the lower layer's calls are made with ADVOC_IMG_DY_CURRENT |
   * ADVOC_IMG_DY_BOUNDED.  hdr[0] receives the largest magnitude written (zeroed by this call first), hdr[1] = 2^-s.
   * With ADVOC_DX_IMAGE_ONLY the fp32 tensor dx0 is NOT written at all (the pointer still gives the geometry): the image,
   * the column sums and the magnitude are the only outputs -- the lower layer's image pass (a read and a write of the
   * whole tensor) disappears without the epilogue storing one byte more than it did.  Honoured by the patch kernels'
   * backward-data launches on grids without remainder columns (advoc_conv_emits_dx_image() == 3), without dropout
   * mask on either destination and without accumulation -- except accum0 together with `bound_add` (device): float bits of a
   * bound of |what dx0 already holds| (advoc_conv_layer.dx1_amax of the call that wrote it), four-phase and per-tap launches. */
  struct { uint16_t* img; uint32_t* hdr; float* colsum; float* table; int32_t mode; int32_t reserved;
           const uint32_t* bound_add; } dx_img;
  /* optional (backward-data calls of the image kernels on a two-source layer, r5): the largest |value| the call writes to dx1 is
   * raised here (float bits, device; the caller zeroes the word first).  It is the `bound_add` of the LATER call that
   * accumulates into the same tensor and writes its image: a decoder's skip gradient arrives first, the encoder's
   * backward-data pass adds its own and -- with dx_img under ADVOC_DX_BOUNDED | ADVOC_DX_IMAGE_ONLY and accum0 -- leaves the
   * sum as the lower encoder's output-gradient image only. */
  uint32_t* dx1_amax;
} advoc_conv_layer;
#define ADVOC_DX_BOUNDED 1
#define ADVOC_DX_IMAGE_ONLY 2
#define ADVOC_Y_BOUNDED 1
#define ADVOC_Y_IMAGE_ONLY 2
#define ADVOC_WGRAD_TABLE_BYTES 262144

/* amax_out[i] = float bits of max |base[offsets[i] .. offsets[i] + sizes[i])| for `count` tensors of one arena, in one
 * launch (offsets / sizes / amax_out in device memory; amax_out is zeroed first).  Feeds advoc_conv_layer.w_amax. */
int advoc_segmented_amax_f32(const float* base, const int64_t* offsets, const int64_t* sizes, int32_t count,
                             uint32_t* amax_out, advoc_stream_t stream);

/* out5_host = {taps, n_total, ktot, b_kn, bytes} of the weight image the forward (direction 0) / backward-data (1) call of
 * this layer reads; bytes = 0 when that direction does not run on the image kernels (host-side, no launch). */
int advoc_conv_weight_image_desc(const advoc_conv_layer* layer, int32_t direction, int64_t* out5_host);

/* Builds `count` weight images in one launch.  table (device): 8 int64 per image {offset of the kernel tensor from `base`
 * in elements, taps, n_total, ktot, b_kn (the first four numbers of advoc_conv_weight_image_desc), index of the tensor in
 * `amax` (advoc_segmented_amax_f32's output, already current on `stream`), byte offset of the image in `pool` (256-byte
 * aligned), index of its 4-word header in `hdrs`}.  Feeds advoc_conv_layer.w_img / w_img_hdr. */
int advoc_weight_images_f32(const float* base, const uint32_t* amax, const int64_t* table, int32_t count, void* pool,
                            uint32_t* hdrs, advoc_stream_t stream);
/* (r5) The same with ADVOC_WEIGHT_HDR_L1_WORDS-word headers (index of table column 7 in units of that many words; the headers
 * of ALL `count` images are zeroed first, so the table must use indices 0 .. count - 1): besides {max |w|, 2^-s} in words 0, 1 a
 * header receives word 2 = taps (0: none of the following; images of more than 16 taps), word 3 = K and words 4 .. 4 + taps =
 * float bits of max over the image's rows n of sum_k |w[tap][n][k]| (rounded up) -- the per-tap factors of the a-priori bounds
 * the launches that write an operand image under ADVOC_DX_BOUNDED / ADVOC_Y_BOUNDED use instead of max |w| * K (about 5 x
 * tighter: two more bits for the small values of the image); pass ADVOC_IMG_W_L1 with such headers. */
#define ADVOC_WEIGHT_HDR_L1_WORDS 32
int advoc_weight_images_l1_f32(const float* base, const uint32_t* amax, const int64_t* table, int32_t count, void* pool,
                               uint32_t* hdrs, advoc_stream_t stream);

/* 1: the layer's output-gradient image pass can carry the bias gradient (db_fused above): dy_img present and cout such
 * that a thread of the image pass keeps one group of 8 channels (32 <= cout <= 1024, 256 % (cout / 8) == 0) */
int advoc_conv_bias_fusable(const advoc_conv_layer* layer);
/* != 0: a forward call on this layer writes the consumers' images of advoc_conv_layer.y_img (it runs on the image kernels
 * and has the workspace they need; 2: ADVOC_Y_BOUNDED / ADVOC_Y_IMAGE_ONLY are honoured too); 0: y_img is ignored and the
 * consumers must build their images themselves */
int advoc_conv_emits_images(const advoc_conv_layer* layer);
/* 1: this layer's backward-data call can gate its activation gradient on the signs of x_img (ADVOC_IMG_X_GATES): a patch
 * kernel on a grid without remainder columns, inputs with an activation and without batch-norm affine / dropout mask */
int advoc_conv_gates_on_image(const advoc_conv_layer* layer);
/* != 0: a backward-data call on this layer writes advoc_conv_layer.dx_img (2: the <= 2-output-channel matrix kernel, one-pass
 * scale; 3: a patch kernel, ADVOC_DX_BOUNDED [| ADVOC_DX_IMAGE_ONLY] required); 0: dx_img is refused */
int advoc_conv_emits_dx_image(const advoc_conv_layer* layer);

#define ADVOC_IMG_X_CURRENT 1
#define ADVOC_IMG_DY_CURRENT 2
/* the header of the buffer holds the largest magnitude of an image written to it before (an earlier call on data of the
 * same scale, e.g. the previous train step): the image may be built in ONE pass with the scale derived from that
 * magnitude (2^6 of head room) instead of a magnitude pass followed by the image pass.  When a value leaves the head
 * room (counted in header word 3), the tensor has shrunk by more than 2^6 or the header holds no usable magnitude, the
 * call rebuilds the image with the exact scale on the device (always-launched refit kernel, counted in header word 5):
 * the result never depends on a clamped operand. */
#define ADVOC_IMG_X_DELAYED 4
#define ADVOC_IMG_DY_DELAYED 8
/* with ADVOC_IMG_X_CURRENT: x_img was written by the producers of the inputs (advoc_conv_layer.y_img of their layers)
 * under the one-pass scale since the last forward call: this call runs the refit check / header rotation first */
#define ADVOC_IMG_X_EMITTED 16
/* with ADVOC_IMG_DY_CURRENT: dy_img was written by the backward-data call of the layer above (its advoc_conv_layer.dx_img)
 * under the one-pass scale: this call runs the refit check / header rotation first */
#define ADVOC_IMG_DY_EMITTED 32
/* with ADVOC_IMG_DY_CURRENT: dy_img was written by the layer above under ADVOC_DX_BOUNDED: final as it stands (dy_hdr[0] =
 * its largest magnitude, dy_hdr[1] = 2^-s), no refit check, no header rotation */
#define ADVOC_IMG_DY_BOUNDED 64
/* with ADVOC_IMG_X_CURRENT: x_img was written by the producer under ADVOC_Y_BOUNDED: final as it stands, no refit check */
#define ADVOC_IMG_X_BOUNDED 128
/* backward-data calls: the fp32 inputs x0 / x1 were never written (ADVOC_Y_IMAGE_ONLY producers); the activation gradient is
 * gated by the sign of x_img's high plane.  Patch kernels only (ADVOC_ERR_UNSUPPORTED elsewhere). */
#define ADVOC_IMG_X_GATES 256
/* w_img_hdr[] are ADVOC_WEIGHT_HDR_L1_WORDS-word headers filled by advoc_weight_images_l1_f32 (per-tap row-L1 maxima for the
 * a-priori bounds); without the flag the bounds use max |w| * taps * K */
#define ADVOC_IMG_W_L1 512

/* Bytes of the persistent operand image `which` (0: inputs, 1: output gradient) the layer can use; 0 when the layer's
 * shapes are outside the image-based kernels. */
int64_t advoc_conv_image_bytes(const advoc_conv_layer* layer, int32_t which);
/* Builds the persistent operand image `which` (0: x_img from the layer's inputs; 1: dy_img from `dy`) on its own: the
 * magnitude pass and the fp16 pair image pass of advoc_amd/csrc/image.hip.  ADVOC_ERR_NULL without the buffer. */
int advoc_conv_make_image(const advoc_conv_layer* layer, int32_t which, const float* dy, advoc_stream_t stream);

/* Scratch the layer can use for `direction` (0 forward, 1 backward-data, 2 backward-weight); 0 when it needs none. */
int64_t advoc_conv_workspace_bytes(const advoc_conv_layer* layer, int32_t direction);
/* bytes of advoc_conv_layer.wgrad_ws this layer's backward-weight call can use (0: it would not use any) */
int64_t advoc_conv_wgrad_ws_bytes(const advoc_conv_layer* layer);

/* Forward.  Replaces TF Conv2D / Conv2DBackpropInput(+BiasAdd, activations, concat, dropout)
 * built at advoc_model.py:89-158 (generator) and :184-202 (discriminator). */
int advoc_conv_forward(const advoc_conv_layer* layer, advoc_stream_t stream);

/* Gradient w.r.t. the layer's inputs AFTER their (optional) scale/shift: dx0 / dx1 have the
 * geometry of x0 / x1.
 *   dx = in_mask * in_mask_scale * act'(scale * x + shift) * conv_backward_data(dy * drop_mask * drop_scale, w)
 * accum != 0 adds into the destination (an encoder output receives gradient from its next
 * encoder AND its skip decoder).  dx1 may be NULL when x1 is absent or its gradient is unwanted;
 * dx0 may be NULL likewise.  Pixels of dx0/dx1 at columns >= logical w are NOT written.
 * Replaces the TF gradient ops of the same graph (tf.gradients via AdamOptimizer.minimize,
 * advoc_model.py:254-257). */
int advoc_conv_backward_data(const advoc_conv_layer* layer, const float* dy, float* dx0, float* dx1,
                             int32_t accum0, int32_t accum1, advoc_stream_t stream);

/* Bias gradient alone: db[co] = sum over pixels of dy * drop_mask * drop_scale (BiasAddGrad). */
int advoc_conv_backward_bias(const advoc_conv_layer* layer, const float* dy, float* db,
                             int32_t accumulate, advoc_stream_t stream);

/* Diagnostics: name of the kernel template instance a call on `layer` launches, e.g.
 * "gather_gemm_kernel<2, 2, 2, 2, true>" -- the string rocprofv3 shows for it.
 * direction: 0 forward, 1 backward-data, 2 backward-weight.  buf_host is HOST memory. */
int advoc_conv_kernel_name(const advoc_conv_layer* layer, int32_t direction, char* buf_host,
                           int32_t buf_len);

/* Gradient w.r.t. kernel and bias: dw has the layout of layer->w, db is [cout] (NULL = skip).
 * accumulate == 0 overwrites dw / db, != 0 adds to them (a variable shared by two passes, e.g. the
 * discriminator on real and on fake inputs).  Replaces Conv2DBackpropFilter / BiasAddGrad.
 * db: for layers with 1-2 input channels (encoder_1, discriminator layer_1) and a workspace the per-channel sums are
 * taken inside the weight-gradient kernel, which reads every dy element exactly once anyway; elsewhere a separate pass
 * over dy (or pass NULL and use advoc_conv_layer.db_fused / advoc_conv_backward_bias). */
int advoc_conv_backward_weight(const advoc_conv_layer* layer, const float* dy, float* dw, float* db,
                               int32_t accumulate, advoc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Batch normalisation (use_batchnorm=True; always batch statistics, advoc_model.py:77-84,173-177)
 * ---------------------------------------------------------------------------------------- */

/* Statistics of z [npix, c] (dense NHWC, c % 4 == 0) and the affine a consuming conv folds into its
 * loads:  mean, invstd = 1/sqrt(biased var + epsilon);  scale = gamma * invstd;
 * shift = beta - mean * scale.   work: 4*c floats of scratch (8-byte aligned; sums are combined in
 * double).  The normalised tensor itself is never materialised. */
int advoc_bn_forward(const float* z, int64_t npix, int32_t c, const float* gamma, const float* beta,
                     float epsilon, float* scale, float* shift, float* mean, float* invstd, float* work,
                     advoc_stream_t stream);

/* In place g: dL/dy -> dL/dz for y = gamma * (z - mean) * invstd + beta with batch statistics;
 * dgamma = sum g * (z - mean) * invstd, dbeta = sum g (overwritten, or added to when
 * accumulate != 0).  work: 4*c floats, 8-byte aligned. */
int advoc_bn_backward(const float* z, float* g, int64_t npix, int32_t c, const float* gamma,
                      const float* mean, const float* invstd, float* dgamma, float* dbeta,
                      int32_t accumulate, float* work, advoc_stream_t stream);

/* Split forms of the two calls above, for batch norm whose statistics span several replicas
 * (data parallelism): *_stats leaves 2*c doubles in `work` -- forward: (sum z, sum z^2); backward:
 * (sum g, sum g (z - mean)) -- which the caller adds up across replicas (e.g. an all-reduce of `work`
 * viewed as float64) before calling *_finalize / *_apply with the GLOBAL pixel count.  The parameter
 * gradients written by advoc_bn_backward_stats are this replica's contribution.  With count_total ==
 * npix and no exchange they reproduce advoc_bn_forward / advoc_bn_backward. */
int advoc_bn_forward_stats(const float* z, int64_t npix, int32_t c, float* work, advoc_stream_t stream);
int advoc_bn_forward_finalize(const float* work, int64_t count_total, int32_t c, const float* gamma,
                              const float* beta, float epsilon, float* scale, float* shift, float* mean,
                              float* invstd, advoc_stream_t stream);
int advoc_bn_backward_stats(const float* z, const float* g, int64_t npix, int32_t c, const float* mean,
                            const float* invstd, float* dgamma, float* dbeta, int32_t accumulate, float* work,
                            advoc_stream_t stream);
int advoc_bn_backward_apply(const float* z, float* g, int64_t npix, int32_t c, const float* gamma,
                            const float* mean, const float* invstd, const float* work, int64_t count_total,
                            advoc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Losses, optimiser, dropout masks
 * ---------------------------------------------------------------------------------------- */

/* Discriminator loss (advoc_model.py:238) on raw layer_5 logits (the sigmoid of :201 is fused):
 *   loss_sum[0] = sum_i -(log(sigmoid(zr_i) + 1e-12) + log(1 - sigmoid(zf_i) + 1e-12))
 *   dlogit_* = d(mean loss)/dz  (NULL = not wanted).  discrim_loss = loss_sum[0] / n. */
int advoc_gan_d_loss(const float* logit_real, const float* logit_fake, int64_t n, float* dlogit_real,
                     float* dlogit_fake, float* loss_sum, advoc_stream_t stream);

/* Generator loss (advoc_model.py:239-245):  gan_weight * mean(-log(sigmoid(zf) + 1e-12))
 *                                          + l1_weight * mean(|target - gen|)
 *   loss_sums[0] = sum -log(sigmoid(zf)+1e-12), loss_sums[1] = sum |target - gen|
 *   dlogit_fake  = d gen_loss / d zf;  dgen (+)= l1 part of d gen_loss / d gen (accum_dgen != 0 adds).
 * logit_fake may be NULL (gan_weight <= 0, or the eval L1 metric of train_evaluate.py:143). */
int advoc_gan_g_loss(const float* logit_fake, int64_t n_logits, const float* gen, const float* target,
                     int64_t n_spec, float gan_weight, float l1_weight, float* dlogit_fake, float* dgen,
                     int32_t accum_dgen, float* loss_sums, advoc_stream_t stream);

/* One TF-style Adam step over a flat parameter arena (tf.train.AdamOptimizer, advoc_model.py:250-257):
 *   g' = grad_scale * g;  m = b1 m + (1-b1) g';  v = b2 v + (1-b2) g'^2;
 *   param -= lr_t * m / (sqrt(v) + epsilon),   lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t) from the caller.
 * All four arrays must be 16-byte aligned. */
int advoc_adam_tf_f32(float* param, const float* grad, float* m, float* v, int64_t count, float lr_t,
                      float beta1, float beta2, float epsilon, float grad_scale, advoc_stream_t stream);

/* tf.nn.dropout keep mask (advoc_model.py:144-149): mask[i] = floor(keep_prob + u_i) in {0,1},
 * u_i = Philox-4x32-10(seed, offset + i) -- a pure function of (seed, offset + i), so a global
 * batch sharded over GPUs draws the same mask as the unsharded batch.  offset % 4 == 0. */
/* prob[i] = 1 / (1 + exp(-logits[i])): the discriminator's output activation (models/advoc/advoc_model.py:201) for
 * Advoc.build_discriminator; the train step never materialises it (fused into the loss kernels above) */
int advoc_sigmoid_f32(const float* logits, float* prob, int64_t count, advoc_stream_t stream);

/* dst[0 .. count) = 0 (dst 16-byte aligned): the gradient arenas at the start of a D / G update -- tf.gradients
 * (models/advoc/advoc_model.py:254-257) builds fresh sums every step, the weight- and bias-gradient kernels here accumulate */
int advoc_zero_f32(float* dst, int64_t count, advoc_stream_t stream);

int advoc_dropout_mask_u8(uint8_t* mask, int64_t count, uint64_t seed, uint64_t offset, float keep_prob,
                          advoc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ADVOC_HIP_H_ */
