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

/* ------------------------------------------------------------------------------------------
 * Feature extractor
 * ---------------------------------------------------------------------------------------- */

/* |STFT| of a batch of mono waveforms.
 * Replaces tf.abs(tf.contrib.signal.stft(...)) reached from advoc/spectral.py:60-83 (stft_tf)
 * via advoc/loader.py:116-128 (magspec branch).
 *   wav     [batch, nsamps]                 float32
 *   window  [nfft]                          float32  (lws sqrt-Hann, advoc/spectral.py:44-57)
 *   mag     [batch, nframes, nfft/2+1]      float32
 * Frame t covers samples [t*nhop, t*nhop+nfft); samples >= nsamps read as zero (pad_end).
 * nfft must be 1024 (the only size the reference's tensor path is used with). */
int advoc_stft_mag_f32(const float* wav, int64_t batch, int64_t nsamps, const float* window,
                       int32_t nfft, int32_t nhop, int64_t nframes, float* mag,
                       advoc_stream_t stream);

/* Complex STFT, interleaved (re, im) float32 pairs: out [batch, nframes, nfft/2+1, 2].
 * Replaces tf.contrib.signal.stft at advoc/spectral.py:79. */
int advoc_stft_c64(const float* wav, int64_t batch, int64_t nsamps, const float* window,
                   int32_t nfft, int32_t nhop, int64_t nframes, float* out,
                   advoc_stream_t stream);

/* out[r, n] = sum_k x[r, k] * w[n, k]      (x @ w^T over the last dimension)
 * Replaces tf.tensordot / tf.matmul at models/advoc/spectral_util.py:29-32 (mag->mel,
 * w = mel filterbank [80,513]), :34-43 (mel->mag, w = pinv [513,80]) and
 * advoc/spectral.py:204-208.
 *   x [rows, k]   w [n, k]   out [rows, n]      all float32, dense row-major */
int advoc_matmul_nt_f32(const float* x, const float* w, float* out, int64_t rows, int32_t k,
                        int32_t n, advoc_stream_t stream);

/* In-place r9y9 dB normalisation of a linear mel spectrogram:
 *   v = clip((20*log10(max(min_level, v)) - ref_db - min_db) / -min_db, 0, 1)
 * Replaces advoc/spectral.py:210-225. */
int advoc_mel_dbnorm_f32(float* v, int64_t count, float min_level, float ref_db, float min_db,
                         advoc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ADVOC_HIP_H_ */
