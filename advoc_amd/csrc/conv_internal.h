// Internal declarations shared by conv.hip, igemm.hip, wgrad.hip and edge.hip.
#pragma once
#include "common.h"
#include "igemm.h"

namespace advoc {

int validate_layer(const advoc_conv_layer* L);

// ---- thin layers (<= 2 input channels or 1 output channel): direct HBM-bound kernels ----
// Both take the SAME GatherGemmParams a gather-GEMM launch would get.
//   gather_dot  : n_total <= 2 outputs per grid point, K channels wide (wave per grid point)
//   gather_outer: K = c0 + c1 <= 2 input channels, n_total wide (thread per output element)
int launch_gather_dot(const GatherGemmParams& p, bool b_kn, hipStream_t stream,
                      const char** name_only = nullptr);
int launch_gather_outer(const GatherGemmParams& p, bool b_kn, hipStream_t stream,
                        const char** name_only = nullptr);

// Second stage of the two-stage path for <= 2 output channels (edge.hip): out = bias + sum over
// the taps of S[pixel @ tap][wtap * n_total + n], S being the per-pixel, per-tap partial dot
// products a pointwise MFMA GEMM produced.  `p` is the ORIGINAL problem (taps, phases, epilogue).
int launch_tap_sum(const GatherGemmParams& p, const float* S, int s_channels, hipStream_t stream,
                   const char** name_only = nullptr);

// The two-stage path in one launch (edge.hip, fused_taps_kernel): the per-pixel, per-tap partial dot products stay in LDS.
// `p` is the original problem; weights [tap][n][k] (k contiguous; for N == 1 the [tap][k][n] layout is the same).
bool fused_taps_ok(const GatherGemmParams& p);
int launch_fused_taps(const GatherGemmParams& p, hipStream_t stream, const char** name_only = nullptr);

// MFMA versions of the thin kernels (thin.hip): K = c0 + c1 <= 2 and taps * K <= 32, N % 32 == 0
int launch_thin_k_gemm(const GatherGemmParams& p, bool b_kn, hipStream_t stream,
                       const char** name_only = nullptr);

// ---- weight gradient ----
// An NHWC activation view (optionally the channel concat of two tensors) with the layer's fused
// input transform.
struct Operand {
  const float* p0;
  const float* p1;
  int c0, c1;
  int pitch0, pitch1;
  int h, w;            // rows per image, logical width
  int act;
  const float* scale;  // optional per-channel affine before act
  const float* shift;
  const uint8_t* mask; // optional {0,1} mask on the channels of source 0, indexed like p0
  float mask_scale;
};

// dw[tap][a][b] = sum over grid points g of P[g*s + d(tap)][a] * Q[g][b]
struct WgradParams {
  Operand P;   // gathered operand -> rows a
  Operand Q;   // grid operand     -> cols b
  int batch, gh, gw;
  int sy, sx;
  int ntaps;
  int tap[kMaxTaps];   // (dy & 0xff) | (dx & 0xff) << 8 | wtap << 16
  float* dw;           // accumulated with atomics; zero-filled by the launcher unless `accumulate`
  int accumulate;
  // thin_wgrad only (Q = the output gradient): the per-channel sums of the transformed Q values -- the layer's bias
  // gradient -- are taken on the way: added to qsum_table[kColsumReplicas][cb] (zeroed scratch), folded into qsum_out
  float* qsum_out;
  float* qsum_table;
  // wgrad_h3 only: scratch for the K slices' partial tiles (advoc_conv_layer.wgrad_ws); when it holds them all the slices
  // are stored plainly and summed IN SLICE ORDER by a second launch -- no zero fill, no atomics, a deterministic sum
  float* part_ws;
  int64_t part_ws_bytes;
};

int launch_wgrad_mfma(const WgradParams& p, hipStream_t stream,
                      const char** name_only = nullptr);   // both operands wide (% 32)
int launch_wgrad_thin(const WgradParams& p, hipStream_t stream,
                      const char** name_only = nullptr);   // P has <= 2 channels

// Operand-image weight gradient (wgrad_h3.hip): both operands as fp16 pair images.  *_make_image writes one operand's
// image (source 1 behind source 0 at its 256-byte-rounded size) and {amax, 2^-s} header; *_operand_bytes sizes it.
bool wgrad_h3_eligible(const WgradParams& p);
int64_t wgrad_h3_operand_bytes(const Operand& o, int batch, int64_t* b0, int64_t* b1);
int wgrad_h3_make_image(const Operand& o, int batch, uint16_t* img, unsigned* hdr, bool delayed, hipStream_t stream,
                        float* colsum0 = nullptr, float* colsum_table = nullptr);
int launch_wgrad_h3(const WgradParams& p, const uint16_t* p_img, const unsigned* p_hdr, const uint16_t* q_img,
                    const unsigned* q_hdr, hipStream_t stream, const char** name_only = nullptr);
int64_t wgrad_h3_partial_bytes(const WgradParams& p);      // what part_ws must hold for the atomics-free path (0: n/a)

int launch_wgrad_thin_mfma(const WgradParams& p, hipStream_t stream,
                           const char** name_only = nullptr);   // P <= 2 channels, Q % 32, on MFMA

// db[c] = sum over pixels of dy[., c] (* mask * scale)
int launch_bias_grad(const float* dy, const uint8_t* mask, float mask_scale, int64_t rows, int w,
                     int pitch, int c, float* db, int accumulate, hipStream_t stream);

}  // namespace advoc
