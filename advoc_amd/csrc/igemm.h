// Parameter block of the gather-GEMM ("implicit GEMM") kernels shared by conv.hip (host-side
// translation of advoc_conv_layer) and igemm.hip (the gfx950 MFMA kernel).
//
// GEMM view of every conv / transposed-conv direction in the model:
//   rows  m  = (image, gy, gx) over a grid of batch x gh x gw points
//   cols  n  = output channel
//   depth k  = (tap, input channel);  A[m, k] = act(in[image, gy*sy + dy[tap], gx*sx + dx[tap], ci])
//                                      (zero outside the logical input), B[k, n] = w_tap[ci][n]
//   result row m is stored at pixel (gy*osy + ooy[phase], gx*osx + oox[phase]) of the destination.
// Stride-2 transposed convolutions (decoder forward, encoder backward-data) run as 4 sub-pixel
// phases (blockIdx.z), each a dense 2x2-tap GEMM -- no zero-insertion, no wasted MACs.
#pragma once
#include <stdint.h>

namespace advoc {

constexpr int kMaxPhases = 4;
constexpr int kMaxTaps = 16;

// where one consumer wants the image of a forward launch's output d[0] (image_emit.h): the start of the tensor's region in
// the consumer's operand image, the operand's rotated 8-word header ([2] = previous magnitude, [0] = accumulator, [3] =
// out-of-window count) and the slope of the consumer's activation (1 none, 0.2 leaky ReLU, 0 ReLU); img == null: none
struct ImgOut {
  uint16_t* img;
  unsigned* hdr;
  float slope;
};

struct GemmDest {
  float* p;           // destination tensor (NHWC, `c` channels, rows `pitch` pixels apart)
  const float* xpre;  // same geometry: pre-activation forward value, for act'(x) gating (or null)
  int pitch;
  int c;
  int accum;          // 1: add into destination
  // act'(.) is evaluated at gscale * xpre + gshift (the BN affine the forward applied to this
  // input, per channel of THIS destination; null = identity) and the result is multiplied by
  // gmask * gmask_scale (the dropout mask the forward applied to this input; null = none)
  const float* gscale;
  const float* gshift;
  const uint8_t* gmask;
  float gmask_scale;
  // (r5) != null: the tensor behind xpre was never written -- its producer left only the operand image of it (the consuming
  // layer's x_img: act(x) 2^s as fp16 pairs) -- and the activation gradient is gated by the SIGN of the image's high plane
  // (h0 > 0 <=> x > 0 up to values below 2^-25 of the scaled range): 2 bytes per value instead of 4.  Patch kernels only.
  const uint16_t* ximg;
};

struct GatherGemmParams {
  // ---- A operand ----
  const float* a0;
  const float* a1;
  int c0, c1;              // channels per source; c1 == 0 -> single source
  int a0_pitch, a1_pitch;  // row pitch in pixels
  int a_h;                 // rows per image (physical == logical)
  int in_h, in_w;          // logical gather bounds
  const float* in_scale;   // optional per-channel affine applied before the activation
  const float* in_shift;
  int in_act;
  const uint8_t* a_mask;   // optional {0,1} mask multiplied into the channels of SOURCE 0 (indexed like a0)
  float a_mask_scale;
  // ---- grid / taps ----
  int batch, gh, gw;
  int gx_off;              // the launch covers grid columns [gx_off, gx_off + gw) (igemm_h3.hip only; 0 elsewhere)
  int sy, sx;
  int nphase, ntaps;
  int tap[kMaxPhases][kMaxTaps];  // (dy & 0xff) | (dx & 0xff) << 8 | wtap << 16
  // ---- B operand ----
  const float* w;
  const unsigned* w_amax;  // optional: float bits of max |w| already on the device (advoc_segmented_amax_f32)
  const uint16_t* w_img;   // optional: the CURRENT fp16 pair image of w for this direction (advoc_weight_images_f32) and
  const unsigned* w_img_hdr;   // its 4-word header ([1] = 2^-s): the call builds no weight image
  int w_img_l1;            // != 0: w_img_hdr is an ADVOC_WEIGHT_HDR_L1_WORDS-word header (advoc_weight_images_l1_f32)
  int n_total;
  int k_order;             // 0: channel slices inner, taps outer; 1: taps inner
  int n_valid;             // 0 = n_total; else only the first n_valid columns exist in w / are stored
  // ---- output ----
  int osy, osx;
  int ooy[kMaxPhases], oox[kMaxPhases];
  int out_h, out_w;
  GemmDest d[2];
  int n_split;             // channels [0, n_split) -> d[0], the rest -> d[1]
  const float* bias;
  const uint8_t* y_mask;   // forward dropout mask, indexed like d[0]
  float y_mask_scale;
  int grad_act;            // != ADVOC_ACT_NONE: multiply result by act'(d[i].xpre)
  // ---- split-bf16 path (filled in by the launcher): the weights pre-split into three bf16 planes,
  // wq[plane][tap][n_total][c0 + c1] (contraction axis contiguous, zeros for n >= n_valid) ----
  const uint16_t* wq;
  int wq_taps;
  // ---- operand-image path (igemm_h3.hip, filled in by its launcher): the A sources as arrays of 128-byte K slices,
  // [pixel][channel / 32][plane][32] fp16 with the tensor's own geometry (image.hip); wq is then the fp16 pair
  // weight image [tap][n][k / 32][plane][32]; a_hdr / b_hdr: {largest magnitude bits, 2^-s} per operand ----
  const uint16_t* a0_img;
  const uint16_t* a1_img;
  int a0_img_bytes, a1_img_bytes;
  const unsigned* a_hdr;
  const unsigned* b_hdr;
  // caller-provided home of the A image / header (persistent per-layer buffers); null: the launch workspace
  uint16_t* a_img_out;
  unsigned* a_hdr_out;
  int a_img_current;       // != 0: a_img_out / a_hdr_out already hold this operand's image (skip the image passes)
  int a_img_delayed;       // != 0: a_hdr_out holds the magnitude of a previous image of this operand: one-pass image
  int a_img_bounded;       // != 0 (with a_img_current): written by the layer above under an a-priori scale (oimg_bounded
                           // below): final -- no refit check, header word 0 = its largest magnitude
  int a_img_emitted;       // != 0 (with a_img_current): the image was written by the producers' epilogues under the delayed
                           // scale (image_emit.h): run the refit check (and the header rotation) before reading it
  // forward launches: up to two consumers' images of the OUTPUT d[0], written by the epilogue (image_emit.h); honoured by
  // the image kernels' non-atomic epilogues only (launch_gather_gemm_h3 reports through emit_report what it will do)
  ImgOut oimg[2];
  int* emit_report;        // host pointer, name_only launches: set to 1 when this launch would write oimg
  // backward-data launches of the thin matrix kernel (thin.hip): per-channel sums of destination 0 over its pixels, added to
  // ocolsum_out[c] through the replica table ocolsum_table (kColsumBytes, zeroed by the launcher) -- the bias gradient of
  // the layer below when oimg[0] is that layer's output-gradient image
  float* ocolsum_out;
  float* ocolsum_table;
  // (r5) backward-data launches of the PATCH kernels (igemm_patch.hip) under an A-PRIORI scale: oimg[0] is the output-gradient
  // image of the layer below and its scale comes from a bound of |dx| known before the launch --
  //     |dx| <= max|dy| * max|w| * (taps per output * K)  (+ obound_add: a bound of what is already in the destination)
  // -- placed at [2^14, 2^15): no value can leave the fp16 range, so there is no history, no refit check and no need for
  // the fp32 tensor to rebuild the image from; the largest magnitude actually written is raised in oimg[0].hdr[0] (zeroed
  // by the launcher), 2^-s goes to hdr[1].  a_amax / b_hdr[0] = float bits of max |A operand| / max |w| on the device.
  int oimg_bounded;
  const unsigned* w_l1;    // persistent weight header with per-tap row-L1 maxima (advoc_weight_images_l1_f32: word 2 = taps,
                           // word 3 = K, words 4.. = max_n sum_k |w[tap][n][k]|), or null: the bound uses max|w| * taps * K
  const unsigned* a_amax;
  const unsigned* obound_add;
  int d0_no_store;         // != 0: destination 0 exists as the image only, its fp32 tensor is not written
  unsigned* d1_amax_out;   // != null (backward-data): max |value written to destination 1| is raised here (float bits; zeroed by the
                           // caller) -- the bound of "what the destination already holds" for the launch that ACCUMULATES into that
                           // tensor later and writes its image (obound_add of that launch: a decoder's skip gradient, finished by
                           // the encoder's backward-data pass)
  float* a_colsum;         // != null: the image pass of source 0 adds its per-channel sums over the logical pixels here (the
                           // bias gradient, when A is an output gradient); only honoured where image_colsum_ok(c0)
  // ---- tail split (filled in by the launcher, see launch_cfg) ----
  int tail_main;           // > 0: 1-D launch; tiles [0, tail_main) whole, the rest in tail_split K slices each
  int tail_split;
  float* tail_ws;          // [tail tiles][tail_split][BM * BN] partial accumulators
  int* tail_cnt;           // [tail tiles] arrival counters (zero before and after every launch)
};

// B_KN = true : weights stored [tap][K][N] (N contiguous)   conv fwd, deconv bwd-data
// B_KN = false: weights stored [tap][N][K] (K contiguous)   deconv fwd, conv bwd-data
// name_only != nullptr: do not launch, just report the kernel instance that would run.
// scratch / scratch_bytes: optional caller workspace for the tail split (see launch_cfg); without it
// the launch is a plain one.  scratch_query != nullptr: do not launch, report the bytes wanted.
int launch_gather_gemm(const GatherGemmParams& p, bool b_kn, hipStream_t stream,
                       const char** name_only = nullptr, float* scratch = nullptr,
                       int64_t scratch_bytes = 0, int64_t* scratch_query = nullptr);

// ---- shared launch helpers (igemm.hip) ----
int device_cu_count();
int persistent_cu_count();    // device_cu_count() minus ADVOC_RESERVE_CUS, a multiple of 8
// taps / columns / K of the weight image a gather launch reads (igemm_h3.hip); false: not on the image kernels
bool h3_weight_image_shape(const GatherGemmParams& p, int* taps, int* n_total, int* ktot);
// A launch of T equal workgroups on C compute units costs ceil(T / C) rounds when they are all resident: the last
// T mod C tiles are cut into `split` K slices each so that the extra round is 1 / split of a tile long.
struct TailPlan { int main = 0, rem = 0, split = 0; };
TailPlan plan_tail(int64_t tiles, int nkt);
int* tail_counter_slot();

// Operand-image variant (igemm_h3.hip): same problem description; the launcher writes the weight image and the
// activation image(s) into the caller workspace first.  ADVOC_ERR_UNSUPPORTED when the problem is outside what
// it takes (too small, channel counts, no / too little workspace): the caller falls back to launch_gather_gemm.
int launch_gather_gemm_h3(const GatherGemmParams& p, bool b_kn, hipStream_t stream, const char** name_only,
                           float* scratch, int64_t scratch_bytes, int64_t* scratch_query);

// Patch variant for stride-1 gathers (igemm_patch.hip): a workgroup owns a 16 x 16 patch of grid points of one image
// and loads its input halo once per K slice.  patch_plan: 0 = not a patch launch, else the kernel form (4: four fused
// sub-pixel phases of 2x2 taps, 1: a 4x4 stride-1 gather, 2 | 3: a 4x4 stride-2 gather as four parity planes with 256 |
// 128 columns per workgroup) with the geometry filled in.  The launcher expects the operand images / headers of
// launch_gather_gemm_h3 in `p`.
struct PatchGeom {
  int dy0, dx0;      // smallest tap offsets: halo pixel (0, 0) is input (gy0 + dy0, gx0 + dx0)
  int hh, hw;        // halo rows / columns
  int py, px;        // patches per image along y / x
  int rem;           // > 0: the patches cover grid columns [0, 16 px) only, the last `rem` (<= 4) columns go to a per-tap launch
  int nblocks;       // 8-pixel DMA blocks of the halo
  int delay2;        // (r6, nph 6) shader cycles the second workgroup of a CU starts behind the first
  int ablate;        // timing experiments only (ADVOC_H3_PATCH_ABLATE bits: 1 no DMA, 2 no MFMA, 4 no barrier); 0 in use
  // stride-2 gathers as four parity planes of the input (nph 2 | 3): plane (py, px) holds input pixels (2 y + py, 2 x + px);
  // s2_a0y[py] / s2_a0x[px] = smallest plane-row / plane-column offset of the plane's taps (its halo origin),
  // s2_tap[4 (2 py + px) + t] = (row offset - s2_a0y) | (column offset - s2_a0x) << 8 | weight tap << 16
  int s2_a0y[2], s2_a0x[2];
  int s2_tap[16];
};
int patch_plan(const GatherGemmParams& p, PatchGeom* g);
int launch_patch_gemm_h3(const GatherGemmParams& p, const PatchGeom& g, int nph, hipStream_t stream,
                         const char** name_only);

}  // namespace advoc
