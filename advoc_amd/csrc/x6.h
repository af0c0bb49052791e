// Split-bf16 arithmetic shared by image.hip, igemm.hip, igemm_x6d.hip and wgrad.hip.
//
// Every fp32 operand is written as x = x0 + x1 + x2 EXACTLY, each term a bf16 (8 significand bits, by
// truncation), and an fp32 product a * b is accumulated as six bf16 MFMA products with fp32 accumulation,
// smallest first: a1 b1, a0 b2, a2 b0, a0 b1, a1 b0, a0 b0.  bf16 x bf16 products are exact in fp32; the dropped
// a1 b2, a2 b1, a2 b2 are <= 2^-22 of the product: fp32-level error (2.5e-7 rel-L2 vs float64 on whole layers,
// tools/micro/x6_check.py) at 6 / 16 of the fp32 MFMA cost.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace advoc {

// the high halves of the three words are the bf16 terms; x1 / x2 take the next 8 / the last <= 8 bits
__device__ __forceinline__ void split3(float x, unsigned& h0, unsigned& h1, unsigned& h2) {
  h0 = __float_as_uint(x) & 0xffff0000u;
  const float r1 = x - __uint_as_float(h0);
  h1 = __float_as_uint(r1) & 0xffff0000u;
  h2 = __float_as_uint(r1 - __uint_as_float(h1));
}
// two bf16 (high halves of lo / hi) in one dword, lo in the low half
__device__ __forceinline__ unsigned pack_hi16(unsigned lo, unsigned hi) {
  return __builtin_amdgcn_perm(hi, lo, 0x07060302u);
}

// image.hip
// ---- fp16 pair images (igemm_h3.hip): see image.hip for the arithmetic ----
// hdr: two words per GEMM operand in device memory: [0] bit pattern of the largest |value| (zero it, then run
// launch_amax over every source of the operand), [1] the factor 2^-s that undoes the operand's scaling (written by
// the image kernels, read by the GEMM epilogue).
int launch_amax(const float* x, int64_t elems, int c, const float* scale, const float* shift, int act,
                const uint8_t* mask, float mask_scale, unsigned* amax, hipStream_t stream);
// act(scale * x + shift) * mask * mask_scale of `elems` fp32 values (channels innermost, c % 32 == 0) as 128-byte
// K slices img[elems / 32][plane][32] fp16, scaled by the power of two derived from hdr[0]
int launch_pair_image(const float* x, uint16_t* img, int64_t elems, int c, const float* scale, const float* shift,
                      int act, const uint8_t* mask, float mask_scale, unsigned* hdr, bool delayed, hipStream_t stream,
                      float* colsum = nullptr, int w_log = 0, int w_pitch = 0, float* colsum_table = nullptr);
// One GEMM operand = one or two channel-concatenated sources under ONE scale.  make_operand_image writes the image
// (source 1 behind source 0 at its 256-byte-rounded size) and the 32-byte header {[0] magnitude accumulator of the pass in
// flight (0 between calls), [1] 2^-s of the image in the buffer, [2] largest magnitude of that image (the scale source
// of the next one-pass image), [3] values that left the one-pass head room (cumulative), [4] word 3 before the last
// pass, [5] exact refits taken (cumulative), [6] arrival counter of the refit kernel, [7] reserved}; every call leaves
// the header "rotated" ([2] <- [0], [0] <- 0, [4] <- [3]: by the last workgroup of the refit kernel, or by a one-thread
// kernel behind an exact image).  delayed ==
// false: the exact two-pass form (magnitude pass, then image pass, largest magnitude at [2^13, 2^14)).  delayed == true:
// ONE pass, scale from the previous image's magnitude placed at [2^9, 2^10), this image's magnitude recorded for the
// next call, followed by refit_image_kernel, which rebuilds the image with the exact scale whenever a value left the
// head room, the tensor shrank by more than 2^6 or the header held no usable magnitude -- on the device, in the same
// stream, so the consumer never sees a clamped operand (image.hip).
struct ImageSource {
  const float* x;
  int64_t elems;
  int c;
  const float* scale;
  const float* shift;
  int act;
  const uint8_t* mask;
  float mask_scale;
};
// colsum0 != null: the image pass of source 0 also adds its per-channel sums over the logical pixels (x < w_log of rows
// w_pitch pixels apart) to colsum0[c] -- the bias gradient when the operand is an output gradient (image_colsum_ok(c))
// colsum_table: kColsumBytes of scratch (the sums are collected in kColsumReplicas copies first)
// keep_history: the header is a persistent one (a layer's x_hdr / dy_hdr): an exact image leaves it rotated for a one-pass
// image next time.  false for the per-call images in the launch workspace, whose header nobody reads again
int make_operand_image(const ImageSource& s0, const ImageSource& s1, uint16_t* img, unsigned* hdr, bool delayed,
                       hipStream_t stream, float* colsum0 = nullptr, int w_log = 0, int w_pitch = 0,
                       float* colsum_table = nullptr, bool keep_history = true);
int launch_image_refit(const ImageSource& s0, const ImageSource& s1, uint16_t* img, unsigned* hdr, hipStream_t stream);
bool image_colsum_ok(int c);
// out[ch] += sum over the kColsumReplicas copies of table[r][ch] (the fold of every replica table of this library)
int launch_colsum_reduce(const float* table, float* out, int c, hipStream_t stream);
// max |x| over a flat fp32 array, raised in *amax (float bits, zeroed by the caller)
int launch_amax_any(const float* x, int64_t elems, unsigned* amax, hipStream_t stream);
constexpr int kColsumReplicas = 64;
constexpr int64_t kColsumBytes = (int64_t)kColsumReplicas * 1024 * 4;
// weights [tap][k][n] (b_kn) or [tap][n][k] -> wq[tap][n_total][ktot / 32][plane][32] fp16 (amax pass included;
// hdr[0] must be zero on entry)
// amax_src != null: the largest |w| is already known (advoc_segmented_amax_f32: one launch per arena instead of one
// magnitude pass per weight image) -- the magnitude pass is skipped and hdr[0] is not read
int launch_pair_weights(const float* w, uint16_t* wq, int taps, int n_total, int ktot, bool b_kn, unsigned* hdr,
                        hipStream_t stream, const unsigned* amax_src = nullptr);
// ---- bf16 triple weights (register-split kernels of igemm.hip): wq[plane][tap][n_total][ktot], zeros for n >= n_valid
int launch_split_weights(const float* w, uint16_t* wq, int taps, int n_total, int n_valid, int ktot, bool b_kn,
                         hipStream_t stream);

}  // namespace advoc
