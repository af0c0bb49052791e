// Patch gather-GEMM on operand images: every gather of the model on a grid of at least 16 x 16 points (transposed-conv
// forward and conv backward-data as four sub-pixel phases of 2x2 taps; the 4x4 stride-1 convolution of the
// discriminator's layer_4 in both directions; the 4x4 stride-2 gathers -- conv forward, transposed-conv backward-data --
// as four parity planes of the input) with the A operand read ONCE per workgroup instead of once per tap.
//
// Same arithmetic and operand images as igemm_h3.hip (fp16 pairs, three MFMA products per fp32 product, replaces the
// cuDNN / Eigen Conv2DBackpropInput / Conv2D kernels TF1 runs for models/advoc/advoc_model.py:25-69,185-199).  What
// differs is the tile: igemm_h3.hip walks K = (tap, channel slice) and fetches a fresh [rows x 128 B] A tile for every
// tap -- with stride-1 gathers the same input pixel is fetched 4 (2x2 phases, x4 phases) or 16 (4x4) times by one
// workgroup and the launch is bound by the L2 -> LDS volume (NOTEBOOK.md §4.2).  Here a workgroup owns a 16 x 16 PATCH
// of grid points of one image:
//   * per 32-channel K slice the patch's input HALO ((16 + e) x (16 + e) pixels, e = 2 or 3; one 128-byte line per
//     pixel) goes global -> LDS once (LDS-DMA, zero padding from the descriptor's range check) and serves every tap
//     and every phase: the A fragment of tap (dy, dx) is the same LDS rows shifted by dy * halo_width + dx;
//   * the K loop is (slice, tap step); a step streams ONE 256-row x 128-byte B tile (2 LDS stages) -- four phases x 64
//     output channels of tap t of each phase (NPH = 4), or 256 output channels of tap t (NPH = 1) -- and runs 48
//     MFMAs per wave on it; the halo of the NEXT slice arrives in pieces during the steps of the current one
//     (2 halo buffers);
//   * 8 waves, each a 128-point x 64-column accumulator (4 x 2 blocks of 32 x 32): wave -> (half of the patch,
//     phase | column quarter).  1024 output pixels x 64 channels (NPH = 4) or 256 x 256 (NPH = 1) per workgroup;
//   * L2 -> LDS bytes per K-slice: ~41 KiB of halo + 4 x 32 KiB of weights for 4.2 algorithmic MFLOP x 4
//     (10 B / kFLOP) against 46 (128 x 64 tile), 31 (128 x 128) and 15.6 (256 x 256) of igemm_h3.hip.
// LDS rows are 128 bytes; chunk c of the halo pixel in halo column x sits at position c ^ ((x >> 1) & 7).  A
// ds_read_b128 is served in four fixed 16-lane groups that are NOT contiguous ({0-3, 12-15, 20-27}, ...): with 32-point
// blocks of 2 patch rows x 16 columns a group is columns {0-3, 12-15} of one row and {4-11} of the next, i.e. 16
// different columns, and (address bit 7, position) -- the 16-byte slot of the 256-byte bank row -- is different for
// each of them whatever the tap offset and the halo width (a swizzle by the linear pixel index is 2-way conflicted
// whenever the halo pitch is odd: 25 % of the LDS cycles of the 4x4 kernel were replays, 450 -> 510 TFLOP/s with the
// DMAs ablated once fixed).
#include <utility>
#include <string>

#include "common.h"
#include "igemm.h"
#include "tuning.h"
#include "image_emit.h"
#include "lds_dma.h"
#include "x6.h"

namespace advoc {
namespace {

// (r6, NPH = 6) arrivals of two-per-CU workgroups per (XCC, CU): see the staggered start in the kernel body
__device__ unsigned g_p4w_arrivals[8 * 256];

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_void_p;

__device__ __forceinline__ float act_slope_p(int act) {
  return act == ADVOC_ACT_LRELU02 ? 0.2f : (act == ADVOC_ACT_RELU ? 0.f : 1.f);
}

// NPH selects the gather a workgroup fuses:
//   4  four sub-pixel phases of 2x2 taps (transposed-conv forward, conv backward-data): 256 points x 4 phases x 64 columns
//   1  one 4x4 stride-1 gather (layer_4 both directions): 256 points x 256 columns
//   2  one 4x4 STRIDE-2 gather (conv forward, transposed-conv backward-data) as four PARITY PLANES of the input, each a
//      2x2-tap stride-1 gather over the output grid: the K loop walks (channel slice, plane), the halo of a slice is the
//      17 x 17 pixels of ONE plane -- every other pixel of the ordinary image, de-interleaved by the DMA's source
//      addresses, so the planes exist in LDS only: 256 points x 256 columns
//   3  as 2 with 128 columns (the 128-channel layers): 8 waves = 4 point quarters x 2 column halves
//   5  as 1 with 128 columns (r4: AdVoc-small's layer_4 backward-data, 256 -> 128 channels), waves as in 3
// (W = 4 wavefronts, one per SIMD with 512 registers and register-carried fragments, was built and measured 0.85-1.0 x
// the 8-wave form: the code paths are kept behind CARRY / DOUBLE_B, only W = 8 is instantiated.)
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

template <int NPH, int W>
struct PCfg {
  static constexpr bool S2 = NPH == 2 || NPH == 3;
  static constexpr int WAVES = W, THREADS = 64 * W;
  static constexpr bool S1 = NPH == 1 || NPH == 5;         // the 4x4 stride-1 gather
  static constexpr bool N128 = NPH == 3 || NPH == 5;       // 128 columns per workgroup
  // (r6) NPH = 6: the four-phase gather of NPH = 4 on TWO 4-WAVE WORKGROUPS PER CU (see "two workgroups per CU" below): a
  // workgroup owns 8 x 16 grid points, a wave is a phase (128 points x 64 columns, the accumulator of an NPH = 4 wave), the
  // weights of a phase are read by that wave alone and live in a ring of its own
  static constexpr bool P4W = NPH == 6;
  static constexpr bool FOURPH = NPH == 4 || NPH == 6;
  static_assert(!P4W || W == 4, "the two-per-CU instance has four waves");
  static constexpr int PROWS = P4W ? 8 : 16;               // grid rows of a patch (16 columns always)
  static constexpr int NST = S1 ? 16 : 4;                  // tap steps per K slice
  static constexpr int BN = FOURPH ? 64 : (N128 ? 128 : 256);   // output channels per workgroup
  static constexpr int BROWS = N128 ? 128 : 256;           // rows of a B stage (NPH = 4: four phases x 64 columns)
  static constexpr int MT = N128 ? 2 : ((NPH == 4 && W == 4) ? 8 : 4);   // 32-point blocks per wave
  static constexpr int NT = (NPH == 1 && W == 4) ? 4 : 2;  // 32-column blocks per wave
  static constexpr bool CARRY = W == 4 && !P4W;            // next step's first A fragments fetched before the barrier
#ifdef ADVOC_P3_DOUBLE_B        // (A/B builds only: with the ablation switches compile-time the 4x4 instances have the registers)
  static constexpr bool DOUBLE_B = (W == 4 && !P4W) || NPH == 1 || NPH == 5;
#else
  static constexpr bool DOUBLE_B = W == 4 && !P4W;         // B fragments of both k steps in registers at once (W = 8: the
                                                           // second set costs the 16 registers that tip the loop into scratch)
#endif
  static constexpr int HW = FOURPH ? 18 : (S1 ? 19 : 17);  // halo width (= height but for P4W; patch_plan takes only these)
  static constexpr int HROWS = P4W ? 10 : HW;              // halo rows
  static constexpr int HP = S1 ? 20 : 18;                  // halo row pitch in LDS, EVEN: address bit 7 (the half of the
                                                           // 256-byte bank row) must follow the column's parity
  static constexpr int HALO_BLOCKS = (HROWS * HP + 7) / 8; // 8-pixel DMA blocks: 41 | 48 | 39 | 23
  static constexpr int HPS = S1 ? 1 : ((W == 4 && !P4W) ? 3 : 2);   // halo DMA slots per wave and step
  static constexpr int BPW = BROWS / 8 / W;                // B DMA blocks (8 rows) per wave and step
  // P4W: a wave's ring = two stages of [64 columns x one 16-channel k step x 2 planes] = 64 rows x 64 bytes, stage = k step
  static constexpr int KH_STAGE = 64 * 64;
  // where in a step the DMAs of the next one are issued: 1 after the step's fragment reads, 2 after its first MFMA group
  // (measured with a run-time switch: +4-5 % on the four-phase kernel, neutral on the 4x4 one; at the top of the step, 0,
  // the DMA's LDS writes collide with the fragment reads that follow the barrier)
  static constexpr int DMA_POS = S1 ? 1 : 2;
  static constexpr int HALO_BYTES = HALO_BLOCKS * 1024;
  static constexpr int B_STAGE = P4W ? W * KH_STAGE : BROWS * 128;     // (P4W: "stage" h = k step h of every wave's ring)
  static constexpr int OFF_B = 2 * HALO_BYTES;
  // NPH = 3 (24 MFMAs per wave and tap, half the other instances'): FOUR tap buffers, two taps per rendezvous
  static constexpr int NBUF = NPH == 3 ? 4 : 2;
#ifdef ADVOC_P3_NO_STAGGER     // (A/B builds only: the r3 loop, all eight waves in step)
  static constexpr bool STAGGER = false;
#else
  // two wave groups one phase apart (K loop below): measured on one box against the in-step loop (tools/micro/lib_ab2.sh):
  // <1,0> 1.957 -> 1.894 ms, <1,1> 1.896 -> 1.879; but <4,0> 0.675 -> 0.730, <2,0> 0.444 -> 0.471, <2,1> 0.697 -> 0.756,
  // <4,1> 0.584 -> 0.588 -- a single wave per SIMD does not keep the matrix pipe full through its 24-MFMA phase (the
  // intervals come out at ~925 cycles where 768 were expected), and the instances with more DMA slots per step lose more
  // than the overlap returns.  Only the 4x4 stride-1 instance keeps it.
  static constexpr bool STAGGER = W == 8 && NBUF == 2 && NPH == 1;
#endif
  static constexpr int LDS_BYTES = OFF_B + NBUF * B_STAGE;  // 146 | 160 | 142 | 142 | 78 KiB
  static constexpr int PTS_W = 32 * MT;                    // grid points per wave
  // (r5) the epilogue's scratch -- per wave a 32 x 36-word transpose tile and the pixel table of its points -- in the LDS the
  // NEXT tile's prologue does not write (halo buffer 0, weight stage 0 and, with four stages, 1): the tiles in halo buffer
  // 1, the tables in weight stage 1 (2)
#ifdef ADVOC_P3_LATE_PROLOGUE   // (A/B builds only: the r4 order, every tile's prologue behind the previous tile's epilogue)
  static constexpr bool EARLY_PROLOGUE = false;
#else
  static constexpr bool EARLY_PROLOGUE = true;
#endif
  // (r5) With the run-time ablation switches gone from the product build the K loop is one basic block per step, and the
  // scheduler moves fragment reads, DMA issue and MFMAs across what used to be block boundaries: <1,.> -5 %, <3,0> -8 %, but
  // <4,.> / <2,.> +1...4 % (same box, alternating: profiles/r05_a_*) -- their hand-placed DMA issue point (DMA_POS = 2:
  // behind the first MFMA group) is what the boundaries had been protecting.  PIN puts a scheduling fence where those
  // boundaries were (around every MFMA group and DMA slot) for the instances that lost.
#ifdef ADVOC_P3_PIN_ALL
  static constexpr bool PIN = true;
#elif defined(ADVOC_P3_PIN_NONE)
  static constexpr bool PIN = false;
#else
  static constexpr bool PIN = NPH == 4 || NPH == 2 || NPH == 6;
#endif
#ifdef ADVOC_P3_PIN_DMA_ONLY    // (A/B builds only)
  static constexpr bool PIN_MFMA = false;
#else
  static constexpr bool PIN_MFMA = PIN;
#endif
  static constexpr int EPI_T_BYTES = WAVES * 32 * 36 * 4, EPI_PIX_BYTES = WAVES * 2 * PTS_W * 4;
  static constexpr int EPI_T_OFF = HALO_BYTES;
  // (P4W: the next tile's prologue fills BOTH stages of every ring; tiles and tables share halo buffer 1)
  static constexpr int EPI_PIX_OFF = P4W ? HALO_BYTES + EPI_T_BYTES : OFF_B + (NBUF == 4 ? 2 : 1) * B_STAGE;
  static_assert(WAVES * NST * HPS >= HALO_BLOCKS, "every halo block has a DMA slot");
  static_assert(LDS_BYTES <= 160 * 1024 && (!P4W || LDS_BYTES <= 80 * 1024), "LDS budget");
  static_assert(P4W ? EPI_T_BYTES + EPI_PIX_BYTES <= HALO_BYTES
                    : (EPI_T_BYTES <= HALO_BYTES && EPI_PIX_BYTES <= B_STAGE && EPI_PIX_OFF + EPI_PIX_BYTES <= LDS_BYTES),
                "the epilogue's scratch fits the buffers the next tile's prologue leaves alone");
};

// BWD = 0: forward epilogue (bias, dropout mask on the result, one destination); BWD = 1: backward-data epilogue
// (activation gradient at the pre-activation value, the consumer's BN affine / dropout mask, accumulate, two
// destinations)
// NE: consumers whose operand images the (forward) epilogue writes, in oimg[0 .. NE): a compile-time count, so that a
// launch pays for the image arithmetic of the consumers it has and no more
// HALF (NPH = 4 only): a 32-column problem on the 64-column tile -- the wave's second 32-column block does not exist: no
// fragments, no MFMAs, no epilogue for it, and the waves whose stage rows are those columns issue no weight DMA
// GI (backward-data): the activation gradient is gated by the sign of the consuming layer's operand image (GemmDest::ximg:
// the fp32 pre-activation tensor was never written) -- 8 bytes per 4 values instead of 16, no batch-norm affine (the image
// has it applied)
template <int NPH, int W, int BWD, int NE, int HALF = 0, int GI = 0>
__device__ __forceinline__ void patch_gemm_h3_body(const GatherGemmParams& p, const PatchGeom& g) {
  using C = PCfg<NPH, W>;
  static_assert(!HALF || (NPH == 4 && C::NT == 2), "half tiles exist for the four-phase instance");
  constexpr int NST = C::NST, BN = C::BN, HPS = C::HPS, MT = C::MT, NT = HALF ? 1 : C::NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned char* smem_b = reinterpret_cast<unsigned char*>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // wave -> (part wm of the patch's points, phase, first of its columns), W = 8:
  //   NPH = 4: (half wave / 4, phase = wave % 4), 64 columns;  NPH = 1, 2: (half wave / 4, column quarter wave % 4);
  //   NPH = 3: (quarter wave / 2, column half wave % 2)
  constexpr bool S2 = C::S2;
  const int wm = C::P4W ? 0 : (C::N128 ? wave >> 1 : (W == 8 ? wave >> 2 : (NPH == 4 ? 0 : wave >> 1)));
  const int phase = C::FOURPH ? (wave & 3) : 0;
  const int ncol0 = C::FOURPH ? 0 : (C::N128 ? (wave & 1) * 64 : (W == 8 ? (wave & 3) * 64 : (wave & 1) * 128));
  const int brow0 = C::P4W ? 0 : (NPH == 4 ? phase * 64 : ncol0);     // first B-stage row of the wave's columns
  const int b_phase = C::P4W ? wave : (NPH == 4 ? wave * (256 / W) / 64 : 0);   // phase of the B rows this wave LOADS
  const int ppi = g.py * g.px;                       // patches per image
  const int npatch = p.batch * ppi;
  const int ktot = p.c0 + p.c1;
  const int nslices = (S2 ? 4 : 1) * (ktot / 32);      // S2: K slice s = (channel slice s / 4, parity plane s % 4)
  constexpr int hw = C::HP;                          // halo row pitch: compile-time, so that block i of a fragment read is an
  constexpr int hpix = C::HROWS * C::HP;             // immediate offset

  const __amdgpu_buffer_rsrc_t rs_a0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(p.a0_img), 0, p.a0_img_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(p.a1_img ? p.a1_img : p.a0_img), 0, p.a1_img_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(p.wq), 0, p.wq_taps * p.n_total * ktot * 4, 0x00020000);

  // (r5) the same three descriptors for the inline-assembly DMA of lds_dma.h: the next tile's prologue is issued in front of
  // the epilogue, and behind a DMA the compiler knows about it waits for vmcnt(0) before the epilogue's first LDS access
  // (ISA checked) -- the round trip the early issue is there to hide.  Untracked, those DMAs are simply the oldest entries
  // of the memory queue: every counted wait of the epilogue covers them.
  const u32x4s ra_a0 = dma_rsrc(p.a0_img, (unsigned)p.a0_img_bytes);
  const u32x4s ra_a1 = dma_rsrc(p.a1_img ? p.a1_img : p.a0_img, (unsigned)p.a1_img_bytes);
  const u32x4s ra_b = dma_rsrc(p.wq, (unsigned)(p.wq_taps * p.n_total * ktot * 4));
  constexpr bool kAsmDma = false;       // shadowed by `true` where the next tile's prologue is issued
  // Launch constants of the epilogue, read ONCE per workgroup (r5; per tile before): a load whose value the epilogue uses
  // waits for everything older in the wave's memory queue, i.e. for the next tile's prologue DMAs issued in front of it.
  auto sgpr_f = [](float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); };
  const float unscale = sgpr_f(__uint_as_float(p.a_hdr[1]) * __uint_as_float(p.b_hdr[1]));   // exact: powers of two
  // forward: consumers' operand images of the output (image_emit.h), one-pass scale from the consumer header.
  // (r5) backward-data, NE = 1: oimg[0] is the OUTPUT-GRADIENT image of the layer below (destination 0 seen as that layer's
  // dy), under the a-priori scale of GatherGemmParams::oimg_bounded: |dx| <= max|dy| max|w| (taps per output x K) [+ what
  // the destination already holds] -- known before the first MFMA, so nothing can leave the fp16 range, nothing is ever
  // refitted, and the fp32 tensor need not exist (d0_no_store); the per-channel sums of the tensor -- the lower layer's
  // bias gradient, which used to ride in its image pass -- are taken here too (ocolsum_table).
  constexpr bool emit0 = NE >= 1, emit1 = !BWD && NE >= 2;
  constexpr bool kNoMask = BWD && NE >= 1;       // the BWD-emitting instances take no dropout mask on their destinations
  float eup0_ = 1.f;
  if (emit0) {
    if (BWD) {
      float bound = __uint_as_float(*p.a_amax) * emit_weight_bound(p, ktot);
      if (p.obound_add) bound += __uint_as_float(*p.obound_add);
      eup0_ = emit_up_scale_bounded(bound);
    } else if (p.oimg_bounded) {
      // forward under the a-priori scale: |y| <= max|act(x)| max|w| taps K + max|b| (the consumer's activation has slope <= 1)
      float bm = 0.f;
      if (p.bias)
        for (int n = lane; n < p.n_total; n += 64) bm = fmaxf(bm, fabsf(p.bias[n]));
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) bm = fmaxf(bm, __shfl_xor(bm, off, 64));
      eup0_ = emit_up_scale_bounded(__uint_as_float(*p.a_amax) * emit_weight_bound(p, ktot) + bm);
    } else {
      eup0_ = emit_up_scale(p.oimg[0].hdr[2]);
    }
  }
  const float eup0 = sgpr_f(eup0_);
  const float eup1 = emit1 ? sgpr_f(emit_up_scale(p.oimg[1].hdr[2])) : 1.f;
  // (the BWD-emitting instances are LEAN: no dropout mask on the destination, no accumulation -- the launcher checks -- so
  // that the epilogue's prefetched operands, the image arithmetic and the column sums fit the registers next to the tile)
  // (NE = 2, backward-data: the same with an ACCUMULATING destination 0 -- the value already there is loaded and added before
  // the image is written: the encoder chain, where a decoder's skip gradient arrived first)
  constexpr bool kLean = BWD && NE == 1;
  float dmax1 = 0.f;       // max |value| written to destination 1 (d1_amax_out)
  if (emit0 && threadIdx.x == 0) {
    if (emit0) p.oimg[0].hdr[1] = __float_as_uint(1.f / eup0);
    if (emit1) p.oimg[1].hdr[1] = __float_as_uint(1.f / eup1);
  }
  // largest |activation x scale| this wave has emitted, over ALL its tiles: emit_finish (cross-lane reduction, a header load
  // whose wait drains every store of the epilogue, two atomics) runs once per launch instead of once per tile
  float evmax0 = 0.f, evmax1 = 0.f;

  const float gslope = act_slope_p(p.grad_act);
  const int src_c0 = p.c0, src_c1 = p.c1, src_p0 = p.a0_pitch, src_p1 = p.a1_pitch;
#ifdef ADVOC_P3_ABL        // (timing experiments, tools/micro/build_ablations.sh: the switches below as a COMPILE-TIME constant --
  constexpr int abl = ADVOC_P3_ABL;     // no branch in the K loop, the product's schedule minus what is left out)
#elif defined(ADVOC_DIAG)
  const int abl = g.ablate;       // timing experiments only (ADVOC_H3_PATCH_ABLATE): 1 no DMA, 2 no MFMA, 4 no barrier, 8 no halo DMA, 16 no B DMA, 64 no epilogue, 256 / 512 no A / B fragment reads, 1024 every B tile from one L2-resident slab, 2048 halo address arithmetic without its DMA, 4096 no halo pieces in the K loop, 128 epilogue stores dropped
#else
  // (r5) a compile-time zero in the product library: as a run-time value every `if (abl & ...)` was a branch in the K loop --
  // one in front of each rendezvous, MFMA group and DMA slot -- i.e. a basic-block boundary the scheduler does not move
  // fragment reads, DMA issue or MFMAs across
  constexpr int abl = 0;
#endif

  // ---- persistent workgroups: the launch holds one workgroup per CU; XCD x walks its own contiguous range of tiles
  // (column tile slowest, so the workgroups an XCD runs together stream the same weights), and the stores of one
  // tile's epilogue drain while the next tile's operands arrive ----
  const int ntiles = npatch * ((p.n_total + BN - 1) / BN);      // (a 32-column problem runs on the 64-column instance: see patch_plan)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
  const int tq_ = ntiles >> 3, tr_ = ntiles & 7;
  const int t_lo = xcd < tr_ ? xcd * (tq_ + 1) : tr_ * (tq_ + 1) + (xcd - tr_) * tq_;
  const int t_hi = t_lo + tq_ + (xcd < tr_ ? 1 : 0);

  // (the backward-data instances used to take ONE tile per workgroup: with the K loop's per-lane constants kept alive
  // across the epilogue for a next tile, their epilogue -- accumulators + the block's prefetched operands -- spilled;
  // since the constants are derived from an opaque copy of the lane id per tile, below, they walk tiles too: 1-2 %)
  // (P4W) The two workgroups of a CU start half a tile apart -- and stay apart: whichever is in its epilogue leaves the
  // matrix pipe to the other's K loop.  (Started together they would run their K loops together, at half speed each, and
  // their epilogues together, pipe idle: the 8-wave instance again.)  The dispatcher places workgroups 0 .. per_xcd / 2 - 1 of
  // an XCD on its CUs first and the second half next to them.
  // Which two workgroups share a CU is the dispatcher's business: each workgroup reads its CU (HW_ID: CU, shader array and
  // engine; XCC_ID) and counts its arrival there in a table that is never reset -- two arrivals per CU and launch keep the
  // parity of the count meaning "first / second"; the second sleeps (its first wave does: the others meet it at the first
  // slice's barrier).
  if (C::P4W && g.delay2 > 0 && wave == 0) {
    const unsigned cu = __builtin_amdgcn_s_getreg((7 << 11) | (8 << 6) | 4);       // HW_REG_HW_ID[15:8]
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);     // HW_REG_XCC_ID[3:0]
    unsigned old = 0;
    if (lane == 0) old = atomicAdd(&g_p4w_arrivals[((xcc & 7u) << 8) | (cu & 255u)], 1u);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old & 1u) {
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)g.delay2) __builtin_amdgcn_s_sleep(32);
    }
  }
  for (int tile = t_lo + slot; tile < t_hi; tile += per_xcd) {
  // The per-lane constants of the K loop are derived from an OPAQUE copy of the lane id inside the tile loop: hoisted out
  // of it they would stay alive across the epilogue, whose accumulators + prefetched operands then spill (80-230
  // registers in the backward-data instances).
  int ln = lane;
  asm volatile("" : "+v"(ln));
  const int lrow = ln >> 3, lpos = ln & 7;
  constexpr int hmagic = (65536 + hw - 1) / hw;      // h / hw == (h * hmagic) >> 16 for h < 512, hw <= 20
  const int half = ln >> 5, l32 = ln & 31;
  // halo pixel of this lane's row in the wave's first 32-point block (2 patch rows x 16 columns) for tap (0, 0); block
  // i is 2 i halo rows further
  const int h_base = (wm * (C::PTS_W / 16) + (l32 >> 4)) * hw + (l32 & 15);
  // fragment chunk (plane, k step, half) = 4 plane + 2 ks + half sits at position chunk ^ swizzle = (half ^ swizzle) ^
  // (4 plane + 2 ks): one byte offset per row, the (plane, ks) part is an XOR with a constant
  // (P4W: 64-byte rows -- chunk 2 plane + half of the stage's k step at position chunk ^ ((row >> 2) & 3): the 16 rows of
  // a ds_read_b128 service group, {0-3, 12-15, 20-27} or {4-11, 16-19, 28-31}, take every residue mod 4 four times with
  // four different (row >> 2) & 3, i.e. 16 different 16-byte slots of the 256-byte bank row)
  const int bfrag = C::P4W ? l32 * 64 + ((half ^ ((l32 >> 2) & 3)) * 16) : (brow0 + l32) * 128 + ((half ^ ((l32 >> 1) & 7)) * 16);
  // the tap tables of the wave's compute phase and of the phase whose B rows it loads, one tap per lane: read back with
  // v_readlane inside the K loop (an s_load there would put an lgkmcnt(0) wait -- SMEM returns out of order -- in front
  // of every step's LDS reads).  (S2: one table for all waves, entry 4 plane + t, offsets already relative to the
  // plane's halo origin)
  const int tapv_c = S2 ? g.s2_tap[ln & (kMaxTaps - 1)] : p.tap[phase][ln & (kMaxTaps - 1)];
  // the parity planes' first rows / columns as four scalars: indexed as g.s2_a0y[parity] in the K loop they were scalar
  // LOADS from the kernel arguments, each followed by s_waitcnt lgkmcnt(0), three times per K slice
  const int s2y0 = g.s2_a0y[0], s2y1 = g.s2_a0y[1], s2x0 = g.s2_a0x[0], s2x1 = g.s2_a0x[1];
  const int tapv_b = S2 ? tapv_c : p.tap[b_phase][ln & (kMaxTaps - 1)];
  const int nt = tile / npatch;
  const int pid = tile - nt * npatch;
  const int img = pid / ppi;
  const int pin = pid - img * ppi;
  const int gy0 = (pin / g.px) * C::PROWS, gx0 = (pin % g.px) * 16;
  const int n0 = nt * BN;

  // ---- B DMA lanes: the wave loads stage rows (256 / W) wave .. in 8-row blocks; stage row q holds (phase q / 64,
  // column q % 64) for NPH = 4, column q for NPH = 1.  Block k is 8 k rows further (that part goes into the scalar
  // offset); the swizzle of row q, (q >> 1) & 7, is lrow >> 1 for even k and that ^ 4 for odd k ----
  int b_off[2];
  // (NPH = 4: the wave's stage rows are one phase's columns [32 (wave & 1), + 32); beyond n_total -- a 32-column problem --
  // they do not exist in the weight image: the DMA gets an out-of-range offset and writes zeros)
  const bool b_cols = NPH != 4 || n0 + ((wave * (C::BROWS / W)) & 63) < p.n_total;
  {
    const int q = wave * (C::BROWS / W) + lrow;
    const int col = NPH == 4 ? (q & 63) : q;
    const int rowb = ((n0 + col) * ktot) * 4;
    b_off[0] = rowb + ((lpos ^ (lrow >> 1)) * 16);
    b_off[1] = rowb + ((lpos ^ (lrow >> 1) ^ 4) * 16);
    if (C::P4W) {      // one DMA = 16 rows x 64 bytes: lane -> (row ln >> 2, position ln & 3) holding chunk position ^ ((row >> 2) & 3)
      const int c = (ln & 3) ^ ((ln >> 4) & 3);
      b_off[0] = ((n0 + (ln >> 2)) * ktot) * 4 + (c >> 1) * 64 + (c & 1) * 16;
    }
  }

  // Halo pieces of K slice SL (channels [32 SL, 32 SL + 32) of the concatenated sources) that step T carries, into
  // halo buffer HB.  Slots beyond the halo issue nothing (every wait in the K loop is vmcnt(0): no counting to keep).
#define ADVOC_P3_HALO(SL, T, HB)                                                                          \
  {                                                                                                       \
    if (C::PIN) __builtin_amdgcn_sched_barrier(0);                                                        \
    const int k0_ = (S2 ? (SL) >> 2 : (SL)) * 32;                                                         \
    const int ppy_ = ((SL) >> 1) & 1, ppx_ = (SL) & 1;        /* S2: parity plane of the slice */           \
    const bool second_ = k0_ >= p.c0;                                                                     \
    const int c_ = second_ ? src_c1 : src_c0, pitch_ = second_ ? src_p1 : src_p0;                         \
    const int kk_ = second_ ? k0_ - p.c0 : k0_;                                                           \
    _Pragma("unroll") for (int j = 0; j < HPS; ++j) {                                                     \
      const int blk_ = ((T) * C::WAVES + wave) * HPS + j;                                                 \
      int lr_ = lrow;                                                                                     \
      asm volatile("" : "+v"(lr_));  /* recompute per use: hoisted out of the K loop these cost a VGPR per slot */ \
      const int h_ = blk_ * 8 + lr_;                                                                      \
      const int hy_ = (h_ * hmagic) >> 16;                                                                \
      const int hx_ = h_ - hy_ * hw;                                                                      \
      const int iy_ = S2 ? 2 * (gy0 + hy_ + (ppy_ ? s2y1 : s2y0)) + ppy_ : gy0 + hy_ + g.dy0;              \
      const int ix_ = S2 ? 2 * (gx0 + hx_ + (ppx_ ? s2x1 : s2x0)) + ppx_ : gx0 + hx_ + g.dx0;              \
      const bool ok_ = h_ < hpix && hx_ < C::HW && (unsigned)iy_ < (unsigned)p.in_h &&                    \
                       (unsigned)ix_ < (unsigned)p.in_w;                                                  \
      const int voff_ = ok_ ? (((img * p.a_h + iy_) * pitch_ + ix_) * c_ + kk_) * 4 +                     \
                                  ((lpos ^ ((hx_ >> 1) & 7)) * 16)                                        \
                            : (int)0x80000000;                                                            \
      unsigned char* d_ = smem_b + (HB) * C::HALO_BYTES + blk_ * 1024;                                    \
      if (blk_ >= C::HALO_BLOCKS || (abl & 9)) continue;                                                  \
      if (abl & 2048) { asm volatile("" ::"v"(voff_), "s"(lds_address(d_))); continue; }                  \
      if (kAsmDma) dma16(second_ ? ra_a1 : ra_a0, lds_address(d_), voff_, 0);                             \
      else if (second_) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a1, (lds_void_p)(d_), 16, voff_, 0, 0, 0); \
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a0, (lds_void_p)(d_), 16, voff_, 0, 0, 0);         \
    }                                                                                                     \
    if (C::PIN) __builtin_amdgcn_sched_barrier(0);                                                        \
  }

  // The B tile of (slice SL, tap step T) into stage ST
#define ADVOC_P3_B(SL, T, ST)                                                                             \
  {                                                                                                       \
    if (C::PIN) __builtin_amdgcn_sched_barrier(0);                                                        \
    const int wtap_ = __builtin_amdgcn_readlane(tapv_b, S2 ? ((SL) & 3) * 4 + (T) : (T)) >> 16;           \
    const int wslab_ = (abl & 1024) ? 0 : (wtap_ * p.n_total * ktot + (S2 ? (SL) >> 2 : (SL)) * 32) * 4;  \
    _Pragma("unroll") for (int k = 0; k < C::BPW; ++k) {                                                  \
      unsigned char* d_ = smem_b + C::OFF_B + (ST) * C::B_STAGE + (wave * C::BPW + k) * 1024;             \
      if (!(abl & 17) && (!HALF || b_cols)) {                                                             \
        if (kAsmDma) dma16(ra_b, lds_address(d_), b_cols ? b_off[k & 1] : (int)0x80000000, wslab_ + k * 8 * ktot * 4); \
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_void_p)(d_), 16, b_cols ? b_off[k & 1] : (int)0x80000000, \
                                                      wslab_ + k * 8 * ktot * 4, 0, 0);                   \
      }                                                                                                   \
    }                                                                                                     \
    if (C::PIN) __builtin_amdgcn_sched_barrier(0);                                                        \
  }

  // (P4W) k step H of the wave's own 64 columns of (slice SL, tap step T) into stage H of its ring: 4 DMAs of 16 rows
#define ADVOC_P3_BQ(SL, T, H)                                                                             \
  {                                                                                                       \
    if (C::PIN) __builtin_amdgcn_sched_barrier(0);                                                        \
    const int wtap_ = __builtin_amdgcn_readlane(tapv_b, (T)) >> 16;                                       \
    const int wslab_ = (wtap_ * p.n_total * ktot + (SL) * 32) * 4 + (H) * 32;                             \
    _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                       \
      unsigned char* d_ = smem_b + C::OFF_B + (H) * C::B_STAGE + wave * C::KH_STAGE + k * 1024;           \
      if (!(abl & 17)) {                                                                                  \
        if (kAsmDma) dma16(ra_b, lds_address(d_), b_off[0], wslab_ + k * 16 * ktot * 4);                  \
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_void_p)(d_), 16, b_off[0], wslab_ + k * 16 * ktot * 4, 0, 0); \
      }                                                                                                   \
    }                                                                                                     \
    if (C::PIN) __builtin_amdgcn_sched_barrier(0);                                                        \
  }

  // A fragments of tap step T, k step KS, from halo buffer HB into REG[MT][2]
#define ADVOC_P3_LOAD_A(REG, HB, T, KS)                                                                   \
  {                                                                                                       \
    const int tp_ = __builtin_amdgcn_readlane(tapv_c, S2 ? (s & 3) * 4 + (T) : (T));                      \
    const int dxo_ = S2 ? (tp_ >> 8) & 0xff : (int)(int8_t)((tp_ >> 8) & 0xff) - g.dx0;                   \
    const int toff_ = (S2 ? tp_ & 0xff : (int)(int8_t)(tp_ & 0xff) - g.dy0) * hw + dxo_;                  \
    const unsigned char* Hx = smem_b + (HB) * C::HALO_BYTES;                                              \
    const int sl_ = ((half ^ ((((l32 & 15) + dxo_) >> 1) & 7)) ^ (2 * (KS))) * 16;                        \
    const int ad_ = (h_base + toff_) * 128 + sl_;                                                         \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                        \
      _Pragma("unroll") for (int pl = 0; pl < 2; ++pl) {                                                  \
        if (abl & 256) asm volatile("" : "=v"(REG[i][pl]));        /* (timing only: no A fragment reads) */ \
        else REG[i][pl] = *reinterpret_cast<const f16x8*>(Hx + (ad_ ^ (4 * pl * 16)) + i * (2 * hw * 128)); \
      }                                                                                                   \
  }
#define ADVOC_P3_LOAD_B(REG, ST, KS)                                                                      \
  {                                                                                                       \
    const unsigned char* Bx = smem_b + C::OFF_B + (ST) * C::B_STAGE;                                      \
    _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                        \
      _Pragma("unroll") for (int pl = 0; pl < 2; ++pl) {                                                  \
        if (abl & 512) asm volatile("" : "=v"(REG[j][pl]));        /* (timing only: no B fragment reads) */ \
        else if (C::P4W) REG[j][pl] = *reinterpret_cast<const f16x8*>(smem_b + C::OFF_B + (KS) * C::B_STAGE + wave * C::KH_STAGE + \
                                                                      (bfrag ^ (pl * 32)) + j * 32 * 64);   \
        else REG[j][pl] = *reinterpret_cast<const f16x8*>(Bx + (bfrag ^ ((4 * pl + 2 * (KS)) * 16)) + j * 32 * 128); \
      }                                                                                                   \
  }
  // three fp16 products per 32x32x16 block, small terms first (a0 b1, a1 b0, a0 b0); product-major so that consecutive
  // MFMAs write different accumulators
#define ADVOC_P3_MFMA(AR, BR)                                                                             \
  if (C::PIN_MFMA) __builtin_amdgcn_sched_barrier(0);                                                     \
  if (abl & 2) {                                                                                          \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) asm volatile("" ::"v"(AR[i][0]), "v"(AR[i][1]));       \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) asm volatile("" ::"v"(BR[j][0]), "v"(BR[j][1]));       \
  } else {                                                                                                \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                        \
      _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                      \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AR[i][0], BR[j][1], acc[i][j], 0, 0, 0);       \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                        \
      _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                      \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AR[i][1], BR[j][0], acc[i][j], 0, 0, 0);       \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                        \
      _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                      \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AR[i][0], BR[j][0], acc[i][j], 0, 0, 0);       \
  }                                                                                                       \
  if (C::PIN_MFMA) __builtin_amdgcn_sched_barrier(0);

  floatx16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // The rendezvous of a step: own DMAs landed (vmcnt 0), OWN FRAGMENT READS RETURNED (lgkmcnt 0: the DMAs the others issue
  // behind the barrier overwrite the stage those reads came from -- lds_dma.h, dma_ring_barrier), everybody here.
#ifdef ADVOC_RING_NO_LGKM      // (A/B builds only: the r3 rendezvous)
#define ADVOC_P3_WAIT "s_waitcnt vmcnt(0)"
#else
#define ADVOC_P3_WAIT "s_waitcnt vmcnt(0) lgkmcnt(0)"
#endif
#define ADVOC_P3_RENDEZVOUS()                                                                             \
  {                                                                                                       \
    if (abl & 4) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                              \
    else asm volatile(ADVOC_P3_WAIT "\n\ts_barrier" ::: "memory");                                        \
  }

  // ---- prologue: the whole halo of slice 0, the B tile of step 0.  (r5) Only a workgroup's FIRST tile issues it here: the
  // prologue of every later tile goes out behind the previous tile's K loop, in front of its epilogue (below), so that the
  // DMAs' round trip runs under the epilogue instead of in front of an idle matrix pipe ----
  if (!C::EARLY_PROLOGUE || tile == t_lo + slot) {
#pragma unroll
    for (int t = 0; t < NST; ++t) ADVOC_P3_HALO(0, t, 0);
    if constexpr (C::P4W) {
      ADVOC_P3_BQ(0, 0, 0);
      ADVOC_P3_BQ(0, 0, 1);
    } else {
      ADVOC_P3_B(0, 0, 0);
      if (C::NBUF == 4) ADVOC_P3_B(0, 1, 1);
    }
  }

  // ---- K loop: one barrier per step.  Wait for the own DMAs of this step's B tile (and, at a slice boundary, of the
  // halo), barrier (everyone's data landed, everyone finished the step before), issue the next step's B tile into the
  // stage just vacated and a piece of the next slice's halo, then compute.  NST is even: stage = step parity.
  // Fragments are double-buffered in registers: the second k step's are fetched under the MFMAs of the first; the
  // halo of the running slice has been in LDS since the slice began, so the A fragments of the NEXT step's first k
  // step are fetched under the MFMAs of the second (a0 is carried across the barrier) and only the B fragments wait
  // for the barrier. ----
  f16x8 a0[MT][2], a1[MT][2], b0[NT][2], b1[NT][2];
  if constexpr (C::P4W) {
    // ---- (r6) TWO WORKGROUPS PER CU.  What the epilogue of a patch tile waits for -- 384-640 KiB per tile and CU at the
    // rate the chip sustains when every CU streams, behind ONE in-order vmcnt that the next tile's first rendezvous has to
    // drain -- nothing inside a workgroup can run under (profiles/r06_epilogue_ceiling.md: 10-35 % of these launches'
    // cycles, matrix pipe idle).  A second, independent workgroup on the CU can: 4 waves and 78 KiB each, one wave of each
    // per SIMD, started half a tile apart (below).  What makes the half-size workgroup cheap for the four-phase gather: a
    // wave IS a phase, so the weights of a tap step are read by ONE wave -- they need no rendezvous at all.  Each wave
    // streams its own 64 columns through a two-stage ring (stage = k step: 64 rows x 64 bytes), waits on its own vmcnt,
    // refills a stage as soon as its own reads of it have returned; only the halo is shared, and the workgroup meets ONCE
    // PER K SLICE (96 MFMAs per wave), where the 8-wave instance meets every tap step.
    // In-order vmcnt, per wave and tap step: [F0: 4 DMAs, next step's k step 0] [H: 0-2 halo pieces] [F1: 4 DMAs, k step 1].
    // "k step 0 of this tap has landed" = everything but the youngest F1 (4) [+ the H in front of it] -> vmcnt(4);
    // "k step 1 has landed" = everything but the F0 just issued -> vmcnt(4); the last tap step of the last slice issues
    // nothing, its second wait is vmcnt(0).
    for (int s = 0; s < nslices; ++s) {
      const int hb = s & 1;
      const bool more = s + 1 < nslices;
      // the slice's halo: own pieces landed (12 weight DMAs were issued behind the last one), own reads of the buffer
      // the NEXT slice's pieces go to returned, everybody here; covers "k step 0 of tap 0 has landed" too
      if (abl & 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      static_for<NST>([&](auto tc_) {
        constexpr int t = decltype(tc_)::value;
        if constexpr (t > 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        // stage 0's reads first: the stage is refilled as soon as THEY have returned (the 4 MT A reads behind them may be
        // in flight), a whole tap step before its next use -- issued behind the first MFMA group the refill had half a
        // step, less than a DMA's round trip under load
        ADVOC_P3_LOAD_B(b0, 0, 0);
        ADVOC_P3_LOAD_A(a0, hb, t, 0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * MT) : "memory");       // (a 4-bit counter: 2 MT = 8 A reads behind them)
        if constexpr (t + 1 < NST) { ADVOC_P3_BQ(s, t + 1, 0); } else { if (more) ADVOC_P3_BQ(s + 1, 0, 0); }
        ADVOC_P3_LOAD_A(a1, hb, t, 1);
        ADVOC_P3_MFMA(a0, b0);
        if constexpr (t + 1 < NST) {
          asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
          if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if constexpr (t * W * HPS < C::HALO_BLOCKS) { if (more) ADVOC_P3_HALO(s + 1, t, hb ^ 1); }
        ADVOC_P3_LOAD_B(b0, 1, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (t + 1 < NST) { ADVOC_P3_BQ(s, t + 1, 1); } else { if (more) ADVOC_P3_BQ(s + 1, 0, 1); }
        ADVOC_P3_MFMA(a1, b0);
      });
    }
  } else if constexpr (C::NBUF == 4) {
    // Two taps per rendezvous (NST = 4: taps 0, 1 in buffers 0, 1, taps 2, 3 in buffers 2, 3): the pair that is not being
    // read is filled, with the halo pieces of both its steps, behind the first MFMA group of the pair that is.
    static_assert(C::NBUF != 4 || (NST == 4 && !C::CARRY && !C::DOUBLE_B), "two tap pairs per K slice");
#define ADVOC_P3_TAP(T, BUF, ISSUE)                                                                       \
  ADVOC_P3_LOAD_A(a0, hb, T, 0);                                                                          \
  ADVOC_P3_LOAD_B(b0, BUF, 0);                                                                            \
  ADVOC_P3_LOAD_A(a1, hb, T, 1);                                                                          \
  ADVOC_P3_MFMA(a0, b0);                                                                                  \
  ISSUE;                                                                                                  \
  ADVOC_P3_LOAD_B(b0, BUF, 1);                                                                            \
  ADVOC_P3_MFMA(a1, b0);
    for (int s = 0; s < nslices; ++s) {
      const int hb = s & 1;
      const bool more = s + 1 < nslices;
      ADVOC_P3_RENDEZVOUS();
      ADVOC_P3_TAP(0, 0, {
        ADVOC_P3_B(s, 2, 2);
        ADVOC_P3_B(s, 3, 3);
        if (more) { ADVOC_P3_HALO(s + 1, 0, hb ^ 1); ADVOC_P3_HALO(s + 1, 1, hb ^ 1); }
      })
      ADVOC_P3_TAP(1, 1, {})
      ADVOC_P3_RENDEZVOUS();
      ADVOC_P3_TAP(2, 2, {
        if (more) {
          ADVOC_P3_B(s + 1, 0, 0);
          ADVOC_P3_B(s + 1, 1, 1);
          ADVOC_P3_HALO(s + 1, 2, hb ^ 1);
          ADVOC_P3_HALO(s + 1, 3, hb ^ 1);
        }
      })
      ADVOC_P3_TAP(3, 3, {})
    }
#undef ADVOC_P3_TAP
  } else if constexpr (C::STAGGER) {
    // ---- (r4) TWO WAVE GROUPS, ONE PHASE APART.  A step of a wave is four phases: R1 (fragment reads of the step: A of
    // both k steps, B of the first), M1 (24 MFMAs), R2 (B of the second k step), M2 (24 MFMAs), a workgroup barrier behind
    // each.  Waves 4-7 run ONE BARRIER BEHIND waves 0-3 (an extra barrier in front of their loop, one behind the others'),
    // and wave w shares its SIMD with wave w + 4: whenever one wave of a SIMD multiplies the other one reads, so the matrix
    // pipe never waits for a fragment read and the LDS never serves all eight waves at once -- what r3 measured as the
    // "rendezvous window" (NOTEBOOK.md section 7a item 5: -33 % with the barrier ablated) without giving up the rendezvous.
    //   * early waves issue the DMAs of step n + 1 (B tile, halo piece) at the end of their R1(n), late waves theirs at the
    //     start of their M2(n - 1): the same barrier interval, right behind the barrier at which every wave has retired
    //     (lgkmcnt 0) its reads of the stage being refilled, and four intervals before anyone reads the new tile;
    //   * everybody waits for its own DMAs (vmcnt 0) in front of the last barrier before early's R1(n + 1);
    //   * every barrier is one asm statement with the waits in it and a scheduling fence either side: MFMAs stay in
    //     their phase.
    const bool late = wave >= W / 2;
    const int nsteps = nslices * NST;
    // (A/B builds only, ADVOC_P3_SETPRIO: the wave that multiplies outranks the wave of its SIMD that reads / issues DMAs)
#ifdef ADVOC_P3_SETPRIO
#define ADVOC_P3_PRIO(X) __builtin_amdgcn_s_setprio(X)
#else
#define ADVOC_P3_PRIO(X)
#endif
#define ADVOC_P3_BAR(WAITS)                                                                               \
    {                                                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                  \
      if (abl & 4) asm volatile(WAITS "s_nop 0" ::: "memory");     /* (timing only: the waits without the barrier) */ \
      else asm volatile(WAITS "s_barrier" ::: "memory");                                                  \
      __builtin_amdgcn_sched_barrier(0);                                                                  \
    }
    // the DMAs that feed step N: its B tile into stage N & 1, and piece (N - 1) % NST of the halo of the slice behind
    // step N - 1's -- i.e. the piece the unstaggered loop issues during step N - 1
#define ADVOC_P3_FEED(N)                                                                                  \
    {                                                                                                     \
      const int n_ = (N);                                                                                 \
      if (n_ < nsteps) {                                                                                  \
        const int sl_ = n_ / NST, tt_ = n_ - sl_ * NST;                                                   \
        if (n_ & 1) { ADVOC_P3_B(sl_, tt_, 1); } else { ADVOC_P3_B(sl_, tt_, 0); }                        \
      }                                                                                                   \
      if (n_ >= 1 && !(abl & 4096)) {                                                                     \
        const int ps_ = (n_ - 1) / NST, pt_ = (n_ - 1) - ps_ * NST;                                       \
        if (ps_ + 1 < nslices) {                                                                          \
          if (ps_ & 1) { ADVOC_P3_HALO(ps_ + 1, pt_, 0); } else { ADVOC_P3_HALO(ps_ + 1, pt_, 1); }       \
        }                                                                                                 \
      }                                                                                                   \
    }
    if (late) ADVOC_P3_FEED(1);              // (their "M2(-1)")
    ADVOC_P3_BAR("s_waitcnt vmcnt(0)\n\t");  // slice 0's halo and B(0) have landed, from everybody
    if (late) ADVOC_P3_BAR("");
#ifndef ADVOC_P3_STEP_LOOP
    // (r5) THE SAME SCHEDULE WITH A K SLICE AS THE LOOP BODY.  Measured in shader cycles with parts of the loop compiled out
    // (profiles/r05_layer4_k_loop_cycles.md): fragment reads 0, barriers -2 %, but the halo pieces -9 % although they are a
    // tenth of the DMA bytes, and only -3 % when the piece's DMA instruction goes and its address arithmetic stays: what cost
    // was the CONTROL FLOW of the feeds -- in the step loop below (r4; ADVOC_P3_STEP_LOOP builds) slice and tap come from a
    // division of the step index and every feed asks at run time which stage, which halo buffer, which piece, is there a
    // next slice, is there a next step: basic-block boundaries around every MFMA group, the same thing the ablation switches
    // had been.  With the 16 tap steps of a slice unrolled the tap is a compile-time constant: stage, halo piece (and whether
    // the step carries one at all: 6 of 16) and whether a B feed stays inside the slice are decided by the compiler; what is
    // left at run time is "is there a next slice" in 9 of a slice's 32 feeds.  layer_4 forward, 128 images: 4.18 M -> 3.82 M
    // cycles (-8.6 %), 2.30 -> 2.20 ms (the clock gives 4 % back: 1.82 -> 1.74 GHz).  (Two loops -- halo steps / plain steps --
    // or one per wave group spill 900-2 400 registers; a single loop over the unrolled slice does not.)
    for (int s = 0; s < nslices; ++s) {
      const int hb = s & 1;
      const bool more = s + 1 < nslices;
      static_for<NST>([&](auto tc_) {
        constexpr int t = decltype(tc_)::value, u = t & 1;
        ADVOC_P3_LOAD_A(a0, hb, t, 0);
        ADVOC_P3_LOAD_B(b0, u, 0);
        ADVOC_P3_LOAD_A(a1, hb, t, 1);
        if (!late) {
          if constexpr (t + 1 < NST) { ADVOC_P3_B(s, t + 1, u ^ 1); } else { if (more) ADVOC_P3_B(s + 1, 0, u ^ 1); }
          if constexpr (t * W * HPS < C::HALO_BLOCKS) { if (more) ADVOC_P3_HALO(s + 1, t, hb ^ 1); }
        }
        ADVOC_P3_BAR("s_waitcnt lgkmcnt(0)\n\t");
        ADVOC_P3_MFMA(a0, b0);
        ADVOC_P3_BAR("");
        ADVOC_P3_LOAD_B(b0, u, 1);
        if (late) ADVOC_P3_BAR("s_waitcnt vmcnt(0) lgkmcnt(0)\n\t") else ADVOC_P3_BAR("s_waitcnt lgkmcnt(0)\n\t");
        if (late) {
          if constexpr (t + 2 < NST) { ADVOC_P3_B(s, t + 2, u); } else { if (more) ADVOC_P3_B(s + 1, t + 2 - NST, u); }
          if constexpr (t + 1 < NST) {
            if constexpr ((t + 1) * W * HPS < C::HALO_BLOCKS) { if (more) ADVOC_P3_HALO(s + 1, t + 1, hb ^ 1); }
          } else {
            if (s + 2 < nslices) ADVOC_P3_HALO(s + 2, 0, hb);
          }
        }
        ADVOC_P3_MFMA(a1, b0);
        if (late) ADVOC_P3_BAR("") else ADVOC_P3_BAR("s_waitcnt vmcnt(0)\n\t");
      });
    }
#else
    for (int n = 0; n < nsteps; n += 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int s = (n + u) / NST, t = (n + u) - s * NST;
        const int hb = s & 1;
        // R1
        ADVOC_P3_LOAD_A(a0, hb, t, 0);
        ADVOC_P3_LOAD_B(b0, u, 0);
        ADVOC_P3_LOAD_A(a1, hb, t, 1);
        if (!late) ADVOC_P3_FEED(n + u + 1);
        ADVOC_P3_BAR("s_waitcnt lgkmcnt(0)\n\t");
        // M1
        ADVOC_P3_PRIO(1);
        ADVOC_P3_MFMA(a0, b0);
        ADVOC_P3_PRIO(0);
        ADVOC_P3_BAR("");
        // R2
        ADVOC_P3_LOAD_B(b0, u, 1);
        if (late) ADVOC_P3_BAR("s_waitcnt vmcnt(0) lgkmcnt(0)\n\t") else ADVOC_P3_BAR("s_waitcnt lgkmcnt(0)\n\t");
        // M2
        if (late) ADVOC_P3_FEED(n + u + 2);
        ADVOC_P3_PRIO(1);
        ADVOC_P3_MFMA(a1, b0);
        ADVOC_P3_PRIO(0);
        if (late) ADVOC_P3_BAR("") else ADVOC_P3_BAR("s_waitcnt vmcnt(0)\n\t");
      }
    }
#endif
    if (!late) ADVOC_P3_BAR("");
#undef ADVOC_P3_FEED
#undef ADVOC_P3_BAR
#undef ADVOC_P3_PRIO
  } else
  for (int s = 0; s < nslices; ++s) {
    const int hb = s & 1;
    const bool more = s + 1 < nslices;
    // (unrolling this loop -- the tap step a compile-time constant here too -- changes nothing: 1.00 on every instance, r5)
    for (int t = 0; t < NST; t += 2) {
#ifdef ADVOC_P3_IGLP
      __builtin_amdgcn_iglp_opt(ADVOC_P3_IGLP);
#endif
      // step t (stage 0)
      ADVOC_P3_RENDEZVOUS();
      if (C::DMA_POS == 0) {
        ADVOC_P3_B(s, t + 1, 1);
        if (more) ADVOC_P3_HALO(s + 1, t, hb ^ 1);
      }
      if (!C::CARRY || t == 0) ADVOC_P3_LOAD_A(a0, hb, t, 0);
      ADVOC_P3_LOAD_B(b0, 0, 0);
      ADVOC_P3_LOAD_A(a1, hb, t, 1);
      if (C::DOUBLE_B) ADVOC_P3_LOAD_B(b1, 0, 1);
      if (C::DMA_POS == 1) {
        ADVOC_P3_B(s, t + 1, 1);
        if (more) ADVOC_P3_HALO(s + 1, t, hb ^ 1);
      }
      ADVOC_P3_MFMA(a0, b0);
      if (C::DMA_POS == 2) {
        ADVOC_P3_B(s, t + 1, 1);
        if (more) ADVOC_P3_HALO(s + 1, t, hb ^ 1);
      }
      if (C::CARRY) ADVOC_P3_LOAD_A(a0, hb, t + 1, 0);
      if (C::DOUBLE_B) {
        ADVOC_P3_MFMA(a1, b1);
      } else {
        ADVOC_P3_LOAD_B(b0, 0, 1);
        ADVOC_P3_MFMA(a1, b0);
      }
      // step t + 1 (stage 1)
      ADVOC_P3_RENDEZVOUS();
      if (C::DMA_POS == 0) {
        if (t + 2 < NST) {
          ADVOC_P3_B(s, t + 2, 0);
        } else if (more) {
          ADVOC_P3_B(s + 1, 0, 0);
        }
        if (more) ADVOC_P3_HALO(s + 1, t + 1, hb ^ 1);
      }
      if (!C::CARRY) ADVOC_P3_LOAD_A(a0, hb, t + 1, 0);
      ADVOC_P3_LOAD_B(b0, 1, 0);
      ADVOC_P3_LOAD_A(a1, hb, t + 1, 1);
      if (C::DOUBLE_B) ADVOC_P3_LOAD_B(b1, 1, 1);
      if (C::DMA_POS == 1) {
        if (t + 2 < NST) {
          ADVOC_P3_B(s, t + 2, 0);
        } else if (more) {
          ADVOC_P3_B(s + 1, 0, 0);
        }
        if (more) ADVOC_P3_HALO(s + 1, t + 1, hb ^ 1);
      }
      ADVOC_P3_MFMA(a0, b0);
      if (C::DMA_POS == 2) {
        if (t + 2 < NST) {
          ADVOC_P3_B(s, t + 2, 0);
        } else if (more) {
          ADVOC_P3_B(s + 1, 0, 0);
        }
        if (more) ADVOC_P3_HALO(s + 1, t + 1, hb ^ 1);
      }
      if (C::CARRY && t + 2 < NST) ADVOC_P3_LOAD_A(a0, hb, t + 2, 0);
      if (C::DOUBLE_B) {
        ADVOC_P3_MFMA(a1, b1);
      } else {
        ADVOC_P3_LOAD_B(b0, 1, 1);
        ADVOC_P3_MFMA(a1, b0);
      }
    }
  }
#undef ADVOC_P3_LOAD_A
#undef ADVOC_P3_LOAD_B
#undef ADVOC_P3_MFMA
#undef ADVOC_P3_RENDEZVOUS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // the bias of the wave's column blocks, loaded in front of the next tile's prologue (see `unscale` above)
  float4 bias4v[NT];
  {
    int lb = lane;
    asm volatile("" : "+v"(lb));
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      bias4v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      const int nb = n0 + ncol0 + j * 32 + 4 * (lb & 7);
      if (!BWD && p.bias && nb < p.n_total) bias4v[j] = *reinterpret_cast<const float4*>(p.bias + nb);
    }
  }
  // ---- (r5) the NEXT tile's prologue, issued before this tile's epilogue: halo buffer 0 and weight stage 0 (1) are dead from
  // here on -- every fragment read of the tile has returned behind the barrier above -- and the epilogue's scratch lives in
  // halo buffer 1 and the weight stage the prologue does not fill (EPI_T_OFF / EPI_PIX_OFF).  The DMAs are the oldest
  // entries of the wave's memory queue: every counted wait of the epilogue covers them, the K loop's first rendezvous
  // (vmcnt 0) finds them landed.
  if (C::EARLY_PROLOGUE) {
    const int tile_n = tile + per_xcd;
    if (tile_n < t_hi) {
      const int nt = tile_n / npatch;
      const int pid = tile_n - nt * npatch;
      const int img = pid / ppi;
      const int pin = pid - img * ppi;
      const int gy0 = (pin / g.px) * C::PROWS, gx0 = (pin % g.px) * 16;
      const int n0 = nt * BN;
      int b_off[2];
      const bool b_cols = NPH != 4 || n0 + ((wave * (C::BROWS / W)) & 63) < p.n_total;
      {
        const int q = wave * (C::BROWS / W) + lrow;
        const int col = NPH == 4 ? (q & 63) : q;
        const int rowb = ((n0 + col) * ktot) * 4;
        b_off[0] = rowb + ((lpos ^ (lrow >> 1)) * 16);
        b_off[1] = rowb + ((lpos ^ (lrow >> 1) ^ 4) * 16);
        if (C::P4W) {
          const int c = (ln & 3) ^ ((ln >> 4) & 3);
          b_off[0] = ((n0 + (ln >> 2)) * ktot) * 4 + (c >> 1) * 64 + (c & 1) * 16;
        }
      }
      constexpr bool kAsmDma = true;
#pragma unroll
      for (int t = 0; t < NST; ++t) ADVOC_P3_HALO(0, t, 0);
      if constexpr (C::P4W) {
        ADVOC_P3_BQ(0, 0, 0);
        ADVOC_P3_BQ(0, 0, 1);
      } else {
        ADVOC_P3_B(0, 0, 0);
        if (C::NBUF == 4) ADVOC_P3_B(0, 1, 1);
      }
    }
  }
#undef ADVOC_P3_HALO
#undef ADVOC_P3_B
#undef ADVOC_P3_BQ
  if (abl & 64) continue;         // (timing experiments only: the K loop without its epilogue -- and, the accumulators being dead, without its MFMAs)
  if (abl & 8192) {               // (r6, timing only: no epilogue, the accumulators kept alive -- the ceiling of a perfectly hidden epilogue)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) asm volatile("" ::"v"(acc[i][j]));
    __syncthreads();
    continue;
  }

  // ---- epilogue (igemm_h3.hip's, per wave): pixel table of the wave's points for its phase, LDS transpose,
  // 16-byte stores with the fused bias / dropout / activation-gradient / two-destination logic ----
  int le = lane;
  asm volatile("" : "+v"(le));
  const int half_e = le >> 5, l32_e = le & 31;
  constexpr int PW_ = C::PTS_W;
  int* s_pix = reinterpret_cast<int*>(smem_b + C::EPI_PIX_OFF) + wave * 2 * PW_;
  for (int r = le; r < PW_; r += 64) {
    const int pt = wm * PW_ + r;
    const int gy = gy0 + (pt >> 4), gx = gx0 + (pt & 15);
    int pix0 = -1, pix1 = -1;
    if (gy < p.gh && gx < p.gw) {
      const int oy = gy * p.osy + p.ooy[phase], ox = gx * p.osx + p.oox[phase];
      if (oy < p.out_h && ox < p.out_w) {
        pix0 = (img * p.out_h + oy) * p.d[0].pitch + ox;
        pix1 = (img * p.out_h + oy) * p.d[1].pitch + ox;
      }
    }
    s_pix[r] = pix0;
    s_pix[PW_ + r] = pix1;
  }
  wave_lds_sync();

  constexpr int LDT = 36;
  float* T = reinterpret_cast<float*>(smem_b + C::EPI_T_OFF) + wave * (32 * LDT);
  const int trow = le >> 3, tq = le & 7;
  const bool use_grad = BWD && p.grad_act != ADVOC_ACT_NONE && !(abl & 32);     // (DIAG builds, timing only: 32 = no pre-activation loads)
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    // destination of this column block (channels [0, n_split) -> d[0], the rest -> d[1]) and its per-channel vectors
    const int nt0 = n0 + ncol0 + j * 32;
    const int di = nt0 >= p.n_split ? 1 : 0;
    float* const dp = di ? p.d[1].p : p.d[0].p;
    if (dp == nullptr || nt0 >= p.n_total) continue;
    const float* const d_xpre = di ? p.d[1].xpre : p.d[0].xpre;
    const uint16_t* const d_ximg = di ? p.d[1].ximg : p.d[0].ximg;
    const uint8_t* const d_gmask = di ? p.d[1].gmask : p.d[0].gmask;
    const float d_gmask_scale = di ? p.d[1].gmask_scale : p.d[0].gmask_scale;
    const int d_c = di ? p.d[1].c : p.d[0].c;
    const bool d_accum = (di ? p.d[1].accum : p.d[0].accum) != 0;
    const int ch = (di ? nt0 - p.n_split : nt0) + 4 * tq;
    const float4 bias4 = bias4v[j];
    float4 gs4 = make_float4(1.f, 1.f, 1.f, 1.f), gh4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* const gsc = di ? p.d[1].gscale : p.d[0].gscale;
    const float* const gsh = di ? p.d[1].gshift : p.d[0].gshift;
    if (BWD && gsc) {
      gs4 = *reinterpret_cast<const float4*>(gsc + ch);
      gh4 = *reinterpret_cast<const float4*>(gsh + ch);
    }
    // Loads and stores retire through ONE in-order counter here, so a global load placed after a store waits for the
    // store's round trip: the plain per-row load -> wait -> store chain costs one memory latency per 16 bytes (measured:
    // 2/3 of the time of the 64-column backward-data launches).  Per 32-row block ALL global loads (pre-activation values,
    // masks, the value to add to) are issued together, and those of block i + 1 go out BEFORE the stores of block i
    // (their registers are free once block i's values are computed).
    // Every access is a BUFFER instruction whose offset is out of range for the rows that store nothing (loads return
    // zero, stores are dropped) and for the tensors this launch does not have: no branch and no exec mask around any of
    // them, so the number of stores between a load and its use is known at compile time and the wait for block i + 1's
    // loads leaves block i's four stores in flight (s_waitcnt vmcnt(4)) -- with `if (row valid) store` the compiler had
    // to wait for vmcnt(0), i.e. for the round trip of every block's stores.
    constexpr unsigned kOob = 0xffffff00u;              // >= num_records of every descriptor below
    // (d0_no_store: destination 0 exists as the image only -- every fp32 store to it is dropped by the range check)
    const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(
        dp, 0, (emit0 && di == 0 && p.d0_no_store) ? 0u : kOob, 0x00020000);
    const bool has_mask = BWD ? d_gmask != nullptr : p.y_mask != nullptr;
    const uint8_t* const mask_p = BWD ? d_gmask : p.y_mask;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        GI ? reinterpret_cast<void*>(const_cast<uint16_t*>(use_grad ? d_ximg : reinterpret_cast<const uint16_t*>(dp)))
           : reinterpret_cast<void*>(const_cast<float*>(use_grad ? d_xpre : dp)), 0, use_grad ? kOob : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc(
        has_mask ? const_cast<uint8_t*>(mask_p) : reinterpret_cast<uint8_t*>(dp), 0, has_mask ? kOob : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(dp, 0, (BWD && d_accum) ? kOob : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_e0 = __builtin_amdgcn_make_buffer_rsrc(
        emit0 ? reinterpret_cast<void*>(p.oimg[0].img) : reinterpret_cast<void*>(dp), 0, emit0 ? kOob : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_e1 = __builtin_amdgcn_make_buffer_rsrc(
        emit1 ? reinterpret_cast<void*>(p.oimg[1].img) : reinterpret_cast<void*>(dp), 0, emit1 ? kOob : 0u, 0x00020000);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    unsigned off[4];                                   // element offset of the row's 4 channels, kOob for rows without a pixel
    u32x4 xp[4], old[4];
    unsigned mk[4];
    // per-channel sums of this column block over the tile's points (BWD emission: the lower layer's bias gradient): this
    // lane's 4 channels, reduced over the 8 lanes that share them and added to one of kColsumReplicas copies per tile
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
#define ADVOC_P3_PRELOAD(I)                                                                               \
    _Pragma("unroll") for (int ps = 0; ps < 4; ++ps) {                                                    \
      const int pix = s_pix[di * PW_ + (I) * 32 + trow + 8 * ps];                                         \
      off[ps] = pix < 0 ? kOob : (unsigned)(pix * d_c + ch);                                              \
      const unsigned ob = pix < 0 ? kOob : off[ps] * 4u;                                                  \
      if (BWD && GI) {       /* element e -> image halves (e >> 5) * 64 + (e & 31), high plane: 8 bytes for 4 channels */ \
        typedef unsigned u32x2g __attribute__((ext_vector_type(2)));                                      \
        const unsigned ib = pix < 0 ? kOob : (((off[ps] >> 5) << 6) + (off[ps] & 31)) * 2u;               \
        const u32x2g t_ = __builtin_amdgcn_raw_buffer_load_b64(rs_x, ib, 0, 0);                           \
        xp[ps].x = t_.x; xp[ps].y = t_.y;                                                                 \
      } else if (BWD) {                                                                                   \
        xp[ps] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ob, 0, 0);                                   \
      }                                                                                                   \
      if (BWD && !kLean) old[ps] = __builtin_amdgcn_raw_buffer_load_b128(rs_o, ob, 0, 0);                 \
      if (!kNoMask) mk[ps] = __builtin_amdgcn_raw_buffer_load_b32(rs_m, off[ps], 0, 0);                   \
    }
    ADVOC_P3_PRELOAD(0);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * half_e) * LDT + l32_e] = acc[i][j][r];
      wave_lds_sync();
      float4 v[4];
      unsigned so[4];
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        v[ps] = *reinterpret_cast<const float4*>(T + (trow + 8 * ps) * LDT + 4 * tq);
        v[ps].x = fmaf(v[ps].x, unscale, bias4.x); v[ps].y = fmaf(v[ps].y, unscale, bias4.y);
        v[ps].z = fmaf(v[ps].z, unscale, bias4.z); v[ps].w = fmaf(v[ps].w, unscale, bias4.w);
        if (GI && use_grad) {
          // fp16 h > 0 <=> its bits, minus one, are below 0x7fff as an unsigned (zero, negatives and -0 fall out)
          const unsigned lo = xp[ps].x, hi = xp[ps].y;
          v[ps].x *= ((lo & 0xffffu) - 1u) < 0x7fffu ? 1.f : gslope; v[ps].y *= ((lo >> 16) - 1u) < 0x7fffu ? 1.f : gslope;
          v[ps].z *= ((hi & 0xffffu) - 1u) < 0x7fffu ? 1.f : gslope; v[ps].w *= ((hi >> 16) - 1u) < 0x7fffu ? 1.f : gslope;
        } else if (use_grad) {
          float4 x = make_float4(__uint_as_float(xp[ps].x), __uint_as_float(xp[ps].y), __uint_as_float(xp[ps].z),
                                 __uint_as_float(xp[ps].w));
          x.x = x.x * gs4.x + gh4.x; x.y = x.y * gs4.y + gh4.y; x.z = x.z * gs4.z + gh4.z; x.w = x.w * gs4.w + gh4.w;
          v[ps].x *= x.x > 0.f ? 1.f : gslope; v[ps].y *= x.y > 0.f ? 1.f : gslope;
          v[ps].z *= x.z > 0.f ? 1.f : gslope; v[ps].w *= x.w > 0.f ? 1.f : gslope;
        }
        if (!kNoMask && has_mask) {
          const float ms = BWD ? d_gmask_scale : p.y_mask_scale;
          v[ps].x *= (float)(mk[ps] & 0xffu) * ms; v[ps].y *= (float)((mk[ps] >> 8) & 0xffu) * ms;
          v[ps].z *= (float)((mk[ps] >> 16) & 0xffu) * ms; v[ps].w *= (float)(mk[ps] >> 24) * ms;
        }
        if (!kLean && BWD && d_accum) {
          v[ps].x += __uint_as_float(old[ps].x); v[ps].y += __uint_as_float(old[ps].y);
          v[ps].z += __uint_as_float(old[ps].z); v[ps].w += __uint_as_float(old[ps].w);
        }
        if (BWD && di == 1 && p.d1_amax_out && off[ps] != kOob)
          dmax1 = fmaxf(fmaxf(dmax1, fmaxf(fabsf(v[ps].x), fabsf(v[ps].y))), fmaxf(fabsf(v[ps].z), fabsf(v[ps].w)));
        so[ps] = (off[ps] == kOob || (abl & 128)) ? kOob : off[ps] * 4u;       // (128: timing only, every store dropped by the range check)
      }
      if (i + 1 < MT) { ADVOC_P3_PRELOAD(i + 1); }
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        u32x4 sv;
        sv.x = __float_as_uint(v[ps].x); sv.y = __float_as_uint(v[ps].y);
        sv.z = __float_as_uint(v[ps].z); sv.w = __float_as_uint(v[ps].w);
        __builtin_amdgcn_raw_buffer_store_b128(sv, rs_d, so[ps], 0, 0);
      }
      if (emit0 || emit1) {  // (compile-time; unconditional buffer stores: see emit4_buffer)
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          const bool ok = so[ps] != kOob && di == 0;
          if (emit0) emit4_buffer(rs_e0, BWD ? 1.f : p.oimg[0].slope, eup0, v[ps], ok ? so[ps] : kOob, ok, evmax0);
          if (emit1) emit4_buffer(rs_e1, p.oimg[1].slope, eup1, v[ps], ok ? so[ps] : kOob, ok, evmax1);
          if (BWD && emit0 && ok) {
            cs.x += v[ps].x; cs.y += v[ps].y; cs.z += v[ps].z; cs.w += v[ps].w;
          }
        }
      }
      wave_lds_sync();
    }
#undef ADVOC_P3_PRELOAD
    if (BWD && emit0 && p.ocolsum_table) {
#pragma unroll
      for (int sh = 8; sh < 64; sh <<= 1) {
        cs.x += __shfl_xor(cs.x, sh, 64); cs.y += __shfl_xor(cs.y, sh, 64);
        cs.z += __shfl_xor(cs.z, sh, 64); cs.w += __shfl_xor(cs.w, sh, 64);
      }
      if (trow == 0 && di == 0) {
        float* row = p.ocolsum_table + (size_t)(blockIdx.x & (kColsumReplicas - 1)) * d_c + ch;
        unsafeAtomicAdd(row, cs.x); unsafeAtomicAdd(row + 1, cs.y); unsafeAtomicAdd(row + 2, cs.z); unsafeAtomicAdd(row + 3, cs.w);
      }
    }
  }
  __syncthreads();          // the next tile's DMAs overwrite the LDS this epilogue read
  }  // tiles
  if (BWD && p.d1_amax_out) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dmax1 = fmaxf(dmax1, __shfl_xor(dmax1, off, 64));
    if (lane == 0 && dmax1 > 0.f &&
        __float_as_uint(dmax1) > __hip_atomic_load(p.d1_amax_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMax(p.d1_amax_out, __float_as_uint(dmax1));
  }
  if (emit0) emit_finish(p.oimg[0], eup0, evmax0);
  if (emit1) emit_finish(p.oimg[1], eup1, evmax1);
}

template <int NPH, int BWD, int NE, int HALF = 0, int GI = 0>
__global__ __launch_bounds__(512, 1) void patch_gemm_h3_kernel(const GatherGemmParams p, const PatchGeom g) {
#ifdef ADVOC_CLOCK_PROBE
  // Diagnostic build only (tools/micro/build_variant.sh clk "-DADVOC_CLOCK_PROBE"): the shader clock this launch ran at =
  // s_memtime ticks (shader cycles) per s_memrealtime tick (100 MHz), first wave of a few workgroups.
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#endif
  patch_gemm_h3_body<NPH, 8, BWD, NE, HALF, GI>(p, g);
#ifdef ADVOC_CLOCK_PROBE
  __syncthreads();         // (the workgroup's life, not its first wave's: without the K loop's barriers they differ)
  if (threadIdx.x == 0 && (blockIdx.x & 63) == 0) {
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    printf("clk <%d,%d,%d,%d,%d> wg %3d: %llu cycles in %llu ticks of 10 ns = %.3f GHz\n", NPH, BWD, NE, HALF, GI, (int)blockIdx.x,
           c1 - c0, r1 - r0, 0.1 * (double)(c1 - c0) / (double)(r1 - r0));
  }
#endif
}

// (r6) the two-per-CU instance of the four-phase gather: 4 waves, <= 80 KiB of LDS, 256 registers per wave
template <int BWD, int NE, int GI = 0>
__global__ __launch_bounds__(256, 2) void patch4w_gemm_h3_kernel(const GatherGemmParams p, const PatchGeom g) {
#ifdef ADVOC_CLOCK_PROBE
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#endif
  patch_gemm_h3_body<6, 4, BWD, NE, 0, GI>(p, g);
#ifdef ADVOC_CLOCK_PROBE
  __syncthreads();
  if (threadIdx.x == 0 && (blockIdx.x & 63) == 0) {
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    printf("clk <%d,%d,%d,%d,%d> wg %3d: %llu cycles in %llu ticks of 10 ns = %.3f GHz\n", 6, BWD, NE, 0, GI, (int)blockIdx.x,
           c1 - c0, r1 - r0, 0.1 * (double)(c1 - c0) / (double)(r1 - r0));
  }
#endif
}

template <int NPH, int BWD, int NE = 0>
int launch_patch(const GatherGemmParams& p_in, const PatchGeom& g, hipStream_t stream, const char** name_only) {
  using C = PCfg<NPH, NPH == 6 ? 4 : 8>;
  if (BWD && NE == 0 && !name_only && p_in.oimg[0].img) {
    if (p_in.d[0].accum) {       // (the accumulating form exists for the four-phase instance: the encoders' backward-data)
      if constexpr (NPH == 4 || NPH == 6) return launch_patch<NPH, BWD, BWD ? 2 : 0>(p_in, g, stream, name_only);
      else return ADVOC_ERR_UNSUPPORTED;
    }
    return launch_patch<NPH, BWD, BWD ? 1 : 0>(p_in, g, stream, name_only);
  }
  if (!BWD && NE == 0 && !name_only && (p_in.oimg[0].img || p_in.oimg[1].img)) {
    // forward launch with image consumers: the instance compiled for their number, consumers packed into oimg[0 .. n)
    GatherGemmParams q = p_in;
    if (!q.oimg[0].img) { q.oimg[0] = q.oimg[1]; q.oimg[1] = ImgOut{}; }
    if (q.oimg[1].img) return launch_patch<NPH, BWD, BWD ? 0 : 2>(q, g, stream, name_only);
    return launch_patch<NPH, BWD, BWD ? 0 : 1>(q, g, stream, name_only);
  }
  const GatherGemmParams& p = p_in;
  const Tuning& t = tuning();
  if (name_only) {
    static const std::string name = std::string("patch_gemm_h3_kernel<") + std::to_string(NPH) + ", " +
                                    std::to_string(BWD) + ">";
    *name_only = name.c_str();
    return ADVOC_OK;
  }
  void (*kern)(const GatherGemmParams, const PatchGeom);
  if constexpr (NPH == 6) {
    if (p.n_total % 64) return ADVOC_ERR_UNSUPPORTED;
    kern = patch4w_gemm_h3_kernel<BWD, NE>;
  } else {
    kern = (NPH == 4 && p.n_total == 32) ? patch_gemm_h3_kernel<NPH, BWD, NE, NPH == 4 ? 1 : 0> : patch_gemm_h3_kernel<NPH, BWD, NE>;
  }
  if (p.d[0].ximg) {       // (r5) gates from the consuming layer's image: both destinations, full tiles, backward-data
    if (!BWD || (NPH == 4 && p.n_total == 32) || (p.d[1].p && !p.d[1].ximg) || p.d[0].gscale || p.d[1].gscale)
      return ADVOC_ERR_UNSUPPORTED;
    if constexpr (NPH == 6) kern = patch4w_gemm_h3_kernel<BWD, NE, BWD ? 1 : 0>;
    else kern = patch_gemm_h3_kernel<NPH, BWD, NE, 0, BWD ? 1 : 0>;
  }
  const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
  if (attr != hipSuccess) { note_hip_error(attr); return ADVOC_ERR_HIP; }
  // workgroups in whole rows of the 8 XCDs; persistent (default; ADVOC_H3_PATCH_PERSIST=0 for one tile per workgroup):
  // one per CU (160 KB of LDS each), every workgroup walks tiles -- 2-4 % faster on every layer of the model
  const int64_t tiles = (int64_t)p.batch * g.py * g.px * ((p.n_total + C::BN - 1) / C::BN);
  int64_t wgs = (tiles + 7) / 8 * 8;
  if (t.h3_patch_persist && (!BWD || t.h3_patch_persist == 2)) {
    const int64_t cus = persistent_cu_count() * (NPH == 6 ? 2 : 1);      // (NPH = 6: two workgroups per CU)
    if (wgs > cus && cus >= 8) wgs = cus;
  }
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(C::THREADS), C::LDS_BYTES, stream, p, g);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

}  // namespace

// 0: not a patch launch; else NPH (4 | 1) with `g` filled in
int patch_plan(const GatherGemmParams& p, PatchGeom* g) {
  const Tuning& t = tuning();
  if (!t.h3_patch) return 0;
  int nph = 0;
  // (r4) 32 columns (AdVoc-small's decoder_2 forward, encoder_2 / layer_2 backward-data) take the 64-column instance of the
  // four-phase kernel with the upper half of the weight tile masked in the DMA and never stored: half its matrix work is on
  // zeros, and it is still 1.4-2.4 x the masked per-tap tile these launches ran on (140 / 75 TFLOP/s)
  if (p.sy == 1 && p.sx == 1 && p.nphase == 4 && p.ntaps == 4 && p.osy == 2 && p.osx == 2 &&
      (p.n_total % 64 == 0 || (p.n_total == 32 && t.h3_patch_n32)))
    nph = 4;
  else if (p.sy == 1 && p.sx == 1 && p.nphase == 1 && p.ntaps == 16 && p.osy == 1 && p.osx == 1 && p.n_total % 256 == 0)
    nph = 1;
  else if (t.h3_patch_s1n128 && p.sy == 1 && p.sx == 1 && p.nphase == 1 && p.ntaps == 16 && p.osy == 1 && p.osx == 1 &&
           p.n_total % 128 == 0)
    nph = 5;        // (r4) 128 columns: AdVoc-small's layer_4 backward-data ran the per-tap 128 x 128 tile at 290 TFLOP/s
  else if (t.h3_patch_s2 && p.sy == 2 && p.sx == 2 && p.nphase == 1 && p.ntaps == 16 && p.osy == 1 && p.osx == 1 &&
           p.n_total % 128 == 0)
    nph = p.n_total % 256 == 0 ? 2 : 3;
  if (!nph) return 0;
  // (r6) the four-phase gather on two 4-wave workgroups per CU (NPH = 6): ADVOC_H3_PATCH_2WG 1 = backward-data launches, 2 = all
  if (nph == 4 && p.n_total % 64 == 0 && t.h3_patch_2wg) {
    const bool bwd = p.grad_act != ADVOC_ACT_NONE || p.d[0].xpre || p.d[1].p || p.d[0].accum;
    if (bwd || t.h3_patch_2wg >= 2) nph = 6;
  }
  if (p.gh < (nph == 6 ? 8 : 16) || p.gw < 16) return 0;
  *g = PatchGeom{};
  if (nph >= 2 && nph <= 3) {
    // every tap (dy, dx) reads input (2 gy + dy, 2 gx + dx) = pixel (gy + a, gx + b) of plane (py, px), dy = 2 a + py
    int cnt[4] = {0, 0, 0, 0};
    int ay[16], ax[16];
    g->s2_a0y[0] = g->s2_a0y[1] = g->s2_a0x[0] = g->s2_a0x[1] = 127;
    for (int i = 0; i < 16; ++i) {
      const int dy = (int)(int8_t)(p.tap[0][i] & 0xff), dx = (int)(int8_t)((p.tap[0][i] >> 8) & 0xff);
      const int py = dy & 1, px = dx & 1;                  // (two's complement: -1 & 1 == 1)
      ay[i] = (dy - py) / 2; ax[i] = (dx - px) / 2;
      g->s2_a0y[py] = ay[i] < g->s2_a0y[py] ? ay[i] : g->s2_a0y[py];
      g->s2_a0x[px] = ax[i] < g->s2_a0x[px] ? ax[i] : g->s2_a0x[px];
    }
    for (int i = 0; i < 16; ++i) {
      const int dy = (int)(int8_t)(p.tap[0][i] & 0xff), dx = (int)(int8_t)((p.tap[0][i] >> 8) & 0xff);
      const int py = dy & 1, px = dx & 1, pl = 2 * py + px;
      const int ry = ay[i] - g->s2_a0y[py], rx = ax[i] - g->s2_a0x[px];
      if (cnt[pl] >= 4 || ry < 0 || ry > 1 || rx < 0 || rx > 1) return 0;     // 2x2 taps per plane: 17 x 17 halos
      g->s2_tap[4 * pl + cnt[pl]++] = ry | (rx << 8) | ((p.tap[0][i] >> 16) << 16);
    }
    if (cnt[0] != 4 || cnt[1] != 4 || cnt[2] != 4 || cnt[3] != 4) return 0;
    g->hh = g->hw = 17;
  } else {
    int dy0 = 127, dy1 = -127, dx0 = 127, dx1 = -127;
    for (int ph = 0; ph < p.nphase; ++ph)
      for (int i = 0; i < p.ntaps; ++i) {
        const int dy = (int)(int8_t)(p.tap[ph][i] & 0xff), dx = (int)(int8_t)((p.tap[ph][i] >> 8) & 0xff);
        dy0 = dy < dy0 ? dy : dy0; dy1 = dy > dy1 ? dy : dy1;
        dx0 = dx < dx0 ? dx : dx0; dx1 = dx > dx1 ? dx : dx1;
      }
    const int e = (nph == 4 || nph == 6) ? 2 : 3;                 // the kernels are compiled for halos of exactly (16 + e) x (16 + e)
    if (dy1 - dy0 != e || dx1 - dx0 != e) return 0;
    g->dy0 = dy0; g->dx0 = dx0;
    g->hh = (nph == 6 ? 8 : 16) + e; g->hw = 16 + e;
  }
  const int prows = nph == 6 ? 8 : 16;
  g->py = (p.gh + prows - 1) / prows; g->px = (p.gw + 15) / 16;
  if (nph == 6)      // half of a tile's life at two waves per SIMD: slices x 4 tap steps x 48 MFMAs x 32 cycles (x 2 / 2)
    g->delay2 = (int)((int64_t)((p.c0 + p.c1) / 32) * 4 * 48 * 32 * t.h3_patch_2wg_delay / 100);
  int gw_cov = p.gw;
  if (t.h3_patch_rem && p.gw >= 32 && p.gw % 16 >= 1 && p.gw % 16 <= 4) {      // 16 px + a few columns: see the launcher
    g->px = p.gw / 16;
    g->rem = p.gw % 16;
    gw_cov = g->px * 16;
  }
  g->nblocks = (g->hh * g->hw + 7) / 8;
  g->ablate = t.h3_patch_ablate;
  // rows the patches add beyond the grid are computed and thrown away
  if ((int64_t)g->py * g->px * prows * 16 * 100 > (int64_t)p.gh * gw_cov * 125) return 0;
  const int bn = (nph == 4 || nph == 6) ? 64 : ((nph == 3 || nph == 5) ? 128 : 256);
  const int64_t wgs = (int64_t)p.batch * g->py * g->px * ((p.n_total + bn - 1) / bn);
  if (wgs < t.h3_patch_min_wgs) return 0;
  return nph;
}

int launch_patch_gemm_h3(const GatherGemmParams& p, const PatchGeom& g, int nph, hipStream_t stream,
                         const char** name_only) {
  // the backward-data description is the one with an activation gradient or a second / accumulating destination
  const bool bwd = p.grad_act != ADVOC_ACT_NONE || p.d[0].xpre || p.d[1].p || p.d[0].accum;
  if (nph == 4) return bwd ? launch_patch<4, 1>(p, g, stream, name_only) : launch_patch<4, 0>(p, g, stream, name_only);
  if (nph == 6) return bwd ? launch_patch<6, 1>(p, g, stream, name_only) : launch_patch<6, 0>(p, g, stream, name_only);
  if (nph == 2) return bwd ? launch_patch<2, 1>(p, g, stream, name_only) : launch_patch<2, 0>(p, g, stream, name_only);
  if (nph == 3) return bwd ? launch_patch<3, 1>(p, g, stream, name_only) : launch_patch<3, 0>(p, g, stream, name_only);
  if (nph == 5) return bwd ? launch_patch<5, 1>(p, g, stream, name_only) : launch_patch<5, 0>(p, g, stream, name_only);
  return bwd ? launch_patch<1, 1>(p, g, stream, name_only) : launch_patch<1, 0>(p, g, stream, name_only);
}

}  // namespace advoc
