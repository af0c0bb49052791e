// Diagnostic switches of the library, read from the environment ONCE (first use) instead of on every launch;
// advoc_tuning_reload() (C ABI) re-reads them -- tests and A/B measurements flip a variable and call it.
#pragma once

namespace advoc {

struct Tuning {
  int igemm_splitk;     // ADVOC_IGEMM_SPLITK   0: every launch one K pass (bitwise run-to-run reproducible)
  int igemm_tail;       // ADVOC_IGEMM_TAIL     0: no tail split
  int igemm_x6;         // ADVOC_IGEMM_X6       0: every contraction on the fp32 MFMA kernels
  long long igemm_x6_wide;   // ADVOC_IGEMM_X6_WIDE  >= 0: tile threshold of the 128x256 register-split tile
  int igemm_bk;         // ADVOC_IGEMM_BK       16 | 32 (fp32 kernels)
  int igemm_x6_n32;     // ADVOC_IGEMM_X6_N32   0: 32-channel outputs stay on fp32 MFMA
  int igemm_x6_tile;    // ADVOC_IGEMM_X6_TILE  1..3: force a register-split tile
  int igemm_tile;       // ADVOC_IGEMM_TILE     1..4: force an fp32 tile
  int igemm_korder;     // ADVOC_IGEMM_KORDER   0 | 1, -1: per-layer rule
  int wgrad_x6;         // ADVOC_WGRAD_X6       0: fp32 weight-gradient kernels; 2: also the 128x64 split tile
  int wgrad_h3;         // ADVOC_WGRAD_H3       0: weight gradients stay on the register-split / fp32 kernels
  int wgrad_h3_min_m;   // ADVOC_WGRAD_H3_MIN_M smallest pixel grid (batch x gh x gw) that takes the image-based weight gradient
  int wgrad_h3_tile;    // ADVOC_WGRAD_H3_TILE  1: 128 x 128, 2: 256 x 256 forced (where the shape allows)
  int wgrad_h3_ordered; // ADVOC_WGRAD_H3_ORDERED 0: K slices of the image weight gradient always meet in fp32 atomics; 1: the 256 x 256 tile's through wgrad_ws when the layer has it; 2: the 128 x 128 tile's too
  int wgrad_h3_rows;    // ADVOC_WGRAD_H3_ROWS    1: the image weight gradient's address arithmetic per grid ROW (scalar) where rows have >= 32 points; 0: always per slot (flat); 2: row mode or unsupported (tests)
  int wgrad_h3_rounds;  // ADVOC_WGRAD_H3_ROUNDS  > 0: K chunks per tile = this many rounds of the chip (default: 1)
  int h3;               // ADVOC_H3             0: no operand-image kernels (register-split path instead)
  int h3_tile;          // ADVOC_H3_TILE        1: 128x128, 4: 128x64, 5: 256x256 (8 waves) forced
  int h3_skip_prep;     // ADVOC_H3_SKIP_PREP   1: (micro-benchmarks only; -DADVOC_DIAG builds only) reuse the images already in the workspace
  int h3_min_tiles;     // ADVOC_H3_MIN_TILES   smallest launch (128-row x 128-column tiles) that takes the image path (4: with the
                        //                      workspace K split the image kernels beat the r1 ones down to the 1 x 3-point layers)
  int h3_patch;         // ADVOC_H3_PATCH       0: stride-1 gathers stay on the per-tap tiles of igemm_h3.hip
  int h3_patch_min_wgs; // ADVOC_H3_PATCH_MIN_WGS  smallest launch (workgroups) that takes the patch kernel
  int h3_patch_s2;      // ADVOC_H3_PATCH_S2    0: stride-2 gathers stay on the per-tap tiles (no parity-plane patch kernel)
  int h3_patch_n32;     // ADVOC_H3_PATCH_N32   0: 32-column four-phase gathers stay on the masked 128 x 64 per-tap tile
  int h3_patch_rem;     // ADVOC_H3_PATCH_REM   0: grids of 16 n + 1..4 columns get a whole extra patch column instead of a per-tap launch
  int h3_patch_persist; // ADVOC_H3_PATCH_PERSIST  0: one tile per workgroup instead of one workgroup per CU walking tiles;
                        //                          1: only the forward instances walk tiles (2, default: all)
  int thin_wgrad_bias;  // ADVOC_THIN_WGRAD_BIAS  0: the thin layers' bias gradient stays a pass of its own over dy
  int fused_taps;       // ADVOC_FUSED_TAPS     0: <= 2 output columns over a wide K stay on the two-stage path (pointwise GEMM to the workspace + tap_sum)
  int thin_fwd_spec;    // ADVOC_THIN_FWD_SPEC  0: forward calls of the thin layers run the run-time-generic instance of thin_k_gemm_kernel
  int thin_wgrad_nt;    // ADVOC_THIN_WGRAD_NT  widest column tile (32-channel blocks per wave: 1 | 2 | 4) of thin_wgrad_kernel
  int h3_deep_wgs_per_cu;  // ADVOC_H3_DEEP_WGS_PER_CU  workgroups per CU the workspace K split of the deep layers aims at
  int h3_deep_split_div;   // ADVOC_H3_DEEP_SPLIT_DIV   K tiles per slice, at least
  int h3_rem_ws;        // ADVOC_H3_REM_WS      1: the K slices of the remainder launch meet in the workspace (no zero fill, no atomics)
  int h3_rem_wgs_per_cu;   // ADVOC_H3_REM_WGS_PER_CU  workgroups per CU the K split of that launch aims at
  int h3_deep_split;    // ADVOC_H3_DEEP_SPLIT    > 0: that many K slices for every deep-layer launch (experiments)
  int h3_patch_s1n128;  // ADVOC_H3_PATCH_S1N128  1: the 128-column instance of the 4x4 stride-1 patch kernel (<5,.>)
  int emit_dx;          // ADVOC_EMIT_DX          1: backward-data calls honour advoc_conv_layer.dx_img (0: kill switch); (the lower layer's dy image from the epilogue)
  int h3_deep_plan;     // ADVOC_H3_DEEP_PLAN     1: r4's tile / K-slice choice for the launches under one round of tiles, 0: r3's
  int h3_deep_stages;   // ADVOC_H3_DEEP_STAGES   LDS stages of the 128 x 64 per-tap tile (2 | 3 (default since r6) | 4)
  int h3_rem_split_div; // ADVOC_H3_REM_SPLIT_DIV  K tiles per slice, at least
  int reserve_cus;      // ADVOC_RESERVE_CUS     CUs the PERSISTENT launches (patch kernels, image weight gradient) leave free: set by
                        //                       advoc_amd.parallel from ADVOC_DP_RESERVE_CUS when world_size > 1, so that RCCL's kernels
                        //                       find a CU next to the 110-160 KB-LDS workgroups (multiples of 8: one CU per XCD)
  int h3_patch_2wg;     // ADVOC_H3_PATCH_2WG   (r6) 0 (default: measured, not faster -- profiles/r06_two_workgroups_per_cu.md): four-phase gathers on one 8-wave workgroup per CU; 1: backward-data launches on two
                        //                       4-wave workgroups per CU (patch_gemm_h3_kernel<6, .>); 2: forward launches too
  int h3_patch_2wg_delay;   // ADVOC_H3_PATCH_2WG_DELAY  percent of half a tile's estimated life the CU's second workgroup starts late
  int h3_patch_ablate;  // ADVOC_H3_PATCH_ABLATE  (-DADVOC_DIAG builds only) timing experiments: bits 1 no DMA, 2 no MFMA, 4 no barrier (results are garbage)
};

const Tuning& tuning();

}  // namespace advoc
