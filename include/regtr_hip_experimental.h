/* regtr_hip_experimental.h -- entry points of libregtr_hip.experimental.so that are NOT part of the drop-in boundary (include/regtr_hip.h) and
 * NOT covered by REGTR_ABI_VERSION: kernels that were built, tested and MEASURED SLOWER than the path the product runs (kept so the
 * measurements in docs/NEGATIVES.md can be reproduced), and diagnostics.  They are compiled ONLY under -DREGTR_EXPERIMENTAL, i.e. into
 * libregtr_hip.experimental.so (`python -m regtr_amd.build --experimental`, regtr_amd/experimental.py); the shipped libregtr_hip.so
 * does not contain them and the product has no route to them.  Signatures here may change or disappear without a version bump. */
#ifndef REGTR_HIP_EXPERIMENTAL_H
#define REGTR_HIP_EXPERIMENTAL_H

#include "regtr_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* KPConv.forward (kpconv_blocks.py:269-414) in ONE launch for the level-0 shape -- 32 -> 32 channels, 15 kernel points, rows of at
 * most 40 neighbours (regtr_kpconv_fused_supported): gather, kernel-point correlation, contraction with W [480,32] and the division
 * by the neighbour count, the weighted features staying in LDS (they are 4.6 GB per convolution of a 64-pair forward otherwise).
 * x [ns,32]: FINAL features; s_xyzf [ns,4]: (x, y, z, positivity flag) records (regtr_instnorm_apply's row_xyz form);
 * planes = regtr_gemm_split_weights(W as [480,32], transposed = 1).  out [nq,32]. */
int regtr_kpconv_fused_supported(int Cin, int Cout, int KP, int H);
int regtr_kpconv_fused(const float* q_xyz, int nq, int ns, const int* nbr, int H, const float* x, const float* s_xyzf,
                       const float* kernel_points, int KP, float extent, const void* planes, float* out, void* stream);


/* The same tail when the second summand already exists as an [M, N] array R (identity shortcut: r_stats NULL) or is a shortcut product
 * with its own statistics r_stats [n_clouds, N, 2]:  Y = LeakyReLU_slope( InstanceNorm(A1' W1) + [InstanceNorm](R) ), A1' =
 * LeakyReLU_a1_slope(InstanceNorm(A1)) by a1_stats [n_clouds, K1, 2] (kpconv_blocks.py:727-741).  A1' W1 is never written.
 * Shape served: K1 = 64, N % 64 == 0.  tile_info = regtr_tile_segments(seg_off, n_clouds, M, 256, ..). */
int regtr_block_tail_res_supported(int M, int N, int K1);
size_t regtr_block_tail_res_ws_bytes(int n_clouds, int max_len, int N, int K1);
int regtr_block_tail_res(const float* A1, int lda1, const float* a1_stats, float a1_slope, const float* W1, const float* R, int ldr,
                         const float* r_stats, const int* seg_off, int n_clouds, int max_len, const void* tile_info, int M, int N, int K1,
                         float eps, float slope, float* Y, int ldy, void* ws, size_t ws_bytes, void* stream);


/* diagnostic: resident workgroups per CU of the row-strip split kernel (cw 2|4 column blocks, ar 2|3|4 A-ring mode, stats epilogue) */
int regtr_gemm_x3_strip_occupancy(int cw, int ar, int stats);

#ifdef __cplusplus
}
#endif
#endif /* REGTR_HIP_EXPERIMENTAL_H */
