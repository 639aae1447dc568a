/*
 * regtr_hip.h -- C ABI of libregtr_hip.so: the gfx950 (MI355X) kernels of the RegTR correspondence-inference hot
 * path.  Plain pointers and sizes only, no torch types: any host (ctypes, cgo, JNI, a C++ pipeline) can bind it.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its comment says "host";
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is enqueued, nothing synchronises;
 *   - return value: 0 = OK, <0 = error (REGTR_ERR_*); nothing throws across the boundary.  The Python shim raises
 *     RuntimeError on a non-zero status, as the reference's CPython wrappers do
 *     (cpp_neighbors/wrapper.cpp:77,95,133,203);
 *   - clouds of a batch are STACKED: rows [seg_off[c], seg_off[c+1]) belong to cloud c; seg_off is int32 [n_clouds+1]
 *     ON THE DEVICE so that data-dependent level sizes never have to visit the host between kernels.  `*_cap`
 *     arguments are host-known upper bounds used only to size launches and buffers.
 *   - indices are int32; the shadow / pad index is the total number of support rows (neighbors.cpp:323-324).
 *
 * Reference interfaces replaced (paths relative to /root/reference/src):
 *   regtr_grid_subsample      cpp_subsampling.subsample_batch     models/backbone_kpconv/cpp_wrappers/cpp_subsampling/wrapper.cpp:62-333
 *                             = batch_grid_subsampling            .../grid_subsampling/grid_subsampling.cpp:109-211
 *                             (GPU twin batch_grid_subsampling_kpconv_gpu, models/backbone_kpconv/kpconv.py:213-240)
 *   regtr_cellgrid_build +
 *   regtr_radius_query        cpp_neighbors.batch_query           .../cpp_neighbors/wrapper.cpp:58-238
 *                             = batch_nanoflann_neighbors         .../cpp_neighbors/neighbors/neighbors.cpp:211-332
 *                             (GPU twin batch_neighbors_kpconv_gpu, kpconv.py:261-288)
 *   regtr_kpconv_gather +
 *   regtr_gemm_f32            KPConv.forward                      models/backbone_kpconv/kpconv_blocks.py:269-414
 *   regtr_maxpool_gather      max_pool                            kpconv_blocks.py:127-143
 *   regtr_instnorm_*          BatchNormBlock (InstanceNorm1d) + LeakyReLU + residual   kpconv_blocks.py:497-519,556-561,741
 *   regtr_gemm_f32 / _x3 /
 *   regtr_gemm_stream         nn.Linear call sites                kpconv_blocks.py:557, regtr.py:145,432-436, transformers.py:197-238
 *   regtr_block_tail          ResnetBottleneckBlock tail (unary2 + unary_shortcut + sum + LeakyReLU), SimpleBlock after its gather
 *                                                                 kpconv_blocks.py:727-741, 590-646
 *   regtr_encoder_fwd         KPFEncoder.forward (blocks sequenced) models/backbone_kpconv/kpconv.py:81-88, kpconv_blocks.py:632-646,706-741
 *   regtr_layernorm           nn.LayerNorm (+ with_pos_embed)     transformers.py:116-119,194-195,213-215,232
 *   regtr_posemb_sine         PositionEmbeddingCoordsSine.forward models/transformer/position_embedding.py:29-50
 *   regtr_mha_fwd             nn.MultiheadAttention core          transformers.py:197-226
 *   regtr_attn_xyz            CorrespondenceDecoder.simple_attention   models/regtr.py:316-351
 *   regtr_weighted_procrustes pose assembly + compute_rigid_transform   regtr.py:185-203, utils/se3_torch.py:108-154
 */
#ifndef REGTR_HIP_H
#define REGTR_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define REGTR_OK 0
#define REGTR_ERR_LAUNCH (-1)
#define REGTR_ERR_ARG (-2)
#define REGTR_ERR_WORKSPACE (-3)

/* ABI version of THIS header.  Entry points have gained arguments between versions (regtr_maxpool_gather, regtr_radius_query,
 * regtr_kpconv_gather, regtr_instnorm_apply, regtr_mha_fwd, regtr_gemm_x3): a binding generated from another version of the header
 * would pass shifted arguments, so every binding must compare regtr_abi_version() with the REGTR_ABI_VERSION it was written against
 * before its first call (regtr_amd/_lib.py does).  Bumped on any signature change. */
#define REGTR_ABI_VERSION 11
int regtr_abi_version(void);

/* The STATUS WORD: an optional device int (zeroed by the caller, e.g. once per forward) that kernels OR bits into -- conditions that
 * only the data can reveal, reported without a host round trip per launch.  `status` arguments may be NULL.
 *   REGTR_STATUS_F16_RANGE       an f16 pair product (regtr_gemm_x3 with n_planes = 4, regtr_mha_fwd precision 3) came out non-finite:
 *                                an operand reached f16's range (|x| >= 65504).  The caller re-runs in the bf16x3 format (n_planes 3 /
 *                                precision 0), which has float32's range; regtr_amd.RegTR.forward does, once per forward.
 *   REGTR_STATUS_NONFINITE_POSE  regtr_weighted_procrustes wrote a non-finite R|t (whatever the cause upstream). */
#define REGTR_STATUS_F16_RANGE 1
#define REGTR_STATUS_NONFINITE_POSE 2

/* ---- preprocessing ---------------------------------------------------------------------------------------- */

size_t regtr_grid_subsample_ws_bytes(int n_cap, int n_clouds);

/* Voxel-grid barycentres of every cloud.  xyz [n_cap,3] (live rows: seg_off[n_clouds]); out_xyz [n_cap,3] receives
 * M <= n rows, clouds stacked in order, voxels of a cloud in order of first appearance in the input;
 * out_seg_off [n_clouds+1].  Barycentres are bit-identical to the reference's (float32 sums in input order). */
int regtr_grid_subsample(const float* xyz, const int* seg_off, int n_clouds, int n_cap, float dl, float* out_xyz,
                         int* out_seg_off, void* ws, size_t ws_bytes, void* stream);

/* The same with a choice of output row order: row_order 0 = first appearance (above); 1 = the reference's own order, i.e.
 * the iteration order of the libstdc++ std::unordered_map<size_t, .> it fills in input order (grid_subsampling.cpp:48,58-59,85)
 * -- the parity mode (cfg.kpconv_ref_row_order); one thread per cloud replays the container (csrc/ref_umap.h). */
size_t regtr_grid_subsample_ordered_ws_bytes(int n_cap, int n_clouds, int row_order);
/* key_mode: which of the reference's two voxel rules.
 *   0  floor((p - origin) / dl), origin = floor(min corner * (1 / dl)) * dl, linear size_t key -- the CPU Preprocessor's
 *      cpp_subsampling (grid_subsampling.cpp:25-31,53-56); bit-exact against the unmodified reference C++;
 *   1  floor(p / dl), no origin shift, float32 IEEE division -- PreprocessorGPU, the class the reference model instantiates
 *      (regtr.py:29; kpconv.py:213-240: MinkowskiEngine quantisation of points / sampleDl);
 *   2  floor(p * (1 / dl)), reciprocal rounded to float32 -- the same rule as torch's CUDA division by a host scalar evaluates it.
 * 0 and 1 give DIFFERENT voxel sets wherever points sit on voxel faces (3DMatch fragments lie on a lattice: red-kitchen pair
 * 9 977 vs 10 088 level-1 points).  Barycentre arithmetic and the first-appearance row order are the same for every mode
 * (MinkowskiEngine's own output order and summation order are unspecified).  row_order 1 requires key_mode 0.
 * out_cap: rows of out_xyz (<= 0 or > n_cap: n_cap).  The output size is data dependent and only known on the device; a caller that
 * allocates less than n_cap rows gets the first out_cap voxels (first-appearance order), out_seg_off SATURATED at out_cap, and detects
 * the full level by out_seg_off[n_clouds] == out_cap (then repeats with a larger capacity).  row_order 1 needs out_cap = n_cap. */
int regtr_grid_subsample_ordered(const float* xyz, const int* seg_off, int n_clouds, int n_cap, float dl, int row_order, int key_mode,
                                 int out_cap, float* out_xyz, int* out_seg_off, void* ws, size_t ws_bytes, void* stream);

size_t regtr_cellgrid_ws_bytes(int ns_cap, int n_clouds);

/* Builds the support-point cell grid for `radius` into ws (kept by the caller, reused by any number of queries). */
int regtr_cellgrid_build(const float* s_xyz, const int* s_seg_off, int n_clouds, int ns_cap, float radius, void* ws,
                         size_t ws_bytes, void* stream);

/* Fixed-radius neighbours within the same cloud, strict d2 < r2 in the reference's float32 arithmetic, rows padded with
 * Ns_total = s_seg_off[n_clouds].  out_idx [nq_cap,K]:
 *   order 0  the K NEAREST supports in the ball, ascending (d2, support index) -- the reference's CPU Preprocessor
 *            (nanoflann radius search + sort, kpconv.py:243-258; cpp_neighbors/neighbors.cpp:211-332);
 *   order 1  the FIRST K supports in the ball by support index, ascending by index -- the reference's PreprocessorGPU
 *            (pytorch3d ball_query, kpconv.py:261-288), the class its model instantiates (regtr.py:29).
 * The two differ only on rows whose ball holds more than K supports.  out_count [nq_cap] (optional):
 * untruncated in-ball count; out_max_count (optional, device int zeroed by the caller): max over out_count, i.e. the
 * row width the reference's batch_query would return.  1 <= K <= 448.  ns_cap / ws_bytes as given to the build. */
int regtr_radius_query(const float* q_xyz, const int* q_seg_off, int nq_cap, const int* s_seg_off, int ns_cap,
                       int n_clouds, float radius, int K, int order, const void* grid_ws, size_t ws_bytes, int* out_idx,
                       int* out_count, int* out_max_count, void* stream);

/* The same table for the grid's OWN supports as queries (every conv table of the pyramid): a cell-centric kernel -- one wave
 * per occupied cell stages the 27 neighbouring runs once in LDS and answers all of the cell's queries from there.  out_idx
 * [ns_cap, K] is indexed by the original support row; results are identical to regtr_radius_query(s_xyz, s_seg_off, ...). */
int regtr_radius_query_self(const int* s_seg_off, int ns_cap, int n_clouds, float radius, int K, int order, const void* grid_ws,
                            size_t ws_bytes, int* out_idx, int* out_count, int* out_max_count, void* stream);

/* ---- ground-truth overlap (training / validation side) ------------------------------------------- */

/* utils/pointcloud.py:8-65 compute_overlap: index of the NEAREST support of the query's cloud with d2 < radius^2, distances
 * in float64 like open3d's KDTreeFlann (float32 coordinates widened), -1 when the ball is empty.  Query cloud c searches
 * support cloud c (stack src clouds as supports and tgt clouds as queries, then the other way round).  grid_ws: a cell grid
 * built by regtr_cellgrid_build over the supports with grid_radius * (1 + 1e-6) >= radius.  out_idx [nq_cap]. */
int regtr_nearest_in_radius(const float* q_xyz, const int* q_seg_off, int nq_cap, const int* s_seg_off, int ns_cap,
                            int n_clouds, double radius, float grid_radius, const void* grid_ws, size_t ws_bytes, int* out_idx,
                            void* stream);

/* models/backbone_kpconv/kpconv.py:553-562 compute_overlaps, one pyramid level: out[q] = clamp(mean of ov over the valid
 * (< ns) entries of the first H columns of row q of nbr, 0, 1); a row without a valid entry gives NaN as in the reference. */
int regtr_overlap_avgpool(const float* ov, int ns, const int* nbr, int ld_nbr, int nq, int H, float* out, void* stream);

/* (Parity mode's neighbour tables in the reference's KD-tree / std::sort row order: include/regtr_hip_parity.h, libregtr_parity.so --
 * since ABI 11 a library of its own, so that the product library holds no nanoflann-derived code.) */

/* ---- KPConv encoder --------------------------------------------------------------------------------------- */

/* flag[j] = (sum_c x'[j,c] > 0) ? 1 : 0   -- the per-support term of the reference's normaliser (kpconv_blocks.py:409-410).
 * x' = x, or LeakyReLU_slope(InstanceNorm(x)) when stats [n_seg,C,2] + seg_off [n_seg+1] are given (fused UnaryBlock tail). */
int regtr_rowsum_positive(const float* x, int n, int C, const float* stats, const int* seg_off, int n_seg, float slope,
                          float* flag, void* stream);

/* 1 when regtr_kpconv_gather derives the positivity flags from the feature rows it gathers anyway (Cin == 1 or a
 * multiple of 32 with H <= 64, 16-byte aligned x / wf / x_stats): `flag` may then be NULL and regtr_rowsum_positive skipped. */
int regtr_kpconv_gather_computes_flag(int Cin, int H);

/* wf [nq, KP*Cin] (kernel point major, channel minor), num [nq] = max(1, #positive neighbours).  nbr [nq,H] int32,
 * x [ns,Cin], flag [ns] (or NULL, see above), kernel_points [KP,3], KP <= 16.  x_stats [n_seg,Cin,2] + q_seg_off [n_seg+1] (optional): the
 * gathered features are LeakyReLU_slope(InstanceNorm(x)) computed on the fly (cloud of a neighbour = cloud of its query).
 * s_xyzf (optional, 16-byte aligned [ns,4], no x_stats): per-support records (x, y, z, f) read with ONE 16-byte load per neighbour
 * instead of four scattered 4-byte ones -- the gathers are bound by the texture-address path (TA ~76 % busy, 40 % of its lines were
 * these dwords).  f = the positivity flag (regtr_instnorm_apply writes the records: x is then final, no fold, no row sums); for
 * Cin == 1, f = the feature itself.  ld_wf: row stride of wf in floats, 0 = KP*Cin; only Cin == 1 takes another value (16: rows padded
 * with zeros, the A operand of regtr_block_tail's first-block form). */
int regtr_kpconv_gather(const float* q_xyz, int nq, const float* s_xyz, int ns, const int* nbr, int H, const float* x,
                        int Cin, const float* flag, const float* s_xyzf, const float* kernel_points, int KP, float extent,
                        const float* x_stats, const int* q_seg_off, int n_seg, float slope, float* wf, int ld_wf, float* num,
                        void* stream);

/* out[q,:] = max over the first H columns of row q of nbr (row stride ld_nbr >= H) of x[nbr[q,h],:], the shadow index
 * ns standing for a zero row (kpconv_blocks.py:127-143).  H < ld_nbr serves the reference's CPU tables, whose width is
 * min(max in-ball count, neighborhood_limit) (kpconv.py:255-258): a full row then holds no shadow and its maximum may be negative. */
int regtr_maxpool_gather(const float* x, int ns, int C, const int* nbr, int ld_nbr, int nq, int H, float* out, void* stream);

size_t regtr_instnorm_ws_bytes(int n_clouds, int max_len, int C);
int regtr_instnorm_stats(const float* x, const int* seg_off, int n_clouds, int max_len, int C, float eps, float* stats,
                         void* ws, size_t ws_bytes, void* stream);
/* y = act((x-mean)*rstd [+ residual | + (residual-rmean)*rrstd]); act: 0 none, 1 LeakyReLU(slope); y may alias x.
 * row_positive (optional, C <= 256): [rows] 1.0 where sum_c y[row,c] > 0 -- KPConv's per-support normaliser flag
 * (kpconv_blocks.py:409-410), ready for regtr_kpconv_gather's `flag`; with row_xyz [rows,3] it is written as 16-byte records
 * [rows,4] = (x, y, z, flag) instead: regtr_kpconv_gather's `s_xyzf`. */
int regtr_instnorm_apply(const float* x, const int* seg_off, int n_clouds, int max_len, int C, const float* stats,
                         const float* residual, const float* res_stats, int act, float slope, float* y, const float* row_xyz,
                         float* row_positive, void* stream);

/* ---- dense ------------------------------------------------------------------------------------------------- */

/* C[M,N] = act(A'[M,K] B[K,N] / row_div[m] + bias[n]) + residual[m,n] ; float32 MFMA ; act: 0 none, 1 ReLU.
 * A' = A, or LeakyReLU_slope(InstanceNorm(A)) when a_stats [n_seg,K,2] + a_seg_off [n_seg+1] are given.
 * ws: regtr_gemm_f32_ws_bytes(M,N,K) bytes of scratch for the split-K path (0 for most shapes; ws may then be NULL). */
size_t regtr_gemm_f32_ws_bytes(int M, int N, int K);
int regtr_gemm_f32(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                   const float* bias, const float* row_div, const float* residual, int ldr, int act,
                   const float* a_stats, const int* a_seg_off, int n_seg, float a_slope, void* ws, size_t ws_bytes,
                   void* stream);

/* The same contraction at float32 accuracy on the bf16 matrix cores (16x the f32-MFMA rate on gfx950): every float32 is
 * split exactly into three bf16 (x = x0 + x1 + x2) and the product is evaluated as six bf16 MFMAs with f32 accumulation;
 * the dropped cross terms are below one f32 ulp of each product.  Weights are split once:
 *   regtr_gemm_split_weights(W, ld, N, K, transposed, planes)   W = [N,K] (nn.Linear.weight as stored; transposed = 0) or
 *                                                               [K,N] (transposed = 1); planes: .._bytes(N, K) bytes
 * regtr_gemm_x3 then has the contract of regtr_gemm_f32 with `planes` in place of B.  Shapes it does not take
 * (regtr_gemm_x3_supported == 0: N not a multiple of 64, K not a multiple of 4) go to regtr_gemm_f32. */
int regtr_gemm_x3_supported(int M, int N, int K);
int regtr_gemm_x3_preferred(int M, int N, int K);   /* supported AND measured faster than regtr_gemm_f32 (K >= 32) */
size_t regtr_gemm_split_weights_bytes(int N, int K);
int regtr_gemm_split_weights(const float* W, int ld, int N, int K, int transposed, void* planes, void* stream);
/* The f16 pair operand format (n_planes = 4 of regtr_gemm_x3): x = h0 + h1 / 2048, h0 = f16(x), h1 = f16((x - h0) * 2048) -- 22 mantissa
 * bits in two planes; a product is three v_mfma_f32_32x32x16_f16 (the two low terms in a second, scaled accumulator) at float32-grade
 * accuracy (error vs float64 within 3x of the six-term bf16 split's on RegTR's shapes), half the matrix-pipe work of the bf16 split.
 * Operands must stay below 65504 in magnitude.  regtr_gemm_x3_f16_supported(M, N, K, with_stats): every shape regtr_gemm_x3 supports. */
int regtr_gemm_x3_f16_supported(int M, int N, int K, int with_stats);      /* with_stats: the call passes stat_partial, a_stats or tile_info
                                                                             * (the launch then keeps regtr_gemm_x3_tile_rows' tile height) */
size_t regtr_gemm_split_weights_f16_bytes(int N, int K);
int regtr_gemm_split_weights_f16(const float* W, int ld, int N, int K, int transposed, void* planes, void* stream);
size_t regtr_gemm_x3_ws_bytes(int M, int N, int K);
int regtr_gemm_x3(const float* A, int lda, const void* planes, float* C, int ldc, int M, int N, int K,
                  const float* bias, const float* row_div, const float* residual, int ldr, int act,
                  const float* a_stats, const int* a_seg_off, int n_seg, float a_slope, void* ws, size_t ws_bytes,
                  double* stat_partial, const int* stat_seg_off, int n_stat_seg, int n_planes, const void* tile_info, int* status,
                  void* stream);
/* status (optional): the status word above; n_planes = 4 reports REGTR_STATUS_F16_RANGE.
 * tile_info (optional): regtr_tile_segments(seg_off, n_seg, M, regtr_gemm_x3_tile_rows(M,N,K), ..) -- 16 bytes per row tile that
 * replace the per-workgroup cloud search (a chain of dependent memory round trips) when a_stats / stat_partial are used; a_seg_off
 * and stat_seg_off must then be the same array. */
int regtr_tile_segments(const int* seg_off, int n_seg, int M, int rows, void* out, void* stream);
int regtr_gemm_x3_tile_rows(int M, int N, int K);
/* n_planes: 3 = the float32-grade six-term product (default everywhere); 2 = three leading terms (a0 w0 + a0 w1 + a1 w0,
 * ~2^-16 relative per product); 1 = plain bf16 operands with float32 accumulation (cfg.compute_dtype 'bf16').  1 and 2 do not
 * combine with a_stats / stat_partial (the KPConv encoder always runs float32-grade). */
/* One-shot strip variant for the shallow encoder levels (K in {32, 64, 128}, N <= 512, millions of rows; csrc/gemm_stream.hip):
 * every wave takes 32 rows from global memory straight into MFMA fragments, the weight planes sit in LDS per 256-row workgroup,
 * nothing is loaded after a store.  Same float32-grade product as regtr_gemm_x3:  C = A' W.
 *   a_stats [n_seg,K,2] (K <= 64): A' = LeakyReLU_a_slope(InstanceNorm(A)); seg_off [n_seg+1]: cloud offsets of the rows;
 *   tile_info = regtr_tile_segments(seg_off, n_seg, M, regtr_gemm_stream_tile_rows(), ..);
 *   stat_partial (optional) = (ceil(M / 256) + n_seg) * N double2 of per-(tile, cloud) column sums of C for
 *   regtr_instnorm_finalize_tiles(tile_rows = 256). */
int regtr_gemm_stream_supported(int M, int N, int K);
int regtr_gemm_stream_tile_rows(void);
int regtr_gemm_stream(const float* A, int lda, const void* planes, float* C, int ldc, int M, int N, int K,
                      const float* a_stats, float a_slope, const int* seg_off, int n_seg, const void* tile_info,
                      double* stat_partial, void* stream);
/* Linear -> InstanceNorm [+ Linear -> InstanceNorm of a second input] -> LeakyReLU in one pass over the NARROW inputs
 * (csrc/block_tail.hip):  Y = LeakyReLU_slope( InstanceNorm(A1' W1) [+ InstanceNorm(A2 W2)] ).  The per-cloud statistics of a product
 * come from the K x K second moments of its input (float64), so no product is ever written.  Two served forms
 * (regtr_block_tail_supported):
 *   K1 = 32, K2 = 64, N = 128: the tail of the level-0 resnet block (kpconv_blocks.py:727-741) -- replaces unary2 GEMM + shortcut
 *     GEMM + regtr_instnorm_apply; A1' = LeakyReLU_a1_slope(InstanceNorm(A1)) by a1_stats [n_clouds,K1,2] (the conv output's);
 *   K1 = 16, K2 = 0, N = 64: the first block (:590-646), A1 = the Cin = 1 gather's WF rows at ld_wf = 16, A1' = A1 / row_div1[row]
 *     (the neighbour count, :411), a1_stats / A2 / W2 NULL -- replaces contraction GEMM + statistics + regtr_instnorm_apply.
 * W1 [K1,N] / W2 [K2,N] float32 row-major, seg_off [n_clouds+1], max_len = longest cloud, tile_info = regtr_tile_segments(seg_off,
 * n_clouds, M, 256, ..).  out_stats (optional) [1 or 2,n_clouds,N,2] receives the (mean, rstd) of the products. */
int regtr_block_tail_supported(int M, int N, int K1, int K2);
size_t regtr_block_tail_ws_bytes(int n_clouds, int max_len, int N, int K1, int K2);
int regtr_block_tail(const float* A1, int lda1, const float* a1_stats, float a1_slope, const float* row_div1, const float* A2, int lda2,
                     const float* W1, const float* W2, const int* seg_off, int n_clouds, int max_len, const void* tile_info,
                     int M, int N, int K1, int K2, float eps, float slope, float* Y, int ldy, void* ws, size_t ws_bytes,
                     float* out_stats, void* stream);

/* InstanceNorm statistics of C straight from the GEMM epilogue (no second pass over C): when
 * R = regtr_gemm_x3_stat_tile_rows(M,N,K) > 0, pass stat_partial = (ceil(M/R) + n_stat_seg) * N * 2 doubles and the cloud
 * offsets of C's rows; then regtr_instnorm_finalize_tiles(stat_partial, seg_off, n_clouds, N, R, eps, stats) yields the
 * same [n_clouds, N, 2] (mean, rstd) table as regtr_instnorm_stats(C).  R = 0 (split-K shapes): not available. */
int regtr_gemm_x3_stat_tile_rows(int M, int N, int K);
int regtr_instnorm_finalize_tiles(const double* partial, const int* seg_off, int n_clouds, int C, int tile_rows, float eps,
                                  float* stats, void* stream);

/* out = a + b over n floats (16-byte aligned pointers): with_pos_embed of the post-norm layer, transformers.py:118-119 */
int regtr_add_f32(const float* a, const float* b, size_t n, float* out, void* stream);

int regtr_layernorm(const float* x, int n, int D, const float* gamma, const float* beta, float eps, const float* add,
                    float* y, float* y_plain, void* stream);

int regtr_posemb_sine(const float* xyz, int n, int npf, int d_model, float scale, const float* dim_t, float* pe,
                      void* stream);

/* ---- attention + pose -------------------------------------------------------------------------------------- */

/* softmax(q k^T * scale) v per head on packed clouds: cloud c's rows attend the rows of cloud kv_of[c]; head_dim = 32.
 * precision: 0 = float32-grade on the bf16 matrix cores (every operand split exactly into three bf16, six MFMAs per product),
 * 1 = plain bf16 operands with float32 softmax / accumulation (cfg.compute_dtype 'bf16'), 2 = exact-f32 MFMA, 3 = float32-grade by the
 * f16 pair split (x = h0 + h1 / 2048: two planes, three MFMAs per product, a scaled second accumulator; operands below 65504 --
 * q, k, v are projections of LayerNorm outputs, the probabilities are <= 1; what cfg.compute_dtype 'fp32' uses). */
int regtr_mha_fwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out, int ldo,
                  const int* seg_off, const int* kv_of, int n_clouds, int max_len, int n_heads, int head_dim, float scale,
                  int precision, int* status, void* stream);      /* status: optional status word (precision 3 reports REGTR_STATUS_F16_RANGE) */

/* The whole pre-norm cross-encoder stack (transformers.py:183-244 forward_pre for every layer, :37-59 the final norm of every
 * layer's output) enqueued by one call -- the same launches, in the same order, as calling regtr_layernorm / regtr_gemm_x3 /
 * regtr_mha_fwd layer by layer (12 per layer); it exists because at one pair per forward the host, not the GPU, is the bound.
 *   x [n_tok, d_model] packed tokens (left untouched)  ->  outs [n_layers | 1, n_tok, d_model]
 *   layer_params  HOST array of n_layers * regtr_cross_encoder_per_layer_params() DEVICE pointers, per layer:
 *                 norm1 gamma, beta | self_attn in_proj planes, bias | out_proj planes, bias | norm2 gamma, beta |
 *                 multihead_attn in_proj planes, bias | out_proj planes, bias | norm3 gamma, beta | linear1 planes, bias |
 *                 linear2 planes, bias          (planes = regtr_gemm_split_weights of the nn.Linear weight)
 *   layer_eps     HOST array, 3 floats per layer;  final_gamma NULL = no final LayerNorm (plain copy)
 *   pe            [n_tok, d_model] added to the normalised tokens for q, k and v (sa/ca_val_has_pos_emb = true), or NULL
 *   supported():  head_dim 32 and every Linear on the split kernel; otherwise issue the launches one by one.
 *   ws            regtr_cross_encoder_ws_bytes(n_tok, d_model, d_ff) bytes */
int regtr_cross_encoder_per_layer_params(void);
int regtr_cross_encoder_supported(int n_tok, int d_model, int d_ff, int n_heads);
size_t regtr_cross_encoder_ws_bytes(int n_tok, int d_model, int d_ff);
int regtr_cross_encoder_fwd(const float* x, int n_tok, int d_model, int d_ff, int n_heads, int n_layers,
                            const void* const* layer_params, const float* layer_eps, const float* final_gamma,
                            const float* final_beta, float final_eps, int return_intermediate, const float* pe,
                            const int* seg_off, const int* kv_self, const int* kv_cross, int n_clouds, int max_len,
                            int gemm_planes, int attn_precision, void* ws, size_t ws_bytes, float* outs, int* status, void* stream);

/* The KPConv encoder's blocks (kpconv.py:81-88 KPFEncoder.forward over kpconv_blocks.py:632-646 SimpleBlock.forward and :706-741
 * ResnetBottleneckBlock.forward) enqueued by ONE call, for the SMALL-batch regime: a pair or two per forward -- the reference's own
 * operating mode (conf/3dmatch.yaml:11 test_batch_size 1, trainer.py:202-206), where the host, not the GPU, bounds an op-by-op forward.
 * The same launches, with the same arguments, in the same order as regtr_amd/kpconv.py issues them through the entry points above
 * (outputs bit-identical); intermediates carved from one workspace.  HOST-side descriptors: */
typedef struct {
    const float* kn;        /* [K, N] float32 row-major (regtr_gemm_f32's B) */
    const void* planes;     /* regtr_gemm_split_weights of the matrix, or NULL (shape not served by regtr_gemm_x3) */
    const void* planes16;   /* regtr_gemm_split_weights_f16, or NULL (a weight beyond the f16 range: stays on `planes`) */
    int N, K;               /* N == 0: no such Linear in the block (identity) */
} regtr_weight_t;
typedef struct {
    int kind;               /* 0 SimpleBlock, 1 ResnetBottleneckBlock */
    int strided;            /* 1: the convolution reads level `layer` and writes level `layer + 1` (pool table, max-pooled shortcut) */
    int layer;
    int n_kp;               /* kernel points (<= 16) */
    float extent;           /* KP_extent of the block's KPConv */
    const float* kernel_points;     /* [n_kp, 3] */
    regtr_weight_t unary1;  /* kind 1: Linear in front of the KPConv (N == 0: nn.Identity) */
    regtr_weight_t conv;    /* KPConv.weights viewed as [n_kp * Cin, Cout] */
    regtr_weight_t unary2;  /* kind 1 */
    regtr_weight_t shortcut;        /* kind 1: unary_shortcut (N == 0: nn.Identity) */
} regtr_encoder_block_t;
typedef struct {
    const float* points;    /* [n, 3] */
    int n;                  /* live rows of the level (host-known: the one read-back of a forward) */
    const int* conv_idx;    /* [n, K] neighbour table of the level, or NULL */
    const int* pool_idx;    /* [n of the NEXT level, K] supports of this level per next-level point, or NULL */
    int K;                  /* columns of both tables */
    int pool_width;         /* columns of pool_idx the strided shortcut's max-pool reads (K; the parity mode's narrower CPU tables) */
    const int* seg_off;     /* [n_clouds + 1] */
    int max_len;            /* longest cloud of the level */
} regtr_encoder_level_t;
/* supported(): the shapes; the CALLER keeps the call to its small-batch regime (regtr_amd/ops.py SMALL_REGIME_ROWS = 131072 level-0 rows:
 * below it no large-batch kernel form applies -- which is the routing this function implements); otherwise issue the launches one by one.  x_in [rows of block `first`'s level, its input width] -> out [rows of block `last - 1`'s level, its width];
 * blocks [first, last).  f16_pair: contractions in the f16 pair format where planes16 exists and the kernel serves the shape (what
 * cfg.compute_dtype 'fp32' runs); ws: regtr_encoder_ws_bytes(...) bytes for the same arguments. */
int regtr_encoder_supported(const regtr_encoder_block_t* blocks, int n_blocks, const regtr_encoder_level_t* levels, int n_levels,
                            int n_clouds, int first, int last);
size_t regtr_encoder_ws_bytes(const regtr_encoder_block_t* blocks, int n_blocks, const regtr_encoder_level_t* levels, int n_levels,
                              int n_clouds, int first, int last, int f16_pair);
int regtr_encoder_fwd(const regtr_encoder_block_t* blocks, int n_blocks, const regtr_encoder_level_t* levels, int n_levels, int n_clouds,
                      int first, int last, const float* x_in, float* out, int f16_pair, float slope, float eps, void* ws, size_t ws_bytes,
                      int* status, void* stream);

/* CorrespondenceDecoder.simple_attention (regtr.py:316-351, `direct_regress_coor: False`): single-head attention whose values
 * are coordinates.  q, k [n_layers, n_total, head_dim] contiguous (projections of the conditioned features), xyz [n_total,3],
 * out [n_layers, n_total, 3]; cloud c attends cloud kv_of[c]; head_dim in {32, 64, 128, 256}. */
int regtr_attn_xyz(const float* q, const float* k, const float* xyz, float* out, const int* seg_off, const int* kv_of,
                   int n_clouds, int n_total, int n_layers, int max_len, int head_dim, float scale, void* stream);

/* kp [n_total,3], corr [L,n_total,3], logit [L,n_total], seg_off [2*n_pairs+1] -> pose [L,n_pairs,3,4]; status (optional): the
 * status word, REGTR_STATUS_NONFINITE_POSE when an R|t came out non-finite */
int regtr_weighted_procrustes(const float* kp, const float* corr, const float* logit, const int* seg_off, int n_pairs,
                              int n_total, int n_layers, float* pose, int* status, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* REGTR_HIP_H */
