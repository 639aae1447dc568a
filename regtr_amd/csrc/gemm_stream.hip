// One-shot strip GEMM for the SHALLOW encoder levels: C[M, N] = epilogue(A'[M, K] W[K, N]) with K in {32, 64, 128} input
// columns, N <= 512 outputs and millions of rows -- the UnaryBlock / shortcut Linears of the KPConv levels with >= 131072 rows
// (/root/reference/src/models/backbone_kpconv/kpconv_blocks.py:556-561, 722-741).  Float32-grade on the bf16 matrix cores by the
// same exact three-way split as gemm_x3.hip (six v_mfma_f32_32x32x16_bf16 per product block, smallest terms first).
//
// These launches are HBM streams (a few flops per byte).  The tiled kernel of gemm_x3.hip stages A through LDS behind two
// barriers per k-tile: its eight waves move in lockstep phases, so a CU holds two independent streams, and the short-K launches
// run at 2.5-2.8 TB/s; a read+write probe with the same access patterns and occupancy reaches 5.3 TB/s (tools/probe).  Here
//   * every WAVE owns one strip of 32 rows and goes load -> split -> MFMA -> store -> exit on its own: activations travel from
//     global memory straight into MFMA fragments (the A operand of the 32x32x16 MFMA is 8 consecutive k of one row per lane =
//     two float4 loads), the InstanceNorm+LeakyReLU fold of the producer (a_stats) and the bf16 split happen in registers;
//   * nothing is loaded after a store (on CDNA one counter tracks loads AND stores in issue order: a persistent variant that
//     prefetched its next rows waited on its own store acknowledgements and ran at 2.1 TB/s);
//   * the weight planes (K NB 6 bytes <= 48 KB) are copied into LDS once per 8-wave workgroup (256 rows) while the A loads fly:
//     the only barrier;
//   * the clouds of a 256-row tile come from the per-level table of regtr_tile_segments (one 16-byte load, no search);
//   * InstanceNorm statistics of the result: per-lane float64 sums over the wave's rows, cross-wave reduction in LDS, one partial
//     per (256-row tile, cloud) -- the slot convention of regtr_instnorm_finalize_tiles with tile_rows = 256.
// Measured on MI355X, 2.4 M rows (tools/stream_bench.py), strip vs tiled kernel, both writing C and the statistics partials:
//   64 -> 128: 522 vs 660 us   32 -> 128 (folded A): 395 vs 580   64 -> 32: 184 vs 338 (5.1 TB/s)   128 -> 32: 283 vs 499 (5.5 TB/s)
//   0.6 M rows: 64 -> 256: 284 vs 381   128 -> 64: 151 vs 150.
// (Also built and dropped: the resnet-block tail as an epilogue -- statistics-only pass, then normalise + shortcut + LeakyReLU from
//  the accumulators.  Both passes pay the split + MFMA work again (~350 us at level 0) and the residual tile next to four live
//  accumulators spills: 0.33 + 1.75 ms against 0.41 + 0.71 ms for this kernel followed by k_instnorm_apply.)
#include "common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int SG_WAVES = 8;
constexpr int SG_ROWS = 32 * SG_WAVES;       // rows per workgroup = statistics tile

struct SgArgs {
    const float* A; const uint16_t* Wt; float* C;
    const float2* a_stats;
    const int* seg_off;          // cloud offsets of the rows (of A and C alike), n_seg + 1
    const int4* tile_info;       // regtr_tile_segments(seg_off, n_seg, M, 256): (first cloud, last cloud, first's begin, first's end)
    double2* stat_partial;       // optional [(ceil(M / 256) + n_seg) * N]
    size_t plane;                // elements per weight plane (Npad * Kp)
    int M, N, Kp, lda, ldc, n_seg;
    float a_slope;
};

__device__ __forceinline__ unsigned sg_pack(float a, float b)
{
    bf16x2 v;
    v.x = (__bf16)a; v.y = (__bf16)b;
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void sg_split2(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2)
{
    p0 = sg_pack(a, b);
    const float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);
    p1 = sg_pack(ra, rb);
    p2 = sg_pack(ra - __uint_as_float(p1 << 16), rb - __uint_as_float(p1 & 0xffff0000u));
}

// LDS image of the weights: plane p, column n = one row of K bf16 (CPR = K / 8 sixteen-byte chunks); chunk c of row n sits at
// c ^ swz(n), swz chosen so that the 16 lanes a ds_read_b128 services together (rows {0-3,12-15,20-27} / {4-11,16-19,28-31})
// touch all 64 banks once: CPR 4 -> (n >> 2) & 3, CPR 8 -> (n >> 1) & 7, CPR 16 -> n & 15.
template <int KT> __device__ __forceinline__ unsigned sg_swz(unsigned n)
{
    constexpr int CPR = 2 * KT;
    return CPR == 4 ? ((n >> 2) & 3u) : (CPR == 8 ? ((n >> 1) & 7u) : (n & 15u));
}

// KT = K / 16 (2, 4, 8); NT = 32-column blocks per workgroup column (1, 2, 4); INFOLD: A' = LeakyReLU(InstanceNorm(A))
template <int KT, int NT, bool INFOLD>
__global__ void __launch_bounds__(SG_WAVES* RG_WAVE, (KT <= 4 || NT <= 2) ? 4 : 2) k_gemm_strip(SgArgs g)
{
    constexpr int K = 16 * KT, CPR = 2 * KT, ROWB = K * 2, NB = 32 * NT;
    extern __shared__ __align__(16) unsigned char Ws[];          // [3][NB][K] bf16, chunk-swizzled | double2 red[SG_WAVES][NB]
    const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // Workgroup -> (row tile, column block), round 6: a 1-D launch in which the column blocks of ONE 256-row tile are consecutive workgroups of
    // ONE XCD (workgroup id -> XCD id % 8): the tile's A rows are fetched from HBM once and served to the other column blocks by that XCD's L2.
    // (Until round 5 the launch was (row tiles, column blocks) with the row tiles fastest: every column block re-read A from HBM a whole pass
    //  later, which kept wide outputs off this kernel -- N <= 256 and at most two column blocks.)
    const int ncol = g.N / NB;
    const int wg_k = (int)(blockIdx.x >> 3), row_tile = (wg_k / ncol) * 8 + (int)(blockIdx.x & 7u);
    if (row_tile * SG_ROWS >= g.M) return;                        // (the whole workgroup, before any barrier)
    const int N = g.N, n0 = (wg_k % ncol) * NB;
    const int m0 = row_tile * SG_ROWS, r0 = m0 + 32 * wave;       // the wave's strip
    const int row = r0 + l31;
    const bool row_ok = row < g.M;
#ifdef SG_PROF
    long long pt[8]; int pi = 0;
#define SG_STAMP() do { pt[pi++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SG_STAMP() do {} while (0)
#endif
    SG_STAMP();

    // ---- the strip's rows first (the long-latency loads), the tile's clouds, then the weights while those fly
    float4 raw[KT][2];
    {
        const float* ap = g.A + (size_t)(row_ok ? row : g.M - 1) * g.lda + 8 * hi;
#pragma unroll
        for (int ks = 0; ks < KT; ks++) {
            raw[ks][0] = *(const float4*)(ap + 16 * ks);
            raw[ks][1] = *(const float4*)(ap + 16 * ks + 4);
        }
    }
    const int4 ti = g.tile_info[row_tile];
    {
        // the weight planes by LDS-DMA (global_load_lds_dwordx4: 64 sixteen-byte pieces per instruction, no staging registers, every
        // instruction of the workgroup in flight at once behind the A loads).  A DMA fills LDS lane-linearly, so lane i of DMA q is slot
        // s = 64 q + i = (p NB + n) CPR + cs of the image and fetches logical chunk cs ^ swz(n) of column n, plane p.
        // (As a load -> ds_write loop the copy was six dependent round trips per thread -- hipcc keeps the loop rolled, one
        //  s_waitcnt vmcnt(0) per piece: 11-40 k of a wave's ~50 k cycles, profiles/r04_strip_phase_clocks.md.)
        constexpr int NDMA = 3 * NB * CPR / RG_WAVE;
        static_assert(3 * NB * CPR % RG_WAVE == 0, "whole DMA instructions");
        typedef __attribute__((address_space(3))) void* lds_ptr;
#pragma unroll
        for (int q0 = 0; q0 < NDMA; q0 += SG_WAVES) {
            const int q = q0 + wave;
            if (q < NDMA) {                                          // wave-uniform
                const int sidx = q * RG_WAVE + lane;
                const int p = sidx / (NB * CPR), rem = sidx - p * (NB * CPR), n = rem / CPR, cs = rem - n * CPR;
                const uint16_t* src = g.Wt + (size_t)p * g.plane + (size_t)(n0 + n) * g.Kp + (((unsigned)cs ^ sg_swz<KT>((unsigned)n)) * 8);
                __builtin_amdgcn_global_load_lds((const void*)src, (lds_ptr)(Ws + (size_t)q * 1024), 16, 0, 0);
            }
        }
    }
    const int s_lo = ti.x, s_hi = ti.y;
    const bool one_cloud = s_lo == s_hi;                          // workgroup-uniform; almost always
    SG_STAMP();
    __syncthreads();                                              // (carries the vmcnt(0) that lands the DMA)
    SG_STAMP();

    // ---- A fragments: fold, split
    bf16x8 fa[KT][3];
    {
        const float2* sp = nullptr;
        if (INFOLD) {
            const int my_seg = one_cloud ? s_lo : rg_find_segment(g.seg_off, g.n_seg, row_ok ? row : g.M - 1);
            sp = g.a_stats + (size_t)my_seg * K + 8 * hi;
        }
#pragma unroll
        for (int ks = 0; ks < KT; ks++) {
            float x[8] = {raw[ks][0].x, raw[ks][0].y, raw[ks][0].z, raw[ks][0].w, raw[ks][1].x, raw[ks][1].y, raw[ks][1].z, raw[ks][1].w};
            if (INFOLD) {
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const float4 s4 = *(const float4*)(sp + 16 * ks + e);
                    float u = (x[e] - s4.x) * s4.y;
                    x[e] = fmaxf(u, u * g.a_slope);
                    u = (x[e + 1] - s4.z) * s4.w;
                    x[e + 1] = fmaxf(u, u * g.a_slope);
                }
            }
            unsigned w[4][3];
#pragma unroll
            for (int e = 0; e < 4; e++) sg_split2(row_ok ? x[2 * e] : 0.f, row_ok ? x[2 * e + 1] : 0.f, w[e][0], w[e][1], w[e][2]);
#pragma unroll
            for (int p = 0; p < 3; p++) fa[ks][p] = __builtin_bit_cast(bf16x8, make_uint4(w[0][p], w[1][p], w[2][p], w[3][p]));
        }
    }
    SG_STAMP();
    unsigned f_off[KT];
#pragma unroll
    for (int ks = 0; ks < KT; ks++) f_off[ks] = (unsigned)l31 * ROWB + (((unsigned)(2 * ks + hi)) ^ sg_swz<KT>((unsigned)l31)) * 16u;

    // ---- all MFMAs of the strip (NT independent accumulators)
    floatx16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KT; ks++)
#pragma unroll
        for (int j = 0; j < NT; j++) {
            bf16x8 fb[3];
#pragma unroll
            for (int p = 0; p < 3; p++)
                fb[p] = __builtin_bit_cast(bf16x8, *(const uint4*)(Ws + (size_t)(p * NB + 32 * j) * ROWB + f_off[ks]));
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][2], fb[0], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][1], fb[1], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][0], fb[2], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][1], fb[0], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][0], fb[1], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][0], fb[0], acc[j], 0, 0, 0);
        }

    const int rbase = r0 + 4 * hi;
#ifdef SG_PROF
    { float z = 0.f; for (int j = 0; j < NT; j++) z += acc[j][0]; if (z == 1.2345e30f) pt[0] = 0; }   // (the accumulators are complete)
#endif
    SG_STAMP();
    // ---- statistics of the Linear's own output: per cloud of the tile, per column (kpconv_blocks.py:510-519) -- BEFORE the stores: the
    // transposing store below consumes the accumulators in place
    if (g.stat_partial) {                                        // workgroup-uniform
    double2* red = (double2*)(Ws + (size_t)3 * NB * ROWB);        // [SG_WAVES][NB]
    for (int sg = s_lo; sg <= s_hi; sg++) {                      // workgroup-uniform; one pass almost always
        const int c_lo = sg == s_lo ? ti.z : g.seg_off[sg], c_hi = min(sg == s_lo ? ti.w : g.seg_off[sg + 1], g.M);
#pragma unroll
        for (int j = 0; j < NT; j++) {
            double s = 0.0, q = 0.0;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int rw = rbase + (r & 3) + 8 * (r >> 2);
                if (rw >= c_lo && rw < c_hi) { const double v = (double)acc[j][r]; s += v; q += v * v; }
            }
            s += __shfl_xor(s, 32, RG_WAVE); q += __shfl_xor(q, 32, RG_WAVE);
            if (hi == 0) red[wave * NB + 32 * j + l31] = make_double2(s, q);
        }
        __syncthreads();
        if (t < NB) {
            double2 a = red[t];
#pragma unroll
            for (int w = 1; w < SG_WAVES; w++) { const double2 b = red[w * NB + t]; a.x += b.x; a.y += b.y; }
            g.stat_partial[(size_t)(row_tile + sg) * N + n0 + t] = a;
        }
        if (sg < s_hi) __syncthreads();                          // red is reused by the next cloud
    }
    }
    SG_STAMP();
    // ---- store: a lane holds a COLUMN strip (rows 4 hi + (r & 3) + 8 (r >> 2) of column l31), every instruction writes two full 128-byte lines.
    // (Measured, round 4: the 4 x 4 blocks of a lane quad transposed by DPP quad_perm butterflies so that a lane owns four consecutive columns of
    //  one row -- 4 NT sixteen-byte stores per lane instead of 16 NT dwords, eight full lines per instruction: bit-identical and NOT faster,
    //  313 vs 287 us on the 0.6 M x 64 -> 256 shape; the wave's time in its store phase is queueing in the memory system, not instruction issue:
    //  profiles/r04_strip_phase_clocks.md.)
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int rw = rbase + (r & 3) + 8 * (r >> 2);
            if (rw < g.M) g.C[(size_t)rw * g.ldc + n0 + 32 * j + l31] = acc[j][r];
        }
#ifdef SG_PROF
    SG_STAMP();
    __builtin_amdgcn_s_waitcnt(0);
    SG_STAMP();
    if (lane == 0 && (row_tile == 700 || row_tile == 1500) && n0 == 0 && (wave == 0 || wave == 5))
        printf("gemm_strip KT %d NT %d row tile %d wave %d: loads+Wcopy %lld barrier %lld split %lld mfma %lld stats %lld store-issue %lld drain %lld total %lld\n", KT, NT, row_tile, wave,
               pt[1] - pt[0], pt[2] - pt[1], pt[3] - pt[2], pt[4] - pt[3], pt[5] - pt[4], pt[6] - pt[5], pt[7] - pt[6], pt[7] - pt[0]);
#endif
}

// columns per workgroup column: the widest of 128 / 64 / 32 that divides N and whose weight planes + reduction buffer fit 64 KB of LDS (two
// workgroups per CU, four for the small ones)
int sg_cols_per_wg(int N, int K)
{
    for (int nb = 128; nb >= 32; nb /= 2)
        if (nb <= N && N % nb == 0 && (size_t)nb * K * 6 + (size_t)SG_WAVES * nb * 16 <= 64 * 1024) return nb;
    return 0;
}

}  // namespace

extern "C" {

// 1 when regtr_gemm_stream serves the shape: K in {32, 64, 128}, N a multiple of 32 up to 512
int regtr_gemm_stream_supported(int M, int N, int K)
{
    return (M >= 0 && (K == 32 || K == 64 || K == 128) && N >= 32 && N <= 512 && N % 32 == 0 && sg_cols_per_wg(N, K) > 0) ? 1 : 0;
}

// rows per workgroup = statistics tile = the `rows` of the tile_info table (regtr_tile_segments) and of regtr_instnorm_finalize_tiles
int regtr_gemm_stream_tile_rows(void) { return SG_ROWS; }

// C[M, N] = A' W for the short-K, tall shapes of the shallow encoder levels; see the file header.
//   planes        weight planes of regtr_gemm_split_weights(W, .., N, K, ..)
//   a_stats       optional [n_seg, K, 2]: A' = LeakyReLU_a_slope(InstanceNorm(A)) per cloud (K <= 64)
//   seg_off       cloud offsets of the rows [n_seg + 1];  tile_info = regtr_tile_segments(seg_off, n_seg, M, 256, ..)
//   stat_partial  optional: (ceil(M / 256) + n_seg) * N double2 of per-(tile, cloud) column sums of C for
//                 regtr_instnorm_finalize_tiles(tile_rows = 256)
int regtr_gemm_stream(const float* A, int lda, const void* planes, float* C, int ldc, int M, int N, int K,
                      const float* a_stats, float a_slope, const int* seg_off, int n_seg, const void* tile_info,
                      double* stat_partial, void* stream)
{
    if (!A || !planes || !C || !regtr_gemm_stream_supported(M, N, K) || lda < K || ldc < N) return RG_ERR_ARG;
    if ((lda % 4) || ((uintptr_t)A % 16) || ((uintptr_t)planes % 16)) return RG_ERR_ARG;
    if (!seg_off || n_seg < 1 || !tile_info || ((uintptr_t)tile_info % 16)) return RG_ERR_ARG;
    if (a_stats && (K > 64 || ((uintptr_t)a_stats % 16))) return RG_ERR_ARG;
    if (stat_partial && ((uintptr_t)stat_partial % 16)) return RG_ERR_ARG;
    if (M == 0) return RG_OK;
    const int Npad = rg_cdiv(N, 128) * 128, Kp = rg_cdiv(K, 32) * 32;
    SgArgs g{A, (const uint16_t*)planes, C, (const float2*)a_stats, seg_off, (const int4*)tile_info, (double2*)stat_partial,
             (size_t)Npad * Kp, M, N, Kp, lda, ldc, n_seg, a_slope};
    const int NB = sg_cols_per_wg(N, K);
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)3 * NB * K * 2 + (size_t)SG_WAVES * NB * 16;
    const long long n_wg = (long long)rg_cdiv(rg_cdiv(M, SG_ROWS), 8) * 8 * (N / NB);     // (row tiles padded to the 8 XCDs) x column blocks
    if (n_wg > 0x7fffffffLL) return RG_ERR_ARG;
    const dim3 grid((unsigned)n_wg);
#define SG_L3(KT_, NT_) do { if (a_stats) k_gemm_strip<KT_, NT_, (KT_ <= 4)><<<grid, SG_WAVES * RG_WAVE, lds, st>>>(g); \
                             else k_gemm_strip<KT_, NT_, false><<<grid, SG_WAVES * RG_WAVE, lds, st>>>(g); } while (0)
#define SG_L2(KT_) do { if (NB == 32) SG_L3(KT_, 1); else if (NB == 64) SG_L3(KT_, 2); else SG_L3(KT_, 4); } while (0)
    if (K == 32) SG_L2(2);
    else if (K == 64) SG_L2(4);
    else SG_L2(8);
#undef SG_L2
#undef SG_L3
    RG_RETURN_IF_LAUNCH_FAILED();
    return RG_OK;
}

}  // extern "C"
