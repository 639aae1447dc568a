// C[M,N] = epilogue( A[M,K] * B[K,N] )   float32 in / float32 accumulate on the CDNA4 matrix cores
// (v_mfma_f32_32x32x2_f32: exact f32, bit-for-bit an fmaf chain in k order).
//
// Serves every dense contraction of the RegTR hot path: KPConv's [Nq,15*Cin] x [15*Cin,Cout] kernel-point
// contraction (kpconv_blocks.py:401-406), the unary / shortcut linears (kpconv_blocks.py:557), feat_proj
// (regtr.py:145), the attention in/out projections and FFN (transformers.py:197-238) and the correspondence head
// (regtr.py:432-436).  Weights are stored pre-transposed as B[K,N] row-major so B-fragment loads are contiguous.
//
// Tiles: 256-thread workgroups of 4 waves, one 32x32 accumulator per wave, arranged 2x2 (64x64 tile) or, for thin
// outputs (N <= 32: the level-0 KPConv contraction and unary1), 4x1 (128x32 tile) so no MFMA work is spent on
// padding columns.  BK = 32, LDS double-buffered with the next tile's global loads issued before the current MFMAs.
// Small-M / deep-K problems (the 750-token transformer and level-3 KPConv contractions: a few hundred tiles with a
// serial chain of K/2 MFMAs each) are split along K across workgroups; partial tiles go to a workspace and a second
// kernel reduces them in fixed order (deterministic) and applies the epilogue.
// Fused epilogue:   v = acc ; v /= row_div[m] ; v += bias[n] ; act ; v += residual[m,n]
// Optional A-operand transform (per-cloud InstanceNorm + LeakyReLU folded into the load, kpconv_blocks.py:727-730):
//   a[m,k] <- lrelu((a[m,k] - mean[cloud(m),k]) * rstd[cloud(m),k])
#include "common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int LDA_S = BK + 1;   // odd stride: A-fragment column reads hit 32 distinct banks

struct GemmArgs {
    const float* A; const float* B; float* C;
    const float* bias; const float* row_div; const float* residual;
    const float2* a_stats; const int* a_seg_off;   // optional fused InstanceNorm+LeakyReLU on A
    float* partial;                                  // split-K workspace [S][M][N] (raw accumulators)
    int M, N, K, lda, ldb, ldc, ldr, act, n_seg, k_chunk;
    float a_slope;
};

// Operand tiles into registers.  Every load is UNCONDITIONAL, from a clamped row / column / k, and NOTHING here consumes a loaded
// value (clamped garbage lands in rows and columns that are never stored; what lies beyond k_end is zeroed by stage() from the
// returned mask, when the registers go to LDS two tiles later): a load under a per-lane predicate, or one whose value feeds a
// select right away, compiles to load + s_waitcnt vmcnt(0) -- the tile's loads became serial memory round trips in front of the
// MFMAs they were meant to hide behind (tools/isa_scan.py).  mask: bit 4 i + j = element j of A vector i is inside k_end,
// bit 16 + 4 i + j the same for B (also inside N).
template <bool ALIGNED_A, bool ALIGNED_B, int BM, int BN>
__device__ __forceinline__ unsigned load_tiles(const GemmArgs& g, int m0, int n0, int k0, int k_end,
                                               float (&ra)[BM * BK / 1024][4], float (&rb)[BK * BN / 1024][4], const int* row_seg)
{
    const int t = threadIdx.x;
    constexpr int KV = BK / 4;                      // float4 per A row
    unsigned mask = 0;
#pragma unroll
    for (int i = 0; i < BM * BK / 1024; i++) {      // A: BM rows x BK k -> (row = idx / KV, k4 = (idx % KV) * 4)
        const int idx = t + i * 256;
        const int row = min(m0 + idx / KV, g.M - 1), k = k0 + (idx % KV) * 4;
        if (ALIGNED_A) {                            // (k_end % 4 == 0: a float4 is wholly inside or wholly outside)
            const float4 v = *(const float4*)(g.A + (size_t)row * g.lda + min(k, k_end - 4));
            ra[i][0] = v.x; ra[i][1] = v.y; ra[i][2] = v.z; ra[i][3] = v.w;
            mask |= (k < k_end ? 0xfu : 0u) << (4 * i);
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                ra[i][j] = g.A[(size_t)row * g.lda + min(k + j, k_end - 1)];
                mask |= (k + j < k_end ? 1u : 0u) << (4 * i + j);
            }
        }
        if (g.a_stats) {                            // (the folded-InstanceNorm operand: rare on this kernel, transformed at once)
            const float2* st = g.a_stats + (size_t)row_seg[i] * g.K;
            float2 s[4];
#pragma unroll
            for (int j = 0; j < 4; j++) s[j] = st[min(k + j, k_end - 1)];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float v = (ra[i][j] - s[j].x) * s[j].y;
                ra[i][j] = v > 0.f ? v : v * g.a_slope;
            }
        }
    }
    constexpr int NV = BN / 4;                      // float4 per B row
#pragma unroll
    for (int i = 0; i < BK * BN / 1024; i++) {      // B: BK k x BN n -> (k = idx / NV, n4 = (idx % NV) * 4)
        const int idx = t + i * 256;
        const int k = k0 + idx / NV, n = n0 + (idx % NV) * 4;
        const int kc = min(k, k_end - 1);
        if (ALIGNED_B) {                            // (N % 4 == 0)
            const float4 v = *(const float4*)(g.B + (size_t)kc * g.ldb + min(n, g.N - 4));
            rb[i][0] = v.x; rb[i][1] = v.y; rb[i][2] = v.z; rb[i][3] = v.w;
            mask |= ((k < k_end && n < g.N) ? 0xfu : 0u) << (16 + 4 * i);
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                rb[i][j] = g.B[(size_t)kc * g.ldb + min(n + j, g.N - 1)];
                mask |= ((k < k_end && n + j < g.N) ? 1u : 0u) << (16 + 4 * i + j);
            }
        }
    }
    return mask;
}

// WMW x WNW waves, each a 32x32 accumulator
// ALIGNED_A / ALIGNED_B: the operand may be read as float4 (the thin head Linears -- N = 3, 1 -- still stream their A operand so)
template <bool ALIGNED_A, bool ALIGNED_B, int WMW, int WNW>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) k_gemm_f32(GemmArgs g)
{
    constexpr int BM = 32 * WMW, BN = 32 * WNW, LDB_S = BN + 1;
    __shared__ float As[2][BM * LDA_S];
    __shared__ float Bs[2][BK * LDB_S];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WNW, wn = wave % WNW;
    const int l31 = lane & 31, hi = lane >> 5;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int k_begin = blockIdx.z * g.k_chunk;
    const int k_end = min(g.K, k_begin + g.k_chunk);
    const int nk = (k_end - k_begin + BK - 1) / BK;

    constexpr int AV = BM * BK / 1024, BV = BK * BN / 1024, KV = BK / 4, NV = BN / 4;
    int row_seg[AV];
    int seg_first = 0, seg_last = 0;      // (wave-parallel lookups; only tiles straddling clouds search per row)
    if (g.a_stats) {
        seg_first = rg_find_segment_wave(g.a_seg_off, g.n_seg, m0);
        seg_last = rg_find_segment_wave(g.a_seg_off, g.n_seg, min(m0 + BM, g.M) - 1);
    }
#pragma unroll
    for (int i = 0; i < AV; i++) {
        row_seg[i] = 0;
        const int row = m0 + (t + i * 256) / KV;
        if (g.a_stats && row < g.M) row_seg[i] = seg_first == seg_last ? seg_first : rg_find_segment(g.a_seg_off, g.n_seg, row);
    }

    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;

    // Two register stages (R0/R1) feed two LDS buffers: the global loads of tile t+2 are issued before the MFMAs of
    // tile t and only written to LDS after the MFMAs of tile t+1, so every load has ~2 tiles of matrix-core work
    // (>= 2048 cycles) to cover its L2 / Infinity-Cache latency even with a single workgroup per CU.
    float ra0[AV][4], rb0[BV][4], ra1[AV][4], rb1[BV][4];
    unsigned in0 = 0, in1 = 0;                    // load_tiles' in-range masks of the two register stages
    auto stage = [&](int buf, float (&ra)[AV][4], float (&rb)[BV][4], unsigned in) {
#pragma unroll
        for (int i = 0; i < AV; i++) {
            const int idx = t + i * 256;
            float* a = &As[buf][(idx / KV) * LDA_S + (idx % KV) * 4];
#pragma unroll
            for (int j = 0; j < 4; j++) a[j] = (in >> (4 * i + j) & 1u) ? ra[i][j] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < BV; i++) {
            const int idx = t + i * 256;
            float* b = &Bs[buf][(idx / NV) * LDB_S + (idx % NV) * 4];
#pragma unroll
            for (int j = 0; j < 4; j++) b[j] = (in >> (16 + 4 * i + j) & 1u) ? rb[i][j] : 0.f;
        }
    };
    auto compute = [&](int buf) {
        const float* a = &As[buf][(wm * 32 + l31) * LDA_S + hi];
        const float* b = &Bs[buf][hi * LDB_S + wn * 32 + l31];
#pragma unroll
        for (int s = 0; s < BK / 2; s++)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * s], b[2 * s * LDB_S], acc, 0, 0, 0);
    };
    in0 = load_tiles<ALIGNED_A, ALIGNED_B, BM, BN>(g, m0, n0, k_begin, k_end, ra0, rb0, row_seg);
    if (nk > 1) in1 = load_tiles<ALIGNED_A, ALIGNED_B, BM, BN>(g, m0, n0, k_begin + BK, k_end, ra1, rb1, row_seg);
    stage(0, ra0, rb0, in0);
    __syncthreads();
    // (the prefetches are unconditional: past the last tile they re-read a clamped in-range tile that is never staged -- a load
    //  under `if (kt + 2 < nk)` into these loop-carried registers would be waited for on the spot)
    const int k_last = k_begin + (nk - 1) * BK;      // the last real tile's k0 (aligned like every k0)
    for (int kt = 0; kt < nk; kt += 2) {
        // even tile kt lives in LDS buffer 0
        in0 = load_tiles<ALIGNED_A, ALIGNED_B, BM, BN>(g, m0, n0, min(k_begin + (kt + 2) * BK, k_last), k_end, ra0, rb0, row_seg);
        compute(0);
        if (kt + 1 < nk) stage(1, ra1, rb1, in1);
        __syncthreads();
        if (kt + 1 >= nk) break;
        // odd tile kt + 1 lives in LDS buffer 1
        in1 = load_tiles<ALIGNED_A, ALIGNED_B, BM, BN>(g, m0, n0, min(k_begin + (kt + 3) * BK, k_last), k_end, ra1, rb1, row_seg);
        compute(1);
        if (kt + 2 < nk) stage(0, ra0, rb0, in0);
        __syncthreads();
    }

    const int col = n0 + wn * 32 + l31;
    if (col >= g.N) return;
    if (g.partial) {   // split-K: raw accumulators, epilogue happens in k_splitk_reduce
        float* P = g.partial + (size_t)blockIdx.z * g.M * g.N;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (row < g.M) P[(size_t)row * g.N + col] = acc[r];
        }
        return;
    }
    const float bv = g.bias ? g.bias[col] : 0.f;
    float rd[16], rs[16];                                  // clamped rows, all loads in flight before the first use
    if (g.row_div) {
#pragma unroll
        for (int r = 0; r < 16; r++) rd[r] = g.row_div[min(m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, g.M - 1)];
    }
    if (g.residual) {
#pragma unroll
        for (int r = 0; r < 16; r++) rs[r] = g.residual[(size_t)min(m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, g.M - 1) * g.ldr + col];
    }
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        float v = acc[r];
        if (g.row_div) v = v / rd[r];
        v += bv;
        if (g.act == 1) v = fmaxf(v, 0.f);
        if (g.residual) v += rs[r];
        if (row < g.M) g.C[(size_t)row * g.ldc + col] = v;
    }
}

__global__ void __launch_bounds__(256) k_splitk_reduce(GemmArgs g, int S)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)g.M * g.N) return;
    const int row = (int)(e / g.N), col = (int)(e % g.N);
    float v = 0.f;
    for (int s = 0; s < S; s++) v += g.partial[(size_t)s * g.M * g.N + e];   // fixed order: deterministic
    if (g.row_div) v = v / g.row_div[row];
    if (g.bias) v += g.bias[col];
    if (g.act == 1) v = fmaxf(v, 0.f);
    if (g.residual) v += g.residual[(size_t)row * g.ldr + col];
    g.C[(size_t)row * g.ldc + col] = v;
}

// number of K splits for a problem (host policy; 1 = no split)
int choose_splits(int M, int N, int K, int bm, int bn)
{
    const long long tiles = (long long)rg_cdiv(M, bm) * rg_cdiv(N, bn);
    if (tiles >= 384 || K < 512) return 1;
    int s = (int)((768 + tiles - 1) / tiles);
    const int max_by_k = K / 128;          // keep >= 128 of K per split
    if (s > max_by_k) s = max_by_k;
    if (s > 16) s = 16;
    return s < 2 ? 1 : s;
}

}  // namespace

extern "C" {

size_t regtr_gemm_f32_ws_bytes(int M, int N, int K)
{
    const int bm = N <= 32 ? 128 : 64, bn = N <= 32 ? 32 : 64;
    const int s = choose_splits(M, N, K, bm, bn);
    return s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
}

// act: 0 none, 1 ReLU.  bias [N], row_div [M], residual [M, ldr] are optional (NULL).
// a_stats [n_seg, K, 2] + a_seg_off [n_seg + 1] (optional): fold lrelu(InstanceNorm(A)) into the A load.
// ws: regtr_gemm_f32_ws_bytes(M, N, K) bytes (may be NULL when that is 0).
int regtr_gemm_f32(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                   const float* bias, const float* row_div, const float* residual, int ldr, int act,
                   const float* a_stats, const int* a_seg_off, int n_seg, float a_slope, void* ws, size_t ws_bytes,
                   void* stream)
{
    if (!A || !B || !C || M < 0 || N < 1 || K < 1 || lda < K || ldb < N || ldc < N) return RG_ERR_ARG;
    if (a_stats && (!a_seg_off || n_seg < 1)) return RG_ERR_ARG;
    if (M == 0) return RG_OK;
    const bool thin = N <= 32;
    const int bm = thin ? 128 : 64, bn = thin ? 32 : 64;
    const int S = choose_splits(M, N, K, bm, bn);
    if (S > 1 && (!ws || ws_bytes < (size_t)S * M * N * sizeof(float))) return RG_ERR_WORKSPACE;
    int k_chunk = K;
    if (S > 1) k_chunk = rg_cdiv(rg_cdiv(K, S), BK) * BK;
    const int S_eff = rg_cdiv(K, k_chunk);
    GemmArgs g{A, B, C, bias, row_div, residual, (const float2*)a_stats, a_seg_off, S_eff > 1 ? (float*)ws : nullptr,
               M, N, K, lda, ldb, ldc, ldr, act, n_seg, k_chunk, a_slope};
    const bool aligned_a = (K % 4 == 0) && (lda % 4 == 0) && ((uintptr_t)A % 16 == 0) && (k_chunk % 4 == 0);
    const bool aligned_b = (N % 4 == 0) && (ldb % 4 == 0) && ((uintptr_t)B % 16 == 0);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(rg_cdiv(M, bm), rg_cdiv(N, bn), S_eff);
    if (thin) {
        if (aligned_a && aligned_b) k_gemm_f32<true, true, 4, 1><<<grid, 256, 0, st>>>(g);
        else if (aligned_a) k_gemm_f32<true, false, 4, 1><<<grid, 256, 0, st>>>(g);
        else k_gemm_f32<false, false, 4, 1><<<grid, 256, 0, st>>>(g);
    } else {
        if (aligned_a && aligned_b) k_gemm_f32<true, true, 2, 2><<<grid, 256, 0, st>>>(g);
        else k_gemm_f32<false, false, 2, 2><<<grid, 256, 0, st>>>(g);
    }
    if (S_eff > 1) k_splitk_reduce<<<rg_cdiv((long long)M * N, 256), 256, 0, st>>>(g, S_eff);
    RG_RETURN_IF_LAUNCH_FAILED();
    return RG_OK;
}

}  // extern "C"
