// C[M,N] = epilogue( A[M,K] * B[K,N] )   float32 in / float32 accumulate on the CDNA4 matrix cores
// (v_mfma_f32_32x32x2_f32: exact f32, bit-for-bit an fmaf chain in k order).
//
// Serves every dense contraction of the RegTR hot path: KPConv's [Nq,15*Cin] x [15*Cin,Cout] kernel-point
// contraction (kpconv_blocks.py:401-406), the unary / shortcut linears (kpconv_blocks.py:557), feat_proj
// (regtr.py:145), the attention in/out projections and FFN (transformers.py:197-238) and the correspondence head
// (regtr.py:432-436).  Weights are stored pre-transposed as B[K,N] row-major so B-fragment loads are contiguous.
//
// Tile: 64x64 per 256-thread workgroup (4 waves, 2x2, one 32x32 accumulator each), BK = 16, LDS double-buffered
// with the next tile's global loads issued before the MFMAs of the current one.  Fused epilogue:
//   v = acc ; v /= row_div[m] ; v += bias[n] ; act ; v += residual[m,n]
#include "common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int BM = 64, BN = 64, BK = 16;
constexpr int LDA_S = BK + 1;   // odd stride: A-fragment column reads hit 32 distinct banks
constexpr int LDB_S = BN + 1;

struct GemmArgs {
    const float* A; const float* B; float* C;
    const float* bias; const float* row_div; const float* residual;
    int M, N, K, lda, ldb, ldc, ldr, act;
};

template <bool ALIGNED>
__device__ __forceinline__ void load_tiles(const GemmArgs& g, int m0, int n0, int k0, float (&ra)[4], float (&rb)[4])
{
    const int t = threadIdx.x;
    {   // A: 64 rows x 16 k  -> thread (row = t/4, k4 = (t%4)*4)
        const int row = m0 + (t >> 2), k = k0 + (t & 3) * 4;
        if (ALIGNED) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < g.M && k < g.K) v = *(const float4*)(g.A + (size_t)row * g.lda + k);
            ra[0] = v.x; ra[1] = v.y; ra[2] = v.z; ra[3] = v.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) ra[j] = (row < g.M && k + j < g.K) ? g.A[(size_t)row * g.lda + k + j] : 0.f;
        }
    }
    {   // B: 16 k x 64 n -> thread (k = t/16, n4 = (t%16)*4)
        const int k = k0 + (t >> 4), n = n0 + (t & 15) * 4;
        if (ALIGNED) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < g.K && n < g.N) v = *(const float4*)(g.B + (size_t)k * g.ldb + n);
            rb[0] = v.x; rb[1] = v.y; rb[2] = v.z; rb[3] = v.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) rb[j] = (k < g.K && n + j < g.N) ? g.B[(size_t)k * g.ldb + n + j] : 0.f;
        }
    }
}

template <bool ALIGNED>
__global__ void __launch_bounds__(256) k_gemm_f32(GemmArgs g)
{
    __shared__ float As[2][BM * LDA_S];
    __shared__ float Bs[2][BK * LDB_S];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int nk = (g.K + BK - 1) / BK;

    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;

    float ra[4], rb[4];
    load_tiles<ALIGNED>(g, m0, n0, 0, ra, rb);
    auto stage = [&](int buf) {
        float* a = &As[buf][(t >> 2) * LDA_S + (t & 3) * 4];
        a[0] = ra[0]; a[1] = ra[1]; a[2] = ra[2]; a[3] = ra[3];
        float* b = &Bs[buf][(t >> 4) * LDB_S + (t & 15) * 4];
        b[0] = rb[0]; b[1] = rb[1]; b[2] = rb[2]; b[3] = rb[3];
    };
    stage(0);
    __syncthreads();

    for (int kt = 0; kt < nk; kt++) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tiles<ALIGNED>(g, m0, n0, (kt + 1) * BK, ra, rb);
        const float* a = &As[cur][(wm * 32 + l31) * LDA_S + hi];
        const float* b = &Bs[cur][hi * LDB_S + wn * 32 + l31];
#pragma unroll
        for (int s = 0; s < BK / 2; s++)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * s], b[2 * s * LDB_S], acc, 0, 0, 0);
        if (kt + 1 < nk) stage(cur ^ 1);
        __syncthreads();
    }

    const int col = n0 + wn * 32 + l31;
    if (col >= g.N) return;
    const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (row >= g.M) continue;
        float v = acc[r];
        if (g.row_div) v = v / g.row_div[row];
        v += bv;
        if (g.act == 1) v = fmaxf(v, 0.f);
        if (g.residual) v += g.residual[(size_t)row * g.ldr + col];
        g.C[(size_t)row * g.ldc + col] = v;
    }
}

}  // namespace

extern "C" {

// act: 0 none, 1 ReLU.  bias [N], row_div [M], residual [M, ldr] are optional (NULL).
int regtr_gemm_f32(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                   const float* bias, const float* row_div, const float* residual, int ldr, int act, void* stream)
{
    if (!A || !B || !C || M < 0 || N < 1 || K < 1 || lda < K || ldb < N || ldc < N) return RG_ERR_ARG;
    if (M == 0) return RG_OK;
    GemmArgs g{A, B, C, bias, row_div, residual, M, N, K, lda, ldb, ldc, ldr, act};
    const bool aligned = (K % 4 == 0) && (N % 4 == 0) && (lda % 4 == 0) && (ldb % 4 == 0) &&
                         (((uintptr_t)A | (uintptr_t)B) % 16 == 0);
    dim3 grid(rg_cdiv(M, BM), rg_cdiv(N, BN));
    if (aligned) k_gemm_f32<true><<<grid, 256, 0, (hipStream_t)stream>>>(g);
    else k_gemm_f32<false><<<grid, 256, 0, (hipStream_t)stream>>>(g);
    RG_RETURN_IF_LAUNCH_FAILED();
    return RG_OK;
}

}  // extern "C"
