// The tail of a resnet bottleneck block whose shortcut is a Linear, as ONE pass over the block's narrow inputs:
//     y = LeakyReLU( InstanceNorm(x1' W2) + InstanceNorm(f Ws) )          x1' = LeakyReLU(InstanceNorm(conv output))
// (/root/reference/src/models/backbone_kpconv/kpconv_blocks.py:727-741: unary2 and unary_shortcut are Linear -> InstanceNorm, :556-561).
//
// The straightforward form writes both products (out_dim columns each), reads them back to normalise and add, and writes y: at level
// 0 of a 64-pair forward that is 7.1 GB of HBM traffic for 0.9 GB of inputs and 1.2 GB of result.  InstanceNorm needs the per-cloud
// statistics of a product before any of it can be finalised -- but they follow from the statistics of its INPUT: for u = x W,
//     mean(u_j) = sum_a mean(x_a) W_aj,        var(u_j) = sum_ab W_aj W_bj Cov(x)_ab,
// so one pass over the narrow inputs (K <= 64 columns) that accumulates their K x K second moments per cloud replaces the pass over
// the wide products.  Three launches:
//   k_moments       per (2048-row chunk, cloud): S1 = sum x, S2 = sum x x^T on v_mfma_f32_32x32x2_f32 (operands A = B = the loaded
//                   value itself: lane (row parity, channel)), float32 over a wave's 512 rows then float64; waves of a workgroup are added in
//                   fixed order through LDS: deterministic.
//   k_tail_prepare  per cloud: float64 covariance, mean / rstd of both products, the weights scaled by rstd_j and split ONCE PER CLOUD
//                   into the three bf16 planes of gemm_x3.hip's exact split; the column means are handed over as INPUT means
//                   (the kernel centres its operands, so no bias term and no cancellation of two large numbers).
//   k_tail_strip    the strip GEMM of gemm_stream.hip with two A sources and the cloud's planes: 32 rows per wave straight from global
//                   memory into MFMA fragments (fold, centre, split in registers), one accumulator set for both products, LeakyReLU,
//                   store.  A 256-row tile that straddles clouds repeats the MFMAs per cloud with that cloud's planes (1 tile in 70).
// Float32-grade like every dense op here: six bf16 MFMAs per product block; the statistics are those of the exact products (float64),
// where the straightforward form takes them from float32-rounded values: the two agree to ~1e-7 relative.
#include "common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int MO_WAVES = 4;
constexpr int MO_ROWS_WAVE = 512;
constexpr int MO_ROWS_WG = MO_WAVES * MO_ROWS_WAVE;      // rows of one cloud per workgroup

// partial[(cloud * n_chunks + chunk) * (K * K + K)]: K x K second moments (32 x 32 blocks (ta, tb) with ta <= tb only; the rest is the
// transpose, k_tail_prepare reads it from there) then the K sums
// k_valid < 32 (KC = 1 only): the rows hold k_valid columns, lanes beyond contribute zeros and the output is the compact
// k_valid x k_valid layout.  row_div (optional): x = A[row] / row_div[row] (KPConv's neighbour-count normaliser, kpconv_blocks.py:411).
template <int KC, bool INFOLD, bool ROWDIV>      // K = 32 KC
__global__ void __launch_bounds__(MO_WAVES* RG_WAVE, 4) k_moments(const float* __restrict__ A, int lda, const float2* __restrict__ stats,
                                                               float slope, const float* __restrict__ row_div, int k_valid,
                                                               const int* __restrict__ seg_off, int n_chunks,
                                                               double* __restrict__ partial, float* __restrict__ pivot)
{
    constexpr int K = 32 * KC, NTILE = KC * (KC + 1) / 2;
    __shared__ double red[K * K + K];
    const int cloud = blockIdx.y, chunk = blockIdx.x;
    const int c_end = seg_off[cloud + 1];
    const int wg_begin = seg_off[cloud] + chunk * MO_ROWS_WG;
    if (wg_begin >= c_end) return;                                   // k_tail_prepare knows how many chunks a cloud has
    const int lane = rg_lane(), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, ch = lane & 31;
    const bool ch_ok = ch < k_valid;
    const int chc = ch_ok ? ch : 0;
    const int w_begin = wg_begin + wave * MO_ROWS_WAVE, w_end = min(c_end, w_begin + MO_ROWS_WAVE);
    float mu[KC], rs[KC];
#pragma unroll
    for (int c = 0; c < KC; c++) {
        mu[c] = 0.f; rs[c] = 1.f;
        if (INFOLD) { const float2 s = stats[(size_t)cloud * K + 32 * c + chc]; mu[c] = s.x; rs[c] = s.y; }
    }
    // The moments are those of x - pivot: covariances do not change under a shift, and the float32 rounding of the sums (~1e-6 of
    // sum x^2) then scales with (spread + |mean - pivot|)^2 of a channel instead of with its mean^2 -- a near-constant channel (variance <<
    // mean^2) would otherwise lose its variance to that rounding.  pivot = the trimmed mean (largest and smallest dropped) of SIXTEEN rows spread evenly over the cloud (every
    // wave and chunk of the cloud loads the same ones and adds them in the same order: one value per cloud and channel).  Round 5: it
    // used to be the cloud's FIRST row -- on a real 3DMatch fragment that row can be an outlier (an isolated point whose kernel-point sums
    // are a tenth of the typical ones), the pivot then sits ~6 sigma from the mean and the statistics lose a factor ~36: 3e-5 on the encoder
    // output of one cloud in nine (profiles/r05_real_diag.txt), against 2e-6 from a pivot near the mean.  k_tail_prepare adds the pivot back.
    float pv[KC];
    {
        const int r0 = seg_off[cloud];
        const long long n_c = c_end - r0;
        float v16[16][KC], d16[ROWDIV ? 16 : 1];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int row = r0 + (int)(((2 * i + 1) * n_c) >> 5);
#pragma unroll
            for (int c = 0; c < KC; c++) v16[i][c] = A[(size_t)row * lda + 32 * c + chc];
            if (ROWDIV) d16[i] = row_div[row];
        }
#pragma unroll
        for (int c = 0; c < KC; c++) {
            float acc16 = 0.f, lo = 3.0e38f, hi = -3.0e38f;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                float v = v16[i][c];
                if (ROWDIV) v = v / d16[ROWDIV ? i : 0];
                if (INFOLD) { const float u = (v - mu[c]) * rs[c]; v = fmaxf(u, u * slope); }
                acc16 += v; lo = fminf(lo, v); hi = fmaxf(hi, v);
            }
            pv[c] = ch_ok ? (acc16 - lo - hi) * (1.f / 14.f) : 0.f;      // trimmed: one sampled outlier does not move it
        }
    }
    // float32 accumulation over the wave's <= 512 rows (rounding ~1e-6 of a sum, independent between the ~40 waves of a cloud),
    // float64 from there on; per-64-row float64 accumulators cost 96 registers and a third of the occupancy
    floatx16 acc[NTILE];
    float s1[KC];
#pragma unroll
    for (int t = 0; t < NTILE; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
#pragma unroll
    for (int c = 0; c < KC; c++) s1[c] = 0.f;
    // 32-row sub-chunks, double-buffered in registers: the 16 KC independent 4-byte loads per lane of the next sub-chunk are in flight
    // under the MFMAs of the current one
    constexpr int SUB = 16;
    constexpr int DS = ROWDIV ? SUB : 1;
    auto load = [&](float (&xv)[SUB][KC], float (&dv)[DS], int base) {
#pragma unroll
        for (int s = 0; s < SUB; s++) {
            const int row = base + 2 * s + half;
            const int rcl = row < w_end ? row : w_end - 1;
            const float* ap = A + (size_t)rcl * lda + chc;
#pragma unroll
            for (int c = 0; c < KC; c++) xv[s][c] = ap[32 * c];
            if (ROWDIV) dv[s] = row_div[rcl];
        }
    };
    auto consume = [&](const float (&xv)[SUB][KC], const float (&dv)[DS], int base) {
#pragma unroll
        for (int s = 0; s < SUB; s++) {
            const bool ok = base + 2 * s + half < w_end && ch_ok;
            float x[KC];
#pragma unroll
            for (int c = 0; c < KC; c++) {
                float v = xv[s][c];
                if (ROWDIV) v = v / dv[ROWDIV ? s : 0];
                if (INFOLD) { const float u = (v - mu[c]) * rs[c]; v = fmaxf(u, u * slope); }
                x[c] = ok ? v - pv[c] : 0.f;
                s1[c] += x[c];
            }
            int t = 0;
#pragma unroll
            for (int ta = 0; ta < KC; ta++)
#pragma unroll
                for (int tb = ta; tb < KC; tb++, t++)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[ta], x[tb], acc[t], 0, 0, 0);
        }
    };
    float xa[SUB][KC], xb[SUB][KC], da[DS], db[DS];
    if (w_begin < w_end) load(xa, da, w_begin);
    for (int base = w_begin; base < w_end; base += 4 * SUB) {        // wave-uniform
        load(xb, db, base + 2 * SUB);
        consume(xa, da, base);
        load(xa, da, base + 4 * SUB);
        consume(xb, db, base + 2 * SUB);
    }
    // ---- the workgroup's waves, added in wave order (deterministic).  D layout: row m = (r & 3) + 8 (r >> 2) + 4 half, column ch
    for (int w = 0; w < MO_WAVES; w++) {
        if (wave == w && w_begin < w_end) {
            int t = 0;
#pragma unroll
            for (int ta = 0; ta < KC; ta++)
#pragma unroll
                for (int tb = ta; tb < KC; tb++, t++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int a = 32 * ta + (r & 3) + 8 * (r >> 2) + 4 * half, b = 32 * tb + ch;
                        const double v = (w == 0 ? 0.0 : red[a * K + b]) + (double)acc[t][r];
                        red[a * K + b] = v;                          // (no mirrored write: 32-way bank conflict)
                    }
#pragma unroll
            for (int c = 0; c < KC; c++) {
                const double v = (double)s1[c] + (double)__shfl_xor(s1[c], 32, RG_WAVE);
                if (half == 0) red[K * K + 32 * c + ch] = (w == 0 ? 0.0 : red[K * K + 32 * c + ch]) + v;
            }
        }
        __syncthreads();
    }
    const int kv = k_valid;
    double* out = partial + ((size_t)cloud * n_chunks + chunk) * (kv * kv + kv);
    for (int i = threadIdx.x; i < kv * kv + kv; i += MO_WAVES * RG_WAVE) {
        int src = K * K + (i - kv * kv);                             // the sums
        if (i < kv * kv) { const int a = i / kv, b = i - a * kv; src = a * K + b; }
        out[i] = red[src];
    }
    if (chunk == 0 && wave == 0 && half == 0 && ch_ok) {
#pragma unroll
        for (int c = 0; c < KC; c++) pivot[(size_t)cloud * kv + 32 * c + ch] = pv[c];
    }
}

__device__ __forceinline__ unsigned bt_pack(float a, float b)
{
    bf16x2 v;
    v.x = (__bf16)a; v.y = (__bf16)b;
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void bt_split2(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2)
{
    p0 = bt_pack(a, b);
    const float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);
    p1 = bt_pack(ra, rb);
    p2 = bt_pack(ra - __uint_as_float(p1 << 16), rb - __uint_as_float(p1 & 0xffff0000u));
}

// One source of one cloud: covariance into LDS, then per output column j: mean / rstd of (x W)_j and the scaled + split weight row.
// 256 threads = 64 columns x 4 row quarters; the workgroup covers columns [col0, col0 + 64).
// planes[cloud][p][N][K] bf16 (k contiguous), in_mean[cloud][K], out_stats (optional) [cloud][N] (mean, rstd).
template <int K>
__device__ void bt_prepare_source(const double* __restrict__ partial, int n_chunks_total, int n_valid, int n_rows, int cloud,
                                  const float* __restrict__ W_kn, int N, int col0, float eps, double* cov /*LDS [K*K + K]*/,
                                  double2* quad /*LDS [256]*/, uint16_t* __restrict__ planes, float* __restrict__ in_mean,
                                  float2* __restrict__ out_stats, const float* __restrict__ pivot)
{
    const double inv_n = 1.0 / (double)n_rows;
    for (int i = threadIdx.x; i < K * K + K; i += blockDim.x) {
        int src = i;
        if (i < K * K) {                                             // blocks below the diagonal: the transpose
            const int a = i / K, b = i - a * K;
            if ((a >> 5) > (b >> 5)) src = b * K + a;
        }
        double s = 0.0;
        for (int c = 0; c < n_valid; c++) s += partial[((size_t)cloud * n_chunks_total + c) * (K * K + K) + src];   // fixed order
        cov[i] = s * inv_n;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < K * K; i += blockDim.x) {          // centred second moments (of x - pivot: the same as of x)
        const int a = i / K, b = i - a * K;
        cov[i] -= cov[K * K + a] * cov[K * K + b];
    }
    __syncthreads();
    if (threadIdx.x < K) {                                           // mean(x) = pivot + mean(x - pivot)
        cov[K * K + threadIdx.x] += (double)pivot[(size_t)cloud * K + threadIdx.x];
        if (col0 == 0) in_mean[(size_t)cloud * K + threadIdx.x] = (float)cov[K * K + threadIdx.x];
    }
    __syncthreads();
    const int j = col0 + (threadIdx.x & 63), h = threadIdx.x >> 6;   // column, row quarter (= wave)
    float wj[K];                                                     // the column of W, once (coalesced across the threads)
#pragma unroll
    for (int a = 0; a < K; a++) wj[a] = W_kn[(size_t)a * N + j];
    double mean = 0.0, var = 0.0;
#pragma unroll 1
    for (int a = h * (K / 4); a < (h + 1) * (K / 4); a++) {          // var_j = sum_a W_aj (sum_b Cov_ab W_bj)
        double t = 0.0;
#pragma unroll
        for (int b = 0; b < K; b++) t += cov[a * K + b] * (double)wj[b];          // LDS broadcast reads
        const double wa = (double)W_kn[(size_t)a * N + j];           // (wj[a] by a dynamic index would spill the array; L1 hit)
        mean += cov[K * K + a] * wa;
        var += wa * t;
    }
    quad[threadIdx.x] = make_double2(mean, var);
    __syncthreads();
    {
        const int c = threadIdx.x & 63;
        const double2 q0 = quad[c], q1 = quad[64 + c], q2 = quad[128 + c], q3 = quad[192 + c];
        mean = (q0.x + q1.x) + (q2.x + q3.x);                        // the same order in all four threads of a column
        var = (q0.y + q1.y) + (q2.y + q3.y);
    }
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (out_stats && h == 0) out_stats[(size_t)cloud * N + j] = make_float2((float)mean, rstd);
    uint16_t* row = planes + ((size_t)cloud * 3 * N + j) * K;        // plane p at + p N K
    for (int k = h * (K / 4); k < (h + 1) * (K / 4); k += 2) {
        unsigned p0, p1, p2;
        bt_split2(W_kn[(size_t)k * N + j] * rstd, W_kn[(size_t)(k + 1) * N + j] * rstd, p0, p1, p2);   // (L1 hits)
        *(unsigned*)(row + k) = p0;
        *(unsigned*)(row + (size_t)N * K + k) = p1;
        *(unsigned*)(row + (size_t)2 * N * K + k) = p2;
    }
    __syncthreads();                                                 // cov / quad are reused by the next source
}

// grid (n_clouds, N / 64)
template <int K1, int K2>
__global__ void __launch_bounds__(256) k_tail_prepare(const double* __restrict__ part1, const double* __restrict__ part2,
                                                      const int* __restrict__ seg_off, int n_chunks, const float* __restrict__ W1,
                                                      const float* __restrict__ W2, int N, float eps, uint16_t* __restrict__ planes1,
                                                      uint16_t* __restrict__ planes2, float* __restrict__ mean1,
                                                      float* __restrict__ mean2, float2* __restrict__ out_stats, int n_clouds,
                                                      const float* __restrict__ pivot1, const float* __restrict__ pivot2)
{
    constexpr int KM = K1 > K2 ? K1 : K2;
    __shared__ double cov[KM * KM + KM];
    __shared__ double2 quad[256];
    const int cloud = blockIdx.x, col0 = blockIdx.y * 64;
    const int n = seg_off[cloud + 1] - seg_off[cloud];
    if (n <= 0) return;                                              // no row of this cloud reaches k_tail_strip
    const int n_valid = (n + MO_ROWS_WG - 1) / MO_ROWS_WG;
    bt_prepare_source<K1>(part1, n_chunks, n_valid, n, cloud, W1, N, col0, eps, cov, quad, planes1, mean1, out_stats, pivot1);
    if constexpr (K2 > 0)
        bt_prepare_source<K2>(part2, n_chunks, n_valid, n, cloud, W2, N, col0, eps, cov, quad, planes2, mean2,
                              out_stats ? out_stats + (size_t)n_clouds * N : nullptr, pivot2);
}

// ------------------------------------------------------------------------------------------------------------------ the strip
constexpr int TS_WAVES = 8;
constexpr int TS_ROWS = 32 * TS_WAVES;

struct TailArgs {
    const float* A1; const float* A2; float* Y;
    const float2* a1_stats;                       // [n_seg, K1] (mean, rstd) of the conv output: A1' = LeakyReLU(InstanceNorm(A1))
    const float* row_div1;                        // optional [M]: A1' = A1 / row_div1[row] (when there is no fold)
    const float* mean1; const float* mean2;       // [n_seg, K1] / [n_seg, K2]: per-cloud means of A1' / A2 (the operands are centred)
    const uint16_t* planes1; const uint16_t* planes2;   // [n_seg][3][N][K1] / [..][K2] bf16: weights x rstd of the product's column
    const int* seg_off; const int4* tile_info;
    int M, N, lda1, lda2, ldy, n_seg;
    float a1_slope, slope;
    const float* R; const float2* r_stats; int ldr;   // RES: + R[row, col] (identity shortcut) or + InstanceNorm(R) by r_stats [n_seg, N] (Linear shortcut)
};

// chunk swizzle of a K-bf16 LDS row (see gemm_stream.hip): CPR = K / 8 sixteen-byte chunks
template <int KT> __device__ __forceinline__ unsigned ts_swz(unsigned n)
{
    constexpr int CPR = 2 * KT;
    return CPR == 2 ? 0u : (CPR == 4 ? ((n >> 2) & 3u) : (CPR == 8 ? ((n >> 1) & 7u) : (n & 15u)));   // CPR 2: conflict free as is
}

template <int KT, int NB>
__device__ __forceinline__ void ts_copy_planes(const uint16_t* __restrict__ src /*[3][N][K] of the cloud, at column n0*/, size_t plane,
                                               unsigned char* dst, int t)
{
    // LDS-DMA, as k_gemm_strip's weight copy (csrc/gemm_stream.hip): lane i of DMA q is slot 64 q + i = (p NB + n) CPR + cs of the image and
    // fetches logical chunk cs ^ swz(n) of column n, plane p.  (The load -> ds_write loop this replaces was up to nine dependent round
    // trips per thread, and the kernel has no registers to batch them in.)
    constexpr int CPR = 2 * KT, K = 16 * KT, NDMA = 3 * NB * CPR / RG_WAVE;
    static_assert(3 * NB * CPR % RG_WAVE == 0, "whole DMA instructions");
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
#pragma unroll 1                 // (rolled: a DMA returns nothing to wait for, and the addresses of nine of them at once would spill)
    for (int q0 = 0; q0 < NDMA; q0 += TS_WAVES) {
        const int q = q0 + wave;
        if (q < NDMA) {                                              // wave-uniform
            const int sidx = q * RG_WAVE + lane;
            const int p = sidx / (NB * CPR), rem = sidx - p * (NB * CPR), n = rem / CPR, cs = rem - n * CPR;
            __builtin_amdgcn_global_load_lds((const void*)(src + (size_t)p * plane + (size_t)n * K + (((unsigned)cs ^ ts_swz<KT>((unsigned)n)) * 8)),
                                             (lds_ptr)(dst + (size_t)q * 1024), 16, 0, 0);
        }
    }
}

// NT 32-column accumulators at a time, NPASS passes over the workgroup's NB = 32 NT NPASS columns: the rows stay in registers (raw),
// fragments are remade per pass -- with all 128 columns' accumulators live the kernel needs 228 registers = one workgroup per CU
// KT2 == 0: one source only (SimpleBlock: KPConv -> InstanceNorm -> LeakyReLU); FOLD1: source 1 is LeakyReLU(InstanceNorm(A1))
// RES: the second operand of the block's sum is not a product computed here but a finished [M, N] array -- the block input itself
// (identity shortcut, kpconv_blocks.py:736-739) or a shortcut product with its own statistics -- added in the epilogue.
template <int KT1, int KT2, int NT, int NPASS, bool FOLD1, bool RES = false>
__global__ void __launch_bounds__(TS_WAVES* RG_WAVE, NT <= 2 ? 4 : 2) k_tail_strip(TailArgs g)
{
    constexpr int K1 = 16 * KT1, K2 = 16 * KT2, NB = 32 * NT * NPASS, KT2A = KT2 > 0 ? KT2 : 1;
    extern __shared__ __align__(16) unsigned char Ws[];              // [3][NB][K1] | [3][NB][K2]  bf16, chunk-swizzled
    unsigned char* Ws2 = Ws + (size_t)3 * NB * K1 * 2;
    const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int N = g.N, n0 = blockIdx.y * NB;
    const int m0 = blockIdx.x * TS_ROWS, r0 = m0 + 32 * wave;
    const int row = r0 + l31;
    const bool row_ok = row < g.M;
    const int rc = row_ok ? row : g.M - 1;

    float4 raw1[KT1][2], raw2[KT2A][2];
    float div1 = 1.f;
    {
        const float* ap = g.A1 + (size_t)rc * g.lda1 + 8 * hi;
#pragma unroll
        for (int ks = 0; ks < KT1; ks++) { raw1[ks][0] = *(const float4*)(ap + 16 * ks); raw1[ks][1] = *(const float4*)(ap + 16 * ks + 4); }
        if constexpr (KT2 > 0) {
            const float* bp = g.A2 + (size_t)rc * g.lda2 + 8 * hi;
#pragma unroll
            for (int ks = 0; ks < KT2; ks++) { raw2[ks][0] = *(const float4*)(bp + 16 * ks); raw2[ks][1] = *(const float4*)(bp + 16 * ks + 4); }
        }
        if (!FOLD1 && g.row_div1) div1 = g.row_div1[rc];
    }
    const int4 ti = g.tile_info[blockIdx.x];
    const int s_lo = ti.x, s_hi = ti.y;
    ts_copy_planes<KT1, NB>(g.planes1 + ((size_t)s_lo * 3 * N + n0) * K1, (size_t)N * K1, Ws, t);
    if constexpr (KT2 > 0) ts_copy_planes<KT2, NB>(g.planes2 + ((size_t)s_lo * 3 * N + n0) * K2, (size_t)N * K2, Ws2, t);
    const int my_seg = s_lo == s_hi ? s_lo : rg_find_segment(g.seg_off, g.n_seg, rc);
    __syncthreads();

    // ---- fold + centre once, in place (the statistics / mean loads must not sit in the pass loop: hoisted out of it by the compiler
    // they cost 80 registers); the bf16 split is redone just in time per k-step and pass (holding all fragments = 72 registers)
    {
        const float* mp = g.mean1 + (size_t)my_seg * K1 + 8 * hi;
        if constexpr (FOLD1) {
            const float2* sp = g.a1_stats + (size_t)my_seg * K1 + 8 * hi;
#pragma unroll
            for (int ks = 0; ks < KT1; ks++)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const float4 sa = *(const float4*)(sp + 16 * ks + 4 * h), sb = *(const float4*)(sp + 16 * ks + 4 * h + 2);
                    const float4 m = *(const float4*)(mp + 16 * ks + 4 * h);
                    float4 v = raw1[ks][h];
                    float u;
                    u = (v.x - sa.x) * sa.y; v.x = row_ok ? fmaxf(u, u * g.a1_slope) - m.x : 0.f;
                    u = (v.y - sa.z) * sa.w; v.y = row_ok ? fmaxf(u, u * g.a1_slope) - m.y : 0.f;
                    u = (v.z - sb.x) * sb.y; v.z = row_ok ? fmaxf(u, u * g.a1_slope) - m.z : 0.f;
                    u = (v.w - sb.z) * sb.w; v.w = row_ok ? fmaxf(u, u * g.a1_slope) - m.w : 0.f;
                    raw1[ks][h] = v;
                }
        } else {
#pragma unroll
            for (int ks = 0; ks < KT1; ks++)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const float4 m = *(const float4*)(mp + 16 * ks + 4 * h);
                    float4 v = raw1[ks][h];
                    if (g.row_div1) { v.x = v.x / div1; v.y = v.y / div1; v.z = v.z / div1; v.w = v.w / div1; }   // as k_moments does
                    v.x = row_ok ? v.x - m.x : 0.f; v.y = row_ok ? v.y - m.y : 0.f; v.z = row_ok ? v.z - m.z : 0.f; v.w = row_ok ? v.w - m.w : 0.f;
                    raw1[ks][h] = v;
                }
        }
        if constexpr (KT2 > 0) {
            const float* mq = g.mean2 + (size_t)my_seg * K2 + 8 * hi;
#pragma unroll
            for (int ks = 0; ks < KT2; ks++)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const float4 m = *(const float4*)(mq + 16 * ks + 4 * h);
                    float4 v = raw2[ks][h];
                    v.x = row_ok ? v.x - m.x : 0.f; v.y = row_ok ? v.y - m.y : 0.f; v.z = row_ok ? v.z - m.z : 0.f; v.w = row_ok ? v.w - m.w : 0.f;
                    raw2[ks][h] = v;
                }
        }
    }
    auto frag = [&](const float4 (&r)[2], bf16x8 (&fa)[3]) {
        unsigned w[4][3];
        bt_split2(r[0].x, r[0].y, w[0][0], w[0][1], w[0][2]);
        bt_split2(r[0].z, r[0].w, w[1][0], w[1][1], w[1][2]);
        bt_split2(r[1].x, r[1].y, w[2][0], w[2][1], w[2][2]);
        bt_split2(r[1].z, r[1].w, w[3][0], w[3][1], w[3][2]);
#pragma unroll
        for (int p = 0; p < 3; p++) fa[p] = __builtin_bit_cast(bf16x8, make_uint4(w[0][p], w[1][p], w[2][p], w[3][p]));
    };
    unsigned f1_off[KT1], f2_off[KT2A];
#pragma unroll
    for (int ks = 0; ks < KT1; ks++) f1_off[ks] = (unsigned)l31 * (K1 * 2) + (((unsigned)(2 * ks + hi)) ^ ts_swz<KT1>((unsigned)l31)) * 16u;
    if constexpr (KT2 > 0) {
#pragma unroll
        for (int ks = 0; ks < KT2; ks++) f2_off[ks] = (unsigned)l31 * (K2 * 2) + (((unsigned)(2 * ks + hi)) ^ ts_swz<KT2A>((unsigned)l31)) * 16u;
    }

    const int rbase = r0 + 4 * hi;
    for (int sg = s_lo; sg <= s_hi; sg++) {                          // workgroup-uniform; one pass almost always
        if (sg > s_lo) {                                             // a straddling tile: the next cloud's planes
            __syncthreads();
            ts_copy_planes<KT1, NB>(g.planes1 + ((size_t)sg * 3 * N + n0) * K1, (size_t)N * K1, Ws, t);
            if constexpr (KT2 > 0) ts_copy_planes<KT2, NB>(g.planes2 + ((size_t)sg * 3 * N + n0) * K2, (size_t)N * K2, Ws2, t);
            __syncthreads();
        }
        const int c_lo = sg == s_lo ? ti.z : g.seg_off[sg], c_hi = min(sg == s_lo ? ti.w : g.seg_off[sg + 1], g.M);
#pragma unroll 1
        for (int ps = 0; ps < NPASS; ps++) {
            const int cb = 32 * NT * ps;                             // first column of the pass within the workgroup's NB
            floatx16 acc[NT];
#pragma unroll
            for (int j = 0; j < NT; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
            // RES: the pass's residual values are requested BEFORE its MFMAs, branch free (clamped rows; a predicated load in the store
            // loop would be waited for one at a time): they arrive under the matrix work
            float rv[RES ? NT : 1][RES ? 16 : 1];
            if constexpr (RES) {
#pragma unroll
                for (int j = 0; j < NT; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int rw = min(rbase + (r & 3) + 8 * (r >> 2), g.M - 1);
                        rv[j][r] = g.R[(size_t)rw * g.ldr + n0 + cb + 32 * j + l31];
                    }
            }
#define TS_STEP(FA, WS_, OFF, KK)                                                                                            \
            _Pragma("unroll") for (int j = 0; j < NT; j++) {                                                                   \
                bf16x8 fb[3];                                                                                                  \
                _Pragma("unroll") for (int p = 0; p < 3; p++)                                                                  \
                    fb[p] = __builtin_bit_cast(bf16x8, *(const uint4*)(WS_ + (size_t)(p * NB + cb + 32 * j) * (KK * 2) + OFF)); \
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[2], fb[0], acc[j], 0, 0, 0);                               \
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[1], fb[1], acc[j], 0, 0, 0);                               \
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0], fb[2], acc[j], 0, 0, 0);                               \
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[1], fb[0], acc[j], 0, 0, 0);                               \
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0], fb[1], acc[j], 0, 0, 0);                               \
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0], fb[0], acc[j], 0, 0, 0);                               \
            }
#pragma unroll
            for (int ks = 0; ks < KT1; ks++) { bf16x8 fa[3]; frag(raw1[ks], fa); TS_STEP(fa, Ws, f1_off[ks], K1) }
            if constexpr (KT2 > 0) {
#pragma unroll
                for (int ks = 0; ks < KT2; ks++) { bf16x8 fa[3]; frag(raw2[ks], fa); TS_STEP(fa, Ws2, f2_off[ks], K2) }
            }
#undef TS_STEP
            // opaque per pass: otherwise the 16 row addresses + predicates are hoisted out of the pass loop (48 registers, spilled)
            int rb = rbase;
            asm volatile("" : "+v"(rb));
#pragma unroll
            for (int j = 0; j < NT; j++) {
                const int col = n0 + cb + 32 * j + l31;
                float2 rst = make_float2(0.f, 1.f);
                if constexpr (RES) { if (g.r_stats) rst = g.r_stats[(size_t)sg * N + col]; }
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int rw = rb + (r & 3) + 8 * (r >> 2);
                    float v = acc[j][r];
                    if (rw >= c_lo && rw < c_hi) {
                        if constexpr (RES) v += (rv[j][r] - rst.x) * rst.y;
                        g.Y[(size_t)rw * g.ldy + col] = fmaxf(v, v * g.slope);
                    }
                }
            }
        }
    }
}

int bt_chunks(int max_len) { return rg_cdiv(max_len > 0 ? max_len : 1, MO_ROWS_WG); }
#ifdef REGTR_EXPERIMENTAL
size_t bt_align(size_t b);
size_t bt_res_ws_bytes(int n_clouds, int max_len, int N, int K1)
{
    const size_t nc = (size_t)bt_chunks(max_len);
    return ((n_clouds * nc * (K1 * K1 + K1) * 8 + 255) & ~(size_t)255) + (((size_t)n_clouds * 3 * N * K1 * 2 + 255) & ~(size_t)255) +
           2 * (((size_t)n_clouds * K1 * 4 + 255) & ~(size_t)255);
}
#endif
size_t bt_align(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

extern "C" {

// 1 when regtr_block_tail serves the shape:
//   K1 = 32, K2 = 64, N = 128   the resnet block of level 0 of both shipped encoders (kpconv_blocks.py:649-741, in 64 -> out 128)
//   K1 = 16, K2 = 0,  N = 64    their first block (SimpleBlock, :590-646): KPConv -> InstanceNorm -> LeakyReLU with 15 kernel points x 1
//                               input channel (rows padded to 16 columns) -- one source, no fold, rows divided by the neighbour count
int regtr_block_tail_supported(int M, int N, int K1, int K2)
{
    return (M >= 0 && ((K1 == 32 && K2 == 64 && N == 128) || (K1 == 16 && K2 == 0 && N == 64))) ? 1 : 0;
}

size_t regtr_block_tail_ws_bytes(int n_clouds, int max_len, int N, int K1, int K2)
{
    if (n_clouds < 1 || !regtr_block_tail_supported(0, N, K1, K2)) return 0;
    const size_t nc = (size_t)bt_chunks(max_len);
    return bt_align(n_clouds * nc * (K1 * K1 + K1) * 8) + bt_align(n_clouds * nc * (K2 * K2 + K2) * 8) +
           bt_align((size_t)n_clouds * 3 * N * K1 * 2) + bt_align((size_t)n_clouds * 3 * N * K2 * 2) +
           2 * (bt_align((size_t)n_clouds * K1 * 4) + bt_align((size_t)n_clouds * K2 * 4));
}

// Y[M, N] = LeakyReLU_slope( InstanceNorm(A1' W1) [+ InstanceNorm(A2 W2)] )
//   A1 [M, K1] (lda1); A1' = LeakyReLU_a1_slope(InstanceNorm(A1)) by a1_stats [n_clouds, K1, 2] when given, else A1 / row_div1[row]
//   (row_div1 optional [M]);  A2 [M, K2] (lda2) and W2 only when K2 > 0;  W1 [K1, N], W2 [K2, N] float32 row-major (k, n)
//   seg_off [n_clouds + 1] cloud offsets of the rows, max_len = longest cloud, tile_info = regtr_tile_segments(seg_off, .., M, 256, ..)
//   out_stats (optional) [1 or 2, n_clouds, N, 2]: (mean, rstd) of the products, as InstanceNorm would report them
int regtr_block_tail(const float* A1, int lda1, const float* a1_stats, float a1_slope, const float* row_div1, const float* A2, int lda2,
                     const float* W1, const float* W2, const int* seg_off, int n_clouds, int max_len, const void* tile_info,
                     int M, int N, int K1, int K2, float eps, float slope, float* Y, int ldy, void* ws, size_t ws_bytes,
                     float* out_stats, void* stream)
{
    if (!A1 || !W1 || !seg_off || !tile_info || !Y || !ws || n_clouds < 1 || max_len < 0) return RG_ERR_ARG;
    if (!regtr_block_tail_supported(M, N, K1, K2) || lda1 < K1 || ldy < N || (lda1 % 4)) return RG_ERR_ARG;
    if (K2 > 0 && (!A2 || !W2 || !a1_stats || row_div1 || lda2 < K2 || (lda2 % 4) || (uintptr_t)A2 % 16)) return RG_ERR_ARG;
    if (K2 == 0 && (A2 || W2 || a1_stats || !row_div1)) return RG_ERR_ARG;
    if (((uintptr_t)A1 | (uintptr_t)a1_stats | (uintptr_t)tile_info | (uintptr_t)ws) % 16) return RG_ERR_ARG;
    if (ws_bytes < regtr_block_tail_ws_bytes(n_clouds, max_len, N, K1, K2)) return RG_ERR_WORKSPACE;
    if (M == 0 || max_len == 0) return RG_OK;
    const int nc = bt_chunks(max_len);
    unsigned char* p = (unsigned char*)ws;
    double* part1 = (double*)p; p += bt_align((size_t)n_clouds * nc * (K1 * K1 + K1) * 8);
    double* part2 = (double*)p; p += bt_align((size_t)n_clouds * nc * (K2 * K2 + K2) * 8);
    uint16_t* planes1 = (uint16_t*)p; p += bt_align((size_t)n_clouds * 3 * N * K1 * 2);
    uint16_t* planes2 = (uint16_t*)p; p += bt_align((size_t)n_clouds * 3 * N * K2 * 2);
    float* mean1 = (float*)p; p += bt_align((size_t)n_clouds * K1 * 4);
    float* mean2 = (float*)p; p += bt_align((size_t)n_clouds * K2 * 4);
    float* pivot1 = (float*)p; p += bt_align((size_t)n_clouds * K1 * 4);
    float* pivot2 = (float*)p;
    hipStream_t st = (hipStream_t)stream;
    const dim3 mgrid(nc, n_clouds), pgrid(n_clouds, N / 64), sgrid(rg_cdiv(M, TS_ROWS), 1);
    TailArgs g{A1, A2, Y, (const float2*)a1_stats, row_div1, mean1, mean2, planes1, planes2, seg_off, (const int4*)tile_info,
               M, N, lda1, lda2, ldy, n_clouds, a1_slope, slope, nullptr, nullptr, 0};
    const size_t lds = (size_t)3 * N * (K1 + K2) * 2;                // the workgroup holds all N columns
    if (K2 > 0) {
        k_moments<1, true, false><<<mgrid, MO_WAVES * RG_WAVE, 0, st>>>(A1, lda1, (const float2*)a1_stats, a1_slope, nullptr, K1, seg_off, nc, part1, pivot1);
        k_moments<2, false, false><<<mgrid, MO_WAVES * RG_WAVE, 0, st>>>(A2, lda2, nullptr, 0.f, nullptr, K2, seg_off, nc, part2, pivot2);
        k_tail_prepare<32, 64><<<pgrid, 256, 0, st>>>(part1, part2, seg_off, nc, W1, W2, N, eps, planes1, planes2, mean1, mean2,
                                                      (float2*)out_stats, n_clouds, pivot1, pivot2);
        if (!rg_allow_dynamic_lds<k_tail_strip<2, 4, 2, 2, true>>(lds)) return RG_ERR_ARG;
        k_tail_strip<2, 4, 2, 2, true><<<sgrid, TS_WAVES * RG_WAVE, lds, st>>>(g);
    } else {
        k_moments<1, false, true><<<mgrid, MO_WAVES * RG_WAVE, 0, st>>>(A1, lda1, nullptr, 0.f, row_div1, K1, seg_off, nc, part1, pivot1);
        k_tail_prepare<16, 0><<<pgrid, 256, 0, st>>>(part1, nullptr, seg_off, nc, W1, nullptr, N, eps, planes1, nullptr, mean1, nullptr,
                                                     (float2*)out_stats, n_clouds, pivot1, nullptr);
        if (!rg_allow_dynamic_lds<k_tail_strip<1, 0, 2, 1, false>>(lds)) return RG_ERR_ARG;
        k_tail_strip<1, 0, 2, 1, false><<<sgrid, TS_WAVES * RG_WAVE, lds, st>>>(g);
    }
    RG_RETURN_IF_LAUNCH_FAILED();
    return RG_OK;
}

#ifdef REGTR_EXPERIMENTAL      // measured slower (docs/NEGATIVES.md): experiment variant only
// The tail of a resnet block whose second summand already exists (csrc/block_tail.hip, RES):
//     Y = LeakyReLU_slope( InstanceNorm(A1' W1) + R )                      r_stats == NULL: identity shortcut (kpconv_blocks.py:736-741)
//     Y = LeakyReLU_slope( InstanceNorm(A1' W1) + InstanceNorm_r_stats(R) ) r_stats [n_clouds, N, 2]: a Linear shortcut's product
// with A1' = LeakyReLU_a1_slope(InstanceNorm(A1)) by a1_stats.  The product A1' W1 is never written: its statistics come from the K1 x K1
// second moments of A1', the normalisation is folded into per-cloud weight planes, R is added in the strip GEMM's epilogue -- one pass
// (read A1, R; write Y) instead of GEMM + statistics + normalise-add pass.  Shape served: K1 = 64, N a multiple of 64 (level 1 of the
// 3DMatch encoder: 64 -> 256).  ws: regtr_block_tail_res_ws_bytes.
int regtr_block_tail_res_supported(int M, int N, int K1) { return (M >= 0 && K1 == 64 && N >= 64 && N % 64 == 0) ? 1 : 0; }

size_t regtr_block_tail_res_ws_bytes(int n_clouds, int max_len, int N, int K1)
{
    if (n_clouds < 1 || !regtr_block_tail_res_supported(0, N, K1)) return 0;
    return bt_res_ws_bytes(n_clouds, max_len, N, K1);
}

int regtr_block_tail_res(const float* A1, int lda1, const float* a1_stats, float a1_slope, const float* W1, const float* R, int ldr,
                         const float* r_stats, const int* seg_off, int n_clouds, int max_len, const void* tile_info, int M, int N, int K1,
                         float eps, float slope, float* Y, int ldy, void* ws, size_t ws_bytes, void* stream)
{
    if (!A1 || !a1_stats || !W1 || !R || !seg_off || !tile_info || !Y || !ws || n_clouds < 1 || max_len < 0) return RG_ERR_ARG;
    if (!regtr_block_tail_res_supported(M, N, K1) || lda1 < K1 || ldy < N || ldr < N || (lda1 % 4)) return RG_ERR_ARG;
    if (((uintptr_t)A1 | (uintptr_t)a1_stats | (uintptr_t)tile_info | (uintptr_t)ws) % 16) return RG_ERR_ARG;
    if (ws_bytes < bt_res_ws_bytes(n_clouds, max_len, N, K1)) return RG_ERR_WORKSPACE;
    if (M == 0 || max_len == 0) return RG_OK;
    const int nc = bt_chunks(max_len);
    unsigned char* p = (unsigned char*)ws;
    double* part1 = (double*)p; p += bt_align((size_t)n_clouds * nc * (K1 * K1 + K1) * 8);
    uint16_t* planes1 = (uint16_t*)p; p += bt_align((size_t)n_clouds * 3 * N * K1 * 2);
    float* mean1 = (float*)p; p += bt_align((size_t)n_clouds * K1 * 4);
    float* pivot1 = (float*)p;
    hipStream_t st = (hipStream_t)stream;
    const bool wide = N % 256 == 0;                                  // a workgroup holds 256 (else 64) columns of the cloud's planes: A is read once
    const int nbw = wide ? 256 : 64;
    const dim3 mgrid(nc, n_clouds), pgrid(n_clouds, N / 64), sgrid(rg_cdiv(M, TS_ROWS), N / nbw);
    TailArgs g{A1, nullptr, Y, (const float2*)a1_stats, nullptr, mean1, nullptr, planes1, nullptr, seg_off, (const int4*)tile_info,
               M, N, lda1, 0, ldy, n_clouds, a1_slope, slope, R, (const float2*)r_stats, ldr};
    const size_t lds = (size_t)3 * nbw * K1 * 2;
    k_moments<2, true, false><<<mgrid, MO_WAVES * RG_WAVE, 0, st>>>(A1, lda1, (const float2*)a1_stats, a1_slope, nullptr, K1, seg_off, nc, part1, pivot1);
    k_tail_prepare<64, 0><<<pgrid, 256, 0, st>>>(part1, nullptr, seg_off, nc, W1, nullptr, N, eps, planes1, nullptr, mean1, nullptr, nullptr, n_clouds,
                                                 pivot1, nullptr);
    if (wide) {
        if (!rg_allow_dynamic_lds<k_tail_strip<4, 0, 4, 2, true, true>>(lds)) return RG_ERR_ARG;
        k_tail_strip<4, 0, 4, 2, true, true><<<sgrid, TS_WAVES * RG_WAVE, lds, st>>>(g);
    } else {
        k_tail_strip<4, 0, 2, 1, true, true><<<sgrid, TS_WAVES * RG_WAVE, lds, st>>>(g);
    }
    RG_RETURN_IF_LAUNCH_FAILED();
    return RG_OK;
}

#endif  // REGTR_EXPERIMENTAL

}  // extern "C"
