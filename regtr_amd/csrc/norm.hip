// Normalisation / elementwise kernels of the RegTR hot path for gfx950 (all HBM-bound, float4 accesses):
//   * per-cloud InstanceNorm (+ LeakyReLU 0.1, + residual) -- the reference's BatchNormBlock with
//     nn.InstanceNorm1d: per cloud, per channel, biased variance, eps 1e-5, no affine
//     (/root/reference/src/models/backbone_kpconv/kpconv_blocks.py:489,510-519,556-561,741)
//   * LayerNorm (+ positional embedding add)   (models/transformer/transformers.py:194-195,213-215,232)
//   * sine positional embedding                (models/transformer/position_embedding.py:29-50)
// Statistics are accumulated in float64 with a fixed reduction tree, so results are run-to-run deterministic.
#include "common.h"

namespace {

constexpr int IN_ROWS = 128;   // rows of one cloud handled by one workgroup (large launches)

// Rows per workgroup of a launch over clouds of at most max_len rows.  128 everywhere a launch fills the chip anyway; a pair or two per forward
// leaves the deep levels with a handful of 128-row workgroups (751 rows x 1024 channels: 8 workgroups, each thread walking 128 rows -- 35 us
// for 9 MB), so the chunk is halved until the launch has ~1000 workgroups or a workgroup is down to four row steps / 8 rows.  Host-side
// and a function of the launch geometry only (never of the data); the statistics kernels add a cloud's chunks in chunk order whatever their
// length, so results stay run-to-run deterministic.
inline int in_rows(int max_len, int n_clouds, int C)
{
    const int TR = 256 / (C >> 2);
    int rows = IN_ROWS;
    while (rows > 4 * TR && rows > 8 && (long long)rg_cdiv(max_len > 0 ? max_len : 1, rows) * n_clouds < 1024) rows >>= 1;
    return rows;
}

// Thread mapping shared by the InstanceNorm kernels: C4 = C/4 float4 columns (a power of two <= 256); thread
// (tx = t % C4, ty = t / C4) owns column group tx for rows ty, ty + TR, ... so that every row is one contiguous read and
// the per-channel statistics a thread needs stay in registers for the whole tile.

// partial[(cloud * nchunk + chunk) * C + c] = (sum, sumsq) over the chunk's rows, accumulated in float64
__global__ void __launch_bounds__(256) k_instnorm_partial(const float* __restrict__ x, const int* __restrict__ seg_off, int C,
                                                          int nchunk, int rows, double2* __restrict__ partial)
{
    __shared__ double sh[256 * 8];
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int r0 = seg_off[b] + chunk * rows, r1 = min(seg_off[b + 1], r0 + rows);
    if (r0 >= r1) return;
    const int C4 = C >> 2, TR = 256 / C4;
    const int tx = threadIdx.x % C4, ty = threadIdx.x / C4;
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    const float* base = x + 4 * tx;
    auto add = [&](const float4& v) {
        s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
        ss[0] += (double)v.x * v.x; ss[1] += (double)v.y * v.y; ss[2] += (double)v.z * v.z; ss[3] += (double)v.w * v.w;
    };
    // full groups of four rows with four unconditional loads in flight, then the tail row by row: the same rows in the same order
    // as one guarded loop, but a guarded load (`rr < r1 ? load : 0`, even from a clamped row) is compiled to branch + load + wait
    int r = r0 + ty;
    for (; r + 3 * TR < r1; r += 4 * TR) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = *(const float4*)(base + (size_t)(r + u * TR) * C);
#pragma unroll
        for (int u = 0; u < 4; u++) add(v[u]);
    }
    for (; r < r1; r += TR) add(*(const float4*)(base + (size_t)r * C));
#pragma unroll
    for (int j = 0; j < 4; j++) { sh[threadIdx.x * 8 + j] = s[j]; sh[threadIdx.x * 8 + 4 + j] = ss[j]; }
    __syncthreads();
    if (ty == 0) {
        for (int y = 1; y < TR; y++)
#pragma unroll
            for (int j = 0; j < 4; j++) { s[j] += sh[(y * C4 + tx) * 8 + j]; ss[j] += sh[(y * C4 + tx) * 8 + 4 + j]; }
        double2* o = partial + ((size_t)b * nchunk + chunk) * C + 4 * tx;
#pragma unroll
        for (int j = 0; j < 4; j++) o[j] = make_double2(s[j], ss[j]);
    }
}

// stats[(cloud * C + c)] = (mean, 1/sqrt(var + eps)); one wave per (cloud, channel) sums the chunks in a fixed tree
__global__ void __launch_bounds__(256) k_instnorm_finalize(const double2* __restrict__ partial, const int* __restrict__ seg_off,
                                                           int C, int nchunk, int rows, float eps, float2* __restrict__ stats)
{
    const int b = blockIdx.y, c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (c >= C) return;
    const int lane = rg_lane();
    const int n = seg_off[b + 1] - seg_off[b];
    const int used = (n + rows - 1) / rows;
    double s = 0, ss = 0;
    for (int k = lane; k < used; k += RG_WAVE) {
        const double2 p = partial[((size_t)b * nchunk + k) * C + c];
        s += p.x; ss += p.y;
    }
    s = rg_wave_sum(s); ss = rg_wave_sum(ss);
    if (lane == 0) {
        float mean = 0.f, rstd = 0.f;
        if (n > 0) {
            const double m = s / n;
            double var = ss / n - m * m;
            if (var < 0) var = 0;
            mean = (float)m;
            rstd = (float)(1.0 / sqrt(var + (double)eps));
        }
        stats[(size_t)b * C + c] = make_float2(mean, rstd);
    }
}

// Same, from the per-(row tile, cloud) partial sums a GEMM epilogue wrote (gemm_x3.hip): the rows of cloud b live in
// tiles off[b] / R .. (off[b+1] - 1) / R, tile t's share of cloud b is slot t + b.  One THREAD per (cloud, channel), channels
// across the lanes (coalesced 16-byte loads), the cloud's few slots (3 ... 75) added in slot order: a wave per (cloud, channel)
// with a shuffle tree spent its time on launch geometry -- 131 072 one-load waves for the 1024-channel level (37 us; now 6).
__global__ void __launch_bounds__(256) k_instnorm_finalize_tiles(const double2* __restrict__ partial, const int* __restrict__ seg_off,
                                                                 int C, int tile_rows, float eps, float2* __restrict__ stats)
{
    const int b = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int r0 = seg_off[b], r1 = seg_off[b + 1], n = r1 - r0;
    float mean = 0.f, rstd = 0.f;
    if (n > 0) {
        const int t0 = r0 / tile_rows, t1 = (r1 - 1) / tile_rows;
        const double2* p = partial + (size_t)(t0 + b) * C + c;
        double s = 0, ss = 0;
        int t = t0;
        for (; t + 3 <= t1; t += 4) {                      // four independent loads in flight, added in slot order
            const double2 a0 = p[0], a1 = p[(size_t)C], a2 = p[2 * (size_t)C], a3 = p[3 * (size_t)C];
            s = (((s + a0.x) + a1.x) + a2.x) + a3.x; ss = (((ss + a0.y) + a1.y) + a2.y) + a3.y;
            p += 4 * (size_t)C;
        }
        for (; t <= t1; t++) { const double2 a = *p; s += a.x; ss += a.y; p += C; }
        const double m = s / n;
        double var = ss / n - m * m;
        if (var < 0) var = 0;
        mean = (float)m;
        rstd = (float)(1.0 / sqrt(var + (double)eps));
    }
    stats[(size_t)b * C + c] = make_float2(mean, rstd);
}

// The same table for SMALL cloud sets (a pair or two per forward): one WAVE per (cloud, channel), the cloud's slots dealt to the
// lanes and added by a fixed shuffle tree.  With two clouds the thread-per-channel form above is a handful of workgroups walking
// up to ~300 slots one dependent load after another (31 us for a 20k-point cloud at 128-row tiles, most of a one-pair forward's
// InstanceNorm time); here that is five loads per lane.  Deterministic like the other form, but a different summation order: which
// of the two runs depends on the launch geometry only (n_clouds * C), never on the data.
__global__ void __launch_bounds__(256) k_instnorm_finalize_tiles_wave(const double2* __restrict__ partial, const int* __restrict__ seg_off,
                                                                      int C, int tile_rows, float eps, float2* __restrict__ stats)
{
    const int b = blockIdx.y, c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (c >= C) return;
    const int lane = rg_lane();
    const int r0 = seg_off[b], r1 = seg_off[b + 1], n = r1 - r0;
    float mean = 0.f, rstd = 0.f;
    if (n > 0) {                                                      // (wave-uniform)
        const int t0 = r0 / tile_rows, t1 = (r1 - 1) / tile_rows;
        double s = 0, ss = 0;
        for (int t = t0 + lane; t <= t1; t += RG_WAVE) {
            const double2 a = partial[(size_t)(t + b) * C + c];
            s += a.x; ss += a.y;
        }
        s = rg_wave_sum(s); ss = rg_wave_sum(ss);
        const double m = s / n;
        double var = ss / n - m * m;
        if (var < 0) var = 0;
        mean = (float)m;
        rstd = (float)(1.0 / sqrt(var + (double)eps));
    }
    if (lane == 0) stats[(size_t)b * C + c] = make_float2(mean, rstd);
}

// y = act( norm(x) [+ (res_stats ? norm(res) : res)] ) ; act: 0 none, 1 LeakyReLU(slope)
// development A/B (REGTR_VARIANT_FLAGS=-DIN_APPLY_NT=1): non-temporal loads / stores for the streamed tensors of the apply pass
#ifndef IN_APPLY_NT
#define IN_APPLY_NT 0
#endif
typedef float in_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 IN_LD4(const float* p)
{
#if IN_APPLY_NT
    const in_f4 v = __builtin_nontemporal_load((const in_f4*)p);
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *(const float4*)p;
#endif
}
__device__ __forceinline__ void IN_ST4(float* p, float4 v)
{
#if IN_APPLY_NT
    __builtin_nontemporal_store(in_f4{v.x, v.y, v.z, v.w}, (in_f4*)p);
#else
    *(float4*)p = v;
#endif
}

__global__ void __launch_bounds__(256) k_instnorm_apply(const float* __restrict__ x, const int* __restrict__ seg_off, int C,
                                                        const float2* __restrict__ stats, const float* __restrict__ res,
                                                        const float2* __restrict__ res_stats, int act, float slope,
                                                        float* __restrict__ y, const float* __restrict__ row_xyz,
                                                        float* __restrict__ row_positive, int rows)
{
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int r0 = seg_off[b] + chunk * rows, r1 = min(seg_off[b + 1], r0 + rows);
    if (r0 >= r1) return;
    const int C4 = C >> 2, TR = 256 / C4;
    const int tx = threadIdx.x % C4, ty = threadIdx.x / C4;
    float mu[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {1.f, 1.f, 1.f, 1.f}, rmu[4] = {0.f, 0.f, 0.f, 0.f}, rrs[4] = {1.f, 1.f, 1.f, 1.f};
    if (stats) {
        const float4* st = (const float4*)(stats + (size_t)b * C + 4 * tx);              // (mean, rstd) of channels 4 tx .. 4 tx + 3
        const float4 a = st[0], c = st[1];
        mu[0] = a.x; rs[0] = a.y; mu[1] = a.z; rs[1] = a.w; mu[2] = c.x; rs[2] = c.y; mu[3] = c.z; rs[3] = c.w;
    }
    if (res_stats) {
        const float4* st = (const float4*)(res_stats + (size_t)b * C + 4 * tx);
        const float4 a = st[0], c = st[1];
        rmu[0] = a.x; rrs[0] = a.y; rmu[1] = a.z; rrs[1] = a.w; rmu[2] = c.x; rrs[2] = c.y; rmu[3] = c.z; rrs[3] = c.w;
    }
    for (int r = r0 + ty; r < r1; r += 4 * TR) {
        // Loads from CLAMPED rows, unconditional: under the per-lane `rr < r1` predicate every load was followed by its own
        // s_waitcnt vmcnt(0) (tools/isa_scan.py) -- four to eight serial round trips per iteration instead of all in flight.
        float4 v[4], rv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = IN_LD4(x + (size_t)min(r + u * TR, r1 - 1) * C + 4 * tx);
        if (res) {
#pragma unroll
            for (int u = 0; u < 4; u++) rv[u] = IN_LD4(res + (size_t)min(r + u * TR, r1 - 1) * C + 4 * tx);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int rr = r + u * TR;
            if (rr >= r1) continue;
            float o[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int j = 0; j < 4; j++) o[j] = (o[j] - mu[j]) * rs[j];
            if (res) {
                const float q[4] = {rv[u].x, rv[u].y, rv[u].z, rv[u].w};
#pragma unroll
                for (int j = 0; j < 4; j++) o[j] += (q[j] - rmu[j]) * rrs[j];
            }
            if (act == 1) {
#pragma unroll
                for (int j = 0; j < 4; j++) o[j] = o[j] > 0.f ? o[j] : o[j] * slope;
            }
            IN_ST4(y + (size_t)rr * C + 4 * tx, make_float4(o[0], o[1], o[2], o[3]));
            if (row_positive) {   // C4 is a power of two <= 64 here: the row's lanes sit in one wave, aligned to C4
                float sum = (o[0] + o[1]) + (o[2] + o[3]);
                for (int m = 1; m < C4; m <<= 1) sum += __shfl_xor(sum, m, RG_WAVE);
                if (tx == 0) {
                    const float f = sum > 0.f ? 1.f : 0.f;
                    if (row_xyz) *(float4*)(row_positive + 4 * (size_t)rr) = make_float4(row_xyz[3 * (size_t)rr], row_xyz[3 * (size_t)rr + 1],
                                                                                         row_xyz[3 * (size_t)rr + 2], f);
                    else row_positive[rr] = f;
                }
            }
        }
    }
}

// one wave per row: y = LN(x) * gamma + beta (+ add)
__global__ void __launch_bounds__(256) k_layernorm(const float* __restrict__ x, int n, int D, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, float eps, const float* __restrict__ add,
                                                   float* __restrict__ y, float* __restrict__ y_plain)
{
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n) return;
    const int lane = rg_lane();
    const float* xr = x + (size_t)row * D;
    float s = 0.f;
    for (int c = lane * 4; c < D; c += RG_WAVE * 4) {
        const float4 v = *(const float4*)(xr + c);
        s += (v.x + v.y) + (v.z + v.w);
    }
    const float mean = rg_wave_sum(s) / (float)D;
    float ss = 0.f;
    for (int c = lane * 4; c < D; c += RG_WAVE * 4) {
        const float4 v = *(const float4*)(xr + c);
        const float a = v.x - mean, b = v.y - mean, cc = v.z - mean, d = v.w - mean;
        ss += (a * a + b * b) + (cc * cc + d * d);
    }
    const float rstd = 1.0f / sqrtf(rg_wave_sum(ss) / (float)D + eps);
    for (int c = lane * 4; c < D; c += RG_WAVE * 4) {
        const float4 v = *(const float4*)(xr + c);
        const float4 gm = *(const float4*)(gamma + c), bt = *(const float4*)(beta + c);
        float4 o = make_float4((v.x - mean) * rstd * gm.x + bt.x, (v.y - mean) * rstd * gm.y + bt.y,
                               (v.z - mean) * rstd * gm.z + bt.z, (v.w - mean) * rstd * gm.w + bt.w);
        if (y_plain) *(float4*)(y_plain + (size_t)row * D + c) = o;
        if (add) {
            const float4 a = *(const float4*)(add + (size_t)row * D + c);
            o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
        }
        *(float4*)(y + (size_t)row * D + c) = o;
    }
}

// pe[i, a*npf + f] = f even ? sin(xyz[i,a]*scale / dim_t[f]) : cos(...), zero padded to d_model
__global__ void __launch_bounds__(256) k_posemb_sine(const float* __restrict__ xyz, int n, int npf, int d_model, float scale,
                                                     const float* __restrict__ dim_t, float* __restrict__ pe)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)n * d_model) return;
    const int i = (int)(e / d_model), d = (int)(e % d_model);
    float v = 0.f;
    if (d < 3 * npf) {
        const int a = d / npf, f = d % npf;
        const float p = __fdiv_rn(__fmul_rn(xyz[3 * (size_t)i + a], scale), dim_t[f]);
        v = (f & 1) ? cosf(p) : sinf(p);
    }
    pe[e] = v;
}

}  // namespace

extern "C" {

size_t regtr_instnorm_ws_bytes(int n_clouds, int max_len, int C)
{
    if (n_clouds < 1 || C < 4 || C % 4 || C > 1024 || 256 % (C / 4)) return 256;
    const size_t nchunk = (size_t)rg_cdiv(max_len > 0 ? max_len : 1, in_rows(max_len, n_clouds, C));
    return nchunk * n_clouds * C * sizeof(double2) + 256;
}

// stats [n_clouds, C, 2] = (mean, rstd) of x [N, C] per cloud segment.  max_len = longest segment (host-known bound).
int regtr_instnorm_stats(const float* x, const int* seg_off, int n_clouds, int max_len, int C, float eps, float* stats,
                         void* ws, size_t ws_bytes, void* stream)
{
    if (!x || !seg_off || !stats || n_clouds < 1 || C < 4 || C % 4 || max_len < 0) return RG_ERR_ARG;
    if (C > 1024 || 256 % (C / 4)) return RG_ERR_ARG;      // C/4 float4 columns must be a power of two <= 256
    if (ws_bytes < regtr_instnorm_ws_bytes(n_clouds, max_len, C)) return RG_ERR_WORKSPACE;
    if (max_len == 0) return RG_OK;
    hipStream_t st = (hipStream_t)stream;
    const int rows = in_rows(max_len, n_clouds, C);
    const int nchunk = rg_cdiv(max_len, rows);
    k_instnorm_partial<<<dim3(nchunk, n_clouds), 256, 0, st>>>(x, seg_off, C, nchunk, rows, (double2*)ws);
    k_instnorm_finalize<<<dim3(rg_cdiv(C, 4), n_clouds), 256, 0, st>>>((const double2*)ws, seg_off, C, nchunk, rows, eps,
                                                                         (float2*)stats);
    RG_RETURN_IF_LAUNCH_FAILED();
    return RG_OK;
}

// stats from the partial sums regtr_gemm_x3 wrote in its epilogue (tile_rows = regtr_gemm_x3_stat_tile_rows of that call)
int regtr_instnorm_finalize_tiles(const double* partial, const int* seg_off, int n_clouds, int C, int tile_rows, float eps,
                                  float* stats, void* stream)
{
    if (!partial || !seg_off || !stats || n_clouds < 1 || C < 1 || tile_rows < 1) return RG_ERR_ARG;
    if ((long long)n_clouds * C <= 4096) {          // a pair or two per forward: few, long slot lists -- a wave per (cloud, channel)
        k_instnorm_finalize_tiles_wave<<<dim3(rg_cdiv(C, 4), n_clouds), 256, 0, (hipStream_t)stream>>>(
            (const double2*)partial, seg_off, C, tile_rows, eps, (float2*)stats);
    } else {
        const int bs = C >= 256 ? 256 : (C >= 128 ? 128 : 64);
        k_instnorm_finalize_tiles<<<dim3(rg_cdiv(C, bs), n_clouds), bs, 0, (hipStream_t)stream>>>(
            (const double2*)partial, seg_off, C, tile_rows, eps, (float2*)stats);
    }
    RG_RETURN_IF_LAUNCH_FAILED();
    return RG_OK;
}

// y = act( (x - mean) * rstd  [+ residual | + (residual - rmean) * rrstd] ), act 1 = LeakyReLU(slope).
// stats may be NULL (x used as is).  y may alias x.  row_positive (optional, C <= 256): [rows] 1.0 where sum_c y[row, c] > 0 --
// the per-support flag of KPConv's neighbour-count normaliser (kpconv_blocks.py:409-410), for regtr_kpconv_gather's `flag`; with
// row_xyz [rows,3] it is written as 16-byte records [rows,4] = (x, y, z, flag), regtr_kpconv_gather's `s_xyzf`.
int regtr_instnorm_apply(const float* x, const int* seg_off, int n_clouds, int max_len, int C, const float* stats,
                         const float* residual, const float* res_stats, int act, float slope, float* y, const float* row_xyz,
                         float* row_positive, void* stream)
{
    if (!x || !seg_off || !y || n_clouds < 1 || C < 4 || C % 4 || max_len < 0) return RG_ERR_ARG;
    if (C > 1024 || 256 % (C / 4) || (row_positive && C > 256) || (row_xyz && (!row_positive || (uintptr_t)row_positive % 16)))
        return RG_ERR_ARG;
    if (max_len == 0) return RG_OK;
    const int rows = in_rows(max_len, n_clouds, C);
    k_instnorm_apply<<<dim3(rg_cdiv(max_len, rows), n_clouds), 256, 0, (hipStream_t)stream>>>(
        x, seg_off, C, (const float2*)stats, residual, (const float2*)res_stats, act, slope, y, row_xyz, row_positive, rows);
    RG_RETURN_IF_LAUNCH_FAILED();
    return RG_OK;
}

// out = a + b (float4 body, scalar tail): with_pos_embed of the post-norm encoder layer (transformers.py:118-119,131,142)
static __global__ void __launch_bounds__(256) k_add(const float* __restrict__ a, const float* __restrict__ b, size_t n, float* __restrict__ out)
{
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        const float4 x = *(const float4*)(a + i), y = *(const float4*)(b + i);
        *(float4*)(out + i) = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    } else {
        for (size_t j = i; j < n; j++) out[j] = a[j] + b[j];
    }
}

// y = LayerNorm(x) * gamma + beta (+ add) ; y_plain (optional) receives the value before `add`.
int regtr_add_f32(const float* a, const float* b, size_t n, float* out, void* stream)
{
    if (!a || !b || !out || (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) % 16)) return RG_ERR_ARG;
    if (n == 0) return RG_OK;
    k_add<<<rg_cdiv((long long)((n + 3) / 4), 256), 256, 0, (hipStream_t)stream>>>(a, b, n, out);
    RG_RETURN_IF_LAUNCH_FAILED();
    return RG_OK;
}

int regtr_layernorm(const float* x, int n, int D, const float* gamma, const float* beta, float eps, const float* add,
                    float* y, float* y_plain, void* stream)
{
    if (!x || !gamma || !beta || !y || n < 0 || D < 4 || D % 4) return RG_ERR_ARG;
    if (n == 0) return RG_OK;
    k_layernorm<<<rg_cdiv(n, 4), 256, 0, (hipStream_t)stream>>>(x, n, D, gamma, beta, eps, add, y, y_plain);
    RG_RETURN_IF_LAUNCH_FAILED();
    return RG_OK;
}

int regtr_posemb_sine(const float* xyz, int n, int npf, int d_model, float scale, const float* dim_t, float* pe,
                      void* stream)
{
    if (!xyz || !dim_t || !pe || n < 0 || npf < 1 || d_model < 3 * npf) return RG_ERR_ARG;
    if (n == 0) return RG_OK;
    k_posemb_sine<<<rg_cdiv((long long)n * d_model, 256), 256, 0, (hipStream_t)stream>>>(xyz, n, npf, d_model, scale,
                                                                                          dim_t, pe);
    RG_RETURN_IF_LAUNCH_FAILED();
    return RG_OK;
}

}  // extern "C"
