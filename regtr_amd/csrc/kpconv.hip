// KPConv irregular neighbour gather + kernel-point correlation for gfx950, and the strided-block max-pool gather.
// Reference: KPConv.forward (non-deformable, 'linear' influence, 'sum' aggregation)
//   /root/reference/src/models/backbone_kpconv/kpconv_blocks.py:269-414, max_pool :127-143.
//
// The reference materialises (Nq,H,15,3) differences, (Nq,H,15) influences and (Nq,H,Cin) gathered features as
// separate tensors; here they never exist.  Three kernels, chosen by the launcher from Cin:
//   k_kpconv_gather_mfma<J,V>  Cin a multiple of 32 (every block of both shipped configs but the first): the per-query
//                              correlation runs on the f32 matrix cores, see the comment above the kernel;
//   k_kpconv_gather_c1         Cin == 1 (first block, features = ones): lanes = (query, kernel point);
//   k_kpconv_gather<LQ>        any other Cin -- the general LDS-tile form:
//     phase 1  lanes = neighbours: coalesced read of the index row, gather of the neighbour xyz, centred offsets and the
//              "feature sum > 0" flags into LDS;
//     phase 2  lanes = (neighbour, kernel point): 15 linear influences max(0, 1 - |y - kp_k| / extent) into an LDS-staged
//              [H][16] tile (each lane keeps its kernel point in registers);
//     phase 3  lanes = channels: the neighbour feature rows are gathered once (row-contiguous, line-sized reads),
//              influences are broadcast from LDS as 128-bit reads and 15 accumulators per channel stay in registers.
// Output is the weighted-feature matrix WF[q][k*Cin + c] that feeds the kernel-point contraction with the
// [15*Cin, Cout] weights (gemm_x3.hip / gemm.hip), plus the reference's data-dependent normaliser
//   num[q] = max(1, #{h : sum_c x[n_qh, c] > 0})            kpconv_blocks.py:409-411
// which the GEMM epilogue divides by.
#include "common.h"

namespace {

constexpr int KP_PAD = 16;   // kernel points padded to 16 per neighbour (15 used)
constexpr int GATHER_WAVES = 4;

// flag[j] = (sum_c x'[j, c] > 0) for every support row, x' = x or lrelu(InstanceNorm(x)) when stats are given;
// the shadow row (index ns) is 0 by construction.
__global__ void __launch_bounds__(256) k_rowsum_positive(const float* __restrict__ x, int n, int C, const float2* __restrict__ stats,
                                                         const int* __restrict__ seg_off, int n_seg, float slope,
                                                         float* __restrict__ flag)
{
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n) return;
    const int lane = rg_lane();
    const float2* st = stats ? stats + (size_t)rg_find_segment(seg_off, n_seg, row) * C : nullptr;
    float s = 0.f;
    for (int c = lane; c < C; c += RG_WAVE) {
        float v = x[(size_t)row * C + c];
        if (st) { v = (v - st[c].x) * st[c].y; v = v > 0.f ? v : v * slope; }
        s += v;
    }
    s = rg_wave_sum(s);
    if (lane == 0) flag[row] = s > 0.f ? 1.f : 0.f;
}

struct GatherArgs {
    const float* q_xyz; const float* s_xyz; const int* nbr; const float* x; const float* flag; const float* s_xyzf; const float* kp;
    float* wf; float* num;
    const float2* x_stats; const int* q_seg_off;     // optional fused lrelu(InstanceNorm(x)) on the gathered features
    int nq, ns, H, Cin, KP, n_seg, ld_wf;
    float extent, slope;
    int qpw;      // k_kpconv_gather_mfma: queries each wave handles one after another (<= MG_QPW; fewer on small launches)
};

// LQ = lanes per query (16, 32 or 64); a wave handles 64 / LQ queries at a time.
template <int LQ>
__global__ void __launch_bounds__(GATHER_WAVES * RG_WAVE) k_kpconv_gather(GatherArgs g)
{
    constexpr int QW = RG_WAVE / LQ;
    extern __shared__ __align__(16) float smem[];
    const int wave = threadIdx.x >> 6, lane = rg_lane();
    const int H = g.H;
    // per-wave LDS: w[QW][H][16] | rel[QW][H][3] | nidx[QW][H] | flag[QW][H] | x[QW][H] (Cin == 1 only)
    const int per_wave = (QW * H * (KP_PAD + 6) + 3) & ~3;   // keep every wave's tile 16-B aligned
    float* w_s = smem + (size_t)wave * per_wave;
    float* rel_s = w_s + QW * H * KP_PAD;
    int* idx_s = (int*)(rel_s + QW * H * 3);
    float* flg_s = (float*)(idx_s + QW * H);
    float* xs_s = flg_s + QW * H;

    const int q0 = (rg_xcd_block(blockIdx.x, gridDim.x) * GATHER_WAVES + wave) * QW;      // XCD-contiguous query ranges
    if (q0 >= g.nq) return;   // wave-uniform

    // ---- phase 1: neighbour indices, centred offsets, positivity flags
    for (int e = lane; e < QW * H; e += RG_WAVE) {
        const int qi = e / H, h = e - qi * H, q = q0 + qi;
        int idx = g.ns;
        float rx = 1e6f, ry = 1e6f, rz = 1e6f, f = 0.f, x1 = 0.f;
        if (q < g.nq) {
            idx = g.nbr[(size_t)q * H + h];
            float sx = 1e6f, sy = 1e6f, sz = 1e6f;            // shadow support point  (kpconv_blocks.py:309)
            if (idx < g.ns) {
                sx = g.s_xyz[3 * (size_t)idx]; sy = g.s_xyz[3 * (size_t)idx + 1]; sz = g.s_xyz[3 * (size_t)idx + 2];
                f = g.flag[idx];
                if (g.Cin == 1) x1 = g.x[idx];
            }
            rx = sx - g.q_xyz[3 * (size_t)q]; ry = sy - g.q_xyz[3 * (size_t)q + 1]; rz = sz - g.q_xyz[3 * (size_t)q + 2];
        }
        idx_s[e] = idx; flg_s[e] = f; xs_s[e] = x1;
        rel_s[3 * e] = rx; rel_s[3 * e + 1] = ry; rel_s[3 * e + 2] = rz;
    }
    __builtin_amdgcn_wave_barrier();

    // ---- phase 2: linear influences; lane's kernel point is fixed (k = lane % 16)
    {
        const int k = lane & (KP_PAD - 1);
        const bool kvalid = k < g.KP;
        const float kx = kvalid ? g.kp[3 * k] : 0.f, ky = kvalid ? g.kp[3 * k + 1] : 0.f, kz = kvalid ? g.kp[3 * k + 2] : 0.f;
        const float inv_extent = 1.0f / g.extent;
        for (int e = lane; e < QW * H * KP_PAD; e += RG_WAVE) {
            const int qh = e >> 4;   // (qi*H + h)
            const float dx = rel_s[3 * qh] - kx, dy = rel_s[3 * qh + 1] - ky, dz = rel_s[3 * qh + 2] - kz;
            float d2;
            {
#pragma clang fp contract(off)
                d2 = (dx * dx + dy * dy) + dz * dz;                               // kpconv_blocks.py:326-329
            }
            // :368  1 - sqrt(d2)/extent with the hardware sqrt (1 ulp) and a precomputed reciprocal: |error| <~ 2e-7
            float wv = 1.f - __builtin_amdgcn_sqrtf(d2) * inv_extent;
            w_s[e] = (kvalid && wv > 0.f) ? wv : 0.f;
        }
    }
    __builtin_amdgcn_wave_barrier();

    // ---- phase 3: lanes = channels of one query
    const int qi = lane / LQ, cl = lane % LQ, q = q0 + qi;
    if (q >= g.nq) return;
    const float* wq = w_s + (size_t)qi * H * KP_PAD;
    const int* iq = idx_s + qi * H;
    const int Cin = g.Cin;
    {
        const float2* st = nullptr;
        if (g.x_stats) st = g.x_stats + (size_t)rg_find_segment(g.q_seg_off, g.n_seg, q) * Cin;
        for (int c = cl; c < Cin; c += LQ) {
            float acc[KP_PAD];
#pragma unroll
            for (int k = 0; k < KP_PAD; k++) acc[k] = 0.f;
            float mu = 0.f, rs = 1.f;
            if (st) { mu = st[c].x; rs = st[c].y; }
            // neighbour rows are fetched 8 at a time before any of them is consumed: 8 independent gathers in flight per
            // lane hide the L2 / Infinity-Cache latency that a load-use chain per neighbour would expose
            for (int h0 = 0; h0 < H; h0 += 8) {
                float xv[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int h = h0 + u;
                    const int idx = h < H ? iq[h] : g.ns;
                    xv[u] = idx < g.ns ? g.x[(size_t)idx * Cin + c] : 0.f;          // zero shadow feature (:388)
                    if (st && idx < g.ns) { const float t = (xv[u] - mu) * rs; xv[u] = t > 0.f ? t : t * g.slope; }
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int h = h0 + u;
                    if (h < H) {
                        const float4* w4 = (const float4*)(wq + h * KP_PAD);
#pragma unroll
                        for (int j = 0; j < KP_PAD / 4; j++) {
                            const float4 wv = w4[j];
                            acc[4 * j + 0] = fmaf(wv.x, xv[u], acc[4 * j + 0]);
                            acc[4 * j + 1] = fmaf(wv.y, xv[u], acc[4 * j + 1]);
                            acc[4 * j + 2] = fmaf(wv.z, xv[u], acc[4 * j + 2]);
                            acc[4 * j + 3] = fmaf(wv.w, xv[u], acc[4 * j + 3]);
                        }
                    }
                }
            }
            float* o = g.wf + (size_t)q * g.KP * Cin + c;
#pragma unroll
            for (int k = 0; k < KP_PAD; k++)
                if (k < g.KP) o[(size_t)k * Cin] = acc[k];
        }
    }
    if (cl == 0) {
        float cnt = 0.f;
        const float* fq = flg_s + qi * H;
        for (int h = 0; h < H; h++) cnt += fq[h];
        g.num[q] = fmaxf(cnt, 1.f);                                               // :410-411
    }
}


// ------------------------------------------------------------------------------------------------------------------
// Matrix-core gather (Cin a multiple of 32).  For one query the kernel-point correlation
//   WF[k, c] = sum_h w[h, k] * x[n_h, c]
// is a [16 x H] x [H x Cin] product, so it runs on v_mfma_f32_16x16x4_f32 (exact f32, an fmaf chain over h in order):
//   A operand  lane (k = l & 15, hh = l >> 4)  holds  w[h = 4 j + hh][k]  -- computed by that very lane from the centred
//              neighbour offset and ITS kernel point, so influences go from the VALU straight into the MFMA with no LDS
//              tile and no broadcast reads;
//   B operand  lane (n = l & 15, hh)  holds  x[n_{4 j + hh}][c0 + V n + v], v < V: ONE 8- or 16-byte load per lane per
//              neighbour group, the 16 lanes of a group reading one whole 128-B / 256-B feature row segment; component
//              v feeds MFMA v, whose output column n therefore is channel c0 + V n + v.  All J loads of a pass are
//              issued before the first MFMA, so tens of KB per CU are in flight and L2 / Infinity-Cache latency hides;
//   D          lane holds WF[k = 4 hh + r][c0 + V n + v] in acc[v][r]: V consecutive channels -> one 8/16-byte store.
// The per-support positivity flag of the normaliser (kpconv_blocks.py:409-410) is a row sum over the very channels the
// wave has just gathered: per-lane partial sums + a 4-step DPP rotate-add over the 16 lanes of a group, so no separate
// pass over x (and no flag gather) is needed.
// ------------------------------------------------------------------------------------------------------------------
typedef float floatx4 __attribute__((ext_vector_type(4)));
#ifndef RG_MG_QPW
#define RG_MG_QPW 8
#endif
constexpr int MG_QPW = RG_MG_QPW;      // queries handled one after another by each wave

template <int V> struct RgVec;
template <> struct RgVec<2> { typedef float2 type; };
template <> struct RgVec<4> { typedef float4 type; };
__device__ __forceinline__ float rg_comp(const float2& v, int i) { return i == 0 ? v.x : v.y; }
__device__ __forceinline__ float rg_comp(const float4& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }
__device__ __forceinline__ void rg_set(float2& v, int i, float f) { if (i == 0) v.x = f; else v.y = f; }
__device__ __forceinline__ void rg_set(float4& v, int i, float f) { if (i == 0) v.x = f; else if (i == 1) v.y = f; else if (i == 2) v.z = f; else v.w = f; }

// sum over the 16 lanes of a DPP row (every lane of the row ends with the same, order-symmetric total)
__device__ __forceinline__ float rg_row16_sum(float v)
{
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));   // row_ror:8
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));   // row_ror:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));   // row_ror:2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));   // row_ror:1
    return v;
}

typedef unsigned rg_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned rg_u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned RG_OOB = 0x80000000u;      // byte offset beyond any buffer this kernel binds (all are < 2 GiB)

// Raw buffer access (buffer_load/store ... offen): 32-bit byte offsets against a scalar resource -- one address VGPR per
// access instead of a 64-bit pair plus the VALU that builds it -- and hardware range checking: an offset of RG_OOB
// loads zeros / drops the store, which is exactly the "shadow row" (kpconv_blocks.py:388) and the k >= KP mask.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rg_rsrc(const void* p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
template <int V> __device__ __forceinline__ typename RgVec<V>::type rg_buf_load(__amdgpu_buffer_rsrc_t r, unsigned off);
template <> __device__ __forceinline__ float2 rg_buf_load<2>(__amdgpu_buffer_rsrc_t r, unsigned off)
{
    const rg_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
    return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
}
template <> __device__ __forceinline__ float4 rg_buf_load<4>(__amdgpu_buffer_rsrc_t r, unsigned off)
{
    const rg_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
// Cache policy of the WF stores: nt (non-temporal, aux bit 1).  WF is written once, read once by the next kernel and is far
// larger than every cache (4.6 GB at level 0 of a 64-pair forward): as default-policy stores it evicted the feature rows the
// gather re-reads ~40 times from L2.  Measured on MI355X (tools/gather_bench.py, 64 pairs): all gather launches of a forward
// 5.40 -> 4.94 ms (Cin 64: 950 -> 786 us, Cin 128: 438 -> 358 us); without any store they take 4.42 ms.
#ifndef RG_WF_STORE_AUX
#define RG_WF_STORE_AUX 2
#endif
__device__ __forceinline__ void rg_buf_store(__amdgpu_buffer_rsrc_t r, unsigned off, const float2& v)
{
#ifndef RG_WF_SKIP_STORE
    __builtin_amdgcn_raw_buffer_store_b64(rg_u32x2{__float_as_uint(v.x), __float_as_uint(v.y)}, r, off, 0, RG_WF_STORE_AUX);
#else
    if (v.x == 1.2345e30f) __builtin_amdgcn_raw_buffer_store_b64(rg_u32x2{__float_as_uint(v.x), __float_as_uint(v.y)}, r, off, 0, 0);
#endif
}
__device__ __forceinline__ void rg_buf_store(__amdgpu_buffer_rsrc_t r, unsigned off, const float4& v)
{
#ifndef RG_WF_SKIP_STORE
    __builtin_amdgcn_raw_buffer_store_b128(rg_u32x4{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}, r, off, 0, RG_WF_STORE_AUX);
#else
    if (v.x == 1.2345e30f) __builtin_amdgcn_raw_buffer_store_b128(rg_u32x4{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}, r, off, 0, 0);
#endif
}

typedef float rg_f32x2 __attribute__((ext_vector_type(2)));
#ifndef RG_C1_PIPELINE
#define RG_C1_PIPELINE 1          // development A-B (REGTR_VARIANT_FLAGS=-DRG_C1_PIPELINE=0): the one-group-per-wave Cin = 1 gather
#endif
#ifndef RG_MG_WAVES_PER_EU
#define RG_MG_WAVES_PER_EU 3
#endif
#ifndef RG_MG_CHAINS
#define RG_MG_CHAINS 1             // development A-B (REGTR_VARIANT_FLAGS=-DRG_MG_CHAINS=2): see the MFMA loop
#endif
// PRE: g.s_xyzf holds (x, y, z, positivity flag) records and the features are final (no InstanceNorm fold) -- the launcher's choice
// when the caller passes `s_xyzf` and no x_stats.  Per neighbour ONE 16-byte load replaces three coordinate dwords (plus the flag):
// PMC on the level-0 launch showed the texture-address path 76 % busy, 148 of ~260 lines per query being those scattered dwords,
// and waves 68 % of their time in issue stalls behind it; per query ~170 of ~290 VALU instructions (fold, row sums) go as well.
template <int J, int V, bool PRE>    // J = ceil(H / 4) neighbour groups (H <= 4 J); V floats per lane per row per pass
__global__ void __launch_bounds__(GATHER_WAVES * RG_WAVE, RG_MG_WAVES_PER_EU) k_kpconv_gather_mfma(GatherArgs g)
{
    typedef typename RgVec<V>::type vec;
    constexpr int HP = 4 * J;
    __shared__ float4 nb_sh[GATHER_WAVES][HP];        // (rel.x, rel.y, rel.z, feature-row byte offset) per neighbour
    // readfirstlane: tells the compiler the wave index (hence q and everything derived from it) is wave-uniform, so the
    // per-query buffer resource lives in SGPRs instead of being "waterfalled" lane by lane
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = rg_lane();
    const int k = lane & 15, hh = lane >> 4;
    const int H = g.H, Cin = g.Cin, ns = g.ns, nq = g.nq;
    float4* nb_s = nb_sh[wave];
    const bool kvalid = k < g.KP;
    const int kc = kvalid ? k : 0;
    // a lane without a kernel point (k = 15) sits infinitely far away: its influence clamps to 0 with no extra select
    const float kx = kvalid ? g.kp[3 * kc] : 1e30f, ky = g.kp[3 * kc + 1], kz = g.kp[3 * kc + 2];
    const float inv_extent = 1.0f / g.extent;
    const unsigned row_bytes = (unsigned)Cin * 4u;
    const __amdgpu_buffer_rsrc_t x_rs = rg_rsrc(g.x, (unsigned)ns * row_bytes);
    const __amdgpu_buffer_rsrc_t sxyz_rs = rg_rsrc(g.s_xyz, (unsigned)ns * 12u);
    const __amdgpu_buffer_rsrc_t xyzf_rs = rg_rsrc(PRE ? g.s_xyzf : g.s_xyz, PRE ? (unsigned)ns * 16u : 0u);
    const unsigned lane_off = (unsigned)(V * k) * 4u;                 // this lane's channel bytes within a 16 V pass
    unsigned st_off[4];                                               // WF store offsets of the lane's 4 kernel-point rows
#pragma unroll
    for (int r = 0; r < 4; r++) st_off[r] = (4 * hh + r) < g.KP ? (unsigned)(4 * hh + r) * row_bytes + lane_off : RG_OOB;
    const unsigned wf_q_bytes = (unsigned)g.KP * row_bytes;

    // All gathers are BRANCH FREE (range-checked buffer loads, clamped rows): predicated loads would split the code into
    // basic blocks, and hipcc drains every outstanding load (s_waitcnt vmcnt(0)) at block boundaries, serialising the
    // prefetch.
    // XCD-aware: each XCD (own L2) takes one contiguous eighth of the queries -- neighbouring queries gather the same rows
    const int qpw = g.qpw;
    const int qbase = (rg_xcd_block(blockIdx.x, gridDim.x) * GATHER_WAVES + wave) * qpw;
    if (qbase >= nq) return;
    const int hl = lane < H ? lane : H - 1;
    auto load_idx = [&](int q) -> int {
        const int v = g.nbr[(size_t)(q < nq ? q : nq - 1) * H + hl];
        return (q < nq && lane < H) ? v : ns;
    };
    struct Nb { float rx, ry, rz, f; };
    auto load_nb = [&](int q, int idx) -> Nb {
        const bool real = idx < ns;                                  // else: shadow support point at 1e6 (kpconv_blocks.py:309)
        float sx, sy, sz, f = 0.f;
        if (PRE) {
            const rg_u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(xyzf_rs, real ? (unsigned)idx * 16u : RG_OOB, 0, 0);
            sx = __uint_as_float(r.x); sy = __uint_as_float(r.y); sz = __uint_as_float(r.z); f = __uint_as_float(r.w);
        } else {
            const unsigned so = real ? (unsigned)idx * 12u : RG_OOB;
            sx = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(sxyz_rs, so, 0, 0));
            sy = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(sxyz_rs, so + 4u, 0, 0));
            sz = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(sxyz_rs, so + 8u, 0, 0));
        }
        const unsigned qc = (unsigned)(q < nq ? q : nq - 1);
        const float qx = g.q_xyz[3 * qc], qy = g.q_xyz[3 * qc + 1], qz = g.q_xyz[3 * qc + 2];
        Nb n;
        n.rx = (real ? sx : 1e6f) - qx; n.ry = (real ? sy : 1e6f) - qy; n.rz = (real ? sz : 1e6f) - qz; n.f = f;
        return n;
    };
    // Software pipeline over the wave's queries: the index row of query q+2 and the neighbour coordinates of query q+1
    // are requested while query q is in its gather / MFMA phase, so the idx -> xyz -> features dependency chain of one
    // query overlaps the matrix work of the previous one instead of being exposed three memory latencies deep.
    int idx_cur = load_idx(qbase);
    Nb nb_cur = load_nb(qbase, idx_cur);
    int idx_nxt = load_idx(qbase + 1);
    int seg_cur = 0, seg_end = -1;      // cloud of the current query and its end row (x_stats path)
#pragma unroll 1
    for (int qq = 0; qq < qpw; qq++) {
        const int q = qbase + qq;
        if (q >= nq) return;            // wave-uniform
        __builtin_amdgcn_wave_barrier();
        if (lane < HP) {
            const unsigned ro = idx_cur < ns ? (unsigned)idx_cur * row_bytes : RG_OOB;
            nb_s[lane] = make_float4(nb_cur.rx, nb_cur.ry, nb_cur.rz, __uint_as_float(ro));
        }
        __builtin_amdgcn_wave_barrier();
        // ---- influences of kernel point k for neighbours h = 4 j + hh   (the A operands)
        // (Measured, round 4: the same arithmetic on the packed-float32 instructions -- neighbour pairs interleaved in LDS, v_pk_add / v_pk_mul /
        //  v_pk_fma, 75 instead of 120 vector instructions per query -- is bit-identical and NOT faster here: level 0 2118 vs 2114 us, level-0
        //  pool 545 vs 520, level 1 594 vs 584; six more registers cost the Cin = 32 form a wave per SIMD.  This kernel is not bound by the
        //  influence arithmetic; the Cin = 1 kernel below, which is nothing else, keeps the packed form.)
        float w[J];
        unsigned row[J];       // byte offset of the neighbour's feature row; RG_OOB for a shadow neighbour (reads as zeros)
#pragma unroll
        for (int j = 0; j < J; j++) {
            const float4 nb = nb_s[4 * j + hh];
            const float dx = nb.x - kx, dy = nb.y - ky, dz = nb.z - kz;
            float d2;
            {
#pragma clang fp contract(off)
                d2 = (dx * dx + dy * dy) + dz * dz;                                   // kpconv_blocks.py:326-329
            }
            w[j] = fmaxf(1.f - __builtin_amdgcn_sqrtf(d2) * inv_extent, 0.f);         // :368
            row[j] = __float_as_uint(nb.w) + lane_off;
        }
        const float2* st = nullptr;
        if (!PRE && g.x_stats) {
            // q is wave-uniform and the wave's queries are consecutive: the cloud changes at most rarely, and when it does
            // the whole wave finds it in one round trip (a per-lane binary search would be log2(n) DEPENDENT loads here)
            if (q >= seg_end) {
                seg_cur = rg_find_segment_wave(g.q_seg_off, g.n_seg, q);
                seg_end = g.q_seg_off[seg_cur + 1];
            }
            st = g.x_stats + (size_t)seg_cur * Cin;
        }
        const __amdgpu_buffer_rsrc_t wf_rs = rg_rsrc(g.wf + (size_t)q * g.KP * Cin, wf_q_bytes);
        Nb nb_nxt = nb_cur;
        int idx_nn = ns;
        const float f_cur = nb_cur.f;
        float rsum[J];
#pragma unroll
        for (int j = 0; j < J; j++) rsum[j] = 0.f;
        // ---- channel passes of 16 V channels
#pragma unroll 1
        for (int c0 = 0; c0 < Cin; c0 += 16 * V) {
            const unsigned c0b = (unsigned)c0 * 4u;
            vec xv[J];
#pragma unroll
            for (int j = 0; j < J; j++) xv[j] = rg_buf_load<V>(x_rs, row[j] + c0b);
            if (c0 == 0) {   // prefetch for the next queries, queued behind this query's feature gathers
                nb_nxt = load_nb(q + 1, idx_nxt);
                idx_nn = load_idx(q + 2);
            }
            if (!PRE && st) {   // fused lrelu(InstanceNorm(x)) of the preceding UnaryBlock (wave-uniform branch)
                float2 ms[V];
#pragma unroll
                for (int v = 0; v < V; v += 2) {
                    const float4 t = *(const float4*)(st + c0 + V * k + v);
                    ms[v] = make_float2(t.x, t.y); ms[v + 1] = make_float2(t.z, t.w);
                }
#pragma unroll
                for (int j = 0; j < J; j++)
#pragma unroll
                    for (int v = 0; v < V; v++) {
                        const float t = (rg_comp(xv[j], v) - ms[v].x) * ms[v].y;
                        rg_set(xv[j], v, fmaxf(t, t * g.slope));          // LeakyReLU, 0 < slope < 1
                    }
            }
            // A shadow neighbour's row needs no zeroing for WF: its influence w is exactly 0 (it sits 1e6 away).  Its row
            // sum is discarded below.
            if (!PRE)
#pragma unroll
            for (int j = 0; j < J; j++) {   // partial row sums for the normaliser (:409)
                float s = rg_comp(xv[j], 0);
#pragma unroll
                for (int v = 1; v < V; v++) s += rg_comp(xv[j], v);
                rsum[j] += s;
            }
            floatx4 acc[V];
#pragma unroll
            for (int v = 0; v < V; v++) acc[v] = floatx4{0.f, 0.f, 0.f, 0.f};
#if RG_MG_CHAINS == 2
            // probe (round 5): the J dependent MFMAs of a channel slice as TWO independent accumulator chains (even / odd
            // neighbour groups), added at the end -- does the issue-stall share fall?  (profiles/r05_gather_chains.md; not the product build)
            floatx4 accb[V];
#pragma unroll
            for (int v = 0; v < V; v++) accb[v] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < J; j++)
#pragma unroll
                for (int v = 0; v < V; v++) {
                    if (j & 1) accb[v] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j], rg_comp(xv[j], v), accb[v], 0, 0, 0);
                    else acc[v] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j], rg_comp(xv[j], v), acc[v], 0, 0, 0);
                }
#pragma unroll
            for (int v = 0; v < V; v++) acc[v] += accb[v];
#else
#pragma unroll
            for (int j = 0; j < J; j++)
#pragma unroll
                for (int v = 0; v < V; v++)
                    acc[v] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j], rg_comp(xv[j], v), acc[v], 0, 0, 0);
#endif
#pragma unroll
            for (int r = 0; r < 4; r++) {
                vec o;
#pragma unroll
                for (int v = 0; v < V; v++) rg_set(o, v, acc[v][r]);
                rg_buf_store(wf_rs, st_off[r] + c0b, o);
            }
        }
        // ---- normaliser: positive-row count over the query's real neighbours   (:409-411)
        float cnt = 0.f;
        if (PRE) {   // lanes = neighbours here (a lane beyond H or on a shadow neighbour loaded 0)
            cnt = (float)__builtin_popcountll(__ballot(f_cur > 0.f));
        } else {
#pragma unroll
            for (int j = 0; j < J; j++) cnt += (rg_row16_sum(rsum[j]) > 0.f && row[j] < RG_OOB) ? 1.f : 0.f;
            cnt += __shfl_xor(cnt, 16, RG_WAVE);
            cnt += __shfl_xor(cnt, 32, RG_WAVE);
        }
        if (lane == 0) g.num[q] = fmaxf(cnt, 1.f);
        idx_cur = idx_nxt; nb_cur = nb_nxt; idx_nxt = idx_nn;
    }
}

// (Rounds 3-4 built this gather on the 16-bit matrix pipe as well -- feature rows as f16 pair planes, staged row-wise in LDS and read
//  back with ds_read_b64_tr_b16 as v_mfma_f32_16x16x32_f16 operands; round 4 software-pipelined the row loads, swapped the operand roles
//  for 16-byte stores and cut the contraction to 48 slots.  Correct to 4e-7, a third of the ALU cycles, and still 12-20 % slower than the
//  kernel above: 40-50 % of a wave's time went into ISSUING the row loads against a full vector-memory pipe -- both kernels ingest the
//  gathered rows at ~15 B per clock and CU from L2 / Infinity Cache, and that is the limiter at Cin >= 64.  Removed after the
//  measurement; numbers and phase clocks in profiles/r04_f16_gather_v2.md, docs/NEGATIVES.md.)

#ifdef REGTR_EXPERIMENTAL      // measured slower (docs/NEGATIVES.md): built only into the experiment variant, never into the shipped library
// ------------------------------------------------------------------------------------------------------------------
// Gather FUSED with the kernel-point contraction, for the level-0 shape (Cin = Cout = 32, 15 kernel points): the weighted features
// never go to HBM (4.6 GB written and 4.6 GB read back per level-0 convolution of a 64-pair forward otherwise).
//   * one workgroup = 16 waves; each wave gathers one query per ROUND exactly as k_kpconv_gather_mfma<J, 2, true> does (final
//     features, packed support records) and leaves its WF[15][32] tile in LDS -- 16 queries per round = the M of a 16x16x16 MFMA;
//   * the three bf16 planes of W[480][32] (92 KB, rows padded to 488 for the banks) are resident in LDS for the workgroup's life;
//   * barrier; the 16 waves share the [16 x 480] x [480 x 32] product: wave (kg, nt) takes 3-4 of the 30 k-steps of one 16-column
//     half (A fragment = 16 bytes of a query's WF row, split exactly into three bf16 planes; six MFMAs per step: float32-grade like
//     gemm_x3.hip), partial 16 x 16 tiles go to LDS; barrier; 512 threads add the 8 partials of an output in fixed order
//     (deterministic), divide by the neighbour count (kpconv_blocks.py:411) and store.
// The price is two workgroup barriers per query round (measured on the plain gather: +22 %); what it buys is the whole contraction
// launch and the WF stores.
typedef short rg_s4 __attribute__((ext_vector_type(4)));
constexpr int FU_WAVES = 16, FU_QPW = 16, FU_K = 480, FU_WROW = 488, FU_N = 32;

struct FusedArgs {
    const float* q_xyz; const int* nbr; const float* x; const float* s_xyzf; const float* kp; const uint16_t* Wt;
    float* out;
    size_t plane;                 // elements per weight plane in global memory (Npad * Kp)
    int nq, ns, H, KP, Kp;
    float extent;
};

__device__ __forceinline__ unsigned fu_pack(float a, float b)
{
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    b2 v;
    v.x = (__bf16)a; v.y = (__bf16)b;
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void fu_split2(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2)
{
    p0 = fu_pack(a, b);
    const float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);
    p1 = fu_pack(ra, rb);
    p2 = fu_pack(ra - __uint_as_float(p1 << 16), rb - __uint_as_float(p1 & 0xffff0000u));
}

template <int J>
__global__ void __launch_bounds__(FU_WAVES * RG_WAVE) k_kpconv_fused(FusedArgs g)
{
    constexpr int HP = 4 * J, Cin = 32;
    extern __shared__ __align__(16) unsigned char fsm[];
    uint16_t* Wl = (uint16_t*)fsm;                                            // [3][32][FU_WROW] bf16
    float* wf_s = (float*)(fsm + (size_t)3 * FU_N * FU_WROW * 2);             // [16 queries][480]
    float* part_s = wf_s + FU_WAVES * FU_K;                                   // [8 k-groups][2 halves][16 x 16]
    float4* nb_all = (float4*)(part_s + 8 * 2 * 256);                         // [16 waves][HP]
    float* num_s = (float*)(nb_all + FU_WAVES * HP);                          // [2][16]
    int* qidx_s = (int*)(num_s + 2 * FU_WAVES);                               // [2][16]
    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = rg_lane();
    const int k = lane & 15, hh = lane >> 4;
    const int H = g.H, ns = g.ns, nq = g.nq;
    float4* nb_s = nb_all + wave * HP;
    const bool kvalid = k < g.KP;
    const int kc = kvalid ? k : 0;
    const float kx = kvalid ? g.kp[3 * kc] : 1e30f, ky = g.kp[3 * kc + 1], kz = g.kp[3 * kc + 2];
    const float inv_extent = 1.0f / g.extent;
    constexpr unsigned row_bytes = Cin * 4u;
    const __amdgpu_buffer_rsrc_t x_rs = rg_rsrc(g.x, (unsigned)ns * row_bytes);
    const __amdgpu_buffer_rsrc_t xyzf_rs = rg_rsrc(g.s_xyzf, (unsigned)ns * 16u);
    const unsigned lane_off = (unsigned)(2 * k) * 4u;

    // weight planes -> LDS (16-byte chunks; 60 per 480-bf16 row)
    for (int idx = t; idx < 3 * FU_N * 60; idx += FU_WAVES * RG_WAVE) {
        const int p = idx / (FU_N * 60), rem = idx - p * (FU_N * 60), n = rem / 60, c = rem - n * 60;
        *(uint4*)(Wl + (size_t)(p * FU_N + n) * FU_WROW + c * 8) = *(const uint4*)(g.Wt + (size_t)p * g.plane + (size_t)n * g.Kp + c * 8);
    }

    const int qbase = (rg_xcd_block(blockIdx.x, gridDim.x) * FU_WAVES + wave) * FU_QPW;
    const int hl = lane < H ? lane : H - 1;
    auto load_idx = [&](int q) -> int {
        const int v = g.nbr[(size_t)(q < nq ? q : nq - 1) * H + hl];
        return (q < nq && lane < H) ? v : ns;
    };
    struct Nb { float rx, ry, rz, f; };
    auto load_nb = [&](int q, int idx) -> Nb {
        const bool real = idx < ns;
        const rg_u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(xyzf_rs, real ? (unsigned)idx * 16u : RG_OOB, 0, 0);
        const unsigned qc = (unsigned)(q < nq ? q : nq - 1);
        const float qx = g.q_xyz[3 * qc], qy = g.q_xyz[3 * qc + 1], qz = g.q_xyz[3 * qc + 2];
        Nb n;
        n.rx = (real ? __uint_as_float(r.x) : 1e6f) - qx; n.ry = (real ? __uint_as_float(r.y) : 1e6f) - qy;
        n.rz = (real ? __uint_as_float(r.z) : 1e6f) - qz; n.f = real ? __uint_as_float(r.w) : 0.f;
        return n;
    };
    int idx_cur = load_idx(qbase);
    Nb nb_cur = load_nb(qbase, idx_cur);
    int idx_nxt = load_idx(qbase + 1);
    // stage-2 role of this wave: 16-column half nt, k-steps [ks0, ks0 + nks) of the 30
    const int nt = wave & 1, kg = wave >> 1;
    const int ks0 = kg < 6 ? 4 * kg : 24 + 3 * (kg - 6), nks = kg < 6 ? 4 : 3;
    int par = 0;
#pragma unroll 1
    for (int qq = 0; qq < FU_QPW; qq++) {                      // every wave runs every round (barriers inside); q >= nq: a zero tile
        const int q = qbase + qq;
        __builtin_amdgcn_wave_barrier();
        if (lane < HP) {
            const unsigned ro = idx_cur < ns ? (unsigned)idx_cur * row_bytes : RG_OOB;
            nb_s[lane] = make_float4(nb_cur.rx, nb_cur.ry, nb_cur.rz, __uint_as_float(ro));
        }
        __builtin_amdgcn_wave_barrier();
        float w[J];
        unsigned row[J];
#pragma unroll
        for (int j = 0; j < J; j++) {
            const float4 nb = nb_s[4 * j + hh];
            const float dx = nb.x - kx, dy = nb.y - ky, dz = nb.z - kz;
            float d2;
            {
#pragma clang fp contract(off)
                d2 = (dx * dx + dy * dy) + dz * dz;                                   // kpconv_blocks.py:326-329
            }
            w[j] = fmaxf(1.f - __builtin_amdgcn_sqrtf(d2) * inv_extent, 0.f);         // :368
            row[j] = __float_as_uint(nb.w) + lane_off;
        }
        float2 xv[J];
#pragma unroll
        for (int j = 0; j < J; j++) xv[j] = rg_buf_load<2>(x_rs, row[j]);
        const float f_cur = nb_cur.f;
        Nb nb_nxt = load_nb(q + 1, idx_nxt);
        const int idx_nn = load_idx(q + 2);
        floatx4 acc0 = floatx4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
        for (int j = 0; j < J; j++) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j], xv[j].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j], xv[j].y, acc1, 0, 0, 0);
        }
        // WF tile -> LDS: lane holds WF[kk = 4 hh + r][c = 2 k, 2 k + 1]
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int kk = 4 * hh + r;
            if (kk < 15) *(float2*)(wf_s + wave * FU_K + kk * Cin + 2 * k) = make_float2(acc0[r], acc1[r]);
        }
        const float cnt = (float)__builtin_popcountll(__ballot(f_cur > 0.f));
        if (lane == 0) { num_s[par * FU_WAVES + wave] = fmaxf(cnt, 1.f); qidx_s[par * FU_WAVES + wave] = q < nq ? q : -1; }
        idx_cur = idx_nxt; nb_cur = nb_nxt; idx_nxt = idx_nn;
        __syncthreads();                                                       // the 16 tiles (and, first round, the weights) are in LDS
        // ---- stage 2: [16 queries x 480] x [480 x 32], this wave's share
        // all fragments of the wave's (up to four) k-steps first, then three independent MFMA chains: between two workgroup barriers
        // the latency of this stage is what every wave of the workgroup waits for
        float4 av[4];
        uint2 bv[4][3];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int ks = ks0 + (i < nks ? i : 0);
            av[i] = *(const float4*)(wf_s + k * FU_K + 16 * ks + 4 * hh);     // query slot k (= lane & 15), 4 consecutive k-indices
#pragma unroll
            for (int p = 0; p < 3; p++)
                bv[i][p] = *(const uint2*)(Wl + (size_t)(p * FU_N + 16 * nt + k) * FU_WROW + 16 * ks + 4 * hh);
        }
        floatx4 o0 = floatx4{0.f, 0.f, 0.f, 0.f}, o1 = o0, o2 = o0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (i < nks) {                                                     // wave-uniform
                unsigned a0[3], a1[3];
                fu_split2(av[i].x, av[i].y, a0[0], a0[1], a0[2]);
                fu_split2(av[i].z, av[i].w, a1[0], a1[1], a1[2]);
                rg_s4 fa[3], fb[3];
#pragma unroll
                for (int p = 0; p < 3; p++) {
                    fa[p] = __builtin_bit_cast(rg_s4, make_uint2(a0[p], a1[p]));
                    fb[p] = __builtin_bit_cast(rg_s4, bv[i][p]);
                }
                o2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(fa[2], fb[0], o2, 0, 0, 0);      // the 2^-16 terms
                o2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(fa[1], fb[1], o2, 0, 0, 0);
                o2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(fa[0], fb[2], o2, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(fa[1], fb[0], o1, 0, 0, 0);      // the 2^-8 terms
                o1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(fa[0], fb[1], o1, 0, 0, 0);
                o0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(fa[0], fb[0], o0, 0, 0, 0);
            }
        }
        floatx4 o;
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = (o2[i] + o1[i]) + o0[i];            // smallest first
        // D: lane holds out[query slot 4 hh + i][column 16 nt + k]
#pragma unroll
        for (int i = 0; i < 4; i++) part_s[(kg * 2 + nt) * 256 + (4 * hh + i) * 16 + k] = o[i];
        __syncthreads();
        if (t < 512) {
            const int qs = t >> 5, col = t & 31, h2 = col >> 4, c = col & 15;
            float sum = part_s[h2 * 256 + qs * 16 + c];
#pragma unroll
            for (int g2 = 1; g2 < 8; g2++) sum += part_s[(g2 * 2 + h2) * 256 + qs * 16 + c];
            const int qi = qidx_s[par * FU_WAVES + qs];
            if (qi >= 0) g.out[(size_t)qi * FU_N + col] = sum / num_s[par * FU_WAVES + qs];
        }
        par ^= 1;
    }
}

#endif  // REGTR_EXPERIMENTAL

// Cin == 1 (first encoder block, features = ones): no channel dimension to spread over lanes, so lanes are
// (query, kernel point) pairs: 4 queries x 16 kernel points per wave, each lane walks its query's neighbours once and
// accumulates influence x feature directly -- no influence tile in LDS, 5 floats of LDS per neighbour instead of 22.
// The kernel is pure vector ALU (600 influences per query), so the neighbours are walked TWO at a time on the packed-float32
// instructions (v_pk_add / v_pk_mul / v_pk_fma: the same IEEE operations, two per instruction): the LDS image holds neighbour PAIRS
// (rows padded to an even length), one 8-byte read per quantity and pair.
__global__ void __launch_bounds__(GATHER_WAVES * RG_WAVE) k_kpconv_gather_c1(GatherArgs g)
{
    constexpr int QW = 4;
    extern __shared__ __align__(16) float smem[];
    const int wave = threadIdx.x >> 6, lane = rg_lane();
    const int H = g.H, HP = (H + 1) & ~1, NPAIR = HP >> 1;
    // per (query, neighbour pair): relx[2] | rely[2] | relz[2] | x[2] | flag[2] -- 40 bytes, so the walk below needs ONE address register
    float* pair_s = smem + (size_t)wave * QW * NPAIR * 10;
    const int q0 = (rg_xcd_block(blockIdx.x, gridDim.x) * GATHER_WAVES + wave) * QW;
    if (q0 >= g.nq) return;
    for (int e = lane; e < QW * HP; e += RG_WAVE) {
        const int qi = e / HP, h = e - qi * HP, q = q0 + qi;
        float rx = 1e6f, ry = 1e6f, rz = 1e6f, f = 0.f, x1 = 0.f;
        if (q < g.nq && h < H) {
            const int idx = g.nbr[(size_t)q * H + h];
            float sx = 1e6f, sy = 1e6f, sz = 1e6f;
            if (idx < g.ns) {
                if (g.s_xyzf) {   // (x, y, z, feature) in one 16-byte load
                    const float4 r = *(const float4*)(g.s_xyzf + 4 * (size_t)idx);
                    sx = r.x; sy = r.y; sz = r.z; x1 = r.w; f = x1 > 0.f ? 1.f : 0.f;
                } else {
                    sx = g.s_xyz[3 * (size_t)idx]; sy = g.s_xyz[3 * (size_t)idx + 1]; sz = g.s_xyz[3 * (size_t)idx + 2];
                    x1 = g.x[idx]; f = g.flag ? g.flag[idx] : (x1 > 0.f ? 1.f : 0.f);
                }
            }
            rx = sx - g.q_xyz[3 * (size_t)q]; ry = sy - g.q_xyz[3 * (size_t)q + 1]; rz = sz - g.q_xyz[3 * (size_t)q + 2];
        }
        float* d = pair_s + (qi * NPAIR + (h >> 1)) * 10 + (h & 1);      // (the pad slot of an odd H: a shadow neighbour, influence 0)
        d[0] = rx; d[2] = ry; d[4] = rz; d[6] = x1; d[8] = f;
    }
    __builtin_amdgcn_wave_barrier();
    const int qi = lane >> 4, k = lane & 15, q = q0 + qi;
    if (q >= g.nq) return;
    const bool kvalid = k < g.KP;
    const float kx = kvalid ? g.kp[3 * k] : 0.f, ky = kvalid ? g.kp[3 * k + 1] : 0.f, kz = kvalid ? g.kp[3 * k + 2] : 0.f;
    const float inv_extent = 1.0f / g.extent;
    const rg_f32x2 ninv{-inv_extent, -inv_extent}, one{1.f, 1.f};
    float acc = 0.f, cnt = 0.f;
    const float* pp = pair_s + qi * NPAIR * 10;
    for (int p = 0; p < NPAIR; p++, pp += 10) {
        const rg_f32x2 dx = *(const rg_f32x2*)(pp) - kx, dy = *(const rg_f32x2*)(pp + 2) - ky, dz = *(const rg_f32x2*)(pp + 4) - kz;
        const rg_f32x2 xv = *(const rg_f32x2*)(pp + 6), fv = *(const rg_f32x2*)(pp + 8);
        rg_f32x2 d2;
        {
#pragma clang fp contract(off)
            d2 = (dx * dx + dy * dy) + dz * dz;                                            // kpconv_blocks.py:326-329
        }
        const rg_f32x2 sq{__builtin_amdgcn_sqrtf(d2.x), __builtin_amdgcn_sqrtf(d2.y)};
        const rg_f32x2 t = __builtin_elementwise_fma(sq, ninv, one);                        // 1 - d / extent (:368)
        acc = fmaf(fmaxf(t.x, 0.f), xv.x, acc);                                            // (neighbour order kept)
        acc = fmaf(fmaxf(t.y, 0.f), xv.y, acc);
        cnt += fv.x + fv.y;                                                                // (0 / 1 flags: exact in any order)
    }
    if (k < g.ld_wf) g.wf[(size_t)q * g.ld_wf + k] = kvalid ? acc : 0.f;      // ld_wf = KP, or 16 with a zero pad column
    if (k == 0) g.num[q] = fmaxf(cnt, 1.f);
}

// The same, software-pipelined over several 4-query groups per wave (packed support records only): the kernel above spends most of a wave's
// life in the idx -> record dependency chain of its single group (two memory round trips before the first influence).  Measured at level 0
// of a 64-pair forward (2.4 M queries, tools/gather_bench.py --levels 7, same box): scalar influences 772 / 760 us, packed pairs 720 / 732,
// + this pipeline 676 / 681 -- bit-identical outputs; ~570 us is the vector-ALU time of the loop (two quarter-rate v_sqrt_f32 per pair).  Here group g + 1's records and group g + 2's index rows are in flight
// while group g's influences are computed from LDS (two LDS images, ping-pong).  NS = element slots per lane: 4 HP <= 64 NS.
template <int NS>
__global__ void __launch_bounds__(GATHER_WAVES * RG_WAVE) k_kpconv_gather_c1p(GatherArgs g)
{
    constexpr int QW = 4;
    extern __shared__ __align__(16) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = rg_lane();
    const int H = g.H, HP = (H + 1) & ~1, NPAIR = HP >> 1, nq = g.nq, ns = g.ns;
    float* buf_s = smem + (size_t)wave * 2 * QW * NPAIR * 10;            // [2][QW][NPAIR][10] (layout of k_kpconv_gather_c1)
    const int groups = g.qpw;                                            // 4-query groups per wave
    const int qbase = (rg_xcd_block(blockIdx.x, gridDim.x) * GATHER_WAVES + wave) * QW * groups;
    if (qbase >= nq) return;
    // element slots of this lane: e = lane + 64 s -> (query e / HP of the group, neighbour column e % HP)
    int e_q[NS], e_h[NS], e_dst[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const int e = lane + RG_WAVE * s;
        e_q[s] = e / HP; e_h[s] = e - e_q[s] * HP;
        if (e >= QW * HP) { e_q[s] = -1; e_h[s] = 0; }
        e_dst[s] = (e_q[s] * NPAIR + (e_h[s] >> 1)) * 10 + (e_h[s] & 1);
    }
    // branch-free loads from clamped addresses (a predicated load drains every outstanding one: tools/isa_scan.py)
    auto issue_idx = [&](int qg, int (&idx)[NS]) {
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const int q = qg + (e_q[s] < 0 ? 0 : e_q[s]);
            const int v = g.nbr[(size_t)(q < nq ? q : nq - 1) * H + (e_h[s] < H ? e_h[s] : H - 1)];
            idx[s] = (e_q[s] >= 0 && q < nq && e_h[s] < H) ? v : -1;     // -1: not an element (pad slot, query beyond the end)
        }
    };
    struct Rec { float4 r; float qx, qy, qz; };
    auto issue_rec = [&](int qg, const int (&idx)[NS], Rec (&rec)[NS]) {
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const int q = qg + (e_q[s] < 0 ? 0 : e_q[s]);
            const unsigned qc = (unsigned)(q < nq ? q : nq - 1);
            const int ic = idx[s] < 0 ? 0 : (idx[s] < ns ? idx[s] : ns - 1);
            rec[s].r = *(const float4*)(g.s_xyzf + 4 * (size_t)ic);
            rec[s].qx = g.q_xyz[3 * qc]; rec[s].qy = g.q_xyz[3 * qc + 1]; rec[s].qz = g.q_xyz[3 * qc + 2];
        }
    };
    auto stage = [&](float* buf, const int (&idx)[NS], const Rec (&rec)[NS]) {
#pragma unroll
        for (int s = 0; s < NS; s++) {
            if (e_q[s] < 0) continue;
            const bool elem = idx[s] >= 0, real = elem && idx[s] < ns;
            // (as k_kpconv_gather_c1: a shadow neighbour sits at 1e6 - q, a pad slot at 1e6; both have influence 0 and feature 0)
            const float sx = real ? rec[s].r.x : 1e6f, sy = real ? rec[s].r.y : 1e6f, sz = real ? rec[s].r.z : 1e6f;
            const float x1 = real ? rec[s].r.w : 0.f;
            float* d = buf + e_dst[s];
            d[0] = elem ? sx - rec[s].qx : 1e6f; d[2] = elem ? sy - rec[s].qy : 1e6f; d[4] = elem ? sz - rec[s].qz : 1e6f;
            d[6] = x1; d[8] = x1 > 0.f ? 1.f : 0.f;
        }
    };
    const int qi = lane >> 4, k = lane & 15;
    const bool kvalid = k < g.KP;
    const float kx = kvalid ? g.kp[3 * k] : 0.f, ky = kvalid ? g.kp[3 * k + 1] : 0.f, kz = kvalid ? g.kp[3 * k + 2] : 0.f;
    const float inv_extent = 1.0f / g.extent;
    const rg_f32x2 ninv{-inv_extent, -inv_extent}, one{1.f, 1.f};

    int idx_a[NS], idx_b[NS];
    Rec rec[NS];
    issue_idx(qbase, idx_a);
    issue_rec(qbase, idx_a, rec);
    issue_idx(qbase + QW, idx_b);
    stage(buf_s, idx_a, rec);
#pragma unroll 1
    for (int gi = 0; gi < groups; gi++) {
        const int qg = qbase + gi * QW;
        if (qg >= nq) return;                                            // wave-uniform
        int idx_c[NS];
        issue_rec(qg + QW, idx_b, rec);                                  // group gi + 1's records (its index row arrived a group ago)
        issue_idx(qg + 2 * QW, idx_c);
        __builtin_amdgcn_wave_barrier();
        const int q = qg + qi;
        const float* pp = buf_s + (gi & 1) * QW * NPAIR * 10 + qi * NPAIR * 10;
        float acc = 0.f, cnt = 0.f;
        for (int p = 0; p < NPAIR; p++, pp += 10) {
            const rg_f32x2 dx = *(const rg_f32x2*)(pp) - kx, dy = *(const rg_f32x2*)(pp + 2) - ky, dz = *(const rg_f32x2*)(pp + 4) - kz;
            const rg_f32x2 xv = *(const rg_f32x2*)(pp + 6), fv = *(const rg_f32x2*)(pp + 8);
            rg_f32x2 d2;
            {
#pragma clang fp contract(off)
                d2 = (dx * dx + dy * dy) + dz * dz;                                        // kpconv_blocks.py:326-329
            }
            const rg_f32x2 sq{__builtin_amdgcn_sqrtf(d2.x), __builtin_amdgcn_sqrtf(d2.y)};
            const rg_f32x2 t = __builtin_elementwise_fma(sq, ninv, one);                    // 1 - d / extent (:368)
            acc = fmaf(fmaxf(t.x, 0.f), xv.x, acc);
            acc = fmaf(fmaxf(t.y, 0.f), xv.y, acc);
            cnt += fv.x + fv.y;
        }
        if (q < nq) {
            if (k < g.ld_wf) g.wf[(size_t)q * g.ld_wf + k] = kvalid ? acc : 0.f;
            if (k == 0) g.num[q] = fmaxf(cnt, 1.f);
        }
        __builtin_amdgcn_wave_barrier();
        stage(buf_s + ((gi + 1) & 1) * QW * NPAIR * 10, idx_b, rec);
#pragma unroll
        for (int s = 0; s < NS; s++) idx_b[s] = idx_c[s];
    }
}

// out[q, c] = max_h x_pad[nbr[q, h], c]   with a zero shadow row   (kpconv_blocks.py:127-143)
// QW queries per wave: C / 4 lanes (one float4 each) serve a query when C <= 128, so that no lane idles on narrow rows
template <int QW>
__global__ void __launch_bounds__(256) k_maxpool_gather(const float* __restrict__ x, int ns, int C, const int* __restrict__ nbr,
                                                        int ld_nbr, int nq, int H, float* __restrict__ out)
{
    constexpr int LQ = RG_WAVE / QW;                 // lanes per query
    const int lane = rg_lane();
    const int q = (rg_xcd_block(blockIdx.x, gridDim.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * QW + lane / LQ;      // XCD-contiguous
    if (q >= nq) return;
    const int* row = nbr + (size_t)q * ld_nbr;
    for (int c = (lane % LQ) * 4; c < C; c += LQ * 4) {
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        for (int h0 = 0; h0 < H; h0 += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int idx = h0 + u < H ? row[h0 + u] : -1;
                v[u] = m;                                                    // out-of-range slot: neutral
                if (idx >= 0) v[u] = idx < ns ? *(const float4*)(x + (size_t)idx * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                m.x = fmaxf(m.x, v[u].x); m.y = fmaxf(m.y, v[u].y); m.z = fmaxf(m.z, v[u].z); m.w = fmaxf(m.w, v[u].w);
            }
        }
        *(float4*)(out + (size_t)q * C + c) = m;
    }
}

}  // namespace

// The same, BRANCH FREE.  The .s of the kernel above is a chain of `global_load ; s_waitcnt vmcnt(0)` pairs -- every index load and every
// row load sits in its own predicated block, so the "8 rows in flight" never were: one dependent round trip after the other (index,
// row, index, row ...).  Here the index row is read with clamped column numbers (a repeated neighbour does not change a maximum), the
// feature rows with range-checked buffer loads (an offset of RG_OOB returns zeros: exactly the zero shadow row), HB rows per batch
// all issued before the first maximum.  Needs ns * C * 4 < 2^32 - 256 (32-bit unsigned buffer offsets; MP_OOB, not RG_OOB, is the out-of-range
// offset here: the level-0 rows of a 192-pair forward are 3.7 GB, and with RG_OOB's 2 GiB bound that launch fell back to the predicated kernel
// above -- 2.65 ms against 1.9 for three times the 64-pair launch).
namespace {
constexpr unsigned MP_OOB = 0xfffffff0u;
template <int QW, int HB>
__global__ void __launch_bounds__(256) k_maxpool_gather_buf(const float* __restrict__ x, int ns, int C, const int* __restrict__ nbr,
                                                            int ld_nbr, int nq, int H, float* __restrict__ out)
{
    constexpr int LQ = RG_WAVE / QW;                 // lanes per query
    const int lane = rg_lane();
    const int q = (rg_xcd_block(blockIdx.x, gridDim.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * QW + lane / LQ;      // XCD-contiguous
    const int qc = q < nq ? q : nq - 1;              // a dead lane repeats the last query (it never stores)
    const int* row = nbr + (size_t)qc * ld_nbr;
    const unsigned row_bytes = (unsigned)C * 4u;
    const __amdgpu_buffer_rsrc_t x_rs = rg_rsrc(x, (unsigned)ns * row_bytes);
    for (int c = (lane % LQ) * 4; c < C; c += LQ * 4) {
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        const unsigned cb = (unsigned)c * 4u;
        for (int h0 = 0; h0 < H; h0 += HB) {
            int idx[HB];
#pragma unroll
            for (int u = 0; u < HB; u++) idx[u] = row[min(h0 + u, H - 1)];
            float4 v[HB];
#pragma unroll
            for (int u = 0; u < HB; u++) v[u] = rg_buf_load<4>(x_rs, (unsigned)idx[u] < (unsigned)ns ? (unsigned)idx[u] * row_bytes + cb : MP_OOB);
#pragma unroll
            for (int u = 0; u < HB; u++) {
                m.x = fmaxf(m.x, v[u].x); m.y = fmaxf(m.y, v[u].y); m.z = fmaxf(m.z, v[u].z); m.w = fmaxf(m.w, v[u].w);
            }
        }
        if (q < nq) *(float4*)(out + (size_t)q * C + c) = m;
    }
}
}  // namespace

extern "C" {

int regtr_rowsum_positive(const float* x, int n, int C, const float* stats, const int* seg_off, int n_seg, float slope,
                          float* flag, void* stream)
{
    if (!x || !flag || n < 0 || C < 1 || (stats && (!seg_off || n_seg < 1))) return RG_ERR_ARG;
    if (n == 0) return RG_OK;
    k_rowsum_positive<<<rg_cdiv(n, 4), 256, 0, (hipStream_t)stream>>>(x, n, C, (const float2*)stats, seg_off, n_seg, slope,
                                                                       flag);
    RG_RETURN_IF_LAUNCH_FAILED();
    return RG_OK;
}

// 1 when regtr_kpconv_gather derives the positivity flags from the rows it gathers (flag may then be NULL)
int regtr_kpconv_gather_computes_flag(int Cin, int H) { return (Cin == 1 || (Cin % 32 == 0 && H <= 64)) ? 1 : 0; }

// wf [nq, KP*Cin] (k-major, channel-minor: matches weights.view(KP*Cin, Cout)), num [nq].
int regtr_kpconv_gather(const float* q_xyz, int nq, const float* s_xyz, int ns, const int* nbr, int H, const float* x,
                        int Cin, const float* flag, const float* s_xyzf, const float* kernel_points, int KP, float extent,
                        const float* x_stats, const int* q_seg_off, int n_seg, float slope, float* wf, int ld_wf, float* num,
                        void* stream)
{
    if (ld_wf == 0) ld_wf = KP * Cin;
    if (ld_wf != KP * Cin && !(Cin == 1 && ld_wf == KP_PAD)) return RG_ERR_ARG;
    if (!q_xyz || !s_xyz || !nbr || !x || !kernel_points || !wf || !num || nq < 0 || ns < 0 || H < 1 ||
        Cin < 1 || KP < 1 || KP > KP_PAD || !(extent > 0.f) || (x_stats && (!q_seg_off || n_seg < 1)) ||
        (s_xyzf && (x_stats || (uintptr_t)s_xyzf % 16 || (long long)ns * 16 >= (1LL << 31))))
        return RG_ERR_ARG;
    if (!flag && !regtr_kpconv_gather_computes_flag(Cin, H)) return RG_ERR_ARG;
    if (nq == 0) return RG_OK;
    GatherArgs g{q_xyz, s_xyz, nbr, x, flag, s_xyzf, kernel_points, wf, num, (const float2*)x_stats, q_seg_off,
                 nq, ns, H, Cin, KP, n_seg, ld_wf, extent, slope, MG_QPW};
    hipStream_t st = (hipStream_t)stream;
    if (Cin == 1) {
        if (x_stats) return RG_ERR_ARG;
        const int HP1 = (H + 1) & ~1;
        if (RG_C1_PIPELINE && s_xyzf && 4 * HP1 <= 4 * RG_WAVE && nq > 0 && ns > 0) {
            // groups of 4 queries per wave: up to 8 where the level is large enough to keep ~8 workgroups per CU anyway
            int groups = (int)(((long long)nq + 256LL * 8 * GATHER_WAVES * 4 - 1) / (256LL * 8 * GATHER_WAVES * 4));
            groups = groups < 1 ? 1 : (groups > 8 ? 8 : groups);
          if (groups >= 2) {      // (a pair or two per forward: one group per wave fills the chip better -- 14.3 vs 15.6 us at one pair)
            g.qpw = groups;
            const size_t ldsp = (size_t)GATHER_WAVES * 2 * 4 * HP1 * 5 * sizeof(float);
            const int gridp = rg_xcd_grid(rg_cdiv(nq, GATHER_WAVES * 4 * groups));
            if (4 * HP1 <= 2 * RG_WAVE) k_kpconv_gather_c1p<2><<<gridp, GATHER_WAVES * RG_WAVE, ldsp, st>>>(g);
            else if (4 * HP1 <= 3 * RG_WAVE) k_kpconv_gather_c1p<3><<<gridp, GATHER_WAVES * RG_WAVE, ldsp, st>>>(g);
            else k_kpconv_gather_c1p<4><<<gridp, GATHER_WAVES * RG_WAVE, ldsp, st>>>(g);
            RG_RETURN_IF_LAUNCH_FAILED();
            return RG_OK;
          }
        }
        const size_t lds1 = (size_t)GATHER_WAVES * 4 * HP1 * 5 * sizeof(float);
        k_kpconv_gather_c1<<<rg_xcd_grid(rg_cdiv(nq, GATHER_WAVES * 4)), GATHER_WAVES * RG_WAVE, lds1, st>>>(g);
        RG_RETURN_IF_LAUNCH_FAILED();
        return RG_OK;
    }
    const bool aligned16 = (((uintptr_t)x | (uintptr_t)wf | (uintptr_t)x_stats) & 15) == 0;
    if (!flag && !(aligned16 && ns > 0 && (long long)ns * Cin < (1LL << 29))) return RG_ERR_ARG;
    if (regtr_kpconv_gather_computes_flag(Cin, H) && aligned16 && ns > 0 && (long long)ns * Cin < (1LL << 29)) {   // matrix-core path
        // queries per wave: MG_QPW (8) on large levels (the software pipeline over a wave's queries hides the idx -> xyz -> rows chain);
        // a small level (one pair: ~5000 queries at level 1) would be ~150 workgroups of 32 serial-ish queries on 256 CUs, so it gets
        // as few per wave as still puts ~12 waves on every CU
        int qpw = (int)(((long long)nq + 256 * 12 - 1) / (256 * 12));
        qpw = qpw < 1 ? 1 : (qpw > MG_QPW ? MG_QPW : qpw);
        g.qpw = qpw;
        const int grid_m = rg_xcd_grid(rg_cdiv(nq, GATHER_WAVES * qpw));
        const int J = H <= 40 ? 10 : (H <= 52 ? 13 : 16);
        const bool v4 = Cin % 64 == 0;
        const bool pre = s_xyzf != nullptr;
#define RG_LAUNCH_MG2(JJ, VV) do { if (pre) k_kpconv_gather_mfma<JJ, VV, true><<<grid_m, GATHER_WAVES * RG_WAVE, 0, st>>>(g); \
                                   else k_kpconv_gather_mfma<JJ, VV, false><<<grid_m, GATHER_WAVES * RG_WAVE, 0, st>>>(g); } while (0)
#define RG_LAUNCH_MG(JJ) do { if (v4) RG_LAUNCH_MG2(JJ, 4); else RG_LAUNCH_MG2(JJ, 2); } while (0)
        if (J == 10) RG_LAUNCH_MG(10); else if (J == 13) RG_LAUNCH_MG(13); else RG_LAUNCH_MG(16);
#undef RG_LAUNCH_MG
#undef RG_LAUNCH_MG2
        RG_RETURN_IF_LAUNCH_FAILED();
        return RG_OK;
    }
    const int LQ = Cin <= 16 ? 16 : (Cin <= 32 ? 32 : 64);
    const int QW = RG_WAVE / LQ;
    const size_t lds = (size_t)GATHER_WAVES * ((QW * H * (KP_PAD + 6) + 3) & ~3) * sizeof(float);
    if (lds > 160 * 1024) return RG_ERR_ARG;
    const int grid = rg_xcd_grid(rg_cdiv(nq, GATHER_WAVES * QW));
    if (LQ == 16) k_kpconv_gather<16><<<grid, GATHER_WAVES * RG_WAVE, lds, st>>>(g);
    else if (LQ == 32) k_kpconv_gather<32><<<grid, GATHER_WAVES * RG_WAVE, lds, st>>>(g);
    else k_kpconv_gather<64><<<grid, GATHER_WAVES * RG_WAVE, lds, st>>>(g);
    RG_RETURN_IF_LAUNCH_FAILED();
    return RG_OK;
}

#ifdef REGTR_EXPERIMENTAL
// 1 when regtr_kpconv_fused serves the convolution: 32 -> 32 channels, 15 kernel points, rows of at most 40 neighbours
int regtr_kpconv_fused_supported(int Cin, int Cout, int KP, int H) { return (Cin == 32 && Cout == 32 && KP == 15 && H >= 1 && H <= 40) ? 1 : 0; }

// KPConv.forward in ONE launch for the level-0 shape: out[nq, 32] = (gather + kernel-point correlation) x W / neighbour count
// (kpconv_blocks.py:269-414).  x [ns,32] final features, s_xyzf [ns,4] = (x, y, z, positivity flag) records (regtr_instnorm_apply),
// planes = regtr_gemm_split_weights(W viewed as [480, 32], transposed = 1).
int regtr_kpconv_fused(const float* q_xyz, int nq, int ns, const int* nbr, int H, const float* x, const float* s_xyzf,
                       const float* kernel_points, int KP, float extent, const void* planes, float* out, void* stream)
{
    if (!q_xyz || !nbr || !x || !s_xyzf || !kernel_points || !planes || !out || nq < 0 || ns < 1 || !(extent > 0.f)) return RG_ERR_ARG;
    if (!regtr_kpconv_fused_supported(32, 32, KP, H) || (long long)ns * 32 >= (1LL << 29)) return RG_ERR_ARG;
    if (((uintptr_t)x | (uintptr_t)s_xyzf | (uintptr_t)planes | (uintptr_t)out) % 16) return RG_ERR_ARG;
    if (nq == 0) return RG_OK;
    const int Kp = FU_K, Npad = 128;                              // regtr_gemm_split_weights(N = 32, K = 480): Kp = 480, Npad = 128
    FusedArgs g{q_xyz, nbr, x, s_xyzf, kernel_points, (const uint16_t*)planes, out, (size_t)Npad * Kp, nq, ns, H, KP, Kp, extent};
    constexpr int J = 10, HP = 4 * J;
    const size_t lds = (size_t)3 * FU_N * FU_WROW * 2 + (size_t)FU_WAVES * FU_K * 4 + 8 * 2 * 256 * 4 + (size_t)FU_WAVES * HP * 16 + 2 * FU_WAVES * 8;
    if (!rg_allow_dynamic_lds<k_kpconv_fused<J>>(lds)) return RG_ERR_ARG;
    const int grid = rg_xcd_grid(rg_cdiv(nq, FU_WAVES * FU_QPW));
    k_kpconv_fused<J><<<grid, FU_WAVES * RG_WAVE, lds, (hipStream_t)stream>>>(g);
    RG_RETURN_IF_LAUNCH_FAILED();
    return RG_OK;
}

#endif  // REGTR_EXPERIMENTAL

int regtr_maxpool_gather(const float* x, int ns, int C, const int* nbr, int ld_nbr, int nq, int H, float* out, void* stream)
{
    if (!x || !nbr || !out || ns < 0 || nq < 0 || H < 1 || ld_nbr < H || C < 4 || C % 4) return RG_ERR_ARG;
    if (nq == 0) return RG_OK;
    if (ns > 0 && (unsigned long long)ns * C * 4ull < 0xffffff00ull && ((uintptr_t)x % 16) == 0) {      // the branch-free form (32-bit buffer offsets)
        if (C <= 64) k_maxpool_gather_buf<4, 8><<<rg_xcd_grid(rg_cdiv(nq, 16)), 256, 0, (hipStream_t)stream>>>(x, ns, C, nbr, ld_nbr, nq, H, out);
        else if (C <= 128) k_maxpool_gather_buf<2, 8><<<rg_xcd_grid(rg_cdiv(nq, 8)), 256, 0, (hipStream_t)stream>>>(x, ns, C, nbr, ld_nbr, nq, H, out);
        else k_maxpool_gather_buf<1, 8><<<rg_xcd_grid(rg_cdiv(nq, 4)), 256, 0, (hipStream_t)stream>>>(x, ns, C, nbr, ld_nbr, nq, H, out);
        RG_RETURN_IF_LAUNCH_FAILED();
        return RG_OK;
    }
    if (C <= 64) k_maxpool_gather<4><<<rg_xcd_grid(rg_cdiv(nq, 16)), 256, 0, (hipStream_t)stream>>>(x, ns, C, nbr, ld_nbr, nq, H, out);
    else if (C <= 128) k_maxpool_gather<2><<<rg_xcd_grid(rg_cdiv(nq, 8)), 256, 0, (hipStream_t)stream>>>(x, ns, C, nbr, ld_nbr, nq, H, out);
    else k_maxpool_gather<1><<<rg_xcd_grid(rg_cdiv(nq, 4)), 256, 0, (hipStream_t)stream>>>(x, ns, C, nbr, ld_nbr, nq, H, out);
    RG_RETURN_IF_LAUNCH_FAILED();
    return RG_OK;
}

}  // extern "C"
