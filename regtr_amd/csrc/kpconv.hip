// KPConv irregular neighbour gather + kernel-point correlation for gfx950, and the strided-block max-pool gather.
// Reference: KPConv.forward (non-deformable, 'linear' influence, 'sum' aggregation)
//   /root/reference/src/models/backbone_kpconv/kpconv_blocks.py:269-414, max_pool :127-143.
//
// The reference materialises (Nq,H,15,3) differences, (Nq,H,15) influences and (Nq,H,Cin) gathered features as
// separate tensors.  Here one pass per query tile does everything on chip:
//   phase 1  lanes = neighbours: coalesced read of the index row, gather of the neighbour xyz, centred offsets and
//            the "feature sum > 0" flags into LDS;
//   phase 2  lanes = (neighbour, kernel point): 15 linear influences max(0, 1 - |y - kp_k| / extent) into an
//            LDS-staged [H][16] tile (each lane keeps its kernel point in registers);
//   phase 3  lanes = channels: the neighbour feature rows are gathered once (row-contiguous, line-sized reads),
//            influences are broadcast from LDS as 128-bit reads and 15 accumulators per channel stay in registers.
// Output is the weighted-feature matrix WF[q][k*Cin + c] that feeds the MFMA contraction with the [15*Cin, Cout]
// kernel weights (gemm.hip), plus the reference's data-dependent normaliser
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
    const float* q_xyz; const float* s_xyz; const int* nbr; const float* x; const float* flag; const float* kp;
    float* wf; float* num;
    const float2* x_stats; const int* q_seg_off;     // optional fused lrelu(InstanceNorm(x)) on the gathered features
    int nq, ns, H, Cin, KP, n_seg;
    float extent, slope;
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

    const int q0 = (blockIdx.x * GATHER_WAVES + wave) * QW;
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
// Matrix-core gather.  For one query the kernel-point correlation  WF[k, c] = sum_h w[h, k] * x[n_h, c]  is a
// [16 x H] x [H x Cin] product, so it runs on v_mfma_f32_16x16x4_f32 (exact f32, an fmaf chain over h in order):
//   A operand  lane (k = l & 15, hh = l >> 4)  holds  w[h = 4 j + hh][k]      -- computed by that very lane from the
//              centred neighbour offset and ITS kernel point, so influences go from the VALU straight into the MFMA
//              with no LDS tile and no broadcast reads;
//   B operand  lane (c = l & 15, hh = l >> 4)  holds  x[n_{4 j + hh}][c0 + c] -- one dword of a gathered row; all
//              J x (Cin/16) gathers of a query are issued before the first MFMA, so tens of independent loads per lane
//              are in flight and L2 / Infinity-Cache latency disappears behind them;
//   D          lane holds WF[k = 4 hh + r][c0 + c], r = 0..3, written straight to the weighted-feature matrix.
// The VALU only computes 10-16 influences per lane per query (hardware sqrt); the 15 x H x Cin multiply-adds run on
// the matrix pipe at its full f32 rate while other waves' loads and influence maths overlap.
// ------------------------------------------------------------------------------------------------------------------
typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr int MG_QPW = 8;      // queries handled one after another by each wave

template <int J, int MG_CB>    // J = ceil(H / 4) neighbour groups (H <= 4 J); MG_CB 16-channel blocks per pass
__global__ void __launch_bounds__(GATHER_WAVES * RG_WAVE) k_kpconv_gather_mfma(GatherArgs g)
{
    constexpr int HP = 4 * J;
    __shared__ float rel_sh[GATHER_WAVES][HP * 3];
    __shared__ int idx_sh[GATHER_WAVES][HP];
    __shared__ float flg_sh[GATHER_WAVES][HP];
    const int wave = threadIdx.x >> 6, lane = rg_lane();
    const int k = lane & 15, hh = lane >> 4;
    const int H = g.H, Cin = g.Cin, ns = g.ns, nq = g.nq;
    float* rel_s = rel_sh[wave];
    int* idx_s = idx_sh[wave];
    float* flg_s = flg_sh[wave];
    const bool kvalid = k < g.KP;
    const int kc = kvalid ? k : 0;
    const float kx = g.kp[3 * kc], ky = g.kp[3 * kc + 1], kz = g.kp[3 * kc + 2];
    const float inv_extent = 1.0f / g.extent;

    // All gathers below are BRANCH FREE: out-of-range rows / shadow neighbours load from a clamped (valid) address and
    // are replaced by a select afterwards.  Predicated loads would split the code into basic blocks, and hipcc drains
    // every outstanding load (s_waitcnt vmcnt(0)) at such block boundaries, which serialises the prefetch.
    const int qbase = (blockIdx.x * GATHER_WAVES + wave) * MG_QPW;
    if (qbase >= nq) return;
    const int hl = lane < H ? lane : H - 1;
    auto load_idx = [&](int q) -> int {
        const int v = g.nbr[(size_t)(q < nq ? q : nq - 1) * H + hl];
        return (q < nq && lane < H) ? v : ns;
    };
    struct Nb { float rx, ry, rz, f; };
    auto load_nb = [&](int q, int idx) -> Nb {
        const unsigned ic = (unsigned)(idx < ns ? idx : ns - 1), qc = (unsigned)(q < nq ? q : nq - 1);
        const float sx = g.s_xyz[3 * ic], sy = g.s_xyz[3 * ic + 1], sz = g.s_xyz[3 * ic + 2], f = g.flag[ic];
        const float qx = g.q_xyz[3 * qc], qy = g.q_xyz[3 * qc + 1], qz = g.q_xyz[3 * qc + 2];
        const bool real = idx < ns;                                  // else: shadow support point at 1e6 (kpconv_blocks.py:309)
        Nb n;
        n.rx = (real ? sx : 1e6f) - qx; n.ry = (real ? sy : 1e6f) - qy; n.rz = (real ? sz : 1e6f) - qz;
        n.f = real ? f : 0.f;
        return n;
    };
    // Software pipeline over the wave's queries: the index row of query q+2 and the neighbour coordinates of query q+1
    // are requested while query q is in its gather / MFMA phase, so the idx -> xyz -> features dependency chain of one
    // query overlaps the matrix work of the previous one instead of being exposed three memory latencies deep.
    int idx_cur = load_idx(qbase);
    Nb nb_cur = load_nb(qbase, idx_cur);
    int idx_nxt = load_idx(qbase + 1);
#pragma unroll 1
    for (int qq = 0; qq < MG_QPW; qq++) {
        const int q = qbase + qq;
        if (q >= nq) return;            // wave-uniform
        __builtin_amdgcn_wave_barrier();
        if (lane < HP) {
            rel_s[3 * lane] = nb_cur.rx; rel_s[3 * lane + 1] = nb_cur.ry; rel_s[3 * lane + 2] = nb_cur.rz;
            idx_s[lane] = idx_cur; flg_s[lane] = nb_cur.f;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- influences of kernel point k for neighbours h = 4 j + hh   (the A operands)
        float w[J];
        int nidx[J];
        float fsum = 0.f;
#pragma unroll
        for (int j = 0; j < J; j++) {
            const int h = 4 * j + hh;
            const float dx = rel_s[3 * h] - kx, dy = rel_s[3 * h + 1] - ky, dz = rel_s[3 * h + 2] - kz;
            float d2;
            {
#pragma clang fp contract(off)
                d2 = (dx * dx + dy * dy) + dz * dz;                                   // kpconv_blocks.py:326-329
            }
            const float wv = 1.f - __builtin_amdgcn_sqrtf(d2) * inv_extent;           // :368
            w[j] = (kvalid && wv > 0.f) ? wv : 0.f;
            nidx[j] = idx_s[h];
            fsum += flg_s[h];
        }
        // normaliser: every 16-lane group saw the flags of its hh; combine the four groups   (:409-411)
        fsum += __shfl_xor(fsum, 16, RG_WAVE);
        fsum += __shfl_xor(fsum, 32, RG_WAVE);
        if (lane == 0) g.num[q] = fmaxf(fsum, 1.f);

        const float2* st = nullptr;
        if (g.x_stats) st = g.x_stats + (size_t)rg_find_segment(g.q_seg_off, g.n_seg, q) * Cin;
        float* wf_q = g.wf + (size_t)q * g.KP * Cin;
        Nb nb_nxt = nb_cur;
        int idx_nn = ns;
        // ---- channel passes of up to 16 * MG_CB channels
        for (int c0 = 0; c0 < Cin; c0 += 16 * MG_CB) {
            float xv[J][MG_CB];
#pragma unroll
            for (int j = 0; j < J; j++) {
                // 32-bit element offsets (host guarantees ns * Cin < 2^30): one address VGPR per gather, not two
                const unsigned row = (unsigned)(nidx[j] < ns ? nidx[j] : ns - 1) * (unsigned)Cin;
#pragma unroll
                for (int cb = 0; cb < MG_CB; cb++) {
                    const int c = c0 + cb * 16 + k;
                    xv[j][cb] = g.x[row + (unsigned)(c < Cin ? c : Cin - 1)];
                }
            }
            if (c0 == 0) {   // prefetch for the next queries, queued behind this query's feature gathers
                nb_nxt = load_nb(q + 1, idx_nxt);
                idx_nn = load_idx(q + 2);
            }
            if (st) {   // fused lrelu(InstanceNorm(x)) of the preceding UnaryBlock (wave-uniform branch)
#pragma unroll
                for (int cb = 0; cb < MG_CB; cb++) {
                    const int c = c0 + cb * 16 + k;
                    const float2 ms = st[c < Cin ? c : Cin - 1];
#pragma unroll
                    for (int j = 0; j < J; j++) {
                        const float t = (xv[j][cb] - ms.x) * ms.y;
                        xv[j][cb] = t > 0.f ? t : t * g.slope;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < J; j++) {   // zero shadow row (:388) and out-of-range channels
                const bool real = nidx[j] < ns;
#pragma unroll
                for (int cb = 0; cb < MG_CB; cb++) xv[j][cb] = (real && c0 + cb * 16 + k < Cin) ? xv[j][cb] : 0.f;
            }
            floatx4 acc[MG_CB];
#pragma unroll
            for (int cb = 0; cb < MG_CB; cb++) acc[cb] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < J; j++)
#pragma unroll
                for (int cb = 0; cb < MG_CB; cb++)
                    acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j], xv[j][cb], acc[cb], 0, 0, 0);
#pragma unroll
            for (int cb = 0; cb < MG_CB; cb++) {
                const int c = c0 + cb * 16 + k;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int kk = 4 * hh + r;
                    if (kk < g.KP && c < Cin) wf_q[(unsigned)kk * (unsigned)Cin + (unsigned)c] = acc[cb][r];
                }
            }
        }
        idx_cur = idx_nxt; nb_cur = nb_nxt; idx_nxt = idx_nn;
    }
}

// Cin == 1 (first encoder block, features = ones): no channel dimension to spread over lanes, so lanes are
// (query, kernel point) pairs: 4 queries x 16 kernel points per wave, each lane walks its query's neighbours once and
// accumulates influence x feature directly -- no influence tile in LDS, 6 floats of LDS per neighbour instead of 22.
__global__ void __launch_bounds__(GATHER_WAVES * RG_WAVE) k_kpconv_gather_c1(GatherArgs g)
{
    constexpr int QW = 4;
    extern __shared__ __align__(16) float smem[];
    const int wave = threadIdx.x >> 6, lane = rg_lane();
    const int H = g.H;
    float* rel_s = smem + (size_t)wave * QW * H * 5;      // rel[QW][H][3] | x[QW][H] | flag[QW][H]
    float* xs_s = rel_s + QW * H * 3;
    float* flg_s = xs_s + QW * H;
    const int q0 = (blockIdx.x * GATHER_WAVES + wave) * QW;
    if (q0 >= g.nq) return;
    for (int e = lane; e < QW * H; e += RG_WAVE) {
        const int qi = e / H, h = e - qi * H, q = q0 + qi;
        float rx = 1e6f, ry = 1e6f, rz = 1e6f, f = 0.f, x1 = 0.f;
        if (q < g.nq) {
            const int idx = g.nbr[(size_t)q * H + h];
            float sx = 1e6f, sy = 1e6f, sz = 1e6f;
            if (idx < g.ns) {
                sx = g.s_xyz[3 * (size_t)idx]; sy = g.s_xyz[3 * (size_t)idx + 1]; sz = g.s_xyz[3 * (size_t)idx + 2];
                f = g.flag[idx]; x1 = g.x[idx];
            }
            rx = sx - g.q_xyz[3 * (size_t)q]; ry = sy - g.q_xyz[3 * (size_t)q + 1]; rz = sz - g.q_xyz[3 * (size_t)q + 2];
        }
        rel_s[3 * e] = rx; rel_s[3 * e + 1] = ry; rel_s[3 * e + 2] = rz; xs_s[e] = x1; flg_s[e] = f;
    }
    __builtin_amdgcn_wave_barrier();
    const int qi = lane >> 4, k = lane & 15, q = q0 + qi;
    if (q >= g.nq) return;
    const bool kvalid = k < g.KP;
    const float kx = kvalid ? g.kp[3 * k] : 0.f, ky = kvalid ? g.kp[3 * k + 1] : 0.f, kz = kvalid ? g.kp[3 * k + 2] : 0.f;
    const float inv_extent = 1.0f / g.extent;
    float acc = 0.f, cnt = 0.f;
    for (int h = 0; h < H; h++) {
        const int e = qi * H + h;
        const float dx = rel_s[3 * e] - kx, dy = rel_s[3 * e + 1] - ky, dz = rel_s[3 * e + 2] - kz;
        float d2;
        {
#pragma clang fp contract(off)
            d2 = (dx * dx + dy * dy) + dz * dz;
        }
        const float wv = fmaxf(1.f - __builtin_amdgcn_sqrtf(d2) * inv_extent, 0.f);
        acc = fmaf(wv, xs_s[e], acc);
        cnt += flg_s[e];
    }
    if (kvalid) g.wf[(size_t)q * g.KP + k] = acc;
    if (k == 0) g.num[q] = fmaxf(cnt, 1.f);
}

// out[q, c] = max_h x_pad[nbr[q, h], c]   with a zero shadow row   (kpconv_blocks.py:127-143)
__global__ void __launch_bounds__(256) k_maxpool_gather(const float* __restrict__ x, int ns, int C, const int* __restrict__ nbr,
                                                        int nq, int H, float* __restrict__ out)
{
    const int q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (q >= nq) return;
    const int lane = rg_lane();
    const int* row = nbr + (size_t)q * H;
    for (int c = lane * 4; c < C; c += RG_WAVE * 4) {
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

// wf [nq, KP*Cin] (k-major, channel-minor: matches weights.view(KP*Cin, Cout)), num [nq].
int regtr_kpconv_gather(const float* q_xyz, int nq, const float* s_xyz, int ns, const int* nbr, int H, const float* x,
                        int Cin, const float* flag, const float* kernel_points, int KP, float extent,
                        const float* x_stats, const int* q_seg_off, int n_seg, float slope, float* wf, float* num,
                        void* stream)
{
    if (!q_xyz || !s_xyz || !nbr || !x || !flag || !kernel_points || !wf || !num || nq < 0 || ns < 0 || H < 1 ||
        Cin < 1 || KP < 1 || KP > KP_PAD || !(extent > 0.f) || (x_stats && (!q_seg_off || n_seg < 1)))
        return RG_ERR_ARG;
    if (nq == 0) return RG_OK;
    GatherArgs g{q_xyz, s_xyz, nbr, x, flag, kernel_points, wf, num, (const float2*)x_stats, q_seg_off,
                 nq, ns, H, Cin, KP, n_seg, extent, slope};
    hipStream_t st = (hipStream_t)stream;
    if (Cin == 1) {
        if (x_stats) return RG_ERR_ARG;
        const size_t lds1 = (size_t)GATHER_WAVES * 4 * H * 5 * sizeof(float);
        k_kpconv_gather_c1<<<rg_cdiv(nq, GATHER_WAVES * 4), GATHER_WAVES * RG_WAVE, lds1, st>>>(g);
        RG_RETURN_IF_LAUNCH_FAILED();
        return RG_OK;
    }
    if (Cin >= 16 && H <= 64 && ns > 0 && (long long)ns * Cin < (1LL << 30) && (long long)nq * H < (1LL << 31)) {   // matrix-core path
        const int grid_m = rg_cdiv(nq, GATHER_WAVES * MG_QPW);
        const int cb = Cin > 32 ? 4 : (Cin > 16 ? 2 : 1);
        const int J = H <= 40 ? 10 : (H <= 52 ? 13 : 16);
#define RG_LAUNCH_MG(JJ, CC) k_kpconv_gather_mfma<JJ, CC><<<grid_m, GATHER_WAVES * RG_WAVE, 0, st>>>(g)
        if (J == 10) { if (cb == 4) RG_LAUNCH_MG(10, 4); else if (cb == 2) RG_LAUNCH_MG(10, 2); else RG_LAUNCH_MG(10, 1); }
        else if (J == 13) { if (cb == 4) RG_LAUNCH_MG(13, 4); else if (cb == 2) RG_LAUNCH_MG(13, 2); else RG_LAUNCH_MG(13, 1); }
        else { if (cb == 4) RG_LAUNCH_MG(16, 4); else if (cb == 2) RG_LAUNCH_MG(16, 2); else RG_LAUNCH_MG(16, 1); }
#undef RG_LAUNCH_MG
        RG_RETURN_IF_LAUNCH_FAILED();
        return RG_OK;
    }
    const int LQ = Cin <= 16 ? 16 : (Cin <= 32 ? 32 : 64);
    const int QW = RG_WAVE / LQ;
    const size_t lds = (size_t)GATHER_WAVES * ((QW * H * (KP_PAD + 6) + 3) & ~3) * sizeof(float);
    if (lds > 160 * 1024) return RG_ERR_ARG;
    const int grid = rg_cdiv(nq, GATHER_WAVES * QW);
    if (LQ == 16) k_kpconv_gather<16><<<grid, GATHER_WAVES * RG_WAVE, lds, st>>>(g);
    else if (LQ == 32) k_kpconv_gather<32><<<grid, GATHER_WAVES * RG_WAVE, lds, st>>>(g);
    else k_kpconv_gather<64><<<grid, GATHER_WAVES * RG_WAVE, lds, st>>>(g);
    RG_RETURN_IF_LAUNCH_FAILED();
    return RG_OK;
}

int regtr_maxpool_gather(const float* x, int ns, int C, const int* nbr, int nq, int H, float* out, void* stream)
{
    if (!x || !nbr || !out || ns < 0 || nq < 0 || H < 1 || C < 4 || C % 4) return RG_ERR_ARG;
    if (nq == 0) return RG_OK;
    k_maxpool_gather<<<rg_cdiv(nq, 4), 256, 0, (hipStream_t)stream>>>(x, ns, C, nbr, nq, H, out);
    RG_RETURN_IF_LAUNCH_FAILED();
    return RG_OK;
}

}  // extern "C"
