// Multi-head attention core (scaled QK^T -> softmax -> AV) on the CDNA4 matrix cores, float32 in / float32
// accumulate (v_mfma_f32_32x32x2_f32).  Implements the arithmetic of nn.MultiheadAttention between the in- and
// out-projections for the four attention calls of a RegTR cross-encoder layer
//   /root/reference/src/models/transformer/transformers.py:197-201,206-210 (self) and :217-226 (cross)
// on PACKED variable-length sequences: cloud c's queries attend the keys/values of cloud kv_of[c]
// (self: kv_of[c] = c ; cross: the partner cloud of the pair).  No padding, no key_padding_mask needed.
//
// One 64-lane wavefront owns a 32-query tile of one head and streams 32-key tiles, flash style:
//   S^T = K Q^T is computed with K as the A operand and the (pre-scaled) queries as B, so each lane ends up with the
//   scores of ONE query row (column of S^T) for 16 keys; the two half-waves hold complementary key sets, so the row
//   max / row sum need 15 in-register ops and a single cross-half exchange -- no LDS round trip for the softmax.
//   P^T is then fed straight back as the B operand of O^T += V^T P^T: step s of the contraction uses the key that
//   accumulator register s of this half-wave already corresponds to, so the probabilities never leave registers.
// K / V tiles are staged through LDS with an odd row stride (33) so both fragment read patterns are conflict free.
#include "common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int HD = 32;       // head dimension handled by this kernel
constexpr int TQ = 32, TK = 32;
constexpr int LDS_STRIDE = HD + 1;

struct MhaArgs {
    const float* q; const float* k; const float* v;   // row-major, leading dims ldq / ldk / ldv
    float* out;
    const int* seg_off; const int* kv_of;
    int ldq, ldk, ldv, ldo, n_heads;
    float scale;
};

// accumulator register r of half-wave `hi` holds matrix row  (r & 3) + 8 * (r >> 2) + 4 * hi
__device__ __forceinline__ int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

__global__ void __launch_bounds__(RG_WAVE) k_mha_fwd(MhaArgs g)
{
    __shared__ float Ks[TK * LDS_STRIDE];
    __shared__ float Vs[TK * LDS_STRIDE];
    const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5;
    const int cloud = blockIdx.z, head = blockIdx.y;
    const int q_begin = g.seg_off[cloud], q_end = g.seg_off[cloud + 1];
    const int q0 = q_begin + blockIdx.x * TQ;
    if (q0 >= q_end) return;
    const int kc = g.kv_of[cloud];
    const int k_begin = g.seg_off[kc], nk = g.seg_off[kc + 1] - k_begin;
    const int hoff = head * HD;

    // B operand of S^T = K Q^T : lane holds Q[q0 + l31][2 s + hi] * scale for s = 0..15
    float qreg[16];
    {
        const int qrow = q0 + l31;
        const bool live = qrow < q_end;
#pragma unroll
        for (int s = 0; s < 16; s++)
            qreg[s] = live ? g.q[(size_t)qrow * g.ldq + hoff + 2 * s + hi] * g.scale : 0.f;
    }

    floatx16 o;
#pragma unroll
    for (int r = 0; r < 16; r++) o[r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // K / V tiles: 32 rows x 32 floats each; a row is one 128-B line, 8 lanes per row.  The next tile's global loads are
    // issued before the MFMAs of the current one (register staging), so HBM/L2 latency hides under compute.
    float4 kreg[4], vreg[4];
    auto fetch = [&](int kt) {
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int krow = kt + it * 8 + (lane >> 3), c4 = (lane & 7) * 4;
            kreg[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            vreg[it] = kreg[it];
            if (krow < nk) {
                kreg[it] = *(const float4*)(g.k + (size_t)(k_begin + krow) * g.ldk + hoff + c4);
                vreg[it] = *(const float4*)(g.v + (size_t)(k_begin + krow) * g.ldv + hoff + c4);
            }
        }
    };
    fetch(0);
    for (int kt = 0; kt < nk; kt += TK) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int row = it * 8 + (lane >> 3), c4 = (lane & 7) * 4;
            float* kd = &Ks[row * LDS_STRIDE + c4];
            kd[0] = kreg[it].x; kd[1] = kreg[it].y; kd[2] = kreg[it].z; kd[3] = kreg[it].w;
            float* vd = &Vs[row * LDS_STRIDE + c4];
            vd[0] = vreg[it].x; vd[1] = vreg[it].y; vd[2] = vreg[it].z; vd[3] = vreg[it].w;
        }
        __syncthreads();
        if (kt + TK < nk) fetch(kt + TK);

        // S^T tile: rows = keys, cols = queries
        floatx16 sc;
#pragma unroll
        for (int r = 0; r < 16; r++) sc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; s++)
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[l31 * LDS_STRIDE + 2 * s + hi], qreg[s], sc, 0, 0, 0);

        // online softmax for query column l31 (keys acc_row(r, hi) of this tile)
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if (kt + acc_row(r, hi) >= nk) sc[r] = -INFINITY;
            mx = fmaxf(mx, sc[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, RG_WAVE));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            sc[r] = expf(sc[r] - m_new);
            psum += sc[r];
        }
        psum += __shfl_xor(psum, 32, RG_WAVE);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; r++) o[r] *= alpha;

        // O^T += V^T P^T : contraction step s uses key acc_row(s, hi) on both operands
#pragma unroll
        for (int s = 0; s < 16; s++)
            o = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[acc_row(s, hi) * LDS_STRIDE + l31], sc[s], o, 0, 0, 0);
    }

    const int qrow = q0 + l31;
    if (qrow < q_end) {
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;   // empty key set -> zeros
        float* dst = g.out + (size_t)qrow * g.ldo + hoff;
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) {
            const int d = 8 * r4 + 4 * hi;   // acc_row(4 * r4 + j, hi) = j + 8 * r4 + 4 * hi
            *(float4*)(dst + d) = make_float4(o[4 * r4] * inv, o[4 * r4 + 1] * inv, o[4 * r4 + 2] * inv, o[4 * r4 + 3] * inv);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Single-head dot-product attention whose VALUES are coordinates: CorrespondenceDecoder.simple_attention
// (/root/reference/src/models/regtr.py:316-351, the `direct_regress_coor: False` head).  out[l, q, :] =
// softmax_s( Q[l, q, :] . K[l, s, :] * scale ) @ xyz[s, :]  over the keys s of the partner cloud kv_of[cloud(q)], for every
// decoder layer l; Q / K are (L, N, HD) row-major projections of the conditioned features, xyz (N, 3) is shared by the layers.
// Same flash-style structure as k_mha_fwd (S^T = K Q^T with one query column per lane, P^T fed back as the B operand of
// O^T += V^T P^T); the head dimension is a template parameter and V^T has 3 live rows.  A rarely used configuration:
// written for correctness, K tiles are staged without prefetch.
// ------------------------------------------------------------------------------------------------------------------
struct AttnXyzArgs {
    const float* q; const float* k; const float* xyz; float* out;
    const int* seg_off; const int* kv_of;
    int n_total;
    float scale;
};

template <int HDX>
__global__ void __launch_bounds__(RG_WAVE) k_attn_xyz(AttnXyzArgs g)
{
    constexpr int KS = HDX + 1;                       // odd row stride: conflict-free fragment reads
    extern __shared__ float smem_x[];
    float* Ks = smem_x;                               // [TK][HDX + 1]
    float* Vs = smem_x + TK * KS;                     // [TK][LDS_STRIDE], columns >= 3 are zero
    const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5;
    const int cloud = blockIdx.z, layer = blockIdx.y;
    const int q_begin = g.seg_off[cloud], q_end = g.seg_off[cloud + 1];
    const int q0 = q_begin + blockIdx.x * TQ;
    if (q0 >= q_end) return;
    const int kc = g.kv_of[cloud];
    const int k_begin = g.seg_off[kc], nk = g.seg_off[kc + 1] - k_begin;
    const float* Q = g.q + (size_t)layer * g.n_total * HDX;
    const float* K = g.k + (size_t)layer * g.n_total * HDX;

    float qreg[HDX / 2];                              // B operand of S^T = K Q^T: Q[q0 + l31][2 s + hi] * scale
    {
        const int qrow = q0 + l31;
        const bool live = qrow < q_end;
#pragma unroll
        for (int s = 0; s < HDX / 2; s++) qreg[s] = live ? Q[(size_t)qrow * HDX + 2 * s + hi] * g.scale : 0.f;
    }
    floatx16 o;
#pragma unroll
    for (int r = 0; r < 16; r++) o[r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    for (int kt = 0; kt < nk; kt += TK) {
        __syncthreads();
        for (int row = 0; row < TK; row++) {          // one key row (HDX floats) per pass: 64 lanes x 4 floats
            const bool live = kt + row < nk;
            for (int c4 = lane * 4; c4 < HDX; c4 += RG_WAVE * 4) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (live) v = *(const float4*)(K + (size_t)(k_begin + kt + row) * HDX + c4);
                float* kd = &Ks[row * KS + c4];
                kd[0] = v.x; kd[1] = v.y; kd[2] = v.z; kd[3] = v.w;
            }
        }
        for (int e = lane; e < TK * LDS_STRIDE; e += RG_WAVE) {
            const int row = e / LDS_STRIDE, d = e - row * LDS_STRIDE;
            Vs[e] = (d < 3 && kt + row < nk) ? g.xyz[(size_t)(k_begin + kt + row) * 3 + d] : 0.f;
        }
        __syncthreads();

        floatx16 sc;
#pragma unroll
        for (int r = 0; r < 16; r++) sc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < HDX / 2; s++)
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[l31 * KS + 2 * s + hi], qreg[s], sc, 0, 0, 0);

        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if (kt + acc_row(r, hi) >= nk) sc[r] = -INFINITY;
            mx = fmaxf(mx, sc[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, RG_WAVE));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            sc[r] = expf(sc[r] - m_new);
            psum += sc[r];
        }
        psum += __shfl_xor(psum, 32, RG_WAVE);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; r++) o[r] *= alpha;
#pragma unroll
        for (int s = 0; s < 16; s++)
            o = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[acc_row(s, hi) * LDS_STRIDE + l31], sc[s], o, 0, 0, 0);
    }
    // O^T rows 0..2 = x, y, z live in registers 0..2 of the lower half-wave (acc_row(r, 0) = r for r < 4)
    const int qrow = q0 + l31;
    if (qrow < q_end && hi == 0) {
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
        float* dst = g.out + ((size_t)layer * g.n_total + qrow) * 3;
        dst[0] = o[0] * inv; dst[1] = o[1] * inv; dst[2] = o[2] * inv;
    }
}

}  // namespace

extern "C" {

// q, k, v: [N_total, *] row-major views (leading dims ldq/ldk/ldv) holding n_heads * 32 columns each;
// out [N_total, ldo].  seg_off [n_clouds + 1] and kv_of [n_clouds] live on the device.
// max_len = longest query segment (host-known bound used for the launch grid).
int regtr_mha_fwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out, int ldo,
                  const int* seg_off, const int* kv_of, int n_clouds, int max_len, int n_heads, int head_dim, float scale,
                  void* stream)
{
    if (!q || !k || !v || !out || !seg_off || !kv_of || n_clouds < 1 || n_heads < 1 || max_len < 0) return RG_ERR_ARG;
    if (head_dim != HD) return RG_ERR_ARG;
    if ((ldk | ldv | ldo) % 4 || (((uintptr_t)k | (uintptr_t)v | (uintptr_t)out) % 16)) return RG_ERR_ARG;
    if (max_len == 0) return RG_OK;
    MhaArgs g{q, k, v, out, seg_off, kv_of, ldq, ldk, ldv, ldo, n_heads, scale};
    k_mha_fwd<<<dim3(rg_cdiv(max_len, TQ), n_heads, n_clouds), RG_WAVE, 0, (hipStream_t)stream>>>(g);
    RG_RETURN_IF_LAUNCH_FAILED();
    return RG_OK;
}

// CorrespondenceDecoder.simple_attention (regtr.py:316-351): q, k (n_layers, n_total, head_dim) contiguous, xyz (n_total, 3),
// out (n_layers, n_total, 3); cloud c attends the keys / coordinates of cloud kv_of[c].  head_dim in {32, 64, 128, 256}.
int regtr_attn_xyz(const float* q, const float* k, const float* xyz, float* out, const int* seg_off, const int* kv_of,
                   int n_clouds, int n_total, int n_layers, int max_len, int head_dim, float scale, void* stream)
{
    if (!q || !k || !xyz || !out || !seg_off || !kv_of || n_clouds < 1 || n_total < 0 || n_layers < 1 || max_len < 0)
        return RG_ERR_ARG;
    if ((((uintptr_t)q | (uintptr_t)k) % 16)) return RG_ERR_ARG;
    if (max_len == 0 || n_total == 0) return RG_OK;
    AttnXyzArgs g{q, k, xyz, out, seg_off, kv_of, n_total, scale};
    const dim3 grid(rg_cdiv(max_len, TQ), n_layers, n_clouds);
    hipStream_t st = (hipStream_t)stream;
#define RG_ATTN_XYZ(HDX_) k_attn_xyz<HDX_><<<grid, RG_WAVE, (TK * (HDX_ + 1) + TK * LDS_STRIDE) * sizeof(float), st>>>(g)
    if (head_dim == 256) RG_ATTN_XYZ(256);
    else if (head_dim == 128) RG_ATTN_XYZ(128);
    else if (head_dim == 64) RG_ATTN_XYZ(64);
    else if (head_dim == 32) RG_ATTN_XYZ(32);
    else return RG_ERR_ARG;
#undef RG_ATTN_XYZ
    RG_RETURN_IF_LAUNCH_FAILED();
    return RG_OK;
}

}  // extern "C"
