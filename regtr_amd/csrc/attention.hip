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
    int* status;      // optional: REGTR_STATUS_F16_RANGE is OR-ed in when the f16 pair core (precision 3) produced a non-finite output
};

// accumulator register r of half-wave `hi` holds matrix row  (r & 3) + 8 * (r >> 2) + 4 * hi
__device__ __forceinline__ int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// KS = key splits: KS waves per workgroup share ONE 32-query tile, wave w streaming key tiles w, w + KS, ... with its own running
// (max, sum, O) and its own LDS tiles; the partial results are merged through LDS in wave order (flash-decoding style).  KS = 4 is the
// launcher's choice for a pair or two per forward, where a launch is only ~200 single-wave workgroups and its duration is one wave's
// serial walk over all keys: 32 -> 13 us per launch at one 3DMatch pair, twelve launches per forward.
template <int KS>
__global__ void __launch_bounds__(KS * RG_WAVE) __attribute__((amdgpu_waves_per_eu(2))) k_mha_fwd(MhaArgs g)
{
    __shared__ float KVs[KS][2 * TK * LDS_STRIDE];
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* Ks = KVs[wave];
    float* Vs = KVs[wave] + TK * LDS_STRIDE;
    const int cloud = blockIdx.z, head = blockIdx.y;
    const int q_begin = g.seg_off[cloud], q_end = g.seg_off[cloud + 1];
    const int q0 = q_begin + blockIdx.x * TQ;
    if (q0 >= q_end) return;                                        // (workgroup-uniform)
    const int kc = g.kv_of[cloud];
    const int k_begin = g.seg_off[kc], nk = g.seg_off[kc + 1] - k_begin;
    const int hoff = head * HD;

    // B operand of S^T = K Q^T : lane holds Q[q0 + l31][2 s + hi] * scale for s = 0..15
    float qreg[16];
    {
        const int qrow = q0 + l31;
        const bool live = qrow < q_end;
        // branch free (clamped row, scale 0 for a dead lane): `live ? load : 0` is sixteen predicated loads with a full wait each
        const float* qp = g.q + (size_t)(live ? qrow : q_begin) * g.ldq + hoff + hi;
        const float scl = live ? g.scale : 0.f;
#pragma unroll
        for (int s = 0; s < 16; s++) qreg[s] = qp[2 * s] * scl;
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
            // clamped rows, no predicate (see k_mha_fwd_bf16): keys past nk are masked to -inf below, their values meet p = 0
            const int krow = min(kt + it * 8 + (lane >> 3), nk > 0 ? nk - 1 : 0), c4 = (lane & 7) * 4;
            const int kb = nk > 0 ? k_begin : q_begin;
            kreg[it] = *(const float4*)(g.k + (size_t)(kb + krow) * g.ldk + hoff + c4);
            vreg[it] = *(const float4*)(g.v + (size_t)(kb + krow) * g.ldv + hoff + c4);
        }
    };
    // (the tiles are private to the wave: wave-level barriers order its LDS writes and reads; the waves' trip counts differ)
    fetch(wave * TK);
    for (int kt = wave * TK; kt < nk; kt += KS * TK) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int row = it * 8 + (lane >> 3), c4 = (lane & 7) * 4;
            float* kd = &Ks[row * LDS_STRIDE + c4];
            kd[0] = kreg[it].x; kd[1] = kreg[it].y; kd[2] = kreg[it].z; kd[3] = kreg[it].w;
            float* vd = &Vs[row * LDS_STRIDE + c4];
            vd[0] = vreg[it].x; vd[1] = vreg[it].y; vd[2] = vreg[it].z; vd[3] = vreg[it].w;
        }
        __builtin_amdgcn_wave_barrier();
        fetch(kt + KS * TK);             // unconditional: rows are clamped

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

    if (KS > 1) {
        // merge the waves' partial (m, l, O) in wave order: m = max m_w, l = sum l_w e^(m_w - m), O = sum O_w e^(m_w - m).  A wave
        // that saw no key carries m = -inf, l = 0, O = 0 and is skipped (e^(-inf - (-inf)) would be NaN).
        __builtin_amdgcn_wave_barrier();
        float* mine = KVs[wave];                                    // [16][64] O | [64] m | [64] l : 4608 B of the wave's 8448
        if (wave > 0) {
#pragma unroll
            for (int r = 0; r < 16; r++) mine[r * 64 + lane] = o[r];
            mine[16 * 64 + lane] = m_run;
            mine[17 * 64 + lane] = l_run;
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll 1
        for (int w = 1; w < KS; w++) {
            const float* p = KVs[w];
            const float mw = p[16 * 64 + lane], lw = p[17 * 64 + lane];
            if (mw == -INFINITY) continue;                          // (per lane: both halves of a query column agree)
            const float m_new = fmaxf(m_run, mw);
            const float a0 = expf(m_run - m_new), a1 = expf(mw - m_new);          // m_run = -inf: a0 = 0, o and l_run are 0 anyway
            l_run = l_run * a0 + lw * a1;
#pragma unroll
            for (int r = 0; r < 16; r++) o[r] = o[r] * a0 + p[r * 64 + lane] * a1;
            m_run = m_new;
        }
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
// The same attention on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, 16x the f32-MFMA rate on gfx950), float32
// softmax and accumulation.  NP = number of bf16 planes every operand is split into:
//   NP = 3  float32-grade: q, k, v and the probabilities are EXACT sums of three bf16 (x = x0 + x1 + x2) and each product is
//           six MFMAs, smallest terms first (see gemm_x3.hip) -- the default; 2.7x fewer matrix-pipe cycles than the f32 MFMA;
//   NP = 1  plain bf16 operands (cfg.compute_dtype = 'bf16', BASELINE configs[1]): one MFMA per product.
// Workgroup = 4 waves = four 32-query tiles of one (cloud, head) sharing every K / V tile: the 256 threads load one float4 of K
// and two float2 of V each, split them once, and stage them into double-buffered LDS (K row-major, V TRANSPOSED -- the A
// operand of O^T += V^T P^T is 8 consecutive contraction slots of one head channel); the next tile's global loads are issued
// before the MFMAs of the current one; one barrier per tile.  LDS rows are 64 B (32 bf16) with the four 16-byte chunks of
// row r at chunk ^ ((r >> 2) & 3): conflict-free ds_read_b128 fragments.  As in k_mha_fwd, S^T = K Q^T leaves every lane with
// ONE query's scores, and P^T goes back in as the B operand without leaving registers: contraction slot j of k-step ks of
// half-wave hi is the key accumulator register 8 ks + j of that half-wave belongs to, so V^T is stored with its key columns
// permuted accordingly (bits 2 and 3 of the key index swapped).  Scores carry log2(e): softmax uses the hardware exp2.
// ------------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int BW = 4;                 // waves (32-query tiles) per workgroup; clouds of more than 128 tokens: BW8 = 8 (round 6, below)
constexpr int BW8 = 8;
constexpr int BROW = 64;              // bytes per LDS row

__device__ __forceinline__ unsigned bf_pack(float a, float b)
{
    bf16x2v v;
    v.x = (__bf16)a; v.y = (__bf16)b;
    return __builtin_bit_cast(unsigned, v);
}
// F16 (NP = 2): the f16 pair of gemm_x3.hip -- x = h0 + h1 / 2048, h0 = f16(x), h1 = f16((x - h0) * 2048): 22 mantissa bits in two planes,
// three MFMA terms per product (the two low ones in a second accumulator, scaled by 1 / 2048 where it is read)
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
constexpr float MHA_F16_SCALE = 2048.f;
// development A-B switches (REGTR_VARIANT_FLAGS): measured on one box with tools/mha_bench.py (128 clouds of ~295 tokens), per launch:
// accumulators in AGPRs (no register budget) 132-135 us, in VGPRs 126-130 us, + the last-tile-only key mask and the
// v_permlane32_swap exchange 122-125 us.  (Vector conversions -- v_cvt_pk_f16_f32 -- for the pair split: no change, hipcc's SLP pass
// already packs most of the scalar form.)
#ifndef MHA_WAVES_ATTR
#define MHA_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(3)))
#endif
#ifndef MHA_PACKED
#define MHA_PACKED 0                  // softmax / pair-split arithmetic on two-element vectors (v_pk_*_f32): measured SLOWER (below); A/B: -DMHA_PACKED=1
#endif
#ifndef MHA_WIDE_MIN_WG
#define MHA_WIDE_MIN_WG 4096         // eight-wave workgroups only on launches with at least this many of them (A/B: -DMHA_WIDE_MIN_WG=..)
#endif
#ifndef MHA_WIDE
#define MHA_WIDE 1                    // 8-wave workgroups for clouds of more than 128 tokens (A/B: -DMHA_WIDE=0)
#endif
#ifndef MHA_OPT_SWAP
#define MHA_OPT_SWAP 1
#endif
#ifndef MHA_OPT_MASK
#define MHA_OPT_MASK 1
#endif
__device__ __forceinline__ unsigned f16_pack(float a, float b)
{
    f16x2v v;
    v.x = (_Float16)a; v.y = (_Float16)b;
    return __builtin_bit_cast(unsigned, v);
}
// (a, b) -> NP packed bf16 pairs; for NP = 3: a = a0 + a1 + a2 exactly (likewise b)
template <int NP, bool F16 = false>
__device__ __forceinline__ void bf_split2(float a, float b, unsigned (&p)[NP])
{
    if constexpr (F16) {
        static_assert(NP == 2, "the f16 pair has two planes");
        p[0] = f16_pack(a, b);
        const f16x2v h = __builtin_bit_cast(f16x2v, p[0]);
#if MHA_PACKED
        const f32x2 r = (f32x2{a, b} - f32x2{(float)h.x, (float)h.y}) * f32x2{MHA_F16_SCALE, MHA_F16_SCALE};      // (v_pk_add_f32 + v_pk_mul_f32)
        p[1] = f16_pack(r.x, r.y);
#else
        p[1] = f16_pack((a - (float)h.x) * MHA_F16_SCALE, (b - (float)h.y) * MHA_F16_SCALE);
#endif
        return;
    }
    p[0] = bf_pack(a, b);
    if (NP > 1) {
        const float ra = a - __uint_as_float(p[0] << 16), rb = b - __uint_as_float(p[0] & 0xffff0000u);
        p[1] = bf_pack(ra, rb);
        if (NP > 2) p[2] = bf_pack(ra - __uint_as_float(p[1] << 16), rb - __uint_as_float(p[1] & 0xffff0000u));
    }
}
// 8 floats -> NP fragments of 8 bf16
template <int NP, bool F16 = false>
__device__ __forceinline__ void bf_split8(const float (&x)[8], bf16x8 (&f)[NP])
{
    unsigned w[4][NP];
#pragma unroll
    for (int i = 0; i < 4; i++) bf_split2<NP, F16>(x[2 * i], x[2 * i + 1], w[i]);
#pragma unroll
    for (int p = 0; p < NP; p++) f[p] = __builtin_bit_cast(bf16x8, make_uint4(w[0][p], w[1][p], w[2][p], w[3][p]));
}
// A x B with both operands split: the NP (NP + 1) / 2 terms whose exponent sum stays below NP, smallest first
template <int NP>
__device__ __forceinline__ floatx16 bf_mma(const bf16x8 (&a)[NP], const bf16x8 (&b)[NP], floatx16 c)
{
    if (NP == 3) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
    }
    if (NP >= 2) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
    }
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
}

// f16 pair product: hi += a0 b0, lo += a1 b0 + a0 b1 (alternating accumulators)
__device__ __forceinline__ void f16_mma(const bf16x8 (&a)[2], const bf16x8 (&b)[2], floatx16& hi, floatx16& lo)
{
    lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8v, a[1]), __builtin_bit_cast(f16x8v, b[0]), lo, 0, 0, 0);
    hi = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8v, a[0]), __builtin_bit_cast(f16x8v, b[0]), hi, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8v, a[0]), __builtin_bit_cast(f16x8v, b[1]), lo, 0, 0, 0);
}

// combine a value of lane l with that of lane l ^ 32 (the two half-waves hold complementary key sets of the same 32 queries):
// v_permlane32_swap_b32 exchanges the upper half of one register with the lower half of the other, so after swapping x with a copy of
// itself the two registers hold (low half's value, high half's value) in EVERY lane -- one VALU instruction where __shfl_xor goes
// through ds_bpermute_b32 and a wait on LDS.  max and + are commutative: both halves get bit-identical results.
__device__ __forceinline__ float mha_max_halves(float x)
{
#if !MHA_OPT_SWAP
    return fmaxf(x, __shfl_xor(x, 32, RG_WAVE));
#endif
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float mha_sum_halves(float x)
{
#if !MHA_OPT_SWAP
    return x + __shfl_xor(x, 32, RG_WAVE);
#endif
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// NW (round 6) = waves per workgroup, 4 or 8.  Every workgroup of a (cloud, head) stages ALL of the cloud's K / V tiles, so the split + staging
// work per query tile -- ~900 of the ~3200 cycles a wave spends per key tile in the f16 pair form -- is divided by the waves that share it: with
// eight waves a thread stages one float2 of K and one key pair of ONE V channel (two split calls, one 4-byte LDS store per plane each) instead of a
// float4 and two channels.  Same registers per wave, same 16 KB of LDS, two workgroups per CU instead of four; identical arithmetic per query, so
// the outputs are bit-identical to the four-wave form.  The launcher takes eight waves when a cloud needs more than one 4-wave workgroup anyway.
template <int NP, bool F16 = false, int NW = BW>
// (waves_per_eu: with a register budget of at most 256 per lane hipcc keeps the MFMA accumulators in VGPRs; without it the budget is 512, the
//  accumulators go to AGPRs and every softmax step pays v_accvgpr_read / _write copies -- 288 of the 1151 vector instructions of the f16 pair
//  form, and 160 registers instead of 124: three waves per SIMD instead of four)
__global__ void __launch_bounds__(NW * RG_WAVE) MHA_WAVES_ATTR k_mha_fwd_bf16(MhaArgs g)
{
    __shared__ __align__(16) unsigned char Ks[2][NP][TK * BROW];      // [buffer][plane][key][32 channels]
    __shared__ __align__(16) unsigned char Vt[2][NP][HD * BROW];      // [buffer][plane][channel][32 key slots]
    const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int cloud = blockIdx.z, head = blockIdx.y;
    const int q_begin = g.seg_off[cloud], q_end = g.seg_off[cloud + 1];
    if (q_begin + (int)blockIdx.x * NW * TQ >= q_end) return;          // the whole workgroup (before any barrier)
    const int q0 = q_begin + (blockIdx.x * NW + wave) * TQ;
    const bool wave_live = q0 < q_end;                                 // wave-uniform; dead waves still stage K / V
    const int kc = g.kv_of[cloud];
    const int k_begin = g.seg_off[kc], nk = g.seg_off[kc + 1] - k_begin;
    const int hoff = head * HD;

    // B operand of S^T = K Q^T: lane (query l31, half hi) holds Q[q][16 ks + 8 hi + j] * scale * log2(e), j < 8
    bf16x8 qf[2][NP];
    {
        const int qrow = q0 + l31;
        const bool live = wave_live && qrow < q_end;
        const float sc2 = g.scale * 1.44269504088896340736f;
        const float* qp = g.q + (size_t)(live ? qrow : q_begin) * g.ldq + hoff + 8 * hi;
        // (all four 16-byte loads first, from the clamped row, THEN the select: `live ? load : 0` per element compiles to sixteen
        //  predicated dword loads with a full wait each)
        float4 qv[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ks++) { qv[ks][0] = *(const float4*)(qp + 16 * ks); qv[ks][1] = *(const float4*)(qp + 16 * ks + 4); }
        const float scl = live ? sc2 : 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            const float x[8] = {qv[ks][0].x * scl, qv[ks][0].y * scl, qv[ks][0].z * scl, qv[ks][0].w * scl,
                                qv[ks][1].x * scl, qv[ks][1].y * scl, qv[ks][1].z * scl, qv[ks][1].w * scl};
            bf_split8<NP, F16>(x, qf[ks]);
        }
    }

    // staging roles.  NW = 4: K -- thread (key = t / 8, channels 4 (t % 8) ..+3); V -- thread (key pair u = t / 16, channels 2 (t % 16), +1)
    //                 NW = 8: K -- thread (key = t / 16, channels 2 (t % 16), +1);  V -- thread (key pair u = t / 32, channel t % 32)
    constexpr int KPT = NW == 4 ? 4 : 2;                             // K channels per thread
    const int skey = t / (32 / KPT), sdk = (t % (32 / KPT)) * KPT;
    const int su = NW == 4 ? t >> 4 : t >> 5, sd2 = NW == 4 ? (t & 15) * 2 : (t & 31);
    const unsigned k_dst = (unsigned)skey * BROW + (((unsigned)(sdk >> 3) ^ (((unsigned)skey >> 2) & 3u)) * 16u) + ((unsigned)sdk & 7u) * 2u;
    const unsigned vpos = ((2u * su) & 0x13u) | (((2u * su) & 4u) << 1) | (((2u * su) & 8u) >> 1);     // slot of key 2 u (even)
    unsigned v_dst[2];
#pragma unroll
    for (int dd = 0; dd < 2; dd++) {
        const unsigned row = (unsigned)(sd2 + dd);
        v_dst[dd] = row * BROW + (((vpos >> 3) ^ ((row >> 2) & 3u)) * 16u) + (vpos & 7u) * 2u;
    }
    // K / V rows travel global -> registers -> (split) -> LDS.  TWO register sets: the rows of tile t + 2 are requested while tile t
    // is multiplied and tile t + 1's set -- requested a whole tile earlier -- is split and stored; with one set the store pass waited
    // ~700 cycles per tile for loads issued one compute phase before (phase clocks, profiles/r03_mha_phase_clocks.md).
    struct KV { float4 k; float2 v0, v1; };             // (NW = 8: k.x, k.y and v0.x, v1.x only)
    // BRANCH-FREE loads from clamped rows: a predicated load (`x = 0; if (row < nk) x = load`) becomes load + select, and the select's
    // wait exposes the whole L2 latency right at the issue point -- phase clocks showed 1575 of 4800 cycles per key tile there.  Keys past
    // nk need no zeroing: their scores are masked to -inf below, so their (finite, clamped-row) values meet p = 0.
    const int nk1 = nk > 0 ? nk - 1 : 0;
    const int kb = nk > 0 ? k_begin : q_begin;         // an empty key cloud: read (and never use) a row of the live query cloud
    auto fetch = [&](KV& r, int kt) {
        const int kr = min(kt + skey, nk1), v0 = min(kt + 2 * su, nk1), v1 = min(kt + 2 * su + 1, nk1);
        if constexpr (NW == 4) {
            r.k = *(const float4*)(g.k + (size_t)(kb + kr) * g.ldk + hoff + sdk);
            r.v0 = *(const float2*)(g.v + (size_t)(kb + v0) * g.ldv + hoff + sd2);
            r.v1 = *(const float2*)(g.v + (size_t)(kb + v1) * g.ldv + hoff + sd2);
        } else {
            const float2 k2 = *(const float2*)(g.k + (size_t)(kb + kr) * g.ldk + hoff + sdk);
            r.k.x = k2.x; r.k.y = k2.y;
            r.v0.x = g.v[(size_t)(kb + v0) * g.ldv + hoff + sd2];
            r.v1.x = g.v[(size_t)(kb + v1) * g.ldv + hoff + sd2];
        }
    };
    auto stage = [&](const KV& r, int buf) {
        unsigned a[NP], b[NP];
        bf_split2<NP, F16>(r.k.x, r.k.y, a);
        if constexpr (NW == 4) {
            bf_split2<NP, F16>(r.k.z, r.k.w, b);
#pragma unroll
            for (int p = 0; p < NP; p++) *(uint2*)(&Ks[buf][p][k_dst]) = make_uint2(a[p], b[p]);
        } else {
#pragma unroll
            for (int p = 0; p < NP; p++) *(unsigned*)(&Ks[buf][p][k_dst]) = a[p];
        }
        bf_split2<NP, F16>(r.v0.x, r.v1.x, a);        // (key 2u, key 2u + 1) of channel sd2
        if constexpr (NW == 4) bf_split2<NP, F16>(r.v0.y, r.v1.y, b);        // ... of channel sd2 + 1
#pragma unroll
        for (int p = 0; p < NP; p++) {
            *(unsigned*)(&Vt[buf][p][v_dst[0]]) = a[p];
            if constexpr (NW == 4) *(unsigned*)(&Vt[buf][p][v_dst[1]]) = b[p];
        }
    };

    floatx16 o, o_lo;                                  // (o_lo: the f16 pair's scaled low terms of O)
#pragma unroll
    for (int r = 0; r < 16; r++) { o[r] = 0.f; o_lo[r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;
    unsigned f_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ks++) f_off[ks] = (unsigned)l31 * BROW + ((((unsigned)(2 * ks + hi)) ^ (((unsigned)l31 >> 2) & 3u)) * 16u);

// development A-B (variant build -DMHA_SETPRIO=1): raise the wave's issue priority around its MFMA groups -- the explicit form of the
// MFMA / VALU "ping-pong" between the 3 waves a SIMD holds (144 VGPRs + 16 AGPRs, 24 KB of LDS per workgroup: three workgroups per CU,
// so the hardware already interleaves one wave's MFMAs with its neighbours' softmax).  Measured on a 64-pair forward, alternating
// runs on one box: 165 us per launch against 155-156 us without (gpurun_out/r03_y4): the kernel is VALU-bound (per 32-key tile and
// wave 768 matrix-pipe cycles against ~1500 of softmax / split arithmetic + ~900 of K/V split and staging), and prioritising the
// pipe that has slack delays the one that has none.  Off.
#ifdef MHA_SETPRIO
#define MHA_PRIO(p) __builtin_amdgcn_s_setprio(p)
#else
#define MHA_PRIO(p)
#endif
#ifdef MHA_PROF          // development: per-wave phase clocks (REGTR_VARIANT_FLAGS=-DMHA_PROF)
    long long pt[6] = {0, 0, 0, 0, 0, 0}, pc = clock64(), pstart = pc;
#define MHA_STAMP(I) do { const long long n_ = clock64(); pt[I] += n_ - pc; pc = n_; } while (0)
#else
#define MHA_STAMP(I) do {} while (0)
#endif
    KV ra, rb;
    fetch(ra, 0);
    stage(ra, 0);
    fetch(ra, TK);                                     // tile 1 (clamped rows past the end: never used)
    __syncthreads();
    MHA_STAMP(0);
    // one key tile: request tile kt + 2 into `rnew`, multiply tile kt from LDS slot `buf`, split + store tile kt + 1 (`rold`, requested a
    // tile ago) into the other slot.  The fetch is unconditional (rows are clamped): under `if (more)` the loop-carried registers get phi
    // copies placed right behind the loads, and a wait for them at the issue point.
    auto tile = [&](int kt, int buf, KV& rold, KV& rnew) {
        const bool more = kt + TK < nk;                // workgroup-uniform
        fetch(rnew, kt + 2 * TK);
        MHA_STAMP(1);
        if (wave_live) {
            floatx16 sc, sc_lo;
#pragma unroll
            for (int r = 0; r < 16; r++) { sc[r] = 0.f; sc_lo[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                bf16x8 kf[NP];
#pragma unroll
                for (int p = 0; p < NP; p++) kf[p] = __builtin_bit_cast(bf16x8, *(const uint4*)(&Ks[buf][p][f_off[ks]]));
                MHA_PRIO(2);
                if constexpr (F16) f16_mma(kf, qf[ks], sc, sc_lo);
                else sc = bf_mma<NP>(kf, qf[ks], sc);
                MHA_PRIO(0);
            }
            // (PAIRS, round 6: written on two-element vectors so that hipcc emits the packed-f32 instructions -- v_pk_fma_f32 / v_pk_add_f32 do two
            //  lanes' worth of elements per issue slot.  The scalar form compiled to 17 v_sub + 17 v_add (+ 16 v_fma in the f16 pair form) per key
            //  tile next to 17 quarter-rate v_exp: the softmax is what bounds this kernel at head dimension 32.)
            if constexpr (F16) {
#if MHA_PACKED
                const f32x2 lsc = {1.0f / MHA_F16_SCALE, 1.0f / MHA_F16_SCALE};
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 v = f32x2{sc_lo[r], sc_lo[r + 1]} * lsc + f32x2{sc[r], sc[r + 1]};
                    sc[r] = v.x; sc[r + 1] = v.y;
                }
#else
#pragma unroll
                for (int r = 0; r < 16; r++) sc[r] += sc_lo[r] * (1.0f / MHA_F16_SCALE);
#endif
            }
            MHA_STAMP(2);
            if (!MHA_OPT_MASK || kt + TK > nk) {       // (workgroup-uniform: only the last tile of a cloud has keys past the end)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    if (kt + acc_row(r, hi) >= nk) sc[r] = -INFINITY;
            }
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; r++) mx = fmaxf(mx, sc[r]);
            mx = mha_max_halves(mx);
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            float pr[16];
#if MHA_PACKED
            f32x2 ps2 = {0.f, 0.f};
            const f32x2 mm = {m_new, m_new};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 d = f32x2{sc[r], sc[r + 1]} - mm;
                pr[r] = __builtin_amdgcn_exp2f(d.x);
                pr[r + 1] = __builtin_amdgcn_exp2f(d.y);
                ps2 += f32x2{pr[r], pr[r + 1]};
            }
            float psum = mha_sum_halves(ps2.x + ps2.y);
#else
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r++) { pr[r] = __builtin_amdgcn_exp2f(sc[r] - m_new); psum += pr[r]; }
            psum = mha_sum_halves(psum);
#endif
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int r = 0; r < 16; r++) { o[r] *= alpha; if (F16) o_lo[r] *= alpha; }
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; j++) x[j] = pr[8 * ks + j];
                bf16x8 pf[NP], vf[NP];
                bf_split8<NP, F16>(x, pf);
#pragma unroll
                for (int p = 0; p < NP; p++) vf[p] = __builtin_bit_cast(bf16x8, *(const uint4*)(&Vt[buf][p][f_off[ks]]));
                MHA_PRIO(2);
                if constexpr (F16) f16_mma(vf, pf, o, o_lo);
                else o = bf_mma<NP>(vf, pf, o);
                MHA_PRIO(0);
            }
            MHA_STAMP(3);
        }
        if (more) stage(rold, buf ^ 1);
        MHA_STAMP(4);
        __syncthreads();
        MHA_STAMP(5);
    };
    for (int kt = 0; kt < nk; kt += 2 * TK) {
        tile(kt, 0, ra, rb);
        if (kt + TK >= nk) break;
        tile(kt + TK, 1, rb, ra);
    }
#ifdef MHA_PROF
    if ((blockIdx.z == 3 || blockIdx.z == 77) && blockIdx.y == 2 && lane == 0)
        printf("mha NP %d cloud %d wg %d wave %d live %d nk %d: prologue %lld fetch-issue %lld qk %lld softmax+pv %lld stage %lld barrier %lld total %lld (cycles)\n", NP,
               (int)blockIdx.z, (int)blockIdx.x, wave, (int)wave_live, nk, pt[0], pt[1], pt[2], pt[3], pt[4], pt[5], clock64() - pstart);
#endif

    if constexpr (F16) {
        // a q / k / v value beyond f16's range converts to Inf (its residual to NaN): scores, and with them the outputs, come out
        // non-finite -- reported through the status word (x * 0 is NaN exactly for a non-finite x)
        float chk = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) { o[r] += o_lo[r] * (1.0f / MHA_F16_SCALE); chk = fmaf(o[r], 0.f, chk); }
        chk = fmaf(l_run, 0.f, chk);
        if (g.status && wave_live && chk != chk) atomicOr(g.status, REGTR_STATUS_F16_RANGE);
    }
    const int qrow = q0 + l31;
    if (wave_live && qrow < q_end) {
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;   // empty key set -> zeros
        float* dst = g.out + (size_t)qrow * g.ldo + hoff;
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) {
            const int d = 8 * r4 + 4 * hi;
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
__global__ void __launch_bounds__(RG_WAVE) __attribute__((amdgpu_waves_per_eu(2))) k_attn_xyz(AttnXyzArgs g)
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
        // branch free (clamped row, scale 0 for a dead lane): `live ? load : 0` compiles to a load + full wait per element
        const float* qp = Q + (size_t)(live ? qrow : q_begin) * HDX + hi;
        const float scl = live ? g.scale : 0.f;
#pragma unroll
        for (int s = 0; s < HDX / 2; s++) qreg[s] = qp[2 * s] * scl;
    }
    floatx16 o;
#pragma unroll
    for (int r = 0; r < 16; r++) o[r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    for (int e = lane; e < TK * LDS_STRIDE; e += RG_WAVE) Vs[e] = 0.f;      // columns >= 3 stay zero for the whole kernel

    // Key rows and coordinates come from CLAMPED rows with no predicate, eight rows' loads in flight (keys past nk are masked to
    // -inf below, so their finite clamped values meet p = 0); lanes beyond the row's HDX floats re-read its last float4 and store nothing.
    constexpr int LPR = HDX / 4 < RG_WAVE ? HDX / 4 : RG_WAVE;        // lanes that own a float4 of a key row
    const int c4 = (lane < LPR ? lane : LPR - 1) * 4;
    const int nk1 = nk > 0 ? nk - 1 : 0;
    const int kb = nk > 0 ? k_begin : q_begin;                        // an empty key cloud: read (and mask) rows of the query cloud
    for (int kt = 0; kt < nk; kt += TK) {
        __syncthreads();
#pragma unroll 1
        for (int row0 = 0; row0 < TK; row0 += 8) {
            float4 v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = *(const float4*)(K + (size_t)(kb + min(kt + row0 + i, nk1)) * HDX + c4);
            if (lane < LPR) {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    float* kd = &Ks[(row0 + i) * KS + c4];
                    kd[0] = v[i].x; kd[1] = v[i].y; kd[2] = v[i].z; kd[3] = v[i].w;
                }
            }
        }
        {
            const float* xp = g.xyz + (size_t)(kb + min(kt + l31, nk1)) * 3;
            const float x = xp[0], y = xp[1], z = xp[2];
            if (hi == 0) { Vs[l31 * LDS_STRIDE] = x; Vs[l31 * LDS_STRIDE + 1] = y; Vs[l31 * LDS_STRIDE + 2] = z; }
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
// precision: 0 = float32-grade on the bf16 matrix cores (three-way split operands), 1 = plain bf16 operands
// (float32 softmax / accumulation), 2 = the exact-f32 MFMA kernel (one wave per 32-query tile; the A/B reference), 3 = float32-grade by
// the f16 pair split (two planes, three MFMA terms; operands below 65504: q, k, v are projections of LayerNorm outputs, p <= 1).
int regtr_mha_fwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out, int ldo,
                  const int* seg_off, const int* kv_of, int n_clouds, int max_len, int n_heads, int head_dim, float scale,
                  int precision, int* status, void* stream)
{
    if (!q || !k || !v || !out || !seg_off || !kv_of || n_clouds < 1 || n_heads < 1 || max_len < 0) return RG_ERR_ARG;
    if (head_dim != HD || precision < 0 || precision > 3) return RG_ERR_ARG;
    if ((ldq | ldk | ldv | ldo) % 4 || (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) % 16)) return RG_ERR_ARG;
    if (max_len == 0) return RG_OK;
    MhaArgs g{q, k, v, out, seg_off, kv_of, ldq, ldk, ldv, ldo, n_heads, scale, status};
    hipStream_t st = (hipStream_t)stream;
    // Small problems (a pair or two per forward: fewer than two 4-wave workgroups per CU): the single-wave exact-f32 kernel puts
    // four times as many workgroups on the chip and stages K / V without the split -- 32 us against 56 us per launch at one
    // 3DMatch pair.  Both are float32-grade, so precision 0 may take either.
    const bool small = (long long)rg_cdiv(max_len, BW * TQ) * n_heads * n_clouds < 512;
    if (precision == 2 || ((precision == 0 || precision == 3) && small)) {
        const dim3 grid(rg_cdiv(max_len, TQ), n_heads, n_clouds);
        // very small launches (one pair: ~200 query tiles on 256 CUs): four waves split the keys of a tile
        if ((long long)grid.x * grid.y * grid.z <= 1024 && max_len > 4 * TK) k_mha_fwd<4><<<grid, 4 * RG_WAVE, 0, st>>>(g);
        else k_mha_fwd<1><<<grid, RG_WAVE, 0, st>>>(g);
    }
    else {
        if ((ldv % 2) || ((uintptr_t)v % 8)) return RG_ERR_ARG;
        // clouds of more than 128 tokens need two 4-wave workgroups per head anyway: eight waves share every K / V tile instead (k_mha_fwd_bf16)
        // -- on launches of many rounds of workgroups only.  Measured (tools/mha_bench.py, profiles/r06_f_mha_wide.txt; us per launch, 4 -> 8 waves):
        // 384 clouds of 330-460 tokens, f16 pair 439.7 -> 419.5, bf16x3 553.7 -> 515.3; 512 clouds of 560-640, bf16 572.9 -> 554.9 (inside the
        // ModelNet forward 547 -> 464); but 128 clouds of 230-360 (2048 eight-wave workgroups, four rounds on 256 CUs x 2) 93.9 -> 110.8.
        if (MHA_WIDE && max_len > BW * TQ && (long long)rg_cdiv(max_len, BW8 * TQ) * n_heads * n_clouds >= MHA_WIDE_MIN_WG) {
            const dim3 grid(rg_cdiv(max_len, BW8 * TQ), n_heads, n_clouds);
            if (precision == 0) k_mha_fwd_bf16<3, false, BW8><<<grid, BW8 * RG_WAVE, 0, st>>>(g);
            else if (precision == 3) k_mha_fwd_bf16<2, true, BW8><<<grid, BW8 * RG_WAVE, 0, st>>>(g);
            else k_mha_fwd_bf16<1, false, BW8><<<grid, BW8 * RG_WAVE, 0, st>>>(g);
        } else {
            const dim3 grid(rg_cdiv(max_len, BW * TQ), n_heads, n_clouds);
            if (precision == 0) k_mha_fwd_bf16<3><<<grid, BW * RG_WAVE, 0, st>>>(g);
            else if (precision == 3) k_mha_fwd_bf16<2, true><<<grid, BW * RG_WAVE, 0, st>>>(g);
            else k_mha_fwd_bf16<1><<<grid, BW * RG_WAVE, 0, st>>>(g);
        }
    }
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
