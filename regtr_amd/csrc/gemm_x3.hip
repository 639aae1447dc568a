// C[M,N] = epilogue( A[M,K] * W[K,N] )  at float32 accuracy on the CDNA4 16-bit matrix cores.
//
// TWO operand formats share the kernels of this file (FMT template parameter, n_planes of regtr_gemm_x3):
//   bf16x3 (n_planes 3, below): x = x0 + x1 + x2 exactly, six MFMA terms, float32's full range -- the original form;
//   f16 pair (n_planes 4, round 3): x = h0 + h1 / 2048, h0 = f16(x), h1 = f16((x - h0) * 2048): 22 mantissa bits in two planes,
//     a w = a0 w0 + (a0 w1 + a1 w0) / 2048 + O(2^-22 |a w|) = THREE v_mfma_f32_32x32x16_f16, the two low terms in a second
//     accumulator set scaled once in the epilogue (the scale keeps the residual plane out of f16's subnormal range).  Measured
//     against float64 it is as accurate as the six-term form (fewer f32 accumulation roundings) at half the matrix-pipe work and
//     two thirds of the weight bytes; operands must stay below 65504 (an overflow yields a non-finite result).  What
//     cfg.compute_dtype 'fp32' uses; see x3_split2_f16, the FMT = 1 paths of both kernels and the f16 hand loop of k_gemm_x3d.
//
// gfx950 runs f32-input MFMA at 1/16 of its bf16 MFMA rate (157 TF vs 2.5 PF).  A float32 value is EXACTLY the sum of
// three bf16 values (8 significant bits each):  x = x0 + x1 + x2,  x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1),
// and every bf16 x bf16 product is exact in f32.  So
//     a * w  =  a0 w0 + (a0 w1 + a1 w0) + (a0 w2 + a1 w1 + a2 w0)  +  O(2^-24 |a w|)
// and the GEMM is six v_mfma_f32_32x32x16_bf16 per (32 x 32 x 16) block with f32 accumulation -- float32-grade results
// (the dropped terms are below one f32 ulp of each product) at up to 16/6 of the f32-MFMA rate.  Terms are issued
// smallest first.  (Finite inputs only: an infinite operand splits into Inf + NaN, so it yields NaN where the exact-f32 kernel
// yields Inf; a NaN stays a NaN.)  Serves the same call sites as gemm.hip (kpconv_blocks.py:401-406,557; regtr.py:145,432-436;
// transformers.py:197-238) whenever N is a multiple of 64; thin / odd shapes stay on the exact-f32 kernel.
//
// Weights are split ONCE (regtr_gemm_split_weights) into three bf16 planes of W^T, Wt[p][n][k] (k contiguous, K padded
// to 32 with zeros): a B fragment is then 8 consecutive k of one column = one 16-byte LDS read.  Activations are
// split on the fly while they are staged into LDS (5.5 VALU ops per element, amortised over the BN columns of the tile).
//
// Workgroup = 4 waves (2 x 2), wave tile (32 WM) x (32 WN), BK = 32.  LDS rows are 64 B (32 bf16) with the four 16-byte
// chunks of row r stored at chunk ^ ((r >> 2) & 3): the 16-byte fragment reads of the 32x32x16 MFMA and the 8-byte
// staging stores are then bank-conflict free without padding, and the image stays lane-linear so the weight planes can
// be streamed by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass; the swizzle is applied to
// the per-lane SOURCE address).  Per k-tile: the A float4 loads and the B DMA of tile t+1 are issued, then the 48 MFMAs
// per wave of tile t run from LDS while they are in flight (B double-buffered, A split + stored after the MFMAs);
// two workgroups per CU interleave.  Split-K (deterministic two-pass) for small-M / deep-K shapes.
// (Measured and dropped: persistent workgroups walking several tiles as one k-tile stream, so that a tile's epilogue overlaps
// the next tile's first loads -- the prefetched operands stay live across the epilogue and every SOUT / STATS variant
// spilled or lost a third of its occupancy under hipcc.)
// (Measured and dropped: an all-DMA variant -- raw float32 A tiles in a 3/4-stage LDS ring issued from inline asm with counted
// vmcnt waits, the split done at fragment time -- ran 5-30 % SLOWER on every RegTR shape: the fragment-time split is
// repeated by every wave sharing the rows, and these few-hundred-tile problems are bound by tile quantisation and L2
// traffic of the 6-byte weight planes, not by prefetch depth.)
#include "common.h"
#include <stdlib.h>

// Development switches of the tile planner (A/B sweeps: tools/x3_bench.py): read from the environment ONLY in variant builds
// (REGTR_VARIANT_FLAGS=-DREGTR_DEV_ENV=1, regtr_amd/build.py); the shipped library compiles them to their defaults, so a stray
// environment variable cannot change production tiling.
#if defined(REGTR_DEV_ENV) && REGTR_DEV_ENV
static inline int x3_dev_env(const char* name, int dflt) { const char* v = getenv(name); return (v && *v) ? atoi(v) : dflt; }
#define X3_DEV_ENV(NAME, DFLT) x3_dev_env(NAME, DFLT)
#else
#define X3_DEV_ENV(NAME, DFLT) (DFLT)
#endif

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int XBK = 32;          // k per tile
constexpr int XROW = 64;         // bytes per LDS row: 32 bf16, chunk-swizzled

typedef __attribute__((address_space(3))) void* x3_lds_ptr;

struct X3Args {
    const float* A; const uint16_t* Wt; float* C;
    const float* bias; const float* row_div; const float* residual;
    const float2* a_stats; const int* a_seg_off;
    float* partial;
    double2* stat_partial;       // optional: per (row-tile, cloud) column sums / sums of squares of C  [slot][N]
    const int* stat_seg_off;     // cloud offsets of the rows of C (n_stat_seg + 1)
    const int4* tile_info;       // optional: per row tile (first cloud, last cloud, first cloud's begin row, its end row)
    size_t plane;                // elements per weight plane = Npad * Kp
    int M, N, K, Kp, lda, ldc, ldr, act, n_seg, k_chunk, n_stat_seg;
    float a_slope;
    int* status;                 // optional: REGTR_STATUS_F16_RANGE is OR-ed in when an f16 pair product came out non-finite
};

// f16 pair: an operand at or beyond f16's range converts to +-Inf and its residual plane to NaN, so every product of that row (or
// column) is non-finite -- the raw accumulators tell.  x * 0 is 0 for a finite x and NaN otherwise: one FMA per accumulator value.
__device__ __forceinline__ void x3_report_range(int* status, float chk)
{
    if (status && chk != chk) atomicOr(status, REGTR_STATUS_F16_RANGE);
}

__device__ __forceinline__ unsigned x3_pack(float a, float b)
{
    bf16x2 v;
    v.x = (__bf16)a; v.y = (__bf16)b;            // v_cvt_pk_bf16_f32, round to nearest even
    return __builtin_bit_cast(unsigned, v);
}

// (a, b) -> three packed bf16 pairs, a = a0 + a1 + a2 exactly (likewise b)
__device__ __forceinline__ void x3_split2(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2)
{
    p0 = x3_pack(a, b);
    const float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);
    p1 = x3_pack(ra, rb);
    p2 = x3_pack(ra - __uint_as_float(p1 << 16), rb - __uint_as_float(p1 & 0xffff0000u));
}

// ---- f16 pair split (operand format 1, "f16x2s"): x = h0 + h1 / 2048 with h0 = f16(x), h1 = f16((x - h0) * 2048) -- 22 mantissa bits in
// TWO planes where the bf16 split needs three for 24.  A product is a0 w0 + (a0 w1 + a1 w0) / 2048 + O(2^-22 |a w|): THREE
// v_mfma_f32_32x32x16_f16 instead of six bf16 ones, the two low terms in a second accumulator that is scaled once in the epilogue
// (the scale keeps the low planes out of f16's subnormal range, where an unscaled residual of anything below 0.12 would sit).
// Range: |x| < 65504 (f16); values below 6.1e-5 have a subnormal (coarse) h0 whose rounding the scaled h1 picks up again.
typedef _Float16 x3_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 x3_f16x2 __attribute__((ext_vector_type(2)));
constexpr float X3_F16_SCALE = 2048.f;
__device__ __forceinline__ unsigned x3_pack_f16(float a, float b)
{
    x3_f16x2 v;
    v.x = (_Float16)a; v.y = (_Float16)b;
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void x3_split2_f16(float a, float b, unsigned& p0, unsigned& p1)
{
    p0 = x3_pack_f16(a, b);
    const x3_f16x2 h = __builtin_bit_cast(x3_f16x2, p0);
    p1 = x3_pack_f16((a - (float)h.x) * X3_F16_SCALE, (b - (float)h.y) * X3_F16_SCALE);
}

// NP = bf16 planes per operand: 3 = float32-grade (six MFMA terms, the default); 2 = the three leading terms a0 w0 + a0 w1 +
// a1 w0 (relative error ~2^-16 per product); 1 = plain bf16 operands (cfg.compute_dtype 'bf16').
// FMT = 1 (NP = 2): the f16 pair format -- two f16 planes per operand, three MFMA terms, the two low ones in a second accumulator set
template <int MW, int NW, int WM, int WN, bool STATS, bool SOUT, int NP = 3, int FMT = 0>      // MW x NW waves, wave tile (32 WM) x (32 WN)
__global__ void __launch_bounds__(64 * MW * NW, (MW * NW >= 8) ? 4 : 2) k_gemm_x3(X3Args g)
{
    static_assert(FMT == 0 || (FMT == 1 && NP == 2), "the f16 pair format has two planes");
    constexpr int NT = 64 * MW * NW, RPP = NT / 8; // threads; A rows staged per pass (8 float4 per row)
    constexpr int BM = 32 * WM * MW, BN = 32 * WN * NW;
    constexpr int AV = BM / RPP;                  // float4 of A per thread per tile
    constexpr int NQ = NP * BN / (16 * MW * NW);  // LDS-DMA instructions (1 KiB each) per wave per tile
    static_assert(BM % RPP == 0 && (NP * BN) % (16 * MW * NW) == 0, "tile must split evenly over the waves");
    constexpr int A_BYTES = NP * BM * XROW, B_BYTES = NP * BN * XROW;
    __shared__ __align__(1024) unsigned char As[A_BYTES];
    __shared__ __align__(1024) unsigned char Bs0[B_BYTES];     // two SEPARATE objects: the compiler can then tell that the
    __shared__ __align__(1024) unsigned char Bs1[B_BYTES];     // DMA into one does not alias fragment reads of the other
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);      // wave-uniform (SGPR): LDS-DMA bases live in M0
    const int wm = wave / NW, wn = wave % NW;
    const int l31 = lane & 31, hi = lane >> 5;
    // Workgroup -> tile map, XCD aware: workgroup b runs on XCD b % 8 (each XCD has its own L2).  The N / BN column tiles
    // that share one A row-tile get the same b % 8 and consecutive b / 8, so the row-tile is fetched from HBM once and
    // hit in that XCD's L2 by the others; the weight planes are small enough to live in every L2.
    int tile_m, tile_n;
    {
        const int nc = g.N / BN, b = blockIdx.x, x = b & 7, q = b >> 3;
        tile_n = q % nc;
        tile_m = (q / nc) * 8 + x;
        if (tile_m * BM >= g.M) return;            // grid is rounded up to 8 row-tiles (before any barrier: whole WG exits)
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int k_begin = blockIdx.z * g.k_chunk;
    const int k_end = min(g.K, k_begin + g.k_chunk);
    const int nk = (k_end - k_begin + XBK - 1) / XBK;

    // ---- A staging: float4 #i of this thread = row (t / 8 + RPP i), k4 = (t % 8) * 4.  Loads are branch free (clamped
    // address + select) so that nothing splits the loop body into blocks the compiler would drain loads at.
    const int a_row = t >> 3, a_k4 = (t & 7) * 4;
    const float* a_ptr[AV];
    const float2* st_ptr[AV];
    bool a_ok[AV];
    // clouds of the tile's first / last row, found by the whole wave in one round trip each; almost every tile lies inside
    // one cloud, and only rows of a straddling tile pay the per-lane binary search (log2(n) dependent loads)
    // The epilogue's cloud range (SOUT) is looked up here too, so that its round trips overlap the first operand loads
    // instead of being a serial tail: on the short-K GEMMs the per-workgroup chain of dependent memory round trips, not
    // bandwidth or MFMA, sets the time.
    int seg_first = 0, seg_last = 0, s_lo = 0, s_hi = 0, s_lo_begin = 0, s_lo_end = 0;
    const int row_last = min(m0 + BM, g.M) - 1;
    if ((SOUT || STATS) && g.tile_info) {
        // ONE 16-byte load instead of a chain of dependent round trips (boundary search, then the found cloud's offsets): on the
        // short-K launches that chain, times the rounds of workgroups per CU, set the kernel's time.  The table is per level
        // (regtr_tile_segments), shared by every launch over the level's rows.
        const int4 ti = g.tile_info[tile_m];
        s_lo = ti.x; s_hi = ti.y; s_lo_begin = ti.z; s_lo_end = ti.w;
        seg_first = s_lo; seg_last = s_hi;
    } else {
    if (SOUT) {
        s_lo = rg_find_segment_wave(g.stat_seg_off, g.n_stat_seg, m0);
        s_hi = rg_find_segment_wave(g.stat_seg_off, g.n_stat_seg, row_last);
        s_lo_begin = g.stat_seg_off[s_lo]; s_lo_end = g.stat_seg_off[s_lo + 1];
    }
    if (STATS) {
        if (SOUT && g.a_seg_off == g.stat_seg_off && g.n_seg == g.n_stat_seg) { seg_first = s_lo; seg_last = s_hi; }
        else {
            seg_first = rg_find_segment_wave(g.a_seg_off, g.n_seg, m0);
            seg_last = rg_find_segment_wave(g.a_seg_off, g.n_seg, row_last);
        }
    }
    }
#pragma unroll
    for (int i = 0; i < AV; i++) {
        const int row = m0 + a_row + RPP * i;
        a_ok[i] = row < g.M;
        const int rc = a_ok[i] ? row : g.M - 1;
        a_ptr[i] = g.A + (size_t)rc * g.lda;
        st_ptr[i] = nullptr;
        if (STATS) {
            const int sg = seg_first == seg_last ? seg_first : rg_find_segment(g.a_seg_off, g.n_seg, rc);
            st_ptr[i] = g.a_stats + (size_t)sg * g.K;
        }
    }
    const unsigned a_st_off = (unsigned)a_row * XROW + ((((unsigned)a_k4 >> 3) ^ (((unsigned)a_row >> 2) & 3u)) * 16u) + ((unsigned)t & 1u) * 8u;
    // load_a only ISSUES loads (raw values, nothing consumed): with loads and DMA in flight hipcc waits vmcnt(0) at the
    // first use of any load result, so every use (guards, InstanceNorm fold, bf16 split) is deferred to store_a, which
    // runs after the MFMAs of the current tile.
    float4 ra[AV], rs01[STATS ? AV : 1], rs23[STATS ? AV : 1];
    bool ra_kin = false;
    auto load_a = [&](int k0) {
        const int k = k0 + a_k4;
        ra_kin = k < k_end;
        const int kc = ra_kin ? k : 0;
#pragma unroll
        for (int i = 0; i < AV; i++) {
            ra[i] = *(const float4*)(a_ptr[i] + kc);
            if (STATS) {
                rs01[i] = *(const float4*)(st_ptr[i] + kc);
                rs23[i] = *(const float4*)(st_ptr[i] + kc + 2);
            }
        }
    };
    auto store_a = [&]() {
#pragma unroll
        for (int i = 0; i < AV; i++) {
            float4 v = ra[i];
            if (STATS) {
                float u;
                u = (v.x - rs01[i].x) * rs01[i].y; v.x = fmaxf(u, u * g.a_slope);
                u = (v.y - rs01[i].z) * rs01[i].w; v.y = fmaxf(u, u * g.a_slope);
                u = (v.z - rs23[i].x) * rs23[i].y; v.z = fmaxf(u, u * g.a_slope);
                u = (v.w - rs23[i].z) * rs23[i].w; v.w = fmaxf(u, u * g.a_slope);
            }
            const bool ok = a_ok[i] && ra_kin;
            v = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
            unsigned p0a, p1a, p2a = 0, p0b, p1b, p2b = 0;
            if (FMT == 1) { x3_split2_f16(v.x, v.y, p0a, p1a); x3_split2_f16(v.z, v.w, p0b, p1b); }
            else { x3_split2(v.x, v.y, p0a, p1a, p2a); x3_split2(v.z, v.w, p0b, p1b, p2b); }
            unsigned char* dst = As + a_st_off + i * RPP * XROW;
            *(uint2*)(dst) = make_uint2(p0a, p0b);
            if (NP > 1) *(uint2*)(dst + BM * XROW) = make_uint2(p1a, p1b);
            if (NP > 2) *(uint2*)(dst + 2 * BM * XROW) = make_uint2(p2a, p2b);
        }
    };
    // ---- B streaming: DMA q of this wave fills LDS bytes [(wave NQ + q) 1024, +1024) of the buffer; lane i is slot
    // s = (wave NQ + q) 64 + i = ((p BN + n) 4 + pc), holding logical chunk pc ^ ((n >> 2) & 3) of column n, plane p
    const uint16_t* b_src[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const int sidx = (wave * NQ + q) * 64 + lane;
        const int p = sidx / (BN * 4), r = sidx % (BN * 4), n = r >> 2, kc = (r & 3) ^ ((n >> 2) & 3);
        b_src[q] = g.Wt + (size_t)p * g.plane + (size_t)(n0 + n) * g.Kp + kc * 8;
    }
    auto dma_b = [&](int k0, unsigned char* Bb) {
#pragma unroll
        for (int q = 0; q < NQ; q++)
            __builtin_amdgcn_global_load_lds((const void*)(b_src[q] + k0), (x3_lds_ptr)(Bb + (wave * NQ + q) * 1024), 16, 0, 0);
    };

    floatx16 acc[WM][WN];
    floatx16 acc_lo[FMT == 1 ? WM : 1][FMT == 1 ? WN : 1];     // f16 pair: the scaled low terms
#pragma unroll
    for (int i = 0; i < WM; i++)
#pragma unroll
        for (int j = 0; j < WN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) { acc[i][j][r] = 0.f; if (FMT == 1) acc_lo[FMT == 1 ? i : 0][FMT == 1 ? j : 0][r] = 0.f; }

    // fragment byte offset of this lane within a 32-row block: row l31, logical chunk (2 ks + hi)
    unsigned f_off[XBK / 16];
#pragma unroll
    for (int ks = 0; ks < XBK / 16; ks++) f_off[ks] = (unsigned)l31 * XROW + ((((unsigned)(2 * ks + hi)) ^ (((unsigned)l31 >> 2) & 3u)) * 16u);

    auto compute = [&](const unsigned char* Bb) {
#pragma unroll
        for (int ks = 0; ks < XBK / 16; ks++) {
            bf16x8 fa[WM][NP], fb[WN][NP];
#pragma unroll
            for (int i = 0; i < WM; i++)
#pragma unroll
                for (int p = 0; p < NP; p++)
                    fa[i][p] = __builtin_bit_cast(bf16x8, *(const uint4*)(As + (p * BM + (wm * WM + i) * 32) * XROW + f_off[ks]));
#pragma unroll
            for (int j = 0; j < WN; j++)
#pragma unroll
                for (int p = 0; p < NP; p++)
                    fb[j][p] = __builtin_bit_cast(bf16x8, *(const uint4*)(Bb + (p * BN + (wn * WN + j) * 32) * XROW + f_off[ks]));
            // six terms, smallest first; the (i, j) loops sit inside so consecutive MFMAs hit different accumulators
#define X3_TERM(PA, PB)                                                                                             \
            _Pragma("unroll") for (int i = 0; i < WM; i++)                                                           \
                _Pragma("unroll") for (int j = 0; j < WN; j++)                                                       \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][PA], fb[j][PB], acc[i][j], 0, 0, 0);
#define X3_TERM_F16(ACC, PA, PB)                                                                                    \
            _Pragma("unroll") for (int i = 0; i < WM; i++)                                                           \
                _Pragma("unroll") for (int j = 0; j < WN; j++)                                                       \
                    ACC[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(x3_f16x8, fa[i][PA]),      \
                                                                       __builtin_bit_cast(x3_f16x8, fb[j][PB]), ACC[i][j], 0, 0, 0);
            if constexpr (FMT == 1) { X3_TERM_F16(acc_lo, 1, 0) X3_TERM_F16(acc, 0, 0) X3_TERM_F16(acc_lo, 0, 1) }
            else {
            if (NP == 3) { X3_TERM(2, 0) X3_TERM(1, 1) X3_TERM(0, 2) }
            if (NP >= 2) { X3_TERM((NP >= 2 ? 1 : 0), 0) X3_TERM(0, (NP >= 2 ? 1 : 0)) }
            X3_TERM(0, 0)
            }
#undef X3_TERM
#undef X3_TERM_F16
        }
    };

    load_a(k_begin);
    dma_b(k_begin, Bs0);
    store_a();
    __syncthreads();                               // (carries the vmcnt(0) that lands the DMA)
    for (int kt = 0; kt < nk; kt += 2) {
        // even tile: MFMAs from Bs0 while tile kt+1 streams into Bs1 / registers
        bool more = kt + 1 < nk;                   // wave-uniform
        if (more) {
            load_a(k_begin + (kt + 1) * XBK);
            dma_b(k_begin + (kt + 1) * XBK, Bs1);
        }
        compute(Bs0);
        __syncthreads();                           // every wave is done reading As
        if (!more) break;
        store_a();
        __syncthreads();                           // As (and the DMA'd B buffer) visible
        // odd tile: MFMAs from Bs1 while tile kt+2 streams into Bs0 / registers
        more = kt + 2 < nk;
        if (more) {
            load_a(k_begin + (kt + 2) * XBK);
            dma_b(k_begin + (kt + 2) * XBK, Bs0);
        }
        compute(Bs1);
        __syncthreads();
        if (!more) break;
        store_a();
        __syncthreads();
    }

    if constexpr (FMT == 1) {                              // f16 pair: fold the scaled low terms in
        float chk = 0.f;
#pragma unroll
        for (int i = 0; i < WM; i++)
#pragma unroll
            for (int j = 0; j < WN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) { acc[i][j][r] += acc_lo[i][j][r] * (1.0f / X3_F16_SCALE); chk = fmaf(acc[i][j][r], 0.f, chk); }
        x3_report_range(g.status, chk);
    }
    // ---- epilogue (C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5))
#pragma unroll
    for (int i = 0; i < WM; i++)
#pragma unroll
        for (int j = 0; j < WN; j++) {
            const int col = n0 + (wn * WN + j) * 32 + l31;
            const int rbase = m0 + (wm * WM + i) * 32 + 4 * hi;
            if (g.partial) {   // split-K: raw accumulators, epilogue happens in k_x3_splitk_reduce
                float* P = g.partial + (size_t)blockIdx.z * g.M * g.N;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2);
                    if (row < g.M) P[(size_t)row * g.N + col] = acc[i][j][r];
                }
                continue;
            }
            const float bv = g.bias ? g.bias[col] : 0.f;
            // row_div / residual from clamped rows, eight rows' loads in flight at once (see k_gemm_x3d's epilogue; eight, not
            // sixteen: the <2,4,..> variants sit at their 128-register launch bound)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                float rd[8], rs[8];
                if (g.row_div) {
#pragma unroll
                    for (int q = 0; q < 8; q++) { const int r = 8 * h + q; rd[q] = g.row_div[min(rbase + (r & 3) + 8 * (r >> 2), g.M - 1)]; }
                }
                if (g.residual) {
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const int r = 8 * h + q;
                        rs[q] = g.residual[(size_t)min(rbase + (r & 3) + 8 * (r >> 2), g.M - 1) * g.ldr + col];
                    }
                }
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int r = 8 * h + q, row = rbase + (r & 3) + 8 * (r >> 2);
                    float v = acc[i][j][r];
                    if (g.row_div) v = v / rd[q];
                    v += bv;
                    if (g.act == 1) v = fmaxf(v, 0.f);
                    if (g.residual) v += rs[q];
                    if (row >= g.M) continue;
                    g.C[(size_t)row * g.ldc + col] = v;
                    if (SOUT) acc[i][j][r] = v;
                }
            }
        }

    // ---- optional: InstanceNorm statistics of the rows just produced (kpconv_blocks.py:510-519), so that no separate
    // pass re-reads C.  For every cloud s that owns rows of this tile: per-column (sum, sum of squares) in float64 over
    // the tile's rows of s -> stat_partial[(tile_m + s) * N + col]   (slot tile_m + s is unique: both only grow along M).
    // regtr_instnorm_finalize_tiles adds the slots of a cloud in fixed order: deterministic, float64 like the stand-alone
    // statistics kernel.
    if (SOUT) {
        double2* red = (double2*)As;                       // [MW][BN], As is free after the last barrier of the k loop
        for (int sg = s_lo; sg <= s_hi; sg++) {            // workgroup-uniform; one cloud per tile almost always
            const int r_lo = sg == s_lo ? s_lo_begin : g.stat_seg_off[sg];
            const int r_hi = min(sg == s_lo ? s_lo_end : g.stat_seg_off[sg + 1], g.M);
#pragma unroll
            for (int j = 0; j < WN; j++) {
                double sm = 0.0, sq = 0.0;
#pragma unroll
                for (int i = 0; i < WM; i++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int row = m0 + (wm * WM + i) * 32 + 4 * hi + (r & 3) + 8 * (r >> 2);
                        if (row >= r_lo && row < r_hi) { const double v = (double)acc[i][j][r]; sm += v; sq += v * v; }
                    }
                sm += __shfl_xor(sm, 32, RG_WAVE); sq += __shfl_xor(sq, 32, RG_WAVE);
                if (hi == 0) red[wm * BN + (wn * WN + j) * 32 + l31] = make_double2(sm, sq);
            }
            __syncthreads();
            if (t < BN) {
                double2 a = red[t];
#pragma unroll
                for (int w = 1; w < MW; w++) { const double2 b = red[w * BN + t]; a.x += b.x; a.y += b.y; }
                g.stat_partial[(size_t)(tile_m + sg) * g.N + n0 + t] = a;
            }
            __syncthreads();
        }
    }
}

typedef float x3_f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void x3_asm_dma16(const void* gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// ------------------------------------------------------------------------------------------------------------------
// ROW-STRIP, ALL-DMA form of the same product for the tall problems (the launcher's choice from a few hundred 128-row tiles on,
// K a multiple of 32, no InstanceNorm folded into A).  A workgroup is four waves; wave w owns rows [32 w, 32 w + 32) of the 128-row
// tile and ALL 32 CW columns, so no other wave ever needs its A rows:
//   * the raw float32 A rows arrive by LDS-DMA in whole 128-byte lines (each wave fetches exactly its own 32 rows: the A stream needs
//     no barrier, only the wave's own counted wait) into an XOR-swizzled image (16-byte chunk c of row r at c ^ ((r >> 1) & 7): the
//     two ds_read_b128 of a fragment are conflict free), and are split into the three bf16 planes AT FRAGMENT TIME, exactly once
//     per element (the all-DMA form of the tiled kernel lost there: every column wave repeated the split);
//   * the weight planes -- shared by the four waves -- are staged by LDS-DMA into the other half of the two-slot ring; one barrier per
//     k-tile (the tiled kernel: two, plus an A store pass of 5.5 VALU + one 8-byte LDS store per element per COLUMN tile);
//   * 0.5 16-byte LDS reads per MFMA (tiled: 0.75).
// What was measured on the way (MI355X, RegTR's shapes at 64 pairs per forward, tools/x3_bench.py; tables in profiles/r03_x3_*):
//   * the same strip layout with A loaded by each lane in fragment shape straight into registers (no LDS for A), compiler-scheduled:
//     equal to the tiled kernel (sum over 18 shapes 5.34 vs 5.05 ms).  Ablations: no A loads -30 %, no barrier / weight DMA -17 %, no
//     split -5 %, no fragment reads -5 %, MFMAs alone 2.3x faster -- every cost ADDS, nothing overlaps;
//   * that form with both operand streams issued from inline asm and a two-tile-deep register prefetch of A (counted vmcnt across
//     the barrier): -3 %.  Phase clocks (s_memtime) per k-tile and wave: issuing 6 weight DMAs + 4 fragment-shaped A loads 1100-1650
//     cycles (each A load gathers 32 different 128-byte lines; the wave issues in order, so no MFMA goes out meanwhile), 48 MFMAs 750,
//     wait + barrier 1000, split 150;
//   * A by DMA as well (this kernel's data path), compiler-scheduled k loop: 505 vs 545 us on the level-3 KPConv contraction; PMC:
//     matrix pipe 44 % busy, waves 48 % of their cycles in issue stalls, 26 % in waits, 5 VALU per MFMA at 128 x 64 tiles;
//     tile / split-K sweeps (3 tiles x 5 splits) move nothing by more than 5 %: the bound is per-CU, not occupancy or quantisation;
//   * the .s showed why: hipcc hoists each step's split (44 dependent VALU) in front of that step's MFMAs and issues every fragment
//     read right before its consumer (`ds_read ; s_waitcnt lgkmcnt(0) ; v_mfma x5`); __builtin_amdgcn_sched_group_barrier did not
//     change the order.  Hence the hand-scheduled loop below (NP = 3): -11 ... -14 % on the long-K shapes.
// ---- single-instruction asm wrappers of the hand-scheduled k loop (k_gemm_x3d, NP = 3): volatile asm statements keep their
// program order, which is the point -- hipcc's list scheduler hoists the split ahead of the MFMAs and issues each fragment read
// right in front of its consumer (the .s of the plain-C++ loop: `ds_read ; s_waitcnt lgkmcnt(0) ; v_mfma x5`), and
// sched_group_barrier did not change that.  Waits are counted by hand: LDS operations return in order.
__device__ __forceinline__ void x3h_mfma(floatx16& c, const bf16x8& a, const bf16x8& b)
{
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <int OFF>
__device__ __forceinline__ void x3h_lds128(uint4& d, unsigned addr)
{
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
// first half of the exact three-way split of a float pair: p0 = bf16 pair, (ra, rb) = residuals
__device__ __forceinline__ void x3h_split_a(float a, float b, unsigned& p0, float& ra, float& rb)
{
    unsigned t0, t1;
    asm volatile("v_cvt_pk_bf16_f32 %0, %5, %6\n\t"
                 "v_lshlrev_b32 %3, 16, %0\n\t"
                 "v_and_b32 %4, 0xffff0000, %0\n\t"
                 "v_sub_f32 %1, %5, %3\n\t"
                 "v_sub_f32 %2, %6, %4"
                 : "=&v"(p0), "=&v"(ra), "=&v"(rb), "=&v"(t0), "=&v"(t1) : "v"(a), "v"(b));
}
// second half: p1 = bf16 pair of the residuals, p2 = bf16 pair of what is left
__device__ __forceinline__ void x3h_split_b(float ra, float rb, unsigned& p1, unsigned& p2)
{
    unsigned t0, t1;
    float sa, sb;
    asm volatile("v_cvt_pk_bf16_f32 %0, %6, %7\n\t"
                 "v_lshlrev_b32 %2, 16, %0\n\t"
                 "v_and_b32 %3, 0xffff0000, %0\n\t"
                 "v_sub_f32 %4, %6, %2\n\t"
                 "v_sub_f32 %5, %7, %3\n\t"
                 "v_cvt_pk_bf16_f32 %1, %4, %5"
                 : "=&v"(p1), "=&v"(p2), "=&v"(t0), "=&v"(t1), "=&v"(sa), "=&v"(sb) : "v"(ra), "v"(rb));
}
#define X3H_LGKM(N) asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory")
// ---- the same for the f16 pair format (three MFMA terms; x = h0 + h1 / 2048)
__device__ __forceinline__ void x3h_mfma_f16(floatx16& c, const bf16x8& a, const bf16x8& b)      // (operands are eight f16; the type only carries the bits)
{
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// first half of the split of a float pair: p0 = f16 pair, (ha, hb) = its two values back in float32
__device__ __forceinline__ void x3h_splitf_a(float a, float b, unsigned& p0, float& ha, float& hb)
{
    asm volatile("v_cvt_pk_f16_f32 %0, %3, %4\n\t"
                 "v_cvt_f32_f16_e32 %1, %0\n\t"
                 "v_cvt_f32_f16_sdwa %2, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1"
                 : "=&v"(p0), "=&v"(ha), "=&v"(hb) : "v"(a), "v"(b));
}
// second half: p1 = f16 pair of the residuals scaled by 2048 (0x45000000)
__device__ __forceinline__ void x3h_splitf_b(float a, float b, float ha, float hb, unsigned& p1)
{
    float t0, t1;
    asm volatile("v_sub_f32 %1, %3, %5\n\t"
                 "v_sub_f32 %2, %4, %6\n\t"
                 "v_mul_f32 %1, 0x45000000, %1\n\t"
                 "v_mul_f32 %2, 0x45000000, %2\n\t"
                 "v_cvt_pk_f16_f32 %0, %1, %2"
                 : "=&v"(p1), "=&v"(t0), "=&v"(t1) : "v"(a), "v"(b), "v"(ha), "v"(hb));
}
#ifndef X3D_HAND
#define X3D_HAND 1       // development (REGTR_VARIANT_FLAGS=-DX3D_HAND=0): the compiler-scheduled k loop for A/B runs
#endif

// MW waves (4 or 8: 128- or 256-row tiles; the weight tile is shared by all of them, so twice the rows halve the weight traffic per
// MFMA), CW 32-column blocks per wave, AR slots of the wave-private A ring (3: the A rows of tile t + 2 are in flight while tile t is
// multiplied -- a full tile more than the weights get, which hit L2 while A comes from HBM), NP planes.
// FMT: operand format -- 0 = NP bf16 planes, 1 = the f16 pair (NP = 2 planes, three MFMA terms, second accumulator for the low terms)
template <int MW, int CW, int AR, bool SOUT, int NP = 3, int FMT = 0>
__global__ void __launch_bounds__(64 * MW, 2) k_gemm_x3d(X3Args g)
{
    static_assert(FMT == 0 || (FMT == 1 && NP == 2 && AR == 2), "the f16 pair format has two planes and runs the compiler-scheduled loop");
    constexpr int NT = 64 * MW, BM = 32 * MW, BN = 32 * CW;
    constexpr int A_BYTES = BM * XBK * 4;                      // raw float32 rows of one k-tile: 128 bytes per row
    constexpr int B_BYTES = NP * BN * XROW;
    constexpr int NQ = B_BYTES / 1024 / MW, NA = 4;            // LDS-DMA instructions (1 KiB) per wave and k-tile: weights / own A rows
    // AR = 4: the INTERLEAVED schedule -- two A slots, A two tiles ahead, every LDS-DMA instruction issued between MFMAs (see the hand loop)
    constexpr bool IL = AR == 4;
    constexpr int A_SLOTS = AR == 3 ? 3 : 2;
    constexpr int A_RING = A_SLOTS * A_BYTES, RING_BYTES = A_RING + 2 * B_BYTES;
    constexpr int STAT_BYTES = SOUT ? BM * BN * 4 + (NT / BN) * BN * 16 : 0;      // the statistics epilogue's tile image + partial sums
    constexpr int LDS_BYTES = RING_BYTES > STAT_BYTES ? RING_BYTES : STAT_BYTES;
    static_assert(NQ * 1024 * MW == B_BYTES && NQ >= 1, "the weight tile must split evenly over the waves");
    static_assert(AR == 2 || AR == 3 || AR == 4, "A ring mode");
    static_assert(!IL || (NP == 3 && X3D_HAND), "the interleaved schedule exists in the hand-scheduled loop only");
    __shared__ __align__(1024) unsigned char sm[LDS_BYTES];   // [A ring: AR x BM rows x 128 B][weight ring: 2 x NP planes x BN x 64 B]
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    int tile_m, tile_n;
    {
        const int nc = (g.N + BN - 1) / BN, b = blockIdx.x, x = b & 7, q = b >> 3;       // XCD-aware map, as in k_gemm_x3
        tile_n = q % nc;                                       // (N = 32: one 64-column tile whose upper half multiplies the planes' zero padding)
        tile_m = (q / nc) * 8 + x;
        if (tile_m * BM >= g.M) return;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int k_begin = blockIdx.z * g.k_chunk;
    const int k_end = min(g.K, k_begin + g.k_chunk);
    const int nk = (k_end - k_begin) / XBK;                    // whole k-tiles only (the launcher checks)

    int s_lo = 0, s_hi = 0, s_lo_begin = 0, s_lo_end = 0;
    if (SOUT) {
        if (g.tile_info) {
            const int4 ti = g.tile_info[tile_m];
            s_lo = ti.x; s_hi = ti.y; s_lo_begin = ti.z; s_lo_end = ti.w;
        } else {
            const int row_last = min(m0 + BM, g.M) - 1;
            s_lo = rg_find_segment_wave(g.stat_seg_off, g.n_stat_seg, m0);
            s_hi = rg_find_segment_wave(g.stat_seg_off, g.n_stat_seg, row_last);
            s_lo_begin = g.stat_seg_off[s_lo]; s_lo_end = g.stat_seg_off[s_lo + 1];
        }
    }
    // ---- A DMA: instruction q of this wave fills rows 32 wave + 8 q + (lane >> 3), LDS slot lane & 7 <- source chunk slot ^ ((row >> 1) & 7)
    const float* a_src[NA];
#pragma unroll
    for (int q = 0; q < NA; q++) {
        const int r = wave * 32 + q * 8 + (lane >> 3);
        const int row = m0 + r, rc = row < g.M ? row : g.M - 1;    // rows past M compute garbage that the epilogue never stores
        a_src[q] = g.A + (size_t)rc * g.lda + k_begin + (((lane & 7) ^ ((r >> 1) & 7)) * 4);
    }
    const uint16_t* b_src[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const int sidx = (wave * NQ + q) * 64 + lane;
        const int p = sidx / (BN * 4), r = sidx % (BN * 4), n = r >> 2, kc = (r & 3) ^ ((n >> 2) & 3);
        b_src[q] = g.Wt + (size_t)p * g.plane + (size_t)(n0 + n) * g.Kp + kc * 8 + k_begin;
    }
    const unsigned lds_base = (unsigned)(unsigned long long)(x3_lds_ptr)(&sm[0]);
    const unsigned a_wave = lds_base + (unsigned)wave * (NA * 1024u);             // this wave's 4 KiB of an A slot
    const unsigned b_wave = lds_base + A_RING + (unsigned)wave * (NQ * 1024u);    // this wave's share of a weight slot
    auto dma_a = [&](int kt, unsigned slot) {                  // own A rows of k-tile kt -> A slot
#pragma unroll
        for (int q = 0; q < NA; q++) x3_asm_dma16((const void*)(a_src[q] + kt * XBK), a_wave + slot * A_BYTES + q * 1024u);
    };
    auto dma_b = [&](int kt, unsigned slot) {                  // this wave's share of the weight planes of k-tile kt -> weight slot
#pragma unroll
        for (int q = 0; q < NQ; q++) x3_asm_dma16((const void*)(b_src[q] + kt * XBK), b_wave + slot * B_BYTES + q * 1024u);
    };
    // fragment addresses: A row 32 wave + l31, 16-byte chunks 4 ks + 2 hi (+ 1), swizzled; B as in k_gemm_x3
    unsigned fa_off[2][2], f_off[2];
    {
        const unsigned r = (unsigned)wave * 32u + (unsigned)l31, sw = (r >> 1) & 7u;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
#pragma unroll
            for (int h = 0; h < 2; h++) fa_off[ks][h] = r * 128u + ((((unsigned)(4 * ks + 2 * hi + h)) ^ sw) * 16u);
            f_off[ks] = (unsigned)l31 * XROW + ((((unsigned)(2 * ks + hi)) ^ (((unsigned)l31 >> 2) & 3u)) * 16u);
        }
    }
    floatx16 acc[CW];
    floatx16 acc_lo[FMT == 1 ? CW : 1];                         // f16 pair: the two scaled low terms
#pragma unroll
    for (int j = 0; j < CW; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) { acc[j][r] = 0.f; if (FMT == 1) acc_lo[FMT == 1 ? j : 0][r] = 0.f; }

    if constexpr (NP == 3 && X3D_HAND) {
    // ---- hand-scheduled k loop.  Per 16-k step and column block: three fragment reads of the NEXT block are issued, the wait
    // leaves exactly those in flight, then the block's six MFMAs go out with the two halves of one float pair's split (of the next
    // step's A piece) behind the first two.  Register sets are static -- fa[2] (step parity), fb[2] (block parity), raw[2] -- and
    // the ring slots are carried in the ADDRESS registers, so the loop body is one tile and no accumulator is ever copied between
    // differently-allocated halves of an unrolled loop.
    // Operand streams per tile t (all LDS-DMA, issued from asm, counted by hand; one in-order counter):
    //   AR = 2:  top: A(t+1) then W(t+1)            mid-tile: vmcnt(NQ) = A(t+1) landed        end: vmcnt(0) + barrier
    //   AR = 3:  top: W(t+1) then A(t+2)            mid-tile: vmcnt(NQ + 4) = A(t+1) landed    end: vmcnt(4) + barrier (A(t+2) stays in flight)
    constexpr int PJ = 4 / CW;
    static_assert(CW == 2 || CW == 4, "the split is spread over 2 or 4 column blocks");
    unsigned vb[2], va[2][2];              // per-lane LDS addresses: weights of the CURRENT slot at step ks; A piece halves of the NEXT tile's slot
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
        vb[ks] = lds_base + A_RING + f_off[ks];
        va[ks][0] = lds_base + A_BYTES + fa_off[ks][0];
        va[ks][1] = lds_base + A_BYTES + fa_off[ks][1];
    }
    unsigned b_slot = 1, a_slot = A_SLOTS - 1, a_read = 1;      // (wave-uniform) weight slot / A slot the next DMAs go to; A slot read next
    uint4 fbq[2][3], rawq[2][2];          // weight fragments (block parity, plane); A piece halves (step parity, half)
    unsigned pl[2][3][4];                  // split planes being built (step parity, plane, float pair)
    bf16x8 fa[2][3];
    float ra = 0.f, rb = 0.f;
#define X3H_FB(KS, J, SET) do { x3h_lds128<(0 * BN + (J) * 32) * XROW>(fbq[SET][0], vb[KS]); \
                                x3h_lds128<(1 * BN + (J) * 32) * XROW>(fbq[SET][1], vb[KS]); \
                                x3h_lds128<(2 * BN + (J) * 32) * XROW>(fbq[SET][2], vb[KS]); } while (0)
#define X3H_RAW(KS, SET) do { x3h_lds128<0>(rawq[SET][0], va[KS][0]); x3h_lds128<0>(rawq[SET][1], va[KS][1]); } while (0)
#define X3H_PAIR_A(SET, Q) do { const uint4 v_ = rawq[SET][(Q) >> 1]; \
        x3h_split_a(__uint_as_float(((Q) & 1) ? v_.z : v_.x), __uint_as_float(((Q) & 1) ? v_.w : v_.y), pl[SET][0][Q], ra, rb); } while (0)
#define X3H_PAIR_B(SET, Q) x3h_split_b(ra, rb, pl[SET][1][Q], pl[SET][2][Q])
#define X3H_PACK(SET) do { _Pragma("unroll") for (int p_ = 0; p_ < 3; p_++) \
        fa[SET][p_] = __builtin_bit_cast(bf16x8, make_uint4(pl[SET][p_][0], pl[SET][p_][1], pl[SET][p_][2], pl[SET][p_][3])); } while (0)
    // the six MFMAs of column block J from fa[FS] x fb[CS] (smallest terms first) with the split of pairs PJ J .. of raw[FS ^ 1] behind them
    // H1 / H2: statements issued behind the fifth / sixth MFMA (the interleaved schedule's LDS-DMA instructions: their issue -- 16 cycles
    // of the CU's texture-address path each, more when eight waves queue up -- overlaps the 32 matrix-pipe cycles of the MFMA in front)
#ifndef X3H_SPLIT_AFTER
#define X3H_SPLIT_AFTER 0          // development A-B: 1 = the six MFMAs of a block back to back, the split and DMA instructions after them.
                                   // Measured (gpurun_out/r03_z7, 18 shapes): 4953 vs 4788 us -- the VALU between same-accumulator MFMAs is
                                   // not what the matrix pipe waits for here (two waves per SIMD fill each other's gaps); off
#endif
#define X3H_BLOCKH(J, FS, CS, H1, H2) do { \
        const bf16x8 b0_ = __builtin_bit_cast(bf16x8, fbq[CS][0]), b1_ = __builtin_bit_cast(bf16x8, fbq[CS][1]), b2_ = __builtin_bit_cast(bf16x8, fbq[CS][2]); \
        if (X3H_SPLIT_AFTER) { \
            x3h_mfma(acc[J], fa[FS][2], b0_); x3h_mfma(acc[J], fa[FS][1], b1_); x3h_mfma(acc[J], fa[FS][0], b2_); \
            x3h_mfma(acc[J], fa[FS][1], b0_); x3h_mfma(acc[J], fa[FS][0], b1_); x3h_mfma(acc[J], fa[FS][0], b0_); \
            X3H_PAIR_A((FS) ^ 1, PJ * (J)); X3H_PAIR_B((FS) ^ 1, PJ * (J)); \
            if (PJ == 2) { X3H_PAIR_A((FS) ^ 1, PJ * (J) + 1); X3H_PAIR_B((FS) ^ 1, PJ * (J) + 1); } \
            H1; H2; \
        } else { \
        x3h_mfma(acc[J], fa[FS][2], b0_); X3H_PAIR_A((FS) ^ 1, PJ * (J)); \
        x3h_mfma(acc[J], fa[FS][1], b1_); X3H_PAIR_B((FS) ^ 1, PJ * (J)); \
        x3h_mfma(acc[J], fa[FS][0], b2_); if (PJ == 2) X3H_PAIR_A((FS) ^ 1, PJ * (J) + 1); \
        x3h_mfma(acc[J], fa[FS][1], b0_); if (PJ == 2) X3H_PAIR_B((FS) ^ 1, PJ * (J) + 1); \
        x3h_mfma(acc[J], fa[FS][0], b1_); H1; \
        x3h_mfma(acc[J], fa[FS][0], b0_); H2; } } while (0)
#define X3H_BLOCK(J, FS, CS) X3H_BLOCKH(J, FS, CS, (void)0, (void)0)
    // one 16-k step KS of the current slot using fa[FS]; at its end fa[FS ^ 1] is complete.  _MID: the first block of the slot's next
    // step is prefetched behind the last block; _END: last step of a tile (the next tile's fragments need the barrier first).  LDS
    // operations return in order: with the next block's three reads just issued, lgkmcnt(3) = everything older has landed (the current
    // block's fragments and, in the tile's second step, the four A-piece reads issued in front of it).
#define X3H_STEP_MID(KS, FS, NKS) do { \
        if constexpr (CW == 4) { \
            X3H_FB(KS, 1, 1); X3H_LGKM(3); X3H_BLOCK(0, FS, 0); \
            X3H_FB(KS, 2, 0); X3H_LGKM(3); X3H_BLOCK(1, FS, 1); \
            X3H_FB(KS, 3, 1); X3H_LGKM(3); X3H_BLOCK(2, FS, 0); \
            X3H_FB(NKS, 0, 0); X3H_LGKM(3); X3H_BLOCK(3 % CW, FS, 1); \
        } else { \
            X3H_FB(KS, 1, 1); X3H_LGKM(3); X3H_BLOCK(0, FS, 0); \
            X3H_FB(NKS, 0, 0); X3H_LGKM(3); X3H_BLOCK(1, FS, 1); \
        } \
        X3H_PACK((FS) ^ 1); } while (0)
#define X3H_STEP_END(KS, FS) do { \
        if constexpr (CW == 4) { \
            X3H_FB(KS, 1, 1); X3H_LGKM(3); X3H_BLOCK(0, FS, 0); \
            X3H_FB(KS, 2, 0); X3H_LGKM(3); X3H_BLOCK(1, FS, 1); \
            X3H_FB(KS, 3, 1); X3H_LGKM(3); X3H_BLOCK(2, FS, 0); \
            X3H_LGKM(0); X3H_BLOCK(3 % CW, FS, 1); \
        } else { \
            X3H_FB(KS, 1, 1); X3H_LGKM(3); X3H_BLOCK(0, FS, 0); \
            X3H_LGKM(0); X3H_BLOCK(1, FS, 1); \
        } \
        X3H_PACK((FS) ^ 1); } while (0)
    // the interleaved schedule's steps: step 0 of a tile carries the NQ weight DMAs of tile t + 1 (X3H_DB), step 1 the NA own-row DMAs of
    // tile t + 2 (X3H_DA), two per column block at most
#define X3H_DB(I) do { if ((I) < NQ && more_b) x3_asm_dma16((const void*)(b_src[(I) < NQ ? (I) : 0] + (kt + 1) * XBK), b_wave + b_slot * B_BYTES + (I) * 1024u); } while (0)
#define X3H_DA(I) do { if ((I) < NA && more_a) x3_asm_dma16((const void*)(a_src[(I) < NA ? (I) : 0] + (kt + 2) * XBK), a_wave + a_tgt * A_BYTES + (I) * 1024u); } while (0)
#define X3H_STEP_MID_IL(KS, FS, NKS) do { \
        if constexpr (CW == 4) { \
            X3H_FB(KS, 1, 1); X3H_LGKM(3); X3H_BLOCKH(0, FS, 0, X3H_DB(0), X3H_DB(1)); \
            X3H_FB(KS, 2, 0); X3H_LGKM(3); X3H_BLOCKH(1, FS, 1, X3H_DB(2), X3H_DB(3)); \
            X3H_FB(KS, 3, 1); X3H_LGKM(3); X3H_BLOCKH(2, FS, 0, X3H_DB(4), X3H_DB(5)); \
            X3H_FB(NKS, 0, 0); X3H_LGKM(3); X3H_BLOCKH(3 % CW, FS, 1, X3H_DB(6), X3H_DB(7)); \
        } else { \
            X3H_FB(KS, 1, 1); X3H_LGKM(3); X3H_BLOCKH(0, FS, 0, X3H_DB(0), X3H_DB(1)); \
            X3H_FB(NKS, 0, 0); X3H_LGKM(3); X3H_BLOCKH(1, FS, 1, X3H_DB(2), X3H_DB(3)); \
        } \
        X3H_PACK((FS) ^ 1); } while (0)
#define X3H_STEP_END_IL(KS, FS) do { \
        if constexpr (CW == 4) { \
            X3H_FB(KS, 1, 1); X3H_LGKM(3); X3H_BLOCKH(0, FS, 0, X3H_DA(0), (void)0); \
            X3H_FB(KS, 2, 0); X3H_LGKM(3); X3H_BLOCKH(1, FS, 1, X3H_DA(1), (void)0); \
            X3H_FB(KS, 3, 1); X3H_LGKM(3); X3H_BLOCKH(2, FS, 0, X3H_DA(2), (void)0); \
            X3H_LGKM(0); X3H_BLOCKH(3 % CW, FS, 1, X3H_DA(3), (void)0); \
        } else { \
            X3H_FB(KS, 1, 1); X3H_LGKM(3); X3H_BLOCKH(0, FS, 0, X3H_DA(0), X3H_DA(1)); \
            X3H_LGKM(0); X3H_BLOCKH(1, FS, 1, X3H_DA(2), X3H_DA(3)); \
        } \
        X3H_PACK((FS) ^ 1); } while (0)
#define X3H_VM(N) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory")
#define X3H_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    // prologue: weights and A of tile 0 (and, with the deeper ring, A of tile 1)
    dma_b(0, 0);
    dma_a(0, 0);
    if ((AR == 3 || IL) && nk > 1) { dma_a(1, 1); X3H_VM(4); } else X3H_VM(0);
    X3H_BARRIER();
    // fa[0] = step 0 of tile 0 (split here, unhidden, once), raw[1] = step 1's piece -- both from A slot 0
    {
        unsigned a00 = lds_base + fa_off[0][0], a01 = lds_base + fa_off[0][1], a10 = lds_base + fa_off[1][0], a11 = lds_base + fa_off[1][1];
        x3h_lds128<0>(rawq[0][0], a00); x3h_lds128<0>(rawq[0][1], a01);
        x3h_lds128<0>(rawq[1][0], a10); x3h_lds128<0>(rawq[1][1], a11);
    }
    X3H_LGKM(0);
#pragma unroll
    for (int q = 0; q < 4; q++) { X3H_PAIR_A(0, q); X3H_PAIR_B(0, q); }
    X3H_PACK(0);
#ifdef X3D_PROF          // development: per-wave phase clocks of three workgroups (REGTR_VARIANT_FLAGS=-DX3D_PROF)
    long long pt[5] = {0, 0, 0, 0, 0}, pc = clock64();
#define X3D_STAMP(I) do { const long long n_ = clock64(); pt[I] += n_ - pc; pc = n_; } while (0)
#else
#define X3D_STAMP(I) do {} while (0)
#endif
    for (int kt = 0; kt < nk; kt++) {
        const bool more_b = kt + 1 < nk, more_a = kt + (IL ? 2 : AR - 1) < nk;  // wave-uniform
        const unsigned a_tgt = a_read ^ 1u;                                      // IL: the slot tile kt's rows came from is free since mid tile kt - 1
        if (IL) {}
        else if (AR == 2) { if (more_a) dma_a(kt + 1, a_slot); if (more_b) dma_b(kt + 1, b_slot); }
        else              { if (more_b) dma_b(kt + 1, b_slot); if (more_a) dma_a(kt + 2, a_slot); }
        X3D_STAMP(0);
        X3H_FB(0, 0, 0);
        if constexpr (IL) X3H_STEP_MID_IL(0, 0, 1); else X3H_STEP_MID(0, 0, 1);
        X3D_STAMP(1);
        // this wave's A rows of tile kt + 1 have landed (after the last tile: stale data, split and never used)
        if (AR == 2 || IL) { if (more_b) X3H_VM(NQ); else X3H_VM(0); }           // (IL: A(kt + 1) went out a tile ago, only W(kt + 1) is younger)
        else { if (more_b && more_a) X3H_VM(NQ + 4); else if (more_b) X3H_VM(NQ); else X3H_VM(0); }
        X3D_STAMP(2);
        X3H_RAW(0, 0);
        X3H_RAW(1, 1);
        if constexpr (IL) X3H_STEP_END_IL(1, 1); else X3H_STEP_END(1, 1);
        X3D_STAMP(3);
        if ((AR == 3 || IL) && more_a) X3H_VM(4); else X3H_VM(0);   // the weights of tile kt + 1 have landed (A of tile kt + 2 stays in flight)
        X3H_BARRIER();                                          // ... everywhere; the current weight slot is free
        X3D_STAMP(4);
        // advance the rings: weights of the slot just filled; A pieces of the slot after the one just read; DMA targets
        {
            const unsigned nb = b_slot;
            b_slot ^= 1u;
            a_read = a_read + 1 == A_SLOTS ? 0 : a_read + 1;
            a_slot = a_slot + 1 == A_SLOTS ? 0 : a_slot + 1;
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                vb[ks] = lds_base + A_RING + nb * B_BYTES + f_off[ks];
                va[ks][0] = lds_base + a_read * A_BYTES + fa_off[ks][0];
                va[ks][1] = lds_base + a_read * A_BYTES + fa_off[ks][1];
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");          // MFMA results -> the epilogue's reads (the compiler does not see the MFMAs)
#ifdef X3D_PROF
    if ((blockIdx.x == 0 || blockIdx.x == 301 || blockIdx.x == gridDim.x - 9) && lane == 0)
        printf("x3d M %d N %d K %d MW %d CW %d AR %d blk %d wave %d nk %d: dma-issue %lld step0 %lld own-A-wait %lld step1 %lld sync %lld (cycles)\n", g.M, g.N, g.K,
               MW, CW, AR, (int)blockIdx.x, wave, nk, pt[0], pt[1], pt[2], pt[3], pt[4]);
#endif
    } else if constexpr (FMT == 1 && X3D_HAND) {
    // ---- hand-scheduled k loop of the f16 pair format, 128 x 64 tiles.  The interleaved schedule of the bf16 loop above (two A slots, own
    // rows two tiles ahead, every LDS-DMA instruction issued between MFMAs), with TWO planes per operand and THREE MFMAs per column
    // block -- a0 w1 into the low accumulator, a0 w0 into the high one, a1 w0 into the low one -- and the f16 split of the next step's A
    // piece (cvt_pk / two cvt back / two sub / two mul / cvt_pk per float pair) behind them.
    static_assert((CW == 2 || CW == 4) && NQ <= 4, "the f16 pair hand loop is written for 128 x 64 and 128 x 128 tiles");
    constexpr int PJ = 4 / CW;                 // float pairs of the next A piece split behind each column block
    unsigned vb[2], va[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
        vb[ks] = lds_base + A_RING + f_off[ks];
        va[ks][0] = lds_base + A_BYTES + fa_off[ks][0];
        va[ks][1] = lds_base + A_BYTES + fa_off[ks][1];
    }
    unsigned b_slot = 1, a_read = 1;           // (wave-uniform) weight slot the next DMAs go to; A slot read next (tile kt + 1's rows)
    uint4 fbq[2][2], rawq[2][2];               // weight fragments (block parity, plane); A piece halves (step parity, half)
    unsigned pl[2][2][4];                       // split planes being built (step parity, plane, float pair)
    bf16x8 fa[2][2];
    float ha = 0.f, hb = 0.f;
#define X3F_FB(KS, J, SET) do { x3h_lds128<(0 * BN + (J) * 32) * XROW>(fbq[SET][0], vb[KS]); \
                                x3h_lds128<(1 * BN + (J) * 32) * XROW>(fbq[SET][1], vb[KS]); } while (0)
#define X3F_RAW(KS, SET) do { x3h_lds128<0>(rawq[SET][0], va[KS][0]); x3h_lds128<0>(rawq[SET][1], va[KS][1]); } while (0)
#define X3F_PAIR_A(SET, Q) do { const uint4 v_ = rawq[SET][(Q) >> 1]; \
        x3h_splitf_a(__uint_as_float(((Q) & 1) ? v_.z : v_.x), __uint_as_float(((Q) & 1) ? v_.w : v_.y), pl[SET][0][Q], ha, hb); } while (0)
#define X3F_PAIR_B(SET, Q) do { const uint4 v_ = rawq[SET][(Q) >> 1]; \
        x3h_splitf_b(__uint_as_float(((Q) & 1) ? v_.z : v_.x), __uint_as_float(((Q) & 1) ? v_.w : v_.y), ha, hb, pl[SET][1][Q]); } while (0)
#define X3F_PACK(SET) do { _Pragma("unroll") for (int p_ = 0; p_ < 2; p_++) \
        fa[SET][p_] = __builtin_bit_cast(bf16x8, make_uint4(pl[SET][p_][0], pl[SET][p_][1], pl[SET][p_][2], pl[SET][p_][3])); } while (0)
#define X3F_BLOCK(J, FS, CS, H1, H2) do { \
        const bf16x8 b0_ = __builtin_bit_cast(bf16x8, fbq[CS][0]), b1_ = __builtin_bit_cast(bf16x8, fbq[CS][1]); \
        x3h_mfma_f16(acc_lo[J], fa[FS][1], b0_); X3F_PAIR_A((FS) ^ 1, PJ * (J)); \
        x3h_mfma_f16(acc[J], fa[FS][0], b0_); X3F_PAIR_B((FS) ^ 1, PJ * (J)); H1; \
        x3h_mfma_f16(acc_lo[J], fa[FS][0], b1_); if (PJ == 2) { X3F_PAIR_A((FS) ^ 1, PJ * (J) + 1); X3F_PAIR_B((FS) ^ 1, PJ * (J) + 1); } H2; } while (0)
#define X3F_DB(I) do { if ((I) < NQ && more_b) x3_asm_dma16((const void*)(b_src[(I) < NQ ? (I) : 0] + (kt + 1) * XBK), b_wave + b_slot * B_BYTES + (I) * 1024u); } while (0)
#define X3F_DA(I) do { if (more_a) x3_asm_dma16((const void*)(a_src[I] + (kt + 2) * XBK), a_wave + a_tgt * A_BYTES + (I) * 1024u); } while (0)
#define X3F_VM(N) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory")
#define X3F_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    // prologue: weights and own rows of tile 0, own rows of tile 1
    dma_b(0, 0);
    dma_a(0, 0);
    if (nk > 1) { dma_a(1, 1); X3F_VM(4); } else X3F_VM(0);
    X3F_BARRIER();
    {
        unsigned a00 = lds_base + fa_off[0][0], a01 = lds_base + fa_off[0][1], a10 = lds_base + fa_off[1][0], a11 = lds_base + fa_off[1][1];
        x3h_lds128<0>(rawq[0][0], a00); x3h_lds128<0>(rawq[0][1], a01);
        x3h_lds128<0>(rawq[1][0], a10); x3h_lds128<0>(rawq[1][1], a11);
    }
    X3H_LGKM(0);
#pragma unroll
    for (int q = 0; q < 4; q++) { X3F_PAIR_A(0, q); X3F_PAIR_B(0, q); }
    X3F_PACK(0);
    for (int kt = 0; kt < nk; kt++) {
        const bool more_b = kt + 1 < nk, more_a = kt + 2 < nk;     // wave-uniform
        const unsigned a_tgt = a_read ^ 1u;                         // tile kt's own rows were read (raw) half a tile ago: their slot takes tile kt + 2
        X3F_FB(0, 0, 0);
        // step 0 (fa[0]); behind it: the split of this tile's second A piece (raw[1] -> fa[1]) and the weight DMAs of tile kt + 1
        if constexpr (CW == 2) {
            X3F_FB(0, 1, 1); X3H_LGKM(2); X3F_BLOCK(0, 0, 0, X3F_DB(0), X3F_DB(1));
            X3F_FB(1, 0, 0); X3H_LGKM(2); X3F_BLOCK(1, 0, 1, (void)0, (void)0);
        } else {
            X3F_FB(0, 1, 1); X3H_LGKM(2); X3F_BLOCK(0, 0, 0, X3F_DB(0), X3F_DB(1));
            X3F_FB(0, 2, 0); X3H_LGKM(2); X3F_BLOCK(1, 0, 1, X3F_DB(2), X3F_DB(3));
            X3F_FB(0, 3, 1); X3H_LGKM(2); X3F_BLOCK(2, 0, 0, (void)0, (void)0);
            X3F_FB(1, 0, 0); X3H_LGKM(2); X3F_BLOCK(3 % CW, 0, 1, (void)0, (void)0);
        }
        X3F_PACK(1);
        if (more_b) X3F_VM(NQ); else X3F_VM(0);                     // this wave's rows of tile kt + 1 have landed (issued a tile ago; only W(kt + 1) is younger)
        X3F_RAW(0, 0);
        X3F_RAW(1, 1);
        // step 1 (fa[1]); behind it: the split of the next tile's first A piece (raw[0] -> fa[0]) and the own-row DMAs of tile kt + 2
        if constexpr (CW == 2) {
            X3F_FB(1, 1, 1); X3H_LGKM(2); X3F_BLOCK(0, 1, 0, X3F_DA(0), X3F_DA(1));
            X3H_LGKM(0); X3F_BLOCK(1, 1, 1, X3F_DA(2), X3F_DA(3));
        } else {
            X3F_FB(1, 1, 1); X3H_LGKM(2); X3F_BLOCK(0, 1, 0, X3F_DA(0), (void)0);
            X3F_FB(1, 2, 0); X3H_LGKM(2); X3F_BLOCK(1, 1, 1, X3F_DA(1), (void)0);
            X3F_FB(1, 3, 1); X3H_LGKM(2); X3F_BLOCK(2, 1, 0, X3F_DA(2), (void)0);
            X3H_LGKM(0); X3F_BLOCK(3 % CW, 1, 1, X3F_DA(3), (void)0);
        }
        X3F_PACK(0);
        if (more_a) X3F_VM(4); else X3F_VM(0);                      // W(kt + 1) has landed (the rows of tile kt + 2 stay in flight)
        X3F_BARRIER();
        {
            const unsigned nb = b_slot;
            b_slot ^= 1u;
            a_read ^= 1u;
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                vb[ks] = lds_base + A_RING + nb * B_BYTES + f_off[ks];
                va[ks][0] = lds_base + a_read * A_BYTES + fa_off[ks][0];
                va[ks][1] = lds_base + a_read * A_BYTES + fa_off[ks][1];
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");          // MFMA results -> the epilogue's reads (the compiler does not see the MFMAs)
    } else {
    // ---- compiler-scheduled k loop (one / two planes per operand: cfg.compute_dtype 'bf16' / 'bf16x2'; development builds -DX3D_HAND=0)
    static_assert(AR == 2, "the compiler-scheduled loop uses the two-slot A ring");
    constexpr int PJ = 4 / CW;
    static_assert(CW == 2 || CW == 4, "the split is spread over 2 or 4 column blocks");
    bf16x8 fa[2][NP], fb[2][NP];
    float4 raw[2][2];                                          // the two 16-byte halves of the A piece of a 16-k step
    unsigned pl[3][4];
    auto read_raw = [&](const unsigned char* Ab, int ks, int set) {
        raw[set][0] = *(const float4*)(Ab + fa_off[ks][0]);
        raw[set][1] = *(const float4*)(Ab + fa_off[ks][1]);
    };
    auto split_pair = [&](int set, int q) {                   // float pair q (0..3) of raw[set] -> pl[.][q]
        const float4 v = raw[set][q >> 1];
        if (FMT == 1) {
            if (q & 1) x3_split2_f16(v.z, v.w, pl[0][q], pl[1][q]);
            else x3_split2_f16(v.x, v.y, pl[0][q], pl[1][q]);
        } else {
            if (q & 1) x3_split2(v.z, v.w, pl[0][q], pl[1][q], pl[2][q]);
            else x3_split2(v.x, v.y, pl[0][q], pl[1][q], pl[2][q]);
        }
    };
    auto pack_fa = [&](int set) {
#pragma unroll
        for (int p = 0; p < NP; p++) fa[set][p] = __builtin_bit_cast(bf16x8, make_uint4(pl[p][0], pl[p][1], pl[p][2], pl[p][3]));
    };
    auto read_fb = [&](const unsigned char* Bb, int ks, int jj, int set) {
#pragma unroll
        for (int p = 0; p < NP; p++)
            fb[set][p] = __builtin_bit_cast(bf16x8, *(const uint4*)(Bb + (p * BN + jj * 32) * XROW + f_off[ks]));
    };
    auto step = [&](const unsigned char* Bb, int ks, int fs, const unsigned char* Bn, int ksn, bool have_next) {
#pragma unroll
        for (int jj = 0; jj < CW; jj++) {
            const int cs = jj & 1;
            if (jj + 1 < CW) read_fb(Bb, ks, jj + 1, cs ^ 1);
            else if (have_next) read_fb(Bn, ksn, 0, cs ^ 1);
#define X3D_TERM(PA, PB) acc[jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[fs][PA], fb[cs][PB], acc[jj], 0, 0, 0);
#define X3D_TERM_F16(ACC, PA, PB) ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(x3_f16x8, fa[fs][PA]), \
                                                                                __builtin_bit_cast(x3_f16x8, fb[cs][PB]), ACC, 0, 0, 0);
            if constexpr (FMT == 1) {
                X3D_TERM_F16(acc_lo[jj], 1, 0) X3D_TERM_F16(acc[jj], 0, 0) X3D_TERM_F16(acc_lo[jj], 0, 1)      // (alternating accumulators)
            } else {
            if (NP == 3) { X3D_TERM(2, 0) X3D_TERM(1, 1) X3D_TERM(0, 2) }
            if (NP >= 2) { X3D_TERM((NP >= 2 ? 1 : 0), 0) X3D_TERM(0, (NP >= 2 ? 1 : 0)) }
            X3D_TERM(0, 0)
            }
#undef X3D_TERM
#undef X3D_TERM_F16
#pragma unroll
            for (int q = 0; q < PJ; q++) split_pair(fs ^ 1, jj * PJ + q);
        }
        pack_fa(fs ^ 1);
    };
    const unsigned char* A0 = &sm[0];
    const unsigned char* A1 = &sm[A_BYTES];
    const unsigned char* B0 = &sm[A_RING];
    const unsigned char* B1 = &sm[A_RING + B_BYTES];
#define X3D_SYNC() asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory")
#define X3D_OWN_A() asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NQ) : "memory")      // this wave's A rows of the next tile have landed (issued before the weights)
    dma_a(0, 0); dma_b(0, 0);
    X3D_SYNC();
    read_raw(A0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; q++) split_pair(0, q);
    pack_fa(0);
    read_raw(A0, 1, 1);
    read_fb(B0, 0, 0, 0);
    for (int kt = 0; kt < nk; kt += 2) {
        if (kt + 1 < nk) { dma_a(kt + 1, 1); dma_b(kt + 1, 1); }
        step(B0, 0, 0, B0, 1, true);
        X3D_OWN_A();
        read_raw(A1, 0, 0);                                    // (stale data after the last tile: split, never used)
        step(B0, 1, 1, B1, 0, false);
        read_raw(A1, 1, 1);
        X3D_SYNC();                                            // tile kt + 1 has landed everywhere; slot 0 is free
        if (kt + 1 >= nk) break;
        read_fb(B1, 0, 0, 0);
        if (kt + 2 < nk) { dma_a(kt + 2, 0); dma_b(kt + 2, 0); }
        step(B1, 0, 0, B1, 1, true);
        X3D_OWN_A();
        read_raw(A0, 0, 0);
        step(B1, 1, 1, B0, 0, false);
        read_raw(A0, 1, 1);
        X3D_SYNC();
        read_fb(B0, 0, 0, 0);
    }
#undef X3D_SYNC
#undef X3D_OWN_A
    }

    // ---- epilogue (C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5))
    // SOUT: the finished values also go to an LDS image of the tile (row-major, BN floats per row: a store instruction covers 2 x 32
    // consecutive floats -- conflict free), from which the per-cloud column sums are taken once the accumulators are dead: doing the
    // float64 sums on the live accumulators put every statistics variant at the 256-register cliff (the CW = 2 one spilled around the
    // asm loop and broke it).
    if constexpr (FMT == 1) {                              // f16 pair: fold the scaled low terms in
        float chk = 0.f;
#pragma unroll
        for (int j = 0; j < CW; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) { acc[j][r] += acc_lo[j][r] * (1.0f / X3_F16_SCALE); chk = fmaf(acc[j][r], 0.f, chk); }
        x3_report_range(g.status, chk);
    }
    float* T = (float*)&sm[0];                             // [BM][BN], free after the k loop's last barrier
#pragma unroll
    for (int j = 0; j < CW; j++) {
        const int col = n0 + j * 32 + l31;
        const int rloc = wave * 32 + 4 * hi;
        if (n0 + j * 32 >= g.N) continue;                  // (wave-uniform) column block beyond N: the thin N = 32 case
        if (g.partial) {   // split-K: raw accumulators, epilogue happens in k_x3_splitk_reduce
            float* P = g.partial + (size_t)blockIdx.z * g.M * g.N;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = m0 + rloc + (r & 3) + 8 * (r >> 2);
                if (row < g.M) P[(size_t)row * g.N + col] = acc[j][r];
            }
            continue;
        }
        const float bv = g.bias ? g.bias[col] : 0.f;
        // row_div / residual come from CLAMPED rows, all sixteen loads in flight before the first use: a load under the per-lane
        // `row < M` predicate compiles to load + s_waitcnt vmcnt(0) -- sixteen serial memory round trips per column block.
        float rd[16], rs[16];
        if (g.row_div) {
#pragma unroll
            for (int r = 0; r < 16; r++) rd[r] = g.row_div[min(m0 + rloc + (r & 3) + 8 * (r >> 2), g.M - 1)];
        }
        if (g.residual) {
#pragma unroll
            for (int r = 0; r < 16; r++) rs[r] = g.residual[(size_t)min(m0 + rloc + (r & 3) + 8 * (r >> 2), g.M - 1) * g.ldr + col];
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int rl = rloc + (r & 3) + 8 * (r >> 2), row = m0 + rl;
            float v = acc[j][r];
            if (g.row_div) v = v / rd[r];
            v += bv;
            if (g.act == 1) v = fmaxf(v, 0.f);
            if (g.residual) v += rs[r];
            if (row < g.M) g.C[(size_t)row * g.ldc + col] = v;
            else v = 0.f;
            if (SOUT) T[rl * BN + j * 32 + l31] = v;
        }
    }
    if constexpr (SOUT) {
        // per cloud s owning rows of this tile: per-column (sum, sum of squares) in float64 over the tile's rows of s ->
        // stat_partial[(tile_m + s) * N + col] (slot tile_m + s is unique: both only grow along M); regtr_instnorm_finalize_tiles adds
        // the slots of a cloud in fixed order.  NT threads = PARTS row classes x BN columns, fixed order everywhere: deterministic.
        constexpr int PARTS = NT / BN;
        static_assert(BM * BN * 4 + PARTS * BN * sizeof(double2) <= LDS_BYTES, "tile image + partial sums must fit the operand rings");
        double2* red = (double2*)&sm[BM * BN * 4];
        const int c = t % BN, part = t / BN;
        __syncthreads();
        for (int sg = s_lo; sg <= s_hi; sg++) {            // workgroup-uniform; one cloud per tile almost always
            const int r_lo = max((sg == s_lo ? s_lo_begin : g.stat_seg_off[sg]) - m0, 0);
            const int r_hi = min(min(sg == s_lo ? s_lo_end : g.stat_seg_off[sg + 1], g.M) - m0, BM);
            double sm_ = 0.0, sq = 0.0;
            if (n0 + c < g.N)
                for (int r = r_lo + part; r < r_hi; r += PARTS) { const double v = (double)T[r * BN + c]; sm_ += v; sq += v * v; }
            red[part * BN + c] = make_double2(sm_, sq);
            __syncthreads();
            if (t < BN && n0 + t < g.N) {
                double2 a = red[t];
#pragma unroll
                for (int w = 1; w < PARTS; w++) { const double2 b = red[w * BN + t]; a.x += b.x; a.y += b.y; }
                g.stat_partial[(size_t)(tile_m + sg) * g.N + n0 + t] = a;
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// 64 x 64 tiles of SMALL problems (a pair or two per forward), operand pipeline THREE k-tiles deep (round 5).
// k_gemm_x3 issues the loads of tile t + 1 before the MFMAs of tile t and meets a __syncthreads() -- s_waitcnt vmcnt(0) -- right
// behind them, so every k-tile pays a full memory round trip: a K = 256 product is ~10 us of which 0.2 us per tile are MFMAs, and with a few
// dozen workgroups on 256 CUs nothing else hides it.  69 such launches are a third of a one-pair forward's GPU time.  Here BOTH operands
// arrive by LDS-DMA (raw float32 A rows, XOR-swizzled like k_gemm_x3d's; weight planes as in k_gemm_x3) into a FOUR-slot ring, the DMA of
// tile t + 3 is issued while tile t is multiplied, and the one barrier per k-tile is preceded by a COUNTED wait -- vmcnt(2 tiles' worth) --
// that leaves tiles t + 1 and t + 2 in flight (the DMA instructions are inline asm: the compiler neither tracks nor waits for them).  A is
// split into its planes at fragment time by the wave that multiplies it (twice per element: the two column waves of a row block; VALU is idle
// here).  No folded InstanceNorm operand (that launch stays on k_gemm_x3); statistics epilogue and split-K partial products as there.
template <bool SOUT, int NP, int FMT>
__global__ void __launch_bounds__(256, 2) k_gemm_x3q(X3Args g)
{
    static_assert((FMT == 0 && NP == 3) || (FMT == 1 && NP == 2), "float32-grade formats only: bf16x3 or the f16 pair");
    constexpr int BM = 64, BN = 64, NT = 256, SLOTS = 4, PD = 3;
    constexpr int A_BYTES = BM * XBK * 4, B_BYTES = NP * BN * XROW, SLOT_BYTES = A_BYTES + B_BYTES;
    constexpr int NA = A_BYTES / 1024 / 4, NQ = B_BYTES / 1024 / 4, N_TILE = NA + NQ;       // LDS-DMA instructions per wave and k-tile
    constexpr int STAT_BYTES = SOUT ? 2 * BN * 16 : 0;
    constexpr int LDS_BYTES = SLOTS * SLOT_BYTES > STAT_BYTES ? SLOTS * SLOT_BYTES : STAT_BYTES;
    __shared__ __align__(1024) unsigned char sm[LDS_BYTES];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    int tile_m, tile_n;
    {
        const int nc = g.N / BN, b = blockIdx.x, x = b & 7, q = b >> 3;       // XCD-aware map, as in k_gemm_x3
        tile_n = q % nc;
        tile_m = (q / nc) * 8 + x;
        if (tile_m * BM >= g.M) return;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int k_begin = blockIdx.z * g.k_chunk;
    const int k_end = min(g.K, k_begin + g.k_chunk);
    const int nk = (k_end - k_begin) / XBK;                    // whole k-tiles only (the launcher checks)
    int s_lo = 0, s_hi = 0, s_lo_begin = 0, s_lo_end = 0;
    if (SOUT) {
        if (g.tile_info) {
            const int4 ti = g.tile_info[tile_m];
            s_lo = ti.x; s_hi = ti.y; s_lo_begin = ti.z; s_lo_end = ti.w;
        } else {
            const int row_last = min(m0 + BM, g.M) - 1;
            s_lo = rg_find_segment_wave(g.stat_seg_off, g.n_stat_seg, m0);
            s_hi = rg_find_segment_wave(g.stat_seg_off, g.n_stat_seg, row_last);
            s_lo_begin = g.stat_seg_off[s_lo]; s_lo_end = g.stat_seg_off[s_lo + 1];
        }
    }
    // A DMA: instruction q of this wave fills rows 16 wave + 8 q + (lane >> 3), LDS chunk lane & 7 <- source chunk (lane & 7) ^ ((row >> 1) & 7)
    const float* a_src[NA];
#pragma unroll
    for (int q = 0; q < NA; q++) {
        const int r = wave * (8 * NA) + q * 8 + (lane >> 3);
        const int row = m0 + r, rc = row < g.M ? row : g.M - 1;    // rows past M compute garbage that the epilogue never stores
        a_src[q] = g.A + (size_t)rc * g.lda + k_begin + (((lane & 7) ^ ((r >> 1) & 7)) * 4);
    }
    const uint16_t* b_src[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const int sidx = (wave * NQ + q) * 64 + lane;
        const int p = sidx / (BN * 4), r = sidx % (BN * 4), n = r >> 2, kc = (r & 3) ^ ((n >> 2) & 3);
        b_src[q] = g.Wt + (size_t)p * g.plane + (size_t)(n0 + n) * g.Kp + kc * 8 + k_begin;
    }
    const unsigned lds_base = (unsigned)(unsigned long long)(x3_lds_ptr)(&sm[0]);
    auto dma = [&](int kt, unsigned slot) {
        const unsigned base = lds_base + slot * SLOT_BYTES;
#pragma unroll
        for (int q = 0; q < NA; q++) x3_asm_dma16((const void*)(a_src[q] + kt * XBK), base + (unsigned)(wave * NA + q) * 1024u);
#pragma unroll
        for (int q = 0; q < NQ; q++) x3_asm_dma16((const void*)(b_src[q] + kt * XBK), base + A_BYTES + (unsigned)(wave * NQ + q) * 1024u);
    };
    // fragment addresses: A row 32 wm + l31, 16-byte chunks 4 ks + 2 hi (+ 1), swizzled; weights as in k_gemm_x3
    unsigned fa_off[2][2], f_off[2];
    {
        const unsigned r = (unsigned)wm * 32u + (unsigned)l31, sw = (r >> 1) & 7u;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
#pragma unroll
            for (int h = 0; h < 2; h++) fa_off[ks][h] = r * 128u + ((((unsigned)(4 * ks + 2 * hi + h)) ^ sw) * 16u);
            f_off[ks] = (unsigned)A_BYTES + ((unsigned)wn * 32u + (unsigned)l31) * XROW + ((((unsigned)(2 * ks + hi)) ^ (((unsigned)l31 >> 2) & 3u)) * 16u);
        }
    }
    floatx16 acc, acc_lo;
#pragma unroll
    for (int r = 0; r < 16; r++) { acc[r] = 0.f; acc_lo[r] = 0.f; }

#pragma unroll
    for (int d = 0; d < PD; d++) dma(d < nk ? d : nk - 1, (unsigned)d);
    for (int kt = 0; kt < nk; kt++) {
        // tile kt has landed (this wave's share: tiles kt + 1, kt + 2 are the younger ones in flight), then everybody's; the barrier also says
        // every wave is done with tile kt - 1, whose slot the next DMA refills
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"((PD - 1) * N_TILE) : "memory");
        { const int kn = kt + PD; dma(kn < nk ? kn : nk - 1, (unsigned)(kn & (SLOTS - 1))); }      // (past the end: a harmless re-read keeps the count constant)
        const unsigned char* S = &sm[(kt & (SLOTS - 1)) * SLOT_BYTES];
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            const float4 r0 = *(const float4*)(S + fa_off[ks][0]), r1 = *(const float4*)(S + fa_off[ks][1]);
            unsigned pl[3][4];
            if constexpr (FMT == 1) {
                x3_split2_f16(r0.x, r0.y, pl[0][0], pl[1][0]); x3_split2_f16(r0.z, r0.w, pl[0][1], pl[1][1]);
                x3_split2_f16(r1.x, r1.y, pl[0][2], pl[1][2]); x3_split2_f16(r1.z, r1.w, pl[0][3], pl[1][3]);
            } else {
                x3_split2(r0.x, r0.y, pl[0][0], pl[1][0], pl[2][0]); x3_split2(r0.z, r0.w, pl[0][1], pl[1][1], pl[2][1]);
                x3_split2(r1.x, r1.y, pl[0][2], pl[1][2], pl[2][2]); x3_split2(r1.z, r1.w, pl[0][3], pl[1][3], pl[2][3]);
            }
            bf16x8 fa[NP], fb[NP];
#pragma unroll
            for (int p = 0; p < NP; p++) {
                fa[p] = __builtin_bit_cast(bf16x8, make_uint4(pl[p][0], pl[p][1], pl[p][2], pl[p][3]));
                fb[p] = __builtin_bit_cast(bf16x8, *(const uint4*)(S + p * BN * XROW + f_off[ks]));
            }
            if constexpr (FMT == 1) {
                acc_lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(x3_f16x8, fa[1]), __builtin_bit_cast(x3_f16x8, fb[0]), acc_lo, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(x3_f16x8, fa[0]), __builtin_bit_cast(x3_f16x8, fb[0]), acc, 0, 0, 0);
                acc_lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(x3_f16x8, fa[0]), __builtin_bit_cast(x3_f16x8, fb[1]), acc_lo, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2], fb[0], acc, 0, 0, 0);       // six terms, smallest first (as k_gemm_x3)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[0], acc, 0, 0, 0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");       // the re-reads past the end have landed before the ring is reused or released

    if constexpr (FMT == 1) {                              // f16 pair: fold the scaled low terms in
        float chk = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) { acc[r] += acc_lo[r] * (1.0f / X3_F16_SCALE); chk = fmaf(acc[r], 0.f, chk); }
        x3_report_range(g.status, chk);
    }
    // ---- epilogue (C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)), as k_gemm_x3
    {
        const int col = n0 + wn * 32 + l31;
        const int rbase = m0 + wm * 32 + 4 * hi;
        if (g.partial) {   // split-K: raw accumulators, epilogue happens in k_x3_splitk_reduce[_stats]
            float* P = g.partial + (size_t)blockIdx.z * g.M * g.N;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (row < g.M) P[(size_t)row * g.N + col] = acc[r];
            }
        } else {
            const float bv = g.bias ? g.bias[col] : 0.f;
            float rd[16], rs[16];
            if (g.row_div) {
#pragma unroll
                for (int r = 0; r < 16; r++) rd[r] = g.row_div[min(rbase + (r & 3) + 8 * (r >> 2), g.M - 1)];
            }
            if (g.residual) {
#pragma unroll
                for (int r = 0; r < 16; r++) rs[r] = g.residual[(size_t)min(rbase + (r & 3) + 8 * (r >> 2), g.M - 1) * g.ldr + col];
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                float v = acc[r];
                if (g.row_div) v = v / rd[r];
                v += bv;
                if (g.act == 1) v = fmaxf(v, 0.f);
                if (g.residual) v += rs[r];
                if (row < g.M) g.C[(size_t)row * g.ldc + col] = v;
                acc[r] = v;
            }
        }
    }
    if constexpr (SOUT) {       // per (tile, cloud) column sums of the finished values (float64, fixed order), as k_gemm_x3
        double2* red = (double2*)&sm[0];                   // [2 row blocks][BN]
        for (int sg = s_lo; sg <= s_hi; sg++) {            // workgroup-uniform; one cloud per tile almost always
            const int r_lo = sg == s_lo ? s_lo_begin : g.stat_seg_off[sg];
            const int r_hi = min(sg == s_lo ? s_lo_end : g.stat_seg_off[sg + 1], g.M);
            double sm_ = 0.0, sq = 0.0;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = m0 + wm * 32 + 4 * hi + (r & 3) + 8 * (r >> 2);
                if (row >= r_lo && row < r_hi) { const double v = (double)acc[r]; sm_ += v; sq += v * v; }
            }
            sm_ += __shfl_xor(sm_, 32, RG_WAVE); sq += __shfl_xor(sq, 32, RG_WAVE);
            if (hi == 0) red[wm * BN + wn * 32 + l31] = make_double2(sm_, sq);
            __syncthreads();
            if (t < BN) {
                double2 a = red[t];
                const double2 b = red[BN + t];
                a.x += b.x; a.y += b.y;
                g.stat_partial[(size_t)(tile_m + sg) * g.N + n0 + t] = a;
            }
            __syncthreads();
        }
    }
}

// (Round 4 built an A-RESIDENT strip kernel for the short-K products -- A fragments kept in registers for a whole range of columns, weight
//  planes through an eight-slot LDS ring, one wave per SIMD -- and removed it after measurement: 107-163 us against 92-127 us for k_gemm_x3d on
//  the K = 256 cross-encoder shapes; phase clocks and the reading in profiles/r04_gemm_ares.md.)
__global__ void __launch_bounds__(256) k_x3_splitk_reduce(X3Args g, int S)
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

// The same reduction WITH the InstanceNorm partial sums of the result (round 5): a split-K launch had no statistics epilogue, so a KPConv
// contraction of a small batch was four launches -- product, this reduction, regtr_instnorm_stats' two -- and the statistics pass re-read
// the C just written.  Here one workgroup owns X3_RS_ROWS rows x all N columns: thread (tx = column quad, ty = row phase) adds the S
// partial products of its elements in split order (the same sum as above), applies the epilogue, stores, and keeps per-column (sum, sum of
// squares) in float64 for every cloud owning rows of the tile; the ty phases are added in fixed order through LDS and the tile's slots
// written in regtr_gemm_x3's own layout -- stat_partial[(tile + cloud) * N + col] -- so regtr_instnorm_finalize_tiles(tile_rows =
// x3_rs_rows(N)) finishes the job as for every other launch.  N / 4 must be a power of two <= 256 (every RegTR width).
// Tile height = ONE row per thread row-phase (1024 / N rows: 4 at N = 256): split-K launches are small problems by construction (fewer than
// 384 tiles), and a first version with 32-row tiles left a 751-row level with 24 workgroups walking eight rows each -- 11.9 us against 4.8 +
// 4.7 us for the two kernels it replaced.
static inline int x3_rs_rows(int N) { return N >= 1024 ? 1 : 1024 / N; }

__global__ void __launch_bounds__(256) k_x3_splitk_reduce_stats(X3Args g, int S, int vec_ok, int X3_RS_ROWS)
{
    __shared__ double sh[256 * 8];
    const int C4 = g.N >> 2, TR = 256 / C4;
    const int tx = threadIdx.x % C4, ty = threadIdx.x / C4;
    const int tile = blockIdx.x, r0 = tile * X3_RS_ROWS, r1 = min(g.M, r0 + X3_RS_ROWS);
    const int s_lo = rg_find_segment(g.stat_seg_off, g.n_stat_seg, r0), s_hi = rg_find_segment(g.stat_seg_off, g.n_stat_seg, r1 - 1);
    const int col = 4 * tx;
    const size_t plane = (size_t)g.M * g.N;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.bias) bv = *(const float4*)(g.bias + col);
    for (int sg = s_lo; sg <= s_hi; sg++) {                              // workgroup-uniform; one cloud per tile almost always
        const int lo = max(r0, g.stat_seg_off[sg]), hi = min(r1, g.stat_seg_off[sg + 1]);
        if (lo >= hi) continue;                                           // an empty cloud between two others
        double sm[4] = {0, 0, 0, 0}, sq[4] = {0, 0, 0, 0};
        for (int r = lo + ty; r < hi; r += TR) {
            const float* P = g.partial + (size_t)r * g.N + col;
            float4 a = *(const float4*)P;
            for (int k = 1; k < S; k++) {                                 // fixed order: deterministic
                const float4 b = *(const float4*)(P + (size_t)k * plane);
                a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            }
            float v[4] = {a.x, a.y, a.z, a.w};
            if (g.row_div) { const float d = g.row_div[r];
#pragma unroll
                for (int j = 0; j < 4; j++) v[j] = v[j] / d; }
            v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
            if (g.act == 1) {
#pragma unroll
                for (int j = 0; j < 4; j++) v[j] = fmaxf(v[j], 0.f);
            }
            if (g.residual) {
#pragma unroll
                for (int j = 0; j < 4; j++) v[j] += g.residual[(size_t)r * g.ldr + col + j];
            }
            if (vec_ok) *(float4*)(g.C + (size_t)r * g.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
            else {
#pragma unroll
                for (int j = 0; j < 4; j++) g.C[(size_t)r * g.ldc + col + j] = v[j];
            }
#pragma unroll
            for (int j = 0; j < 4; j++) { sm[j] += (double)v[j]; sq[j] += (double)v[j] * (double)v[j]; }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) { sh[threadIdx.x * 8 + j] = sm[j]; sh[threadIdx.x * 8 + 4 + j] = sq[j]; }
        __syncthreads();
        if (ty == 0) {
            for (int y = 1; y < TR; y++)
#pragma unroll
                for (int j = 0; j < 4; j++) { sm[j] += sh[(y * C4 + tx) * 8 + j]; sq[j] += sh[(y * C4 + tx) * 8 + 4 + j]; }
            double2* o = g.stat_partial + (size_t)(tile + sg) * g.N + col;
#pragma unroll
            for (int j = 0; j < 4; j++) o[j] = make_double2(sm[j], sq[j]);
        }
        __syncthreads();
    }
}

static inline bool x3_rs_ok(int N) { const int c4 = N >> 2; return N % 4 == 0 && c4 >= 1 && c4 <= 256 && (c4 & (c4 - 1)) == 0; }

// W (rows x cols, leading dimension ld) -> Wt[p][n][k]: n = row (transposed == 0: W is [N, K], an nn.Linear weight) or
// n = col (transposed == 1: W is [K, N]).  Padding (k >= K, n >= N) is zero.
__global__ void __launch_bounds__(256) k_split_weights(const float* __restrict__ W, int ld, int N, int K, int transposed,
                                                       int Npad, int Kp, uint16_t* __restrict__ Wt)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)Npad * Kp) return;
    const int n = (int)(e / Kp), k = (int)(e % Kp);
    float w = 0.f;
    if (n < N && k < K) w = transposed ? W[(size_t)k * ld + n] : W[(size_t)n * ld + k];
    unsigned p0, p1, p2;
    x3_split2(w, 0.f, p0, p1, p2);
    const size_t plane = (size_t)Npad * Kp;
    Wt[e] = (uint16_t)(p0 & 0xffffu); Wt[plane + e] = (uint16_t)(p1 & 0xffffu); Wt[2 * plane + e] = (uint16_t)(p2 & 0xffffu);
}

// the same for the f16 pair format: Wt[2][Npad][Kp] f16 (plane 1 = scaled residual)
__global__ void __launch_bounds__(256) k_split_weights_f16(const float* __restrict__ W, int ld, int N, int K, int transposed,
                                                           int Npad, int Kp, uint16_t* __restrict__ Wt)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)Npad * Kp) return;
    const int n = (int)(e / Kp), k = (int)(e % Kp);
    float w = 0.f;
    if (n < N && k < K) w = transposed ? W[(size_t)k * ld + n] : W[(size_t)n * ld + k];
    unsigned p0, p1;
    x3_split2_f16(w, 0.f, p0, p1);
    const size_t plane = (size_t)Npad * Kp;
    Wt[e] = (uint16_t)(p0 & 0xffffu); Wt[plane + e] = (uint16_t)(p1 & 0xffffu);
}

// tile t of `rows` rows -> (first cloud, last cloud, first cloud's begin, first cloud's end); one thread per tile
__global__ void __launch_bounds__(256) k_tile_segments(const int* __restrict__ seg_off, int n_seg, int M, int rows, int4* __restrict__ out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if ((long long)t * rows >= M) return;
    const int r0 = t * rows, r1 = min(M, r0 + rows) - 1;
    const int lo = rg_find_segment(seg_off, n_seg, r0), hi = rg_find_segment(seg_off, n_seg, r1);
    out[t] = make_int4(lo, hi, seg_off[lo], seg_off[lo + 1]);
}

struct X3Plan { int tile, splits, k_chunk; bool strip; };      // tile: 0 = 128 x 128, 1 = 128 x 64, 2 = 64 x 64; strip: the row-strip kernel where eligible

// tile shape and K split for a problem (host policy)
X3Plan x3_plan(int M, int N, int K)
{
    X3Plan p{2, 1, K, false};
    if (N == 32) { p.tile = 1; p.strip = true; return p; }     // thin: one 64-column strip tile per 128 rows (regtr_gemm_x3_supported)
    static const int strip_on = X3_DEV_ENV("REGTR_X3_STRIP", 1);   // development: A/B runs
    static const int forced = X3_DEV_ENV("REGTR_X3_TILE", -1);   // development: tile A/B runs
    static const int forced_s = X3_DEV_ENV("REGTR_X3_SPLITS", 0);   // development
    if (forced >= 0 && forced <= 2 && (forced != 0 || N % 128 == 0)) {
        p.tile = forced;
        p.strip = strip_on && p.tile != 2;
        if (forced_s >= 2 && K / forced_s >= 64) {
            p.k_chunk = rg_cdiv(rg_cdiv(K, forced_s), XBK) * XBK;
            p.splits = rg_cdiv(K, p.k_chunk);
        }
        return p;
    }
    auto tiles = [&](int bm, int bn) { return (long long)rg_cdiv(M, bm) * (N / bn); };
    // Measured on MI355X (REGTR_X3_TILE = 0 / 1 / 2 sweeps over RegTR's shapes at 16 and 64 pairs per forward): with a few
    // hundred tiles the 64 x 64 tile at four workgroups per CU wins (occupancy hides the per-k-tile latency); from about
    // six tiles per CU on, the larger tiles' lower operand traffic per flop pays: 128 x 64 first for deep K, then the
    // 8-wave 128 x 128 tile.
    p.tile = 2;
    if (tiles(128, 64) >= 512 && (K >= 960 || tiles(128, 64) >= 2048)) p.tile = 1;
    if (N % 128 == 0 && tiles(128, 128) >= 1536) p.tile = 0;
    // deep K on the row-strip kernel: the 128-column tile reads A half as often (level-2 / level-3 KPConv contractions: 483 vs 511, 478 vs 481 us)
    if (strip_on && N % 128 == 0 && K >= 1536 && K % XBK == 0 && tiles(128, 128) >= 512) p.tile = 0;
    const long long tl = p.tile == 0 ? tiles(128, 128) : (p.tile == 1 ? tiles(128, 64) : tiles(64, 64));
    if (tl < 384 && K >= 512) {
        int s = (int)((768 + tl - 1) / tl);
        const int max_by_k = K / 256;          // keep >= 256 of K per split
        if (s > max_by_k) s = max_by_k;
        if (s > 8) s = 8;
        if (s >= 2) {
            p.k_chunk = rg_cdiv(rg_cdiv(K, s), XBK) * XBK;
            p.splits = rg_cdiv(K, p.k_chunk);
        }
    }
    p.strip = strip_on && p.tile != 2;
    return p;
}

}  // namespace

extern "C" {

// 1 when regtr_gemm_x3 accepts the shape (otherwise use regtr_gemm_f32)
// (N = 32 -- the level-0 KPConv contractions -- only on the row-strip kernel: K a multiple of 32, no InstanceNorm folded into A)
int regtr_gemm_x3_supported(int M, int N, int K)
{
    if (M < 0) return 0;
    if (N == 32) return (K >= 64 && K % XBK == 0) ? 1 : 0;
    return (N >= 64 && N % 64 == 0 && K >= 16 && K % 4 == 0) ? 1 : 0;
}

// Per row tile of `rows` rows (regtr_gemm_x3_stat_tile_rows / the launch's tile height): first and last cloud owning rows of the
// tile and the first cloud's row range, 16 bytes per tile.  Built once per pyramid level and handed to every regtr_gemm_x3 launch
// over that level's rows (tile_info), it replaces the boundary search + dependent offset loads at the head of every workgroup.
// Empty clouds are skipped by the search (the containing cloud of a row is the last one starting at or before it).
int regtr_tile_segments(const int* seg_off, int n_seg, int M, int rows, void* out, void* stream)
{
    if (!seg_off || !out || n_seg < 1 || M < 0 || rows < 1) return RG_ERR_ARG;
    if (M == 0) return RG_OK;
    k_tile_segments<<<rg_cdiv(rg_cdiv(M, rows), 256), 256, 0, (hipStream_t)stream>>>(seg_off, n_seg, M, rows, (int4*)out);
    RG_RETURN_IF_LAUNCH_FAILED();
    return RG_OK;
}

// tile height of the regtr_gemm_x3 launch for this shape (the `rows` of its tile_info)
int regtr_gemm_x3_tile_rows(int M, int N, int K)
{
    if (!regtr_gemm_x3_supported(M, N, K)) return 0;
    return x3_plan(M, N, K).tile == 2 ? 64 : 128;
}

// 1 when the split kernel is also the FASTER choice (measured on MI355X, tools/microbench.py): every supported shape with
// at least one full k-tile; thinner contractions are pure streaming and stay on the exact-f32 kernel
int regtr_gemm_x3_preferred(int M, int N, int K)
{
    if (!regtr_gemm_x3_supported(M, N, K)) return 0;
    if (N == 32) return 0;      // measured: the level-0 contraction ([2.4 M x 480] x [480 x 32], a pure A stream) 1.20 ms here vs 1.14 ms on the exact-f32 tiled kernel
    return K >= 32 ? 1 : 0;
}

#ifdef REGTR_EXPERIMENTAL
// Diagnostic: workgroups of the row-strip kernel the runtime will keep resident per CU (hipOccupancyMaxActiveBlocksPerMultiprocessor)
// for column width cw (2 | 4 blocks of 32), A-ring mode ar (2 | 3; 4 = interleaved, cw 4 only) and the statistics epilogue; -1 = no such variant.
int regtr_gemm_x3_strip_occupancy(int cw, int ar, int stats)
{
    int n = -1;
    const void* k = nullptr;
    if (cw == 4 && ar == 2) k = stats ? (const void*)k_gemm_x3d<4, 4, 2, true> : (const void*)k_gemm_x3d<4, 4, 2, false>;
    else if (cw == 4 && ar == 4) k = stats ? (const void*)k_gemm_x3d<4, 4, 4, true> : (const void*)k_gemm_x3d<4, 4, 4, false>;
    else if (cw == 2 && ar == 2) k = stats ? (const void*)k_gemm_x3d<4, 2, 2, true> : (const void*)k_gemm_x3d<4, 2, 2, false>;
    else if (cw == 2 && ar == 3) k = stats ? (const void*)k_gemm_x3d<4, 2, 3, true> : (const void*)k_gemm_x3d<4, 2, 3, false>;
    if (!k || hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, 0) != hipSuccess) return -1;
    return n;
}
#endif  // REGTR_EXPERIMENTAL

size_t regtr_gemm_split_weights_bytes(int N, int K)
{
    const size_t Npad = (size_t)rg_cdiv(N, 128) * 128, Kp = (size_t)rg_cdiv(K, XBK) * XBK;
    return 3 * Npad * Kp * sizeof(uint16_t);
}

// W: float32 weights, [N, K] row-major (transposed = 0, an nn.Linear weight as stored) or [K, N] (transposed = 1, e.g. a
// KPConv weight viewed as [15 Cin, Cout]).  planes: regtr_gemm_split_weights_bytes(N, K) bytes.
int regtr_gemm_split_weights(const float* W, int ld, int N, int K, int transposed, void* planes, void* stream)
{
    if (!W || !planes || N < 1 || K < 1 || ld < (transposed ? N : K)) return RG_ERR_ARG;
    const int Npad = rg_cdiv(N, 128) * 128, Kp = rg_cdiv(K, XBK) * XBK;
    k_split_weights<<<rg_cdiv((long long)Npad * Kp, 256), 256, 0, (hipStream_t)stream>>>(W, ld, N, K, transposed, Npad, Kp,
                                                                                         (uint16_t*)planes);
    RG_RETURN_IF_LAUNCH_FAILED();
    return RG_OK;
}

// Tile plan of the f16 pair format (regtr_gemm_x3 with n_planes = 4): the bf16 plan (strip or tiled kernel alike); without the
// statistics epilogue -- whose slot height is tied to the bf16 plan -- a 64 x 64-tile plan becomes 128 x 64 strips as soon as there
// are 512 of them (the 37.9 k-token out-projection of the cross-encoder: 1184 strips)
static bool x3_plan_f16(int M, int N, int K, bool with_stats, X3Plan& p)
{
    if (!regtr_gemm_x3_supported(M, N, K)) return false;          // (N = 32: the thin strip form, K a multiple of 32, no folded operand)
    p = x3_plan(M, N, K);
    if (!p.strip && !with_stats && p.splits == 1 && K % XBK == 0 && (long long)rg_cdiv(M, 128) * (N / 64) >= 512) { p.tile = 1; p.strip = true; }
    return true;
}

int regtr_gemm_x3_f16_supported(int M, int N, int K, int with_stats)
{
    X3Plan p;
    return x3_plan_f16(M, N, K, with_stats != 0, p) ? 1 : 0;
}

size_t regtr_gemm_split_weights_f16_bytes(int N, int K)
{
    const size_t Npad = (size_t)rg_cdiv(N, 128) * 128, Kp = (size_t)rg_cdiv(K, XBK) * XBK;
    return 2 * Npad * Kp * sizeof(uint16_t);
}

int regtr_gemm_split_weights_f16(const float* W, int ld, int N, int K, int transposed, void* planes, void* stream)
{
    if (!W || !planes || N < 1 || K < 1 || ld < (transposed ? N : K)) return RG_ERR_ARG;
    const int Npad = rg_cdiv(N, 128) * 128, Kp = rg_cdiv(K, XBK) * XBK;
    k_split_weights_f16<<<rg_cdiv((long long)Npad * Kp, 256), 256, 0, (hipStream_t)stream>>>(W, ld, N, K, transposed, Npad, Kp,
                                                                                             (uint16_t*)planes);
    RG_RETURN_IF_LAUNCH_FAILED();
    return RG_OK;
}

size_t regtr_gemm_x3_ws_bytes(int M, int N, int K)
{
    if (!regtr_gemm_x3_supported(M, N, K)) return 0;
    const X3Plan p = x3_plan(M, N, K);
    return p.splits > 1 ? (size_t)p.splits * M * N * sizeof(float) : 0;
}

// Same contract as regtr_gemm_f32 with B given as the planes written by regtr_gemm_split_weights(W, .., N, K, ..).
// rows per statistics tile when regtr_gemm_x3 can emit InstanceNorm partial sums for this shape (stat_partial): the launch's tile height
// when they come from the GEMM epilogue, 1024 / N when a split-K launch's reduction kernel writes them; 0 when it cannot (split-K with a
// column count the reduction's thread map does not take): the caller then runs regtr_instnorm_stats on C instead.
int regtr_gemm_x3_stat_tile_rows(int M, int N, int K)
{
    if (!regtr_gemm_x3_supported(M, N, K) || M < 1) return 0;
    const X3Plan p = x3_plan(M, N, K);
    if (p.splits > 1) return x3_rs_ok(N) ? x3_rs_rows(N) : 0;
    return p.tile == 2 ? 64 : 128;
}

// Same contract as regtr_gemm_f32 with B given as the planes written by regtr_gemm_split_weights(W, .., N, K, ..).
// stat_partial (optional, needs regtr_gemm_x3_stat_tile_rows(M,N,K) = R > 0): (ceil(M / R) + n_stat_seg) * N double2 that
// receive per-(row tile, cloud) column sums of C for regtr_instnorm_finalize_tiles; stat_seg_off [n_stat_seg + 1] are the
// cloud offsets of C's rows.
int regtr_gemm_x3(const float* A, int lda, const void* planes, float* C, int ldc, int M, int N, int K,
                  const float* bias, const float* row_div, const float* residual, int ldr, int act,
                  const float* a_stats, const int* a_seg_off, int n_seg, float a_slope, void* ws, size_t ws_bytes,
                  double* stat_partial, const int* stat_seg_off, int n_stat_seg, int n_planes, const void* tile_info, int* status, void* stream)
{
    if (!A || !planes || !C || M < 0 || lda < K || ldc < N || !regtr_gemm_x3_supported(M, N, K)) return RG_ERR_ARG;
    if (tile_info && a_stats && stat_partial && (a_seg_off != stat_seg_off || n_seg != n_stat_seg)) return RG_ERR_ARG;
    // n_planes: 1 | 2 | 3 bf16 planes (regtr_gemm_split_weights); 4 = the f16 pair of regtr_gemm_split_weights_f16 (three MFMA terms at
    // float32-grade accuracy; row-strip kernel only: regtr_gemm_x3_f16_supported)
    if (n_planes < 1 || n_planes > 4 || ((n_planes == 1 || n_planes == 2) && (a_stats || stat_partial))) return RG_ERR_ARG;
    // (the f16 plan may promote a 64-row plan to 128-row strips; never when a per-tile cloud table or folded statistics ride along --
    //  tile_info / stat_partial slots are laid out at regtr_gemm_x3_tile_rows' height)
    const bool f16_keep_rows = stat_partial != nullptr || a_stats != nullptr || tile_info != nullptr;
    if (n_planes == 4 && !regtr_gemm_x3_f16_supported(M, N, K, f16_keep_rows)) return RG_ERR_ARG;
    if (N == 32 && a_stats) return RG_ERR_ARG;               // the thin case exists on the row-strip kernel only
    if ((lda % 4) || ((uintptr_t)A % 16) || ((uintptr_t)planes % 16)) return RG_ERR_ARG;
    if (a_stats && (!a_seg_off || n_seg < 1 || ((uintptr_t)a_stats % 16))) return RG_ERR_ARG;
    if (M == 0) return RG_OK;
    X3Plan p = x3_plan(M, N, K);
    if (n_planes == 4) x3_plan_f16(M, N, K, f16_keep_rows, p);
    static const int f16_cw4 = X3_DEV_ENV("REGTR_F16_CW4", 1);   // development: A/B runs
    if (n_planes == 4 && p.tile == 0 && !f16_cw4) p.tile = 1;
    // (tiled kernel, f16 pair: the 8-wave 128 x 128 tile's two accumulator sets do not fit its 128-register budget -- 128 x 64 instead;
    //  same 128-row statistics slots)
    if (n_planes == 4 && p.tile == 0 && (!p.strip || a_stats || K % XBK || p.k_chunk % XBK)) p.tile = 1;
    if (p.splits > 1 && (!ws || ws_bytes < (size_t)p.splits * M * N * sizeof(float))) return RG_ERR_WORKSPACE;
    if (stat_partial && ((p.splits > 1 && !x3_rs_ok(N)) || !stat_seg_off || n_stat_seg < 1 || ((uintptr_t)stat_partial % 16))) return RG_ERR_ARG;
    // split-K: the product kernel writes raw partial products; the statistics (if asked for) come from the reduction kernel
    double* const stat_all = stat_partial;
    if (p.splits > 1) stat_partial = nullptr;
    const int Npad = rg_cdiv(N, 128) * 128, Kp = rg_cdiv(K, XBK) * XBK;
    X3Args g{A, (const uint16_t*)planes, C, bias, row_div, residual, (const float2*)a_stats, a_seg_off,
             p.splits > 1 ? (float*)ws : nullptr, (double2*)stat_partial, stat_seg_off, (const int4*)tile_info, (size_t)Npad * Kp,
             M, N, K, Kp, lda, ldc, ldr, act, n_seg, p.k_chunk, n_stat_seg, a_slope, status};
    hipStream_t st = (hipStream_t)stream;
    const int bm = p.tile == 2 ? 64 : 128, bn = p.tile == 0 ? 128 : 64;
    dim3 grid(rg_cdiv(rg_cdiv(M, bm), 8) * 8 * rg_cdiv(N, bn), 1, p.splits);      // see the XCD-aware tile map in the kernel
#define X3_LAUNCH(MW_, NW_, WM_, WN_) do { \
        if (n_planes == 4) { if constexpr (NW_ == 2) { \
            if (a_stats) { if (stat_partial) k_gemm_x3<MW_, NW_, WM_, WN_, true, true, 2, 1><<<grid, 64 * MW_ * NW_, 0, st>>>(g); \
                           else k_gemm_x3<MW_, NW_, WM_, WN_, true, false, 2, 1><<<grid, 64 * MW_ * NW_, 0, st>>>(g); } \
            else { if (stat_partial) k_gemm_x3<MW_, NW_, WM_, WN_, false, true, 2, 1><<<grid, 64 * MW_ * NW_, 0, st>>>(g); \
                   else k_gemm_x3<MW_, NW_, WM_, WN_, false, false, 2, 1><<<grid, 64 * MW_ * NW_, 0, st>>>(g); } } } \
        else if (n_planes == 1) k_gemm_x3<MW_, NW_, WM_, WN_, false, false, 1><<<grid, 64 * MW_ * NW_, 0, st>>>(g); \
        else if (n_planes == 2) k_gemm_x3<MW_, NW_, WM_, WN_, false, false, 2><<<grid, 64 * MW_ * NW_, 0, st>>>(g); \
        else if (a_stats) { if (stat_partial) k_gemm_x3<MW_, NW_, WM_, WN_, true, true><<<grid, 64 * MW_ * NW_, 0, st>>>(g); \
                       else k_gemm_x3<MW_, NW_, WM_, WN_, true, false><<<grid, 64 * MW_ * NW_, 0, st>>>(g); } \
        else { if (stat_partial) k_gemm_x3<MW_, NW_, WM_, WN_, false, true><<<grid, 64 * MW_ * NW_, 0, st>>>(g); \
               else k_gemm_x3<MW_, NW_, WM_, WN_, false, false><<<grid, 64 * MW_ * NW_, 0, st>>>(g); } } while (0)
#define X3D_LAUNCH(MW_, CW_, AR_) do { \
        if (n_planes == 4) { if (stat_partial) k_gemm_x3d<4, CW_, 2, true, 2, 1><<<grid, 256, 0, st>>>(g); \
                             else k_gemm_x3d<4, CW_, 2, false, 2, 1><<<grid, 256, 0, st>>>(g); } \
        else if (n_planes == 1) k_gemm_x3d<4, CW_, 2, false, 1><<<grid, 256, 0, st>>>(g); \
        else if (n_planes == 2) k_gemm_x3d<4, CW_, 2, false, 2><<<grid, 256, 0, st>>>(g); \
        else if (stat_partial) k_gemm_x3d<MW_, CW_, AR_, true><<<grid, 64 * MW_, 0, st>>>(g); \
        else k_gemm_x3d<MW_, CW_, AR_, false><<<grid, 64 * MW_, 0, st>>>(g); } while (0)
    const bool strip = p.strip && !a_stats && K % XBK == 0 && p.k_chunk % XBK == 0;
    static const int a_ring = X3_DEV_ENV("REGTR_X3_ARING", 3);   // development: A/B runs
    static const int deep_pipe = X3_DEV_ENV("REGTR_X3_DEEP", 1);   // development: A/B runs
    // (measured and left out: 8 waves on 256 x 128 tiles, 144 KiB of LDS, one workgroup per CU -- k_gemm_x3d<8, 4, 3> -- halves the
    // weight traffic per MFMA and is no faster: 498 vs 483 us on the level-2 contraction, 555 vs 478 at level 3 where 296 tiles
    // quantise badly over 256 CUs)
    static const int il = X3_DEV_ENV("REGTR_X3_IL", 1);   // development: A/B runs
    // Interleaved schedule (A two tiles ahead in TWO slots, every DMA instruction issued between MFMAs) for the 128 x 128 tile: measured
    // on the 18 RegTR shapes (gpurun_out/r03_z5) sum 4716 -> 4670 us, level-3 contraction 472 -> 458 us; on the 128 x 64 tile it loses
    // on the strided contractions (153 -> 165, 145 -> 151 us) and is not instantiated.  Hiding the ~1250-cycle DMA-issue phase bought
    // 1 %, not the 30 % its share of a tile suggested: with two workgroups per CU (regtr_gemm_x3_strip_occupancy) that phase already
    // overlapped the other workgroup's MFMAs.
    if (strip && il && p.tile == 0) X3D_LAUNCH(4, 4, 4);
    else if (strip && p.tile == 0) X3D_LAUNCH(4, 4, 2);              // 128 x 128: 4 waves of 32 rows x 128 columns, both operands by LDS-DMA
    else if (strip && a_ring == 3) X3D_LAUNCH(4, 2, 3);              // 128 x 64, A rows two tiles ahead
    else if (strip) X3D_LAUNCH(4, 2, 2);                             // 128 x 64
    // small problems on 64 x 64 tiles (at most two workgroups per CU, no folded operand): the three-deep operand pipeline (k_gemm_x3q)
    else if (deep_pipe && p.tile == 2 && !a_stats && (n_planes == 3 || n_planes == 4) && K % XBK == 0 && p.k_chunk % XBK == 0 &&
             (long long)rg_cdiv(M, 64) * (N / 64) * p.splits <= 512) {
        if (n_planes == 4) { if (stat_partial) k_gemm_x3q<true, 2, 1><<<grid, 256, 0, st>>>(g); else k_gemm_x3q<false, 2, 1><<<grid, 256, 0, st>>>(g); }
        else { if (stat_partial) k_gemm_x3q<true, 3, 0><<<grid, 256, 0, st>>>(g); else k_gemm_x3q<false, 3, 0><<<grid, 256, 0, st>>>(g); }
    }
    else if (p.tile == 0) X3_LAUNCH(2, 4, 2, 1);     // 128 x 128, 8 waves of 64 x 32
    else if (p.tile == 1) X3_LAUNCH(2, 2, 2, 1);     // 128 x 64, 4 waves of 64 x 32
    else X3_LAUNCH(2, 2, 1, 1);                      // 64 x 64, 4 waves of 32 x 32
#undef X3_LAUNCH
#undef X3D_LAUNCH
    if (p.splits > 1 && stat_all) {
        X3Args gr = g;
        gr.stat_partial = (double2*)stat_all;
        const int vec_ok = (ldc % 4 == 0 && (uintptr_t)C % 16 == 0) ? 1 : 0;
        k_x3_splitk_reduce_stats<<<rg_cdiv(M, x3_rs_rows(N)), 256, 0, st>>>(gr, p.splits, vec_ok, x3_rs_rows(N));
    } else if (p.splits > 1) {
        k_x3_splitk_reduce<<<rg_cdiv((long long)M * N, 256), 256, 0, st>>>(g, p.splits);
    }
    RG_RETURN_IF_LAUNCH_FAILED();
    return RG_OK;
}

}  // extern "C"
