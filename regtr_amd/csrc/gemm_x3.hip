// C[M,N] = epilogue( A[M,K] * W[K,N] )  at float32 accuracy on the CDNA4 *bf16* matrix cores.
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
};

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

// NP = bf16 planes per operand: 3 = float32-grade (six MFMA terms, the default); 2 = the three leading terms a0 w0 + a0 w1 +
// a1 w0 (relative error ~2^-16 per product); 1 = plain bf16 operands (cfg.compute_dtype 'bf16').
template <int MW, int NW, int WM, int WN, bool STATS, bool SOUT, int NP = 3>      // MW x NW waves, wave tile (32 WM) x (32 WN)
__global__ void __launch_bounds__(64 * MW * NW, (MW * NW >= 8) ? 4 : 2) k_gemm_x3(X3Args g)
{
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
            unsigned p0a, p1a, p2a, p0b, p1b, p2b;
            x3_split2(v.x, v.y, p0a, p1a, p2a);
            x3_split2(v.z, v.w, p0b, p1b, p2b);
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
#pragma unroll
    for (int i = 0; i < WM; i++)
#pragma unroll
        for (int j = 0; j < WN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

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
            if (NP == 3) { X3_TERM(2, 0) X3_TERM(1, 1) X3_TERM(0, 2) }
            if (NP >= 2) { X3_TERM((NP >= 2 ? 1 : 0), 0) X3_TERM(0, (NP >= 2 ? 1 : 0)) }
            X3_TERM(0, 0)
#undef X3_TERM
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
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (row >= g.M) continue;
                float v = acc[i][j][r];
                if (g.row_div) v = v / g.row_div[row];
                v += bv;
                if (g.act == 1) v = fmaxf(v, 0.f);
                if (g.residual) v += g.residual[(size_t)row * g.ldr + col];
                g.C[(size_t)row * g.ldc + col] = v;
                if (SOUT) acc[i][j][r] = v;
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

// tile t of `rows` rows -> (first cloud, last cloud, first cloud's begin, first cloud's end); one thread per tile
__global__ void __launch_bounds__(256) k_tile_segments(const int* __restrict__ seg_off, int n_seg, int M, int rows, int4* __restrict__ out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if ((long long)t * rows >= M) return;
    const int r0 = t * rows, r1 = min(M, r0 + rows) - 1;
    const int lo = rg_find_segment(seg_off, n_seg, r0), hi = rg_find_segment(seg_off, n_seg, r1);
    out[t] = make_int4(lo, hi, seg_off[lo], seg_off[lo + 1]);
}

struct X3Plan { int tile, splits, k_chunk; };      // tile: 0 = 128 x 128, 1 = 128 x 64, 2 = 64 x 64

// tile shape and K split for a problem (host policy)
X3Plan x3_plan(int M, int N, int K)
{
    X3Plan p{2, 1, K};
    static const int forced = (getenv("REGTR_X3_TILE") && *getenv("REGTR_X3_TILE")) ? atoi(getenv("REGTR_X3_TILE")) : -1;   // development: tile A/B runs
    if (forced >= 0 && forced <= 2 && (forced != 0 || N % 128 == 0)) { p.tile = forced; return p; }
    auto tiles = [&](int bm, int bn) { return (long long)rg_cdiv(M, bm) * (N / bn); };
    // Measured on MI355X (REGTR_X3_TILE = 0 / 1 / 2 sweeps over RegTR's shapes at 16 and 64 pairs per forward): with a few
    // hundred tiles the 64 x 64 tile at four workgroups per CU wins (occupancy hides the per-k-tile latency); from about
    // six tiles per CU on, the larger tiles' lower operand traffic per flop pays: 128 x 64 first for deep K, then the
    // 8-wave 128 x 128 tile.
    p.tile = 2;
    if (tiles(128, 64) >= 512 && (K >= 960 || tiles(128, 64) >= 2048)) p.tile = 1;
    if (N % 128 == 0 && tiles(128, 128) >= 1536) p.tile = 0;
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
    return p;
}

}  // namespace

extern "C" {

// 1 when regtr_gemm_x3 accepts the shape (otherwise use regtr_gemm_f32)
int regtr_gemm_x3_supported(int M, int N, int K) { return (N >= 64 && N % 64 == 0 && K >= 16 && K % 4 == 0 && M >= 0) ? 1 : 0; }

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
int regtr_gemm_x3_preferred(int M, int N, int K) { return (regtr_gemm_x3_supported(M, N, K) && K >= 32) ? 1 : 0; }

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

size_t regtr_gemm_x3_ws_bytes(int M, int N, int K)
{
    if (!regtr_gemm_x3_supported(M, N, K)) return 0;
    const X3Plan p = x3_plan(M, N, K);
    return p.splits > 1 ? (size_t)p.splits * M * N * sizeof(float) : 0;
}

// Same contract as regtr_gemm_f32 with B given as the planes written by regtr_gemm_split_weights(W, .., N, K, ..).
// rows per statistics tile when regtr_gemm_x3 can emit InstanceNorm partial sums for this shape in its epilogue
// (stat_partial), 0 when it cannot (split-K shapes): the caller then runs regtr_instnorm_stats on C instead.
int regtr_gemm_x3_stat_tile_rows(int M, int N, int K)
{
    if (!regtr_gemm_x3_supported(M, N, K) || M < 1) return 0;
    const X3Plan p = x3_plan(M, N, K);
    if (p.splits > 1) return 0;
    return p.tile == 2 ? 64 : 128;
}

// Same contract as regtr_gemm_f32 with B given as the planes written by regtr_gemm_split_weights(W, .., N, K, ..).
// stat_partial (optional, needs regtr_gemm_x3_stat_tile_rows(M,N,K) = R > 0): (ceil(M / R) + n_stat_seg) * N double2 that
// receive per-(row tile, cloud) column sums of C for regtr_instnorm_finalize_tiles; stat_seg_off [n_stat_seg + 1] are the
// cloud offsets of C's rows.
int regtr_gemm_x3(const float* A, int lda, const void* planes, float* C, int ldc, int M, int N, int K,
                  const float* bias, const float* row_div, const float* residual, int ldr, int act,
                  const float* a_stats, const int* a_seg_off, int n_seg, float a_slope, void* ws, size_t ws_bytes,
                  double* stat_partial, const int* stat_seg_off, int n_stat_seg, int n_planes, const void* tile_info, void* stream)
{
    if (!A || !planes || !C || M < 0 || lda < K || ldc < N || !regtr_gemm_x3_supported(M, N, K)) return RG_ERR_ARG;
    if (tile_info && a_stats && stat_partial && (a_seg_off != stat_seg_off || n_seg != n_stat_seg)) return RG_ERR_ARG;
    if (n_planes < 1 || n_planes > 3 || (n_planes != 3 && (a_stats || stat_partial))) return RG_ERR_ARG;
    if ((lda % 4) || ((uintptr_t)A % 16) || ((uintptr_t)planes % 16)) return RG_ERR_ARG;
    if (a_stats && (!a_seg_off || n_seg < 1 || ((uintptr_t)a_stats % 16))) return RG_ERR_ARG;
    if (M == 0) return RG_OK;
    const X3Plan p = x3_plan(M, N, K);
    if (p.splits > 1 && (!ws || ws_bytes < (size_t)p.splits * M * N * sizeof(float))) return RG_ERR_WORKSPACE;
    if (stat_partial && (p.splits > 1 || !stat_seg_off || n_stat_seg < 1 || ((uintptr_t)stat_partial % 16))) return RG_ERR_ARG;
    const int Npad = rg_cdiv(N, 128) * 128, Kp = rg_cdiv(K, XBK) * XBK;
    X3Args g{A, (const uint16_t*)planes, C, bias, row_div, residual, (const float2*)a_stats, a_seg_off,
             p.splits > 1 ? (float*)ws : nullptr, (double2*)stat_partial, stat_seg_off, (const int4*)tile_info, (size_t)Npad * Kp,
             M, N, K, Kp, lda, ldc, ldr, act, n_seg, p.k_chunk, n_stat_seg, a_slope};
    hipStream_t st = (hipStream_t)stream;
    const int bm = p.tile == 2 ? 64 : 128, bn = p.tile == 0 ? 128 : 64;
    dim3 grid(rg_cdiv(rg_cdiv(M, bm), 8) * 8 * (N / bn), 1, p.splits);      // see the XCD-aware tile map in the kernel
#define X3_LAUNCH(MW_, NW_, WM_, WN_) do { \
        if (n_planes == 1) k_gemm_x3<MW_, NW_, WM_, WN_, false, false, 1><<<grid, 64 * MW_ * NW_, 0, st>>>(g); \
        else if (n_planes == 2) k_gemm_x3<MW_, NW_, WM_, WN_, false, false, 2><<<grid, 64 * MW_ * NW_, 0, st>>>(g); \
        else if (a_stats) { if (stat_partial) k_gemm_x3<MW_, NW_, WM_, WN_, true, true><<<grid, 64 * MW_ * NW_, 0, st>>>(g); \
                       else k_gemm_x3<MW_, NW_, WM_, WN_, true, false><<<grid, 64 * MW_ * NW_, 0, st>>>(g); } \
        else { if (stat_partial) k_gemm_x3<MW_, NW_, WM_, WN_, false, true><<<grid, 64 * MW_ * NW_, 0, st>>>(g); \
               else k_gemm_x3<MW_, NW_, WM_, WN_, false, false><<<grid, 64 * MW_ * NW_, 0, st>>>(g); } } while (0)
    if (p.tile == 0) X3_LAUNCH(2, 4, 2, 1);          // 128 x 128, 8 waves of 64 x 32
    else if (p.tile == 1) X3_LAUNCH(2, 2, 2, 1);     // 128 x 64, 4 waves of 64 x 32
    else X3_LAUNCH(2, 2, 1, 1);                      // 64 x 64, 4 waves of 32 x 32
#undef X3_LAUNCH
    if (p.splits > 1) k_x3_splitk_reduce<<<rg_cdiv((long long)M * N, 256), 256, 0, st>>>(g, p.splits);
    RG_RETURN_IF_LAUNCH_FAILED();
    return RG_OK;
}

}  // extern "C"
