// KPConv preprocessing for gfx950: voxel-grid barycentre subsampling and batched fixed-radius
// neighbour search.  Replaces the reference's CPU ops
//   batch_grid_subsampling      cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp:109-211
//   batch_nanoflann_neighbors   cpp_wrappers/cpp_neighbors/neighbors/neighbors.cpp:211-332
// (paths relative to /root/reference/src/models/backbone_kpconv/).
//
// Design (not a translation -- the reference is a hash-map loop and a KD-tree):
//   * one spatial hash per call, open addressing over int32 "representative point" slots, allocated for
//     1.5x the point capacity and sized on the device for the live count.  HBM is plentiful (288 GB) so
//     every per-point intermediate is kept.
//   * all sizes that depend on the data (points per level) stay ON THE DEVICE: kernels are launched
//     over the caller's capacity and read the live counts from the segment-offset arrays, so a whole
//     pyramid is enqueued without a single host round trip.
//   * grid subsample: voxel membership lists are rebuilt in ascending input index so that the float32
//     barycentre sums are bit-identical to the reference's sequential accumulation.
//   * radius search: one 64-lane wavefront per query; the 27 candidate cells are looked up by 27 lanes
//     in parallel, candidates are tested 64 at a time, survivors are compacted with ballot/popcount into
//     an LDS list of (d2 bits << 32 | index) keys and ordered by rank counting -> ascending (d2, index).
//   Distance and voxel arithmetic use explicit round-to-nearest intrinsics (no FMA contraction) so the
//   float32 results match the reference's SSE2 arithmetic bit for bit.
#include "common.h"
#include "ref_umap.h"
#include <limits.h>

namespace {

// ------------------------------------------------------------------------------------------------
// exclusive scan over uint64 (three launches; n lives on the device)
// ------------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 4;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ uint64_t block_exclusive_scan(uint64_t v, uint64_t* total, uint64_t* sh /*[SCAN_THREADS/64 + 1]*/)
{
    // wave inclusive scan
    const int lane = rg_lane(), wave = threadIdx.x >> 6;
    uint64_t inc = v;
#pragma unroll
    for (int o = 1; o < RG_WAVE; o <<= 1) {
        uint64_t t = __shfl_up(inc, o, RG_WAVE);
        if (lane >= o) inc += t;
    }
    if (lane == RG_WAVE - 1) sh[wave] = inc;
    __syncthreads();
    uint64_t wave_off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / RG_WAVE; w++) {
        uint64_t s = sh[w];
        if (w < wave) wave_off += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return wave_off + inc - v;
}

__global__ void __launch_bounds__(SCAN_THREADS) k_scan_reduce(const uint64_t* __restrict__ in, const int* __restrict__ n_ptr,
                                                             uint64_t* __restrict__ bsum)
{
    __shared__ uint64_t sh[SCAN_THREADS / RG_WAVE + 1];
    const int n = *n_ptr;
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++)
        if (base + k < n) s += in[base + k];
    uint64_t tot;
    block_exclusive_scan(s, &tot, sh);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// single block: exclusive scan of the block sums in place
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_bsums(uint64_t* __restrict__ bsum, int nblocks)
{
    __shared__ uint64_t sh[SCAN_THREADS / RG_WAVE + 1];
    uint64_t carry = 0;
    for (int base = 0; base < nblocks; base += SCAN_THREADS) {
        const int i = base + threadIdx.x;
        uint64_t v = i < nblocks ? bsum[i] : 0, tot;
        uint64_t ex = block_exclusive_scan(v, &tot, sh);
        if (i < nblocks) bsum[i] = carry + ex;
        carry += tot;
    }
}

__global__ void __launch_bounds__(SCAN_THREADS) k_scan_apply(const uint64_t* __restrict__ in, const int* __restrict__ n_ptr,
                                                            const uint64_t* __restrict__ bsum, uint64_t* __restrict__ out)
{
    __shared__ uint64_t sh[SCAN_THREADS / RG_WAVE + 1];
    const int n = *n_ptr;
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint64_t v[SCAN_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        v[k] = base + k < n ? in[base + k] : 0;
        s += v[k];
    }
    uint64_t tot;
    uint64_t ex = block_exclusive_scan(s, &tot, sh) + bsum[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (base + k < n) out[base + k] = ex;
        ex += v[k];
    }
}

// The same scan in ONE launch for small inputs (at most SCAN_CHAIN_TILES tiles = 65536 items: a pair or two per forward, where the
// pyramid is ~80 launches of 2-30 us paced by the host and every launch removed is ~3 us of host and ~4 us of GPU time): chained scan
// with decoupled look-back.  state[0] = ticket counter, state[1 + t] = descriptor of tile t: 2 flag bits (0 not ready, 1 = the tile's own
// sum, 2 = its inclusive prefix) over a 62-bit value; the caller's clear kernel (k_clear_tables, earlier in the stream) zeroes
// state[0 .. nb].  Tile ids come from the ticket, so every predecessor of a running tile has started: no deadlock whatever the dispatch
// order.  Wave 0 reads ALL predecessors of its tile at once (<= 63: one memory round trip per attempt), takes the nearest published
// prefix and adds the sums in front of it.  Integer addition: the result is the three-launch scan's, bit for bit.
constexpr int SCAN_CHAIN_TILES = 64;
constexpr uint64_t SC_SUM = 1ull << 62, SC_PREFIX = 2ull << 62, SC_VALUE = (1ull << 62) - 1;

__global__ void __launch_bounds__(SCAN_THREADS) k_scan_chained(const uint64_t* __restrict__ in, const int* __restrict__ n_ptr,
                                                              uint64_t* __restrict__ state, uint64_t* __restrict__ out)
{
    __shared__ uint64_t sh[SCAN_THREADS / RG_WAVE + 1];
    __shared__ int s_tile;
    __shared__ uint64_t s_prefix;
    if (threadIdx.x == 0) s_tile = (int)atomicAdd((unsigned*)state, 1u);
    __syncthreads();
    const int tile = s_tile, n = *n_ptr;
    const int base = tile * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint64_t v[SCAN_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        v[k] = base + k < n ? in[base + k] : 0;
        s += v[k];
    }
    uint64_t tot;
    uint64_t ex = block_exclusive_scan(s, &tot, sh);
    if (threadIdx.x < RG_WAVE) {                                      // wave 0: publish, look back
        const int lane = threadIdx.x;
        uint64_t* desc = state + 1;
        if (tile == 0) {
            if (lane == 0) {
                __hip_atomic_store(&desc[0], SC_PREFIX | tot, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                s_prefix = 0;
            }
        } else {
            if (lane == 0) __hip_atomic_store(&desc[tile], SC_SUM | tot, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            // lane l looks at tile - 1 - l; lanes beyond tile 0 stand for "the prefix in front of tile 0" = a published 0
            uint64_t d;
            int fp;
            for (;;) {
                d = lane < tile ? __hip_atomic_load(&desc[tile - 1 - lane], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) : SC_PREFIX;
                const unsigned long long is_p = __ballot((d >> 62) == 2), not_ready = __ballot((d >> 62) == 0);
                fp = __ffsll((long long)is_p) - 1;                      // nearest predecessor holding an inclusive prefix (always >= 0)
                if ((not_ready & ((2ull << fp) - 1ull)) == 0) break;    // everything between it and this tile has published its sum
                __builtin_amdgcn_s_sleep(1);
            }
            uint64_t run = lane <= fp ? (d & SC_VALUE) : 0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) run += __shfl_xor(run, o, RG_WAVE);
            if (lane == 0) {
                __hip_atomic_store(&desc[tile], SC_PREFIX | (run + tot), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                s_prefix = run;
            }
        }
    }
    __syncthreads();
    ex += s_prefix;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (base + k < n) out[base + k] = ex;
        ex += v[k];
    }
}

// bsum: rg_cdiv(n_cap, SCAN_TILE) + 1 entries; when that is at most SCAN_CHAIN_TILES + 1 the caller's clear kernel must have zeroed them
int scan_u64(const uint64_t* in, const int* n_ptr, int n_cap, uint64_t* bsum, uint64_t* out, hipStream_t st)
{
    const int nb = rg_cdiv(n_cap, SCAN_TILE);
    if (nb <= SCAN_CHAIN_TILES) {
        k_scan_chained<<<nb, SCAN_THREADS, 0, st>>>(in, n_ptr, bsum, out);
        return RG_OK;
    }
    k_scan_reduce<<<nb, SCAN_THREADS, 0, st>>>(in, n_ptr, bsum);
    k_scan_bsums<<<1, SCAN_THREADS, 0, st>>>(bsum, nb);
    k_scan_apply<<<nb, SCAN_THREADS, 0, st>>>(in, n_ptr, bsum, out);
    return RG_OK;
}

// ------------------------------------------------------------------------------------------------
// spatial hash: slots hold the index of the first point that claimed them ("representative")
// ------------------------------------------------------------------------------------------------
// Hash tables are ALLOCATED for the level-0 capacity (live counts exist only on the device) but SIZED, on the device, for
// the live count of their level: the power of two >= 1.5 n (>= 64).  Clears, key passes, scans and the packed-slot pass
// then touch 1.5-3 n slots instead of 3 x capacity, and probes of a deep level stay in a table 1/64 the size.
__device__ __forceinline__ unsigned rg_live_table(int n)
{
    unsigned want = (unsigned)n + ((unsigned)n >> 1);
    if (want < 64u) want = 64u;
    return 1u << (32 - __clz((int)(want - 1u)));
}
static inline unsigned rg_table_capacity(int n_cap)       // host twin: upper bound of rg_live_table over n <= n_cap
{
    unsigned want = (unsigned)n_cap + ((unsigned)n_cap >> 1);
    if (want < 64u) want = 64u;
    return rg_next_pow2(want);
}

// (+ the small-scan state of the call's scan_u64 and, for the grid subsample, the per-cloud boxes: one launch instead of three)
__global__ void __launch_bounds__(256) k_clear_tables(const int* __restrict__ n_ptr, int* __restrict__ rep, int* __restrict__ cnt,
                                                      int* __restrict__ fill, int* __restrict__ first, uint64_t* __restrict__ scan_state,
                                                      int n_state, int* __restrict__ bbox, int n_bbox)
{
    const unsigned T = rg_live_table(*n_ptr);
    const unsigned h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h < (unsigned)n_state) scan_state[h] = 0;
    if (h < (unsigned)n_bbox) bbox[h] = (h % 6) < 3 ? INT_MAX : INT_MIN;
    if (h >= T) return;
    rep[h] = -1; cnt[h] = 0; fill[h] = 0;
    if (first) first[h] = 0x7F7F7F7F;
}

#ifndef RG_RANK2
#define RG_RANK2 1            // development A/B (REGTR_VARIANT_FLAGS=-DRG_RANK2=0): for_each_ranked below
#endif
constexpr int CELL_BITS = 21;
constexpr int64_t CELL_BIAS = 1 << 20;

// Home slot of a key: the plain 64-bit mix, salted with the cloud.
// (Measured and dropped, round 6: BLOCK-LOCAL homes for the cell keys of the radius grid -- the 4 x 4 x 4 block of cells picks a 64-slot region,
//  the cell's position in its block the slot inside it -- so that the 27 probes of a neighbourhood fall into <= 8 regions of 2 KB instead of 27 random
//  lines of a table of up to 537 MB, a wave's slot run is one spatial block and the cell-sorted supports are spatially coherent.  Tables bit-identical;
//  192-pair pyramid 12.17 -> 18.46 ms: k_radius_query_self 1219 -> 2533 us per launch, k_radius_query 788 -> 1105, k_insert 267 -> 278
//  (profiles/r06_d_block_hash.md).  A planar surface fills CONTIGUOUS runs of 16 slots of its region; where two blocks share a region a probe walks such a
//  run, and a wave waits for the slowest of its 27 probing lanes -- most neighbourhoods then pay 5-10 dependent round trips instead of 1-2; and the
//  occupied slots bunch into 27 % of the regions, so the cell-centric kernel's slot runs are badly balanced.  The locality bought nothing even where
//  neither effect applies (k_insert): these kernels wait on dependent round trips and issue slots, not on HBM bytes.)
__device__ __forceinline__ unsigned rg_home_slot(uint64_t key, int cid, unsigned mask)
{
    return rg_hash64(key ^ ((uint64_t)(cid + 1) * 0x9E3779B97F4A7C15ULL)) & mask;
}

__device__ __forceinline__ int hash_insert(int* __restrict__ rep, unsigned mask, const uint64_t* __restrict__ pkey,
                                           const int* __restrict__ pcid, int i, uint64_t key, int cid)
{
    unsigned h = rg_home_slot(key, cid, mask);
    for (;;) {
        int r = __hip_atomic_load(&rep[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (r < 0) {
            r = atomicCAS(&rep[h], -1, i);
            if (r < 0) return (int)h;  // we claimed the slot
        }
        // pkey/pcid were written by a previous kernel -> visible
        if (pkey[r] == key && pcid[r] == cid) return (int)h;
        h = (h + 1) & mask;
    }
}

// ------------------------------------------------------------------------------------------------
// grid subsample
// ------------------------------------------------------------------------------------------------
// per-cloud min / max corner  (cloud.cpp:27-66).  Each wave walks BBOX_ITEMS strided points per lane and keeps a running
// box in registers while the cloud stays the same, so the global atomics are ~one set per wave per cloud.
constexpr int BBOX_ITEMS = 8;

__device__ __forceinline__ void bbox_flush(int cid, float (&mn)[3], float (&mx)[3], int* __restrict__ bbox)
{
    if (cid < 0) return;   // wave-uniform
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int a = 0; a < 3; a++) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], o, RG_WAVE));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o, RG_WAVE));
        }
    if (rg_lane() == 0) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            atomicMin(&bbox[cid * 6 + a], rg_f2ord(mn[a]));
            atomicMax(&bbox[cid * 6 + 3 + a], rg_f2ord(mx[a]));
        }
    }
}

__global__ void __launch_bounds__(256) k_bbox(const float* __restrict__ xyz, const int* __restrict__ seg_off, int n_clouds,
                                              int* __restrict__ pcid, int* __restrict__ bbox)
{
    const int n = seg_off[n_clouds];
    const int base = blockIdx.x * (256 * BBOX_ITEMS) + threadIdx.x;
    int cur = -1;   // cloud of the running box (wave-uniform)
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    // every point of the lane requested up front, from CLAMPED rows (round 5): under the `live` predicate each of the eight iterations
    // was load -> wait -> use, eight dependent memory round trips per lane -- 15-19 us on a 40 k-point level whose 18 workgroups
    // have nothing else to hide them behind (measured and NOT the fix: one point per lane in 8x the workgroups, 27 us -- the box
    // atomics, one set per wave, serialise at ~90 ns each)
    float pv[BBOX_ITEMS][3];
    int cids[BBOX_ITEMS];
#pragma unroll
    for (int j = 0; j < BBOX_ITEMS; j++) {
        const int ic = min(base + j * 256, n > 0 ? n - 1 : 0);
        pv[j][0] = xyz[3 * (size_t)ic]; pv[j][1] = xyz[3 * (size_t)ic + 1]; pv[j][2] = xyz[3 * (size_t)ic + 2];
    }
#pragma unroll
    for (int j = 0; j < BBOX_ITEMS; j++) cids[j] = base + j * 256 < n ? rg_find_segment(seg_off, n_clouds, base + j * 256) : -1;
#pragma unroll
    for (int j = 0; j < BBOX_ITEMS; j++) {
        const int i = base + j * 256;
        const bool live = i < n;
        const int cid = cids[j];
        const float p[3] = {pv[j][0], pv[j][1], pv[j][2]};
        if (live) pcid[i] = cid;
        const int cid0 = __shfl(cid, 0, RG_WAVE);
        const bool uniform = __all(cid == cid0) && cid0 >= 0;
        if (uniform) {
            if (cid0 != cur) {
                bbox_flush(cur, mn, mx, bbox);
                cur = cid0;
#pragma unroll
                for (int a = 0; a < 3; a++) { mn[a] = INFINITY; mx[a] = -INFINITY; }
            }
#pragma unroll
            for (int a = 0; a < 3; a++) { mn[a] = fminf(mn[a], p[a]); mx[a] = fmaxf(mx[a], p[a]); }
        } else if (live) {   // wave straddles a cloud boundary (or the tail): per-lane atomics
#pragma unroll
            for (int a = 0; a < 3; a++) {
                atomicMin(&bbox[cid * 6 + a], rg_f2ord(p[a]));
                atomicMax(&bbox[cid * 6 + 3 + a], rg_f2ord(p[a]));
            }
        }
    }
    bbox_flush(cur, mn, mx, bbox);
}

// voxel key of every point.  key_mode 0: exactly as grid_subsampling.cpp:25-31,53-56 computes it (float32, no contraction) -- the
// reference's CPU Preprocessor.  key_mode 1 / 2: the reference's PreprocessorGPU (kpconv.py:213-240: MinkowskiEngine quantisation of
// points / sampleDl, i.e. integer coordinates floor(p / dl) with NO origin shift): 1 = IEEE float32 division (torch on the CPU, numpy),
// 2 = p * (1 / dl) with the reciprocal rounded to float32 (what torch's CUDA division by a host scalar evaluates).  The three integer
// coordinates are packed 21 bits each (|floor(p / dl)| < 2^20: 52 km at dl = 5 cm; beyond that coordinates saturate).
__device__ __forceinline__ uint64_t rg_pack_coord(float f)
{
    const float c = fminf(fmaxf(floorf(f), -1048576.f), 1048575.f);
    return (uint64_t)((int)c + 1048576);
}

__global__ void __launch_bounds__(256) k_voxel_keys(const float* __restrict__ xyz, const int* __restrict__ seg_off, int n_clouds,
                                                    const int* __restrict__ pcid, const int* __restrict__ bbox, float dl, int key_mode,
                                                    uint64_t* __restrict__ pkey)
{
    const int n = seg_off[n_clouds];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int cid = pcid[i];
    const float inv = __fdiv_rn(1.0f, dl);
    float org[3], mx[3], p[3];
#pragma unroll
    for (int a = 0; a < 3; a++) p[a] = xyz[3 * (size_t)i + a];
    if (key_mode != 0) {
        uint64_t c[3];
#pragma unroll
        for (int a = 0; a < 3; a++) c[a] = rg_pack_coord(key_mode == 1 ? __fdiv_rn(p[a], dl) : __fmul_rn(p[a], inv));
        pkey[i] = (c[2] << 42) | (c[1] << 21) | c[0];
        return;
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
        org[a] = __fmul_rn(floorf(__fmul_rn(rg_ord2f(bbox[cid * 6 + a]), inv)), dl);
        mx[a] = rg_ord2f(bbox[cid * 6 + 3 + a]);
    }
    // (size_t)floor(float): negative values wrap exactly like the x86-64 cvttss2si path the reference takes
    const uint64_t NX = (uint64_t)(int64_t)floorf(__fdiv_rn(__fsub_rn(mx[0], org[0]), dl)) + 1;
    const uint64_t NY = (uint64_t)(int64_t)floorf(__fdiv_rn(__fsub_rn(mx[1], org[1]), dl)) + 1;
    const uint64_t iX = (uint64_t)(int64_t)floorf(__fdiv_rn(__fsub_rn(p[0], org[0]), dl));
    const uint64_t iY = (uint64_t)(int64_t)floorf(__fdiv_rn(__fsub_rn(p[1], org[1]), dl));
    const uint64_t iZ = (uint64_t)(int64_t)floorf(__fdiv_rn(__fsub_rn(p[2], org[2]), dl));
    pkey[i] = iX + NX * iY + NX * NY * iZ;
}

__global__ void __launch_bounds__(256) k_insert(const int* __restrict__ n_ptr, const uint64_t* __restrict__ pkey,
                                                const int* __restrict__ pcid, int* __restrict__ rep,
                                                int* __restrict__ slot_of, int* __restrict__ first, int* __restrict__ cnt)
{
    const int n = *n_ptr;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = hash_insert(rep, rg_live_table(n) - 1u, pkey, pcid, i, pkey[i], pcid[i]);
    slot_of[i] = s;
    if (first) atomicMin(&first[s], i);
    atomicAdd(&cnt[s], 1);
}

// scan input: leaders (lowest input index of their voxel) carry (1 << 32 | member count)
__global__ void __launch_bounds__(256) k_leader_flags(const int* __restrict__ n_ptr, const int* __restrict__ slot_of,
                                                      const int* __restrict__ first, const int* __restrict__ cnt,
                                                      uint64_t* __restrict__ scan_in)
{
    const int n = *n_ptr;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = slot_of[i];
    scan_in[i] = first[s] == i ? ((1ULL << 32) | (uint64_t)(unsigned)cnt[s]) : 0ULL;
}

__global__ void __launch_bounds__(256) k_scatter_members(const int* __restrict__ n_ptr, const int* __restrict__ slot_of,
                                                         const int* __restrict__ first, const uint64_t* __restrict__ scan_out,
                                                         int* __restrict__ fill, int* __restrict__ members)
{
    const int n = *n_ptr;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = slot_of[i];
    const unsigned start = (unsigned)(scan_out[first[s]] & 0xFFFFFFFFu);
    const int p = atomicAdd(&fill[s], 1);
    members[start + p] = i;
}

// one thread per voxel leader: order the member list by input index, accumulate in float32 in that order,
// scale by (float)(1.0 / count)   (grid_subsampling.h:74-79, grid_subsampling.cpp:87)
__global__ void __launch_bounds__(256) k_barycentres(const float* __restrict__ xyz, const int* __restrict__ n_ptr,
                                                     const int* __restrict__ slot_of, const int* __restrict__ first,
                                                     const int* __restrict__ cnt, const uint64_t* __restrict__ scan_out,
                                                     int* __restrict__ members, const int* __restrict__ dest, int out_cap,
                                                     float* __restrict__ out_xyz)
{
    const int n = *n_ptr;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = slot_of[i];
    if (first[s] != i) return;
    const uint64_t so = scan_out[i];
    const unsigned start = (unsigned)(so & 0xFFFFFFFFu);
    const unsigned j = dest ? (unsigned)dest[so >> 32] : (unsigned)(so >> 32);     // reference row order (parity mode)
    if (j >= (unsigned)out_cap) return;             // beyond the caller's output capacity: dropped (out_seg_off saturates, see k_out_offsets)
    const int c = cnt[s];
    int* m = members + start;
    for (int a = 1; a < c; a++) {  // insertion sort (lists are a handful of points)
        const int v = m[a];
        int b = a - 1;
        while (b >= 0 && m[b] > v) { m[b + 1] = m[b]; b--; }
        m[b + 1] = v;
    }
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int a = 0; a < c; a++) {
        const size_t p = 3 * (size_t)m[a];
        sx = __fadd_rn(sx, xyz[p]); sy = __fadd_rn(sy, xyz[p + 1]); sz = __fadd_rn(sz, xyz[p + 2]);
    }
    const float w = (float)(1.0 / (double)c);
    out_xyz[3 * (size_t)j] = __fmul_rn(sx, w);
    out_xyz[3 * (size_t)j + 1] = __fmul_rn(sy, w);
    out_xyz[3 * (size_t)j + 2] = __fmul_rn(sz, w);
}

// ---- reference row order (parity mode, ref_umap.h): the voxel keys in first-appearance order, then one thread per cloud
// replays the libstdc++ unordered_map the reference iterates (grid_subsampling.cpp:48,58-59,85) ----
__global__ void __launch_bounds__(256) k_rank_keys(const int* __restrict__ n_ptr, const int* __restrict__ slot_of,
                                                   const int* __restrict__ first, const uint64_t* __restrict__ scan_out,
                                                   const uint64_t* __restrict__ pkey, uint64_t* __restrict__ vkey)
{
    const int n = *n_ptr;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (first[slot_of[i]] == i) vkey[scan_out[i] >> 32] = pkey[i];
}

__global__ void k_umap_order(const uint64_t* __restrict__ vkey, const int* __restrict__ out_seg_off, int n_clouds,
                             int* __restrict__ unext, int* __restrict__ ubefore, int* __restrict__ uorder,
                             int* __restrict__ dest)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_clouds) return;
    const int base = out_seg_off[c], m = out_seg_off[c + 1] - base;
    if (m <= 0) return;
    int* order = uorder + base;
    rg_umap_iteration_order(vkey + base, m, unext + base + c, ubefore + rg_umap_before_offset(base, c), order);
    for (int p = 0; p < m; p++) dest[base + order[p]] = base + p;
}

__global__ void k_out_offsets(const int* __restrict__ seg_off, int n_clouds, const uint64_t* __restrict__ scan_in,
                              const uint64_t* __restrict__ scan_out, int out_cap, int* __restrict__ out_seg_off)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > n_clouds) return;
    const int n = seg_off[n_clouds];
    const int i = seg_off[b];
    int v;
    if (i < n) v = (int)(scan_out[i] >> 32);
    else v = n > 0 ? (int)(scan_out[n - 1] >> 32) + (int)(scan_in[n - 1] >> 32) : 0;
    out_seg_off[b] = v < out_cap ? v : out_cap;      // saturates at the output capacity: a total equal to out_cap tells the caller the level is full
}

// ------------------------------------------------------------------------------------------------
// cell grid over support points + radius query
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void cell_of(float x, float y, float z, double inv_cs, int64_t& cx, int64_t& cy, int64_t& cz)
{
    cx = (int64_t)floor((double)x * inv_cs);
    cy = (int64_t)floor((double)y * inv_cs);
    cz = (int64_t)floor((double)z * inv_cs);
}
__device__ __forceinline__ uint64_t cell_key(int64_t cx, int64_t cy, int64_t cz)
{
    const uint64_t m = (1ULL << CELL_BITS) - 1;
    return ((uint64_t)(cx + CELL_BIAS) & m) | (((uint64_t)(cy + CELL_BIAS) & m) << CELL_BITS) |
           (((uint64_t)(cz + CELL_BIAS) & m) << (2 * CELL_BITS));
}

__global__ void __launch_bounds__(256) k_cell_keys(const float* __restrict__ xyz, const int* __restrict__ seg_off, int n_clouds,
                                                   double inv_cs, uint64_t* __restrict__ pkey, int* __restrict__ pcid)
{
    const int n = seg_off[n_clouds];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t cx, cy, cz;
    cell_of(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], inv_cs, cx, cy, cz);
    pkey[i] = cell_key(cx, cy, cz);
    pcid[i] = rg_find_segment(seg_off, n_clouds, i);
}

// slot -> (cell key, cloud) for lookups, and the scan input (member count per slot)
__global__ void __launch_bounds__(256) k_table_keys(const int* __restrict__ rep, const int* __restrict__ n_ptr,
                                                    const uint64_t* __restrict__ pkey,
                                                    const int* __restrict__ pcid, const int* __restrict__ cnt,
                                                    uint64_t* __restrict__ tkey, int* __restrict__ tcid,
                                                    uint64_t* __restrict__ scan_in, int* __restrict__ table_len)
{
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    const int T = (int)rg_live_table(*n_ptr);
    if (h == 0) *table_len = T;            // device-side length of the scan over the table
    if (h >= T) return;
    const int r = rep[h];
    if (r >= 0) { tkey[h] = pkey[r]; tcid[h] = pcid[r]; }
    scan_in[h] = (uint64_t)(unsigned)cnt[h];
}

// cell-sorted copy of the supports: (x, y, z, index-as-bits), 16-B records for single-instruction loads
__global__ void __launch_bounds__(256) k_scatter_cells(const float* __restrict__ xyz, const int* __restrict__ n_ptr,
                                                       const int* __restrict__ slot_of, const uint64_t* __restrict__ cell_start,
                                                       int* __restrict__ fill, float4* __restrict__ sorted)
{
    const int n = *n_ptr;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = slot_of[i];
    const unsigned pos = (unsigned)cell_start[s] + (unsigned)atomicAdd(&fill[s], 1);
    sorted[pos] = make_float4(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], __int_as_float(i));
}

// One probe = ONE memory round trip: everything a radius query needs from a hash slot, packed into 32 bytes by
// k_pack_slots once the cell starts are known (the separate rep / tkey / tcid / cnt / cell_start arrays would be three
// dependent loads per probe, and the query kernel is latency bound).
struct __align__(16) CellSlot {
    uint64_t key;
    int cid;
    int used;        // >= 0: occupied
    int cnt;
    unsigned start;
    int pad[2];
};

__global__ void __launch_bounds__(256) k_pack_slots(const int* __restrict__ rep, const int* __restrict__ n_ptr,
                                                    const uint64_t* __restrict__ tkey,
                                                    const int* __restrict__ tcid, const int* __restrict__ cnt,
                                                    const uint64_t* __restrict__ cell_start, CellSlot* __restrict__ slots)
{
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= (int)rg_live_table(*n_ptr)) return;
    CellSlot s;
    s.used = rep[h];
    const bool occ = s.used >= 0;
    s.key = occ ? tkey[h] : 0; s.cid = occ ? tcid[h] : -1; s.cnt = occ ? cnt[h] : 0; s.start = occ ? (unsigned)cell_start[h] : 0u;
    s.pad[0] = s.pad[1] = 0;
    slots[h] = s;
}

// lookup in the packed table: (count, start) of cell `key` of cloud `cid`, count 0 when the cell is empty
__device__ __forceinline__ void slot_find(const CellSlot* __restrict__ slots, unsigned mask, uint64_t key, int cid, int& cnt,
                                          int& start)
{
    unsigned h = rg_home_slot(key, cid, mask);
    cnt = 0; start = 0;
    for (;;) {
        const uint4 a = *(const uint4*)&slots[h];                 // key | cid | used
        const uint2 b = *(const uint2*)((const char*)&slots[h] + 16);   // cnt | start
        if ((int)a.w < 0) return;
        if ((((uint64_t)a.y << 32) | a.x) == key && (int)a.z == cid) { cnt = (int)b.x; start = (int)b.y; return; }
        h = (h + 1) & mask;
    }
}

struct GridView {
    const CellSlot* slots;
    const float4* sorted;
    double inv_cs;
};

constexpr int QUERY_WAVES = 4;  // queries in flight per 256-thread workgroup

// rank of every list entry among the n keys (keys are unique: the index is part of the key)
// list lives in LDS (16-byte aligned, room for n + 7 entries); entries e = lane, lane+64, ...
// The keys are read EIGHT per step as four independent 16-byte broadcast reads with one wait: a one-key-per-iteration loop
// (ds_read_b64, s_waitcnt lgkmcnt(0), compare) exposes a full LDS round trip per key -- ~100 cycles x ~30 keys per query was
// most of the radius search's time.  The tail is padded with +inf keys, which rank below nothing.
// Lists of <= 32 keys (a ball holds ~30 supports at 3DMatch densities: nine rows in ten) are ranked by TWO lanes per key (round 6): lane e
// counts the keys below its own among keys 0..15, lane e + 32 among keys 16..31 -- two 8-key steps instead of four, the halves joined by
// one v_permlane32_swap.  The compare-and-count steps are half of the vector instructions a query costs.
#define RG_LT(lo, hi) ((((uint64_t)(hi) << 32) | (lo)) < mine ? 1 : 0)
template <typename F>
__device__ __forceinline__ void for_each_ranked(uint64_t* list, int n, F&& f)
{
    const int lane = rg_lane();
    if (RG_RANK2 && n <= 32) {                       // wave-uniform
        if (lane >= n && lane < 32) list[lane] = ~0ULL;
        __builtin_amdgcn_wave_barrier();
        const uint64_t mine = list[lane & 31];
        const uint64_t* part = list + ((lane >> 5) << 4);
        int rank = 0;
#pragma unroll
        for (int j = 0; j < 16; j += 8) {
            const uint4 a = *(const uint4*)(part + j), b = *(const uint4*)(part + j + 2), c = *(const uint4*)(part + j + 4),
                        d = *(const uint4*)(part + j + 6);
            rank += RG_LT(a.x, a.y) + RG_LT(a.z, a.w) + RG_LT(b.x, b.y) + RG_LT(b.z, b.w) + RG_LT(c.x, c.y) + RG_LT(c.z, c.w) +
                    RG_LT(d.x, d.y) + RG_LT(d.z, d.w);
        }
        const auto r = __builtin_amdgcn_permlane32_swap((unsigned)rank, (unsigned)rank, false, false);
        rank = (int)(r[0] + r[1]);
        if (lane < n) f(rank, mine);
        return;
    }
    const int np = (n + 7) & ~7;
    if (lane < np - n) list[n + lane] = ~0ULL;
    __builtin_amdgcn_wave_barrier();
    for (int e0 = 0; e0 < n; e0 += RG_WAVE) {
        const int e = e0 + lane;
        const uint64_t mine = e < n ? list[e] : ~0ULL;
        int rank = 0;
        for (int j = 0; j < np; j += 8) {
            const uint4 a = *(const uint4*)(list + j), b = *(const uint4*)(list + j + 2), c = *(const uint4*)(list + j + 4),
                        d = *(const uint4*)(list + j + 6);
            rank += RG_LT(a.x, a.y) + RG_LT(a.z, a.w) + RG_LT(b.x, b.y) + RG_LT(b.z, b.w) + RG_LT(c.x, c.y) + RG_LT(c.z, c.w) +
                    RG_LT(d.x, d.y) + RG_LT(d.z, d.w);
        }
        if (e < n) f(rank, mine);
    }
}
#undef RG_LT

__global__ void __launch_bounds__(QUERY_WAVES * RG_WAVE)
k_radius_query(const float* __restrict__ q_xyz, const int* __restrict__ q_seg_off, const int* __restrict__ s_seg_off,
               int n_clouds, GridView g, float radius, int K, int cap, int by_index, int* __restrict__ out_idx,
               int* __restrict__ out_count, int* __restrict__ out_max_count)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = rg_lane();
    // per-wave LDS: list[cap] u64, then 27 run offsets + 27 run deltas (64 ints)
    uint64_t* list = (uint64_t*)smem + (size_t)wave * (cap + 32);

    const int nq = q_seg_off[n_clouds];
    const int ns = s_seg_off[n_clouds];
    const unsigned mask = rg_live_table(ns) - 1u;      // the table was built for ns supports
    // Grid-stride over the LIVE queries: the launch is sized for the level-0 capacity (live counts exist only on the
    // device), and a level with 1/64 of the rows must not pay for 63/64 empty workgroups.
    // (every wave takes a CONTIGUOUS run of queries, and every XCD -- own L2 -- a contiguous eighth of them: neighbouring queries
    //  read the same cell runs)
    const int per_wave = (nq + gridDim.x * QUERY_WAVES - 1) / (gridDim.x * QUERY_WAVES);
    const int q_first = (rg_xcd_block(blockIdx.x, gridDim.x) * QUERY_WAVES + wave) * per_wave;
    const int q_last = min(nq, q_first + per_wave);
    for (int q = q_first; q < q_last; q++) {   // wave-uniform
    const int cid = rg_find_segment_wave(q_seg_off, n_clouds, q);    // q is wave-uniform: one round trip, not log2(n) dependent loads
    const float qx = q_xyz[3 * (size_t)q], qy = q_xyz[3 * (size_t)q + 1], qz = q_xyz[3 * (size_t)q + 2];
    const float r2 = __fmul_rn(radius, radius);  // neighbors.cpp:226

    // 27 candidate cells, one per lane
    int64_t cx, cy, cz;
    cell_of(qx, qy, qz, g.inv_cs, cx, cy, cz);
    int my_cnt = 0, my_start = 0;
    if (lane < 27) {
        const int dx = lane % 3 - 1, dy = (lane / 3) % 3 - 1, dz = lane / 9 - 1;
        slot_find(g.slots, mask, cell_key(cx + dx, cy + dy, cz + dz), cid, my_cnt, my_start);
    }
    // exclusive prefix of the run lengths over lanes
    // (DPP row shifts: Hillis-Steele inside each 16-lane row, then row 0's total is added to row 1 -- the 27 runs live in
    //  lanes 0..26; five dependent ds_bpermute shuffles would cost five LDS-crossbar round trips)
    int inc = my_cnt;
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, true);     // row_shr:1, out-of-row reads 0
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, true);     // row_shr:2
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, true);     // row_shr:4
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xf, 0xf, true);     // row_shr:8
    inc += (lane >= 16 && lane < 32) ? __builtin_amdgcn_readlane(inc, 15) : 0;
    const int total = __builtin_amdgcn_readlane(inc, 26);
    // run table in LDS: run r covers candidates [runs[r], runs[r + 1]) and candidate t of it is support row t + runs[32 + r].
    // (Measured and dropped: the table in 54 SGPRs with 26 compare/selects per lane instead of the 5-step LDS binary
    //  search: 17 % slower; two queries per wave, one per 32-lane half, with per-half DPP scans / ballots: 17 % slower too,
    //  bit-identical tables in both cases.)
    int* runs = (int*)(list + cap);
    if (lane < 27) { runs[lane] = inc - my_cnt; runs[32 + lane] = my_start - (inc - my_cnt); }
    __builtin_amdgcn_wave_barrier();

    int n = 0;          // entries currently in the list
    int n_total = 0;    // all supports inside the ball (untruncated count)
    for (int t0 = 0; t0 < total; t0 += RG_WAVE) {
        const int t = t0 + lane;
        bool in = false;
        uint64_t key = 0;
        if (t < total) {
            int c = 0;  // last run with offset <= t
#pragma unroll
            for (int step = 16; step > 0; step >>= 1) { const int c2 = c + step; if (c2 < 27 && runs[c2] <= t) c = c2; }
            const int d = runs[32 + c];
            const float4 sp = g.sorted[t + d];
            const float dx = __fsub_rn(qx, sp.x), dy = __fsub_rn(qy, sp.y), dz = __fsub_rn(qz, sp.z);
            // nanoflann.hpp:432-440 : ((0 + dx*dx) + dy*dy) + dz*dz, strict '<' (nanoflann.hpp:249-251)
            float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            in = d2 < r2;
            key = (by_index ? 0ULL : ((uint64_t)__float_as_uint(d2) << 32)) | (uint32_t)__float_as_int(sp.w);   // order 1: by support index alone
        }
        const unsigned long long bal = __ballot(in);
        if (in) list[n + __popcll(bal & ((1ULL << lane) - 1ULL))] = key;
        const int add = __popcll(bal);
        n += add;
        n_total += add;
        __builtin_amdgcn_wave_barrier();
        if (n + RG_WAVE > cap) {
            // keep only the K best so far: rank every entry (registers), then survivors go to list[rank]
            uint64_t keep_key[8];
            int keep_rank[8];
#pragma unroll
            for (int a = 0; a < 8; a++) {
                const int e = a * RG_WAVE + lane;
                keep_key[a] = e < n ? list[e] : ~0ULL;
                int rank = 0;
                if (a * RG_WAVE < n)
                    for (int j = 0; j < n; j++) rank += list[j] < keep_key[a] ? 1 : 0;
                keep_rank[a] = e < n ? rank : INT_MAX;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int a = 0; a < 8; a++)
                if (keep_rank[a] < K) list[keep_rank[a]] = keep_key[a];
            n = n < K ? n : K;
            __builtin_amdgcn_wave_barrier();
        }
    }
    int* row = out_idx + (size_t)q * K;
    for_each_ranked(list, n, [&](int rank, uint64_t k) { if (rank < K) row[rank] = (int)(uint32_t)k; });
    for (int k = n + lane; k < K; k += RG_WAVE) row[k] = ns;  // shadow index (neighbors.cpp:323-324)
    if (lane == 0) {
        if (out_count) out_count[q] = n_total;
        if (out_max_count) atomicMax(out_max_count, n_total);
    }
    __builtin_amdgcn_wave_barrier();      // the LDS list is reused by the wave's next query
    }
}

// ---- self query (queries == the grid's supports: every conv table of the pyramid) -------------------------------------
// CELL-centric: a wave takes one occupied cell, looks its 27 neighbour cells up ONCE (27 lanes, one round trip), stages their
// support runs ONCE into LDS (one more round trip), and then answers every query of the cell -- the cell's own points, which
// are among the staged candidates -- from LDS: distance tests, ballot compaction and the rank sort never touch memory again.
// The per-query kernel above pays those two dependent round trips per QUERY (its waves spend two thirds of their cycles in
// s_waitcnt); a cell holds ~6 queries on 3DMatch-density surfaces.  Same float32 arithmetic, same (d2, index) keys, same
// shrink rule: the tables are bit-identical to k_radius_query's.
constexpr int SELF_CAND = 256;        // staged candidates per cell (cells whose 27-neighbourhood holds more read from global)

__global__ void __launch_bounds__(QUERY_WAVES * RG_WAVE)
k_radius_query_self(const int* __restrict__ s_seg_off, int n_clouds, GridView g, float radius, int K, int cap, int by_index,
                    int* __restrict__ out_idx, int* __restrict__ out_count, int* __restrict__ out_max_count)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = rg_lane();
    // per-wave LDS: cand[SELF_CAND] float4 | list[cap] u64 | runs 64 ints
    unsigned char* base = smem + (size_t)wave * (SELF_CAND * 16 + (size_t)cap * 8 + 256);
    float4* cand = (float4*)base;
    uint64_t* list = (uint64_t*)(base + SELF_CAND * 16);
    int* runs = (int*)(base + SELF_CAND * 16 + (size_t)cap * 8);

    const int ns = s_seg_off[n_clouds];
    const unsigned T = rg_live_table(ns), mask = T - 1u;
    const float r2 = __fmul_rn(radius, radius);  // neighbors.cpp:226
    int wave_max = 0;
    // The live table's T slots are dealt to the launched waves in chunks of `spw` consecutive slots (a power of two, 4..64, sized
    // on the device so that every wave gets work: a small cloud set must not leave most of the chip idle while a few waves walk
    // 64 slots -- ~6 cells x ~6 queries -- one after another); the occupied slots of a chunk are handled one after another.
    const unsigned n_waves = gridDim.x * QUERY_WAVES;
    unsigned spw = 4;
    while (spw < 64u && spw * n_waves < T) spw <<= 1;
    for (unsigned h0 = (blockIdx.x * QUERY_WAVES + wave) * spw; h0 < T; h0 += n_waves * spw) {
        const bool mine = (unsigned)lane < spw;
        const CellSlot* sl = &g.slots[h0 + (mine ? lane : 0)];
        uint4 sa = *(const uint4*)sl;                                        // key | cid | used
        const uint2 sb = *(const uint2*)((const char*)sl + 16);               // cnt | start
        if (!mine) sa.w = 0xffffffffu;                                        // lanes beyond the chunk: "unused"
        unsigned long long occ = __ballot((int)sa.w >= 0 && (int)sb.x > 0);
        while (occ) {
            const int src = __ffsll((long long)occ) - 1;
            occ &= occ - 1;
            const uint64_t ckey = ((uint64_t)__shfl((int)sa.y, src, RG_WAVE) << 32) | (unsigned)__shfl((int)sa.x, src, RG_WAVE);
            const int cid = __shfl((int)sa.z, src, RG_WAVE);
            const int c_cnt = __shfl((int)sb.x, src, RG_WAVE);
            const int c_start = __shfl((int)sb.y, src, RG_WAVE);
            const uint64_t m21 = (1ULL << CELL_BITS) - 1;
            const int64_t cx = (int64_t)(ckey & m21) - CELL_BIAS, cy = (int64_t)((ckey >> CELL_BITS) & m21) - CELL_BIAS,
                          cz = (int64_t)((ckey >> (2 * CELL_BITS)) & m21) - CELL_BIAS;
            int my_cnt = 0, my_start = 0;
            if (lane < 27) {
                const int dx = lane % 3 - 1, dy = (lane / 3) % 3 - 1, dz = lane / 9 - 1;
                if (lane == 13) { my_cnt = c_cnt; my_start = c_start; }
                else slot_find(g.slots, mask, cell_key(cx + dx, cy + dy, cz + dz), cid, my_cnt, my_start);
            }
            int inc = my_cnt;
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, true);
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, true);
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, true);
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xf, 0xf, true);
            inc += (lane >= 16 && lane < 32) ? __builtin_amdgcn_readlane(inc, 15) : 0;
            const int total = __builtin_amdgcn_readlane(inc, 26);
            __builtin_amdgcn_wave_barrier();                                  // previous cell's LDS reads are done
            if (lane < 27) { runs[lane] = inc - my_cnt; runs[32 + lane] = my_start - (inc - my_cnt); }
            __builtin_amdgcn_wave_barrier();
            auto cand_global = [&](int t) -> float4 {
                int c = 0;
#pragma unroll
                for (int step = 16; step > 0; step >>= 1) { const int c2 = c + step; if (c2 < 27 && runs[c2] <= t) c = c2; }
                return g.sorted[t + runs[32 + c]];
            };
            const bool staged = total <= SELF_CAND;                           // wave-uniform
            if (staged) {
                for (int t = lane; t < total; t += RG_WAVE) cand[t] = cand_global(t);
                __builtin_amdgcn_wave_barrier();
            }
            // ---- the cell's queries, one after another (wave-uniform loop)
            for (int qi = 0; qi < c_cnt; qi++) {
                // the query is one of the staged candidates (the centre cell is run 13); LDS broadcast instead of a global load
                const float4 qp = staged ? cand[runs[13] + qi] : g.sorted[c_start + qi];
                const int q = __float_as_int(qp.w);
                int n = 0, n_total = 0;
                for (int t0 = 0; t0 < total; t0 += RG_WAVE) {
                    const int t = t0 + lane;
                    bool in = false;
                    uint64_t key = 0;
                    if (t < total) {
                        const float4 sp = staged ? cand[t] : cand_global(t);
                        const float dx = __fsub_rn(qp.x, sp.x), dy = __fsub_rn(qp.y, sp.y), dz = __fsub_rn(qp.z, sp.z);
                        const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                        in = d2 < r2;
                        key = (by_index ? 0ULL : ((uint64_t)__float_as_uint(d2) << 32)) | (uint32_t)__float_as_int(sp.w);   // order 1: by support index alone
                    }
                    const unsigned long long bal = __ballot(in);
                    if (in) list[n + __popcll(bal & ((1ULL << lane) - 1ULL))] = key;
                    const int add = __popcll(bal);
                    n += add;
                    n_total += add;
                    __builtin_amdgcn_wave_barrier();
                    if (n + RG_WAVE > cap) {                                  // keep only the K best so far (as k_radius_query)
                        uint64_t keep_key[8];
                        int keep_rank[8];
#pragma unroll
                        for (int a = 0; a < 8; a++) {
                            const int e = a * RG_WAVE + lane;
                            keep_key[a] = e < n ? list[e] : ~0ULL;
                            int rank = 0;
                            if (a * RG_WAVE < n)
                                for (int j = 0; j < n; j++) rank += list[j] < keep_key[a] ? 1 : 0;
                            keep_rank[a] = e < n ? rank : INT_MAX;
                        }
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int a = 0; a < 8; a++)
                            if (keep_rank[a] < K) list[keep_rank[a]] = keep_key[a];
                        n = n < K ? n : K;
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                int* row = out_idx + (size_t)q * K;
                for_each_ranked(list, n, [&](int rank, uint64_t k) { if (rank < K) row[rank] = (int)(uint32_t)k; });
                for (int k = n + lane; k < K; k += RG_WAVE) row[k] = ns;
                if (lane == 0 && out_count) out_count[q] = n_total;
                wave_max = n_total > wave_max ? n_total : wave_max;
                __builtin_amdgcn_wave_barrier();                              // the list is reused by the next query
            }
        }
    }
    if (out_max_count && lane == 0 && wave_max > 0) atomicMax(out_max_count, wave_max);
}

// Nearest support inside the ball, in float64 -- the ground-truth overlap test of the reference's data pipeline
// (utils/pointcloud.py:8-65: open3d KDTreeFlann.search_radius_vector_3d on double coordinates, d2 < r^2, first = nearest).
// One thread per query over the 27 cells of the support grid; ties on d2 go to the lower index.  Not on the inference
// path, so no wave-level machinery.
__global__ void __launch_bounds__(256) k_nearest_in_radius(const float* __restrict__ q_xyz, const int* __restrict__ q_seg_off,
                                                           const int* __restrict__ s_seg_off, int n_clouds, GridView g, double r2,
                                                           int* __restrict__ out_idx)
{
    const int nq = q_seg_off[n_clouds], ns = s_seg_off[n_clouds];
    const unsigned mask = rg_live_table(ns) - 1u;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += gridDim.x * blockDim.x) {
        const int cid = rg_find_segment(q_seg_off, n_clouds, q);
        const float qx = q_xyz[3 * (size_t)q], qy = q_xyz[3 * (size_t)q + 1], qz = q_xyz[3 * (size_t)q + 2];
        int64_t cx, cy, cz;
        cell_of(qx, qy, qz, g.inv_cs, cx, cy, cz);
        double best = r2;
        int best_i = -1;
        for (int c = 0; c < 27; c++) {
            int cnt, start;
            slot_find(g.slots, mask, cell_key(cx + c % 3 - 1, cy + (c / 3) % 3 - 1, cz + c / 9 - 1), cid, cnt, start);
            for (int t = 0; t < cnt; t++) {
                const float4 sp = g.sorted[start + t];
                const double dx = (double)qx - (double)sp.x, dy = (double)qy - (double)sp.y, dz = (double)qz - (double)sp.z;
                const double d2 = (dx * dx + dy * dy) + dz * dz;
                const int i = __float_as_int(sp.w);
                if (d2 < best || (d2 == best && best_i >= 0 && i < best_i)) { best = d2; best_i = i; }
            }
        }
        out_idx[q] = best_i;
    }
}

// overlap_pyr[p][q] = clamp(mean over the valid entries of pools row q of overlap_pyr[p-1], 0, 1)   (kpconv.py:553-562;
// a row without a valid entry is 0/0 = NaN there and here)
__global__ void __launch_bounds__(256) k_overlap_avgpool(const float* __restrict__ ov, int ns, const int* __restrict__ nbr, int ld,
                                                         int nq, int H, float* __restrict__ out)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    float sum = 0.f, cnt = 0.f;
    for (int h = 0; h < H; h++) {
        const int i = nbr[(size_t)q * ld + h];
        if (i < ns) { sum += ov[i]; cnt += 1.f; }
    }
    const float v = sum / cnt;
    out[q] = fminf(fmaxf(v, 0.f), 1.f);       // torch.clamp propagates NaN; fminf/fmaxf would not:
    if (v != v) out[q] = v;
}

struct GridBuffers {
    uint64_t* pkey; int* pcid; int* slot_of; int* rep; int* cnt; int* fill;
    uint64_t* tkey; int* tcid; uint64_t* scan_in; uint64_t* cell_start; uint64_t* bsum; float4* sorted; CellSlot* slots;
    unsigned T;
    size_t bytes;
};

GridBuffers carve_grid(void* ws, size_t ws_bytes, int ns_cap)
{
    GridBuffers b;
    RgCarver c(ws, ws_bytes);
    const unsigned T = rg_table_capacity(ns_cap);
    b.T = T;
    b.pkey = c.take<uint64_t>(ns_cap); b.pcid = c.take<int>(ns_cap); b.slot_of = c.take<int>(ns_cap);
    b.rep = c.take<int>(T); b.cnt = c.take<int>(T); b.fill = c.take<int>(T);
    b.tkey = c.take<uint64_t>(T); b.tcid = c.take<int>(T);
    b.scan_in = c.take<uint64_t>(T); b.cell_start = c.take<uint64_t>(T);
    b.bsum = c.take<uint64_t>(rg_cdiv(T, SCAN_TILE) + 2);           // scan state (ticket + one descriptor per tile) + the live table length
    b.sorted = c.take<float4>(ns_cap);
    b.slots = c.take<CellSlot>(T);
    b.bytes = rg_align_up(c.off, 256);
    return b;
}

struct SubsampleBuffers {
    uint64_t *pkey, *scan_in, *scan_out, *bsum, *vkey;
    int *pcid, *slot_of, *members, *rep, *first, *cnt, *fill, *bbox, *unext, *ubefore, *uorder, *dest;
    unsigned T;
    size_t bytes;
};

SubsampleBuffers carve_subsample(void* ws, size_t ws_bytes, int n_cap, int n_clouds, int row_order)
{
    SubsampleBuffers b;
    RgCarver c(ws, ws_bytes);
    b.T = rg_table_capacity(n_cap);
    b.pkey = c.take<uint64_t>(n_cap); b.scan_in = c.take<uint64_t>(n_cap); b.scan_out = c.take<uint64_t>(n_cap);
    b.pcid = c.take<int>(n_cap); b.slot_of = c.take<int>(n_cap); b.members = c.take<int>(n_cap);
    b.rep = c.take<int>(b.T); b.first = c.take<int>(b.T); b.cnt = c.take<int>(b.T); b.fill = c.take<int>(b.T);
    b.bsum = c.take<uint64_t>(rg_cdiv(n_cap, SCAN_TILE) + 1);
    b.bbox = c.take<int>((size_t)n_clouds * 6);
    b.vkey = nullptr; b.unext = b.ubefore = b.uorder = b.dest = nullptr;
    if (row_order == 1) {
        b.vkey = c.take<uint64_t>(n_cap);
        b.unext = c.take<int>((size_t)n_cap + n_clouds);
        b.ubefore = c.take<int>(rg_umap_before_offset(n_cap, n_clouds) + 16);
        b.uorder = c.take<int>(n_cap);
        b.dest = c.take<int>(n_cap);
    }
    b.bytes = rg_align_up(c.off, 256);
    return b;
}

}  // namespace

extern "C" {

size_t regtr_grid_subsample_ordered_ws_bytes(int n_cap, int n_clouds, int row_order)
{
    if (n_cap < 1) n_cap = 1;
    if (n_clouds < 1) n_clouds = 1;
    return carve_subsample(nullptr, ~(size_t)0, n_cap, n_clouds, row_order).bytes + 4096;
}
size_t regtr_grid_subsample_ws_bytes(int n_cap, int n_clouds) { return regtr_grid_subsample_ordered_ws_bytes(n_cap, n_clouds, 0); }

int regtr_grid_subsample_ordered(const float* xyz, const int* seg_off, int n_clouds, int n_cap, float dl, int row_order, int key_mode,
                                 int out_cap, float* out_xyz, int* out_seg_off, void* ws, size_t ws_bytes, void* stream)
{
    if (out_cap <= 0 || out_cap > n_cap) out_cap = n_cap;
    if (row_order == 1 && out_cap != n_cap) return RG_ERR_ARG;      // the container order is replayed over whole clouds
    if (!xyz || !seg_off || !out_xyz || !out_seg_off || n_clouds < 1 || n_cap < 0 || !(dl > 0.f) || row_order < 0 || row_order > 1
        || key_mode < 0 || key_mode > 2 || (row_order == 1 && key_mode != 0))      // the container order belongs to the CPU op's keys
        return RG_ERR_ARG;
    if (ws_bytes < regtr_grid_subsample_ordered_ws_bytes(n_cap, n_clouds, row_order)) return RG_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (n_cap == 0) {
        (void)hipMemsetAsync(out_seg_off, 0, sizeof(int) * (n_clouds + 1), st);
        return RG_OK;
    }
    SubsampleBuffers sb = carve_subsample(ws, ws_bytes, n_cap, n_clouds, row_order);
    const unsigned T = sb.T;
    uint64_t *pkey = sb.pkey, *scan_in = sb.scan_in, *scan_out = sb.scan_out, *bsum = sb.bsum;
    int *pcid = sb.pcid, *slot_of = sb.slot_of, *members = sb.members, *rep = sb.rep, *first = sb.first,
        *cnt = sb.cnt, *fill = sb.fill, *bbox = sb.bbox;
    const int* n_ptr = seg_off + n_clouds;
    const int nb = rg_cdiv(n_cap, 256);

    const int n_state = rg_cdiv(n_cap, SCAN_TILE) + 1;
    const int n_clear = (int)T > n_clouds * 6 ? (int)T : n_clouds * 6;       // (n_state <= T always: T >= 1.5 n_cap)
    k_clear_tables<<<rg_cdiv(n_clear, 256), 256, 0, st>>>(n_ptr, rep, cnt, fill, first, n_state <= SCAN_CHAIN_TILES + 1 ? bsum : nullptr,
                                                          n_state <= SCAN_CHAIN_TILES + 1 ? n_state : 0, bbox, n_clouds * 6);
    k_bbox<<<rg_cdiv(n_cap, 256 * BBOX_ITEMS), 256, 0, st>>>(xyz, seg_off, n_clouds, pcid, bbox);
    k_voxel_keys<<<nb, 256, 0, st>>>(xyz, seg_off, n_clouds, pcid, bbox, dl, key_mode, pkey);
    k_insert<<<nb, 256, 0, st>>>(n_ptr, pkey, pcid, rep, slot_of, first, cnt);
    k_leader_flags<<<nb, 256, 0, st>>>(n_ptr, slot_of, first, cnt, scan_in);
    scan_u64(scan_in, n_ptr, n_cap, bsum, scan_out, st);
    k_scatter_members<<<nb, 256, 0, st>>>(n_ptr, slot_of, first, scan_out, fill, members);
    k_out_offsets<<<rg_cdiv(n_clouds + 1, 64), 64, 0, st>>>(seg_off, n_clouds, scan_in, scan_out, out_cap, out_seg_off);
    if (row_order == 1) {
        k_rank_keys<<<nb, 256, 0, st>>>(n_ptr, slot_of, first, scan_out, pkey, sb.vkey);
        k_umap_order<<<rg_cdiv(n_clouds, 64), 64, 0, st>>>(sb.vkey, out_seg_off, n_clouds, sb.unext, sb.ubefore, sb.uorder, sb.dest);
    }
    k_barycentres<<<nb, 256, 0, st>>>(xyz, n_ptr, slot_of, first, cnt, scan_out, members, sb.dest, out_cap, out_xyz);
    RG_RETURN_IF_LAUNCH_FAILED();
    return RG_OK;
}

int regtr_grid_subsample(const float* xyz, const int* seg_off, int n_clouds, int n_cap, float dl, float* out_xyz,
                         int* out_seg_off, void* ws, size_t ws_bytes, void* stream)
{
    return regtr_grid_subsample_ordered(xyz, seg_off, n_clouds, n_cap, dl, 0, 0, n_cap, out_xyz, out_seg_off, ws, ws_bytes, stream);
}

size_t regtr_cellgrid_ws_bytes(int ns_cap, int n_clouds)
{
    (void)n_clouds;
    if (ns_cap < 1) ns_cap = 1;
    return carve_grid(nullptr, ~(size_t)0, ns_cap).bytes + 4096;
}

// Builds the support-point cell grid (cell size = radius * (1 + 1e-6)) into `ws`; the same `ws` can then serve
// any number of regtr_radius_query calls with that radius over the same supports.
int regtr_cellgrid_build(const float* s_xyz, const int* s_seg_off, int n_clouds, int ns_cap, float radius, void* ws,
                         size_t ws_bytes, void* stream)
{
    if (!s_xyz || !s_seg_off || n_clouds < 1 || ns_cap < 0 || !(radius > 0.f)) return RG_ERR_ARG;
    if (ws_bytes < regtr_cellgrid_ws_bytes(ns_cap, n_clouds)) return RG_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int cap = ns_cap > 0 ? ns_cap : 1;
    GridBuffers b = carve_grid(ws, ws_bytes, cap);
    const int* n_ptr = s_seg_off + n_clouds;
    const double inv_cs = 1.0 / ((double)radius * (1.0 + 1e-6));
    const int n_state = rg_cdiv(b.T, SCAN_TILE) + 1;
    k_clear_tables<<<rg_cdiv(b.T, 256), 256, 0, st>>>(n_ptr, b.rep, b.cnt, b.fill, nullptr, n_state <= SCAN_CHAIN_TILES + 1 ? b.bsum : nullptr,
                                                      n_state <= SCAN_CHAIN_TILES + 1 ? n_state : 0, nullptr, 0);
    if (ns_cap > 0) {
        const int nb = rg_cdiv(ns_cap, 256);
        k_cell_keys<<<nb, 256, 0, st>>>(s_xyz, s_seg_off, n_clouds, inv_cs, b.pkey, b.pcid);
        k_insert<<<nb, 256, 0, st>>>(n_ptr, b.pkey, b.pcid, b.rep, b.slot_of, nullptr, b.cnt);
    }
    // the live table length is computed on the device and kept in bsum's tail for the device-n scan
    int* tn = (int*)(b.bsum + rg_cdiv(b.T, SCAN_TILE) + 1);
    k_table_keys<<<rg_cdiv(b.T, 256), 256, 0, st>>>(b.rep, n_ptr, b.pkey, b.pcid, b.cnt, b.tkey, b.tcid, b.scan_in, tn);
    scan_u64(b.scan_in, tn, (int)b.T, b.bsum, b.cell_start, st);
    if (ns_cap > 0)
        k_scatter_cells<<<rg_cdiv(ns_cap, 256), 256, 0, st>>>(s_xyz, n_ptr, b.slot_of, b.cell_start, b.fill, b.sorted);
    k_pack_slots<<<rg_cdiv(b.T, 256), 256, 0, st>>>(b.rep, n_ptr, b.tkey, b.tcid, b.cnt, b.cell_start, b.slots);
    RG_RETURN_IF_LAUNCH_FAILED();
    return RG_OK;
}

// out_idx [nq_cap * K] int32, padded with Ns_total; order 0: the K NEAREST supports in the ball, rows ascending (d2, support index) --
// the reference's CPU Preprocessor (nanoflann, kpconv.py:243-258); order 1: the FIRST K supports in the ball by support index, rows
// ascending by index -- the reference's PreprocessorGPU (pytorch3d ball_query, kpconv.py:261-288).
// out_count (optional) [nq_cap]: untruncated number of supports inside the ball.
// out_max_count (optional): device int, atomically max-ed with the counts (caller zeroes it).
int regtr_radius_query(const float* q_xyz, const int* q_seg_off, int nq_cap, const int* s_seg_off, int ns_cap,
                       int n_clouds, float radius, int K, int order, const void* grid_ws, size_t ws_bytes, int* out_idx,
                       int* out_count, int* out_max_count, void* stream)
{
    if (!q_xyz || !q_seg_off || !s_seg_off || !out_idx || n_clouds < 1 || K < 1 || K > 448 || !(radius > 0.f) || (order != 0 && order != 1))
        return RG_ERR_ARG;
    if (ws_bytes < regtr_cellgrid_ws_bytes(ns_cap, n_clouds)) return RG_ERR_WORKSPACE;
    if (nq_cap <= 0) return RG_OK;
    hipStream_t st = (hipStream_t)stream;
    GridBuffers b = carve_grid((void*)grid_ws, ws_bytes, ns_cap > 0 ? ns_cap : 1);
    GridView g{b.slots, b.sorted, 1.0 / ((double)radius * (1.0 + 1e-6))};
    // LDS list capacity per query: room for the K survivors of a shrink plus one 64-candidate round
    int cap = (2 * K + 63) / 64 * 64;
    if (cap < 256) cap = 256;
    if (cap > 512) cap = 512;          // 8 register-staged chunks of 64 in the shrink path
    if (cap < K + RG_WAVE) return RG_ERR_ARG;   // K <= 448
    const size_t lds = (size_t)QUERY_WAVES * (cap + 32) * sizeof(uint64_t);
    const int grid = rg_cdiv(nq_cap, QUERY_WAVES) < 256 * 64 ? rg_xcd_grid(rg_cdiv(nq_cap, QUERY_WAVES)) : 256 * 64;   // query runs inside
    k_radius_query<<<grid, QUERY_WAVES * RG_WAVE, lds, st>>>(
        q_xyz, q_seg_off, s_seg_off, n_clouds, g, radius, K, cap, order, out_idx, out_count, out_max_count);
    RG_RETURN_IF_LAUNCH_FAILED();
    return RG_OK;
}

// Self query: the rows of EVERY support point of the grid (queries == supports, e.g. the conv tables of the pyramid) by the
// cell-centric kernel; out_idx [ns_cap, K] indexed by the ORIGINAL support row.  Same results as regtr_radius_query(s_xyz, ...).
int regtr_radius_query_self(const int* s_seg_off, int ns_cap, int n_clouds, float radius, int K, int order, const void* grid_ws,
                            size_t ws_bytes, int* out_idx, int* out_count, int* out_max_count, void* stream)
{
    if (!s_seg_off || !out_idx || !grid_ws || n_clouds < 1 || K < 1 || K > 448 || !(radius > 0.f) || (order != 0 && order != 1)) return RG_ERR_ARG;
    if (ws_bytes < regtr_cellgrid_ws_bytes(ns_cap, n_clouds)) return RG_ERR_WORKSPACE;
    if (ns_cap <= 0) return RG_OK;
    GridBuffers b = carve_grid((void*)grid_ws, ws_bytes, ns_cap);
    GridView g{b.slots, b.sorted, 1.0 / ((double)radius * (1.0 + 1e-6))};
    int cap = (2 * K + 63) / 64 * 64;
    if (cap < 256) cap = 256;
    if (cap > 512) cap = 512;
    if (cap < K + RG_WAVE) return RG_ERR_ARG;
    const size_t lds = (size_t)QUERY_WAVES * (SELF_CAND * 16 + (size_t)cap * 8 + 256);
    // waves for 16 slots each of the table's capacity, at most 32 workgroups per CU; the kernel sizes the chunks for the live table
    const long long chunks = ((long long)b.T + 15) / 16;
    const int grid = (int)(rg_cdiv(chunks, QUERY_WAVES) < 256 * 32 ? rg_cdiv(chunks, QUERY_WAVES) : 256 * 32);
    k_radius_query_self<<<grid, QUERY_WAVES * RG_WAVE, lds, (hipStream_t)stream>>>(s_seg_off, n_clouds, g, radius, K, cap, order, out_idx,
                                                                                     out_count, out_max_count);
    RG_RETURN_IF_LAUNCH_FAILED();
    return RG_OK;
}

// Index (into the stacked supports) of the nearest support of the query's cloud with d2 < radius^2 in float64, -1 if none
// (utils/pointcloud.py:44-55).  grid_ws: a cell grid built over the supports with cell size >= radius.
int regtr_nearest_in_radius(const float* q_xyz, const int* q_seg_off, int nq_cap, const int* s_seg_off, int ns_cap,
                            int n_clouds, double radius, float grid_radius, const void* grid_ws, size_t ws_bytes, int* out_idx,
                            void* stream)
{
    if (!q_xyz || !q_seg_off || !s_seg_off || !out_idx || !grid_ws || n_clouds < 1 || !(radius > 0.0) ||
        !((double)grid_radius * (1.0 + 1e-6) >= radius))
        return RG_ERR_ARG;
    if (ws_bytes < regtr_cellgrid_ws_bytes(ns_cap, n_clouds)) return RG_ERR_WORKSPACE;
    if (nq_cap <= 0) return RG_OK;
    GridBuffers b = carve_grid((void*)grid_ws, ws_bytes, ns_cap > 0 ? ns_cap : 1);
    GridView g{b.slots, b.sorted, 1.0 / ((double)grid_radius * (1.0 + 1e-6))};
    const int blocks = rg_cdiv(nq_cap, 256) < 4096 ? rg_cdiv(nq_cap, 256) : 4096;
    k_nearest_in_radius<<<blocks, 256, 0, (hipStream_t)stream>>>(q_xyz, q_seg_off, s_seg_off, n_clouds, g, radius * radius, out_idx);
    RG_RETURN_IF_LAUNCH_FAILED();
    return RG_OK;
}

int regtr_overlap_avgpool(const float* ov, int ns, const int* nbr, int ld_nbr, int nq, int H, float* out, void* stream)
{
    if (!ov || !nbr || !out || ns < 0 || nq < 0 || H < 1 || ld_nbr < H) return RG_ERR_ARG;
    if (nq == 0) return RG_OK;
    k_overlap_avgpool<<<rg_cdiv(nq, 256), 256, 0, (hipStream_t)stream>>>(ov, ns, nbr, ld_nbr, nq, H, out);
    RG_RETURN_IF_LAUNCH_FAILED();
    return RG_OK;
}

}  // extern "C"
