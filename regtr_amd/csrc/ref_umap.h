// Reference-order parity mode (cfg.kpconv_ref_row_order), part 1: the grid-subsample ROW ORDER of the reference's CPU op = the iteration order
// of the libstdc++ std::unordered_map<size_t, SampledData> it fills with emplace in input order and then walks (grid_subsampling.cpp:48,58-59,85).
// A restatement of what libstdc++ (GCC 11: bits/hashtable.h, hashtable_policy.h) does from its documented policies (bucket schedule, insertion
// rule) -- no libstdc++ source is reproduced -- as a plain serial function over caller-provided arrays, so that the SAME code runs inside the HIP
// kernel (one thread per cloud: regtr_grid_subsample_ordered, row_order 1) and, compiled for the host, in tests/test_ref_order.py against the
// real std::unordered_map.  The default (fast) path never calls it.
// (Part 2 -- the neighbour ROW order: nanoflann's KD-tree visiting order + std::sort -- lives in ref_kdtree.h and is compiled into
//  libregtr_parity.so only, not into the product library.)
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define RG_HD __host__ __device__ __forceinline__
#else
#define RG_HD inline
#endif

// ---------------------------------------------------------------------------------------------------------------------
// 1. std::unordered_map<size_t, T> iteration order
// ---------------------------------------------------------------------------------------------------------------------
// Bucket-count schedule of libstdc++'s _Prime_rehash_policy (max_load_factor 1, growth factor 2, __prime_list lookup)
// for a map that starts empty: inserting the element that makes size() == at_size first rehashes to `buckets`.
// Generated from the real container (tests/test_ref_order.py re-checks it against the toolchain's libstdc++).
struct RgUmapGrowth { uint32_t at_size, buckets; };
#define RG_UMAP_GROWTH_LEN 25
#define RG_UMAP_GROWTH_TABLE                                                                                             \
    {{1u, 13u}, {14u, 29u}, {30u, 59u}, {60u, 127u}, {128u, 257u}, {258u, 541u}, {542u, 1109u}, {1110u, 2357u},          \
     {2358u, 5087u}, {5088u, 10273u}, {10274u, 20753u}, {20754u, 42043u}, {42044u, 85229u}, {85230u, 172933u},           \
     {172934u, 351061u}, {351062u, 712697u}, {712698u, 1447153u}, {1447154u, 2938679u}, {2938680u, 5967347u},            \
     {5967348u, 12117689u}, {12117690u, 24607243u}, {24607244u, 49969847u}, {49969848u, 101473717u},                     \
     {101473718u, 206062531u}, {206062532u, 418451333u}}

// bucket count of a map holding m elements (m >= 1); also an upper bound for every smaller map
RG_HD uint32_t rg_umap_bucket_count(uint32_t m)
{
    const RgUmapGrowth g[RG_UMAP_GROWTH_LEN] = RG_UMAP_GROWTH_TABLE;
    uint32_t b = 1;
    for (int i = 0; i < RG_UMAP_GROWTH_LEN; i++)
        if (m >= g[i].at_size) b = g[i].buckets;
    return b;
}

// Where cloud c's `before` array starts inside one shared scratch array, given that the clouds before it hold `base`
// elements in total: rg_umap_bucket_count(m) <= 2.16 m + 13 (worst ratio of the schedule: 5087 buckets at 2358 elements).
RG_HD size_t rg_umap_before_offset(size_t base, size_t c) { return base * 11 / 5 + 16 * c; }

// keys[0 .. m): the distinct keys in insertion order (hash = identity, bucket = key % bucket_count).
// next[m + 1], before[rg_umap_bucket_count(m)]: scratch.  order[m]: order[p] = insertion index of the p-th element
// a range-for over the map visits.
//   _M_insert_bucket_begin (hashtable.h): a node goes to the FRONT of its bucket; a bucket that was empty goes to the front
//   of the whole list (the bucket then "begins" at the before-begin sentinel).  _M_rehash_aux(unique) relinks the nodes in
//   their current list order by the same rule.
RG_HD void rg_umap_iteration_order(const uint64_t* keys, int m, int* next, int* before, int* order)
{
    const RgUmapGrowth g[RG_UMAP_GROWTH_LEN] = RG_UMAP_GROWTH_TABLE;
    const int BB = m, NIL = -1, EMPTY = -2;      // BB: the before-begin sentinel "node"
    uint64_t nb = 1;
    int gi = 0;
    next[BB] = NIL;
    before[0] = EMPTY;
    for (int e = 0; e < m; e++) {
        if (gi < RG_UMAP_GROWTH_LEN && (uint32_t)(e + 1) == g[gi].at_size) {      // _M_rehash_aux, unique keys
            nb = g[gi].buckets;
            gi++;
            for (uint64_t b = 0; b < nb; b++) before[b] = EMPTY;
            int p = next[BB];
            next[BB] = NIL;
            uint64_t bbegin_bkt = 0;
            while (p != NIL) {
                const int nx = next[p];
                const uint64_t bkt = keys[p] % nb;
                if (before[bkt] == EMPTY) {
                    next[p] = next[BB];
                    next[BB] = p;
                    before[bkt] = BB;
                    if (next[p] != NIL) before[bbegin_bkt] = p;
                    bbegin_bkt = bkt;
                } else {
                    next[p] = next[before[bkt]];
                    next[before[bkt]] = p;
                }
                p = nx;
            }
        }
        const uint64_t bkt = keys[e] % nb;                                        // _M_insert_bucket_begin
        if (before[bkt] != EMPTY) {
            next[e] = next[before[bkt]];
            next[before[bkt]] = e;
        } else {
            next[e] = next[BB];
            next[BB] = e;
            if (next[e] != NIL) before[keys[next[e]] % nb] = e;
            before[bkt] = BB;
        }
    }
    int p = next[BB];
    for (int i = 0; i < m; i++) { order[i] = p; p = next[p]; }
}

