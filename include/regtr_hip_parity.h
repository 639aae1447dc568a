/* C ABI of libregtr_parity.so -- the PARITY-MODE neighbour search of the RegTR path on MI355X (gfx950): the reference CPU op's
 * implementation-defined row order (nanoflann KD-tree visiting order + std::sort), reproduced on the GPU so that the product can be held
 * against the reference's own outputs row for row (cfg.kpconv_ref_row_order, cpp_wrappers.reference_order()).
 *
 * A library of its own since ABI 11: its KD-tree functions restate nanoflann 1.3.0 closely (BSD; THIRD_PARTY_NOTICES.md, notice retained in
 * regtr_amd/csrc/ref_kdtree.h) and serve a checking mode only -- the product library libregtr_hip.so (include/regtr_hip.h) neither contains nor
 * calls them.  Same conventions as regtr_hip.h: device pointers + sizes + stream, int status (REGTR_OK ...), no exceptions across the ABI.
 * Reference interface replaced: cpp_neighbors.batch_query in the reference's own row order
 * (/root/reference/src/models/backbone_kpconv/cpp_wrappers/cpp_neighbors/neighbors/neighbors.cpp:211-332, kpconv.py:243-258). */
#ifndef REGTR_HIP_PARITY_H
#define REGTR_HIP_PARITY_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* REGTR_ABI_VERSION of the regtr_hip.h this library was built against (a binding checks both libraries) */
int regtr_parity_abi_version(void);

/* Parity mode (cfg.kpconv_ref_row_order): the same neighbour sets in the REFERENCE's row order -- nanoflann's KD-tree
 * visiting order passed through std::sort on the distance alone (neighbors.cpp:246-267, nanoflann.hpp:857-1003,1348-1412,
 * 1285-1287) -- so that rows truncated to K keep the very supports the reference keeps when distances tie.  One thread
 * builds the tree of one cloud, one thread answers one query (csrc/ref_kdtree.h, csrc/ref_order.hip); a parity tool, not a throughput path.
 *   regtr_kdtree_build          tree of every cloud into ws (regtr_kdtree_ws_bytes)
 *   regtr_kdtree_radius_query   out_idx [nq_cap,K] (first K of each row, pad = Ns_total), out_count / out_max_count as in
 *                               regtr_radius_query; list_cap >= the largest in-ball count (rows with more are cut at
 *                               list_cap BEFORE sorting -- the caller re-runs with list_cap = *out_max_count);
 *                               scratch: regtr_kdtree_query_scratch_bytes(list_cap).
 *                               out_status (optional device int, zeroed by the caller): set to 1 if a query overflowed its
 *                               traversal stack (96 pending subtrees; rows then invalid). */
size_t regtr_kdtree_ws_bytes(int ns_cap, int n_clouds);
size_t regtr_kdtree_query_scratch_bytes(int list_cap);
int regtr_kdtree_build(const float* s_xyz, const int* s_seg_off, int n_clouds, int ns_cap, void* ws, size_t ws_bytes,
                       void* stream);
int regtr_kdtree_radius_query(const float* q_xyz, const int* q_seg_off, int nq_cap, const float* s_xyz,
                              const int* s_seg_off, int ns_cap, int n_clouds, float radius, int K, int list_cap,
                              const void* tree_ws, size_t ws_bytes, void* scratch, size_t scratch_bytes, int* out_idx,
                              int* out_count, int* out_max_count, int* out_status, void* stream);

#ifdef __cplusplus
}
#endif
#endif
