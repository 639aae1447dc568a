// Reference-order parity mode, neighbour tables: the reference's CPU op (batch_nanoflann_neighbors,
// cpp_wrappers/cpp_neighbors/neighbors/neighbors.cpp:211-332) returns rows in nanoflann's KD-tree visiting order passed
// through std::sort on the distance alone, then kpconv.py:255-256 keeps the first neighborhood_limits columns -- so on
// clouds with exactly equidistant supports (the 6 mm lattice of the 3DMatch fragments) WHICH supports a truncated row
// keeps is decided by that tree and that sort.  These kernels replay both (ref_kdtree.h: the same serial functions
// tests/test_ref_order.py checks on the host against the unmodified reference C++): one thread builds the tree of one
// cloud, one thread answers one query.  Built into libregtr_parity.so (include/regtr_hip_parity.h), NOT into the product library.
// Parity tool, not a throughput path -- the default tables come from
// preprocess.hip (cell grid, one wave per query, ascending (d2, index)) and are identical as SETS except on rows whose
// K-th and (K+1)-th distances tie.
#include "common.h"
#include "regtr_hip_parity.h"      // this library's own C ABI: definitions checked against their declarations
#include "ref_kdtree.h"

namespace {

constexpr int KD_STACK = 96;                 // pending far children per query (tree depth is ~20 on 20k-point clouds)
constexpr int KD_QUERY_THREADS = 256;
constexpr int KD_QUERY_BLOCKS = 512;

struct KdBuffers {
    int* vind; RgKdNode* nodes; RgKdFrame* frames; float* boxes; float* root;
    size_t bytes;
};

KdBuffers carve_kd(void* ws, size_t ws_bytes, int ns_cap, int n_clouds)
{
    KdBuffers b;
    RgCarver c(ws, ws_bytes);
    b.vind = c.take<int>(ns_cap);
    b.nodes = c.take<RgKdNode>(2 * (size_t)ns_cap + n_clouds);
    b.frames = c.take<RgKdFrame>((size_t)ns_cap + n_clouds);
    b.boxes = c.take<float>(12 * (size_t)ns_cap + 6 * (size_t)n_clouds);
    b.root = c.take<float>(6 * (size_t)n_clouds);
    b.bytes = rg_align_up(c.off, 256);
    return b;
}

__global__ void k_kd_build(const float* __restrict__ s_xyz, const int* __restrict__ s_seg_off, int n_clouds, KdBuffers b)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_clouds) return;
    const int base = s_seg_off[c], n = s_seg_off[c + 1] - base;
    if (n <= 0) return;
    rg_kd_build(s_xyz + 3 * (size_t)base, n, b.vind + base, b.nodes + 2 * (size_t)base + c, b.frames + base + c,
                b.boxes + 12 * (size_t)base + 6 * (size_t)c, b.root + 6 * (size_t)c);
}

__global__ void __launch_bounds__(KD_QUERY_THREADS)
k_kd_query(const float* __restrict__ q_xyz, const int* __restrict__ q_seg_off, const float* __restrict__ s_xyz,
           const int* __restrict__ s_seg_off, int n_clouds, KdBuffers b, float radius, int K, int list_cap,
           uint64_t* __restrict__ lists, int* __restrict__ stacks, int* __restrict__ out_idx, int* __restrict__ out_count,
           int* __restrict__ out_max_count, int* __restrict__ out_status)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int nq = q_seg_off[n_clouds], ns = s_seg_off[n_clouds];
    uint64_t* list = lists + (size_t)tid * list_cap;
    int* stack = stacks + (size_t)tid * (5 * KD_STACK);
    const float r2 = __fmul_rn(radius, radius);                                   // neighbors.cpp:226
    for (int q = tid; q < nq; q += gridDim.x * blockDim.x) {
        const int c = rg_find_segment(q_seg_off, n_clouds, q);
        const int base = s_seg_off[c], n = s_seg_off[c + 1] - base;
        int cnt = 0;
        if (n > 0) {
            const float qp[3] = {q_xyz[3 * (size_t)q], q_xyz[3 * (size_t)q + 1], q_xyz[3 * (size_t)q + 2]};
            cnt = rg_kd_radius_search(s_xyz + 3 * (size_t)base, b.vind + base, b.nodes + 2 * (size_t)base + c,
                                      b.root + 6 * (size_t)c, qp, r2, list, list_cap, stack, KD_STACK);
        }
        int* row = out_idx + (size_t)q * K;
        if (cnt < 0) {                                                            // traversal stack overflow: reported, row padded
            if (out_status) atomicExch(out_status, 1);
            cnt = 0;
        }
        const int m = cnt < list_cap ? cnt : list_cap;
        for (int k = 0; k < K; k++) row[k] = k < m ? (int)(uint32_t)list[k] + base : ns;       // neighbors.cpp:319-324
        if (out_count) out_count[q] = cnt;
        if (out_max_count) atomicMax(out_max_count, cnt);
    }
}

}  // namespace

extern "C" {

int regtr_parity_abi_version(void) { return REGTR_ABI_VERSION; }

size_t regtr_kdtree_ws_bytes(int ns_cap, int n_clouds)
{
    if (ns_cap < 1) ns_cap = 1;
    if (n_clouds < 1) n_clouds = 1;
    return carve_kd(nullptr, ~(size_t)0, ns_cap, n_clouds).bytes + 4096;
}

size_t regtr_kdtree_query_scratch_bytes(int list_cap)
{
    if (list_cap < 16) list_cap = 16;
    return (size_t)KD_QUERY_BLOCKS * KD_QUERY_THREADS * ((size_t)list_cap * sizeof(uint64_t) + 5 * KD_STACK * sizeof(int)) + 4096;
}

int regtr_kdtree_build(const float* s_xyz, const int* s_seg_off, int n_clouds, int ns_cap, void* ws, size_t ws_bytes,
                       void* stream)
{
    if (!s_xyz || !s_seg_off || !ws || n_clouds < 1 || ns_cap < 0) return RG_ERR_ARG;
    if (ws_bytes < regtr_kdtree_ws_bytes(ns_cap, n_clouds)) return RG_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    KdBuffers b = carve_kd(ws, ws_bytes, ns_cap > 0 ? ns_cap : 1, n_clouds);
    // one thread per cloud, one cloud per workgroup so that clouds build on different CUs
    k_kd_build<<<n_clouds, 1, 0, st>>>(s_xyz, s_seg_off, n_clouds, b);
    RG_RETURN_IF_LAUNCH_FAILED();
    return RG_OK;
}

int regtr_kdtree_radius_query(const float* q_xyz, const int* q_seg_off, int nq_cap, const float* s_xyz,
                              const int* s_seg_off, int ns_cap, int n_clouds, float radius, int K, int list_cap,
                              const void* tree_ws, size_t ws_bytes, void* scratch, size_t scratch_bytes, int* out_idx,
                              int* out_count, int* out_max_count, int* out_status, void* stream)
{
    if (!q_xyz || !q_seg_off || !s_xyz || !s_seg_off || !tree_ws || !scratch || !out_idx || n_clouds < 1 || K < 1 ||
        list_cap < 16 || !(radius > 0.f))
        return RG_ERR_ARG;
    if (ws_bytes < regtr_kdtree_ws_bytes(ns_cap, n_clouds) || scratch_bytes < regtr_kdtree_query_scratch_bytes(list_cap))
        return RG_ERR_WORKSPACE;
    if (nq_cap <= 0) return RG_OK;
    KdBuffers b = carve_kd((void*)tree_ws, ws_bytes, ns_cap > 0 ? ns_cap : 1, n_clouds);
    uint64_t* lists = (uint64_t*)scratch;
    int* stacks = (int*)(lists + (size_t)KD_QUERY_BLOCKS * KD_QUERY_THREADS * list_cap);
    int blocks = rg_cdiv(nq_cap, KD_QUERY_THREADS);
    if (blocks > KD_QUERY_BLOCKS) blocks = KD_QUERY_BLOCKS;
    k_kd_query<<<blocks, KD_QUERY_THREADS, 0, (hipStream_t)stream>>>(q_xyz, q_seg_off, s_xyz, s_seg_off, n_clouds, b, radius, K,
                                                                     list_cap, lists, stacks, out_idx, out_count, out_max_count, out_status);
    RG_RETURN_IF_LAUNCH_FAILED();
    return RG_OK;
}

}  // extern "C"
