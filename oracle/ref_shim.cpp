// TEST INFRASTRUCTURE ONLY -- never linked into or called by the product path.
//
// extern "C" shim over the UNMODIFIED reference C++ (compiled in place from
// /root/reference by oracle/Makefile into oracle/_ref/libref_oracle.so).  The
// reference's own CPython bindings (cpp_neighbors/wrapper.cpp,
// cpp_subsampling/wrapper.cpp) target the numpy-1.x C-API and do not build
// against numpy 2.x, so this shim exposes the two entry points they wrap:
//   batch_nanoflann_neighbors  cpp_neighbors/neighbors/neighbors.cpp:211-332
//   batch_grid_subsampling     cpp_subsampling/grid_subsampling/grid_subsampling.cpp:109-211
// with the same copy-in / copy-out the wrappers do (wrapper.cpp:188-227,
// cpp_subsampling/wrapper.cpp:270-322).
#include "cpp_neighbors/neighbors/neighbors.h"
#include "cpp_subsampling/grid_subsampling/grid_subsampling.h"
#include <cstring>
#include <cstdlib>

extern "C" {

// Returns max_count (row width); *out is malloc'ed [nq * max_count] int32,
// release with ref_free.  Untruncated table, exactly what batch_query returns
// before kpconv.py:255-256 slices it.
int ref_batch_neighbors(const float* queries, int nq, const float* supports, int ns,
                        const int* q_batches, const int* s_batches, int nb,
                        float radius, int** out)
{
    std::vector<PointXYZ> q((const PointXYZ*)queries, (const PointXYZ*)queries + nq);
    std::vector<PointXYZ> s((const PointXYZ*)supports, (const PointXYZ*)supports + ns);
    std::vector<int> qb(q_batches, q_batches + nb), sb(s_batches, s_batches + nb);
    std::vector<int> idx;
    batch_nanoflann_neighbors(q, s, qb, sb, idx, radius);
    int width = nq > 0 ? (int)(idx.size() / (size_t)nq) : 0;
    *out = (int*)malloc(sizeof(int) * (idx.size() + 1));
    memcpy(*out, idx.data(), sizeof(int) * idx.size());
    return width;
}

// Same contract but through the brute-force, insertion-sorted variant
// (neighbors.cpp:125-208): ties keep ascending support index.
int ref_batch_ordered_neighbors(const float* queries, int nq, const float* supports, int ns,
                                const int* q_batches, const int* s_batches, int nb,
                                float radius, int** out)
{
    std::vector<PointXYZ> q((const PointXYZ*)queries, (const PointXYZ*)queries + nq);
    std::vector<PointXYZ> s((const PointXYZ*)supports, (const PointXYZ*)supports + ns);
    std::vector<int> qb(q_batches, q_batches + nb), sb(s_batches, s_batches + nb);
    std::vector<int> idx;
    batch_ordered_neighbors(q, s, qb, sb, idx, radius);
    int width = nq > 0 ? (int)(idx.size() / (size_t)nq) : 0;
    *out = (int*)malloc(sizeof(int) * (idx.size() + 1));
    memcpy(*out, idx.data(), sizeof(int) * idx.size());
    return width;
}

// Returns number of subsampled points; *out_pts malloc'ed [m*3] float,
// out_batches[nb] filled with per-cloud counts.  Row order is the reference's
// (libstdc++ unordered_map iteration order, grid_subsampling.cpp:85).
int ref_batch_grid_subsampling(const float* points, int n, const int* batches, int nb,
                               float sampleDl, int max_p, float** out_pts, int* out_batches)
{
    std::vector<PointXYZ> p((const PointXYZ*)points, (const PointXYZ*)points + n);
    std::vector<int> b(batches, batches + nb);
    std::vector<PointXYZ> sp;
    std::vector<float> f, sf;
    std::vector<int> c, sc, sb;
    batch_grid_subsampling(p, sp, f, sf, c, sc, b, sb, sampleDl, max_p);
    *out_pts = (float*)malloc(sizeof(float) * 3 * (sp.size() + 1));
    memcpy(*out_pts, sp.data(), sizeof(float) * 3 * sp.size());
    for (int i = 0; i < nb; i++) out_batches[i] = sb[i];
    return (int)sp.size();
}

void ref_free(void* p) { free(p); }

}  // extern "C"
