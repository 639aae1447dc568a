// TEST INFRASTRUCTURE ONLY.  Compiles the product's reference-order emulation (regtr_amd/csrc/ref_umap.h + ref_kdtree.h -- the very
// functions the HIP parity-mode kernels run, one thread per cloud / query) for the HOST, next to the real things they
// emulate: libstdc++'s std::unordered_map iteration and std::sort.  tests/test_ref_order.py compares the two, and the
// emulated neighbour tables against the unmodified reference C++ (oracle/_ref).  Nothing in the product loads this.
#include "../regtr_amd/csrc/ref_umap.h"
#include "../regtr_amd/csrc/ref_kdtree.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <unordered_map>
#include <utility>
#include <vector>

extern "C" {

void emul_umap_order(const uint64_t* keys, int m, int* order)
{
    std::vector<int> next(m + 1), before(rg_umap_bucket_count(m > 0 ? m : 1));
    rg_umap_iteration_order(keys, m, next.data(), before.data(), order);
}
void real_umap_order(const uint64_t* keys, int m, int* order)
{
    std::unordered_map<size_t, int> mp;
    for (int i = 0; i < m; i++)
        if (mp.count(keys[i]) < 1) mp.emplace(keys[i], i);       // grid_subsampling.cpp:58-59
    int p = 0;
    for (auto& kv : mp) order[p++] = kv.second;
}
// growth schedule of the live libstdc++: pairs (size at which the bucket count changed, new bucket count)
int real_umap_growth(int max_size, uint32_t* out, int cap)
{
    std::unordered_map<size_t, int> mp;
    size_t bc = mp.bucket_count();
    int n = 0;
    for (int i = 0; i < max_size; i++) {
        mp.emplace((size_t)i * 7919u + 3, 0);
        if (mp.bucket_count() != bc) {
            bc = mp.bucket_count();
            if (n < cap) { out[2 * n] = (uint32_t)mp.size(); out[2 * n + 1] = (uint32_t)bc; }
            n++;
        }
    }
    return n;
}
int emul_umap_growth(uint32_t* out, int cap)
{
    const RgUmapGrowth g[RG_UMAP_GROWTH_LEN] = RG_UMAP_GROWTH_TABLE;
    for (int i = 0; i < RG_UMAP_GROWTH_LEN && i < cap; i++) { out[2 * i] = g[i].at_size; out[2 * i + 1] = g[i].buckets; }
    return RG_UMAP_GROWTH_LEN;
}

void emul_sort(uint64_t* v, int n)
{
    std::vector<int> stack(3 * 80);
    rg_std_sort(v, n, stack.data());
}
void emul_heap_sort(uint64_t* v, int n) { rg_ss_heap_sort(v, 0, n); }
void real_sort(uint64_t* v, int n)
{
    std::vector<std::pair<size_t, float>> a(n);
    for (int i = 0; i < n; i++) { uint32_t u = (uint32_t)(v[i] >> 32); float f; memcpy(&f, &u, 4); a[i] = {(size_t)(uint32_t)v[i], f}; }
    std::sort(a.begin(), a.end(), [](const std::pair<size_t, float>& p1, const std::pair<size_t, float>& p2) { return p1.second < p2.second; });
    for (int i = 0; i < n; i++) { uint32_t u; memcpy(&u, &a[i].second, 4); v[i] = ((uint64_t)u << 32) | (uint32_t)a[i].first; }
}
void real_heap_sort(uint64_t* v, int n)
{
    auto lt = [](uint64_t a, uint64_t b) { return (uint32_t)(a >> 32) < (uint32_t)(b >> 32); };
    std::partial_sort(v, v + n, v + n, lt);
}

// batch_query semantics (neighbors.cpp:211-332) through the emulation: untruncated (nq, max_count) int32 table, pad = ns.
int emul_batch_neighbors(const float* queries, int nq, const float* supports, int ns, const int* q_batches,
                         const int* s_batches, int nb, float radius, int** out)
{
    const float r2 = radius * radius;
    std::vector<std::vector<uint64_t>> rows(nq);
    int qb = 0, sb = 0, max_count = 0;
    for (int b = 0; b < nb; b++) {
        const int n = s_batches[b];
        if (n > 0 && q_batches[b] > 0) {
            const float* pts = supports + 3 * (size_t)sb;
            std::vector<int> vind(n), stack(5 * (n + 2) + 3 * 80);
            std::vector<RgKdNode> nodes(2 * n);
            std::vector<RgKdFrame> frames(n + 1);
            std::vector<float> boxes(12 * (size_t)n);
            float root_box[6];
            rg_kd_build(pts, n, vind.data(), nodes.data(), frames.data(), boxes.data(), root_box);
            std::vector<uint64_t> list(n);
            for (int i = qb; i < qb + q_batches[b]; i++) {
                const int c = rg_kd_radius_search(pts, vind.data(), nodes.data(), root_box, queries + 3 * (size_t)i, r2,
                                                  list.data(), n, stack.data(), n + 18);
                rows[i].assign(list.begin(), list.begin() + c);
                if (c > max_count) max_count = c;
            }
        }
        qb += q_batches[b];
        sb += s_batches[b];
    }
    *out = (int*)malloc(sizeof(int) * ((size_t)nq * max_count + 1));
    qb = 0; sb = 0;
    int b = 0;
    for (int i = 0; i < nq; i++) {
        while (b < nb && i >= qb + q_batches[b]) { qb += q_batches[b]; sb += s_batches[b]; b++; }
        for (int j = 0; j < max_count; j++)
            (*out)[(size_t)i * max_count + j] = j < (int)rows[i].size() ? (int)(uint32_t)rows[i][j] + sb : ns;
    }
    return max_count;
}

void emul_free(void* p) { free(p); }

}  // extern "C"
