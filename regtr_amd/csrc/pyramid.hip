// The preprocessing pyramid of a SMALL batch (/root/reference/src/models/backbone_kpconv/kpconv.py:426-537 PreprocessorGPU.forward /
// :298-414 Preprocessor.forward: per level cell grid -> conv table -> grid subsample -> pool table) enqueued by ONE call per phase.
//
// Why: for a pair or two per forward -- the reference's own operating mode, conf/3dmatch.yaml:11 -- the pyramid is ~70 launches of 2-30 us,
// 0.45 ms of GPU time, and it is PACED BY THE HOST: issued from Python, op by op, it takes 0.55 ms to enqueue, and until the level sizes
// have been read back behind it nothing else of the forward can start.  From C the same launches go out in ~0.2 ms, and in two phases, so
// that the caller can hand the level-0 encoder blocks (which need only level 0's conv table) to another stream in between:
//   phase 1   level 0: cell grid + conv table                         -> the caller records an event, starts the level-0 blocks
//   phase 2   level 0: subsample + pool table; levels 1 ..: everything
// NOTHING is computed differently: the library's own entry points (regtr_cellgrid_build, regtr_radius_query, regtr_grid_subsample_ordered)
// with the arguments regtr_amd/kpconv.py passes them one by one; tables and points are bit-identical (tests/test_gpu_model.py).
// Regime: fewer than 262144 input points (regtr_amd/kpconv.py CAPACITY_MIN_POINTS, regtr_amd/ops.py SELF_QUERY_MIN_POINTS): every level is
// sized at the input capacity and every table comes from the per-query radius kernel.
#include "common.h"

namespace {

constexpr int PYR_SMALL_POINTS = 262144;

bool pyr_check(const regtr_pyramid_level_t* lv, int n_levels, int n_clouds)
{
    if (!lv || n_levels < 1 || n_levels > 16 || n_clouds < 1) return false;
    for (int l = 0; l < n_levels; l++) {
        const regtr_pyramid_level_t& L = lv[l];
        if (L.cap < 1 || L.cap >= PYR_SMALL_POINTS || !(L.radius > 0.f) || L.K < 1 || L.K > 448 || !L.points || !L.seg_off) return false;
        if (L.has_conv && !L.conv_idx) return false;
        if (L.strided && (!(L.dl > 0.f) || L.cap_next < 1 || L.cap_next > L.cap || !L.points_next || !L.seg_next || !L.pool_idx)) return false;
        if (L.strided && l + 1 < n_levels && (lv[l + 1].points != L.points_next || lv[l + 1].seg_off != L.seg_next || lv[l + 1].cap != L.cap_next))
            return false;
    }
    return true;
}

void pyr_sizes(const regtr_pyramid_level_t* lv, int n_levels, int n_clouds, size_t& grid_bytes, size_t& sub_bytes)
{
    grid_bytes = sub_bytes = 0;
    for (int l = 0; l < n_levels; l++) {
        const size_t g = regtr_cellgrid_ws_bytes(lv[l].cap, n_clouds);
        if (g > grid_bytes) grid_bytes = g;
        if (lv[l].strided) {
            const size_t s = regtr_grid_subsample_ordered_ws_bytes(lv[l].cap, n_clouds, 0);
            if (s > sub_bytes) sub_bytes = s;
        }
    }
    grid_bytes = rg_align_up(grid_bytes, 256);
    sub_bytes = rg_align_up(sub_bytes, 256);
}

}  // namespace

extern "C" {

int regtr_pyramid_supported(const regtr_pyramid_level_t* levels, int n_levels, int n_clouds) { return pyr_check(levels, n_levels, n_clouds) ? 1 : 0; }

size_t regtr_pyramid_ws_bytes(const regtr_pyramid_level_t* levels, int n_levels, int n_clouds)
{
    if (!pyr_check(levels, n_levels, n_clouds)) return 0;
    size_t g, s;
    pyr_sizes(levels, n_levels, n_clouds, g, s);
    return g + s + 256;
}

// levels: HOST array (include/regtr_hip.h).  order: regtr_radius_query's (0 nearest K, 1 first K by index); key_mode: regtr_grid_subsample_ordered's.
// phase 0 = the whole pyramid, 1 = level 0's cell grid + conv table, 2 = everything after phase 1 (same ws, untouched in between).
int regtr_pyramid_fwd(const regtr_pyramid_level_t* levels, int n_levels, int n_clouds, int order, int key_mode, int phase, void* ws,
                      size_t ws_bytes, void* stream)
{
    if (!ws || phase < 0 || phase > 2 || order < 0 || order > 1 || key_mode < 0 || key_mode > 2 || !pyr_check(levels, n_levels, n_clouds)) return RG_ERR_ARG;
    size_t grid_bytes, sub_bytes;
    pyr_sizes(levels, n_levels, n_clouds, grid_bytes, sub_bytes);
    if (ws_bytes < grid_bytes + sub_bytes || (uintptr_t)ws % 256) return RG_ERR_WORKSPACE;
    void* grid_ws = ws;
    void* sub_ws = (char*)ws + grid_bytes;
    for (int l = 0; l < n_levels; l++) {
        const regtr_pyramid_level_t& L = levels[l];
        const size_t gb = regtr_cellgrid_ws_bytes(L.cap, n_clouds);
        int rc;
        if (!(phase == 2 && l == 0)) {                                   // (phase 2 finds level 0's grid where phase 1 left it)
            rc = regtr_cellgrid_build(L.points, L.seg_off, n_clouds, L.cap, L.radius, grid_ws, gb, stream);
            if (rc != RG_OK) return rc;
            if (L.has_conv) {                                            // kpconv.py:472 / :349-351
                rc = regtr_radius_query(L.points, L.seg_off, L.cap, L.seg_off, L.cap, n_clouds, L.radius, L.K, order, grid_ws, gb, L.conv_idx,
                                        nullptr, nullptr, stream);
                if (rc != RG_OK) return rc;
            }
        }
        if (phase == 1) return RG_OK;
        if (L.strided) {
            rc = regtr_grid_subsample_ordered(L.points, L.seg_off, n_clouds, L.cap, L.dl, 0, key_mode, L.cap_next, L.points_next, L.seg_next,
                                              sub_ws, regtr_grid_subsample_ordered_ws_bytes(L.cap, n_clouds, 0), stream);     // :486-489 / :363-366
            if (rc != RG_OK) return rc;
            rc = regtr_radius_query(L.points_next, L.seg_next, L.cap_next, L.seg_off, L.cap, n_clouds, L.radius, L.K, order, grid_ws, gb,
                                    L.pool_idx, nullptr, nullptr, stream);                                                     // :499 / :376
            if (rc != RG_OK) return rc;
        }
    }
    return RG_OK;
}

}  // extern "C"
