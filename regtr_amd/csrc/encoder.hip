// The KPConv encoder's blocks (/root/reference/src/models/backbone_kpconv/kpconv.py:81-88 KPFEncoder.forward over kpconv_blocks.py:632-646
// SimpleBlock.forward and :706-741 ResnetBottleneckBlock.forward) enqueued by ONE call, for the SMALL-batch regime -- a pair or two per forward,
// the reference's own operating mode (conf/3dmatch.yaml:11 test_batch_size 1, trainer.py:202-206).
//
// Why: at one pair per forward the encoder is ~130 launches of 4-45 us (1.2 ms of GPU time) and issuing them from Python -- interpreter,
// ctypes marshalling, argument checks, a torch.empty per intermediate -- costs 2.2 ms of host time: the HOST is the bound
// (profiles/r04_x_forward_trace_pairs1.md, r04_x_host_profile_pairs1.txt).  Here the same launches are issued back to back from C, with
// every intermediate carved from one caller-provided workspace.  NOTHING is computed differently: each block calls the library's own entry
// points (regtr_kpconv_gather, regtr_gemm_x3 / _f32, regtr_instnorm_*, regtr_maxpool_gather) with the arguments, in the order, and under the
// routing rules of regtr_amd/kpconv.py + regtr_amd/ops.py for this regime, so the outputs are bit-identical to the op-by-op path
// (tests/test_gpu_model.py::test_one_call_encoder_equals_op_by_op).
//
// The regime: the caller's SMALL-batch regime (regtr_amd/ops.py SMALL_REGIME_ROWS: fewer than 131072 level-0 rows), where none of the
// large-batch forms applies (packed support records / pre-normalised gather, one-shot strip GEMM, block tails from input moments,
// first-block tail); the in-place normalisation before unary2 (>= 8192 rows AND > 64 channels) is implemented.  So the routing is:
//   Linear            regtr_gemm_x3 (tiled split GEMM) when regtr_gemm_x3_preferred and (N >= 64 or no folded operand), else regtr_gemm_f32;
//                     InstanceNorm statistics of the result from the GEMM epilogue (regtr_instnorm_finalize_tiles) when the launch has a
//                     statistics tile, else regtr_instnorm_stats on the result
//   KPConv            regtr_kpconv_gather (InstanceNorm + LeakyReLU of unary1 folded into the gathered rows) + the contraction as a Linear / count
//   block tail        regtr_instnorm_apply (normalise [+ normalised shortcut] + LeakyReLU)
#include "common.h"

namespace {

constexpr int ENC_SMALL_ROWS = 1 << 22;     // a guard only: the CALLER keeps this path to its small-batch regime (regtr_amd/ops.py SMALL_REGIME_ROWS)
constexpr int ENC_PREAPPLY_ROWS = 8192;     // regtr_amd/ops.py PREAPPLY_MIN_ROWS

struct Enc {
    const regtr_encoder_block_t* blocks;
    const regtr_encoder_level_t* levels;
    int n_levels, n_clouds, f16_pair;
    float slope, eps;
    int* status;
    hipStream_t st;
    bool dry;
    // workspace: [feature arena A | feature arena B | row-tile tables of the call | per-block scratch]; the dry pass measures the sizes
    char* base;
    size_t feat_max, tiles_bytes, scratch_max;      // capacities (dry pass: running maxima / totals)
    size_t tiles_off, scratch_off;
    // per-(level, tile height) row-tile cloud tables (regtr_tile_segments), made once per call like ops.tile_segments caches them
    struct TileKey { int level, rows; void* ptr; } tiles[32];
    int n_tiles;
    bool overflow;

    template <typename T>
    T* scratch(size_t n)
    {
        scratch_off = rg_align_up(scratch_off, 256);
        T* p = dry ? nullptr : (T*)(base + 2 * feat_max + tiles_bytes + scratch_off);
        scratch_off += n * sizeof(T);
        if (dry) { if (scratch_off > scratch_max) scratch_max = scratch_off; }
        else if (scratch_off > scratch_max) overflow = true;
        return p;
    }

    int tile_table(int level, int M, int rows, const void** out)
    {
        for (int i = 0; i < n_tiles; i++)
            if (tiles[i].level == level && tiles[i].rows == rows) { *out = tiles[i].ptr; return RG_OK; }
        if (n_tiles >= 32) return RG_ERR_ARG;
        tiles_off = rg_align_up(tiles_off, 256);
        void* p = dry ? (void*)16 : (void*)(base + 2 * feat_max + tiles_off);       // (dry: any non-null marker)
        tiles_off += (size_t)rg_cdiv(M, rows) * 16;
        if (dry) tiles_bytes = rg_align_up(tiles_off, 256);
        else if (tiles_off > tiles_bytes) { overflow = true; return RG_ERR_WORKSPACE; }
        tiles[n_tiles++] = {level, rows, p};
        *out = p;
        if (dry) return RG_OK;
        return regtr_tile_segments(levels[level].seg_off, n_clouds, M, rows, p, st);
    }

    // ops.gemm for this regime: C = A' W [/ row_div] (+ InstanceNorm statistics of C for the rows of `stat_level`).  a_stats: A' =
    // LeakyReLU(InstanceNorm(A)) with the clouds of `a_level`.  -> *stats_out [n_clouds, N, 2] when stat_level >= 0.
    int linear(const float* A, int M, const regtr_weight_t& W, float* C, const float* row_div, const float* a_stats, int a_level,
               int stat_level, float** stats_out)
    {
        const int N = W.N, K = W.K;
        const bool want = stat_level >= 0;
        const int x3_ok = (W.planes && regtr_gemm_x3_supported(M, N, K)) ? 1 : 0;
        const bool use_x3 = x3_ok && (K % 4 == 0) && (N >= 64 || !a_stats) && regtr_gemm_x3_preferred(M, N, K);
        const int* a_seg = a_stats ? levels[a_level].seg_off : nullptr;
        const int n_seg = a_stats ? n_clouds : 0;
        float* stats = nullptr;
        if (want) stats = scratch<float>((size_t)n_clouds * N * 2);
        if (stats_out) *stats_out = stats;
        if (use_x3) {
            const size_t nb = regtr_gemm_x3_ws_bytes(M, N, K);
            void* ws = nb ? (void*)scratch<char>(nb) : nullptr;
            const int R = want ? regtr_gemm_x3_stat_tile_rows(M, N, K) : 0;
            const int rows = regtr_gemm_x3_tile_rows(M, N, K);
            double* partial = nullptr;
            const int* s_off = nullptr;
            if (R) {
                s_off = levels[stat_level].seg_off;
                partial = scratch<double>(((size_t)rg_cdiv(M, R) + n_clouds) * N * 2);
            }
            const int seg_level = R ? stat_level : (a_stats ? a_level : -1);
            const void* ti = nullptr;
            if (seg_level >= 0 && (!a_stats || !R || a_level == stat_level)) {
                const int rc = tile_table(seg_level, M, rows, &ti);
                if (rc != RG_OK) return rc;
            }
            const void* pl = W.planes;
            int npl = 3;
            if (f16_pair && W.planes16 && regtr_gemm_x3_f16_supported(M, N, K, (R > 0 || a_stats || ti) ? 1 : 0)) { pl = W.planes16; npl = 4; }
            if (!dry) {
                if (overflow) return RG_ERR_WORKSPACE;
                const int rc = regtr_gemm_x3(A, K, pl, C, N, M, N, K, nullptr, row_div, nullptr, 0, 0, a_stats, a_seg, n_seg, slope, ws, nb, partial,
                                             s_off, R ? n_clouds : 0, npl, ti, status, st);
                if (rc != RG_OK) return rc;
                if (R) return regtr_instnorm_finalize_tiles(partial, s_off, n_clouds, N, R, eps, stats, st);
            } else if (R) {
                return RG_OK;
            }
        } else {
            const size_t nb = regtr_gemm_f32_ws_bytes(M, N, K);
            void* ws = nb ? (void*)scratch<char>(nb) : nullptr;
            if (!dry) {
                if (overflow) return RG_ERR_WORKSPACE;
                const int rc = regtr_gemm_f32(A, K, W.kn, N, C, N, M, N, K, nullptr, row_div, nullptr, 0, 0, a_stats, a_seg, n_seg, slope, ws, nb, st);
                if (rc != RG_OK) return rc;
            }
        }
        if (want) {        // the launch had no statistics epilogue: a pass over the result (ops.instnorm_stats)
            const regtr_encoder_level_t& L = levels[stat_level];
            const size_t nb = regtr_instnorm_ws_bytes(n_clouds, L.max_len, N);
            void* ws = scratch<char>(nb);
            if (dry) return RG_OK;
            if (overflow) return RG_ERR_WORKSPACE;
            return regtr_instnorm_stats(C, L.seg_off, n_clouds, L.max_len, N, eps, stats, ws, nb, st);
        }
        return RG_OK;
    }

    // ops.kpconv: gather (+ folded InstanceNorm / LeakyReLU of the support features) and the kernel-point contraction / neighbour count
    int kpconv(const regtr_encoder_block_t& b, const float* x, const float* x_stats, float* out, float** stats_out)
    {
        const regtr_encoder_level_t& S = levels[b.layer];
        const int lq = b.layer + (b.strided ? 1 : 0);
        const regtr_encoder_level_t& Q = levels[lq];
        const int* nbr = b.strided ? S.pool_idx : S.conv_idx;
        const int nq = Q.n, ns = S.n, H = S.K, Cin = b.conv.K / b.n_kp;
        float* wf = scratch<float>((size_t)nq * b.n_kp * Cin);
        float* num = scratch<float>((size_t)nq);
        if (!dry) {
            if (overflow) return RG_ERR_WORKSPACE;
            const int rc = regtr_kpconv_gather(Q.points, nq, S.points, ns, nbr, H, x, Cin, nullptr, nullptr, b.kernel_points, b.n_kp, b.extent,
                                               x_stats, x_stats ? Q.seg_off : nullptr, x_stats ? n_clouds : 0, slope, wf, 0, num, st);
            if (rc != RG_OK) return rc;
        }
        return linear(wf, nq, b.conv, out, num, nullptr, -1, lq, stats_out);
    }

    int run(const float* x_in, int first, int last, float* out)
    {
        const float* cur = x_in;
        int flip = 0;
        n_tiles = 0;
        tiles_off = 0;
        overflow = false;
        for (int bi = first; bi < last; bi++) {
            const regtr_encoder_block_t& b = blocks[bi];
            const int lq = b.layer + (b.strided ? 1 : 0);
            const regtr_encoder_level_t& S = levels[b.layer];
            const regtr_encoder_level_t& Q = levels[lq];
            const int out_dim = b.kind == 0 ? b.conv.N : b.unary2.N;
            const size_t out_bytes = (size_t)Q.n * out_dim * sizeof(float);
            if (dry) { if (out_bytes > feat_max) feat_max = out_bytes; }
            else if (out_bytes > feat_max) return RG_ERR_WORKSPACE;
            float* y = (bi == last - 1) ? out : (dry ? nullptr : (float*)(base + (size_t)flip * feat_max));
            scratch_off = 0;
            int rc;
            if (b.kind == 0) {
                // SimpleBlock (kpconv_blocks.py:632-646): KPConv -> InstanceNorm -> LeakyReLU
                float* stt = nullptr;
                rc = kpconv(b, cur, nullptr, y, &stt);
                if (rc != RG_OK) return rc;
                if (!dry) {
                    rc = regtr_instnorm_apply(y, Q.seg_off, n_clouds, Q.max_len, out_dim, stt, nullptr, nullptr, 1, slope, y, nullptr, nullptr, st);
                    if (rc != RG_OK) return rc;
                }
            } else {
                // ResnetBottleneckBlock (kpconv_blocks.py:706-741)
                const int mid = b.conv.N, in_dim = b.unary1.N ? b.unary1.K : b.conv.K / b.n_kp;
                const float* x1 = cur;
                float* x1_stats = nullptr;
                if (b.unary1.N) {                                                       // :722  unary1 = Linear [-> IN -> LReLU folded into the gather]
                    float* t = scratch<float>((size_t)S.n * b.unary1.N);
                    rc = linear(cur, S.n, b.unary1, t, nullptr, nullptr, -1, b.layer, &x1_stats);
                    if (rc != RG_OK) return rc;
                    x1 = t;
                }
                float* conv = scratch<float>((size_t)Q.n * mid);
                float* conv_stats = nullptr;
                rc = kpconv(b, x1, x1_stats, conv, &conv_stats);                         // :726
                if (rc != RG_OK) return rc;
                const float* shortcut = cur;                                             // :734-737
                if (b.strided) {
                    float* mp = scratch<float>((size_t)Q.n * in_dim);
                    if (!dry) {
                        if (overflow) return RG_ERR_WORKSPACE;
                        rc = regtr_maxpool_gather(cur, S.n, in_dim, S.pool_idx, S.K, Q.n, S.pool_width, mp, st);
                        if (rc != RG_OK) return rc;
                    }
                    shortcut = mp;
                }
                float* sc_stats = nullptr;
                if (b.shortcut.N) {
                    float* sc = scratch<float>((size_t)Q.n * b.shortcut.N);
                    rc = linear(shortcut, Q.n, b.shortcut, sc, nullptr, nullptr, -1, lq, &sc_stats);
                    if (rc != RG_OK) return rc;
                    shortcut = sc;
                }
                // :727-730  IN + LReLU of the conv output: folded into unary2's operand -- or, where the fold would route a wide product to
                // the tiled kernel (ops.preapply_unary2: >= 8192 rows and more than 64 channels), applied in place first
                if (conv_stats && Q.n >= ENC_PREAPPLY_ROWS && mid > 64) {
                    if (!dry) {
                        rc = regtr_instnorm_apply(conv, Q.seg_off, n_clouds, Q.max_len, mid, conv_stats, nullptr, nullptr, 1, slope, conv, nullptr, nullptr, st);
                        if (rc != RG_OK) return rc;
                    }
                    conv_stats = nullptr;
                }
                float* y_stats = nullptr;
                rc = linear(conv, Q.n, b.unary2, y, nullptr, conv_stats, lq, lq, &y_stats);
                if (rc != RG_OK) return rc;
                if (!dry) {                                                              // :741  LeakyReLU(IN(unary2) + [IN](shortcut))
                    rc = regtr_instnorm_apply(y, Q.seg_off, n_clouds, Q.max_len, out_dim, y_stats, shortcut, sc_stats, 1, slope, y, nullptr, nullptr, st);
                    if (rc != RG_OK) return rc;
                }
            }
            cur = y;
            flip ^= 1;
        }
        return RG_OK;
    }
};

int enc_check(const regtr_encoder_block_t* blocks, int n_blocks, const regtr_encoder_level_t* levels, int n_levels, int n_clouds, int first, int last)
{
    if (!blocks || !levels || n_levels < 1 || n_clouds < 1 || first < 0 || last > n_blocks || first >= last) return 0;
    if (levels[0].n >= ENC_SMALL_ROWS) return 0;
    for (int l = 0; l < n_levels; l++)
        if (levels[l].n < 1 || levels[l].max_len < 1 || levels[l].K < 1 || levels[l].K > 64 || !levels[l].points || !levels[l].seg_off) return 0;
    for (int bi = first; bi < last; bi++) {
        const regtr_encoder_block_t& b = blocks[bi];
        if (b.kind < 0 || b.kind > 1 || b.layer < 0 || b.layer + (b.strided ? 1 : 0) >= n_levels || b.n_kp < 1 || b.n_kp > 16 || !b.kernel_points) return 0;
        if (!b.conv.kn || b.conv.K % b.n_kp) return 0;
        const int Cin = b.conv.K / b.n_kp;
        if (!regtr_kpconv_gather_computes_flag(Cin, levels[b.layer].K) || (long long)levels[b.layer].n * Cin >= (1LL << 29)) return 0;
        if (!(b.strided ? levels[b.layer].pool_idx : levels[b.layer].conv_idx)) return 0;
        if (b.kind == 1 && (!b.unary2.kn || b.unary2.K != b.conv.N)) return 0;
        if (b.kind == 1 && b.unary1.N && (b.unary1.N != Cin || !b.unary1.kn)) return 0;
        if (b.kind == 1 && b.shortcut.N && (b.shortcut.N != b.unary2.N || !b.shortcut.kn)) return 0;
    }
    return 1;
}

}  // namespace

extern "C" {

int regtr_encoder_supported(const regtr_encoder_block_t* blocks, int n_blocks, const regtr_encoder_level_t* levels, int n_levels, int n_clouds,
                            int first, int last)
{
    return enc_check(blocks, n_blocks, levels, n_levels, n_clouds, first, last);
}

static int enc_plan(Enc& e, const regtr_encoder_block_t* blocks, const regtr_encoder_level_t* levels, int n_levels, int n_clouds, int first,
                    int last, int f16_pair)
{
    e.blocks = blocks; e.levels = levels; e.n_levels = n_levels; e.n_clouds = n_clouds; e.f16_pair = f16_pair;
    e.slope = 0.1f; e.eps = 1e-5f; e.status = nullptr; e.st = nullptr; e.dry = true; e.base = nullptr;
    e.feat_max = 0; e.scratch_max = 0; e.scratch_off = 0; e.tiles_bytes = 0;
    const int rc = e.run(nullptr, first, last, nullptr);
    e.feat_max = rg_align_up(e.feat_max, 256);
    e.scratch_max = rg_align_up(e.scratch_max, 256);
    return rc;
}

size_t regtr_encoder_ws_bytes(const regtr_encoder_block_t* blocks, int n_blocks, const regtr_encoder_level_t* levels, int n_levels, int n_clouds,
                              int first, int last, int f16_pair)
{
    if (!enc_check(blocks, n_blocks, levels, n_levels, n_clouds, first, last)) return 0;
    Enc e;
    if (enc_plan(e, blocks, levels, n_levels, n_clouds, first, last, f16_pair) != RG_OK) return 0;
    return 2 * e.feat_max + e.tiles_bytes + e.scratch_max + 256;
}

// Blocks [first, last) of the encoder.  x_in [levels[blocks[first].layer].n, in_dim]: the features entering block `first` (not modified);
// out [rows of block last - 1's level, its out_dim]: the features leaving block last - 1.  blocks / levels: HOST arrays (include/regtr_hip.h);
// f16_pair: float32-grade contractions in the f16 pair format where the weights allow (planes16 != NULL) and the kernel serves the shape,
// as cfg.compute_dtype 'fp32' runs them; status: the optional status word (REGTR_STATUS_F16_RANGE).
int regtr_encoder_fwd(const regtr_encoder_block_t* blocks, int n_blocks, const regtr_encoder_level_t* levels, int n_levels, int n_clouds,
                      int first, int last, const float* x_in, float* out, int f16_pair, float slope, float eps, void* ws, size_t ws_bytes,
                      int* status, void* stream)
{
    if (!x_in || !out || !ws || !enc_check(blocks, n_blocks, levels, n_levels, n_clouds, first, last)) return RG_ERR_ARG;
    if (((uintptr_t)x_in | (uintptr_t)out | (uintptr_t)ws) % 16) return RG_ERR_ARG;
    Enc e;
    int rc = enc_plan(e, blocks, levels, n_levels, n_clouds, first, last, f16_pair);
    if (rc != RG_OK) return rc;
    if (ws_bytes < 2 * e.feat_max + e.tiles_bytes + e.scratch_max) return RG_ERR_WORKSPACE;
    e.dry = false; e.base = (char*)ws; e.status = status; e.st = (hipStream_t)stream; e.slope = slope; e.eps = eps;
    return e.run(x_in, first, last, out);
}

}  // extern "C"
