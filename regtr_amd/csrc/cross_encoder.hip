// The whole cross-encoder stack (pre-norm layers, /root/reference/src/models/transformer/transformers.py:183-244 forward_pre,
// :37-59 TransformerCrossEncoder.forward) enqueued by ONE call: 12 launches per layer through the library's own entry points
// (regtr_layernorm, regtr_gemm_x3, regtr_mha_fwd), intermediates carved from one caller-provided workspace.
//
// Why this exists: at one pair per forward the step is ~280 launches of 5-50 us and the HOST is the bound (3.1 ms wall, GPU busy
// 2.2 ms; tools/host_profile.py): every launch issued from Python costs ~8 us of interpreter, ctypes marshalling, argument checking
// and torch.empty, of which the hipLaunchKernel itself is ~2.5.  The six layers are 72 of those launches with shapes that depend
// on nothing but the token count, so they are sequenced here, in C.  Nothing is computed differently: the same kernels with the
// same arguments in the same order as regtr_amd/transformer.py issues them one by one (tests/test_gpu_ops.py compares the two).
#include <math.h>

#include "common.h"

namespace {

// per-layer parameter table (device pointers), the order regtr_amd/transformer.py packs it in
enum {
    CE_N1_G, CE_N1_B, CE_SA_IN_W, CE_SA_IN_B, CE_SA_OUT_W, CE_SA_OUT_B,
    CE_N2_G, CE_N2_B, CE_CA_IN_W, CE_CA_IN_B, CE_CA_OUT_W, CE_CA_OUT_B,
    CE_N3_G, CE_N3_B, CE_L1_W, CE_L1_B, CE_L2_W, CE_L2_B, CE_PER_LAYER
};

size_t ce_gemm_ws(int M, int D, int F)
{
    size_t w = regtr_gemm_x3_ws_bytes(M, 3 * D, D);
    const size_t a = regtr_gemm_x3_ws_bytes(M, D, D), b = regtr_gemm_x3_ws_bytes(M, F, D), c = regtr_gemm_x3_ws_bytes(M, D, F);
    if (a > w) w = a;
    if (b > w) w = b;
    if (c > w) w = c;
    return w;
}

}  // namespace

extern "C" {

int regtr_cross_encoder_per_layer_params(void) { return CE_PER_LAYER; }

int regtr_cross_encoder_supported(int n_tok, int d_model, int d_ff, int n_heads)
{
    if (n_tok < 1 || d_model < 64 || d_ff < 64 || n_heads < 1 || d_model % n_heads || d_model / n_heads != 32) return 0;
    return (regtr_gemm_x3_preferred(n_tok, 3 * d_model, d_model) && regtr_gemm_x3_preferred(n_tok, d_model, d_model) &&
            regtr_gemm_x3_preferred(n_tok, d_ff, d_model) && regtr_gemm_x3_preferred(n_tok, d_model, d_ff)) ? 1 : 0;
}

size_t regtr_cross_encoder_ws_bytes(int n_tok, int d_model, int d_ff)
{
    const size_t row = (size_t)n_tok * sizeof(float);
    // x2p | qkv | att | xa | xb | h, each 256-byte aligned, + the split-K workspace of the widest GEMM
    return 6 * 256 + row * ((size_t)d_model * 4 + 3 * (size_t)d_model + d_ff) + rg_align_up(ce_gemm_ws(n_tok, d_model, d_ff), 256) + 256;
}

// x [n_tok, D] packed tokens of all clouds (not modified) -> outs [n_out, n_tok, D]: the final-norm'd output of every layer
// (return_intermediate = 1, n_out = n_layers) or of the last one (0, n_out = 1); final_gamma NULL = no final norm (plain copy).
//   layer_params   HOST array of n_layers * regtr_cross_encoder_per_layer_params() DEVICE pointers: LayerNorm gamma / beta, the
//                  weight planes of regtr_gemm_split_weights for in_proj [3D, D], out_proj [D, D], linear1 [F, D], linear2 [D, F]
//                  and their float32 biases, in the order of the enum above
//   layer_eps      HOST array of 3 floats per layer (norm1, norm2, norm3)
//   pe             positional embedding [n_tok, D] added to the normalised tokens for q, k AND v (sa_val_has_pos_emb = ca_val_has_pos_emb
//                  = true, both shipped configs), or NULL
//   seg_off [n_clouds + 1], kv_self / kv_cross [n_clouds]: cloud c attends cloud kv_*[c];  gemm_planes / attn_precision as in
//   regtr_gemm_x3 / regtr_mha_fwd;  ws: regtr_cross_encoder_ws_bytes(n_tok, D, d_ff) bytes
int regtr_cross_encoder_fwd(const float* x, int n_tok, int d_model, int d_ff, int n_heads, int n_layers,
                            const void* const* layer_params, const float* layer_eps, const float* final_gamma,
                            const float* final_beta, float final_eps, int return_intermediate, const float* pe,
                            const int* seg_off, const int* kv_self, const int* kv_cross, int n_clouds, int max_len,
                            int gemm_planes, int attn_precision, void* ws, size_t ws_bytes, float* outs, int* status, void* stream)
{
    if (!x || !layer_params || !layer_eps || !seg_off || !kv_self || !kv_cross || !outs || n_layers < 1 || n_clouds < 1) return RG_ERR_ARG;
    if (!regtr_cross_encoder_supported(n_tok, d_model, d_ff, n_heads)) return RG_ERR_ARG;
    if (!ws || ws_bytes < regtr_cross_encoder_ws_bytes(n_tok, d_model, d_ff)) return RG_ERR_WORKSPACE;
    const int M = n_tok, D = d_model, F = d_ff, hd = D / n_heads;
    RgCarver cv(ws, ws_bytes);
    float* x2p = cv.take<float>((size_t)M * D);
    float* qkv = cv.take<float>((size_t)M * 3 * D);
    float* att = cv.take<float>((size_t)M * D);
    float* xa = cv.take<float>((size_t)M * D);
    float* xb = cv.take<float>((size_t)M * D);
    float* h = cv.take<float>((size_t)M * F);
    const size_t gws_bytes = ce_gemm_ws(M, D, F);
    void* gws = gws_bytes ? (void*)cv.take<char>(gws_bytes) : nullptr;
    if (!cv.ok()) return RG_ERR_WORKSPACE;
    const float scale = (float)(1.0 / sqrt((double)hd));

#define CE_TRY(call) do { const int st_ = (call); if (st_ != RG_OK) return st_; } while (0)
    auto gemm = [&](const float* A, int K, const void* W, const void* bias, float* C, int N, const float* residual, int relu) {
        return regtr_gemm_x3(A, K, W, C, N, M, N, K, (const float*)bias, nullptr, residual, residual ? D : 0, relu, nullptr, nullptr, 0,
                             0.1f, gws, gws_bytes, nullptr, nullptr, 0, gemm_planes, nullptr, status, stream);
    };
    auto attention = [&](const float* xin, const void* const* P, int g, int in_w, float eps, const int* kv_of, float* xout) {
        // xout = xin + out_proj( MHA(q = k = v = LN(xin) + pe) )          transformers.py:194-229
        int st = regtr_layernorm(xin, M, D, (const float*)P[g], (const float*)P[g + 1], eps, pe, x2p, nullptr, stream);
        if (st != RG_OK) return st;
        st = gemm(x2p, D, P[in_w], P[in_w + 1], qkv, 3 * D, nullptr, 0);
        if (st != RG_OK) return st;
        st = regtr_mha_fwd(qkv, 3 * D, qkv + D, 3 * D, qkv + 2 * D, 3 * D, att, D, seg_off, kv_of, n_clouds, max_len, n_heads, hd, scale,
                           attn_precision, status, stream);
        if (st != RG_OK) return st;
        return gemm(att, D, P[in_w + 2], P[in_w + 3], xout, D, xin, 0);
    };

    const float* cur = x;
    for (int l = 0; l < n_layers; l++) {
        const void* const* P = layer_params + (size_t)l * CE_PER_LAYER;
        const float* eps = layer_eps + 3 * l;
        float* t1 = (cur == xa) ? xb : xa;                        // never the buffer `cur` lives in; x itself is left alone
        float* t2 = (t1 == xa) ? xb : xa;                         // (from the second layer on this IS the layer input: dead once the first
                                                                   //  attention block has consumed it as its residual)
        CE_TRY(attention(cur, P, CE_N1_G, CE_SA_IN_W, eps[0], kv_self, t1));
        CE_TRY(attention(t1, P, CE_N2_G, CE_CA_IN_W, eps[1], kv_cross, t2));
        CE_TRY(regtr_layernorm(t2, M, D, (const float*)P[CE_N3_G], (const float*)P[CE_N3_B], eps[2], nullptr, x2p, nullptr, stream));   // :232
        CE_TRY(gemm(x2p, D, P[CE_L1_W], P[CE_L1_B], h, F, nullptr, 1));
        CE_TRY(gemm(h, F, P[CE_L2_W], P[CE_L2_B], t1, D, t2, 0));                                                                        // :233-238
        cur = t1;
        if (return_intermediate || l == n_layers - 1) {
            float* o = outs + (return_intermediate ? (size_t)l * M * D : 0);
            if (final_gamma) CE_TRY(regtr_layernorm(cur, M, D, final_gamma, final_beta, final_eps, nullptr, o, nullptr, stream));
            else if (hipMemcpyAsync(o, cur, (size_t)M * D * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
                return RG_ERR_LAUNCH;
        }
    }
#undef CE_TRY
    return RG_OK;
}

}  // extern "C"
