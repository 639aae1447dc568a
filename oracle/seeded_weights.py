"""TEST INFRASTRUCTURE ONLY -- deterministic, portable RegTR weights.

The reference ships no checkpoints (trained_models/Readme.txt) and a seeded random-init of the
reference module cannot travel to the GPU box (47 MB).  Parity runs therefore load the SAME
name->tensor dictionary, generated here from a numpy Generator, into the reference module (golden
generation, this container) and into regtr_amd.RegTR (tests).  Names and shapes follow the
reference state_dict (checked against the real module in
oracle/make_golden.py).
"""
import math

import numpy as np
import torch

# 15-point 'center' disposition, unit radius: values of src/kernels/dispositions/k_015_center_3D.ply
# are loaded by the product (regtr_amd.kernel_points); here kernel points are drawn the same way the
# reference does (kernel_points.py:434-461): z-rotation + N(0, 0.01) noise, scaled by the conv radius.


def param_shapes(cfg):
    """Ordered {name: shape} of the reference RegTR state_dict for `cfg`."""
    shapes = {}
    in_dim, out_dim = cfg['in_feats_dim'], cfg['first_feats_dim']
    for bi, block in enumerate(cfg['architecture']):
        p = f'kpf_encoder.encoder_blocks.{bi}.'
        if block.startswith('simple'):                       # kpconv_blocks.py:614-624
            shapes[p + 'KPConv.weights'] = (cfg['num_kernel_points'], in_dim, out_dim // 2)
            shapes[p + 'KPConv.kernel_points'] = (cfg['num_kernel_points'], 3)
            in_dim = out_dim // 2
        else:                                                # kpconv_blocks.py:673-699
            mid = out_dim // 4
            if in_dim != mid:
                shapes[p + 'unary1.mlp.weight'] = (mid, in_dim)
            shapes[p + 'KPConv.weights'] = (cfg['num_kernel_points'], mid, mid)
            shapes[p + 'KPConv.kernel_points'] = (cfg['num_kernel_points'], 3)
            shapes[p + 'unary2.mlp.weight'] = (out_dim, mid)
            if in_dim != out_dim:
                shapes[p + 'unary_shortcut.mlp.weight'] = (out_dim, in_dim)
            in_dim = out_dim
        if 'strided' in block or 'pool' in block:
            out_dim *= 2
    D, FF = cfg['d_embed'], cfg['d_feedforward']
    shapes['feat_proj.weight'] = (D, in_dim)
    shapes['feat_proj.bias'] = (D,)
    for l in range(cfg['num_encoder_layers']):
        p = f'transformer_encoder.layers.{l}.'
        for a in ('self_attn', 'multihead_attn'):
            shapes[p + a + '.in_proj_weight'] = (3 * D, D)
            shapes[p + a + '.in_proj_bias'] = (3 * D,)
            shapes[p + a + '.out_proj.weight'] = (D, D)
            shapes[p + a + '.out_proj.bias'] = (D,)
        shapes[p + 'linear1.weight'] = (FF, D); shapes[p + 'linear1.bias'] = (FF,)
        shapes[p + 'linear2.weight'] = (D, FF); shapes[p + 'linear2.bias'] = (D,)
        for n in ('norm1', 'norm2', 'norm3'):
            shapes[p + n + '.weight'] = (D,); shapes[p + n + '.bias'] = (D,)
    if cfg['pre_norm']:                                          # encoder_norm only with pre_norm (regtr.py:60)
        shapes['transformer_encoder.norm.weight'] = (D,)
        shapes['transformer_encoder.norm.bias'] = (D,)
    q = 'correspondence_decoder.'
    if cfg.get('direct_regress_coor', False):                    # CorrespondenceRegressor (regtr.py:399-443)
        shapes[q + 'coor_mlp.0.weight'] = (D, D); shapes[q + 'coor_mlp.0.bias'] = (D,)
        shapes[q + 'coor_mlp.2.weight'] = (D, D); shapes[q + 'coor_mlp.2.bias'] = (D,)
        shapes[q + 'coor_mlp.4.weight'] = (3, D); shapes[q + 'coor_mlp.4.bias'] = (3,)
    else:                                                        # CorrespondenceDecoder (regtr.py:299-314)
        shapes[q + 'q_norm.weight'] = (D,); shapes[q + 'q_norm.bias'] = (D,)
        shapes[q + 'q_proj.weight'] = (D, D); shapes[q + 'q_proj.bias'] = (D,)
        shapes[q + 'k_proj.weight'] = (D, D); shapes[q + 'k_proj.bias'] = (D,)
    shapes[q + 'conf_logits_decoder.weight'] = (1, D); shapes[q + 'conf_logits_decoder.bias'] = (1,)
    if cfg.get('feature_loss_type', 'infonce') == 'infonce':     # feature_loss.py:261 (training-only params)
        shapes['feature_criterion.W'] = (D, D)
        shapes['feature_criterion_un.W'] = (D, D)
    return shapes


def seeded_state_dict(cfg, seed=0, kernel_dispositions=None):
    """kernel_dispositions: (15,3) unit-radius disposition (regtr_amd.kernel_points.K015_CENTER)."""
    rng = np.random.default_rng(seed)
    sd = {}
    r = cfg['first_subsampling_dl'] * cfg['conv_radius']
    radius_of_block = {}
    for bi, block in enumerate(cfg['architecture']):
        radius_of_block[bi] = r
        if 'strided' in block or 'pool' in block:
            r *= 2
    for name, shape in param_shapes(cfg).items():
        if name.endswith('kernel_points'):
            bi = int(name.split('.')[2])
            kp = np.array(kernel_dispositions, dtype=np.float64)
            th = rng.random() * 2 * np.pi
            c, s = np.cos(th), np.sin(th)
            R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float32)
            kp = kp + rng.normal(scale=0.01, size=kp.shape)
            v = np.matmul(radius_of_block[bi] * kp, R).astype(np.float32)
        elif 'norm' in name and name.endswith('weight'):
            v = (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif name.endswith('bias'):
            v = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif name.endswith('KPConv.weights'):
            bound = 1.0 / math.sqrt(shape[1] * shape[2] / shape[1]) if shape[1] > 0 else 1.0
            bound = 1.0 / math.sqrt(max(shape[2], 1))           # kaiming_uniform(a=sqrt5) on (K,Cin,Cout)
            v = rng.uniform(-bound, bound, shape).astype(np.float32)
        elif name == 'correspondence_decoder.coor_mlp.4.weight':
            # larger than default init so that predicted correspondences spread over metres and the
            # Procrustes covariance is well conditioned (random-init heads collapse to a point).
            v = rng.uniform(-0.5, 0.5, shape).astype(np.float32)
        else:
            bound = 1.0 / math.sqrt(shape[-1])
            v = rng.uniform(-bound, bound, shape).astype(np.float32)
        sd[name] = torch.from_numpy(v)
    return sd


def trained_like(sd, seed=1, linear_gain=4.0, gain_sigma=0.5):
    """A checkpoint-like perturbation of a seeded state_dict (random-init magnitudes are far tamer than trained ones): every dense
    weight of the cross-encoder times `linear_gain` (sharper softmax, larger FFN activations), per-channel log-normal gains
    (sigma `gain_sigma`) and N(0, 0.3) offsets on every LayerNorm, per-input-channel log-normal gains on the encoder's unary and
    KPConv weights (InstanceNorm removes a per-output-channel scale, not a spread over the inputs it mixes), feat_proj times 2.  The
    correspondence head keeps its scale so that outputs stay metre-sized and the 1e-4 absolute bar keeps its meaning."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, v in sd.items():
        v = v.clone()
        if name.endswith('kernel_points') or name.startswith('feature_criterion'):
            pass
        elif 'norm' in name:
            if name.endswith('weight'):
                v = v * torch.from_numpy(np.exp(gain_sigma * rng.standard_normal(v.shape)).astype(np.float32))
            else:
                v = v + torch.from_numpy((0.3 * rng.standard_normal(v.shape)).astype(np.float32))
        elif name.startswith('transformer_encoder.layers') and name.endswith('weight'):
            v = v * linear_gain
        elif name.startswith('kpf_encoder') and (name.endswith('mlp.weight') or name.endswith('KPConv.weights')):
            # gains over the INPUT channels (the following InstanceNorm removes any per-output-channel scale)
            g = torch.from_numpy(np.exp(gain_sigma * rng.standard_normal(v.shape[1])).astype(np.float32))
            v = v * (g[None, :, None] if name.endswith('KPConv.weights') else g[None, :])
        elif name == 'feat_proj.weight':
            v = v * 2.0
        out[name] = v
    return out
