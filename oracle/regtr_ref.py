"""TEST INFRASTRUCTURE ONLY -- plain-torch fp32 CPU restatement of the reference RegTR inference
hot path (floating-point part), function by function.  It is the parity checker for the HIP kernels
and the `cpu_baseline` ("port") leg of bench.py; the product never imports it.

Pinned against the reference's own modules (imported from /root/reference by oracle/ref_loader.py)
by oracle/make_golden.py -> tests/golden/*.npz and tests/test_oracle.py.

All citations are relative to /root/reference/src/.
The restatement works on PACKED token arrays with per-cloud lengths; the reference pads to
(N_max, B, D) and masks (utils/seq_manipulation.py:6-33) which is arithmetically identical because
masked keys receive -inf before the softmax (transformers.py:196-229) and padded rows are dropped
again by unpad_sequences (regtr.py:171-172).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import native


# ------------------------------------------------------------------------------------------------
# preprocessing pyramid  (models/backbone_kpconv/kpconv.py:298-414, CPU `Preprocessor`)
# ------------------------------------------------------------------------------------------------
def preprocess(pts_list, cfg, use_ref_cpp=False):
    """Returns the kpconv_meta dict (kpconv.py:406-412) in the CANONICAL row orders of
    oracle/regtr_oracle.cpp.  neighbour tables have fixed width K = neighborhood_limits[l]
    (the reference's CPU path emits min(max_count, K) columns, kpconv.py:255-258; the extra
    columns are shadow indices and do not change any downstream value).

    use_ref_cpp=True drives the unmodified reference C++ instead (its own row orders) -- used by
    bench.py's cpu_baseline when oracle/_ref is present."""
    arch = cfg['architecture']
    limits = cfg['neighborhood_limits']
    r_normal = cfg['first_subsampling_dl'] * cfg['conv_radius']            # kpconv.py:315
    pts = np.concatenate([np.asarray(p, dtype=np.float32) for p in pts_list], 0)
    lens = np.array([len(p) for p in pts_list], np.int32)
    out = {k: [] for k in ('points', 'neighbors', 'pools', 'upsamples', 'stack_lengths')}
    layer, layer_blocks = 0, []

    def radius(q, s, ql, sl, r, K):
        if cfg.get('kpconv_neighbor_order', 'nearest') == 'index':          # PreprocessorGPU's neighbour sets (kpconv.py:261-288)
            return ball_query_first_k(q, s, ql, sl, r, K)
        if use_ref_cpp:
            t = native.ref_batch_query(q, s, ql, sl, r)[:, :K]             # kpconv.py:254-256
            return t
        return native.radius_neighbors(q, s, ql, sl, r, K)[0]

    for bi, block in enumerate(arch):                                       # kpconv.py:328-404
        if 'global' in block or 'upsample' in block:
            break
        if not ('pool' in block or 'strided' in block):
            layer_blocks.append(block)
            if bi < len(arch) - 1 and 'upsample' not in arch[bi + 1]:
                continue
        K = limits[layer]
        if layer_blocks:
            conv_i = radius(pts, pts, lens, lens, r_normal, K)              # :349-351
        else:
            conv_i = np.zeros((0, 1), np.int32)
        if 'pool' in block or 'strided' in block:
            dl = 2 * r_normal / cfg['conv_radius']                          # :363
            if use_ref_cpp:
                pool_p, pool_b = native.ref_subsample_batch(pts, lens, dl)  # :366
            else:
                # cfg.kpconv_voxel_key: 'origin' = the CPU op (kpconv.py:170-181), 'floor' / 'floor_rcp' = PreprocessorGPU (kpconv.py:213-240)
                pool_p, pool_b = native.grid_subsample(pts, lens, dl, key_mode=native.VOXEL_KEY_MODES[cfg.get('kpconv_voxel_key', 'origin')])
            pool_i = radius(pool_p, pts, pool_b, lens, r_normal, K)         # :376
            # up_i (kpconv.py:380) is never read by RegTR (no decoder, kpconv.py:93-94): skipped.
        else:
            pool_i = np.zeros((0, 1), np.int32)
            pool_p = np.zeros((0, 3), np.float32)
            pool_b = np.zeros((0,), np.int32)
        out['points'].append(torch.from_numpy(pts))
        out['neighbors'].append(torch.from_numpy(conv_i.astype(np.int64)))
        out['pools'].append(torch.from_numpy(pool_i.astype(np.int64)))
        out['upsamples'].append(torch.zeros((0, 1), dtype=torch.int64))
        out['stack_lengths'].append(torch.from_numpy(lens.astype(np.int64)))
        pts, lens = pool_p, pool_b
        r_normal *= 2
        layer += 1
        layer_blocks = []
    return out


# ------------------------------------------------------------------------------------------------
# KPConv encoder  (models/backbone_kpconv/kpconv_blocks.py)
# ------------------------------------------------------------------------------------------------
def kpconv(q_pts, s_pts, neighb_inds, x, weights, kernel_points, KP_extent):
    """KPConv.forward, non-deformable / linear influence / sum aggregation
    (kpconv_blocks.py:269-414)."""
    s_pts = torch.cat((s_pts, torch.zeros_like(s_pts[:1]) + 1e6), 0)       # :309 shadow point
    neighbors = s_pts[neighb_inds] - q_pts.unsqueeze(1)                    # :312-315
    differences = neighbors.unsqueeze(2) - kernel_points                   # :325-326
    sq_distances = torch.sum(differences ** 2, dim=3)                      # :329
    all_weights = torch.clamp(1 - torch.sqrt(sq_distances) / KP_extent, min=0.0)   # :368
    all_weights = all_weights.transpose(1, 2)                              # :369  (Nq, 15, H)
    x = torch.cat((x, torch.zeros_like(x[:1])), 0)                         # :388 shadow feature
    neighb_x = x[neighb_inds]                                              # :391  (Nq, H, Cin)
    weighted = torch.matmul(all_weights, neighb_x)                         # :394  (Nq, 15, Cin)
    kernel_out = torch.matmul(weighted.permute(1, 0, 2), weights)          # :401-402 (15, Nq, Cout)
    out = kernel_out.sum(0)                                                # :406
    nsum = neighb_x.sum(-1)                                                # :409
    num = torch.gt(nsum, 0.0).sum(-1)                                      # :410
    num = torch.max(num, torch.ones_like(num))                             # :411
    return out / num.unsqueeze(1)                                          # :412


def instance_norm(x, lens, eps=1e-5):
    """BatchNormBlock with nn.InstanceNorm1d: per cloud, per channel, biased variance, no affine
    (kpconv_blocks.py:489,510-519)."""
    outs, o = [], 0
    for n in lens.tolist():
        seg = x[o:o + n]
        mu = seg.mean(0, keepdim=True)
        var = seg.var(0, unbiased=False, keepdim=True)
        outs.append((seg - mu) / torch.sqrt(var + eps))
        o += n
    return torch.cat(outs, 0)


def unary(x, w, lens, relu=True):
    """UnaryBlock: Linear(no bias) -> InstanceNorm -> LeakyReLU(0.1) (kpconv_blocks.py:556-561)."""
    x = instance_norm(x @ w.t(), lens)
    return F.leaky_relu(x, 0.1) if relu else x


def max_pool(x, inds):
    """kpconv_blocks.py:127-143 (zero shadow row participates in the max)."""
    x = torch.cat((x, torch.zeros_like(x[:1])), 0)
    return x[inds].max(1)[0]


def encoder(sd, cfg, meta, prefix='kpf_encoder.encoder_blocks.', collect=None):
    """KPFEncoder.forward (kpconv.py:81-88) over SimpleBlock / ResnetBottleneckBlock
    (kpconv_blocks.py:632-646, 706-741).  `collect`, if a list, receives every block output."""
    x = torch.ones_like(meta['points'][0][:, :1])                          # regtr.py:122
    r = cfg['first_subsampling_dl'] * cfg['conv_radius']                   # kpconv.py:28
    layer = 0
    for bi, block in enumerate(cfg['architecture']):
        if 'upsample' in block:
            break
        p = f'{prefix}{bi}.'
        extent = r * cfg['KP_extent'] / cfg['conv_radius']                 # kpconv_blocks.py:603,662
        strided = 'strided' in block
        q_pts = meta['points'][layer + 1] if strided else meta['points'][layer]
        s_pts = meta['points'][layer]
        inds = meta['pools'][layer] if strided else meta['neighbors'][layer]
        lens_pre = meta['stack_lengths'][layer]
        lens_post = meta['stack_lengths'][layer + 1] if strided else lens_pre
        if block.startswith('simple'):
            y = kpconv(q_pts, s_pts, inds, x, sd[p + 'KPConv.weights'], sd[p + 'KPConv.kernel_points'], extent)
            x = F.leaky_relu(instance_norm(y, lens_post), 0.1)             # :645-646
        else:
            feats = x
            if p + 'unary1.mlp.weight' in sd:                              # :722
                y = unary(feats, sd[p + 'unary1.mlp.weight'], lens_pre)
            else:
                y = feats
            y = kpconv(q_pts, s_pts, inds, y, sd[p + 'KPConv.weights'], sd[p + 'KPConv.kernel_points'], extent)
            y = F.leaky_relu(instance_norm(y, lens_post), 0.1)             # :727
            y = unary(y, sd[p + 'unary2.mlp.weight'], lens_post, relu=False)   # :730
            sc = max_pool(feats, inds) if strided else feats               # :734-737
            if p + 'unary_shortcut.mlp.weight' in sd:
                sc = unary(sc, sd[p + 'unary_shortcut.mlp.weight'], lens_post, relu=False)
            x = F.leaky_relu(y + sc, 0.1)                                  # :741
        if collect is not None:
            collect.append(x)
        if 'pool' in block or strided:
            layer += 1
            r *= 2
    return x


# ------------------------------------------------------------------------------------------------
# transformer  (models/transformer/*.py)
# ------------------------------------------------------------------------------------------------
def pos_embed_sine(xyz, d_model=256, scale=1.0, temperature=10000):
    """PositionEmbeddingCoordsSine.forward (position_embedding.py:29-50)."""
    n_dim = xyz.shape[-1]
    npf = d_model // n_dim // 2 * 2
    padding = d_model - npf * n_dim
    dim_t = torch.arange(npf, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode='trunc') / npf)
    pos = (xyz * (scale * 2 * math.pi)).unsqueeze(-1) / dim_t
    emb = torch.stack([pos[..., 0::2].sin(), pos[..., 1::2].cos()], dim=-1).reshape(*xyz.shape[:-1], -1)
    return F.pad(emb, (0, padding))


def mha(q_in, k_in, v_in, w_in, b_in, w_out, b_out, nhead):
    """nn.MultiheadAttention forward for one unpadded sequence (batch_first=False slow path:
    in_proj -> scaled QK^T -> softmax -> AV -> out_proj; torch/nn/functional.py
    multi_head_attention_forward).  q_in (Lq, E); k_in, v_in (Lk, E)."""
    E = q_in.shape[-1]
    hd = E // nhead
    q = q_in @ w_in[:E].t() + b_in[:E]
    k = k_in @ w_in[E:2 * E].t() + b_in[E:2 * E]
    v = v_in @ w_in[2 * E:].t() + b_in[2 * E:]
    q = q.view(-1, nhead, hd).transpose(0, 1) * (1.0 / math.sqrt(hd))
    k = k.view(-1, nhead, hd).transpose(0, 1)
    v = v.view(-1, nhead, hd).transpose(0, 1)
    attn = torch.softmax(q @ k.transpose(1, 2), dim=-1)
    o = (attn @ v).transpose(0, 1).reshape(-1, E)
    return o @ w_out.t() + b_out


def cross_encoder_layer_pre(sd, p, src, tgt, src_pe, tgt_pe, nhead, sa_val_pe=True, ca_val_pe=True):
    """TransformerCrossEncoderLayer.forward_pre for one pair (transformers.py:183-244)."""
    def ln(x, name):
        return F.layer_norm(x, (x.shape[-1],), sd[p + name + '.weight'], sd[p + name + '.bias'])

    def attn(name, q, k, v):
        return mha(q, k, v, sd[p + name + '.in_proj_weight'], sd[p + name + '.in_proj_bias'],
                   sd[p + name + '.out_proj.weight'], sd[p + name + '.out_proj.bias'], nhead)

    s2 = ln(src, 'norm1'); s2p = s2 + src_pe                               # :194-195
    src = src + attn('self_attn', s2p, s2p, s2p if sa_val_pe else s2)      # :197-201
    t2 = ln(tgt, 'norm1'); t2p = t2 + tgt_pe                               # :203-204
    tgt = tgt + attn('self_attn', t2p, t2p, t2p if sa_val_pe else t2)      # :206-210
    s2, t2 = ln(src, 'norm2'), ln(tgt, 'norm2')                            # :213
    sp, tp = s2 + src_pe, t2 + tgt_pe
    s3 = attn('multihead_attn', sp, tp, tp if ca_val_pe else t2)           # :217-221
    t3 = attn('multihead_attn', tp, sp, sp if ca_val_pe else s2)           # :222-226
    src, tgt = src + s3, tgt + t3                                          # :228-229

    def ffn(x):
        x2 = ln(x, 'norm3')
        x2 = F.relu(x2 @ sd[p + 'linear1.weight'].t() + sd[p + 'linear1.bias'])
        return x + (x2 @ sd[p + 'linear2.weight'].t() + sd[p + 'linear2.bias'])   # :232-238
    return ffn(src), ffn(tgt)


def cross_encoder_layer_post(sd, p, src, tgt, src_pe, tgt_pe, nhead, sa_val_pe=True, ca_val_pe=True):
    """TransformerCrossEncoderLayer.forward_post for one pair (transformers.py:121-181)."""
    def ln(x, name):
        return F.layer_norm(x, (x.shape[-1],), sd[p + name + '.weight'], sd[p + name + '.bias'])

    def attn(name, q, k, v):
        return mha(q, k, v, sd[p + name + '.in_proj_weight'], sd[p + name + '.in_proj_bias'],
                   sd[p + name + '.out_proj.weight'], sd[p + name + '.out_proj.bias'], nhead)

    sp = src + src_pe                                                      # :131-139
    src = ln(src + attn('self_attn', sp, sp, sp if sa_val_pe else src), 'norm1')
    tp = tgt + tgt_pe                                                      # :141-148
    tgt = ln(tgt + attn('self_attn', tp, tp, tp if sa_val_pe else tgt), 'norm1')
    sp, tp = src + src_pe, tgt + tgt_pe                                    # :151-152
    s2 = attn('multihead_attn', sp, tp, tp if ca_val_pe else tgt)          # :154-158
    t2 = attn('multihead_attn', tp, sp, sp if ca_val_pe else src)          # :159-163
    src, tgt = ln(src + s2, 'norm2'), ln(tgt + t2, 'norm2')                # :165-166

    def ffn(x):
        x2 = F.relu(x @ sd[p + 'linear1.weight'].t() + sd[p + 'linear1.bias'])
        return ln(x + (x2 @ sd[p + 'linear2.weight'].t() + sd[p + 'linear2.bias']), 'norm3')   # :169-175
    return ffn(src), ffn(tgt)


def transformer(sd, cfg, src, tgt, src_pe, tgt_pe, prefix='transformer_encoder.'):
    """TransformerCrossEncoder.forward with return_intermediate and final norm
    (transformers.py:27-59) -> (L, Ns, D), (L, Nt, D)."""
    so, to = [], []
    layer = cross_encoder_layer_pre if cfg['pre_norm'] else cross_encoder_layer_post     # transformers.py:255-258
    for l in range(cfg['num_encoder_layers']):
        src, tgt = layer(sd, f'{prefix}layers.{l}.', src, tgt, src_pe, tgt_pe, cfg['nhead'],
                         cfg['sa_val_has_pos_emb'], cfg['ca_val_has_pos_emb'])
        if prefix + 'norm.weight' in sd:                                                  # encoder norm only with pre_norm (regtr.py:60)
            nw, nb = sd[prefix + 'norm.weight'], sd[prefix + 'norm.bias']
            so.append(F.layer_norm(src, (src.shape[-1],), nw, nb))
            to.append(F.layer_norm(tgt, (tgt.shape[-1],), nw, nb))
        else:
            so.append(src); to.append(tgt)
    return torch.stack(so), torch.stack(to)


def correspondence_regressor(sd, f, prefix='correspondence_decoder.'):
    """CorrespondenceRegressor (regtr.py:432-436): coor_mlp 256->256->256->3 and conf logit."""
    h = F.relu(f @ sd[prefix + 'coor_mlp.0.weight'].t() + sd[prefix + 'coor_mlp.0.bias'])
    h = F.relu(h @ sd[prefix + 'coor_mlp.2.weight'].t() + sd[prefix + 'coor_mlp.2.bias'])
    corr = h @ sd[prefix + 'coor_mlp.4.weight'].t() + sd[prefix + 'coor_mlp.4.bias']
    logit = f @ sd[prefix + 'conf_logits_decoder.weight'].t() + sd[prefix + 'conf_logits_decoder.bias']
    return corr, logit


def correspondence_decoder_attn(sd, sc, tc, s_xyz, t_xyz, s_pe, t_pe, use_pe, prefix='correspondence_decoder.'):
    """CorrespondenceDecoder.forward / simple_attention for one pair (regtr.py:316-396): sc, tc (L, N, D)."""
    D = sc.shape[-1]

    def attend(qf, kf, xyz):
        q = (qf @ sd[prefix + 'q_proj.weight'].t() + sd[prefix + 'q_proj.bias']) / math.sqrt(D)          # :329
        k = kf @ sd[prefix + 'k_proj.weight'].t() + sd[prefix + 'k_proj.bias']                           # :330
        return torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ xyz                                      # :332-347
    s2, t2 = (sc + s_pe, tc + t_pe) if use_pe else (sc, tc)                                              # :372-373
    s_corr, t_corr = attend(s2, t2, t_xyz), attend(t2, s2, s_xyz)                                        # :374-377
    w, b = sd[prefix + 'conf_logits_decoder.weight'], sd[prefix + 'conf_logits_decoder.bias']
    return s_corr, sc @ w.t() + b, t_corr, tc @ w.t() + b                                                # :379-380


def compute_rigid_transform(a, b, weights):
    """utils/se3_torch.py:108-154 (weighted Kabsch)."""
    wn = weights[..., None] / torch.clamp_min(weights.sum(-1, keepdim=True)[..., None], 1e-6)
    ca = (a * wn).sum(-2)
    cb = (b * wn).sum(-2)
    ac = a - ca[..., None, :]
    bc = b - cb[..., None, :]
    cov = ac.transpose(-2, -1) @ (bc * wn)
    u, s, vh = torch.linalg.svd(cov)
    v = vh.transpose(-2, -1)
    rp = v @ u.transpose(-1, -2)
    vn = v.clone()
    vn[..., 2] *= -1
    rn = vn @ u.transpose(-1, -2)
    rot = torch.where(torch.det(rp)[..., None, None] > 0, rp, rn)
    t = -rot @ ca[..., :, None] + cb[..., :, None]
    return torch.cat((rot, t), dim=-1)


# ------------------------------------------------------------------------------------------------
# RegTR.forward  (models/regtr.py:104-235)
# ------------------------------------------------------------------------------------------------
# ground-truth overlap (training / validation side)
# ------------------------------------------------------------------------------------------------
def compute_overlaps(batch):
    """models/backbone_kpconv/kpconv.py:540-566."""
    overlaps = list(batch['src_overlap']) + list(batch['tgt_overlap'])
    meta = batch['kpconv_meta']
    pyr = {'pyr_0': torch.cat(overlaps, 0).type(torch.float)}                       # :553
    invalid = [s.sum() for s in meta['stack_lengths']]                                # :554
    for p in range(1, len(meta['points'])):
        pools = meta['pools'][p - 1].clone().long()
        valid = pools < invalid[p - 1]                                               # :557
        pools[~valid] = 0
        g = pyr[f'pyr_{p - 1}'][pools] * valid                                       # :561
        g = torch.sum(g, dim=1) / torch.sum(valid, dim=1)                            # :562
        pyr[f'pyr_{p}'] = torch.clamp(g, min=0, max=1)                               # :563
    return pyr


def compute_overlap(src, tgt, search_voxel_size):
    """utils/pointcloud.py:8-65 with open3d's KDTreeFlann.search_radius_vector_3d restated from its documented behaviour (open3d is not
    installed here): coordinates widened to float64, hits are the points with d2 < radius^2 sorted by distance, `knn_indices[0]` the
    nearest.  Pinned to the outputs of the REFERENCE's own function run over a scipy-cKDTree stand-in for that one class
    (oracle/make_golden_overlap.py -> tests/golden/overlap_pairs.npz); open3d's choice among exactly equidistant hits stays unpinned.
    Brute force, small clouds only."""
    src = np.asarray(src, np.float64); tgt = np.asarray(tgt, np.float64)
    r2 = float(search_voxel_size) ** 2

    def nearest(q, s):
        out = np.full(len(q), -1, np.int64)
        for i0 in range(0, len(q), 512):
            d = q[i0:i0 + 512, None, :] - s[None]
            d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
            j = np.argmin(d2, 1)
            hit = d2[np.arange(len(j)), j] < r2
            out[i0:i0 + 512][hit] = j[hit]
        return out
    tgt_corr = nearest(tgt, src)                                                     # :44-49
    src_corr = nearest(src, tgt)                                                     # :50-55
    mutual = np.logical_and(tgt_corr[src_corr] == np.arange(len(src_corr)), src_corr > 0)   # :58-59
    return src_corr >= 0, tgt_corr >= 0, np.stack([np.nonzero(mutual)[0], src_corr[mutual]])


# ------------------------------------------------------------------------------------------------
def regtr_forward(sd, cfg, src_list, tgt_list, meta=None, use_ref_cpp=False, timings=None):
    """Returns the reference's outputs dict (regtr.py:218-235) + 'kpconv_meta'."""
    import time
    t0 = time.perf_counter()
    B = len(src_list)
    if meta is None:
        meta = preprocess([np.asarray(p) for p in src_list + tgt_list], cfg, use_ref_cpp)
    t1 = time.perf_counter()
    feats = encoder(sd, cfg, meta)
    t2 = time.perf_counter()
    un = feats @ sd['feat_proj.weight'].t() + sd['feat_proj.bias']         # regtr.py:145
    xyz_c = meta['points'][-1]
    lens_c = meta['stack_lengths'][-1].tolist()
    pe = pos_embed_sine(xyz_c, cfg['d_embed'], cfg.get('pos_emb_scaling', 1.0))
    un_l = torch.split(un, lens_c); pe_l = torch.split(pe, lens_c); xyz_l = torch.split(xyz_c, lens_c)
    out = {k: [] for k in ('src_feat_un', 'tgt_feat_un', 'src_feat', 'tgt_feat', 'src_kp', 'tgt_kp',
                           'src_kp_warped', 'tgt_kp_warped', 'src_overlap', 'tgt_overlap')}
    poses = []
    zero = torch.zeros_like(pe_l[0][:1]) * 0
    for b in range(B):
        s, t = un_l[b], un_l[B + b]
        spe = pe_l[b] if cfg['transformer_encoder_has_pos_emb'] else zero
        tpe = pe_l[B + b] if cfg['transformer_encoder_has_pos_emb'] else zero
        sc, tc = transformer(sd, cfg, s, t, spe, tpe)                       # regtr.py:160-166
        if cfg.get('direct_regress_coor', False):
            s_corr, s_logit = correspondence_regressor(sd, sc)              # :168
            t_corr, t_logit = correspondence_regressor(sd, tc)
        else:
            s_corr, s_logit, t_corr, t_logit = correspondence_decoder_attn(sd, sc, tc, xyz_l[b], xyz_l[B + b], pe_l[b], pe_l[B + b],
                                                                            cfg['corr_decoder_has_pos_emb'])
        L = sc.shape[0]
        a = torch.cat([xyz_l[b].expand(L, -1, -1), t_corr], 1)              # :187-190
        bb = torch.cat([s_corr, xyz_l[B + b].expand(L, -1, -1)], 1)
        w = torch.cat([torch.sigmoid(s_logit[..., 0]), torch.sigmoid(t_logit[..., 0])], 1)   # :191-194
        poses.append(compute_rigid_transform(a, bb, w))                     # :200-203
        out['src_feat_un'].append(s); out['tgt_feat_un'].append(t)
        out['src_feat'].append(sc); out['tgt_feat'].append(tc)
        out['src_kp'].append(xyz_l[b]); out['tgt_kp'].append(xyz_l[B + b])
        out['src_kp_warped'].append(s_corr); out['tgt_kp_warped'].append(t_corr)
        out['src_overlap'].append(s_logit); out['tgt_overlap'].append(t_logit)
    out['pose'] = torch.stack(poses, 1)
    out['kpconv_meta'] = meta
    t3 = time.perf_counter()
    if timings is not None:
        timings.append((t1 - t0, t2 - t1, t3 - t2))
    return out


def ball_query_first_k(queries, supports, q_lens, s_lens, radius, K):
    """The neighbour table of the reference's PreprocessorGPU (batch_neighbors_kpconv_gpu, kpconv.py:261-288): per query the FIRST K
    supports of its own cloud with d2 < radius^2, in support-index order, padded with the total number of supports -- pytorch3d
    ball_query's documented behaviour (it walks the supports in order and stops at K; float32 d2 = sum of squared differences).
    PARITY UNPINNED: pytorch3d is not installable here, so this restates its documentation / kernel, not its output.
    numpy arrays in, (Nq, K) int64 out."""
    import numpy as np
    queries, supports = np.asarray(queries, np.float32), np.asarray(supports, np.float32)
    out = np.full((len(queries), K), len(supports), np.int64)
    r2 = np.float32(radius) * np.float32(radius)
    qo = so = 0
    for nq, ns in zip(q_lens, s_lens):
        s = supports[so:so + ns]
        for i in range(qo, qo + nq):
            d = s - queries[i]
            d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            hit = np.nonzero(d2 < r2)[0][:K]
            out[i, :len(hit)] = hit + so
        qo += nq; so += ns
    return out
