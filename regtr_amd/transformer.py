"""Cross-encoder transformer on the HIP kernels.

Host-side mirror of /root/reference/src/models/transformer/transformers.py (TransformerCrossEncoderLayer :84-258,
forward_pre :183-244; TransformerCrossEncoder :18-59) with identical parameter names.  The reference pads the two
clouds of each pair to (N_max, B, D) and masks; here every token of every cloud lives in one PACKED (N_total, D)
array with device-side segment offsets, so the shared-weight self-attention of src and tgt, both directions of the
cross-attention and the FFN each become ONE launch over all tokens of the batch.
"""
import copy
import ctypes

import torch
import torch.nn as nn

from . import _lib, context, ops
from .kpconv import _prepared, param_fingerprint


class TransformerCrossEncoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation='relu', normalize_before=False,
                 sa_val_has_pos_emb=False, ca_val_has_pos_emb=False, attention_type='dot_prod'):
        super().__init__()
        if attention_type != 'dot_prod':
            raise NotImplementedError                                             # transformers.py:94-98
        if activation != 'relu':
            raise NotImplementedError('only the ReLU feed-forward of the shipped configs is implemented')
        if dropout != 0.0:
            raise NotImplementedError('inference path: dropout must be 0 (as in both shipped configs)')
        # parameter containers with the reference's names; their torch forward is never called
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.nhead, self.d_model, self.normalize_before = nhead, d_model, normalize_before
        self.sa_val_has_pos_emb, self.ca_val_has_pos_emb = sa_val_has_pos_emb, ca_val_has_pos_emb
        self._cache = {}
        # arithmetic of the dense kernels (set by RegTR from cfg.compute_dtype): bf16 planes per GEMM operand (3 = float32-grade)
        # and the attention core's precision code (ops.mha)
        self.gemm_planes, self.attn_precision = 3, 0

    def _wt(self, name, param, rows=None):
        """The weight (optionally a row block of it: in_proj packs q, k, v) prepared once for the dense kernels."""
        key = name if rows is None else (name, rows)
        return _prepared(self._cache, key, param, lambda w: ops.SplitWeight(w if rows is None else w[rows[0]:rows[1]], 'nk'))

    def _gemm(self, a, w, **kw):
        return ops.gemm(a, w, planes=self.gemm_planes, **kw)

    def _attention(self, attn, tag, x, norm, pe, val_has_pe, seg_off, kv_of, max_len):
        """x + out_proj( MHA(q = k = LN(x) + pe, v = LN(x) [+ pe]) ) for every token (transformers.py:194-229)."""
        D = self.d_model
        b_in = attn.in_proj_bias.detach()
        if pe is None:
            x2p = ops.layernorm(x, norm.weight.detach(), norm.bias.detach(), eps=norm.eps)
            qkv = self._gemm(x2p, self._wt(tag + '_in', attn.in_proj_weight), bias=b_in)
        elif val_has_pe:
            x2p = ops.layernorm(x, norm.weight.detach(), norm.bias.detach(), add=pe, eps=norm.eps)
            qkv = self._gemm(x2p, self._wt(tag + '_in', attn.in_proj_weight), bias=b_in)
        else:
            x2p, x2 = ops.layernorm(x, norm.weight.detach(), norm.bias.detach(), add=pe, eps=norm.eps, want_plain=True)
            qkv = torch.empty((x.shape[0], 3 * D), dtype=torch.float32, device=x.device)
            b_qk = _prepared(self._cache, tag + '_bqk', attn.in_proj_bias, lambda b: b[:2 * D].contiguous())
            b_v = _prepared(self._cache, tag + '_bv', attn.in_proj_bias, lambda b: b[2 * D:].contiguous())
            self._gemm(x2p, self._wt(tag + '_in', attn.in_proj_weight, (0, 2 * D)), bias=b_qk, out=qkv[:, :2 * D])
            self._gemm(x2, self._wt(tag + '_in', attn.in_proj_weight, (2 * D, 3 * D)), bias=b_v, out=qkv[:, 2 * D:])
        att = ops.mha(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], seg_off, kv_of, max_len, self.nhead, self.attn_precision)
        return self._gemm(att, self._wt(tag + '_out', attn.out_proj.weight), bias=attn.out_proj.bias.detach(), residual=x)

    def _attention_post(self, attn, tag, x, norm, pe, val_has_pe, seg_off, kv_of, max_len):
        """LN( x + out_proj( MHA(q = k = x + pe, v = x [+ pe]) ) ) for every token (forward_post, transformers.py:131-166)."""
        D = self.d_model
        b_in = attn.in_proj_bias.detach()
        xp = x if pe is None else ops.add(x, pe)
        if pe is None or val_has_pe:
            qkv = self._gemm(xp, self._wt(tag + '_in', attn.in_proj_weight), bias=b_in)
        else:
            qkv = torch.empty((x.shape[0], 3 * D), dtype=torch.float32, device=x.device)
            b_qk = _prepared(self._cache, tag + '_bqk', attn.in_proj_bias, lambda b: b[:2 * D].contiguous())
            b_v = _prepared(self._cache, tag + '_bv', attn.in_proj_bias, lambda b: b[2 * D:].contiguous())
            self._gemm(xp, self._wt(tag + '_in', attn.in_proj_weight, (0, 2 * D)), bias=b_qk, out=qkv[:, :2 * D])
            self._gemm(x, self._wt(tag + '_in', attn.in_proj_weight, (2 * D, 3 * D)), bias=b_v, out=qkv[:, 2 * D:])
        att = ops.mha(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], seg_off, kv_of, max_len, self.nhead, self.attn_precision)
        y = self._gemm(att, self._wt(tag + '_out', attn.out_proj.weight), bias=attn.out_proj.bias.detach(), residual=x)
        return ops.layernorm(y, norm.weight.detach(), norm.bias.detach(), eps=norm.eps)

    def forward_post(self, x, pe, seg_off, kv_self, kv_cross, max_len):
        """transformers.py:121-181 on packed tokens: both cross-attention directions read the post-norm1 values."""
        x = self._attention_post(self.self_attn, 'sa', x, self.norm1, pe, self.sa_val_has_pos_emb, seg_off, kv_self, max_len)
        x = self._attention_post(self.multihead_attn, 'ca', x, self.norm2, pe, self.ca_val_has_pos_emb, seg_off, kv_cross,
                                 max_len)
        h = self._gemm(x, self._wt('l1', self.linear1.weight), bias=self.linear1.bias.detach(), relu=True)
        y = self._gemm(h, self._wt('l2', self.linear2.weight), bias=self.linear2.bias.detach(), residual=x)       # :168-170
        return ops.layernorm(y, self.norm3.weight.detach(), self.norm3.bias.detach(), eps=self.norm3.eps)

    def forward(self, x, pe, seg_off, kv_self, kv_cross, max_len):
        """x: (N_total, D) tokens of all clouds [src_0..src_{B-1}, tgt_0..tgt_{B-1}]."""
        if not self.normalize_before:                                              # transformers.py:255-258
            return self.forward_post(x, pe, seg_off, kv_self, kv_cross, max_len)
        x = self._attention(self.self_attn, 'sa', x, self.norm1, pe, self.sa_val_has_pos_emb, seg_off, kv_self, max_len)
        x = self._attention(self.multihead_attn, 'ca', x, self.norm2, pe, self.ca_val_has_pos_emb, seg_off, kv_cross,
                            max_len)
        x2 = ops.layernorm(x, self.norm3.weight.detach(), self.norm3.bias.detach(), eps=self.norm3.eps)   # :232
        h = self._gemm(x2, self._wt('l1', self.linear1.weight), bias=self.linear1.bias.detach(), relu=True)
        return self._gemm(h, self._wt('l2', self.linear2.weight), bias=self.linear2.bias.detach(), residual=x)  # :233-238


class TransformerCrossEncoder(nn.Module):
    def __init__(self, cross_encoder_layer, num_layers, norm=None, return_intermediate=False):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(cross_encoder_layer) for _ in range(num_layers)])   # :268-269
        self.num_layers, self.norm, self.return_intermediate = num_layers, norm, return_intermediate
        self._table = None

    def forward(self, x, pe, seg_off, kv_self, kv_cross, max_len):
        """-> (L, N_total, D) if return_intermediate else (1, N_total, D)   (transformers.py:37-59)."""
        n_out = self.num_layers if self.return_intermediate else 1
        outs = torch.empty((n_out,) + tuple(x.shape), dtype=torch.float32, device=x.device)
        if self._one_call_ok(x, pe):
            return self._forward_one_call(x, pe, seg_off, kv_self, kv_cross, max_len, outs)
        for li, layer in enumerate(self.layers):
            x = layer(x, pe, seg_off, kv_self, kv_cross, max_len)
            if self.return_intermediate:
                self._final(x, outs[li])
        if not self.return_intermediate:
            self._final(x, outs[0])
        return outs

    # ---- the whole stack through ONE C call (regtr_cross_encoder_fwd): the same 12 launches per layer, sequenced in C.  At one
    # pair per forward the host is the bound and these are 72 of its ~280 launches (csrc/cross_encoder.hip).
    def _one_call_ok(self, x, pe):
        l0 = self.layers[0]
        ctx = context.current()
        if not (ops.use_one_call_cross_encoder and ctx.mha_records is None and ctx.gemm_records is None and ctx.f16_range_log is None
                and not ops.force_f32_gemm and x.dim() == 2 and x.is_contiguous()
                and x.data_ptr() % 16 == 0 and (pe is None or pe.is_contiguous())):
            return False
        for layer in self.layers:
            if not (layer.normalize_before and (pe is None or (layer.sa_val_has_pos_emb and layer.ca_val_has_pos_emb))
                    and layer.gemm_planes == l0.gemm_planes and layer.attn_precision == l0.attn_precision and layer.nhead == l0.nhead
                    and layer.linear1.out_features == l0.linear1.out_features):
                return False
        return bool(_lib.lib().regtr_cross_encoder_supported(x.shape[0], x.shape[1], l0.linear1.out_features, l0.nhead))

    def _f16_pair(self, n):
        l0 = self.layers[0]
        D, F = l0.d_model, l0.linear1.out_features
        ctx = context.current()
        return bool(ctx.f16_pair and not ctx.force_x3 and l0.gemm_planes >= 2 and ops.f16_pair_ok(n, 3 * D, D) and ops.f16_pair_ok(n, D, D)
                    and ops.f16_pair_ok(n, F, D) and ops.f16_pair_ok(n, D, F))

    def _fingerprint(self):
        """Cheap identity of everything the cached tables point at (kpconv.param_fingerprint)."""
        return param_fingerprint(self)

    def _param_table(self, f16=False):
        """(ctypes array of the layers' device pointers in regtr_cross_encoder_fwd's order, ctypes array of the norm eps) -- rebuilt only
        when a parameter changes (address or version); the tensors behind the pointers are kept alive by the modules / the layers' weight
        caches.  -> (None, None) for f16 when a weight lies beyond the format's range."""
        fp = self._fingerprint()
        if self._table is None or self._table[0] != fp:
            self._table = (fp, {})
        tables = self._table[1]
        if f16 in tables:
            return tables[f16]
        ptrs, eps = [], []
        if f16:       # every weight inside the format's range (audited once per weight version, SplitWeight.f16_ok); else: the bf16 planes
            for layer in self.layers:
                m = layer._modules
                for tag, w in (('sa_in', m['self_attn'].in_proj_weight), ('sa_out', m['self_attn'].out_proj.weight),
                               ('ca_in', m['multihead_attn'].in_proj_weight), ('ca_out', m['multihead_attn'].out_proj.weight),
                               ('l1', m['linear1'].weight), ('l2', m['linear2'].weight)):
                    if not layer._wt(tag, w).f16_ok:
                        tables[f16] = (None, None)
                        return tables[f16]
        for layer in self.layers:
            m = layer._modules
            sa, ca = m['self_attn'], m['multihead_attn']
            n1, n2, n3, l1, l2 = m['norm1'], m['norm2'], m['norm3'], m['linear1'], m['linear2']
            eps += [n1.eps, n2.eps, n3.eps]
            pl = (lambda sw: sw.planes16) if f16 else (lambda sw: sw.planes)
            for t in (n1.weight, n1.bias, pl(layer._wt('sa_in', sa.in_proj_weight)), sa.in_proj_bias,
                      pl(layer._wt('sa_out', sa.out_proj.weight)), sa.out_proj.bias,
                      n2.weight, n2.bias, pl(layer._wt('ca_in', ca.in_proj_weight)), ca.in_proj_bias,
                      pl(layer._wt('ca_out', ca.out_proj.weight)), ca.out_proj.bias,
                      n3.weight, n3.bias, pl(layer._wt('l1', l1.weight)), l1.bias, pl(layer._wt('l2', l2.weight)), l2.bias):
                ptrs.append(t.data_ptr())
        for layer in self.layers:                       # checked once per table: everything the C side dereferences is float32 / bytes on one GPU
            for prm in layer.parameters():
                _lib.ptr(prm.detach())
        tables[f16] = ((ctypes.c_void_p * len(ptrs))(*ptrs), (ctypes.c_float * len(eps))(*eps))
        return tables[f16]

    def _forward_one_call(self, x, pe, seg_off, kv_self, kv_cross, max_len, outs):
        L = _lib.lib()
        l0 = self.layers[0]
        n, D = x.shape
        F = l0.linear1.out_features
        ctx = context.current()
        f16 = self._f16_pair(n)
        planes = 4 if f16 else (3 if ctx.force_x3 else int(l0.gemm_planes))
        attn_precision = 0 if (ctx.force_x3 and l0.attn_precision == 3) else int(l0.attn_precision)
        table, eps = self._param_table(f16)
        if table is None:                   # a weight beyond f16's range: the whole stack on the bf16 planes
            f16, planes = False, (3 if ctx.force_x3 else max(int(l0.gemm_planes), 3 if l0.attn_precision == 3 else 1))
            table, eps = self._param_table(False)
        nb = L.regtr_cross_encoder_ws_bytes(n, D, F)
        ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
        g = b = None
        feps = 0.0
        if self.norm is not None:
            g, b, feps = self.norm.weight.detach(), self.norm.bias.detach(), self.norm.eps
        _lib.check(L.regtr_cross_encoder_fwd(_lib.ptr(x), n, D, F, l0.nhead, self.num_layers, table, eps, _lib.ptr(g), _lib.ptr(b), feps,
                                             1 if self.return_intermediate else 0, _lib.ptr(pe), _lib.iptr(seg_off), _lib.iptr(kv_self),
                                             _lib.iptr(kv_cross), seg_off.numel() - 1, int(max_len), planes,
                                             attn_precision, _lib.bptr(ws), nb, _lib.ptr(outs), ctx.status_ptr(), _lib.stream()),
                   'regtr_cross_encoder_fwd')
        return outs

    def _final(self, x, out):
        if self.norm is None:
            out.copy_(x)
        else:
            ops.layernorm(x, self.norm.weight.detach(), self.norm.bias.detach(), eps=self.norm.eps, out=out)
