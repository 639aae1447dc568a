"""REGTR network, inference forward, on the MI355X HIP kernels.

Drop-in for the reference's `RegTR` (/root/reference/src/models/regtr.py:22-235): same constructor (`RegTR(cfg)`),
same `forward(batch) -> dict` contract (input lists batch['src_xyz'] / batch['tgt_xyz'], side effect
batch['kpconv_meta'], output keys of regtr.py:218-235), same `state_dict` names and shapes, so a reference checkpoint
loads with `load_state_dict(state['state_dict'])` (demo.py:165).  Training-only members (compute_loss, criteria) are
not implemented; their parameters (`feature_criterion.W`, feature_loss.py:261) are kept so strict loading works.

The forward enqueues every stage on the current HIP stream with ONE host synchronisation (the data-dependent level
sizes after preprocessing): preprocess -> KPConv encoder -> feat_proj -> 6 cross-encoder layers on packed tokens ->
correspondence head -> fused weighted-Procrustes.
"""
import logging
import threading

import torch
import torch.nn as nn

from . import _lib, context, devflags, ops
from .config import as_config
from .kpconv import KPFEncoder, PreprocessorGPU, _prepared
from .transformer import TransformerCrossEncoder, TransformerCrossEncoderLayer

_TIMEIT = False   # reference: regtr.py:19 -- stage timer (preprocess / encoder / attention+head / pose / total)


class PositionEmbeddingCoordsSine(nn.Module):
    """position_embedding.py:7-50 (parameter free)."""

    def __init__(self, n_dim=1, d_model=256, temperature=10000, scale=None):
        super().__init__()
        if n_dim != 3:
            raise NotImplementedError
        self.n_dim, self.d_model, self.temperature = n_dim, d_model, temperature
        self.scale = 1.0 if scale is None else scale

    def forward(self, xyz):
        return ops.posemb_sine(xyz, self.d_model, self.scale, self.temperature)


# A-B switch and size gate of the two-stream forward (RegTR._forward)
overlap_preprocessing = devflags.on('REGTR_OVERLAP')          # (read only under REGTR_DEV=1)
OVERLAP_MIN_POINTS = 131072


class CorrespondenceRegressor(nn.Module):
    """regtr.py:399-443: coor_mlp (D->D->D->3, ReLU) and conf_logits_decoder (D->1)."""

    def __init__(self, d_embed):
        super().__init__()
        self.coor_mlp = nn.Sequential(nn.Linear(d_embed, d_embed), nn.ReLU(), nn.Linear(d_embed, d_embed), nn.ReLU(),
                                      nn.Linear(d_embed, 3))
        self.conf_logits_decoder = nn.Linear(d_embed, 1)
        self._cache = {}

    def forward(self, feats):
        """feats (L, N, D) -> corr (L, N, 3), logit (L, N)."""
        Lyr, N, D = feats.shape
        f = feats.view(Lyr * N, D)
        wt = lambda k, lin: _prepared(self._cache, k, lin.weight, lambda w: ops.SplitWeight(w, 'nk'))
        h = ops.gemm(f, wt('0', self.coor_mlp[0]), bias=self.coor_mlp[0].bias.detach(), relu=True)
        h = ops.gemm(h, wt('2', self.coor_mlp[2]), bias=self.coor_mlp[2].bias.detach(), relu=True)
        corr = ops.gemm(h, wt('4', self.coor_mlp[4]), bias=self.coor_mlp[4].bias.detach())
        logit = ops.gemm(f, wt('c', self.conf_logits_decoder), bias=self.conf_logits_decoder.bias.detach())
        return corr.view(Lyr, N, 3), logit.view(Lyr, N)


class CorrespondenceDecoder(nn.Module):
    """regtr.py:299-396 (`direct_regress_coor: False`): correspondences as attention-weighted partner coordinates --
    softmax(q_proj(f [+ pe]) . k_proj(f_partner [+ pe]) / sqrt(D)) @ xyz_partner per decoder layer -- and the overlap logit.
    `q_norm` is a parameter of the reference module that its forward never uses; it is kept for strict loading."""

    def __init__(self, d_embed, use_pos_emb, num_neighbors=0):
        super().__init__()
        if num_neighbors != 0:
            raise NotImplementedError('num_neighbors > 0 (top-k masked attention) is not used by the reference model')
        self.use_pos_emb = use_pos_emb
        self.q_norm = nn.LayerNorm(d_embed)
        self.q_proj = nn.Linear(d_embed, d_embed)
        self.k_proj = nn.Linear(d_embed, d_embed)
        self.conf_logits_decoder = nn.Linear(d_embed, 1)
        self._cache = {}

    def forward(self, feats, pe, xyz, seg_off, kv_cross, max_len):
        """feats (L, N, D) conditioned features of all clouds, pe (N, D), xyz (N, 3) -> corr (L, N, 3), logit (L, N)."""
        Lyr, N, D = feats.shape
        wt = lambda k, lin: _prepared(self._cache, k, lin.weight, lambda w: ops.SplitWeight(w, 'nk'))
        f2 = torch.stack([ops.add(feats[l], pe) for l in range(Lyr)]) if self.use_pos_emb else feats       # :372-373
        q = ops.gemm(f2.view(Lyr * N, D), wt('q', self.q_proj), bias=self.q_proj.bias.detach())
        k = ops.gemm(f2.view(Lyr * N, D), wt('k', self.k_proj), bias=self.k_proj.bias.detach())
        corr = ops.attn_xyz(q.view(Lyr, N, D), k.view(Lyr, N, D), xyz, seg_off, kv_cross, max_len)         # :374-377
        logit = ops.gemm(feats.reshape(Lyr * N, D), wt('c', self.conf_logits_decoder), bias=self.conf_logits_decoder.bias.detach())
        return corr, logit.view(Lyr, N)


def _throttled(n):
    """Log occurrence no. n?  The first three, then every power of two: a permanent condition stays visible without flooding the log."""
    return n <= 3 or (n & (n - 1)) == 0


class _LossParams(nn.Module):
    """Holds InfoNCELossFull.W (feature_loss.py:261) so reference checkpoints load strictly; never used at inference."""

    def __init__(self, d_embed):
        super().__init__()
        self.W = nn.Parameter(torch.zeros(d_embed, d_embed), requires_grad=False)


class RegTR(nn.Module):
    def __init__(self, cfg, *args, **kwargs):
        super().__init__()
        cfg = as_config(cfg)
        self.cfg = cfg
        self.logger = logging.getLogger(self.__class__.__name__)
        self.preprocessor = PreprocessorGPU(cfg)                                  # regtr.py:29
        self.kpf_encoder = KPFEncoder(cfg, cfg.d_embed)                           # :34
        self.feat_proj = nn.Linear(self.kpf_encoder.encoder_skip_dims[-1], cfg.d_embed, bias=True)   # :36
        if cfg.get('pos_emb_type', 'sine') == 'sine':                             # :41-47
            self.pos_embed = PositionEmbeddingCoordsSine(3, cfg.d_embed, scale=cfg.get('pos_emb_scaling', 1.0))
        else:
            raise NotImplementedError('only the sine positional embedding of the shipped configs is implemented')
        encoder_layer = TransformerCrossEncoderLayer(                             # :52-59
            cfg.d_embed, cfg.nhead, cfg.d_feedforward, cfg.dropout, activation=cfg.transformer_act,
            normalize_before=cfg.pre_norm, sa_val_has_pos_emb=cfg.sa_val_has_pos_emb,
            ca_val_has_pos_emb=cfg.ca_val_has_pos_emb, attention_type=cfg.attention_type)
        encoder_norm = nn.LayerNorm(cfg.d_embed) if cfg.pre_norm else None
        # cfg.compute_dtype (not a reference key).  Every dense contraction runs on the 16-bit matrix cores with EXACT operand splits and
        # float32 accumulation.  'fp32' (default): the f16 PAIR split -- x = h0 + h1 / 2048, three f16 MFMA terms, error ~2^-22 per
        # product, 22-bit operands that must stay below f16's 65504 -- in the split GEMMs and the attention core, the six-term bf16 split
        # (~2^-24) in the one-shot strip / block-tail kernels of the shallow levels and wherever the f16 format does not serve a launch.
        # RANGE: InstanceNorm / LayerNorm outputs and their gathered sums sit far below the limit with sane weights (largest |A| 116 on
        # the benchmark workload), but nothing bounds a checkpoint's FFN activations, so it is CHECKED: weights once per version
        # (ops.SplitWeight.f16_ok), activations by the kernels themselves -- a non-finite f16 pair product or pose sets a bit in a
        # device status word that forward() reads once, at its end, and on a trip the forward is re-run in 'fp32x3' arithmetic and the
        # event logged (cfg.f16_range_check: False skips the check and its end-of-forward wait).  Validated against the real reference
        # module's outputs on the goldens in parity mode (worst 2.6e-5, bar 1e-4) and on the benchmarked batch (bench.py's `parity`).
        # 'fp32x3' = six bf16 terms everywhere (float32's operand range, no f16); 'bf16x2' = three bf16 terms in the cross-encoder's
        # Linears without the f16 pair (round-2 callers); 'bf16' = plain bf16 operands with float32 accumulation / softmax in the
        # cross-encoder's Linears and attention core (BASELINE configs[1]; encoder, head, pose as 'fp32').
        dt = cfg.get('compute_dtype', 'fp32')
        if dt not in ('fp32', 'fp32x3', 'bf16', 'bf16x2'):
            raise NotImplementedError(f'compute_dtype {dt!r}: choose fp32, fp32x3, bf16x2 or bf16')
        # (bf16 planes per operand where the f16 pair does not serve a launch: 'fp32' then means six terms, like 'fp32x3')
        encoder_layer.gemm_planes = {'fp32': 3, 'fp32x3': 3, 'bf16x2': 2, 'bf16': 1}[dt]
        encoder_layer.attn_precision = 1 if dt == 'bf16' else (3 if (dt == 'fp32' and ops.f16_pair_default) else 0)      # ops.mha's codes
        self._f16_pair = dt in ('fp32', 'bf16') and ops.f16_pair_default      # ('bf16': the encoder / head GEMMs, which stay float32-grade)
        self._range_check = bool(cfg.get('f16_range_check', True))
        self.f16_range_fallbacks = 0          # forwards re-run in fp32x3 arithmetic because an f16 pair operand left the format's range
        self.nonfinite_pose_forwards = 0      # forwards whose pose came out non-finite with every f16 pair product finite (bad inputs / weights)
        self.transformer_encoder = TransformerCrossEncoder(encoder_layer, cfg.num_encoder_layers, encoder_norm,
                                                           return_intermediate=True)
        if cfg.get('direct_regress_coor', False):                                 # :68-73
            self.correspondence_decoder = CorrespondenceRegressor(cfg.d_embed)
        else:
            self.correspondence_decoder = CorrespondenceDecoder(cfg.d_embed, cfg.corr_decoder_has_pos_emb)
        if cfg.get('feature_loss_type', 'infonce') == 'infonce':                  # :79-81
            self.feature_criterion = _LossParams(cfg.d_embed)
            self.feature_criterion_un = _LossParams(cfg.d_embed)
        self._cache = {}
        self.last_timings = None
        self._params_checked = False

    def _apply(self, fn, *args, **kwargs):          # .to() / .cuda() / .half() / .double(): re-validate at the next forward
        self._params_checked = False
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._params_checked = False
        return super().load_state_dict(*args, **kwargs)

    def _status_word(self, dev):
        """(device int32[1], pinned host int32[1]) the kernels of ONE forward OR bits into.  One pair per (host thread, HIP stream): two
        threads -- or two streams -- driving the SAME model each zero, fill and read their own word, so one forward can neither erase nor
        steal another's overflow bit (a forward is complete on its stream before the next one on that stream zeroes the word)."""
        key = ('status', dev, threading.get_ident(), _lib.stream())
        ent = self._cache.get(key)
        if ent is None:
            if sum(1 for k in self._cache if isinstance(k, tuple) and k and k[0] == 'status') > 64:       # dead threads / streams
                for k in [k for k in self._cache if isinstance(k, tuple) and k and k[0] == 'status']:
                    del self._cache[k]
            ent = self._cache[key] = (torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32).pin_memory())
        return ent

    def _ones(self, n, dev):
        """RegTR's input features (regtr.py:136: ones, one per point) as a view of a cached buffer: constant data, a fill kernel and an
        allocation less per forward.  One buffer per (device, stream), like the status word: created and regrown in stream order, so a second
        thread / stream driving this model can neither read it before its fill kernel ran nor lose it to another stream's regrow."""
        key = ('ones', dev, torch.cuda.current_stream(dev).cuda_stream)
        ent = self._cache.get(key)
        if ent is None or ent.shape[0] < n:
            ent = self._cache[key] = torch.ones((max(n, 1) * 5 // 4 + 16, 1), dtype=torch.float32, device=dev)
        return ent[:n]

    def _side_stream(self, dev):
        """The pyramid's stream (large batches: levels 1-3 are built under the level-0 blocks).  HIGH priority, not for the priority: HIP maps
        the streams of one priority onto a small pool of hardware queues, and a stream that lands on the main stream's queue is serialised
        behind it.  That happened as soon as the process also held an RCCL communicator (its streams shift the mapping): a 192-pair forward
        79.98 ms under a world-1 process group against 77.84 ms without, the rocprofv3 trace showing no overlap at all (busy 78.2 ms inside 79.5 ms
        of wall time, against 84.4 inside 76.6).  A high-priority stream draws from its own queue pool: 78.20 / 77.94 ms (profiles/r05_z_dist_stream.txt)."""
        return _prepared(self._cache, ('side_stream', dev), self.feat_proj.bias,
                         lambda _: torch.cuda.Stream(device=dev, priority=int(devflags.flag('REGTR_SIDE_PRIO', '-1'))))

    @staticmethod
    def _record_meta(meta, stream):
        """Pyramid tensors are allocated on the side stream and read on `stream`: tell the caching allocator."""
        for key in ('points', 'neighbors', 'pools', '_neighbors_i32', '_pools_i32', '_seg_off'):
            for t in meta[key]:
                t.record_stream(stream)

    @property
    def device(self):
        return next(self.parameters()).device

    @torch.no_grad()
    def forward(self, batch):
        dev = batch['src_xyz'][0].device
        if dev.type != 'cuda':
            raise RuntimeError('regtr_amd.RegTR runs on an MI355X (HIP) device only; there is no CPU path')
        clouds = batch['src_xyz'] + batch['tgt_xyz']
        if not self._params_checked:        # once per parameter (re)placement: _apply() below resets it
            bad = [n for n, p in self.named_parameters() if p.dtype != torch.float32]
            if bad:
                raise RuntimeError(f'regtr_amd.RegTR computes in float32; non-float32 parameters: {bad[:3]} ...')
            self._model_device = next(self.parameters()).device
            self._params_checked = True
        if any(p.device != dev for p in clouds) or self._model_device != dev:
            raise RuntimeError(f'RegTR.forward: the model ({self._model_device}) and every input cloud must live on one GPU ({dev})')
        # kernels go to torch's current stream of the CURRENT device: the context makes the tensors' device current for the whole
        # forward and carries the operand format / status word to every launch (thread-local: regtr_amd/context.py)
        check = self._f16_pair and self._range_check
        with context.forward(dev, f16_pair=self._f16_pair, status=None) as ctx:
            if not check:
                return self._forward(batch, dev)
            st_dev, st_host = self._status_word(dev)         # (inside the context: the launch device is current, the stream is this thread's)
            ctx.status = st_dev
            st_dev.zero_()
            out = self._forward(batch, dev)
            st_host.copy_(st_dev, non_blocking=True)
            done = torch.cuda.Event()
            done.record()
            done.synchronize()                  # the one wait at the end of a forward: the status word (4 bytes, pinned)
            bits = int(st_host[0])
        if bits == 0:
            return out
        if not bits & context.STATUS_F16_RANGE:
            # only the pose is non-finite: the f16 pair products were all finite, so the arithmetic is not the cause -- a NaN / Inf in the
            # input clouds or the weights, or a degenerate pair.  A re-run in fp32x3 would return the same NaN at twice the cost.
            self.nonfinite_pose_forwards += 1
            if self.nonfinite_pose_forwards == 1 and self._f16_pair:
                # ... the FIRST time, checked instead of assumed: one re-run with six-term bf16 splits.  A finite pose there means an f16 pair
                # kernel produced a degenerate result WITHOUT raising its range bit -- a defect to report, and the finite result is returned.
                with context.forward(dev, f16_pair=False, force_x3=True, status=None):
                    again = self._forward(batch, dev)
                if bool(torch.isfinite(again['pose']).all()):
                    self.logger.error("non-finite pose with NO f16 range bit set, but the fp32x3 re-run is finite: an f16 pair kernel missed its "
                                      "range report (status %d) -- returning the fp32x3 result; set cfg.compute_dtype: fp32x3 and report this", bits)
                    return again
                self.logger.warning('non-finite pose (status %d): the fp32x3 re-run is non-finite too -- the inputs or the checkpoint hold NaN / Inf', bits)
            if _throttled(self.nonfinite_pose_forwards):
                self.logger.warning('non-finite pose in the output (status %d; forward no. %d with this condition): every f16 pair product was '
                                    'finite, so check the input clouds and the checkpoint for NaN / Inf -- returned as is, not re-run',
                                    bits, self.nonfinite_pose_forwards)
            return out
        # an operand left f16's range: the same forward in fp32x3 arithmetic -- float32's range
        self.f16_range_fallbacks += 1
        if _throttled(self.f16_range_fallbacks):
            self.logger.warning('f16 pair operand range exceeded (status %d; fallback no. %d) -- forward re-run with six-term bf16 splits '
                                "(compute_dtype 'fp32x3' arithmetic, twice the cost of this forward); set cfg.compute_dtype: fp32x3 to skip "
                                'the first attempt', bits, self.f16_range_fallbacks)
        with context.forward(dev, f16_pair=False, force_x3=True, status=None):
            return self._forward(batch, dev)

    def _forward(self, batch, dev):
        B = len(batch['src_xyz'])
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)] if _TIMEIT else None
        if ev: ev[0].record()

        # ---- preprocess (regtr.py:117-122) + KPConv encoder (regtr.py:136)
        clouds = batch['src_xyz'] + batch['tgt_xyz']
        n0 = sum(int(p.shape[0]) for p in clouds)
        n_l0 = self.kpf_encoder.level0_blocks()
        two_streams = (overlap_preprocessing and n_l0 > 0 and not ev and not self.cfg.get('kpconv_ref_row_order', False))
        if two_streams and n0 >= OVERLAP_MIN_POINTS:
            # Large batches: the pyramid is built on a second HIP stream, and the level-0 blocks -- which need only level 0's conv
            # table and sizes the host already knows -- start on the main stream as soon as that table exists: the other 2.4 ms of
            # pyramid building (latency-bound small kernels) run under 6 ms of bandwidth-bound level-0 convolution.
            main = torch.cuda.current_stream()
            side = self._side_stream(dev)
            ev0 = torch.cuda.Event()
            side.wait_stream(main)                                   # the inputs
            with torch.cuda.stream(side):
                state = self.preprocessor.enqueue(clouds, level0_event=ev0)
            meta0 = self.preprocessor.level0_meta(state)
            for t in (meta0['points'][0], meta0['_neighbors_i32'][0], meta0['_seg_off'][0]):
                t.record_stream(main)                                # allocated on `side`, read on `main`
            main.wait_event(ev0)
            feats0 = torch.ones_like(meta0['points'][0][:, 0:1])
            x, skips = self.kpf_encoder(feats0, meta0, 0, n_l0)
            kpconv_meta = self.preprocessor.finish(state)            # host + main wait for the pyramid's `done` event only
            if kpconv_meta is None:      # a level filled its capacity (rare): the pyramid once more, here, at full capacity -- the level-0
                kpconv_meta = self.preprocessor(clouds)          # blocks already enqueued read level 0 only, which is never truncated
            else:
                self._record_meta(kpconv_meta, main)
            batch['kpconv_meta'] = kpconv_meta
            feats_un, _ = self.kpf_encoder(x, kpconv_meta, n_l0, None, skips)
        else:
            kpconv_meta = self.preprocessor(clouds)
            batch['kpconv_meta'] = kpconv_meta
            # (small batches -- a pair or two per forward -- run ONE stream: the level-0 blocks overlapped with pyramid levels 1-3 on a second
            #  stream, with the pyramid sequenced from C in two phases, measured 2.46 vs 2.38 ms per pair: docs/NEGATIVES.md, round 5)
            feats0 = self._ones(kpconv_meta['points'][0].shape[0], dev)
            if ev: ev[1].record()
            feats_un, _ = self.kpf_encoder(feats0, kpconv_meta)
        slens_c = kpconv_meta['_lens_host'][-1]
        src_slens_c, tgt_slens_c = slens_c[:B], slens_c[B:]
        if ev: ev[2].record()

        # ---- projection, positional embedding, cross-encoder on packed tokens (regtr.py:145-166)
        wt = _prepared(self._cache, 'feat_proj', self.feat_proj.weight, lambda w: ops.SplitWeight(w, 'nk'))
        both_feats_un = ops.gemm(feats_un, wt, bias=self.feat_proj.bias.detach())
        xyz_c = kpconv_meta['points'][-1]
        seg_c = kpconv_meta['_seg_off'][-1]
        pe = self.pos_embed(xyz_c) if self.cfg.transformer_encoder_has_pos_emb else None
        kv_self = _prepared(self._cache, ('kv_self', B, dev), self.feat_proj.bias,
                            lambda _: torch.arange(2 * B, dtype=torch.int32, device=dev))
        kv_cross = _prepared(self._cache, ('kv_cross', B, dev), self.feat_proj.bias,
                             lambda _: torch.cat([torch.arange(B, 2 * B), torch.arange(0, B)]).to(torch.int32).to(dev))
        feats_cond = self.transformer_encoder(both_feats_un, pe, seg_c, kv_self, kv_cross, max(slens_c))   # (L, N, D)

        # ---- correspondence head (regtr.py:168) and pose (regtr.py:185-203)
        if isinstance(self.correspondence_decoder, CorrespondenceRegressor):
            corr, logit = self.correspondence_decoder(feats_cond)
        else:
            pe_c = pe if pe is not None else self.pos_embed(xyz_c)
            corr, logit = self.correspondence_decoder(feats_cond, pe_c, xyz_c, seg_c, kv_cross, max(slens_c))
        if ev: ev[3].record()
        pose = ops.weighted_procrustes(xyz_c, corr, logit, seg_c, B)
        if ev:
            ev[4].record()
            torch.cuda.synchronize()
            self.last_timings = {'preprocess': ev[0].elapsed_time(ev[1]) / 1e3, 'encoder': ev[1].elapsed_time(ev[2]) / 1e3,
                                 'attention': ev[2].elapsed_time(ev[3]) / 1e3, 'pose': ev[3].elapsed_time(ev[4]) / 1e3,
                                 'total': ev[0].elapsed_time(ev[4]) / 1e3}

        # ---- unpack into the reference's per-pair lists (views, no copies)
        off = [0]
        for n in slens_c:
            off.append(off[-1] + n)
        sl = lambda c: slice(off[c], off[c + 1])
        logit3 = logit.unsqueeze(-1)
        outputs = {
            'src_feat_un': tuple(both_feats_un[sl(b)] for b in range(B)),
            'tgt_feat_un': tuple(both_feats_un[sl(B + b)] for b in range(B)),
            'src_feat': [feats_cond[:, sl(b)] for b in range(B)],          # List(B) of (N_pred, N_src, D)
            'tgt_feat': [feats_cond[:, sl(B + b)] for b in range(B)],
            'src_kp': tuple(xyz_c[sl(b)] for b in range(B)),
            'src_kp_warped': [corr[:, sl(b)] for b in range(B)],
            'tgt_kp': tuple(xyz_c[sl(B + b)] for b in range(B)),
            'tgt_kp_warped': [corr[:, sl(B + b)] for b in range(B)],
            'src_overlap': [logit3[:, sl(b)] for b in range(B)],
            'tgt_overlap': [logit3[:, sl(B + b)] for b in range(B)],
            'pose': pose,
        }
        return outputs
