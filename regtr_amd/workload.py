"""The benchmarked workloads of bench.py / test.py and the accounting behind the bench line's rooflines (moved out of bench.py in round 6: the tests
build the same objects).  build_workload: model + batch exactly as the timed loop uses them; real_pairs: the three 3DMatch pairs the reference ships
(/root/reference/src/demo.py:26-49) replicated under random rigid motions; probe_head: the linear-probe head that conditions the Procrustes problem of a
random-init network; forward_compulsory_bytes / forward_matrix_seconds: SURVEY.md section 8(d) / Appendix B priced over a whole forward.
The CPU-oracle parity gate (bench.parity_check) stays in bench.py: nothing under regtr_amd/ may import oracle/."""
import os

import numpy as np
import torch

from .synthetic import synth_modelnet_pair, synth_pair

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_BF16_PEAK_TFS = 2500.0   # dense bf16 / f16 MFMA peak (same guide)
MFMA_F32_PEAK_TFS = 157.3     # f32-input MFMA peak


def kpconv_algorithmic_bytes(nq, H, cin, cout, kp=15):
    """B_kp = Nq*H*(4 + 12 + 4*Cin) + Nq*(12 + 4*Cout) + 15*Cin*Cout*4 (fp32 feats, int32 idx)."""
    return nq * H * (4 + 12 + 4 * cin) + nq * (12 + 4 * cout) + kp * cin * cout * 4


# pairs per forward and concurrent forwards per GPU when --pairs / --replicas are not given.
# 3dmatch: THREE concurrent 64-pair forwards = 192 pairs per step (round 6).  Rounds 1-4 ran one 64-pair forward at a time, round 5 one 192-pair
# forward (+3 %: fixed costs spread over three times the pairs).  A forward has two host waits and a 5-6 ms head during which nothing else of it
# can run, so a second and third forward in flight (ReplicaRunner below) fill the chip: same box, pairs/s, one forward at a time -> R threads:
# 3 x 64: 2009 -> 2257 / 2223 / 2219;  2 x 64: 2037 -> 2187;  4 x 64: 2062 -> 2193 / 2128 / 2166;  2 x 96: 2062 -> 2182;  4 x 96: 2058 -> 2213 / 2197;
# 3 x 128: 2056 -> 2171;  2 x 192: 2085 -> 2158;  3 x 192: 2105 -> 2203;  one 192-pair forward: 2085-2117  (profiles/r06_m_concurrency_sweep.txt).
# lomatch: at most 64 per forward, a shard cut into EQUAL forwards (bench.plan_pairs), three in flight; modelnet: three 128-pair forwards in flight
# (one 256-pair forward at a time 4241, 2 x 128 4304-4489, 3 x 86 4374, 4 x 64 4169, 3 x 128 4472 pairs/s: profiles/r06_o_*, r06_mn_*).
DEFAULT_PAIRS = {'3dmatch': 64, 'modelnet': 128, 'lomatch': 64}
DEFAULT_REPLICAS = {'3dmatch': 3, 'modelnet': 3, 'lomatch': 3}
REDUCED_TOL = {'correspondence': 2e-2, 'pose': 1e-1}      # gate of the bf16 / bf16x2 lines against the float32-grade run (see main)

REAL_PAIRS = ('3dmatch_kitchen', '3dmatch_hotel', '3dmatch_home_at')     # tests/golden/*.npz: the clouds of /root/reference/src/demo.py:26-49 examples 0-2


def real_pairs(n_pairs, first_id=0):
    """`--real`: the three REAL 3DMatch pairs the reference ships (demo.py:26-49: red-kitchen 0 / 5, hotel_umd 8 / 15, home_at 38 / 41 -- 6 mm
    lattice ties, home_at with 22.7 % of its level-0 balls over K = 40; the clouds travel as the committed fixtures tests/golden/3dmatch_*.npz)
    replicated to `n_pairs`: slots 0-2 are the originals, every further slot is pair (slot % 3) with each cloud under its own random rigid motion
    (rotation <= 45 deg about a random axis, |t| <= 0.5 m: conf/3dmatch.yaml's augmentation ranges), seeded by the slot id, applied in float32
    -- `n_pairs` different inputs with real-scan neighbourhood statistics.  -> [(src, tgt) float32 numpy]"""
    from regtr_amd.synthetic import random_se3
    base = [np.load(os.path.join(ROOT, 'tests', 'golden', f'{n}.npz')) for n in REAL_PAIRS]
    base = [(np.ascontiguousarray(g['src'], np.float32), np.ascontiguousarray(g['tgt'], np.float32)) for g in base]
    out = []
    for i in range(n_pairs):
        sl = first_id + i
        s, t = base[sl % len(base)]
        if sl >= len(base):
            rng = np.random.default_rng(7000003 + sl)
            (Rs, ts), (Rt, tt) = random_se3(rng, 45.0, 0.5), random_se3(rng, 45.0, 0.5)
            s = (s @ Rs.astype(np.float32).T + ts.astype(np.float32)).astype(np.float32)
            t = (t @ Rt.astype(np.float32).T + tt.astype(np.float32)).astype(np.float32)
        out.append((s, t))
    return out


def probe_head(model, calib, dev, ridge=1.0):
    """head_init 'probe': the output layer of the correspondence MLP (regtr.py:432-436, 3 x 256 + bias) fitted by ridge regression so that
    the head predicts each token's OWN coordinates from the conditioned features of `calib` pairs (all six decoder layers, both clouds).
    Why: the Kabsch covariance (se3_torch.py:108-154) is sum w (a - a_mean)(b - b_mean)^T over a = [src_kp ; tgt_corr], b = [src_corr ;
    tgt_kp].  With a RANDOM output layer the predicted correspondences are spread over the object (singular values 6.7 / 5.0 / 3.4) but
    UNCORRELATED with the key points (a linear fit explains 1.5 % of them): the covariance is a noise matrix, 0.020 / 0.011 / 0.0017, whose
    condition number is an accident -- s1 / (s2 + s3) = 40 ... 380 on the ModelNet-size pairs, where the ORACLE's own float32 Kabsch is up to
    1.1e-4 away from a float64 solve of the same inputs.  A trained head's predictions are a rigid image of the key points; the probe gives a
    random-init network that property (r^2 ~ 0.3: the features carry the sine position embedding), the covariance becomes ~Var(kp), and
    s1 / (s2 + s3) drops to 2 ... 11 on held-out pairs (float32-vs-float64 Kabsch 3e-7 ... 3e-6).  Everything upstream stays random-init."""
    head = model.correspondence_decoder
    with torch.no_grad():
        out = model({'src_xyz': [torch.from_numpy(s).to(dev) for s, _ in calib], 'tgt_xyz': [torch.from_numpy(t).to(dev) for _, t in calib]})
        H, T = [], []
        for side in ('src', 'tgt'):
            for f, kp in zip(out[side + '_feat'], out[side + '_kp']):           # f (6, N, D), kp (N, 3)
                h = torch.relu(torch.nn.functional.linear(f.reshape(-1, f.shape[-1]), head.coor_mlp[0].weight, head.coor_mlp[0].bias))
                h = torch.relu(torch.nn.functional.linear(h, head.coor_mlp[2].weight, head.coor_mlp[2].bias))
                H.append(h.double()); T.append(kp.expand(f.shape[0], -1, -1).reshape(-1, 3).double())
        X = torch.cat(H); X = torch.cat([X, torch.ones_like(X[:, :1])], 1)
        T = torch.cat(T)
        sol = torch.linalg.solve(X.T @ X + ridge * torch.eye(X.shape[1], dtype=X.dtype, device=X.device), X.T @ T)     # (D + 1, 3)
        head.coor_mlp[4].weight.copy_(sol[:-1].T.float())
        head.coor_mlp[4].bias.copy_(sol[-1].float())
        r2 = 1.0 - float(((X @ sol - T) ** 2).sum() / ((T - T.mean(0)) ** 2).sum())
    return r2


def build_workload(config, n_pairs, points, shuffle, rank, dev, dtype, parity_mode=False, first_id=None, distinct=None, real=False,
                   head_init=None):
    """The benchmarked model and batch, exactly as the timed loop uses them (tests/test_gpu_bench_batch.py builds the same objects):
    conf/<config>.yaml architecture with torch.manual_seed(0) random-init weights, `n_pairs` deterministic synthetic pairs
    (ids rank * 100003 + i, or first_id + i) resident on `dev`.  config 'lomatch' = the 3dmatch pipeline on 10-30 %-overlap pairs.
    real: the three shipped real 3DMatch pairs replicated under random rigid motions instead of synthetic rooms (real_pairs).
    distinct: generate only that many different pairs and cycle through them (setup time; nothing is cached between pairs).
    head_init: 'probe' (default since round 6) or 'uniform' (U(-0.5, 0.5): the synthetic 3DMatch-size lines of rounds 4-5) -- see below / probe_head.
    -> (cfg, model, pairs [(src, tgt) numpy], batch {'src_xyz': [...], 'tgt_xyz': [...]})"""
    from regtr_amd import RegTR, load_config
    cfg = load_config(os.path.join(ROOT, 'regtr_amd', 'conf', f'{"3dmatch" if config == "lomatch" else config}.yaml'))
    cfg.update({'compute_dtype': dtype})
    if parity_mode:
        cfg.update({'kpconv_ref_row_order': True})
    torch.manual_seed(0); np.random.seed(0)
    model = RegTR(cfg).to(dev).eval()
    # The default nn.Linear init makes the head's last layer so small that every predicted correspondence collapses onto one point
    # (spread 1.5 cm against 50 cm of key-point spread): the Kabsch covariance is then nearly rank one (s1 / (s2 + s3) 100 - 330 on
    # these pairs) and R amplifies the last-bit differences between ANY two float32 implementations by that factor -- one pair in eight
    # lands beyond 1e-4 on the pose with correspondences equal to 8e-7 (profiles/r04_a_bench_default_init.json).  "pose err vs ref" is
    # meant to measure the kernels, so the benchmark draws that one 3 x 256 matrix from U(-0.5, 0.5) (as oracle/seeded_weights.py does
    # for the goldens): predictions spread over metres, the Procrustes problem is well conditioned and the 1e-4 bar on R|t means what
    # it says.  Still random-init weights; the throughput does not depend on their values.
    # ModelNet-size pairs need more than spread (probe_head): there the output layer is a linear probe for the tokens' own coordinates.
    # round 6: the probe head everywhere.  The occlusion-calibrated rooms (synthetic.py) raised the uniform layer's Kabsch condition number from 9-29
    # to 17-48 and the all-pairs sweep found a pose at 7.8e-5 with correspondences at 8.7e-6 (tests/test_gpu_bench_batch.py): a 1.3 x margin on a
    # number that measures conditioning, not kernels.  `--head-init uniform` keeps the round-4/5 layer.
    head_init = head_init or 'probe'
    last = getattr(model.correspondence_decoder, 'coor_mlp', None)
    with torch.no_grad():
        if last is not None:
            last[4].weight.uniform_(-0.5, 0.5)
    if head_init == 'probe' and last is not None:
        gen_c = synth_modelnet_pair if config == 'modelnet' else (lambda i: synth_pair(i, points, shuffle, overlap='lomatch' if config == 'lomatch' else None))
        calib = real_pairs(4, 3) if real else [gen_c(900000 + i) for i in range(4)]       # calibration pairs: outside every benchmarked id range
        model.head_probe_r2 = probe_head(model, calib, dev)
    model.head_init = head_init
    base = rank * 100003 if first_id is None else first_id
    n_gen = n_pairs if not distinct else min(n_pairs, distinct)
    if real:
        gen = real_pairs(n_gen, base)
    elif config == 'modelnet':
        gen = [synth_modelnet_pair(base + i) for i in range(n_gen)]
    else:
        gen = [synth_pair(base + i, points, shuffle, overlap='lomatch' if config == 'lomatch' else None) for i in range(n_gen)]
    pairs = [gen[i % n_gen] for i in range(n_pairs)]
    dev_pairs = [(torch.from_numpy(s).to(dev), torch.from_numpy(t).to(dev)) for s, t in gen]
    batch = {'src_xyz': [dev_pairs[i % n_gen][0] for i in range(n_pairs)], 'tgt_xyz': [dev_pairs[i % n_gen][1] for i in range(n_pairs)]}
    return cfg, model, pairs, batch


def parity_slots(sizes, n):
    """Which slots of a forward's batch the parity check takes: the first and the last (packing offsets at both ends), the largest and
    the smallest pair (by points), then evenly spaced others up to `n`."""
    m = len(sizes)
    want = [0, m - 1, int(np.argmax(sizes)), int(np.argmin(sizes))]
    slots = []
    for sl in want + [int(round(i * (m - 1) / max(n, 1))) for i in range(1, n + 1)] + list(range(m)):
        if sl not in slots and len(slots) < min(n, m):
            slots.append(sl)
    return sorted(slots)




def forward_compulsory_bytes(model, level_points, H, n_tokens, d_embed=256):
    """Compulsory HBM bytes of ONE forward -- every array of the hot path touched once (SURVEY.md Appendix B's `compulsory MB` column, priced on the
    level sizes of THIS batch): per KPConv block  Nq H 4 (index table) + Ns (12 + 4 Cin) + Nq (12 + 4 Cout) + 15 Cin Cout 4, a strided block also
    its max-pooled shortcut's input  Ns 4 C_block_in;  preprocessing  24 N_l + 12 N_{l+1} per level (section 8d);  the token arrays of the
    cross-encoder read and written once (4 d per token each way) and its weights once.  The unary / InstanceNorm intermediates, the WF tensor and
    every re-read are NOT in it: that is the point of the ratio.  -> bytes"""
    total = 0.0
    for blk in model.kpf_encoder.encoder_blocks:
        kp = blk.KPConv
        strided = 'strided' in blk.block_name
        l = blk.layer_ind
        ns = level_points[l]
        nq = level_points[l + 1] if strided else ns
        cin, cout = kp.in_channels, kp.out_channels
        total += nq * H[l] * 4 + ns * (12 + 4 * cin) + nq * (12 + 4 * cout) + kp.K * cin * cout * 4
        if strided:
            total += ns * 4 * _block_in_dim(blk)
    for l in range(len(level_points) - 1):
        total += 24 * level_points[l] + 12 * level_points[l + 1]
    total += n_tokens * d_embed * 4 * 2
    total += sum(p.numel() * 4 for n, p in model.named_parameters() if not n.startswith('kpf_encoder'))
    return total


def _block_in_dim(blk):
    """Channels of a resnet block's input (the max-pooled shortcut of a strided block reads them, kpconv_blocks.py:734-737)."""
    u1 = getattr(blk, 'unary1', None)
    return u1.in_dim if hasattr(u1, 'in_dim') else blk.KPConv.in_channels


def forward_matrix_seconds(gemm_records, gemm_reps, gather_records, gather_reps, attention_flops, attention_terms):
    """Matrix-pipe time of ONE forward at the peaks: every dense launch 2 M N K x (terms issued) / 2.5 PFLOP/s (exact-f32 launches at 157.3 TFLOP/s),
    the KPConv gathers' correlation Nq H (150 + 30 Cin) flops on the f32 MFMA (SURVEY.md 8d's F_kp without the contraction, which is a dense
    launch), the attention core 4 d (Ns^2 + Nt^2 + 2 Ns Nt) x terms / 2.5 PFLOP/s.
    -> {'dense_s', 'gather_s', 'attention_s', 'algorithmic_flops', 'issued_flops'} per forward"""
    dense = sum(2.0 * m['M'] * m['N'] * m['K'] * (m['terms'] / (MFMA_BF16_PEAK_TFS * 1e12) if m['terms'] else 1.0 / (MFMA_F32_PEAK_TFS * 1e12))
                for _, _, m in gemm_records) / gemm_reps
    f_dense = sum(2.0 * m['M'] * m['N'] * m['K'] for _, _, m in gemm_records) / gemm_reps
    f_dense_issued = sum(2.0 * m['M'] * m['N'] * m['K'] * max(m['terms'], 1) for _, _, m in gemm_records) / gemm_reps
    f_gather = sum(r[3] * r[4] * (150.0 + 30.0 * r[5]) for r in gather_records) / gather_reps
    return {'dense_s': dense, 'gather_s': f_gather / (MFMA_F32_PEAK_TFS * 1e12), 'attention_s': attention_flops * attention_terms / (MFMA_BF16_PEAK_TFS * 1e12),
            'algorithmic_flops': f_dense + f_gather + attention_flops, 'issued_flops': f_dense_issued + f_gather + attention_flops * attention_terms}


class ReplicaRunner:
    """Concurrent forwards of one GPU (round 6): R model replicas (same weights) driven by R host threads on R HIP streams, replica r taking the
    chunks r, r + R, ... of the batch.  Why: a forward has two host waits (the pyramid's level sizes, the status word) and a 5-6 ms head (cell grid +
    level-0 conv table) during which nothing else of it can run, so ONE thread cannot keep the chip busy across forwards; a second forward in flight
    fills both.  Measured on one box (tools/concurrency_probe.py, profiles/r06_k_concurrency.txt): 2 x 192 pairs +4.1 %, 3 x 192 +4.7 %, 4 x 192 +3.1 %,
    3 x 128 +6.1 % over the same forwards one after the other -- poses bit-identical.  Replicas, not one shared module: a model owns its
    preprocessing workspaces.  `run(n)` lets every thread run its chunks n times WITHOUT meeting the others in between (free-running threads settle
    into a stagger; a join per step would start every forward's head at the same moment) and returns the outputs of the last pass per chunk."""

    def __init__(self, models, batch, chunks, device):
        import threading
        self._threading = threading
        self.models, self.batch, self.chunks, self.device = list(models), batch, list(chunks), torch.device(device)
        self.R = len(self.models)
        self.cuda = self.device.type == 'cuda'
        self.streams = [torch.cuda.Stream(self.device) for _ in range(self.R)] if (self.cuda and self.R > 1) else [None] * self.R

    def _chunk(self, lo, hi):
        return {'src_xyz': self.batch['src_xyz'][lo:hi], 'tgt_xyz': self.batch['tgt_xyz'][lo:hi]}

    def _work(self, r, n, outs, errs):
        try:
            mine = list(range(r, len(self.chunks), self.R))

            def loop():
                for _ in range(n):
                    for c in mine:
                        outs[c] = self.models[r](self._chunk(*self.chunks[c]))
            if self.streams[r] is not None:
                with torch.cuda.device(self.device), torch.cuda.stream(self.streams[r]):
                    loop()
            else:
                loop()
        except BaseException as e:          # re-raised on the calling thread
            errs.append(e)

    def run(self, n=1):
        """n passes over the batch -> [output dict of the last pass for chunk 0, 1, ...] (enqueued; the caller synchronises)."""
        outs, errs = [None] * len(self.chunks), []
        if self.R == 1:
            self._work(0, n, outs, errs)
        else:
            cur = torch.cuda.current_stream(self.device) if self.cuda else None
            for s in self.streams:
                if s is not None:
                    s.wait_stream(cur)                    # the inputs (and whatever the caller enqueued before)
            th = [self._threading.Thread(target=self._work, args=(r, n, outs, errs)) for r in range(self.R)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            for s in self.streams:
                if s is not None:
                    cur.wait_stream(s)                    # the caller's stream sees every replica's results
        if errs:
            raise errs[0]
        return outs

    def poses(self, outs):
        """(n_pairs, 3, 4) final-layer poses in batch order, safe to use on the current stream."""
        ps = [o['pose'][-1] for o in outs]
        p = ps[0] if len(ps) == 1 else torch.cat(ps)
        if self.cuda and self.R > 1:
            for x in ps:
                x.record_stream(torch.cuda.current_stream(self.device))
        return p


def replicate(model, cfg, n, device):
    """[model, n - 1 replicas with the same weights] (RegTR modules own their preprocessing workspaces: concurrent forwards need one module each)."""
    from regtr_amd import RegTR
    out = [model]
    for _ in range(n - 1):
        m = RegTR(cfg).to(device).eval()
        m.load_state_dict(model.state_dict())
        for attr in ('head_init', 'head_probe_r2', '_range_check'):
            if hasattr(model, attr):
                setattr(m, attr, getattr(model, attr))
        out.append(m)
    return out
