"""compute_rigid_transform drop-in (/root/reference/src/utils/se3_torch.py:108-154) on the fused Procrustes kernel."""
import torch

from . import ops


def compute_rigid_transform(a: torch.Tensor, b: torch.Tensor, weights: torch.Tensor = None):
    """a, b ([*,] N, 3), weights ([*,] N) in [0, 1] -> ([*,] 3, 4) with T a = b (weighted Kabsch).

    The kernel consumes the model's native layout (key points, predicted coordinates, overlap LOGITS); this generic
    entry point re-expresses its arguments in that layout: a is passed as the 'key points' of a pure-source pair,
    b as its predicted coordinates, and logit = log(w / (1 - w))."""
    assert a.shape == b.shape and a.shape[-1] == 3
    batch_shape = a.shape[:-2]
    N = a.shape[-2]
    a2 = a.reshape(-1, N, 3).to(torch.float32).contiguous()
    b2 = b.reshape(-1, N, 3).to(torch.float32).contiguous()
    P = a2.shape[0]
    if weights is None:
        w = torch.full((P, N), 0.5, dtype=torch.float32, device=a.device)
    else:
        assert a.shape[:-1] == weights.shape
        w = weights.reshape(P, N).to(torch.float32)
    logit = torch.log(w) - torch.log1p(-w)
    # P independent "pairs", each with N source rows and 0 target rows, one "layer"
    seg = torch.cat([torch.arange(P + 1, dtype=torch.int32) * N,
                     torch.full((P,), P * N, dtype=torch.int32)]).to(a.device)
    pose = ops.weighted_procrustes(a2.view(P * N, 3), b2.view(1, P * N, 3), logit.view(1, P * N).contiguous(), seg, P)
    return pose[0].reshape(*batch_shape, 3, 4)
