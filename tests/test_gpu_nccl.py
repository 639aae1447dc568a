"""GPU (-m gpu): the multi-rank entry path on REAL RCCL, as far as one GPU allows: bench.py under
torch.distributed.run with the "nccl" backend at world size 1 (RCCL initialisation with device_id=, the device-tensor
all_gather_into_tensor of the poses, barriers, the per-rank clocks), and two ranks sharing cuda:0 when RCCL permits that."""
import json
import os
import socket
import subprocess
import sys

import pytest

from tests.util import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _torchrun(nproc, script_args, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={nproc}', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port())] + script_args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_under_torchrun_nccl_world1():
    """The driver's N > 1 command line at N = 1: bench.py as a rank of a torch.distributed.run job with the RCCL backend.  The line's
    n_gpus counts ranks whose poses came THROUGH the collective; the parity gate runs as in a plain launch."""
    r = _torchrun(1, [os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1', '--pairs', '4', '--no-cpu-baseline',
                      '--no-roofline', '--parity-pairs', '2'])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 1 and line['value'] > 0 and line['parity']['ok'] and len(line['per_rank_ms_per_step']) == 1
    assert 'all_gather_into_tensor' in line['config']['parallelism']


def test_pose_gather_two_ranks_on_one_gpu_rccl():
    """Two RCCL ranks on cuda:0 (ragged shards: 7 pairs -> 4 + 3): the ONE-collective pose gather on device tensors between real
    ranks.  RCCL may refuse several ranks per device -- then this skips, loudly, with RCCL's own message."""
    r = _torchrun(2, [os.path.join(ROOT, 'tests', 'nccl_worker.py'), '7'], timeout=300)
    if r.returncode != 0 and ('exitcode: 77' in r.stderr or 'exitcode  : 77' in r.stderr or 'Duplicate GPU' in r.stdout + r.stderr):
        why = [l for l in (r.stdout + r.stderr).splitlines() if 'rank' in l and ('Error' in l or 'error' in l or 'Duplicate' in l)][:2]
        pytest.skip(f'RCCL refuses two ranks on one GPU here: {why}')
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count('gathered 7 poses over RCCL') == 2


def test_pose_gather_one_rank_rccl():
    r = _torchrun(1, [os.path.join(ROOT, 'tests', 'nccl_worker.py'), '5'], timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert 'gathered 5 poses over RCCL' in r.stdout
