"""-m gpu, needs >= 2 GPUs (self-skips on the 1-GPU box): the data-parallel step over RCCL — two processes, one per GPU, launched the
way bench.py is (`python -m torch.distributed.run --nproc-per-node 2`), gradient all-reduce through gg_comm_* on the side stream:
replicas stay bit-identical over plain and gradient-penalty steps with hipGraph replay, ncclCommCount == world."""
import os
import subprocess
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def test_two_rank_rccl_training_keeps_replicas_identical():
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29531', str(ROOT / 'tests' / 'rccl_two_rank_worker.py')]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count('replicas identical True') == 2
