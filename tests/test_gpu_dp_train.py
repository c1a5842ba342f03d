"""Data-parallel ForwardTransformer training with two ranks.  RCCL needs one GPU per rank, so on a one-GPU box the two
ranks share GPU 0 and exchange through gloo: the same bucketed exchange, sharding and update code runs, the transport
differs (and with gloo the exchange follows the backward pass instead of overlapping it)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_two_ranks_sum_gradients_and_stay_in_step():
    cmd = [sys.executable, '-u', '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.join('tests', 'dp_train_worker.py')]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count('data-parallel step OK') == 2, r.stdout[-2000:]


def test_two_ranks_autoregressive_trainer():
    cmd = [sys.executable, '-u', '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.join('tests', 'dp_train_ar_worker.py')]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count('autoregressive data-parallel step OK') == 2, r.stdout[-2000:]
