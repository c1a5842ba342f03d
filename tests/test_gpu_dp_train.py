"""Data-parallel ForwardTransformer training with two ranks.  RCCL needs one GPU per rank, so on a one-GPU box the two
ranks share GPU 0 and exchange through gloo: the same bucketed exchange, sharding and update code runs, the transport
differs (and with gloo the exchange follows the backward pass instead of overlapping it)."""
import os
import signal
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


def run_two_ranks(worker, marker):
    """Starts the worker on two ranks under torch.distributed.run and returns its output.  The job takes a few seconds; one
    in a few dozen starts has been seen to stall before the first step (both ranks share one GPU and rendezvous over a local
    TCP store), so a start that is silent for 100 s is ended -- the whole process group -- and started once more on a new port;
    a second stall fails the test with what the ranks had printed."""
    last = ''
    for attempt in range(2):
        cmd = [sys.executable, '-u', '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
               '--master-port', str(free_port()), os.path.join('tests', worker)]
        p = subprocess.Popen(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
        try:
            out, _ = p.communicate(timeout=100)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, signal.SIGKILL)
            out, _ = p.communicate()
            last = f'attempt {attempt}: no result within 100 s\n{out[-3000:]}'
            print(last, flush=True)
            continue
        assert p.returncode == 0, out[-6000:]
        assert out.count(marker) == 2, out[-2000:]
        return out
    raise AssertionError(last)


def test_two_ranks_sum_gradients_and_stay_in_step():
    run_two_ranks('dp_train_worker.py', 'data-parallel step OK')


def test_two_ranks_autoregressive_trainer():
    run_two_ranks('dp_train_ar_worker.py', 'autoregressive data-parallel step OK')
