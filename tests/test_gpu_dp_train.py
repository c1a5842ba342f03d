"""Data-parallel ForwardTransformer training with two ranks.  RCCL needs one GPU per rank, so on a one-GPU box the two
ranks share GPU 0 and exchange through gloo: the same bucketed exchange, sharding and update code runs, the transport
differs (and with gloo the exchange follows the backward pass instead of overlapping it)."""
import os
import signal
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_two_ranks(worker, marker, extra_env=None):
    """Starts the worker on two ranks under torch.distributed.run and returns its output.  `--standalone`: the launcher's
    agent binds its rendezvous store to a port of its own choosing and the ranks talk to that store, so no port number is
    picked here, released and bound again by somebody else (a start that stalled before the first step was once seen with a
    port found that way).  One attempt: the workers log every stage with a time stamp and dump their stacks after 60 s
    without progress, so a job that does not finish within 150 s fails with the stage and frame it stopped in."""
    cmd = [sys.executable, '-u', '-m', 'torch.distributed.run', '--standalone', '--nnodes=1', '--nproc-per-node', '2',
           '--local-addr', '127.0.0.1', os.path.join('tests', worker)]
    env = dict(os.environ)
    env.update(extra_env or {})
    p = subprocess.Popen(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True, env=env)
    try:
        out, _ = p.communicate(timeout=150)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        out, _ = p.communicate()
        raise AssertionError(f'no result within 150 s\n{out[-8000:]}')
    assert p.returncode == 0, out[-8000:]
    assert out.count(marker) == 2, out[-3000:]
    return out


def test_two_ranks_sum_gradients_and_stay_in_step():
    run_two_ranks('dp_train_worker.py', 'data-parallel step OK')


def test_two_ranks_autoregressive_trainer():
    run_two_ranks('dp_train_ar_worker.py', 'autoregressive data-parallel step OK')
