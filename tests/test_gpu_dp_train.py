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


def test_bench_gpus_2_yields_a_two_rank_line_with_the_training_block():
    """`python bench.py --gpus 2` alone (no launcher) must start two ranks and print one line with n_gpus == 2 that carries the
    forward replicas and the data-parallel training block (config 5's step at a small size here); on a one-GPU box the ranks
    share GPU 0 and exchange over gloo."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(ESTTS_DIST_BACKEND='gloo', ESTTS_WATCHDOG_S='200')
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--steps', '5', '--warmup', '2', '--tokens', '30', '--frames', '90',
                        '--train-batch', '2'], cwd=ROOT, capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['scaling'] == 'weak' and line['value'] > 0
    assert line['config']['global_batch'] == 2 and 'replicas x2' in line['config']['parallelism']
    train = line['train']
    assert 'error' not in train, train
    assert train['global_batch'] == 4 and train['steps_per_s'] > 0 and train['launches_per_step'] > 100
    ex = train['exchange']
    assert ex['ranks_seen'] == 2 and ex['backend'] == 'gloo' and ex['buckets'] >= 3 and ex['allreduce_ms'] > 0
    assert ex['exposed_ms'] >= 0 and ex['hidden_ms'] >= 0 and ex['gradient_bytes'] > 70e6
    assert train['loss_last_step'] < train['loss_first_step']
