"""The N > 1 host path on CPU: two gloo ranks shard a list of utterances, agree on a max timing and gather results."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

import estts
from estts import parallel


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    import estts  # noqa: F401  (spawned interpreter)
    from estts import parallel as par
    r, w, _ = par.init_distributed('gloo')
    frames = [560, 120, 870, 400, 400, 333, 640, 200, 555]
    mine = par.shard_by_cost(frames, w)[r]
    par.barrier()
    t = par.max_over_ranks(1.0 + r)
    got = par.gather_to_rank0({'rank': r, 'items': mine, 'frames': sum(frames[i] for i in mine)})
    # data-parallel gradient exchange: flat buffer summed over ranks in buckets (odd sizes, last bucket short)
    g = torch.arange(1003, dtype=torch.float32) * (r + 1)
    par.allreduce_buckets(g, 250)
    ok = bool(torch.equal(g, torch.arange(1003, dtype=torch.float32) * 3))
    # the same exchange overlapped with a backward pass: tails reported as they become final, every element summed once
    g2 = torch.arange(1003, dtype=torch.float32) * (r + 1)
    ov = par.GradientOverlap(g2, 250, min_elems=100)
    for lo in (990, 900, 700, 650, 120):     # 990 and 650 are held back (fewer than 100 new elements)
        ov.ready(lo)
    ov.finish()
    ok = ok and bool(torch.equal(g2, torch.arange(1003, dtype=torch.float32) * 3))
    covered = sorted(ov.calls)
    ok = ok and covered[0][0] == 0 and covered[-1][1] == 1003 and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    ok = ok and all(b - a <= 250 for a, b in covered) and (990, 1003) not in covered
    if r == 0:
        q.put((t, got, ok))
    torch.distributed.destroy_process_group()


def test_two_rank_sharding_and_reductions():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    t, got, grads_ok = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert t == 2.0
    assert grads_ok
    items = sorted(i for g in got for i in g['items'])
    assert items == list(range(9))                       # every utterance exactly once
    loads = [g['frames'] for g in got]
    assert abs(loads[0] - loads[1]) <= 120               # balanced to within the shortest utterance


def test_shard_by_cost_is_deterministic_and_complete():
    costs = list(np.random.default_rng(0).integers(100, 900, size=64))
    a = parallel.shard_by_cost(costs, 8)
    assert a == parallel.shard_by_cost(costs, 8)
    assert sorted(i for s in a for i in s) == list(range(64))
    loads = [sum(costs[i] for i in s) for s in a]
    assert max(loads) - min(loads) < max(costs)
    assert parallel.max_over_ranks(3.5) == 3.5 and parallel.gather_to_rank0('x') == ['x']
    assert parallel.bucket_ranges(10, 4) == [(0, 4), (4, 8), (8, 10)]
    g = torch.ones(5)
    parallel.allreduce_buckets(g, 2)                      # not distributed: untouched
    assert torch.equal(g, torch.ones(5))


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it must start two ranks (one child job under torch.distributed.run)
    and print ONE line with n_gpus == 2.  On a host without a GPU the workload that can run is the gradient exchange on host
    memory over gloo (the hot path itself has no CPU implementation); the launch path is the same for every workload."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['ESTTS_DIST_BACKEND'] = 'gloo'
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--workload', 'exchange', '--elems', '300000',
                        '--steps', '3', '--warmup', '1'], capture_output=True, text=True, timeout=240, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['sum_correct'] is True and line['config']['backend'] == 'gloo'
    assert line['steps'] == 3 and line['value'] > 0
    # a launcher that starts another number of ranks than --gpus says is an error, not a silently different run
    env2 = dict(env, RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--workload', 'exchange', '--elems', '1000',
                        '--steps', '1', '--warmup', '0'], capture_output=True, text=True, timeout=120, env=env2, cwd=root)
    assert r.returncode != 0 and '--gpus 2' in (r.stderr + r.stdout)
