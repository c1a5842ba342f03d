"""One rank of the data-parallel ForwardTransformer training check (started by tests/test_gpu_dp_train.py under
torch.distributed.run, every rank on GPU 0 with the gloo backend).

Each rank trains on a batch of its own.  Checked: the gradient buffer after the step equals the sum over ranks of the
gradients the ranks compute alone on the same batches; the weights after the Adam update are identical on every rank,
also after a second step; the exchange was cut into more than one collective."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dp_common import done, stage  # noqa: E402

stage('process started')

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]

import estts  # noqa: E402,F401
from estts import parallel, weights  # noqa: E402

stage('imports done')
from estts.forward_training import ForwardTransformerTrainer  # noqa: E402
from test_gpu_training import batch, small_cfg  # noqa: E402


def gathered(t):
    h = t.detach().to('cpu')     # gloo sees host tensors only (its device path has stalled with two ranks on one GPU)
    out = [torch.empty_like(h) for _ in range(dist.get_world_size())]
    dist.all_gather(out, h)
    return [o.to(t.device) for o in out]


def main():
    rank, world, _ = parallel.env_ranks()
    torch.cuda.set_device(0)
    torch.cuda.init()
    stage('device set')
    parallel.init_distributed('gloo')
    stage('process group initialised')
    solo = [dist.new_group([r]) for r in range(world)][rank]
    cfg = small_cfg(0.0)
    w = weights.init_forward_transformer(cfg, seed=7)
    # ragged ranks: the ranks' batches pad to different token and frame counts; every rank divides its losses by the global
    # batch's padded size (global_lengths), as scripts/train_forward.py does
    data = [batch(cfg, 3, 19 + 4 * rank, seed=10 + rank), batch(cfg, 2, 23 - 3 * rank, seed=20 + rank)]
    shapes = [torch.tensor([[d[1].shape[1], d[0].shape[1]] for d in data])]
    all_shapes = gathered(shapes[0])
    stage('first all_gather done')
    glob = [tuple(int(v) for v in torch.stack(all_shapes)[:, i].max(dim=0).values) for i in range(2)]   # (T, N) per step
    assert len({int(t[0, 0]) for t in all_shapes}) == world        # the ranks really pad differently

    alone = ForwardTransformerTrainer(w, cfg, process_group=solo)
    assert alone.world == 1
    alone.rank = rank                                  # the same dropout stream as the data-parallel replica (rate 0 here)
    alone.train_step(*data[0], global_lengths=glob[0])
    torch.cuda.synchronize()
    stage('solo train_step done')
    expected = sum(gathered(alone.store.grad.cpu()))

    model = ForwardTransformerTrainer(w, cfg, bucket_mb=0.5)
    assert model.world == world and model.rank == rank
    assert model.store.flat.numel() > 2 * model.bucket_elems
    stage('data-parallel trainer built')
    model.train_step(*data[0], global_lengths=glob[0])
    got = model.store.grad.cpu()
    stage('first data-parallel train_step done')
    # two runs of one step differ a little (float atomics order the column sums differently, and a ReLU gate next to
    # its kink may then fall the other way), so the comparison is over the whole vector; a missing rank gives ~0.7
    scale = float(expected.norm())
    err = float((got - expected).norm())
    assert err <= 5e-3 * scale, (err, scale)
    different = float((got - alone.store.grad.cpu()).norm())
    assert different > 0.1 * scale, different          # the ranks really had different gradients
    model.train_step(*data[1], global_lengths=glob[1])
    stage('second data-parallel train_step issued')
    flats = gathered(model.store.flat.cpu())
    assert all(torch.equal(flats[0], f) for f in flats[1:])
    assert not torch.equal(flats[0], torch.from_numpy(np.asarray(alone.store.flat.cpu())))
    # per-replica BatchNorm statistics differ until a checkpoint broadcasts rank 0's
    states = gathered(model.store.state.cpu())
    assert not torch.equal(states[0], states[1])
    model.broadcast_moving_statistics(src=0)
    states = gathered(model.store.state.cpu())
    assert all(torch.equal(states[0], t) for t in states[1:])
    print(f'rank {rank}: data-parallel step OK (gradient error {err:.2e} of {scale:.2e})', flush=True)
    dist.barrier()
    dist.destroy_process_group()
    done()


if __name__ == '__main__':
    main()
