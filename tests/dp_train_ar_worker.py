"""One rank of the data-parallel AutoregressiveTransformer training check (started by tests/test_gpu_dp_train.py under
torch.distributed.run, every rank on GPU 0 with the gloo backend): the gradient buffer after a step equals the sum over ranks
of the gradients the ranks compute alone on their own batches, and the weights stay identical across ranks after the update."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dp_common import done, stage  # noqa: E402

stage('process started')

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]

import estts  # noqa: E402,F401
from estts import parallel, weights  # noqa: E402

stage('imports done')
from estts.autoregressive_training import AutoregressiveTransformerTrainer  # noqa: E402
from test_gpu_autoregressive_training import batch, small_cfg  # noqa: E402


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
    cfg = small_cfg('speaker_style_text', 0.0)
    w = weights.init_autoregressive_transformer(cfg, seed=3)
    # ragged ranks: the shards pad to different frame counts and every rank divides its losses by the global batch's
    # padded size (global_frames), as scripts/train_autoregressive.py does
    tokens, mel, stop, spk = batch(cfg, 3, 13, 20 + 4 * rank, seed=40 + rank)
    frames = gathered(torch.tensor([mel.shape[1] - 1]))
    Tg = int(torch.stack(frames).max())
    assert len({int(f) for f in frames}) == world
    stage('first all_gather done')

    def make(group):
        m = AutoregressiveTransformerTrainer(w, cfg, process_group=group) if group is not None else AutoregressiveTransformerTrainer(w, cfg)
        m._compile(stop_scaling=8.0, optimizer={'learning_rate': 1e-3})
        m.set_constants(reduction_factor=2)
        return m

    alone = make(solo)
    assert alone.world == 1
    alone.rank = rank
    alone.train_step(mel, tokens, stop, spk, apply=False, global_frames=Tg)
    torch.cuda.synchronize()
    stage('solo train_step done')
    expected = sum(gathered(alone.store.grad.cpu()))
    model = make(None)
    assert model.world == world
    model.train_step(mel, tokens, stop, spk, global_frames=Tg)
    got = model.store.grad.cpu()
    stage('data-parallel train_step done')
    assert len(model._last_overlap.calls) >= 1
    scale, err = float(expected.norm()), float((got - expected).norm())
    assert err <= 5e-3 * scale, (err, scale)
    assert float((got - alone.store.grad.cpu()).norm()) > 0.1 * scale
    flats = gathered(model.store.flat.cpu())
    assert all(torch.equal(flats[0], f) for f in flats[1:])
    print(f'rank {rank}: autoregressive data-parallel step OK (gradient error {err:.2e} of {scale:.2e})', flush=True)
    dist.barrier()
    dist.destroy_process_group()
    done()


if __name__ == '__main__':
    main()
