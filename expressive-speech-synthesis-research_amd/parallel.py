"""One process per GPU: rendezvous from the torchrun environment, utterance sharding, rank reductions, and the
gradient exchange of data-parallel training.

Inference on this path shards by utterance (WaveRNN additionally by fold) with no data-path collective;
collectives are used only to agree on timings / gather results.  Training exchanges one flat gradient buffer per step
(sum over ranks) in a few large buckets.  Backend 'nccl' is RCCL on ROCm; 'gloo' is used by the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def env_ranks():
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('LOCAL_RANK', '0')))


def init_distributed(backend=None):
    """Initialise torch.distributed when WORLD_SIZE > 1.  Returns (rank, world, local_rank)."""
    rank, world, local_rank = env_ranks()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        kw = {}
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
            kw['device_id'] = torch.device('cuda', local_rank)
        dist.init_process_group(backend, **kw)
    return rank, world, local_rank


def shard_by_cost(costs, world):
    """Greedy longest-first partition of items with the given costs (e.g. frames per utterance) into `world` shards of
    near-equal total cost.  Returns a list of index lists, deterministic on every rank."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    shards, load = [[] for _ in range(world)], [0] * world
    for i in order:
        r = min(range(world), key=lambda j: (load[j], j))
        shards[r].append(i)
        load[r] += costs[i]
    return [sorted(s) for s in shards]


def max_over_ranks(value, device=None):
    """Maximum of a Python float over all ranks (identity when not distributed)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_to_rank0(obj):
    """Gather one picklable object per rank on rank 0 (list in rank order; None elsewhere)."""
    if not (dist.is_available() and dist.is_initialized()):
        return [obj]
    out = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(obj, out, dst=0)
    return out


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def _staged(t, group=None):
    """True for a device tensor on a backend other than RCCL (gloo: two ranks rehearsing on one GPU)."""
    return t.is_cuda and dist.get_backend(group) != 'nccl'


def all_reduce_sum_(t, group=None):
    """In-place sum of `t` over all ranks, blocking.  Over gloo a device tensor goes through a host copy made HERE and the
    host tensor is what gloo sees: ProcessGroupGloo's own device path (staging threads and streams of its own) has stalled for
    good with two processes on one GPU -- two collectives in flight behind queued kernels at first
    (tools/gloo_concurrent.py), later a single one, both ranks inside the same all-reduce with two of gloo's threads in an
    uninterruptible driver call -- and the rehearsal transport is not what is being rehearsed."""
    if _staged(t, group):
        h = t.detach().to('cpu')
        dist.all_reduce(h, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, group=group)


def broadcast_(t, src=0, group=None):
    """In-place broadcast from rank `src`, blocking; device tensors over gloo through a host copy (see all_reduce_sum_)."""
    if _staged(t, group):
        h = t.detach().to('cpu')
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src, group=group)


def bucket_ranges(n, bucket_elems):
    """[(start, end)] covering range(n) in pieces of at most bucket_elems elements."""
    bucket_elems = max(int(bucket_elems), 1)
    return [(o, min(o + bucket_elems, n)) for o in range(0, n, bucket_elems)]


def allreduce_buckets(flat, bucket_elems, group=None, comm_stream=None):
    """Sums the 1-D tensor `flat` over all ranks in place, one asynchronous all-reduce per bucket (xGMI rings are
    per-link bound, so a handful of ~25 MB buckets keeps every link busy without the latency of many small calls).
    On a GPU the collectives are issued on `comm_stream` (which first waits for the current stream) and the current
    stream waits for them at the end; on the CPU (gloo) they are simply waited for, and a device tensor over gloo goes
    bucket by bucket through host copies (all_reduce_sum_).  No-op when not distributed."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    ranges = bucket_ranges(flat.numel(), bucket_elems)
    if _staged(flat, group):   # gloo on device memory (two ranks rehearsing on one GPU): bucket by bucket through the host
        torch.cuda.current_stream(flat.device).synchronize()
        for a, b in ranges:
            all_reduce_sum_(flat[a:b], group)
    elif flat.is_cuda and comm_stream is not None:
        cur = torch.cuda.current_stream(flat.device)
        comm_stream.wait_stream(cur)
        with torch.cuda.stream(comm_stream):
            works = [dist.all_reduce(flat[a:b], group=group, async_op=True) for a, b in ranges]
            for w in works:
                w.wait()
        cur.wait_stream(comm_stream)
    else:
        works = [dist.all_reduce(flat[a:b], group=group, async_op=True) for a, b in ranges]
        for w in works:
            w.wait()


class GradientOverlap:
    """Sums a flat gradient buffer over all ranks while the backward pass is still running.

    The buffer is laid out in forward order, so the backward pass finishes it from the tail towards the head: after
    every stage the caller reports the lowest offset that is final (`ready(lo)`), and the finished tail is all-reduced in
    buckets on the communication stream (ordered behind the kernels issued so far) while the compute stream goes on with
    the earlier layers.  `finish()` sends what is left and makes the current stream wait for every collective.
    Tails smaller than `min_elems` are held back until more is ready (xGMI rings are per-link bound: few, large calls).
    Every rank must report the same offsets in the same order.  No-op when not distributed."""

    def __init__(self, flat, bucket_elems, group=None, comm_stream=None, min_elems=None):
        self.flat, self.bucket, self.group, self.comm = flat, bucket_elems, group, comm_stream
        self.min_elems = bucket_elems // 4 if min_elems is None else min_elems
        self.active = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        # gloo stages device buffers through host threads of its own; collectives issued while kernels are still queued
        # stalled two ranks sharing one GPU, so with gloo on device memory the whole exchange waits for finish() and then
        # goes out bucket by bucket through host copies made here
        self.defer = (self.active and flat.is_cuda and dist.get_backend(group) != 'nccl'
                      and os.environ.get('ESTTS_GLOO_OVERLAP', '0') != '1')     # (diagnostic switch: overlap over gloo anyway)
        self.final = False
        self.hi = flat.numel()
        self.works = []
        self.calls = []          # (lo, hi) of every collective issued, for tests
        GradientOverlap.last = self      # (diagnostics: a watchdog can ask which collectives are still pending)

    def ready(self, lo):
        lo = max(0, min(int(lo), self.hi))
        if not self.active or lo >= self.hi or (lo > 0 and self.hi - lo < self.min_elems):
            return
        if self.defer and not self.final:
            return
        ranges = [(lo + a, lo + b) for a, b in bucket_ranges(self.hi - lo, self.bucket)]
        if self.flat.is_cuda and self.comm is not None:
            self.comm.wait_stream(torch.cuda.current_stream(self.flat.device))
            with torch.cuda.stream(self.comm):
                self.works += [dist.all_reduce(self.flat[a:b], group=self.group, async_op=True) for a, b in ranges]
        elif self.defer:   # gloo on device memory: bucket by bucket through the host (see all_reduce_sum_)
            for a, b in ranges:
                all_reduce_sum_(self.flat[a:b], self.group)
        else:
            self.works += [dist.all_reduce(self.flat[a:b], group=self.group, async_op=True) for a, b in ranges]
        self.calls += ranges
        self.hi = lo

    def finish(self):
        if not self.active:
            return
        self.min_elems = 0
        if self.defer:
            self.final, self.comm = True, None
            torch.cuda.current_stream(self.flat.device).synchronize()
        self.ready(0)
        if self.flat.is_cuda and self.comm is not None:
            with torch.cuda.stream(self.comm):
                for w in self.works:
                    w.wait()
            torch.cuda.current_stream(self.flat.device).wait_stream(self.comm)
        else:
            for w in self.works:
                w.wait()
        self.works = []
