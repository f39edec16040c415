"""In-process emulation of an N-rank sequence-parallel group on ONE device: N Python threads, one per emulated rank, run the real
model / kernels on their true shards and exchange tensors through shared memory behind a `threading.Barrier`.  RCCL will not put
two ranks on one GPU, so this is how the N-rank SCHEDULE (sharding, RoPE offsets, K / V^T segments, local-first merge, Ulysses
head split, final gather) is exercised through the production kernels on the 1-GPU boxes (tests/test_round2_gpu.py), and what
`tools/bench_shard.py` uses as its communication-free stand-in.  All threads enqueue on the same stream, so a consumer kernel
launched after the barrier is ordered behind every producer's kernels."""
import threading

import torch

from . import SequenceParallelGroup


class _Shared:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world


class ThreadGroup(SequenceParallelGroup):
    def __init__(self, shared, rank):
        self.shared, self.group, self.world_size, self.rank = shared, None, shared.world, rank

    def _exchange(self, x):
        sh = self.shared
        sh.slots[self.rank] = x
        sh.barrier.wait()
        parts = list(sh.slots)
        sh.barrier.wait()
        return parts

    def all_gather(self, x, dim=1):
        return torch.cat(self._exchange(x.contiguous()), dim=dim)

    def gather_start(self, x):
        x = x.contiguous()
        return torch.stack(self._exchange(x)), None, x

    def all_to_all(self, x):
        parts = self._exchange(x.contiguous())
        return torch.stack([parts[j][self.rank] for j in range(self.world_size)])


def run_ranks(world, fn):
    """fn(group) on `world` threads; returns the per-rank results (re-raises the first exception)."""
    shared = _Shared(world)
    out, err = [None] * world, [None] * world

    def work(r):
        try:
            out[r] = fn(ThreadGroup(shared, r))
        except BaseException as ex:       # noqa: BLE001 - reported to the caller below
            err[r] = ex
            shared.barrier.abort()
    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for e in err:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    for e in err:
        if e is not None:
            raise e
    return out
