"""`world` data-parallel ranks as threads of ONE process on ONE GPU, for the sharded optimiser's GPU test: ShardedExchange with its three
collectives replaced by in-process ones (every rank deposits its tensor, a barrier, every rank reads the others' — all on one stream, so the
device sees the operations in the order the threads issued them). The real collectives are covered by tests/test_sharded_optimizer.py
(gloo, world 2 / 4) and tests/helpers/rccl_world1.py (RCCL)."""
import threading

import torch

from text2nerf_amd.parallel import ShardedExchange


class VirtualGroup:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.turn = threading.Lock()       # one rank at a time from "seed the CPU generator" to "phase 1 submitted"


class VirtualExchange(ShardedExchange):
    @classmethod
    def make(cls, field, vg, rank):
        ex = cls.for_field(field, None, world=vg.world, rank=rank)
        ex.vg = vg
        return ex

    def _deposit(self, t):
        self.vg.slots[self.rank] = t
        self.vg.barrier.wait()
        return list(self.vg.slots)

    def _done(self):
        self.vg.barrier.wait()

    def _reduce_scatter_avg(self, own, body):
        bodies = self._deposit(body)
        n = own.numel()
        acc = torch.stack([b[self.rank * n:(self.rank + 1) * n] for b in bodies]).sum(0) / self.world
        self._done()                       # every rank has READ the deposited slices before anybody overwrites its own
        own.copy_(acc)
        self._done()

    def _all_reduce_avg(self, t):
        ts = self._deposit(t)
        acc = torch.stack(ts).sum(0) / self.world
        self._done()
        t.copy_(acc)
        self._done()

    def _all_gather(self, body, own):
        owns = self._deposit(own)
        n = own.numel()
        for r, o in enumerate(owns):
            if r != self.rank:
                body[r * n:(r + 1) * n].copy_(o)
        self._done()

    def reduce(self, fused_step=None):
        if self.vg.turn.locked():
            try:
                self.vg.turn.release()
            except RuntimeError:
                pass
        super().reduce(fused_step)
