"""Data-parallel gradient reduction over RCCL / xGMI (one process per GPU).

The hot path shards over (sentence, image) pairs: every rank holds a full weight replica and
processes its own micro-batch; the only exchange is one gradient all-reduce (mean) per optimisation step -- the
counterpart of the reference's apex DistributedDataParallel (My_cross_attention.py:768-776).

Because all gradients live in ONE flat fp32 buffer (ParamArena.gflat), laid out in execution order, the reduction is
a handful of large contiguous all-reduces instead of ~200 small ones: buckets are slices of that buffer, taken from
its END (the parameters whose gradients the backward pass finishes first) towards its start.  ``bucket_ready`` lets
the backward launch a bucket's all-reduce on a side stream as soon as every gradient in it is final, overlapping
RCCL traffic over xGMI with the remaining backward kernels; ``finish`` joins the streams.
Works with the ``nccl`` (= RCCL) backend on ROCm devices and with ``gloo`` on CPU tensors (tests).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

from .arena import ParamArena


class GradReducer(object):
    def __init__(self, arena: ParamArena, group=None, bucket_mb: float = 64.0, comm_bf16: bool = False):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.arena = arena
        self.group = group
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.buckets: List[Tuple[int, int]] = arena.buckets(int(bucket_mb * (1 << 20) / 4))
        self.comm_bf16 = comm_bf16 and arena.device.type == "cuda"
        self.is_cuda = arena.device.type == "cuda"
        self.comm_stream = torch.cuda.Stream(device=arena.device) if self.is_cuda else None
        # slot -> bucket index, and per-bucket count of slots whose gradient is not final yet (this step)
        self._bucket_of = {}
        self._nslots = [0] * len(self.buckets)
        for s in arena.order:
            for bi, (lo, hi) in enumerate(self.buckets):
                if lo <= s.off < hi:
                    self._bucket_of[id(s)] = bi
                    self._nslots[bi] += 1
                    break
        self._remaining = list(self._nslots)
        self._launched = [False] * len(self.buckets)
        self._marked = set()      # slots that received a gradient during the current step
        self._calibrated = False  # after the first step only slots that actually get gradients are waited for
        self._stage = None
        if self.comm_bf16:
            self._stage = torch.empty(max(e - s for s, e in self.buckets), dtype=torch.bfloat16, device=arena.device)

    # -------------------------------------------------------------------------------------------------
    def broadcast_parameters(self, src: int = 0) -> None:
        """Make every replica start from rank ``src``'s parameters (one broadcast of the flat buffer)."""
        dist.broadcast(self.arena.flat, src=src, group=self.group)
        self.arena.mark_dirty()

    def _launch(self, idx: int) -> None:
        s, e = self.buckets[idx]
        buf = self.arena.gflat[s:e]
        if self.is_cuda:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self._allreduce(buf)
        else:
            self._allreduce(buf)

    def _allreduce(self, buf: torch.Tensor) -> None:
        if self.backend == "nccl":
            if self.comm_bf16:
                from . import kernels as K
                st = self._stage[:buf.numel()]
                K.cast_f32_to_bf16(buf, st)
                dist.all_reduce(st, op=dist.ReduceOp.AVG, group=self.group)
                K.cast_bf16_to_f32(st, buf)
            else:
                dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.group)
        else:  # gloo has no AVG
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
            buf.mul_(1.0 / self.world)

    def mark_final(self, slots) -> None:
        """Called from backward (ParamArena.flush_final): the gradients of these slots are final for this step.
        A bucket whose slots are all final is all-reduced right away on the side stream (overlap with the rest of
        backward).  Order-independent: nothing is assumed about the order autograd runs the blocks in."""
        for s in slots:
            if id(s) in self._marked:
                continue
            self._marked.add(id(s))
            bi = self._bucket_of[id(s)]
            self._remaining[bi] -= 1
            if self._remaining[bi] == 0 and not self._launched[bi]:
                self._launched[bi] = True
                self._launch(bi)

    def finish(self) -> None:
        """Launch every bucket not launched yet (parameters that got no gradient this step keep their bucket
        waiting until here) and make the compute stream wait for the reductions."""
        for bi in range(len(self.buckets)):
            if not self._launched[bi]:
                self._launch(bi)
        if self.is_cuda:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        if not self._calibrated:
            # parameters that never receive a gradient (e.g. the pooler when only logits are used) must not keep
            # their bucket waiting until finish(): from now on wait only for the slots seen in this first step
            self._nslots = [0] * len(self.buckets)
            for sid in self._marked:
                self._nslots[self._bucket_of[sid]] += 1
            self._calibrated = True
        self._marked = set()
        self._remaining = list(self._nslots)
        self._launched = [n == 0 and False for n in self._nslots]

    def reduce_all(self) -> None:
        """Non-overlapped form: all-reduce every bucket now (after backward)."""
        self.finish()
