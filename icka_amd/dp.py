"""Data-parallel gradient reduction over RCCL / xGMI (one process per GPU).

The hot path shards over (sentence, image) pairs: every rank holds a full weight replica and
processes its own micro-batch; the only exchange is one gradient all-reduce (mean) per optimisation step -- the
counterpart of the reference's apex DistributedDataParallel (My_cross_attention.py:768-776).

Because all gradients live in ONE flat fp32 buffer (ParamArena.gflat), laid out in execution order, the reduction is
a handful of large contiguous all-reduces instead of ~200 small ones: buckets are slices of that buffer, taken from
its END (the parameters whose gradients the backward pass finishes first) towards its start, with the embedding
tables (whose gradient is final only at the very end of backward) in a bucket of their own.  A bucket's all-reduce is
launched on a side stream as soon as every gradient in it is final (``ParamArena.flush_final`` -> ``mark_final``),
overlapping RCCL traffic over xGMI with the remaining backward kernels; ``finish`` joins the streams.

Wire format: with more than one rank on ROCm devices the buckets travel as **bf16** by default (cast -> all-reduce AVG
-> cast back, all on the side stream): 239 MB instead of 477 MB per step for the bert-base path, which is what puts the
exchange under the backward it overlaps with (budget in DESIGN.md section 6).  ``comm_dtype="f32"`` keeps fp32 buckets.

A slot counts as final when it has received ALL the gradient contributions it received in the calibration (first)
step -- a module applied twice per forward contributes twice -- so a bucket is never reduced while a late write into it
is still to come.  Works with the ``nccl`` (= RCCL) backend on ROCm devices and with ``gloo`` (CPU tensors; device
tensors are staged through the host: tests only).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from .arena import ParamArena


import os as _os
_DIAG = _os.environ.get("ICKA_DP_DIAG", "")   # "", or any of "none" / "cast" / "comm" (comma-separated): see _allreduce


class GradReducer(object):
    def __init__(self, arena: ParamArena, group=None, bucket_mb: float = 64.0, comm_dtype: Optional[str] = None,
                 comm_bf16: Optional[bool] = None):
        """``bucket_mb``: minimum bucket size in MB of fp32 gradients.  ``comm_dtype``: "bf16" | "f32" | None (= bf16
        when the world has more than one rank and the arena is on a device, else f32).  ``comm_bf16`` is the round-1
        spelling of the same switch."""
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.arena = arena
        self.group = group
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.is_cuda = arena.device.type == "cuda"
        if comm_dtype is None and comm_bf16 is not None:
            comm_dtype = "bf16" if comm_bf16 else "f32"
        if comm_dtype is None:
            comm_dtype = "bf16" if (self.world > 1 and self.is_cuda) else "f32"
        if comm_dtype not in ("bf16", "f32"):
            raise ValueError("comm_dtype must be 'bf16' or 'f32'")
        if comm_dtype == "bf16" and not self.is_cuda:
            raise ValueError("bf16 buckets need a ROCm device (the casts are HIP kernels)")
        self.comm_bf16 = comm_dtype == "bf16"
        self.buckets: List[Tuple[int, int]] = arena.buckets(int(bucket_mb * (1 << 20) / 4))
        self.comm_stream = torch.cuda.Stream(device=arena.device) if self.is_cuda else None
        # slot -> bucket index
        self._bucket_of: Dict[int, int] = {}
        for s in arena.order:
            for bi, (lo, hi) in enumerate(self.buckets):
                if lo <= s.off < hi:
                    self._bucket_of[id(s)] = bi
                    break
        self._calibrated = False
        self._expected: Dict[int, int] = {}       # slot -> gradient writes per step (counted in the calibration step)
        self._seen: Dict[int, int] = {}
        self._waiting = [0] * len(self.buckets)   # per bucket: slots that have not received all their writes yet
        self._launched = [False] * len(self.buckets)
        # segmented hipGraph capture (graph.SegmentedStep): while ``capture`` is set, a bucket that becomes ready is not
        # launched but reported -- the capture cuts the graph there and the all-reduce is issued eagerly between the
        # replayed segments, so no collective and no cross-stream edge ever sits inside a captured graph
        self.capture = None
        self._stage = None
        if self.comm_bf16:
            self._stage = torch.empty(max(e - s for s, e in self.buckets), dtype=torch.bfloat16, device=arena.device)

    # -------------------------------------------------------------------------------------------------
    def broadcast_parameters(self, src: int = 0) -> None:
        """Make every replica start from rank ``src``'s parameters (one broadcast of the flat buffer)."""
        if self.backend != "nccl" and self.is_cuda:
            host = self.arena.flat.cpu()
            dist.broadcast(host, src=src, group=self.group)
            self.arena.flat.copy_(host)
        else:
            dist.broadcast(self.arena.flat, src=src, group=self.group)
        self.arena.mark_dirty()

    def _launch(self, idx: int) -> None:
        self._launched[idx] = True
        if self.capture is not None:
            self.capture.ready.append(idx)
            return
        self.launch_now(idx)

    def launch_now(self, idx: int) -> None:
        """Issue bucket ``idx``'s all-reduce (side stream on devices, ordered after everything on the current stream)."""
        s, e = self.buckets[idx]
        buf = self.arena.gflat[s:e]
        if self.is_cuda and self.backend == "nccl":
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self._allreduce(buf)
        else:
            self._allreduce(buf)

    def _allreduce(self, buf: torch.Tensor) -> None:
        if _DIAG:   # diagnostic only (tools/fd_sweep.sh): leave out parts of the exchange to price them; wrong gradients
            from . import kernels as K
            if "cast" in _DIAG and self.comm_bf16:
                st = self._stage[:buf.numel()]
                K.cast_f32_to_bf16(buf, st)
                K.cast_bf16_to_f32(st, buf)
            if "comm" in _DIAG:
                dist.all_reduce(self._stage[:buf.numel()] if self.comm_bf16 else buf, op=dist.ReduceOp.AVG, group=self.group)
            return
        if self.backend == "nccl":
            if self.comm_bf16:
                from . import kernels as K
                st = self._stage[:buf.numel()]
                K.cast_f32_to_bf16(buf, st)
                dist.all_reduce(st, op=dist.ReduceOp.AVG, group=self.group)
                K.cast_bf16_to_f32(st, buf)
            else:
                dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.group)
            return
        # gloo has no AVG; device tensors go through the host (2-process tests on one GPU)
        if self.is_cuda:
            if self.comm_bf16:      # same rounding points as the RCCL path: bf16 on the wire, fp32 result
                host = buf.to(torch.bfloat16).float().cpu()
            else:
                host = buf.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            host.mul_(1.0 / self.world)
            if self.comm_bf16:
                host = host.to(torch.bfloat16).float()
            buf.copy_(host)
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
            buf.mul_(1.0 / self.world)

    def mark_final(self, slots) -> None:
        """Called from backward (ParamArena.flush_final) with one entry per gradient WRITE since the last flush.  During
        the calibration step the writes per slot are only counted (buckets are reduced in ``finish``).  Afterwards a
        slot is final once it has received as many writes as in the calibration step; a bucket whose slots are all
        final is all-reduced right away on the side stream.  Order-independent: nothing is assumed about the order in
        which autograd runs the blocks."""
        if not self._calibrated:
            for s in slots:
                self._expected[id(s)] = self._expected.get(id(s), 0) + 1
            return
        for s in slots:
            sid = id(s)
            exp = self._expected.get(sid)
            if exp is None:
                # a parameter that got no gradient in the calibration step: its bucket can no longer be trusted to be
                # complete early -- reduce it in finish()
                bi = self._bucket_of[sid]
                if not self._launched[bi]:
                    self._waiting[bi] = 1 << 30
                continue
            n = self._seen.get(sid, 0) + 1
            self._seen[sid] = n
            if n == exp:
                bi = self._bucket_of[sid]
                self._waiting[bi] -= 1
                if self._waiting[bi] == 0 and not self._launched[bi]:
                    self._launch(bi)
        if self.capture is not None and self.capture.ready:
            self.capture.cut()

    def finish(self) -> None:
        """Launch every bucket not launched yet (parameters that got no gradient this step keep their bucket
        waiting until here) and make the compute stream wait for the reductions."""
        for bi in range(len(self.buckets)):
            if not self._launched[bi]:
                self._launch(bi)
        if self.capture is None:
            self.join()
        self._calibrated = True
        # parameters that never receive a gradient (e.g. the pooler when only logits are used) are not waited for
        self._seen = {}
        self._waiting = [0] * len(self.buckets)
        for sid in self._expected:
            self._waiting[self._bucket_of[sid]] += 1
        self._launched = [False] * len(self.buckets)

    def join(self) -> None:
        """Make the compute stream wait for the reductions issued so far."""
        if self.is_cuda and self.backend == "nccl":
            torch.cuda.current_stream().wait_stream(self.comm_stream)

    def reduce_all(self) -> None:
        """Non-overlapped form: all-reduce every bucket now (after backward)."""
        self.finish()

    # -------------------------------------------------------------------------------------------------
    def wire_bytes(self) -> int:
        """Bytes one rank puts into the all-reduces of a step (before the algorithm's 2(N-1)/N factor)."""
        return sum(e - s for s, e in self.buckets) * (2 if self.comm_bf16 else 4)
