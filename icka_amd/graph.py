"""Whole-step hipGraph capture (torch.cuda.CUDAGraph is a hipGraph on ROCm).

A forward+backward step of the hot path is ~850 short kernel launches; issued eagerly from Python they cost ~10 us of
host time each and the GPU starves once the kernels are fast.  The step is static (fixed shapes, no host sync, all
scratch from the caching allocator, parameters/gradients in the ParamArena), so it is captured once and replayed.
Dropout stays random across replays through the device-side nonce (icka_bump_dropout_nonce is the first node of the
graph).  Gradients are written with beta = 0 (the capture happens right after zero_grad), so every replay leaves
this step's gradients in ``p.grad``; the optimizer runs outside the graph and ``ParamArena.sync`` refreshes the bf16
shadows before the next replay.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from . import kernels as K


class GraphedStep(object):
    def __init__(self, model: torch.nn.Module, step_fn: Callable[[], torch.Tensor], warmup: int = 3):
        """``step_fn`` runs forward + backward (+ gradient all-reduce launches) and returns the loss tensor."""
        dev = next(model.parameters()).device
        self.model = model
        self.nonce = torch.zeros(2, dtype=torch.int32, device=dev)
        K.set_dropout_nonce(self.nonce)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):          # capture prerequisites: allocator / library warm-up off the default stream
            for _ in range(warmup):
                model.zero_grad()
                K.bump_dropout_nonce(self.nonce)
                step_fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.arena = model._icka_arena
        self.graph = torch.cuda.CUDAGraph()
        model.zero_grad()                       # gradients dropped -> captured kernels overwrite (beta = 0)
        # thread_local: only this thread's calls are checked against the capture -- with a process group alive, RCCL's
        # watchdog / heartbeat threads make runtime calls of their own that must not invalidate it
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            K.bump_dropout_nonce(self.nonce)
            self.loss = step_fn()

    def __call__(self) -> torch.Tensor:
        self.arena.sync()                       # parameters may have changed since the last replay (optimizer step)
        self.graph.replay()
        return self.loss
