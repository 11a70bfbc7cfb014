"""Whole-step hipGraph capture (torch.cuda.CUDAGraph is a hipGraph on ROCm).

A forward+backward step of the hot path is ~850 short kernel launches; issued eagerly from Python they cost ~10 us of
host time each and the GPU starves once the kernels are fast.  The step is static (fixed shapes, no host sync, all
scratch from the caching allocator, parameters/gradients in the ParamArena), so it is captured once and replayed.
Dropout stays random across replays through the device-side nonce (icka_bump_dropout_nonce is the first node of the
graph).

Contract
  * Gradients are captured with beta = 0 (the capture happens right after zero_grad): every replay OVERWRITES the
    gradient arena with this step's gradients -- a GraphedStep does not accumulate across calls.  After each replay
    ``p.grad`` is re-attached to the arena views for every parameter that received a gradient at capture, so the
    reference's loop (``model.zero_grad()`` / ``optimizer.zero_grad()`` after every step, My_cross_attention.py:843,
    set_to_none or not) keeps working with ``optimizer.step()`` seeing this step's gradients.
  * The optimizer runs outside the graph.  With the default shadow policy ("always") the bf16 re-cast of the
    parameters is the first kernel inside the captured forward; with "tracked" it runs before the replay when a change
    was seen.
  * ``close()`` (also run by ``__del__``) unregisters the dropout nonce, whose device memory this object owns.
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
        self._grad_slots = [s for s in self.arena.order if s.live]

    def __call__(self) -> torch.Tensor:
        if self.graph is None:
            raise RuntimeError("GraphedStep is closed")
        if self.arena.shadow_policy != "always":   # "always": the cast is a node of the captured forward
            self.arena.sync()
        self.graph.replay()
        self.arena.attach_grads(self._grad_slots)   # replay runs no Python: p.grad may have been dropped by zero_grad
        return self.loss

    def close(self) -> None:
        """Release the graph and unregister the dropout nonce (the kernels keep a raw pointer to it)."""
        if getattr(self, "nonce", None) is not None:
            try:
                K.clear_dropout_nonce_if(self.nonce)
            except Exception:      # interpreter shutdown: the library may already be gone
                pass
            self.nonce = None
        self.graph = None

    def __del__(self):
        self.close()
