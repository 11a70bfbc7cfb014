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

Data parallel: ``FlaggedStep`` (one graph, bucket-ready flag words, eager all-reduces behind flag-wait kernels on the
communication stream) is the default; ``SegmentedStep`` is the fallback it replaced: a captured graph with a side-stream branch per gradient bucket costs ~0.17 ms for the
first fork and ~25 us for each further one on this runtime (tools/graph_fork_probe.py), and needs the collective library
to be capturable.  ``SegmentedStep`` instead cuts the step into LINEAR graphs at the points where a gradient bucket
becomes final and issues the bucket's all-reduce eagerly, on the communication stream, between two segment launches:
the host enqueues everything ahead of the GPU, the compute stream sees the segments back to back, and RCCL is never
captured.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from . import kernels as K


class GraphedStep(object):
    def __init__(self, model: torch.nn.Module, step_fn: Callable[[], torch.Tensor], warmup: int = 3):
        """``step_fn`` runs forward + backward (+ gradient all-reduce launches) and returns the loss tensor.
        (Replaying the grouped weight-gradient launches on a side stream behind flag waits, beside the dependent chain of
        backward, was built and measured in round 3: -0.6 %, inside the noise -- profiles/r03_wgrad_side_stream.txt.)"""
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
        # a persistent LSTM launch of an EARLIER replay that gave up a hand-off (NaN-poisoned outputs): raise at this host
        # touch-point (a host read of a mapped word, no synchronisation)
        K.lstm_check_error("detected before a GraphedStep replay")
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


class _Capture(object):
    """State of a segmented capture: the graph being recorded, the finished (graph, buckets-after-it) segments, and the
    buckets that became ready since the last cut (filled by GradReducer._launch)."""

    def __init__(self):
        self.graph = None
        self.pool = torch.cuda.graph_pool_handle()   # one allocator pool for all segments: later ones read earlier tensors
        self.segments = []
        self.ready = []

    # -- the reducer's capture protocol (dp.GradReducer._launch / mark_final)
    def bucket_ready(self, idx: int) -> None:
        self.ready.append(idx)

    def after_mark(self) -> None:
        if self.ready:
            self.cut()

    def begin(self) -> None:
        self.graph = torch.cuda.CUDAGraph()
        self.graph.capture_begin(pool=self.pool, capture_error_mode="thread_local")

    def end(self) -> None:
        self.graph.capture_end()
        self.segments.append((self.graph, self.ready))
        self.graph, self.ready = None, []

    def cut(self) -> None:
        self.end()
        self.begin()


class SegmentedStep(object):
    """Data-parallel step as a chain of linear hipGraphs with eager bucket all-reduces between them (module docstring).
    ``step_fn`` runs forward + backward + ``reducer.finish()`` and returns the loss; ``reducer`` must already be
    attached to the model's arena.  Same gradient contract as ``GraphedStep``."""

    def __init__(self, model: torch.nn.Module, step_fn: Callable[[], torch.Tensor], reducer, warmup: int = 3):
        dev = next(model.parameters()).device
        self.model, self.reducer = model, reducer
        self.nonce = torch.zeros(2, dtype=torch.int32, device=dev)
        K.set_dropout_nonce(self.nonce)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):          # warm-up (the first of these steps calibrates the reducer's write counts)
            for _ in range(max(warmup, 2)):
                model.zero_grad()
                K.bump_dropout_nonce(self.nonce)
                step_fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.arena = model._icka_arena
        model.zero_grad()                       # gradients dropped -> captured kernels overwrite (beta = 0)
        cap = _Capture()
        reducer.capture = cap
        try:
            # backward on THIS thread: begin / end of a stream capture must come from one thread, and the cuts happen
            # inside backward (GradReducer.mark_final)
            with torch.cuda.stream(side), torch.autograd.set_multithreading_enabled(False):
                cap.begin()
                K.bump_dropout_nonce(self.nonce)
                self.loss = step_fn()           # its reducer.finish() reports the remaining buckets, without joining
                # the embedding backward is the last kernel of the step and makes the last bucket final: the segment
                # opened by that cut would be empty -- give it one (no-op) node
                self._pad = torch.zeros(2, dtype=torch.int32, device=dev)
                K.bump_dropout_nonce(self._pad)
                cap.end()
        finally:
            reducer.capture = None
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.segments = cap.segments
        self._grad_slots = [s for s in self.arena.order if s.live]

    def __call__(self) -> torch.Tensor:
        if self.segments is None:
            raise RuntimeError("SegmentedStep is closed")
        if self.arena.shadow_policy != "always":
            self.arena.sync()
        K.lstm_check_error("detected before a SegmentedStep replay")
        for graph, buckets in self.segments:
            graph.replay()
            for bi in buckets:
                self.reducer.launch_now(bi)
        self.reducer.join()
        self.arena.attach_grads(self._grad_slots)
        return self.loss

    def close(self) -> None:
        if getattr(self, "nonce", None) is not None:
            try:
                K.clear_dropout_nonce_if(self.nonce)
            except Exception:
                pass
            self.nonce = None
        self.segments = None

    def __del__(self):
        self.close()


class _FlagCapture(object):
    """Capture protocol of ``FlaggedStep``: a bucket that becomes ready while the step is being captured gets a flag-set
    node in the graph (icka_dp_flag_set on the capturing stream) instead of a collective."""

    def __init__(self, sync_words: torch.Tensor):
        self.sync = sync_words
        self.order = []

    def bucket_ready(self, idx: int) -> None:
        K.check(K._lib.load().icka_dp_flag_set(self.sync[FlaggedStep.FLAG0 + idx:].data_ptr(), self.sync.data_ptr(), K._stream()),
                "icka_dp_flag_set")
        self.order.append(idx)

    def after_mark(self) -> None:
        pass


class FlaggedStep(object):
    """Data-parallel step as ONE hipGraph with eager, overlapped all-reduces (the default of bench.py at N > 1).

    ``SegmentedStep`` pays for every cut (a hipGraph drains before the next one starts: +0.21 ms for 5 cuts at c2) and an
    external event inside a graph is refused by this ROCm build.  Here the step -- forward, backward and, at every point
    where a gradient bucket becomes final, a one-thread node that stores the step number into the bucket's FLAG WORD
    (icka_dp_flag_set) -- is captured whole.  A replay launches the graph on the compute stream and then, per bucket in the
    order the flags will rise, enqueues on the reducer's communication stream a one-wave kernel that waits for that flag
    (icka_dp_flag_wait: bounded spin with s_sleep; a wait that gives up raises a host-visible error word and poisons the
    bucket with a NaN) followed by the bucket's eager all-reduce -- the tagged-word hand-off of csrc/lstm.hip between two
    streams.  No collective is captured, no graph has a second branch, the compute stream never waits for the
    communication stream before the end of the step.  Same gradient contract as ``GraphedStep``.
    ``step_fn`` runs forward + backward + ``reducer.finish()`` and returns the loss; ``reducer`` is attached to the arena."""

    FLAG0 = 16           # sync words: [0] step counter (bumped by the graph's first node), [FLAG0 + i] flag of bucket i
    WAIT_POLLS = 1 << 20   # ~3 s of s_sleep(64) polls before a wait gives up

    def __init__(self, model: torch.nn.Module, step_fn: Callable[[], torch.Tensor], reducer, warmup: int = 3):
        dev = next(model.parameters()).device
        if not (reducer.is_cuda and reducer.backend == "nccl"):
            raise RuntimeError("FlaggedStep needs the nccl (= RCCL) backend on a ROCm device")
        self.model, self.reducer = model, reducer
        self.nonce = torch.zeros(2, dtype=torch.int32, device=dev)
        K.set_dropout_nonce(self.nonce)
        self.sync = torch.zeros(self.FLAG0 + len(reducer.buckets) + 16, dtype=torch.int32, device=dev)
        K.check(K._lib.load().icka_dp_init(), "icka_dp_init")
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):          # warm-up (the first of these steps calibrates the reducer's write counts)
            for _ in range(max(warmup, 2)):
                model.zero_grad()
                K.bump_dropout_nonce(self.nonce)
                step_fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.arena = model._icka_arena
        model.zero_grad()                       # gradients dropped -> captured kernels overwrite (beta = 0)
        cap = _FlagCapture(self.sync)
        self.graph = torch.cuda.CUDAGraph()
        reducer.capture = cap
        try:
            # backward on THIS thread: the flag nodes are launched from inside backward (GradReducer.mark_final)
            with torch.autograd.set_multithreading_enabled(False), \
                    torch.cuda.graph(self.graph, stream=side, capture_error_mode="thread_local"):
                K.check(K._lib.load().icka_dp_step_bump(self.sync.data_ptr(), K._stream()), "icka_dp_step_bump")
                K.bump_dropout_nonce(self.nonce)
                self.loss = step_fn()           # its reducer.finish() reports the remaining buckets (flags at the end)
        finally:
            reducer.capture = None
        torch.cuda.synchronize()
        self.order = cap.order
        if sorted(self.order) != list(range(len(reducer.buckets))):
            raise RuntimeError("FlaggedStep: buckets flagged during capture %s != all %d buckets" % (self.order, len(reducer.buckets)))
        self._tag = 0
        self._grad_slots = [s for s in self.arena.order if s.live]
        import os
        self._polls = int(os.environ.get("ICKA_DP_WAIT_POLLS", self.WAIT_POLLS))
        self._flag_ptr = [self.sync.data_ptr() + 4 * (self.FLAG0 + i) for i in range(len(reducer.buckets))]

    def __call__(self) -> torch.Tensor:
        if self.graph is None:
            raise RuntimeError("FlaggedStep is closed")
        if self.arena.shadow_policy != "always":
            self.arena.sync()
        K.lstm_check_error("detected before a FlaggedStep replay")
        K.dp_check_error("detected before a FlaggedStep replay")
        self._tag += 1                          # == the step counter the graph's first node is about to write
        self.graph.replay()
        r = self.reducer
        for idx in self.order:
            r.launch_now(idx, wait=(self._flag_ptr[idx], self._tag, self._polls))
        r.join()
        self.arena.attach_grads(self._grad_slots)
        return self.loss

    def close(self) -> None:
        if getattr(self, "nonce", None) is not None:
            try:
                K.clear_dropout_nonce_if(self.nonce)
            except Exception:
                pass
            self.nonce = None
        self.graph = None

    def __del__(self):
        self.close()
