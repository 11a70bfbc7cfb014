"""Whole-step hipGraph capture (torch.cuda.CUDAGraph is a hipGraph on ROCm).

A forward+backward step of the hot path is ~230 kernel launches at c2 (bert-base, 12 layers + 1 cross layer); issued eagerly
from Python through ctypes + autograd they cost more host time than the kernels take on the GPU (bench.py reports both:
``ms_per_step`` for the replayed graph, ``eager_ms_per_step`` for the same step launched eagerly).  A step of fixed shapes is
static (no host sync, all scratch from the caching allocator, parameters/gradients in the ParamArena), so it is captured once
and replayed.  Dropout stays random across replays through the device-side nonce (icka_bump_dropout_nonce is the first node of
every graph).

What the reference's loop does around a step (My_cross_attention.py:797-875) and how the wrappers follow it
  * a NEW batch every step (``batch = tuple(t.to(device) for t in batch)``, :797-798): build the step with ``inputs=`` (the
    first batch; a tuple / list / dict of tensors) and a ``step_fn`` that takes them; the tensors are cloned into STATIC
    device buffers the captured kernels read, and ``gs(*batch)`` copies the new tensors into them (no reallocation, host or
    device sources) before the replay.  ``gs()`` replays on the buffers as they are.
  * batches of OTHER shapes and the other train / eval mode: the loader has no ``drop_last`` (:708: the last batch of an epoch
    is short), the dev pass of every epoch runs in ``eval()`` under ``no_grad`` at its own batch size (:734, :846-875) and the
    test pass at batch 4 (:1022).  ``GraphedStep`` / ``GraphedModule`` therefore keep a small CACHE of captures keyed by
    (call signature, tensor shapes + dtypes, ``module.training``, grad mode): a key seen for the first time is captured (warm-up
    + capture on the spot, the gradients the caller holds are put aside and restored), later calls replay it; beyond
    ``max_captures`` entries -- or when a capture fails -- the call runs the wrapped module / step function EAGERLY.  A call
    never raises because of the wrapper.
  * gradient accumulation over ``gradient_accumulation_steps`` micro-batches (:587-590, :821-822, :831): the step is
    captured twice, lazily -- once with every gradient store as an overwrite (beta = 0) and once as an accumulation
    (beta = 1).  Which one a call replays follows the same rule the eager kernels use per slot (ParamArena.grad_beta):
    gradients the caller still holds (``p.grad`` is the arena view written earlier in this cycle) are accumulated into;
    after ``model.zero_grad()`` / ``optimizer.zero_grad()`` (:843, set_to_none or not) the next call starts a new cycle.
    The 1 / k of the loss (:821-822) stays in the caller's ``step_fn``, as in the reference.
  * ``p.grad`` is re-attached to the arena views after every replay (a replay runs no Python), so ``clip_grad_norm_``,
    ``optimizer.step()`` and ``scheduler.step()`` (:841-843) see this cycle's gradients.
  * The optimizer runs outside the graph.  With the default shadow policy ("always") the bf16 re-cast of the parameters is
    the first kernel inside the captured forward; with "tracked" it runs before the replay when a change was seen.
  * ``close()`` (also run by ``__del__``) unregisters the dropout nonce, whose device memory the wrapper owns.

``GraphedModule`` is the same machinery behind the module's own call: ``model(...)`` replays a forward graph, ``loss.backward()`` a
backward graph, so the reference's loop body does not change at all (it costs one more graph boundary per step); with
``reducer=`` its backward also exchanges the gradients (the flag-word protocol below), as a DDP-wrapped module's does.

Data parallel: ``FlaggedStep`` (one graph, bucket-ready flag words, eager all-reduces behind flag-wait kernels on the
communication stream) is the default; with ``accumulate=k`` only the k-th micro-batch of a cycle exchanges gradients (apex
DDP's delay_allreduce under accumulation; the first k-1 replay graphs captured without the reducer).  ``SegmentedStep`` is
the fallback it replaced: a captured graph with a side-stream branch per gradient bucket costs ~0.17 ms for the first fork
and ~25 us for each further one on this runtime (tools/graph_fork_probe.py), and needs the collective library to be
capturable.  ``SegmentedStep`` instead cuts the step into LINEAR graphs at the points where a gradient bucket becomes final
and issues the bucket's all-reduce eagerly, on the communication stream, between two segment launches: the host enqueues
everything ahead of the GPU, the compute stream sees the segments back to back, and RCCL is never captured.

ONE step form per job.  The forms issue different collective sequences, so every rank must run the same one (the reference has
a single code path per rank, My_cross_attention.py:653-657, :768-776).  Building a data-parallel form has two phases: the
warm-up steps, which exchange gradients and are the same on every rank, and the capture itself, which issues NO process-group
traffic (flag nodes / graph cuts only) and may fail on one rank alone.  After the capture phase the ranks vote over the c10d
store (dp.all_ranks_agree: host-only, no collective); unless every rank captured, ALL of them raise ``dp.CaptureDisagreement``
and take the next form together.  ``build_step`` is that chain (flagged -> segmented -> captured compute + eager all-reduces
-> eager) behind one call; bench.py uses it as its only decision point.
"""
from __future__ import annotations

import os
import sys
from typing import Callable, Optional

import torch

from . import kernels as K


def _sig(values) -> tuple:
    """Shape / dtype signature of a list of step inputs (non-tensors by value)."""
    out = []
    for v in values:
        if isinstance(v, torch.Tensor):
            out.append((tuple(v.shape), v.dtype))
        else:
            try:
                hash(v)
                out.append(("const", v))
            except TypeError:
                out.append(("const", repr(v)))
    return tuple(out)


def _test_fail(form: str) -> None:
    """Test hook, called INSIDE the capture phase of every form: ``ICKA_TEST_FAIL_CAPTURE="<who>:<form>[+<form>...]"`` with who
    = ``all`` | ``rank<k>`` and form in flagged / segmented / step / module makes that capture raise on the named rank(s) --
    after kernels have been recorded, so the clean-up of a half-finished capture is part of what the tests run."""
    spec = os.environ.get("ICKA_TEST_FAIL_CAPTURE")
    if not spec:
        return
    who, _, forms = spec.partition(":")
    if form in forms.split("+") and who in ("all", "rank%s" % os.environ.get("RANK", "0")):
        raise RuntimeError("simulated capture failure (form %r, ICKA_TEST_FAIL_CAPTURE=%s)" % (form, spec))


def _note(msg: str) -> None:
    print("[icka_amd.graph] " + msg, file=sys.stderr, flush=True)


class _GradSnapshot(object):
    """The gradients the caller holds (values, per-slot cycle state, ``p.grad`` objects), put aside while a capture made in
    the middle of a training loop runs its warm-up steps, and restored afterwards: a batch of a new shape may arrive at any
    micro-batch of an accumulation cycle (My_cross_attention.py:708, :831)."""

    def __init__(self, arena):
        self.arena = arena
        self.g = arena.gflat.clone()
        self.state = [(s, s.live, s.param.grad) for s in arena.order]

    def restore(self) -> None:
        self.arena.gflat.copy_(self.g)
        for s, live, grad in self.state:
            s.live = live
            s.param.grad = grad
        self.arena._pending_final = []


class StaticInputs(object):
    """Static device copies of a step's input tensors: the addresses a captured graph reads, refreshed per call.

    ``example``: tuple / list (positional arguments of ``step_fn``) or dict (keyword arguments) of tensors; entries that are
    not tensors (None, numbers) are passed through unchanged and must not change between calls."""

    def __init__(self, example, device):
        if isinstance(example, torch.Tensor):
            example = (example,)
        self.is_dict = isinstance(example, dict)
        items = list(example.items()) if self.is_dict else list(enumerate(example))
        self.keys = [k for k, _ in items]
        self.static = []
        for _, v in items:
            if isinstance(v, torch.Tensor):
                self.static.append(torch.empty(v.shape, dtype=v.dtype, device=device).copy_(v))
            else:
                self.static.append(v)

    def call(self, fn):
        if self.is_dict:
            return fn(**dict(zip(self.keys, self.static)))
        return fn(*self.static)

    def values_of(self, args, kwargs) -> list:
        """The call's inputs in the order of the static buffers (a single tuple / list / dict argument is unpacked); arity and
        keyword names are checked against the captured call."""
        if len(args) == 1 and not kwargs and isinstance(args[0], (tuple, list, dict)):
            if isinstance(args[0], dict):
                args, kwargs = (), args[0]
            else:
                args = tuple(args[0])
        if self.is_dict:
            if args or set(kwargs) != set(self.keys):
                raise TypeError("step inputs: expected keyword tensors %s, got %d positional + %s"
                                % (sorted(map(str, self.keys)), len(args), sorted(kwargs)))
            return [kwargs[k] for k in self.keys]
        if kwargs or len(args) != len(self.keys):
            raise TypeError("step inputs: expected %d positional tensors, got %d (+ keywords %s)"
                            % (len(self.keys), len(args), sorted(kwargs)))
        return list(args)

    def signature(self) -> tuple:
        return (self.is_dict, tuple(self.keys), _sig(self.static))

    def refresh(self, args, kwargs) -> None:
        """Copy a new batch into the static buffers (asynchronous copies on the current stream, ahead of the replay)."""
        self.refresh_values(self.values_of(args, kwargs))

    def refresh_values(self, new) -> None:
        dev_src, dev_dst = [], []
        for k, dst, src in zip(self.keys, self.static, new):
            if not isinstance(dst, torch.Tensor):
                if isinstance(src, torch.Tensor) or src != dst:
                    raise TypeError("step input %r was captured as the constant %r and cannot change per call" % (k, dst))
                continue
            if not isinstance(src, torch.Tensor):
                raise TypeError("step input %r: expected a tensor, got %s" % (k, type(src).__name__))
            if src.shape != dst.shape or src.dtype != dst.dtype:
                # (GraphedStep / GraphedModule never get here: they pick or make the capture of the call's signature first)
                raise ValueError("step input %r: captured as %s %s, got %s %s -- ONE capture has static shapes"
                                 % (k, tuple(dst.shape), dst.dtype, tuple(src.shape), src.dtype))
            if src.data_ptr() == dst.data_ptr():
                continue
            if src.device == dst.device and src.is_contiguous():
                dev_src.append(src)
                dev_dst.append(dst)
            else:
                dst.copy_(src, non_blocking=True)       # host tensors (pinned or pageable), strided sources
        # device-resident sources: ONE launch for all of them (icka_copy_many), 8 per launch
        for i in range(0, len(dev_src), 8):
            K.copy_many(dev_src[i:i + 8], dev_dst[i:i + 8])


class DevicePrefetcher(object):
    """Iterate a host-side batch source (the reference's DataLoader loop, My_cross_attention.py:795-798) one batch AHEAD: while
    step i runs, batch i + 1 is copied host -> device on a copy stream of its own, through a pinned staging copy when the source
    tensors are pageable.  Each item is a tuple of device tensors in one of ``depth`` rotating buffer sets; the consumer's stream
    is made to wait for the copy, and the copy stream waits for the consumer before a set is overwritten, so ``gs(*batch)`` sees
    device sources (one icka_copy_many launch into the captured step's static buffers) and PCIe is off the step's critical path.
    A batch of another shape (the short last batch of an epoch: the reference's loader has no drop_last, :708) gets buffers of
    its own in the slot; only a change of arity is an error.

        for batch in DevicePrefetcher(train_dataloader, "cuda"):
            loss = gs(*batch)
    """

    def __init__(self, source, device, depth: int = 2):
        if depth < 2:
            raise ValueError("DevicePrefetcher needs at least two buffer sets")
        self.source, self.device, self.depth = source, torch.device(device), depth
        self.stream = torch.cuda.Stream(device=self.device)
        self._sig = None

    def __len__(self):
        return len(self.source)

    def _issue(self, slot, batch):
        if isinstance(batch, torch.Tensor):
            batch = (batch,)
        if slot["done"] is not None:
            self.stream.wait_event(slot["done"])            # the consumer is finished with this set
        sig = tuple((tuple(t.shape), t.dtype, bool(t.is_cuda or t.is_pinned())) for t in batch)
        if self._sig is None:
            self._sig = sig
        if len(sig) != len(self._sig):
            raise ValueError("DevicePrefetcher: batches must keep their arity (got %d tensors, the first batch had %d)"
                             % (len(sig), len(self._sig)))
        with torch.cuda.stream(self.stream):
            bufs = slot["bufs"].get(sig)
            if bufs is None:
                bufs = slot["bufs"][sig] = (
                    [torch.empty(t.shape, dtype=t.dtype, device=self.device) for t in batch],
                    [None if (t.is_cuda or t.is_pinned()) else torch.empty(t.shape, dtype=t.dtype).pin_memory() for t in batch])
            slot["dev"], slot["pin"] = bufs
            for t, d, pin in zip(batch, slot["dev"], slot["pin"]):
                if pin is not None:
                    if slot["staged"] is not None:
                        slot["staged"].synchronize()        # the previous copy out of this pinned buffer has left the host
                    pin.copy_(t)
                    t = pin
                d.copy_(t, non_blocking=True)
            slot["staged"] = torch.cuda.Event()
            slot["staged"].record(self.stream)
        return slot

    def __iter__(self):
        slots = [{"bufs": {}, "dev": None, "pin": None, "done": None, "staged": None} for _ in range(self.depth)]
        self._sig = None
        it = iter(self.source)
        pending = []
        i = 0
        try:
            pending.append(self._issue(slots[0], next(it)))
        except StopIteration:
            return
        while pending:
            slot = pending.pop(0)
            i += 1
            try:
                pending.append(self._issue(slots[i % self.depth], next(it)))
            except StopIteration:
                pass
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(slot["staged"])
            yield tuple(slot["dev"])
            slot["done"] = torch.cuda.Event()
            slot["done"].record(torch.cuda.current_stream(self.device))


class _StepBase(object):
    """Shared by the step objects: the dropout nonce, the static inputs, warm-up, and which capture a call replays."""

    def _setup(self, model, step_fn, inputs, nonce=None):
        dev = next(model.parameters()).device
        self.model = model
        self.device = dev
        self.inputs = StaticInputs(inputs, dev) if inputs is not None else None
        self._user_fn = step_fn
        # the nonce is owned by whoever made it: a wrapper with a cache of captures shares ONE among them
        self._owns_nonce = nonce is None
        self.nonce = torch.zeros(2, dtype=torch.int32, device=dev) if nonce is None else nonce
        K.set_dropout_nonce(self.nonce)
        self.side = torch.cuda.Stream(device=dev)

    def _step(self):
        return self.inputs.call(self._user_fn) if self.inputs is not None else self._user_fn()

    def _warm(self, n):
        self.side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):     # capture prerequisites: allocator / library warm-up off the default stream
            for _ in range(n):
                self.model.zero_grad()
                K.bump_dropout_nonce(self.nonce)
                self._step()
        torch.cuda.current_stream().wait_stream(self.side)
        torch.cuda.synchronize()

    def _refresh(self, args, kwargs):
        if args or kwargs:
            if self.inputs is None:
                raise TypeError("this step was built without inputs=: it replays a closure over fixed tensors")
            self.inputs.refresh(args, kwargs)

    def _cycle_state(self):
        """(accumulate, stale): accumulate = some gradient of this step is still held by the caller (ParamArena._is_live:
        written earlier in this accumulation cycle, p.grad still the arena view); stale = the slots that are NOT, which an
        accumulating replay must see as zeros (the eager path's 'mixed' rule, ParamArena.grad_beta)."""
        A = self.arena
        stale = [s for s in self._grad_slots if not A._is_live(s)]
        if len(stale) == len(self._grad_slots):
            return False, ()
        return True, stale

    def _zero(self, slots):
        A = self.arena
        for s in slots:
            A.gflat[s.off:s.off + s.numel].zero_()

    def _close_nonce(self):
        if getattr(self, "nonce", None) is not None:
            if getattr(self, "_owns_nonce", True):
                try:
                    K.clear_dropout_nonce_if(self.nonce)
                except Exception:      # interpreter shutdown: the library may already be gone
                    pass
            self.nonce = None

    def __del__(self):
        self.close()


def _end_capture_quietly() -> None:
    """After a capture that raised: wait for the device and swallow what a half-dead capture may still report."""
    try:
        torch.cuda.synchronize()
    except Exception:   # noqa: BLE001
        pass


class _StepCapture(_StepBase):
    """ONE captured forward + backward step at fixed input shapes and one train / eval mode: an overwrite graph and (lazily) an
    accumulate graph.  ``GraphedStep`` keeps a cache of these."""

    def __init__(self, model, step_fn, warmup=3, inputs=None, nonce=None):
        self._setup(model, step_fn, inputs, nonce)
        self._graphs = {}
        self._loss = {}
        self._warm(warmup)
        self.arena = model._icka_arena
        model.zero_grad()                       # gradients dropped -> captured kernels overwrite (beta = 0)
        self._capture(False)
        self._grad_slots = [s for s in self.arena.order if s.live]
        self.loss = self._loss[False]
        model.zero_grad()                       # a capture executes nothing: the gradients it "wrote" do not exist

    def _capture(self, accumulate: bool) -> None:
        if accumulate:      # every gradient store of the capture must see its slot live: beta = 1, no 'mixed' memsets
            self.arena.attach_grads(self._grad_slots)
        g = torch.cuda.CUDAGraph()
        # thread_local: only this thread's calls are checked against the capture -- with a process group alive, RCCL's
        # watchdog / heartbeat threads make runtime calls of their own that must not invalidate it
        try:
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                K.bump_dropout_nonce(self.nonce)
                self._loss[accumulate] = self._step()
                _test_fail("step")
        except Exception:
            self._loss.pop(accumulate, None)
            _end_capture_quietly()
            raise
        self._graphs[accumulate] = g

    @property
    def graph(self):
        return self._graphs.get(False) if self._graphs is not None else None

    def run(self, values=None) -> torch.Tensor:
        """Refresh the static inputs with ``values`` (already in buffer order; None = replay on the buffers as they are), pick
        the overwrite or the accumulate capture, replay."""
        if self._graphs is None:
            raise RuntimeError("GraphedStep is closed")
        if values is not None:
            self.inputs.refresh_values(values)
        if self.arena.shadow_policy != "always":   # "always": the cast is a node of the captured forward
            self.arena.sync()
        # a persistent LSTM launch of an EARLIER replay that gave up a hand-off (NaN-poisoned outputs): raise at this host
        # touch-point (a host read of a mapped word, no synchronisation)
        K.lstm_check_error("detected before a GraphedStep replay")
        K.gemm_ln_check_error("detected before a GraphedStep replay")
        accumulate, stale = self._cycle_state()
        if accumulate and True not in self._graphs:
            self._capture(True)                 # (a capture executes nothing: the gradients held so far are untouched)
        self._zero(stale)
        self._graphs[accumulate].replay()
        self.arena.attach_grads(self._grad_slots)   # replay runs no Python: p.grad may have been dropped by zero_grad
        self.loss = self._loss[accumulate]
        return self.loss

    def __call__(self, *args, **kwargs) -> torch.Tensor:
        if args or kwargs:
            if self.inputs is None:
                raise TypeError("this step was built without inputs=: it replays a closure over fixed tensors")
            return self.run(self.inputs.values_of(args, kwargs))
        return self.run(None)

    def close(self) -> None:
        """Release the graphs and unregister the dropout nonce (the kernels keep a raw pointer to it)."""
        self._close_nonce()
        self._graphs = None


class GraphedStep(object):
    """``step_fn`` runs forward + backward and returns the loss tensor; with ``inputs`` (tuple / list / dict of tensors: the
    first batch) it is called with static copies of them and ``gs(*batch)`` refreshes those per call (module docstring).

    The object holds up to ``max_captures`` captures keyed by (input shapes + dtypes, ``model.training``): the first batch's is
    made here (an error in it propagates: the caller asked for a captured step); a batch of another shape or a call in the other
    train / eval mode is captured the first time it is seen (the gradients held at that moment are put aside and restored) and
    replayed afterwards; past the cap, or if such a later capture fails, the call runs ``step_fn`` eagerly.  ``stats`` counts
    captures / replays / eager calls.  (Replaying the grouped weight-gradient launches on a side stream behind flag waits,
    beside the dependent chain of backward, was built and measured in round 3: -0.6 %, inside the noise --
    profiles/r03_wgrad_side_stream.txt.)"""

    def __init__(self, model: torch.nn.Module, step_fn: Callable[..., torch.Tensor], warmup: int = 3, inputs=None,
                 max_captures: int = 4):
        if max_captures < 1:
            raise ValueError("max_captures must be >= 1")
        self.model = model
        self._user_fn = step_fn
        self._warmup = warmup
        self.max_captures = int(max_captures)
        dev = next(model.parameters()).device
        self.nonce = torch.zeros(2, dtype=torch.int32, device=dev)
        self._caps = {}
        self._eager_keys = {}
        self.stats = {"captures": 0, "replays": 0, "eager_calls": 0}
        first = _StepCapture(model, step_fn, warmup, inputs, nonce=self.nonce)
        self._primary = self._last = first
        self.arena = first.arena
        self._sigkey0 = first.inputs.signature() if first.inputs is not None else None
        self._caps[(self._sigkey0, bool(model.training))] = first
        self.stats["captures"] = 1

    # -- what callers of the one-capture class used to reach for
    @property
    def inputs(self):
        return self._primary.inputs

    @property
    def loss(self):
        return self._last.loss

    @property
    def graph(self):
        return self._primary.graph

    @property
    def _graphs(self):
        return self._primary._graphs

    @property
    def captures(self) -> int:
        return 0 if self._caps is None else len(self._caps)

    def _new_capture(self, key, values):
        if key in self._eager_keys:
            return None
        if len(self._caps) >= self.max_captures:
            self._eager_keys[key] = "max_captures=%d reached" % self.max_captures
            _note("GraphedStep: %s; calls with inputs %s (training=%s) run eagerly" % (self._eager_keys[key], key[0], key[1]))
            return None
        P = self._primary
        if values is None:
            inputs = None
        elif P.inputs.is_dict:
            inputs = dict(zip(P.inputs.keys, values))
        else:
            inputs = tuple(values)
        snap = _GradSnapshot(self.arena)
        cap = None
        try:
            cap = _StepCapture(self.model, self._user_fn, self._warmup, inputs, nonce=self.nonce)
        except Exception as e:  # noqa: BLE001
            self._eager_keys[key] = "%s: %s" % (type(e).__name__, e)
            _note("GraphedStep: capture for inputs %s (training=%s) failed (%s); such calls run eagerly"
                  % (key[0], key[1], self._eager_keys[key]))
            _end_capture_quietly()
        finally:
            snap.restore()
        if cap is not None:
            self._caps[key] = cap
            self.stats["captures"] += 1
        return cap

    def __call__(self, *args, **kwargs) -> torch.Tensor:
        if self._caps is None:
            raise RuntimeError("GraphedStep is closed")
        P = self._primary
        training = bool(self.model.training)
        if args or kwargs:
            if P.inputs is None:
                raise TypeError("this step was built without inputs=: it replays a closure over fixed tensors")
            values = P.inputs.values_of(args, kwargs)          # arity / keyword names: TypeError, as any wrong call
            sigkey = (P.inputs.is_dict, tuple(P.inputs.keys), _sig(values))
        else:
            values = None
            sigkey = self._last.inputs.signature() if self._last.inputs is not None else None
        key = (sigkey, training)
        cap = self._caps.get(key)
        if cap is None:
            src = values
            if src is None and self._last.inputs is not None:
                src = list(self._last.inputs.static)           # gs() in the other mode: on the buffers as they are
            cap = self._new_capture(key, src)
            if cap is None:
                self.stats["eager_calls"] += 1
                if src is None:
                    return self._user_fn()
                return self._user_fn(**dict(zip(P.inputs.keys, src))) if P.inputs.is_dict else self._user_fn(*src)
        self._last = cap
        self.stats["replays"] += 1
        return cap.run(values)

    def close(self) -> None:
        """Release the graphs and unregister the dropout nonce (the kernels keep a raw pointer to it)."""
        if self._caps is not None:
            for c in self._caps.values():
                c.close()
        self._caps = None
        if self.nonce is not None:
            try:
                K.clear_dropout_nonce_if(self.nonce)
            except Exception:      # interpreter shutdown
                pass
            self.nonce = None

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001
            pass


class _EagerTail(torch.autograd.Function):
    """Identity behind an EAGER call of a GraphedModule that has a reducer: its backward runs first in ``loss.backward()`` --
    where the wrapper decides whether this micro-batch exchanges -- and queues the end-of-backward callback that finishes the
    exchange (what a DDP-wrapped module does with its reducer)."""

    @staticmethod
    def forward(ctx, out, owner):
        ctx.owner = owner
        return out.view_as(out)

    @staticmethod
    def backward(ctx, g):
        owner = ctx.owner
        owner._eager_backward_begins()
        torch.autograd.Variable._execution_engine.queue_callback(owner._eager_backward_ends)
        return g, None


class _ModuleCapture(_StepBase):
    """ONE captured call of a module at a fixed call signature, input shapes, train / eval mode and grad mode: a forward graph
    and -- when the output requires grad -- the backward graphs (overwrite / accumulate, with / without the gradient exchange).
    ``GraphedModule`` keeps a cache of these."""

    def __init__(self, owner, args, kwkeys, kwvalues, warmup):
        self.owner = owner
        module, reducer = owner.model, owner.reducer
        if reducer is not None:
            reducer.arena.reducer = reducer     # attached from the first warm-up backward on (it calibrates there)
        self.reducer = reducer
        self.accumulate = owner.accumulate
        self._nargs = len(args)
        self._kwkeys = list(kwkeys)
        self._setup(module, None, tuple(args) + tuple(kwvalues), nonce=owner.nonce)
        self.gf = None
        self._bwd = {}                          # (accumulate, exchange) -> backward graph
        self._order = {}                        # exchanging captures: the order their bucket flags rise in
        self._grad_slots = []
        model = module

        def fwd():
            st = self.inputs.static
            return model(*st[:self._nargs], **dict(zip(self._kwkeys, st[self._nargs:])))

        self._fwd = fwd
        self.side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            for _ in range(max(1, warmup)):
                model.zero_grad()
                K.bump_dropout_nonce(self.nonce)
                out = fwd()
                if not (isinstance(out, torch.Tensor) and out.is_floating_point()):
                    raise TypeError("GraphedModule captures calls that return ONE floating-point tensor (loss or logits), got %s"
                                    % type(out).__name__)
                if out.requires_grad:
                    out.backward(torch.ones_like(out))
                    if reducer is not None:
                        reducer.finish()       # (eager exchange: the first of these steps calibrates its write counts)
        torch.cuda.current_stream().wait_stream(self.side)
        torch.cuda.synchronize()
        self.arena = model._icka_arena
        if reducer is None and self.arena.reducer is not None:
            raise RuntimeError("this model's arena has a GradReducer attached: pass it as GraphedModule(..., reducer=) so that "
                               "backward() exchanges the gradients (or detach it: arena.reducer = None)")
        if reducer is not None:
            if reducer.arena is not self.arena:
                raise ValueError("GraphedModule(reducer=): the reducer belongs to another model's arena")
            self.arena.reducer = reducer
            self._xch = _FlagExchange(reducer, self.arena, self.device)
        # ---- the capture phase proper: no process-group traffic from here on (the exchanging backward captures carry flag
        #      nodes only), so it may fail on one rank alone -- GraphedModule votes afterwards
        model.zero_grad()
        try:
            self.gf = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.gf, capture_error_mode="thread_local"):
                K.bump_dropout_nonce(self.nonce)
                self.out = fwd()
                _test_fail("module")
            self.gout = torch.zeros_like(self.out)
            if self.out.requires_grad:
                # every capture a backward() can ask for is taken NOW: a capture runs autograd itself and must not start from
                # inside the autograd call that replays it
                k = self.accumulate
                if reducer is None:
                    keys = [(False, False), (True, False)]
                elif k == 1:
                    keys = [(False, True), (True, True)]
                else:
                    keys = [(False, False), (True, False), (True, True)]
                for i, key in enumerate(keys):
                    self._capture_bwd(*key)
                    if i == 0:
                        self._grad_slots = [s for s in self.arena.order if s.live]
        except Exception:
            self.gf, self._bwd = None, {}
            if reducer is not None:
                reducer.abort_step()
                self.arena.reducer = reducer
            self.arena.keep_saved = False
            _end_capture_quietly()
            raise
        model.zero_grad()                       # captures execute nothing
        outer = self

        class _Replay(torch.autograd.Function):
            @staticmethod
            def forward(ctx, anchor):
                outer.gf.replay()
                return outer.out.detach()

            @staticmethod
            def backward(ctx, g):
                outer._replay_backward(g)
                return None

        self._fn = _Replay

    def _replay_backward(self, g: torch.Tensor) -> None:
        if not self._bwd:
            raise RuntimeError("this GraphedModule call was captured without a backward (its output did not require grad)")
        self.gout.copy_(g.expand_as(self.gout) if g.shape != self.gout.shape else g, non_blocking=True)
        accumulate, stale = self._cycle_state()
        exchange = False
        owner = self.owner
        if self.reducer is not None:
            if not accumulate:
                owner._micro = 0                # the caller dropped the gradients: a new cycle starts here
            exchange = owner._micro == self.accumulate - 1
        key = (accumulate, exchange)
        if key not in self._bwd:                # (accumulate = k > 1 and the first backward of a cycle is also its last: k == 1 only)
            raise RuntimeError("GraphedModule: no backward capture for accumulate=%s, exchange=%s" % key)
        self._zero(stale)
        if exchange:
            self._xch.before_replay()
        self._bwd[key].replay()
        if exchange:
            self._xch.after_replay(self._order[key])
            owner._micro = 0
        elif self.reducer is not None:
            owner._micro += 1
        self.arena.attach_grads(self._grad_slots)

    def _capture_bwd(self, accumulate: bool, exchange: bool) -> None:
        arena, r = self.arena, self.reducer
        if accumulate:
            arena.attach_grads(self._grad_slots)
        g = torch.cuda.CUDAGraph()
        key = (accumulate, exchange)
        arena.keep_saved = True
        try:
            if exchange:
                cap = _FlagCapture(self._xch.sync)
                r.capture = cap
                arena.reducer = r
                try:
                    with torch.autograd.set_multithreading_enabled(False), \
                            torch.cuda.graph(g, pool=self.gf.pool(), capture_error_mode="thread_local"):
                        self._xch.first_node()
                        self.out.backward(self.gout, retain_graph=True)
                        r.finish()              # reports the remaining buckets (their flags rise at the end of the graph)
                finally:
                    r.capture = None
                torch.cuda.synchronize()
                if sorted(cap.order) != list(range(len(r.buckets))):
                    raise RuntimeError("GraphedModule: buckets flagged during capture %s != all %d buckets" % (cap.order, len(r.buckets)))
                self._order[key] = cap.order
            else:
                if r is not None:               # a backward that does not exchange: reducer detached, no wire copies, no flags
                    arena.reducer = None
                    r.muted = True
                try:
                    with torch.autograd.set_multithreading_enabled(False), \
                            torch.cuda.graph(g, pool=self.gf.pool(), capture_error_mode="thread_local"):
                        self.out.backward(self.gout, retain_graph=True)
                finally:
                    if r is not None:
                        arena.reducer = r
                        r.muted = False
        finally:
            arena.keep_saved = False
        self._bwd[key] = g

    def run(self, values) -> torch.Tensor:
        if self.gf is None:
            raise RuntimeError("GraphedModule is closed")
        self.inputs.refresh_values(values)
        if self.arena.shadow_policy != "always":
            self.arena.sync()
        K.lstm_check_error("detected before a GraphedModule replay")
        K.gemm_ln_check_error("detected before a GraphedModule replay")
        if self.reducer is not None:
            K.dp_check_error("detected before a GraphedModule replay")
        if torch.is_grad_enabled() and self._bwd:
            return self._fn.apply(self.arena.anchor)
        self.gf.replay()
        return self.out.detach()

    def close(self) -> None:
        self._close_nonce()
        self.gf = None
        self._bwd = {}


class GraphedModule(object):
    """The import swap with NO change to the loop body: wraps a drop-in module so that the reference's own two lines

        loss = model(input_ids, segment_ids, input_mask, added_input_mask, imgs_f, img_att, labels=label_ids)   # :814-817
        loss.backward()                                                                                          # :827

    replay hipGraphs -- one for the forward, one for the backward -- instead of launching ~230 kernels from Python:

        model = icka_amd.graph.GraphedModule(model, example_args, example_kwargs)     # once
        loss = model(*batch_args, labels=label_ids); (loss / k).backward(); ...       # the loop stays as it is
        model.eval(); with torch.no_grad(): logits = model(*dev_batch_args)           # so does the dev / test pass

    ``model(...)`` copies the tensors into static buffers, replays the captured forward and returns the static output (a tensor;
    overwritten by the next call of the same capture, like any graphed callable) hooked into autograd through one Function; its
    backward copies the incoming gradient (the ``1 / k`` of the accumulation scaling arrives here), picks the overwrite or the
    accumulate capture of the backward by the same rule ``GraphedStep`` uses (are the gradients of this cycle still held?),
    replays it and re-attaches ``p.grad``.  Both backward captures read the activations of the ONE captured forward (the layer
    Functions keep their saved state while ``ParamArena.keep_saved`` is set).  No gradient flows to the inputs.

    The whole loop of the reference, not only its steady state (My_cross_attention.py:708 no drop_last; :734, :846-875 dev pass
    in eval() under no_grad at its own batch size; :1022 test at batch 4): captures are cached by (positional count, keyword
    names, tensor shapes + dtypes, ``module.training``, grad mode) -- the example call's is made here, any other is made the first
    time it is seen (up to ``max_captures``; the gradients held at that moment are put aside and restored; a capture made under
    ``no_grad`` has no backward graphs, a grad-mode capture also serves ``no_grad`` calls) and replayed afterwards.  Past the cap,
    for calls that return something else than one floating-point tensor, or when a capture fails, the call runs the wrapped
    module eagerly -- never an exception of the wrapper's own.  ``stats`` counts captures / replays / eager calls.

    Data parallel (the reference wraps the model in apex DDP and keeps the same two lines, My_cross_attention.py:768-776):
    ``GraphedModule(model, args, kwargs, reducer=GradReducer(...), accumulate=k)`` -- the backward capture carries the
    bucket-ready flag nodes of ``FlaggedStep``, and ``loss.backward()`` replays it, then enqueues the flag waits and the eager
    all-reduces on the reducer's communication stream and joins: when ``backward()`` returns (stream-ordered) ``p.grad`` holds
    the exchanged gradients, as after a DDP backward.  With ``accumulate=k`` only the k-th backward of a cycle (a cycle
    restarts at ``zero_grad``) exchanges, the others replay captures without the reducer; eager calls follow the same count.
    Whether a grad-mode call signature is replayed or run eagerly is agreed ACROSS the ranks (module docstring: warm-up =
    collective phase, capture = local phase, then a vote over the store); a disagreement on the example call raises
    ``dp.CaptureDisagreement`` on every rank, on a later signature all ranks run it eagerly."""

    def __init__(self, module: torch.nn.Module, example_args=(), example_kwargs=None, warmup: int = 3, reducer=None,
                 accumulate: int = 1, max_captures: int = 4):
        if accumulate < 1:
            raise ValueError("accumulate must be >= 1")
        if max_captures < 1:
            raise ValueError("max_captures must be >= 1")
        if reducer is None and accumulate != 1:
            raise ValueError("accumulate=k only changes WHEN gradients are exchanged: it needs reducer=")
        if reducer is not None:
            if not (reducer.is_cuda and reducer.backend == "nccl"):
                raise RuntimeError("GraphedModule(reducer=) needs the nccl (= RCCL) backend on a ROCm device")
            if getattr(reducer, "sparse_word", None) is not None:
                raise ValueError("GraphedModule(reducer=) does not take GradReducer(sparse_embeddings=True): use FlaggedStep")
        d = self.__dict__
        d["model"] = module
        d["reducer"] = reducer
        d["accumulate"] = int(accumulate)
        d["max_captures"] = int(max_captures)
        d["_warmup"] = warmup
        d["_micro"] = 0
        d["_caps"] = {}
        d["_eager_keys"] = {}
        d["_eager_exchange"] = False
        d["stats"] = {"captures": 0, "replays": 0, "eager_calls": 0}
        dev = next(module.parameters()).device
        d["nonce"] = torch.zeros(2, dtype=torch.int32, device=dev)
        example_kwargs = dict(example_kwargs or {})
        kwkeys = tuple(sorted(example_kwargs))
        values = list(example_args) + [example_kwargs[k] for k in kwkeys]
        key = (len(example_args), kwkeys, _sig(values), bool(module.training), torch.is_grad_enabled())
        cap, err = self._build(key, example_args, kwkeys, [example_kwargs[k] for k in kwkeys])
        if cap is None:
            from .dp import CaptureDisagreement
            if err is None:
                raise CaptureDisagreement("GraphedModule: another rank could not capture the example call")
            if reducer is not None:
                raise CaptureDisagreement("GraphedModule: this rank could not capture the example call (%s: %s)"
                                          % (type(err).__name__, err)) from err
            raise err
        d["_primary"] = cap
        d["arena"] = cap.arena

    # the wrapped module stays reachable (optimizers, state_dict, zero_grad, train / eval)
    def __getattr__(self, name):
        return getattr(object.__getattribute__(self, "model"), name)

    # -- attributes of the example call's capture that callers of the one-capture class used to reach for
    @property
    def gf(self):
        return None if self._caps is None else self._primary.gf

    @property
    def out(self):
        return self._primary.out

    @property
    def inputs(self):
        return self._primary.inputs

    @property
    def _xch(self):
        return self._primary._xch

    @property
    def _bwd(self):
        return self._primary._bwd

    @property
    def captures(self) -> int:
        return 0 if self._caps is None else len(self._caps)

    def _build(self, key, args, kwkeys, kwvalues):
        """(capture | None, this rank's error | None): warm-up (collective under a reducer) + local capture + -- for a grad-mode
        call under a reducer -- the vote that makes replay-or-eager one decision for all ranks."""
        arena = getattr(self.model, "_icka_arena", None)
        snap = _GradSnapshot(arena) if (arena is not None and self._caps) else None
        cap, err = None, None
        try:
            cap = _ModuleCapture(self, args, kwkeys, kwvalues, self._warmup)
        except Exception as e:  # noqa: BLE001
            err = e
            _end_capture_quietly()
        finally:
            if snap is not None:
                snap.restore()
        if self.reducer is not None and key[4]:
            from .dp import all_ranks_agree
            if not all_ranks_agree(cap is not None, self.reducer.group, "GraphedModule capture %s" % (key[:4],)):
                if cap is not None:
                    cap.close()
                cap = None
        if cap is not None:
            self._caps[key] = cap
            self.stats["captures"] += 1
        return cap, err

    def __call__(self, *args, **kwargs):
        if self._caps is None:
            raise RuntimeError("GraphedModule is closed")
        kwkeys = tuple(sorted(kwargs))
        kwvalues = [kwargs[k] for k in kwkeys]
        values = list(args) + kwvalues
        grad = torch.is_grad_enabled()
        key = (len(args), kwkeys, _sig(values), bool(self.model.training), grad)
        cap = self._caps.get(key)
        if cap is None and not grad:
            cap = self._caps.get(key[:4] + (True,))     # a grad-mode capture replays its forward alone under no_grad
        if cap is None and key not in self._eager_keys:
            if len(self._caps) >= self.max_captures:
                self._eager_keys[key] = "max_captures=%d reached" % self.max_captures
            elif not all(isinstance(v, (torch.Tensor, int, float, bool, str, type(None))) for v in values):
                self._eager_keys[key] = "arguments that are neither tensors nor plain constants"
            else:
                cap, err = self._build(key, args, kwkeys, kwvalues)
                if cap is None:
                    self._eager_keys[key] = "another rank could not capture it" if err is None else "%s: %s" % (type(err).__name__, err)
            if cap is None:
                _note("GraphedModule: calls with signature %s (training=%s, grad=%s) run eagerly: %s"
                      % (key[:3], key[3], key[4], self._eager_keys[key]))
        if cap is None:
            return self._eager(args, kwargs)
        self.stats["replays"] += 1
        return cap.run(values)

    # ---- eager calls (past the cache, un-capturable signatures)
    def _eager(self, args, kwargs):
        self.stats["eager_calls"] += 1
        out = self.model(*args, **kwargs)
        if self.reducer is None or not (torch.is_grad_enabled() and isinstance(out, torch.Tensor) and out.requires_grad):
            return out
        return _EagerTail.apply(out, self)

    def _eager_backward_begins(self) -> None:
        A, r = self.arena, self.reducer
        if not any(A._is_live(s) for s in A.order):
            self._micro = 0                     # the caller dropped the gradients: a new cycle starts here
        self._eager_exchange = self._micro == self.accumulate - 1
        if self._eager_exchange:
            A.reducer = r
        else:
            A.reducer = None
            r.muted = True

    def _eager_backward_ends(self) -> None:
        A, r = self.arena, self.reducer
        if self._eager_exchange:
            r.finish()
            self._micro = 0
        else:
            A.reducer = r
            r.muted = False
            self._micro += 1

    def close(self) -> None:
        d = self.__dict__
        if d.get("_caps") is not None:
            for c in d["_caps"].values():
                c.close()
        d["_caps"] = None
        if d.get("nonce") is not None:
            try:
                K.clear_dropout_nonce_if(d["nonce"])
            except Exception:      # interpreter shutdown
                pass
            d["nonce"] = None

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001
            pass


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

    def abandon(self) -> None:
        """End a capture that raised half-way, so that the stream leaves capture mode; the graphs are dropped."""
        if self.graph is not None:
            try:
                self.graph.capture_end()
            except Exception:   # noqa: BLE001
                pass
            self.graph = None
        self.segments, self.ready = [], []


class SegmentedStep(_StepBase):
    """Data-parallel step as a chain of linear hipGraphs with eager bucket all-reduces between them (module docstring).
    ``step_fn`` runs forward + backward + ``reducer.finish()`` and returns the loss; ``reducer`` must already be
    attached to the model's arena.  ``inputs`` / per-call refresh as ``GraphedStep``; every call overwrites the gradients
    and exchanges them (no accumulation across calls: the fallback form -- ``FlaggedStep`` has ``accumulate=``)."""

    def __init__(self, model: torch.nn.Module, step_fn: Callable[..., torch.Tensor], reducer, warmup: int = 3, inputs=None):
        self._setup(model, step_fn, inputs)
        self.reducer = reducer
        self.segments = None
        dev, side = self.device, self.side
        self._warm(max(warmup, 2))              # collective phase, the same on every rank (the first step calibrates the reducer)
        self.arena = model._icka_arena
        model.zero_grad()                       # gradients dropped -> captured kernels overwrite (beta = 0)
        # ---- capture phase: no process-group traffic (a ready bucket cuts the graph), may fail on one rank alone
        cap = _Capture()
        reducer.capture = cap
        err = None
        try:
            # backward on THIS thread: begin / end of a stream capture must come from one thread, and the cuts happen
            # inside backward (GradReducer.mark_final)
            with torch.cuda.stream(side), torch.autograd.set_multithreading_enabled(False):
                cap.begin()
                try:
                    K.bump_dropout_nonce(self.nonce)
                    self.loss = self._step()    # its reducer.finish() reports the remaining buckets, without joining
                    # the embedding backward is the last kernel of the step and makes the last bucket final: the segment
                    # opened by that cut would be empty -- give it one (no-op) node
                    self._pad = torch.zeros(2, dtype=torch.int32, device=dev)
                    K.bump_dropout_nonce(self._pad)
                    _test_fail("segmented")
                    cap.end()
                except Exception:
                    cap.abandon()
                    raise
        except Exception as e:  # noqa: BLE001
            err = e
        finally:
            reducer.capture = None
        _end_capture_quietly()
        torch.cuda.current_stream().wait_stream(side)
        from .dp import CaptureDisagreement, all_ranks_agree
        if not all_ranks_agree(err is None, reducer.group, "SegmentedStep capture"):
            reducer.abort_step()
            self.arena.reducer = reducer
            model.zero_grad()
            self._close_nonce()
            if err is None:
                raise CaptureDisagreement("SegmentedStep: another rank could not capture the step")
            raise CaptureDisagreement("SegmentedStep: this rank could not capture the step (%s: %s)" % (type(err).__name__, err)) from err
        self.segments = cap.segments
        # row-sparse word-table exchange: the static row / id buffers the captured backward fills -- or, when the step's
        # embedding backward wrote the table densely (fp32-exact mode), a dense mean all-reduce of the slot per replay
        self._sparse_args, self._word_written = None, False
        if getattr(reducer, "sparse_word", None) is not None:
            self._sparse_args, reducer._sparse = reducer._sparse, None
            self._word_written, reducer._word_written = reducer._word_written or self._sparse_args is not None, False
        self._grad_slots = [s for s in self.arena.order if s.live]
        model.zero_grad()                       # a capture executes nothing: the gradients it "wrote" do not exist

    def __call__(self, *args, **kwargs) -> torch.Tensor:
        if self.segments is None:
            raise RuntimeError("SegmentedStep is closed")
        self._refresh(args, kwargs)
        if self.arena.shadow_policy != "always":
            self.arena.sync()
        K.lstm_check_error("detected before a SegmentedStep replay")
        K.gemm_ln_check_error("detected before a SegmentedStep replay")
        for graph, buckets in self.segments:
            graph.replay()
            for bi in buckets:
                self.reducer.launch_now(bi)
        if self._word_written:
            if self._sparse_args is not None:
                self.reducer.set_sparse_rows(*self._sparse_args)
            self.reducer.exchange_sparse(word_written=True)
        self.reducer.join()
        self.arena.attach_grads(self._grad_slots)
        return self.loss

    def close(self) -> None:
        self._close_nonce()
        self.segments = None


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


class _FlagExchange(object):
    """The replay-side half of the flag-word protocol (``FlaggedStep`` docstring) for a capture that is not a whole step
    (``GraphedModule``'s backward): the sync words, the tag of the next exchanging replay, and the flag waits + eager all-reduces
    + join + final poison pass behind a replay."""

    def __init__(self, reducer, arena, device):
        import os
        nb = len(reducer.buckets)
        self.reducer, self.arena = reducer, arena
        self.sync = torch.zeros(FlaggedStep.FLAG0 + 2 * nb + 16, dtype=torch.int32, device=device)
        self.starts = torch.tensor([lo for lo, _ in reducer.buckets], dtype=torch.int64, device=device)
        K.check(K._lib.load().icka_dp_init(), "icka_dp_init")
        self.polls = int(os.environ.get("ICKA_DP_WAIT_POLLS", FlaggedStep.WAIT_POLLS))
        self.flag_ptr = [self.sync.data_ptr() + 4 * (FlaggedStep.FLAG0 + i) for i in range(nb)]
        self.bad_ptr = [self.sync.data_ptr() + 4 * (FlaggedStep.FLAG0 + nb + i) for i in range(nb)]
        self.tag = 0

    def first_node(self) -> None:
        K.check(K._lib.load().icka_dp_step_bump(self.sync.data_ptr(), K._stream()), "icka_dp_step_bump")

    def before_replay(self) -> None:
        self.tag += 1                           # == the step counter the capture's first node is about to write

    def after_replay(self, order) -> None:
        r = self.reducer
        for idx in order:
            r.launch_now(idx, wait=(self.flag_ptr[idx], self.tag, self.polls, self.bad_ptr[idx]))
        nb = len(r.buckets)
        r.share_bad_words(self.sync[FlaggedStep.FLAG0 + nb:FlaggedStep.FLAG0 + 2 * nb])
        r.join()
        K.check(K._lib.load().icka_dp_poison_final(self.bad_ptr[0], len(r.buckets), self.tag & 0xFFFFFFFF, self.arena.gflat.data_ptr(),
                                                  self.starts.data_ptr(), 8, K._stream()), "icka_dp_poison_final")


class FlaggedStep(_StepBase):
    """Data-parallel step as ONE hipGraph with eager, overlapped all-reduces (the default of bench.py at N > 1).

    ``SegmentedStep`` pays for every cut (a hipGraph drains before the next one starts: +0.21 ms for 5 cuts at c2) and an
    external event inside a graph is refused by this ROCm build.  Here the step -- forward, backward and, at every point
    where a gradient bucket becomes final, a one-thread node that stores the step number into the bucket's FLAG WORD
    (icka_dp_flag_set) -- is captured whole.  A replay launches the graph on the compute stream and then, per bucket in the
    order the flags will rise, enqueues on the reducer's communication stream a one-wave kernel that waits for that flag
    (icka_dp_flag_wait: bounded spin with s_sleep; a wait that gives up raises a host-visible error word and the bucket's
    cast / cast-back launches then fill it with NaN, dp.GradReducer._allreduce) followed by the bucket's eager all-reduce --
    the tagged-word hand-off of csrc/lstm.hip between two streams.  No collective is captured, no graph has a second branch,
    the compute stream never waits for the communication stream before the end of the step.
    ``step_fn`` runs forward + backward + ``reducer.finish()`` and returns the loss; ``reducer`` is attached to the arena.

    ``inputs`` / per-call refresh as ``GraphedStep``.  ``accumulate=k`` (the reference's gradient_accumulation_steps,
    My_cross_attention.py:587-590, :831): a cycle is k calls; calls 1..k-1 replay graphs captured WITHOUT the reducer
    (overwrite, then accumulate: no flags, no wire copies, nothing exchanged) and the k-th replays the flagged graph, whose
    gradient stores accumulate (k > 1) and whose buckets carry the sum of the k micro-batches.  A cycle also restarts
    whenever the caller dropped the gradients (``zero_grad``) before its k-th call."""

    FLAG0 = 16           # sync words: [0] step counter (bumped by the graph's first node), [FLAG0 + i] flag of bucket i
    WAIT_POLLS = 1 << 20   # ~3 s of s_sleep(64) polls before a wait gives up

    def __init__(self, model: torch.nn.Module, step_fn: Callable[..., torch.Tensor], reducer, warmup: int = 3, inputs=None,
                 accumulate: int = 1):
        if not (reducer.is_cuda and reducer.backend == "nccl"):
            raise RuntimeError("FlaggedStep needs the nccl (= RCCL) backend on a ROCm device")
        if accumulate < 1:
            raise ValueError("accumulate must be >= 1")
        if accumulate > 1 and getattr(reducer, "sparse_word", None) is not None:
            raise ValueError("FlaggedStep(accumulate > 1) cannot be combined with GradReducer(sparse_embeddings=True): the token "
                             "rows of the micro-batches that do not exchange would stay local")
        self._setup(model, step_fn, inputs)
        self.reducer = reducer
        self.accumulate = int(accumulate)
        self._graphs = None
        self.graph = None
        dev = self.device
        nb = len(reducer.buckets)
        # sync words: [0] step counter, [FLAG0 + i] flag of bucket i, [FLAG0 + nb + i] BAD word of bucket i (the step number
        # of a wait that gave up)
        self.sync = torch.zeros(self.FLAG0 + 2 * nb + 16, dtype=torch.int32, device=dev)
        self._starts = torch.tensor([lo for lo, _ in reducer.buckets], dtype=torch.int64, device=dev)
        K.check(K._lib.load().icka_dp_init(), "icka_dp_init")
        self._warm(max(warmup, 2))              # collective phase, the same on every rank (the first step calibrates the reducer)
        self.arena = model._icka_arena
        self._polls = int(os.environ.get("ICKA_DP_WAIT_POLLS", self.WAIT_POLLS))
        self._flag_ptr = [self.sync.data_ptr() + 4 * (self.FLAG0 + i) for i in range(nb)]
        self._bad_ptr = [self.sync.data_ptr() + 4 * (self.FLAG0 + nb + i) for i in range(nb)]
        self._test_late = ()     # tests: buckets whose wait is made to give up (it polls a word that no node ever sets)
        self._never_ptr = self.sync.data_ptr() + 4 * (self.FLAG0 + 2 * nb + 8)
        self._tag = 0
        self._micro = 0
        self._graphs = {}        # (accumulate, exchange) -> CUDAGraph
        self._loss = {}
        self._order = {}         # exchange graphs: the order their flags rise in
        self._sparse_args = {}
        model.zero_grad()                       # gradients dropped -> captured kernels overwrite (beta = 0)
        first = (False, self.accumulate == 1)
        self._capture_agreed(*first)            # local capture phase, then the vote: raises on EVERY rank or on none
        self._grad_slots = [s for s in self.arena.order if s.live]
        self.loss = self._loss[first]
        self.graph = self._graphs[first]
        self.order = self._order.get(first, [])
        model.zero_grad()                       # a capture executes nothing: the gradients it "wrote" do not exist

    def _capture_agreed(self, accumulate: bool, exchange: bool) -> None:
        """``_capture`` (which issues no process-group traffic: flag nodes only) + the vote over the store: unless the capture
        worked on every rank, every rank raises ``dp.CaptureDisagreement`` -- the ranks fall back together."""
        from .dp import CaptureDisagreement, all_ranks_agree
        err = None
        try:
            self._capture(accumulate, exchange)
        except Exception as e:  # noqa: BLE001
            err = e
            _end_capture_quietly()
        if all_ranks_agree(err is None, self.reducer.group, "FlaggedStep capture (accumulate=%s, exchange=%s)" % (accumulate, exchange)):
            return
        key = (accumulate, exchange)
        self._graphs.pop(key, None)
        self._loss.pop(key, None)
        self.reducer.abort_step()
        self.arena.reducer = self.reducer
        if not self._graphs:                    # the first capture: this object never came to life
            self.model.zero_grad()
            self.close()
        if err is None:
            raise CaptureDisagreement("FlaggedStep: another rank could not capture the step")
        raise CaptureDisagreement("FlaggedStep: this rank could not capture the step (%s: %s)" % (type(err).__name__, err)) from err

    def _capture(self, accumulate: bool, exchange: bool) -> None:
        reducer, arena = self.reducer, self.arena
        if accumulate:
            arena.attach_grads(self._grad_slots)
        g = torch.cuda.CUDAGraph()
        key = (accumulate, exchange)
        if exchange:
            cap = _FlagCapture(self.sync)
            reducer.capture = cap
            arena.reducer = reducer
            try:
                # backward on THIS thread: the flag nodes are launched from inside backward (GradReducer.mark_final)
                with torch.autograd.set_multithreading_enabled(False), \
                        torch.cuda.graph(g, stream=self.side, capture_error_mode="thread_local"):
                    K.check(K._lib.load().icka_dp_step_bump(self.sync.data_ptr(), K._stream()), "icka_dp_step_bump")
                    K.bump_dropout_nonce(self.nonce)
                    self._loss[key] = self._step()   # its reducer.finish() reports the remaining buckets (flags at the end)
                    _test_fail("flagged")
            finally:
                reducer.capture = None
            torch.cuda.synchronize()
            if sorted(cap.order) != list(range(len(reducer.buckets))):
                raise RuntimeError("FlaggedStep: buckets flagged during capture %s != all %d buckets"
                                   % (cap.order, len(reducer.buckets)))
            self._order[key] = cap.order
            # (the captured embedding backward registered its static row / id buffers: the same ones every replay fills; a step
            # whose embedding backward wrote the table densely -- the fp32-exact mode -- leaves None: dense mean of the slot)
            if getattr(reducer, "sparse_word", None) is not None:
                self._sparse_args[key] = reducer._sparse
                reducer._sparse, reducer._word_written = None, False
        else:
            # a micro-batch that is not the last of its cycle: the same step with the reducer detached (no wire copies, no
            # flags) and muted (the step_fn's reducer.finish() returns at once for the duration of the capture)
            arena.reducer = None
            reducer.muted = True                # finish() returns at once: nothing is exchanged, no bucket state moves
            try:
                with torch.autograd.set_multithreading_enabled(False), \
                        torch.cuda.graph(g, stream=self.side, capture_error_mode="thread_local"):
                    K.bump_dropout_nonce(self.nonce)
                    self._loss[key] = self._step()
            finally:
                arena.reducer = reducer
                reducer.muted = False
            torch.cuda.synchronize()
        self._graphs[key] = g

    def __call__(self, *args, **kwargs) -> torch.Tensor:
        if self._graphs is None:
            raise RuntimeError("FlaggedStep is closed")
        self._refresh(args, kwargs)
        if self.arena.shadow_policy != "always":
            self.arena.sync()
        K.lstm_check_error("detected before a FlaggedStep replay")
        K.gemm_ln_check_error("detected before a FlaggedStep replay")
        K.dp_check_error("detected before a FlaggedStep replay")
        accumulate, stale = self._cycle_state()
        if not accumulate:
            self._micro = 0                     # the caller dropped the gradients: a new cycle starts here
        exchange = self._micro == self.accumulate - 1
        key = (accumulate, exchange)
        if key not in self._graphs:
            self._capture_agreed(*key)
        self._zero(stale)
        if exchange:
            self._tag += 1                      # == the step counter the graph's first node is about to write
        self._graphs[key].replay()
        if exchange:
            r = self.reducer
            for idx in self._order[key]:
                late = idx in self._test_late
                r.launch_now(idx, wait=(self._never_ptr if late else self._flag_ptr[idx], self._tag,
                                        64 if late else self._polls, self._bad_ptr[idx]))
            if getattr(r, "sparse_word", None) is not None:
                # the row-sparse word-table exchange: on the communication stream BEHIND the last bucket's flag wait (the
                # embedding backward, which leaves the rows, is what makes that bucket final)
                if self._sparse_args[key] is not None:
                    r.set_sparse_rows(*self._sparse_args[key])
                with torch.cuda.stream(r.comm_stream):
                    r._exchange_on_current = True
                    try:
                        r.exchange_sparse(word_written=True)
                    finally:
                        r._exchange_on_current = False
            nb = len(r.buckets)
            r.share_bad_words(self.sync[self.FLAG0 + nb:self.FLAG0 + 2 * nb])
            r.join()
            # after the join nothing of this step writes gradients any more: a bucket whose wait gave up gets its NaN here
            # for good (one launch; the bad words carry the step number, so nothing is ever reset)
            K.check(K._lib.load().icka_dp_poison_final(self._bad_ptr[0], nb, self._tag & 0xFFFFFFFF, self.arena.gflat.data_ptr(),
                                                      self._starts.data_ptr(), 8, K._stream()), "icka_dp_poison_final")
            self._micro = 0
        else:
            self._micro += 1
        self.arena.attach_grads(self._grad_slots)
        self.loss = self._loss[key]
        return self.loss

    def close(self) -> None:
        self._close_nonce()
        self._graphs = None
        self.graph = None


class _ComputeThenReduce(object):
    """Fallback form of a data-parallel step: forward + backward replayed from ONE capture taken without the reducer, the bucket
    all-reduces launched eagerly after each replay (no overlap, but no host-bound step)."""

    def __init__(self, cap: _StepCapture, reducer):
        self.cap, self.reducer = cap, reducer
        self.inputs, self.nonce = cap.inputs, cap.nonce
        reducer.forget_wire_copies()            # the captured GEMMs wrote none: the bucket casts cover everything

    def __call__(self, *args, **kwargs):
        loss = self.cap(*args, **kwargs)
        self.reducer.reduce_all()
        return loss

    def close(self) -> None:
        self.cap.close()


def build_step(model: torch.nn.Module, step_fn: Callable[..., torch.Tensor], inputs=None, reducer=None, accumulate: int = 1,
               prefer: str = "flagged", graph: bool = True, capture_collectives: bool = False, log=None):
    """The ONE decision point for the form a training step runs in: returns ``(step, mode)`` with ``step(*batch) -> loss``.

    ``step_fn(*inputs)`` runs forward + backward (+ ``reducer.finish()`` under data parallelism) and returns the loss; the
    model has run at least one eager step (its arena exists; the reducer, if any, is attached to it).  Forms, in order:

      no reducer   : ``GraphedStep`` (one hipGraph, capture cache)                         -> eager ``step_fn``
      with reducer : ``FlaggedStep`` (RCCL; ``prefer="flagged"``) -> ``SegmentedStep`` -> one capture of forward + backward
                     with the reducer detached + eager all-reduces after each replay       -> eager ``step_fn``

    Under a reducer every transition is COLLECTIVE: each form's constructor runs its warm-up (gradient exchanges, the same on
    all ranks), captures in a phase that issues no process-group traffic, and votes over the c10d store
    (dp.all_ranks_agree); unless all ranks captured, all of them get ``dp.CaptureDisagreement`` and move to the next form
    together.  No rank ever builds another form than its peers (the reference has one code path per rank,
    My_cross_attention.py:653-657, :768-776)."""
    say = log or (lambda msg: None)
    if not graph:
        return step_fn, "eager"
    arena = model._icka_arena
    if reducer is None:
        try:
            say("capturing the step into a hipGraph")
            return GraphedStep(model, step_fn, inputs=inputs), "hipgraph"
        except Exception as e:  # noqa: BLE001
            say("graph capture failed (%s: %s); running eagerly" % (type(e).__name__, e))
            _end_capture_quietly()
            return step_fn, "eager"
    from .dp import CaptureDisagreement, all_ranks_agree
    nccl = reducer.is_cuda and reducer.backend == "nccl"
    if capture_collectives:
        # the all-reduces INSIDE one hipGraph (side-stream branches): needs a capturable collective library
        err, st = None, None
        try:
            say("capturing the step with its collectives into one hipGraph")
            st = _StepCapture(model, step_fn, inputs=inputs)
        except Exception as e:  # noqa: BLE001
            err = e
            _end_capture_quietly()
        if all_ranks_agree(err is None, reducer.group, "captured-collectives step"):
            return st, "hipgraph(collectives captured, %d buckets)" % len(reducer.buckets)
        say("capture with collectives failed on %s" % ("this rank (%s: %s)" % (type(err).__name__, err) if err else "another rank"))
        reducer.abort_step()
        arena.reducer = reducer
    else:
        forms = (["flagged"] if (prefer == "flagged" and nccl) else []) + ["segmented"]
        for form in forms:
            try:
                if form == "flagged":
                    say("capturing the step as one hipGraph with bucket-ready flags (eager all-reduces on the communication stream)")
                    st = FlaggedStep(model, step_fn, reducer, inputs=inputs, accumulate=accumulate)
                    return st, "hipgraph+flag-waits+eager-allreduce(%d buckets, overlapped)" % len(reducer.buckets)
                say("capturing the step as linear hipGraph segments (eager all-reduces in between)")
                st = SegmentedStep(model, step_fn, reducer, inputs=inputs)
                return st, "hipgraph-segments(%d)+eager-allreduce(%d buckets, overlapped)" % (len(st.segments), len(reducer.buckets))
            except CaptureDisagreement as e:       # raised on every rank alike
                say("%s capture not taken: %s" % (form, e))
    # last graph form: forward + backward only, captured with the reducer detached and muted (step_fn's finish() returns at once)
    err, cap = None, None
    arena.reducer = None
    reducer.muted = True
    try:
        say("capturing forward + backward only (the gradient all-reduce runs eagerly after each replay)")
        cap = _StepCapture(model, step_fn, inputs=inputs)
    except Exception as e:  # noqa: BLE001
        err = e
        _end_capture_quietly()
    finally:
        arena.reducer = reducer
        reducer.muted = False
    if all_ranks_agree(err is None, reducer.group, "compute-only capture"):
        return _ComputeThenReduce(cap, reducer), "hipgraph(compute)+eager-allreduce"
    say("compute-only capture failed on %s; running eagerly"
        % ("this rank (%s: %s)" % (type(err).__name__, err) if err else "another rank"))
    if cap is not None:
        cap.close()
    reducer.abort_step()
    return step_fn, "eager"
