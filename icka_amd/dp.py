"""Data-parallel gradient reduction over RCCL / xGMI (one process per GPU).

The hot path shards over (sentence, image) pairs: every rank holds a full weight replica and
processes its own micro-batch; the only exchange is one gradient all-reduce (mean) per optimisation step -- the
counterpart of the reference's apex DistributedDataParallel (My_cross_attention.py:768-776; process group :653-657).

Because all gradients live in ONE flat fp32 buffer (ParamArena.gflat), laid out in execution order, the reduction is
a handful of large contiguous all-reduces instead of ~200 small ones: buckets are slices of that buffer, taken from
its END (the parameters whose gradients the backward pass finishes first) towards its start, with the embedding
tables (whose gradient is final only at the very end of backward) in a bucket of their own.  A bucket's all-reduce is
launched on a side stream as soon as every gradient in it is final (``ParamArena.flush_final`` -> ``mark_final``),
overlapping RCCL traffic over xGMI with the remaining backward kernels; ``finish`` joins the streams.

Wire format.  ``comm_dtype="f32"`` (the default: what the reference's apex DDP exchanges) all-reduces the gradient buffer in
place (AVG).  ``comm_dtype="bf16"`` (ROCm devices; what bench.py runs: 239 MB instead of 477 MB per step for the bert-base
path, which is what puts the exchange under the backward it overlaps with) keeps a bf16 WIRE BUFFER with
the layout of the gradient buffer:
  * the weight-gradient GEMMs write the wire copy of every matrix gradient from their own epilogue (``ParamArena.wire_of``
    -> icka_gemm_desc.C3): no cast pass over the 85 M matrix gradients of bert-base;
  * what no GEMM produces (bias / LayerNorm vectors, classifier, embedding tables) is cast by ONE launch per bucket over a
    chunk table (icka_dp_cast_chunks), on the side stream;
  * the all-reduce is a SUM of the bucket's wire slice (bf16 on the wire), and one launch brings the bucket back into the f32
    gradient buffer with the 1 / world factor folded in (icka_dp_cast_back_scaled).
The N-rank bf16 ring sum rounds once per hop; tests/test_dp_gloo_cpu.py bounds that error against the fp32 mean.

Row-sparse word-embedding exchange (``sparse_embeddings=True``).  The word-embedding table is 20 % of bert-base's parameters
(23.4 M of 119 M), its gradient is final only with the LAST kernel of backward, and a step touches at most B x S of its rows:
instead of all-reducing the dense [vocab, H] gradient the ranks all-gather their B x S token-gradient rows (wire dtype) and
ids and every rank scatter-adds all N x B x S rows with 1 / N folded in (icka_embed_bwd_rows + icka_embed_scatter_rows); a rank
that brings at least vocab / 4 rows falls back to a dense all-reduce of the slot.  f32 atomics: the sum order differs from the
dense path's in the last bits (tests bound it at 1e-6).  Not combined with accumulation inside a captured step
(FlaggedStep(accumulate > 1) refuses it: the earlier micro-batches' rows would stay local).  Priced in
profiles/r04_dp_budget.md: it halves the bytes of the exposed tail at N <= 4 and is about even at N = 8 x 4096 tokens.

A slot counts as final when it has received ALL the gradient contributions it received in the calibration (first)
step -- a module applied twice per forward contributes twice -- so a bucket is never reduced while a late write into it
is still to come; a write that arrives for a bucket already in flight raises.  Works with the ``nccl`` (= RCCL) backend on
ROCm devices and with ``gloo`` (CPU tensors; device tensors are staged through the host: tests only).
"""
from __future__ import annotations

import sys
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from .arena import ParamArena


_GATHER_SEQ: Dict[str, int] = {}


def all_ranks_gather(value, group=None, what: str = "value", timeout: float = 600.0) -> list:
    """Every rank's ``value`` (JSON-serialisable: bool / int / str / list), in rank order, exchanged over the c10d key-value
    STORE of the process group -- not over a collective.  Host-only: nothing is enqueued on a GPU stream and nothing travels over
    the backend, so ranks may call it while their device state differs (one of them has just failed a stream capture, another
    still holds a finished one) -- the situation in which a backend collective pairs with the wrong call and aborts
    (gloo ``collective mismatch``; a hang under RCCL).  The ranks must call it the same number of times in the same order
    (a per-group call counter names the keys); a rank that does not arrive within ``timeout`` seconds raises here on the others."""
    if not dist.is_initialized():
        return [value]
    world = dist.get_world_size(group)
    if world == 1:
        return [value]
    import json
    from datetime import timedelta
    from torch.distributed import distributed_c10d as c10d
    rank = dist.get_rank(group)
    store = c10d._get_default_store()
    members = dist.get_process_group_ranks(group if group is not None else dist.group.WORLD)
    gkey = "-".join(str(r) for r in members)
    seq = _GATHER_SEQ[gkey] = _GATHER_SEQ.get(gkey, 0) + 1
    base = "icka_amd/gather/%s/%d" % (gkey, seq)
    store.set("%s/%d" % (base, rank), json.dumps(value))
    keys = ["%s/%d" % (base, r) for r in range(world)]
    try:
        store.wait(keys, timedelta(seconds=timeout))
    except Exception as e:  # noqa: BLE001
        raise RuntimeError("icka_amd.dp: not every rank reached the agreement point %r (#%d) within %.0f s: %s"
                           % (what, seq, timeout, e)) from e
    return [json.loads(store.get(k).decode()) for k in keys]


def all_ranks_agree(ok: bool, group=None, what: str = "vote") -> bool:
    """True iff ``ok`` holds on EVERY rank of the group (``all_ranks_gather`` over the store: no collective, no device work).
    The single decision point of everything that may succeed on one rank and fail on another -- above all a hipGraph capture:
    each rank tries in a phase that issues no process-group traffic, then all vote, then all build the agreed form."""
    return all(bool(v) for v in all_ranks_gather(bool(ok), group, what))


class CaptureDisagreement(RuntimeError):
    """A captured data-parallel step could not be built on EVERY rank (raised on all of them, also on those whose own capture
    succeeded): the ranks take the same fallback together."""


class _NullCtx(object):
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

# CUs dp.GradReducer keeps free of persistent BiLSTM blocks while RCCL workgroups may share the GPU (icka_hip.h:
# icka_lstm_set_reserved_cus): RCCL runs one workgroup per channel, at most 64 channels
LSTM_RESERVED_CUS = 64


class GradReducer(object):
    def __init__(self, arena: ParamArena, group=None, bucket_mb: float = 64.0, comm_dtype: Optional[str] = None,
                 comm_bf16: Optional[bool] = None, diag: str = "", lstm_reserved_cus: Optional[int] = None,
                 sparse_embeddings: bool = False):
        """``bucket_mb``: minimum bucket size in MB of fp32 gradients.  ``comm_dtype``: "f32" (default) | "bf16" (ROCm devices
        only; module docstring).  ``comm_bf16`` is the round-1 spelling of the same switch.  ``diag``: DIAGNOSTIC ONLY --
        any of "cast" / "comm" (comma-separated), or "none": leave parts of the exchange out to price the others
        (tools/fd_sweep.sh, bench.py --dp-diag); the gradients are then WRONG and every construction says so on stderr."""
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
            comm_dtype = "f32"
        if comm_dtype not in ("bf16", "f32"):
            raise ValueError("comm_dtype must be 'bf16' or 'f32'")
        if comm_dtype == "bf16" and not self.is_cuda:
            raise ValueError("bf16 buckets need a ROCm device (the casts are HIP kernels)")
        self.comm_bf16 = comm_dtype == "bf16"
        self.diag = diag or ""
        if self.diag:
            print("[icka_amd.dp] DIAGNOSTIC MODE diag=%r: parts of the gradient exchange are skipped -- the gradients of "
                  "this process are WRONG (pricing runs only)" % self.diag, file=sys.stderr, flush=True)
        self.buckets: List[Tuple[int, int]] = arena.buckets(int(bucket_mb * (1 << 20) / 4))
        # ---- row-sparse exchange of the word-embedding gradient (module docstring): the largest embedding table leaves the
        #      dense buckets; its gradient travels as the ranks' token rows (set_sparse_rows / exchange_sparse)
        self.sparse_word = None
        self._sparse = None          # (rows f32 [T, H], ids int64 [T], accumulate) of the step in flight
        self._sparse_ws = {}
        self.sparse_stats = {"sparse": 0, "dense_fallback": 0}
        if sparse_embeddings:
            tables = [sl for sl in arena.order if sl.is_table]
            word = max(tables, key=lambda sl: sl.shape[0]) if tables else None
            if word is not None:
                lo_w, hi_w = word.off, word.off + (word.numel + 7) // 8 * 8
                cut = None
                for bi, (lo, hi) in enumerate(self.buckets):
                    if lo == lo_w and hi_w <= hi:
                        cut = (bi, (hi_w, hi))
                    elif hi == hi_w and lo <= lo_w:
                        cut = (bi, (lo, lo_w))
                if cut is None:
                    print("[icka_amd.dp] sparse_embeddings: %s is not at an edge of a gradient bucket: dense exchange kept"
                          % word.name, file=sys.stderr, flush=True)
                else:
                    bi, (lo, hi) = cut
                    if hi > lo:
                        self.buckets[bi] = (lo, hi)
                    else:
                        del self.buckets[bi]
                    self.sparse_word = word
        self.comm_stream = torch.cuda.Stream(device=arena.device) if self.is_cuda else None
        # slot -> bucket index
        self._bucket_of: Dict[int, int] = {}
        for s in arena.order:
            for bi, (lo, hi) in enumerate(self.buckets):
                if lo <= s.off < hi:
                    self._bucket_of[id(s)] = bi
                    break
        self._word_written = False   # the sparse word table's slot received a (dense) gradient write in the step in flight
        self._agreed_T = set()       # local row counts whose path / buffer sizes the ranks have agreed on (exchange_sparse)
        self._poison_off = [lo for lo, _ in self.buckets]   # where a bucket's give-up NaN goes (see _build_tables)
        self._calibrated = False
        self._expected: Dict[int, int] = {}       # slot -> gradient writes per step (counted in the calibration step)
        self._seen: Dict[int, int] = {}
        self._waiting = [0] * len(self.buckets)   # per bucket: slots that have not received all their writes yet
        self._launched = [False] * len(self.buckets)
        # graph capture (graph.SegmentedStep / graph.FlaggedStep): while ``capture`` is set, a bucket that becomes ready is
        # not launched but reported to it (``bucket_ready(idx)``, then ``after_mark()`` once per mark_final call)
        self.capture = None
        self.muted = False       # graph.FlaggedStep, captures of the micro-batches that do not exchange: finish() is a no-op
        # ---- bf16 wire buffer (module docstring)
        self.gwire = None
        self._wire_ranges = set()                 # (offset, numel) of gradients whose wire copy a GEMM epilogue writes
        self._wire_only = set()                   # ... those of them that are written once per step (see wire_view)
        self._tables: List[Optional[torch.Tensor]] = [None] * len(self.buckets)
        if self.comm_bf16:
            self.gwire = torch.zeros(arena.total, dtype=torch.bfloat16, device=arena.device)
            from . import kernels as K
            K._lib.load().icka_dp_init()
            for bi in range(len(self.buckets)):   # until the calibration step has shown who writes wire copies: cast all
                self._tables[bi] = K.dp_chunk_table([self.buckets[bi]], arena.device)
        # persistent BiLSTM grids must stay co-resident beside RCCL's workgroups: reserve CUs for as long as this reducer
        # lives (close() / __del__ restore what was reserved before)
        self._prev_reserved = None
        # (any backend: over gloo the ranks of a test or rehearsal usually SHARE one card, where launches whose blocks wait for each
        #  other -- the persistent BiLSTM, the fused dense + LayerNorm -- can no longer count on the whole chip either)
        if (self.is_cuda and self.world > 1) or lstm_reserved_cus is not None:
            from . import kernels as K
            self._prev_reserved = K.lstm_set_reserved_cus(LSTM_RESERVED_CUS if lstm_reserved_cus is None else lstm_reserved_cus)

    def abort_step(self) -> None:
        """Forget the step in flight (a capture or an eager step that raised half-way): no bucket is marked launched, no
        write counted, no rows pending, no capture protocol attached.  The calibration is kept."""
        self.capture = None
        self.muted = False
        self._sparse = None
        self._word_written = False
        self._seen = {}
        self._waiting = [0] * len(self.buckets)
        for sid in self._expected:
            self._waiting[self._bucket_of[sid]] += 1
        self._launched = [False] * len(self.buckets)

    def close(self) -> None:
        """Detach from the arena and give back the CU reservation taken for the persistent BiLSTM kernels."""
        if getattr(self, "_prev_reserved", None) is not None:
            try:
                from . import kernels as K
                K.lstm_set_reserved_cus(self._prev_reserved)
            except Exception:      # interpreter shutdown
                pass
            self._prev_reserved = None
        a = getattr(self, "arena", None)
        if a is not None and getattr(a, "reducer", None) is self:
            a.reducer = None

    def __del__(self):
        self.close()

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

    # ------------------------------------------------------------------------------------------------- wire copies
    def wire_view(self, off: int, numel: int, shape):
        """(wire, only): the bf16 wire slice [off, off + numel) shaped like the gradient view a GEMM is about to write
        (ParamArena.wire_of) -- noted, so that the bucket's cast launch skips it once calibration is over -- and whether the
        GEMM may store ONLY the wire copy (icka_gemm_desc.c3_only): after calibration, for gradients written exactly once
        per step.  Their f32 value is produced by the cast-back of the reduced bucket, which every step of a reducer runs
        before anything reads the gradient buffer; the 4-byte store of a value nobody reads is dropped from the epilogue."""
        if self.gwire is None:
            return None, False
        key = (off, numel)
        self._wire_ranges.add(key)
        return self.gwire[off:off + numel].view(shape), key in self._wire_only

    def _build_tables(self) -> None:
        """Chunk tables of what still has to be cast per bucket: the bucket minus the ranges GEMM epilogues fill."""
        from . import kernels as K
        wired = sorted(self._wire_ranges)
        for bi, (lo, hi) in enumerate(self.buckets):
            ranges, cur = [], lo
            for off, n in wired:
                end = off + (n + 7) // 8 * 8          # slots are padded to 8 elements: the pad belongs to nobody
                if end <= lo or off >= hi:
                    continue
                if off > cur:
                    ranges.append((cur, off))
                cur = max(cur, end)
            if cur < hi:
                ranges.append((cur, hi))
            self._tables[bi] = K.dp_chunk_table(ranges, self.arena.device)
            # the NaN of a flag wait that gave up goes where NO GEMM epilogue writes its wire copy (a bias / LayerNorm / table
            # range: their wire values come from the cast launch, which runs on the communication stream BEFORE the poison), so
            # a weight-gradient store of the still-running graph cannot overwrite it before RCCL reads the bucket: every rank
            # receives the NaN through the sum
            for a, b in ranges:
                if b - a >= 8:
                    self._poison_off[bi] = a
                    break
        # gradients written exactly once per step may drop their f32 store (wire_view)
        once = {}
        for s in self.arena.order:
            once[s.off] = (s, self._expected.get(id(s), 0) == 1)
        self._wire_only = set()
        for off, n in wired:
            ok, cur = True, off
            while cur < off + n:                      # every slot the range covers (fused q|k|v: three)
                ent = once.get(cur)
                if ent is None or not ent[1]:
                    ok = False
                    break
                cur += (ent[0].numel + 7) // 8 * 8
            if ok and cur >= off + n:
                self._wire_only.add((off, n))

    def forget_wire_copies(self) -> None:
        """From now on the cast launches cover the WHOLE of every bucket and no GEMM may drop its f32 store: for a step whose
        weight-gradient GEMMs run without this reducer attached (graph.build_step's compute-only capture) and therefore write
        no wire copies."""
        self._wire_ranges = set()
        self._wire_only = set()
        if self.gwire is not None:
            self._build_tables()

    def cast_elements(self) -> int:
        """Elements per step that still go through the cast launches (diagnostics / DESIGN.md)."""
        return sum(int(t[:, 1].sum().item()) for t in self._tables if t is not None and t.numel())

    # ------------------------------------------------------------------------------------------------- launches
    def _launch(self, idx: int) -> None:
        self._launched[idx] = True
        if self.capture is not None:
            self.capture.bucket_ready(idx)
            return
        self.launch_now(idx)

    def launch_now(self, idx: int, wait=None) -> None:
        """Issue bucket ``idx``'s all-reduce: on the side stream on devices, ordered after everything on the current
        stream -- or, with ``wait = (device address of the flag word, tag, poll budget, device address of the bucket's bad
        word)`` (graph.FlaggedStep), after a flag-wait kernel on the side stream and NOT after the current stream (which
        already holds the whole step's graph).  A wait that gives up stores ``tag`` into the bad word; ``_allreduce`` then
        poisons the bucket AFTER its chunk cast (so the NaN is what travels) and again after the cast-back."""
        s, e = self.buckets[idx]
        if self.is_cuda and self.backend == "nccl":
            if wait is None:
                self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                bad = None
                if wait is not None:
                    from . import kernels as K
                    flag, tag, polls, bad_ptr = wait
                    K.check(K._lib.load().icka_dp_flag_wait(flag, tag & 0xFFFFFFFF, bad_ptr, polls, K._stream()),
                            "icka_dp_flag_wait")
                    bad = (bad_ptr, tag & 0xFFFFFFFF)
                self._allreduce(idx, bad)
        else:
            self._allreduce(idx)

    def _poison_if(self, bad, t: torch.Tensor) -> None:
        """NaN into the first elements of ``t`` when the bucket's wait gave up in this step (icka_dp_poison_if)."""
        from . import kernels as K
        K.check(K._lib.load().icka_dp_poison_if(bad[0], bad[1], t.data_ptr(), int(t.dtype == torch.bfloat16), min(8, t.numel()),
                                                K._stream()), "icka_dp_poison_if")

    def _allreduce(self, idx: int, bad=None) -> None:
        s, e = self.buckets[idx]
        buf = self.arena.gflat[s:e]
        if self.diag:   # diagnostic only: leave out parts of the exchange to price them; wrong gradients (see __init__)
            from . import kernels as K
            if "cast" in self.diag and self.comm_bf16:
                K.dp_cast_chunks(self.arena.gflat, self.gwire, self._tables[idx])
                K.dp_cast_back_scaled(self.gwire[s:e], buf, 1.0 / self.world)
            if "comm" in self.diag:
                dist.all_reduce(self.gwire[s:e] if self.comm_bf16 else buf, op=dist.ReduceOp.SUM, group=self.group)
            return
        if self.backend == "nccl":
            if self.comm_bf16:
                from . import kernels as K
                K.dp_cast_chunks(self.arena.gflat, self.gwire, self._tables[idx])
                w = self.gwire[s:e]
                if bad is not None:              # after the cast (which rewrites un-wired slots), before the sum
                    self._poison_if(bad, self.gwire[self._poison_off[idx]:e])
                dist.all_reduce(w, op=dist.ReduceOp.SUM, group=self.group)
                K.dp_cast_back_scaled(w, buf, 1.0 / self.world)
            else:
                if bad is not None:
                    self._poison_if(bad, buf)
                dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.group)
            if bad is not None:
                self._poison_if(bad, buf)        # (a late store of the graph may still land here: FlaggedStep poisons once
                                                 # more on the compute stream after the join, icka_dp_poison_final)
            return
        # gloo has no AVG; device tensors go through the host (2-process tests on one GPU)
        if self.is_cuda:
            if self.comm_bf16:      # same data flow and rounding points as the RCCL path: wire copies + chunk cast, bf16 sum
                from . import kernels as K
                K.dp_cast_chunks(self.arena.gflat, self.gwire, self._tables[idx])
                host = self.gwire[s:e].float().cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                host = host.to(torch.bfloat16).float().mul_(1.0 / self.world)
            else:
                host = buf.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                host.mul_(1.0 / self.world)
            buf.copy_(host)
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
            buf.mul_(1.0 / self.world)

    def mark_final(self, slots) -> None:
        """Called from backward (ParamArena.flush_final) with one entry per gradient WRITE since the last flush.  During
        the calibration step the writes per slot are only counted (buckets are reduced in ``finish``).  Afterwards a
        slot is final once it has received as many writes as in the calibration step; a bucket whose slots are all
        final is all-reduced right away on the side stream.  Order-independent: nothing is assumed about the order in
        which autograd runs the blocks.  A write into a bucket whose all-reduce is already in flight (more writes than in
        the calibration step: a data-dependent branch, a module applied more often) would be lost or race with the
        in-place reduction: it raises."""
        if self.sparse_word is not None:
            n_all = len(slots)
            slots = [s for s in slots if s is not self.sparse_word]
            if len(slots) != n_all:
                self._word_written = True     # exchange_sparse: rows registered -> row exchange, else dense all-reduce of the slot
        if not self._calibrated:
            for s in slots:
                self._expected[id(s)] = self._expected.get(id(s), 0) + 1
            return
        for s in slots:
            sid = id(s)
            bi = self._bucket_of[sid]
            if self._launched[bi]:
                raise RuntimeError(
                    "icka_amd.dp.GradReducer: gradient write into %s after its bucket (%d) was handed to the all-reduce of "
                    "this step -- the step wrote this parameter more often than the calibration (first) step did.  Build a "
                    "new GradReducer after changing what the step computes." % (s.name, bi))
            exp = self._expected.get(sid)
            if exp is None:
                # a parameter that got no gradient in the calibration step: its bucket can no longer be trusted to be
                # complete early -- reduce it in finish()
                self._waiting[bi] = 1 << 30
                continue
            n = self._seen.get(sid, 0) + 1
            self._seen[sid] = n
            if n == exp:
                self._waiting[bi] -= 1
                if self._waiting[bi] == 0:
                    self._launch(bi)
        if self.capture is not None:
            self.capture.after_mark()

    def finish(self) -> None:
        """Launch every bucket not launched yet (parameters that got no gradient this step keep their bucket
        waiting until here) and make the compute stream wait for the reductions."""
        if self.muted:
            return
        if not self._calibrated and self.gwire is not None:
            self._build_tables()      # the calibration step has shown which gradients GEMM epilogues copy to the wire:
                                      # the casts below (and of every later step) leave those ranges alone
        for bi in range(len(self.buckets)):
            if not self._launched[bi]:
                self._launch(bi)
        if self.capture is None:
            self.exchange_sparse()
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

    def share_bad_words(self, words: torch.Tensor) -> None:
        """f32 wire format only: make a flag wait that gave up on ONE rank known to all -- a MAX all-reduce of the buckets' bad
        words (they hold the step number of the give-up) on the communication stream behind the last bucket, so that the final
        poison pass (icka_dp_poison_final) writes the NaN on every rank and no replica takes an optimizer step on the partial
        sums it received.  (With bf16 buckets the NaN itself travels through the sum: ``_poison_off``.)"""
        if self.is_cuda and self.backend == "nccl" and self.world > 1 and not self.comm_bf16:
            with torch.cuda.stream(self.comm_stream):
                dist.all_reduce(words, op=dist.ReduceOp.MAX, group=self.group)

    def reduce_all(self) -> None:
        """Non-overlapped form: all-reduce every bucket now (after backward)."""
        self.finish()

    # ------------------------------------------------------------------------------------------------- row-sparse word table
    def set_sparse_rows(self, rows: torch.Tensor, ids: torch.Tensor, accumulate: bool = False) -> None:
        """Called by the embedding backward (ops.EmbeddingsFn; tests call it directly): this step's per-token gradient rows of
        the word table (f32 [T, H], zero rows for the padding id) and their ids (int64 [T]).  ``accumulate``: the caller still
        holds an earlier gradient of this cycle in the table's slot (the exchanged rows are added to it)."""
        if self.sparse_word is None:
            raise RuntimeError("GradReducer was built without sparse_embeddings=True (or the table is not at a bucket edge)")
        if rows.dim() != 2 or rows.shape[1] != self.sparse_word.shape[1] or ids.numel() != rows.shape[0]:
            raise ValueError("sparse rows must be [T, %d] with one id per row" % self.sparse_word.shape[1])
        if self._sparse is not None:
            raise RuntimeError("GradReducer.set_sparse_rows: the rows of an earlier embedding backward of this step have not been "
                               "exchanged yet (finish() / exchange_sparse() consumes them); a second set would replace them silently")
        self._sparse = (rows, ids, bool(accumulate))

    def _sparse_buffers(self, T: int, H: int, dtype, device):
        key = (T, H, dtype, str(device))
        ws = self._sparse_ws.get(key)
        if ws is None:
            ws = {"rows_w": torch.empty(T, H, dtype=dtype, device=device),
                  "all_rows": torch.empty(self.world * T, H, dtype=dtype, device=device),
                  "all_ids": torch.empty(self.world * T, dtype=torch.int64, device=device)}
            self._sparse_ws[key] = ws
        return ws

    def _exchange_word_dense(self) -> None:
        """The word table's slot was written DENSELY in this step (an embedding backward that does not know the row path: the
        fp32-exact mode, the prompt embeddings of cross_modal): mean all-reduce of the slot, like any bucket."""
        w = self.sparse_word
        g = self.arena.gflat[w.off:w.off + w.numel]
        self.sparse_stats["dense_slot"] = self.sparse_stats.get("dense_slot", 0) + 1
        if self.is_cuda and self.backend == "nccl":
            on_current = getattr(self, "_exchange_on_current", False)
            if not on_current:
                self.comm_stream.wait_stream(torch.cuda.current_stream())
            with (_NullCtx() if on_current else torch.cuda.stream(self.comm_stream)):
                dist.all_reduce(g, op=dist.ReduceOp.AVG, group=self.group)
            return
        host = g.cpu() if self.is_cuda else g
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
        host.mul_(1.0 / self.world)
        if self.is_cuda:
            g.copy_(host)

    def exchange_sparse(self, word_written: Optional[bool] = None) -> None:
        """All-gather every rank's token rows + ids and add (1 / world) * rows into the word table's gradient slot; dense
        fallback (local scatter + all-reduce of the slot) when a rank brings at least vocab / 4 rows.  On devices with RCCL the
        work goes to the communication stream (after everything already on the current stream); ``join`` orders it before the
        consumers of the gradient.  A step whose embedding backward registered NO rows but wrote the slot densely
        (``word_written``; default: what ``mark_final`` saw in the step in flight) gets a plain mean all-reduce of the slot, so
        no path leaves the replicas with local word gradients.  The row count T and with it the path (rows / dense) and the
        all-gather sizes are agreed across the ranks the first time each T is seen (host-side, over the store): ragged counts
        raise on every rank instead of pairing an all-reduce with an all-gather."""
        if self.sparse_word is None:
            return
        written = self._word_written if word_written is None else bool(word_written)
        self._word_written = False
        if self._sparse is None:
            if written:
                self._exchange_word_dense()
            return
        rows, ids, accumulate = self._sparse
        self._sparse = None
        w = self.sparse_word
        V, H = w.shape
        T = rows.shape[0]
        if T not in self._agreed_T:
            Ts = all_ranks_gather(int(T), self.group, "row count of the sparse word-embedding exchange")
            if len(set(Ts)) != 1:
                raise RuntimeError("icka_amd.dp.GradReducer(sparse_embeddings=True): the ranks bring different numbers of token rows "
                                   "%s -- the row exchange needs the same batch x sequence shape on every rank (pad the last batch "
                                   "as the reference's sampler does, or use the dense exchange)" % Ts)
            self._agreed_T.add(T)
        g = self.arena.gflat[w.off:w.off + w.numel].view(V, H)
        inv = 1.0 / self.world
        dense = T * 4 >= V
        self.sparse_stats["dense_fallback" if dense else "sparse"] += 1
        if not self.is_cuda:                                   # CPU arena over gloo (tests): plain tensor arithmetic
            keep = (ids != 0).to(rows.dtype).unsqueeze(1)
            if dense:
                loc = torch.zeros_like(g).index_add_(0, ids, rows * keep)
                dist.all_reduce(loc, op=dist.ReduceOp.SUM, group=self.group)
                g.copy_(g + loc * inv if accumulate else loc * inv)
                return
            all_rows = [torch.empty_like(rows) for _ in range(self.world)]
            all_ids = [torch.empty_like(ids) for _ in range(self.world)]
            dist.all_gather(all_rows, rows.contiguous(), group=self.group)
            dist.all_gather(all_ids, ids.contiguous(), group=self.group)
            if not accumulate:
                g.zero_()
            for r_, i_ in zip(all_rows, all_ids):
                g.index_add_(0, i_, r_ * (i_ != 0).to(r_.dtype).unsqueeze(1) * inv)
            return
        from . import kernels as K
        nccl = self.backend == "nccl"
        on_current = getattr(self, "_exchange_on_current", False)     # graph.FlaggedStep: already on the communication stream
        if nccl and not on_current:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
        ctx = torch.cuda.stream(self.comm_stream) if (nccl and not on_current) else _NullCtx()
        with ctx:
            if dense:
                if accumulate:
                    tmp = self.arena.workspace("sparse_dense", V * H).view(V, H)
                    K.zero_(tmp.view(-1))
                    K.embed_scatter_rows(rows, ids, tmp, padding_idx=0, scale=inv)
                    self._allreduce_tensor(tmp.view(-1))
                    ar = self._sparse_ws.get("arange")
                    if ar is None:
                        ar = self._sparse_ws["arange"] = torch.arange(V, dtype=torch.int64, device=g.device)
                    K.embed_scatter_rows(tmp, ar, g, padding_idx=-1, scale=1.0)      # g += tmp (row-wise atomics kernel)
                else:
                    K.zero_(g.view(-1))
                    K.embed_scatter_rows(rows, ids, g, padding_idx=0, scale=inv)
                    self._allreduce_tensor(g.view(-1))
                return
            wdt = torch.bfloat16 if self.comm_bf16 else torch.float32
            ws = self._sparse_buffers(T, H, wdt, rows.device)
            if self.comm_bf16:
                K.cast_f32_to_bf16(rows.view(-1), ws["rows_w"].view(-1))
                mine = ws["rows_w"]
            else:
                mine = rows
            ids_c = ids.contiguous()
            if nccl:
                dist.all_gather_into_tensor(ws["all_rows"], mine, group=self.group)
                dist.all_gather_into_tensor(ws["all_ids"], ids_c, group=self.group)
            else:                                              # gloo with device tensors (one-GPU tests): through the host
                hr = [torch.empty(T, H, dtype=torch.float32) for _ in range(self.world)]
                hi = [torch.empty(T, dtype=torch.int64) for _ in range(self.world)]
                dist.all_gather(hr, mine.float().cpu(), group=self.group)
                dist.all_gather(hi, ids_c.cpu(), group=self.group)
                ws["all_rows"].copy_(torch.cat(hr).to(wdt))
                ws["all_ids"].copy_(torch.cat(hi))
            if not accumulate:
                K.zero_(g.view(-1))
            K.embed_scatter_rows(ws["all_rows"], ws["all_ids"], g, padding_idx=0, scale=inv)

    def _allreduce_tensor(self, t: torch.Tensor) -> None:
        if self.backend == "nccl":
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        else:
            host = t.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(host)

    # -------------------------------------------------------------------------------------------------
    def wire_bytes(self) -> int:
        """Bytes one rank puts into the all-reduces of a step (before the algorithm's 2(N-1)/N factor)."""
        b = sum(e - s for s, e in self.buckets) * (2 if self.comm_bf16 else 4)
        if self.sparse_word is not None and self._sparse_ws:
            for k, ws in self._sparse_ws.items():
                if isinstance(k, tuple):
                    b += ws["rows_w"].numel() * ws["rows_w"].element_size() + ws["all_ids"].numel() // self.world * 8
                    break
        return b
