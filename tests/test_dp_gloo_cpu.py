"""CPU, world_size 2 and 8, gloo: the data-parallel gradient path (ParamArena buckets + GradReducer).

Identity under test: N ranks each stepping on their own micro-batch and averaging gradients
== one rank stepping on the concatenated batch (loss = mean over the global valid tokens needs the per-rank token
counts, so each rank uses sum-of-token-losses / global_count; here both ranks have the same valid-token count).
The model math on CPU is the oracle (tests may use it); what is exercised is the product's arena layout, bucket
partition, overlap bookkeeping (mark_final / finish) and the all-reduce itself.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from icka_amd import synth
from icka_amd.arena import ParamArena
from icka_amd.config import BertConfig
from icka_amd.modeling import MTCCMBertForMMTokenClassificationCRF
from oracle import mner_oracle as O

CFG = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
           max_position_embeddings=64)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_grads(model, batch):
    """Oracle forward/backward driven by the product module's own (arena-view) parameters."""
    P = dict(model.named_parameters())
    ocfg = O.OracleConfig(vocab_size=512, **CFG)
    logits = O.mner_logits(P, ocfg, batch["input_ids"], batch["segment_ids"], batch["input_mask"],
                           batch["added_attention_mask"], batch["visual_embeds_att"], 1, 36)
    loss = O.token_ce_loss(logits, batch["labels"], batch["input_mask"])
    grads = torch.autograd.grad(loss, [p for p in P.values()], allow_unused=True)
    return loss, dict(zip(P.keys(), grads))


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2 if world <= 2 else 1)
    from icka_amd.dp import GradReducer
    model = MTCCMBertForMMTokenClassificationCRF(BertConfig(512, **CFG), layer_num1=1, num_labels=13, regions=36)
    synth.fill_module_(model)
    if rank >= 1:   # replicas must not depend on identical init: rank 0's parameters are broadcast
        with torch.no_grad():
            for p in model.parameters():
                p.add_(float(rank))
    arena = ParamArena(model)
    red = GradReducer(arena, bucket_mb=0.25)
    assert len(red.buckets) > 3
    red.broadcast_parameters(0)
    per = 2 if world <= 2 else 1           # world 8: the c3 world size (BASELINE configs[2]), one pair per rank
    full = synth.synthetic_batch(per * world, 32, 36, vocab_size=512, seed=5, ragged=False)
    mine = {k: v[rank * per:(rank + 1) * per] for k, v in full.items()}
    for step in range(2):   # second step exercises the calibrated (overlapped) bucket bookkeeping
        _, g = _oracle_grads(model, mine)
        arena.reducer = red
        # emulate backward order: blocks finish from the END of the arena; write grads then flush per "block"
        for s in reversed(arena.order):
            gi = g[s.name]
            if gi is None:
                continue
            arena.grad_beta(s.param)
            arena.g(s.param).copy_(gi)
            arena.flush_final()
        red.finish()
    _, gfull = _oracle_grads(model, full)
    worst = 0.0
    gmax = max(g.abs().max().item() for g in gfull.values() if g is not None)
    for s in arena.order:
        if gfull[s.name] is None:
            continue
        ref = gfull[s.name]
        # key.bias has an exactly-zero true gradient (softmax shift invariance): floor the denominator
        err = (arena.g(s.param) - ref).abs().max().item() / (ref.abs().max().item() + 1e-6 * gmax)
        worst = max(worst, err)
        assert s.param.grad is not None and s.param.grad.data_ptr() == arena.g(s.param).data_ptr()
    torch.save({"worst": worst, "nb": len(red.buckets), "calibrated": red._calibrated}, os.path.join(tmp, "r%d.pt" % rank))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_n_rank_dp_equals_single_rank_on_concatenated_batch(tmp_path, world):
    """SURVEY.md section 8e's identity at world 2 and at the world size BASELINE's c3 names (8 ranks x micro-batch == 1 rank x
    concatenated batch, <= 1e-5 relative, dropout off, f32 buckets).  Reference: one process per GPU (My_cross_attention.py:
    653-657), DistributedSampler shards (:707), apex DDP averages the gradients (:768-776)."""
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        res = torch.load(os.path.join(str(tmp_path), "r%d.pt" % r))
        assert res["worst"] < 1e-5, res
        assert res["calibrated"]


def _twice_worker(rank, world, port, tmp):
    """A slot that receives TWO gradient writes per step (a module applied twice per forward): its bucket must not be
    reduced after the first write (the late write would race with the reduction and never be averaged)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from icka_amd.dp import GradReducer
    model = torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.Linear(64, 64), torch.nn.Linear(64, 64))
    arena = ParamArena(model)
    red = GradReducer(arena, bucket_mb=64 * 65 * 4 / (1 << 20))     # one Linear per bucket
    assert len(red.buckets) == 3
    arena.reducer = red
    launched_early = []
    orig = red._launch
    red._launch = lambda idx: (launched_early.append((idx, dict(red._seen))), orig(idx))[1]
    shared = model[2]                                   # written twice per step, the other two once
    for step in range(3):
        for p in model.parameters():
            p.grad = None
        for layer, times in ((shared, 2), (model[1], 1), (model[0], 1)):
            for t in range(times):
                for p in (layer.weight, layer.bias):
                    beta = arena.grad_beta(p)
                    arena.g(p).mul_(beta).add_(float(rank + 1))
                arena.flush_final()                     # end of one "block backward"
        red.finish()
        assert torch.allclose(arena.g(shared.weight), torch.full_like(shared.weight, 2 * 1.5)), step
        assert torch.allclose(arena.g(model[0].weight), torch.full_like(model[0].weight, 1.5)), step
    # after calibration the shared bucket (index 0 = end of the arena) was launched only once BOTH writes were in
    sid = id(arena.slots[id(shared.weight)])
    early = [seen for idx, seen in launched_early if idx == 0 and seen]
    assert early and all(seen.get(sid) == 2 for seen in early), early
    torch.save({"ok": True}, os.path.join(tmp, "t%d.pt" % rank))
    dist.destroy_process_group()


def test_slot_written_twice_is_reduced_after_its_last_write(tmp_path):
    port = _free_port()
    mp.spawn(_twice_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert torch.load(os.path.join(str(tmp_path), "t%d.pt" % r))["ok"]


def test_bucket_partition_covers_arena_in_reverse_order():
    model = MTCCMBertForMMTokenClassificationCRF(BertConfig(512, **CFG), layer_num1=1, num_labels=13)
    arena = ParamArena(model)
    b = arena.buckets(50_000)
    assert b[0][1] == arena.total and b[-1][0] == 0
    for (s0, e0), (s1, e1) in zip(b[:-1], b[1:]):
        assert e1 == s0 and e0 > s0
    assert all(e - s >= 50_000 for s, e in b[:-2])
    # the embedding tables (gradient final only at the very end of backward) are a bucket of their own
    tables = [s for s in arena.order if s.is_table]
    assert [s.name.split(".")[-2] for s in tables] == ["word_embeddings", "position_embeddings", "token_type_embeddings"]
    assert b[-1] == (0, tables[-1].off + (tables[-1].numel + 7) // 8 * 8)
    assert arena._cast_ranges == [(b[-1][1], arena.total)]      # and they are the only part without a bf16 shadow
    # fused QKV operands are physically adjacent
    sa = model.bert.encoder.layer[0].attention.self
    first, rows = arena._adjacent((sa.query.weight, sa.key.weight, sa.value.weight))
    assert rows == 3 * 128 and first.name.endswith("query.weight")


def test_arena_keeps_state_dict_contract():
    model = MTCCMBertForMMTokenClassificationCRF(BertConfig(512, **CFG), layer_num1=1, num_labels=13)
    synth.fill_module_(model)
    before = {k: v.clone() for k, v in model.state_dict().items()}
    arena = ParamArena(model)
    after = model.state_dict()
    assert list(before) == list(after)
    for k in before:
        assert torch.equal(before[k], after[k]) and after[k].dtype == torch.float32
    # parameters are views into one flat buffer; loading a checkpoint writes through
    sd = {k: v + 1 for k, v in before.items()}
    model.load_state_dict(sd)
    assert torch.equal(arena.flat[arena.order[0].off:arena.order[0].off + 4],
                       sd[arena.order[0].name].reshape(-1)[:4])
    assert arena.valid_for(model)
    with pytest.raises(RuntimeError):
        arena.sync()   # bf16 shadows need a device: no CPU path


def _late_write_worker(rank, world, port, tmp):
    """after calibration a slot that is written MORE often than in the calibration step would be written while
    its bucket's all-reduce is already in flight: mark_final must raise instead of losing / racing the late write."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from icka_amd.dp import GradReducer
    model = torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.Linear(64, 64))
    arena = ParamArena(model)
    red = GradReducer(arena, bucket_mb=64 * 65 * 4 / (1 << 20))     # one Linear per bucket
    assert len(red.buckets) == 2 and not red.comm_bf16                # library default on the wire: f32, like apex DDP
    arena.reducer = red

    def write(layer):
        for p in (layer.weight, layer.bias):
            arena.grad_beta(p)
            arena.g(p).fill_(1.0)
        arena.flush_final()
    for step in range(2):                       # calibration: every layer written once per step
        for p in model.parameters():
            p.grad = None
        write(model[1]); write(model[0])
        red.finish()
    for p in model.parameters():
        p.grad = None
    write(model[1])                             # its bucket is launched right here (all its slots are final)
    raised = False
    try:
        write(model[1])                         # a second write in the same step: the bucket is already in flight
    except RuntimeError as e:
        raised = "after its bucket" in str(e)
    torch.save({"raised": raised}, os.path.join(tmp, "l%d.pt" % rank))
    dist.destroy_process_group()


def test_a_gradient_write_into_a_bucket_already_in_flight_raises(tmp_path):
    mp.spawn(_late_write_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    assert torch.load(os.path.join(str(tmp_path), "l0.pt"))["raised"]


def test_bf16_ring_sum_over_8_ranks_is_bounded_against_the_fp32_mean():
    """comm_dtype="bf16" sums bf16 buckets over the ring -- every hop adds two values in f32 and rounds the
    partial sum to bf16 (N - 1 roundings per element), where the reference's apex DDP averages in fp32.  Emulate the ring
    reduce-scatter order for N = 8 on gradient-like data and bound the error of sum / N against the fp32 mean: relative L2
    within 2^-8 (a bf16 rounding is at most 2^-8 relative -- half an ulp of an 8-bit significand -- per hop; the errors of
    the hops add as a random walk), and never worse than (N - 1) * 2^-8 of the largest partial sum element-wise."""
    N, n = 8, 1 << 16
    g = torch.Generator().manual_seed(11)
    grads = [torch.randn(n, generator=g) * 1e-3 * (1.0 + 0.1 * r) for r in range(N)]     # one gradient slice per rank
    wire = [x.to(torch.bfloat16) for x in grads]                                          # what each rank puts on the wire
    ref = torch.stack([w.float() for w in wire]).mean(0)                                  # fp32 mean of the SAME wire inputs
    # ring reduce-scatter: chunk c starts at rank c + 1 and travels N - 1 hops, rounding to bf16 after every add
    chunks = torch.arange(n).chunk(N)
    out = torch.empty(n)
    worst_partial = torch.zeros(n)
    for c, idx in enumerate(chunks):
        acc = wire[(c + 1) % N][idx]
        for hop in range(2, N + 1):
            acc = (acc.float() + wire[(c + hop) % N][idx].float()).to(torch.bfloat16)
            worst_partial[idx] = torch.maximum(worst_partial[idx], acc.float().abs())
        out[idx] = acc.float() / N                                                        # icka_dp_cast_back_scaled
    rel = ((out - ref).norm() / ref.norm()).item()
    assert rel < 2.0 ** -8, rel
    assert ((out - ref).abs() <= (N - 1) * 2.0 ** -8 * worst_partial / N + 1e-12).all()
    # and against the true fp32 gradients (wire rounding of the inputs included) it stays within 2^-7
    rel_true = ((out - torch.stack(grads).mean(0)).norm() / torch.stack(grads).mean(0).norm()).item()
    assert rel_true < 2.0 ** -7, rel_true
    print("\n[bf16 ring sum, N = 8] rel-L2 vs fp32 mean of the wire inputs %.2e, vs the fp32 gradients %.2e" % (rel, rel_true))


def _sparse_worker(rank, world, port, tmp):
    """Row-sparse exchange of the word-embedding gradient (GradReducer(sparse_embeddings=True)) against the dense all-reduce of
    the same gradient: every rank holds its own token rows + ids; after the exchange the table's gradient slot must equal
    (1 / world) x the scatter of ALL ranks' rows -- which is what the dense path averages."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from icka_amd.dp import GradReducer

    class Tiny(torch.nn.Module):
        def __init__(self, vocab):
            super().__init__()
            self.word = torch.nn.Embedding(vocab, 16, padding_idx=0)
            self.pos = torch.nn.Embedding(8, 16)
            self.lin = torch.nn.Linear(16, 16)

    out = {}
    for vocab, T, expect in ((96, 12, "sparse"), (32, 12, "dense_fallback")):      # 12 rows < 96 / 4; 12 rows >= 32 / 4
        torch.manual_seed(0)
        model = Tiny(vocab)
        arena = ParamArena(model)
        red = GradReducer(arena, bucket_mb=1e-4, sparse_embeddings=True)
        assert red.sparse_word is arena.slot(model.word.weight)
        assert all(not (lo <= red.sparse_word.off < hi) for lo, hi in red.buckets)       # the table left the dense buckets
        arena.reducer = red
        g = torch.Generator().manual_seed(100 + rank)
        ids = torch.randint(0, vocab, (T,), generator=g)
        ids[0] = 0                                                                        # a padding token: contributes nothing
        rows = torch.randn(T, 16, generator=g)
        rows[ids == 0] = 0.0
        # reference: dense mean over ranks of the local scatter
        loc = torch.zeros(vocab, 16).index_add_(0, ids, rows)
        ref = loc.clone()
        dist.all_reduce(ref)
        ref /= world
        for step in range(2):                                                             # calibration step, then a normal one
            for s in reversed(arena.order):
                if s is red.sparse_word:
                    arena.grad_beta(s.param)
                    red.set_sparse_rows(rows, ids)
                else:
                    arena.grad_beta(s.param)
                    arena.g(s.param).fill_(float(rank + 1))
                arena.flush_final()
            red.finish()
        got = arena.g(model.word.weight)
        err = (got - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)
        other = arena.g(model.lin.weight)
        out[expect] = {"err": err, "stats": dict(red.sparse_stats), "row0": got[0].abs().max().item(),
                       "dense_ok": abs(other.mean().item() - (world + 1) / 2.0) < 1e-6}
        # accumulation: a second exchange onto the gradient the caller still holds adds the same mean once more
        arena.grad_beta(model.word.weight)
        red.set_sparse_rows(rows, ids, accumulate=True)
        red.exchange_sparse()
        out[expect]["acc_err"] = (arena.g(model.word.weight) - 2 * ref).abs().max().item() / (ref.abs().max().item() + 1e-12)
        arena.reducer = None
    torch.save(out, os.path.join(tmp, "s%d.pt" % rank))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_row_sparse_word_embedding_exchange_equals_the_dense_mean(tmp_path, world):
    """all-gather of the ranks' token rows + ids and a local scatter-add == the dense all-reduce mean (f32,
    <= 1e-6), at world 2 and 8; the dense fallback when a rank brings at least vocab / 4 rows; the padding row stays zero; the
    other gradients keep going through the dense buckets."""
    port = _free_port()
    mp.spawn(_sparse_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        res = torch.load(os.path.join(str(tmp_path), "s%d.pt" % r))
        for kind in ("sparse", "dense_fallback"):
            assert res[kind]["err"] < 1e-6 and res[kind]["acc_err"] < 1e-6, res
            assert res[kind]["row0"] == 0.0 and res[kind]["dense_ok"], res
        assert res["sparse"]["stats"]["sparse"] >= 2 and res["sparse"]["stats"]["dense_fallback"] == 0, res
        assert res["dense_fallback"]["stats"]["dense_fallback"] >= 2, res


def _agree_worker(rank, world, port, tmp):
    """dp.all_ranks_gather / all_ranks_agree: the vote that decides which form of a step ALL ranks build.
    It runs over the c10d store, so it pairs correctly even while the ranks' collective streams are at different points: here
    rank 0 has an all-reduce IN FLIGHT (async, not yet matched by rank 1) when both vote."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from icka_amd.dp import all_ranks_agree, all_ranks_gather
    out = {}
    out["gather"] = all_ranks_gather(10 + rank, what="test")
    pending = dist.all_reduce(torch.ones(4), async_op=True) if rank == 0 else None      # unmatched so far on rank 1
    out["one_fails"] = all_ranks_agree(rank != 1, what="capture")                        # rank 1 "could not capture"
    out["all_ok"] = all_ranks_agree(True, what="capture")
    if rank != 0:
        pending = dist.all_reduce(torch.ones(4), async_op=True)                           # now the collective pairs up
    pending.wait()
    out["strings"] = all_ranks_gather("r%d" % rank)
    torch.save(out, os.path.join(tmp, "a%d.pt" % rank))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_ranks_vote_over_the_store_not_over_a_collective(tmp_path, world):
    mp.spawn(_agree_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        res = torch.load(os.path.join(str(tmp_path), "a%d.pt" % r))
        assert res["gather"] == [10 + i for i in range(world)], res
        assert res["one_fails"] is False and res["all_ok"] is True, res       # every rank sees the same verdict
        assert res["strings"] == ["r%d" % i for i in range(world)], res


def _sparse_dense_write_worker(rank, world, port, tmp):
    """GradReducer(sparse_embeddings=True) with an embedding backward that does NOT know the row path (the
    fp32-exact mode, the prompt embeddings): the word table's slot is written densely and no rows are registered -- the slot must
    still be averaged (dense all-reduce of the slot), never left local.  Also: ragged row counts raise on every rank, and a second
    set of rows before the first was exchanged raises."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from icka_amd.dp import GradReducer

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.word = torch.nn.Embedding(96, 16, padding_idx=0)
            self.lin = torch.nn.Linear(16, 16)

    model = Tiny()
    arena = ParamArena(model)
    red = GradReducer(arena, bucket_mb=1e-4, sparse_embeddings=True)
    assert red.sparse_word is arena.slot(model.word.weight)
    arena.reducer = red
    out = {}
    for step in range(2):
        for p in model.parameters():
            p.grad = None
        for s in reversed(arena.order):
            arena.grad_beta(s.param)
            arena.g(s.param).fill_(float(rank + 1))            # dense write, also into the word table; no set_sparse_rows
            arena.flush_final()
        red.finish()
        out["word_mean_%d" % step] = arena.g(model.word.weight).mean().item()
        out["lin_mean_%d" % step] = arena.g(model.lin.weight).mean().item()
    out["stats"] = dict(red.sparse_stats)
    # ragged row counts: rank r brings 4 + r rows
    rows = torch.zeros(4 + rank, 16)
    ids = torch.ones(4 + rank, dtype=torch.int64)
    red.set_sparse_rows(rows, ids)
    try:
        red.set_sparse_rows(rows, ids)
        out["pending_raises"] = False
    except RuntimeError as e:
        out["pending_raises"] = "have not been exchanged" in str(e)
    try:
        red.exchange_sparse()
        out["ragged_raises"] = False
    except RuntimeError as e:
        out["ragged_raises"] = "different numbers of token rows" in str(e)
    torch.save(out, os.path.join(tmp, "w%d.pt" % rank))
    dist.destroy_process_group()


def test_sparse_reducer_averages_a_densely_written_word_table_and_refuses_ragged_rows(tmp_path):
    world = 2
    mp.spawn(_sparse_dense_write_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        res = torch.load(os.path.join(str(tmp_path), "w%d.pt" % r))
        for step in range(2):
            assert abs(res["word_mean_%d" % step] - 1.5) < 1e-6 and abs(res["lin_mean_%d" % step] - 1.5) < 1e-6, res
        assert res["stats"].get("dense_slot", 0) >= 2, res
        assert res["pending_raises"] and res["ragged_raises"], res


def _late_rank_worker(rank, world, port, tmp):
    """A rank that never reaches an agreement point must not hang the others: all_ranks_gather raises on them after its timeout."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from icka_amd.dp import all_ranks_gather
    out = {"raised": None}
    if rank == 0:
        try:
            all_ranks_gather(True, what="a vote rank 1 skips", timeout=3.0)
            out["raised"] = False
        except RuntimeError as e:
            out["raised"] = "not every rank reached the agreement point" in str(e)
    torch.save(out, os.path.join(tmp, "q%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_a_rank_that_never_votes_times_the_others_out_instead_of_hanging_them(tmp_path):
    mp.spawn(_late_rank_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert torch.load(os.path.join(str(tmp_path), "q0.pt"))["raised"] is True
