"""Worker of tests/test_dp_gpu.py (child process, world-1 RCCL process group): graph.GraphedModule(reducer=, accumulate=k) -- the
reference's two lines ``loss = model(...)`` / ``loss.backward()`` with the gradient exchange behind ``backward()``, as under the
reference's apex DDP wrapper (My_cross_attention.py:768-776).  ``acc`` different micro-batches per cycle, two cycles, then a cycle
that is restarted by ``zero_grad`` after its first backward; gradients against the eager sum of the micro-batch gradients without
a reducer (world 1: the exchange is an identity up to the bf16 wire rounding); the step word counts exchanging backwards only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

NAMES = ("input_ids", "segment_ids", "input_mask", "added_attention_mask", "visual_embeds_mean", "visual_embeds_att", "labels")


def main():
    port, out, acc, comm = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK="0", WORLD_SIZE="1")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    from icka_amd import kernels as K
    from icka_amd import synth
    from icka_amd.config import BertConfig
    from icka_amd.dp import GradReducer
    from icka_amd.graph import GraphedModule
    from icka_amd.modeling import MTCCMBertForMMTokenClassificationCRF
    cfg = BertConfig(512, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                     max_position_embeddings=64)
    model = MTCCMBertForMMTokenClassificationCRF(cfg, layer_num1=1, num_labels=13, regions=36)
    synth.fill_module_(model)
    model = model.cuda().eval()
    batches = []
    for i in range(3):
        b = synth.synthetic_batch(4, 32, 36, vocab_size=512, seed=50 + i)
        batches.append(tuple(b[k].cuda() for k in NAMES))

    def body(call, b):                       # the reference's loop body (:814-827)
        loss = call(*b[:6], labels=b[6])
        loss = loss / acc
        loss.backward()
        return loss

    model.zero_grad()
    for b in batches[:acc]:
        body(model, b)
    torch.cuda.synchronize()
    ref = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    arena = model._icka_arena
    red = GradReducer(arena, bucket_mb=0.25, comm_dtype=comm)
    model.zero_grad()
    gm = GraphedModule(model, batches[0][:6], {"labels": batches[0][6]}, reducer=red, accumulate=acc, max_captures=2)
    worst, words, losses = 0.0, [], []

    def check():
        nonlocal worst
        torch.cuda.synchronize()
        for n, p in gm.named_parameters():
            if n in ref:
                assert p.grad is not None, n
                worst = max(worst, ((p.grad - ref[n]).norm() / (ref[n].norm() + 1e-12)).item())

    for cycle in range(2):
        gm.zero_grad()
        for b in batches[:acc]:
            losses.append(body(gm, b).item())
            words.append(int(gm._xch.sync[0].item()))
        check()
    # a cycle cut short: one backward, then zero_grad -> the next `acc` backwards form a whole cycle again
    gm.zero_grad()
    body(gm, batches[1])
    words.append(int(gm._xch.sync[0].item()))
    gm.zero_grad()
    for b in batches[:acc]:
        body(gm, b)
        words.append(int(gm._xch.sync[0].item()))
    check()
    with torch.no_grad():                    # forward only: nothing is exchanged
        gm(*batches[0][:6], labels=batches[0][6])
    words.append(int(gm._xch.sync[0].item()))
    # the short last batch of an epoch (the reference's loader has no drop_last, My_cross_attention.py:708) under data parallelism:
    # captured the first time it is seen (second and last cache entry); the next new shape is past max_captures and runs
    # EAGERLY with the exchange behind backward (autograd end-of-backward callback) -- both give the eager step's gradients
    other = {}
    for name, nb in (("captured", 2), ("eager", 1)):
        b = tuple(t[:nb].contiguous() for t in batches[0])
        arena.reducer = None
        model.zero_grad()
        for _ in range(acc):
            body(model, b)
        torch.cuda.synchronize()
        ref2 = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        arena.reducer = red
        w2 = 0.0
        for cycle in range(2):
            gm.zero_grad()
            for _ in range(acc):
                body(gm, b)
            torch.cuda.synchronize()
            for n, p in gm.named_parameters():
                if n in ref2:
                    w2 = max(w2, ((p.grad - ref2[n]).norm() / (ref2[n].norm() + 1e-12)).item())
        other[name] = w2
    res = {"worst": worst, "step_words": words, "captures": sorted(gm._bwd), "buckets": len(red.buckets), "losses": losses,
           "error_word": int(K._lib.load().icka_dp_error()), "other_shapes": other, "stats": dict(gm.stats),
           "n_captures": gm.captures}
    torch.save(res, out)
    gm.close()
    red.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
