"""Worker of tests/test_dp_gpu.py (child process, world-1 RCCL process group): graph.FlaggedStep beyond the plain step.

  accumulate <comm>  -- ``FlaggedStep(..., inputs=, accumulate=3)`` fed three DIFFERENT micro-batches per cycle, two cycles,
                        against the eager sum of the three micro-batch gradients without a reducer (world 1: the exchange is
                        an identity up to the bf16 wire rounding); also checks that only the 3rd call of a cycle exchanges.
  poison <comm>      -- the give-up path: one bucket's wait polls a word nobody sets, with a tiny poll budget:
                        the bucket's first gradients must be NaN after the step (whatever the chunk cast or a late GEMM
                        epilogue wrote there), the global gradient norm NaN, and the next replay raises DpFlagError.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

NAMES = ("input_ids", "segment_ids", "input_mask", "added_attention_mask", "visual_embeds_mean", "visual_embeds_att", "labels")


def main():
    port, out, what, comm = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK="0", WORLD_SIZE="1")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    from icka_amd import kernels as K
    from icka_amd import synth
    from icka_amd.config import BertConfig
    from icka_amd.dp import GradReducer
    from icka_amd.graph import FlaggedStep
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
    acc = 3 if what == "accumulate" else 1

    def fwd_bwd(ids, seg, mask, added, vmean, vatt, labels):
        loss = model(ids, seg, mask, added, vmean, vatt, labels=labels) / acc
        loss.backward()
        return loss

    model.zero_grad()
    for b in batches[:acc]:
        fwd_bwd(*b)
    torch.cuda.synchronize()
    ref = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    arena = model._icka_arena
    red = GradReducer(arena, bucket_mb=0.25, comm_dtype=comm)
    arena.reducer = red

    def step(*b):
        loss = fwd_bwd(*b)
        red.finish()
        return loss

    model.zero_grad()
    fs = FlaggedStep(model, step, red, inputs=batches[0], accumulate=acc)
    res = {"buckets": len(red.buckets)}
    if what == "accumulate":
        worst, words = 0.0, []
        for cycle in range(2):
            model.zero_grad()
            for b in batches:
                fs(*b)
                words.append(int(fs.sync[0].item()))        # the step word counts EXCHANGING replays only
            torch.cuda.synchronize()
            for n, p in model.named_parameters():
                if n in ref:
                    assert p.grad is not None, n
                    worst = max(worst, ((p.grad - ref[n]).norm() / (ref[n].norm() + 1e-12)).item())
        res.update({"worst": worst, "step_words": words, "graphs": sorted(fs._graphs)})
        assert K._lib.load().icka_dp_error() == 0
    else:
        model.zero_grad()
        fs(*batches[0])
        torch.cuda.synchronize()
        res["clean_finite"] = bool(all(torch.isfinite(p.grad).all().item() for p in model.parameters() if p.grad is not None))
        model.zero_grad()
        bad = fs._order[(False, True)][-1]                  # the bucket that becomes final LAST (the embedding tables)
        first = fs._order[(False, True)][0]
        fs._test_late = (bad, first)
        fs(*batches[0])
        torch.cuda.synchronize()
        g = arena.gflat
        res["poisoned"] = [bool(torch.isnan(g[red.buckets[i][0]:red.buckets[i][0] + 8]).all().item()) for i in (bad, first)]
        others = [i for i in range(len(red.buckets)) if i not in (bad, first)]
        res["others_finite"] = bool(all(torch.isfinite(g[red.buckets[i][0]:red.buckets[i][1]]).all().item() for i in others))
        res["norm_is_nan"] = bool(torch.isnan(torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)).item())
        res["error_word"] = int(K._lib.load().icka_dp_error())
        fs._test_late = ()
        try:
            fs(*batches[0])
            res["raised"] = False
        except K.DpFlagError:
            res["raised"] = True
        model.zero_grad()
        fs(*batches[0])                                     # the error word was cleared by the raise: the step works again
        torch.cuda.synchronize()
        res["recovered_finite"] = bool(all(torch.isfinite(p.grad).all().item() for p in model.parameters() if p.grad is not None))
    torch.save(res, out)
    fs.close()
    red.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
