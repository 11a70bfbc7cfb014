"""Worker of tests/test_dp_gpu.py::test_segmented_step (child process): world-1 RCCL process group, GradReducer with
small buckets, the step captured as linear hipGraph segments with eager all-reduces between them (graph.SegmentedStep),
compared with the plain eager step on the same batch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist


def main():
    port, out, comm = sys.argv[1], sys.argv[2], sys.argv[3]
    kind = sys.argv[4] if len(sys.argv) > 4 else "segmented"
    sparse = len(sys.argv) > 5 and sys.argv[5] == "sparse"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK="0", WORLD_SIZE="1")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    from icka_amd import synth
    from icka_amd.config import BertConfig
    from icka_amd.dp import GradReducer
    from icka_amd.graph import FlaggedStep, SegmentedStep
    from icka_amd.modeling import MTCCMBertForMMTokenClassificationCRF
    vocab = 2048 if sparse else 512      # (row-sparse exchange: fewer rows per rank than vocab / 4, else it falls back to dense)
    cfg = BertConfig(vocab, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                     max_position_embeddings=64)
    model = MTCCMBertForMMTokenClassificationCRF(cfg, layer_num1=1, num_labels=13, regions=36)
    synth.fill_module_(model)
    model = model.cuda().eval()
    b = {k: v.cuda() for k, v in synth.synthetic_batch(4, 32, 36, vocab_size=vocab, seed=5).items()}

    def fwd_bwd():
        loss = model(b["input_ids"], b["segment_ids"], b["input_mask"], b["added_attention_mask"],
                     b["visual_embeds_mean"], b["visual_embeds_att"], labels=b["labels"])
        loss.backward()
        return loss

    model.zero_grad()
    ref_loss = fwd_bwd().item()
    torch.cuda.synchronize()
    ref = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    arena = model._icka_arena
    red = GradReducer(arena, bucket_mb=0.25, comm_dtype=comm, sparse_embeddings=sparse)
    arena.reducer = red

    def step():
        loss = fwd_bwd()
        red.finish()
        return loss

    ss = SegmentedStep(model, step, red) if kind == "segmented" else FlaggedStep(model, step, red)
    worst = 0.0
    for it in range(3):
        loss = ss()
        torch.cuda.synchronize()
        for n, p in model.named_parameters():
            if n in ref:
                assert p.grad is not None, n
                worst = max(worst, ((p.grad - ref[n]).norm() / (ref[n].norm() + 1e-12)).item())
        model.zero_grad()                      # the reference loop drops the gradients after every step
    from icka_amd import _lib
    assert _lib.load().icka_dp_error() == 0
    if kind == "segmented":
        res = {"segments": len(ss.segments), "after": [len(a) for _, a in ss.segments]}
    else:
        res = {"segments": 1, "after": [len(ss.order)], "order": list(ss.order),
               "step_word": int(ss.sync[0].item()), "flags": ss.sync[ss.FLAG0:ss.FLAG0 + len(red.buckets)].tolist()}
    res.update({"worst": worst, "buckets": len(red.buckets), "loss": loss.item(), "ref_loss": ref_loss,
                "cast_elements": red.cast_elements() if red.gwire is not None else None,
                "total_elements": sum(e - s for s, e in red.buckets), "wire_ranges": len(red._wire_ranges),
                "sparse_stats": dict(red.sparse_stats)})
    torch.save(res, out)
    ss.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
