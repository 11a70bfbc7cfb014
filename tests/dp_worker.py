"""Worker of tests/test_dp_gpu.py (started as a child process, one per rank): the REAL HIP forward + backward on this
rank's half of a batch, gradients averaged by GradReducer over gloo (two ranks share the one GPU of the test box, which
RCCL refuses, so device buckets are staged through the host), compared with the same process's single-rank step on the
concatenated batch (SURVEY.md section 8e correctness test)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist


def main():
    rank, world, port, out, precision = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    comm = sys.argv[6] if len(sys.argv) > 6 else "f32"
    sparse = len(sys.argv) > 7 and sys.argv[7] == "sparse"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import icka_amd
    from icka_amd import synth
    from icka_amd.config import BertConfig
    from icka_amd.dp import GradReducer
    from icka_amd.modeling import MTCCMBertForMMTokenClassificationCRF
    vocab = 2048 if sparse else 512      # (row-sparse exchange: fewer rows per rank than vocab / 4, else it falls back to dense)
    cfg = BertConfig(vocab, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                     max_position_embeddings=64)
    model = MTCCMBertForMMTokenClassificationCRF(cfg, layer_num1=1, num_labels=13, regions=36)
    synth.fill_module_(model)
    if rank == 1:       # replicas start different: rank 0's parameters are broadcast
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.5)
    model = icka_amd.set_precision(model.cuda().eval(), precision)
    full = {k: v.cuda() for k, v in synth.synthetic_batch(8, 32, 36, vocab_size=vocab, seed=5, ragged=False).items()}
    half = {k: v[rank * 4:(rank + 1) * 4].contiguous() for k, v in full.items()}

    def step(b):
        model.zero_grad()
        loss = model(b["input_ids"], b["segment_ids"], b["input_mask"], b["added_attention_mask"],
                     b["visual_embeds_mean"], b["visual_embeds_att"], labels=b["labels"])
        loss.backward()
        return loss

    step(half)                                          # builds the arena
    arena = model._icka_arena
    red = GradReducer(arena, bucket_mb=0.25, comm_dtype=comm, sparse_embeddings=sparse)
    red.broadcast_parameters(0)
    arena.reducer = red
    early = []
    orig = red._launch
    red._launch = lambda idx: (early.append((idx, red._calibrated)), orig(idx))[1]
    for _ in range(2):                                  # calibration step, then an overlapped step
        step(half)
        red.finish()
    torch.cuda.synchronize()
    dp = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    arena.reducer = None
    step(full)
    torch.cuda.synchronize()
    worst, wkey = 0.0, ""
    gmax = max(p.grad.norm().item() for p in model.parameters() if p.grad is not None)
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        rel = ((dp[n] - p.grad).norm() / (p.grad.norm() + 1e-4 * gmax)).item()
        if rel > worst:
            worst, wkey = rel, n
    overlapped = sum(1 for idx, cal in early if cal)     # buckets launched from inside backward after calibration
    total = sum(e - s for s, e in red.buckets)
    torch.save({"worst": worst, "key": wkey, "buckets": len(red.buckets), "overlapped": overlapped,
                "cast_elements": red.cast_elements() if red.gwire is not None else None, "total_elements": total,
                "wire_ranges": len(red._wire_ranges), "sparse_stats": dict(red.sparse_stats),
                "sparse_word": None if red.sparse_word is None else red.sparse_word.name}, out)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
