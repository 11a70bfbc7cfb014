"""Worker of tests/test_dp_gpu.py::test_two_gpus_rccl_flagged_step (one process per GPU, RCCL over xGMI; runs only on a node
with >= 2 GPUs): the bench's default N > 1 path -- graph.FlaggedStep, bf16 wire buffer written by the weight-gradient GEMM
epilogues (c3_only), flag waits + eager all-reduces on the communication stream -- on this rank's half of a batch, against the
same process's single-rank eager step on the concatenated batch (My_cross_attention.py:653-657, :768-776)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

NAMES = ("input_ids", "segment_ids", "input_mask", "added_attention_mask", "visual_embeds_mean", "visual_embeds_att", "labels")


def main():
    rank, world, port, out, comm = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", rank)
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
    if rank > 0:
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.5)
    model = model.to(dev).eval()
    per = 4
    full = synth.synthetic_batch(per * world, 32, 36, vocab_size=512, seed=5, ragged=False)
    fullt = tuple(full[k].to(dev) for k in NAMES)
    mine = tuple(t[rank * per:(rank + 1) * per].contiguous() for t in fullt)

    def fwd_bwd(ids, seg, mask, added, vmean, vatt, labels):
        loss = model(ids, seg, mask, added, vmean, vatt, labels=labels)
        loss.backward()
        return loss

    model.zero_grad()
    fwd_bwd(*mine)                                     # builds the arena
    arena = model._icka_arena
    red = GradReducer(arena, bucket_mb=0.25, comm_dtype=comm)
    red.broadcast_parameters(0)
    arena.reducer = red

    def step(*b):
        loss = fwd_bwd(*b)
        red.finish()
        return loss

    fs = FlaggedStep(model, step, red, inputs=mine)
    for _ in range(3):
        model.zero_grad()
        fs(*mine)
    torch.cuda.synchronize()
    dp = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    err_word = int(K._lib.load().icka_dp_error())
    arena.reducer = None
    model.zero_grad()
    fwd_bwd(*fullt)
    torch.cuda.synchronize()
    worst, wkey = 0.0, ""
    gmax = max(p.grad.norm().item() for p in model.parameters() if p.grad is not None)
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        rel = ((dp[n] - p.grad).norm() / (p.grad.norm() + 1e-4 * gmax)).item()
        if rel > worst:
            worst, wkey = rel, n
    torch.save({"worst": worst, "key": wkey, "buckets": len(red.buckets), "error_word": err_word,
                "wire_only": len(red._wire_only)}, out)
    fs.close()
    red.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
