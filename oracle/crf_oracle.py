"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the linear-chain CRF the reference calls through the third-party
package ``torchcrf`` (pytorch-crf; imported at Cross_Modal_Interaction_Module.py:3 and my_bert/cl_modeling.py:30, used
at :911 / :1045-1057 and :1269 / :1380-1386).  The package is NOT vendored under /root/reference, is not pinned by the
reference (no requirements file) and is not installed in this image, so this file restates its published algorithm
(pytorch-crf 0.7.2, ``CRF.forward``, ``_compute_score``, ``_compute_normalizer``, ``_viterbi_decode``) in plain PyTorch
and the tests pin it against brute-force enumeration of all tag paths on small cases.  **Parity with the package
itself is unpinned** (no reference test or fixture covers it).
Only tests/ may import this module."""
from __future__ import annotations

import itertools
from typing import List, Optional

import torch

Tensor = torch.Tensor


def crf_llh(emissions: Tensor, tags: Tensor, mask: Optional[Tensor], start: Tensor, end: Tensor, trans: Tensor) -> Tensor:
    """Per-sample log-likelihood, batch-first emissions [B,S,C] (CRF.forward with reduction='none')."""
    B, S, C = emissions.shape
    if mask is None:
        mask = torch.ones(B, S, dtype=torch.bool)
    mask = mask.bool()
    e, t, m = emissions.transpose(0, 1), tags.transpose(0, 1), mask.transpose(0, 1)   # sequence-first, as the package
    ar = torch.arange(B)
    mf = m.to(emissions.dtype)
    # _compute_score
    score = start[t[0]] + e[0, ar, t[0]]
    for i in range(1, S):
        score = score + trans[t[i - 1], t[i]] * mf[i]
        score = score + e[i, ar, t[i]] * mf[i]
    seq_ends = m.long().sum(0) - 1
    score = score + end[t[seq_ends, ar]]
    # _compute_normalizer
    z = start + e[0]
    for i in range(1, S):
        nxt = torch.logsumexp(z.unsqueeze(2) + trans + e[i].unsqueeze(1), dim=1)
        z = torch.where(m[i].unsqueeze(1), nxt, z)
    z = torch.logsumexp(z + end, dim=1)
    return score - z


def crf_reduce(llh: Tensor, mask: Optional[Tensor], reduction: str) -> Tensor:
    if reduction == "none":
        return llh
    if reduction == "sum":
        return llh.sum()
    if reduction == "mean":
        return llh.mean()
    if reduction == "token_mean":
        n = mask.float().sum() if mask is not None else torch.tensor(float(llh.numel()))
        return llh.sum() / n
    raise ValueError("invalid reduction: %s" % reduction)


def crf_decode(emissions: Tensor, mask: Optional[Tensor], start: Tensor, end: Tensor, trans: Tensor) -> List[List[int]]:
    """_viterbi_decode: best tag sequence per sample (length = sum(mask))."""
    B, S, C = emissions.shape
    if mask is None:
        mask = torch.ones(B, S, dtype=torch.bool)
    e, m = emissions.transpose(0, 1), mask.bool().transpose(0, 1)
    score = start + e[0]
    history = []
    for i in range(1, S):
        nxt, idx = (score.unsqueeze(2) + trans + e[i].unsqueeze(1)).max(dim=1)
        score = torch.where(m[i].unsqueeze(1), nxt, score)
        history.append(idx)
    score = score + end
    seq_ends = m.long().sum(0) - 1
    out = []
    for b in range(B):
        best = [int(score[b].argmax())]
        for hist in reversed(history[:int(seq_ends[b])]):
            best.append(int(hist[b][best[-1]]))
        best.reverse()
        out.append(best)
    return out


def brute_force(emissions: Tensor, mask_len: int, start: Tensor, end: Tensor, trans: Tensor):
    """All C^L paths of ONE sample (emissions [S,C], first mask_len positions on): (log Z, best path, best score)."""
    L, C = mask_len, emissions.shape[1]
    scores, best, best_s = [], None, None
    for path in itertools.product(range(C), repeat=L):
        s = start[path[0]] + emissions[0, path[0]]
        for i in range(1, L):
            s = s + trans[path[i - 1], path[i]] + emissions[i, path[i]]
        s = s + end[path[-1]]
        scores.append(s)
        if best_s is None or s > best_s:
            best, best_s = list(path), s
    return torch.logsumexp(torch.stack(scores), 0), best, best_s
