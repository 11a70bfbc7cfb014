"""The reference's current (published) ICKA tagger, Cross_Modal_Interaction_Module.py:887-1057, on the HIP kernels.

SURVEY.md section 8f, last row: the prompt mapping networks (:914-928) and an in-tree replacement for the
prompt-accepting RoBERTa stage (``last_encoder``, :1010-1012), which together with the trunk, the single-query alignment
cross-encoders, the scalar gate, the BiLSTM, the classifier and the CRF complete the model.

``last_encoder`` comes from a package that is NOT in the reference tree (``local_transformers.adapter_transformers``,
imported at My_cross_attention.py:4): its arithmetic is unknown, nothing pins it (parity unpinned).  What IS pinned by
the reference is its call contract, and ``PromptRobertaModel`` below implements exactly that contract on a BERT-layer
stack (RoBERTa's layer is the BERT layer):
  * called as ``last_encoder(input_ids=, token_type_ids=, attention_mask=, prompt_embeddings=[B,P,Hr], input_mask=[B,P],
    offset=)`` and indexed ``[0]`` (:1010-1013);
  * output length = ``input_ids.size(1) - 2 + P`` (comment :1014, offset arithmetic :1022): the two ``<mask>`` tokens of
    the prompt text 'Image is <mask> Bridge between Image and the Text is <mask>' (My_cross_attention.py:293-294) are
    replaced by the prompt vectors -- the vision prefix at the first, the alignment prompt at the second, in the order
    ``prefix_emb = cat([prefix_vision, Alignment_prompt])`` (:1001);
  * the mask tokens sit at token positions 3 and 11 (the reference's own token dump, My_cross_attention.py:402-404).
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
import torch.nn as nn

from . import exact as X
from . import kernels as K
from . import ops
from .config import check_config
from .modeling import (BF16, F32, BertCrossEncoder, BertEncoder, BertLayerNorm, BertModel, BertPreTrainedModel,
                       BertSelfEncoder, _CastFn, _IckaModule, _dims, _is_exact, _mner_trunk, _with_twin, cls_layer_both)


class PromptRobertaEmbeddings(_IckaModule):
    """word / position / token-type tables + LayerNorm + dropout with RoBERTa's conventions: ``padding_idx`` 1,
    positions start at ``padding_idx + 1`` (state_dict keys as HF ``RobertaEmbeddings``)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.padding_idx = getattr(config, "pad_token_id", 1)
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=self.padding_idx)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=getattr(config, "layer_norm_eps", 1e-5))
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, input_ids, src, prompt_embeddings):
        B = input_ids.shape[0]
        S = src.shape[0]
        if S + self.padding_idx + 1 > self.position_embeddings.weight.shape[0]:
            raise IndexError("spliced length %d exceeds max_position_embeddings" % S)
        A = self._arena()
        if _is_exact(self):
            d = _dims(self.config, B, S, 0, self.training, True)
            return X.PromptEmbeddingsFn.apply(A.anchor, prompt_embeddings, self, A, input_ids.contiguous(), src, d,
                                              self.padding_idx + 1).view(B, S, -1)
        d = _dims(self.config, B, S, 0, self.training)
        y, yf = ops.PromptEmbeddingsFn.apply(A.anchor, prompt_embeddings, self, A, input_ids.contiguous(), src, d,
                                             self.padding_idx + 1)
        return _with_twin(y, yf, (B, S, -1))


class PromptRobertaModel(BertPreTrainedModel):
    """In-tree stand-in for the reference's absent ``last_encoder`` (module docstring).  ``config`` is a BertConfig with
    RoBERTa geometry (roberta-large: vocab 50265, hidden 1024, 24 layers, 16 heads, intermediate 4096, 514 positions,
    1 token type, layer_norm_eps 1e-5).  ``mask_positions``: token positions of the prompt text's ``<mask>`` tokens.

    forward(input_ids, token_type_ids, attention_mask, prompt_embeddings, input_mask, offset) -> (last_hidden_state,)
    with last_hidden_state [B, S_in - len(mask_positions) + P, hidden] (bf16, f32 twin attached)."""

    def __init__(self, config, mask_positions: Sequence[int] = (3, 11)):
        super().__init__(config)
        check_config(config)
        self.mask_positions = tuple(int(p) for p in mask_positions)
        if list(self.mask_positions) != sorted(set(self.mask_positions)) or not self.mask_positions:
            raise ValueError("mask_positions must be increasing token positions")
        self.embeddings = PromptRobertaEmbeddings(config)
        self.encoder = BertEncoder(config)
        self._src_cache = {}
        self.apply(self.init_bert_weights)

    def splice_index(self, S_in: int, P: int, device) -> torch.Tensor:
        """int32 [S_in - n_mask + P]: >= 0 token position, < 0 prompt vector -1-v; the P prompt vectors are dealt to
        the mask positions in equal consecutive shares (5 + 5 in the reference)."""
        key = (S_in, P, str(device))
        src = self._src_cache.get(key)
        if src is None:
            n = len(self.mask_positions)
            if P % n or self.mask_positions[-1] >= S_in:
                raise ValueError("cannot deal %d prompt vectors to mask positions %s of a %d-token input"
                                 % (P, self.mask_positions, S_in))
            share, out, j = P // n, [], 0
            for t in range(S_in):
                if t in self.mask_positions:
                    out.extend(-1 - (j + i) for i in range(share))
                    j += share
                else:
                    out.append(t)
            src = torch.tensor(out, dtype=torch.int32, device=device)
            self._src_cache[key] = src
        return src

    def forward(self, input_ids=None, token_type_ids=None, attention_mask=None, prompt_embeddings=None,
                input_mask=None, offset=None):
        if input_ids is None or prompt_embeddings is None:
            raise ValueError("input_ids and prompt_embeddings are required")
        if not input_ids.is_cuda:
            raise TypeError("input_ids must be on a ROCm device: icka_amd has no CPU path")
        B, S_in = input_ids.shape
        P = prompt_embeddings.shape[1]
        H = self.config.hidden_size
        if prompt_embeddings.shape[0] != B or prompt_embeddings.shape[2] != H:
            raise ValueError("prompt_embeddings must be [B, P, %d]" % H)
        self._arena()
        src = self.splice_index(S_in, P, input_ids.device)
        S_out = src.shape[0]
        # Row alignment: the GEMM fast path wants B*S % 128 == 0 (B=32, S=178 -> 5696 rows would fall on the general
        # kernel for every GEMM of the stack).  Up to 7 masked-out positions are appended (they re-read the last token
        # id, get additive mask -10000 and are cut off again below), e.g. 178 -> 180.
        extra = next((e for e in range(8) if (B * (S_out + e)) % 128 == 0), 0)
        if extra and S_out + extra + self.embeddings.padding_idx + 1 <= self.embeddings.position_embeddings.weight.shape[0]:
            key = (S_in, P, extra, str(input_ids.device))
            padded = self._src_cache.get(key)
            if padded is None:
                padded = torch.cat([src, torch.full((extra,), S_in - 1, dtype=torch.int32, device=src.device)])
                self._src_cache[key] = padded
            src = padded
        else:
            extra = 0
        S = src.shape[0]
        # spliced 0/1 mask: token entries from attention_mask, prompt entries from input_mask (index plumbing only)
        am = torch.ones(B, S_in, dtype=torch.int64, device=input_ids.device) if attention_mask is None \
            else attention_mask.long()
        pm = torch.ones(B, P, dtype=torch.int64, device=input_ids.device) if input_mask is None else input_mask.long()
        idx = src.long()
        spliced = torch.where((idx >= 0).unsqueeze(0), am[:, idx.clamp(min=0)], pm[:, (-1 - idx).clamp(min=0)])
        if extra:
            spliced[:, S_out:] = 0
        add_mask = K.additive_mask(spliced.contiguous(), S, torch.empty(B, S, dtype=F32, device=input_ids.device))
        pe = prompt_embeddings
        if _is_exact(self):
            pe = pe if pe.dtype == F32 else _CastFn.apply(pe.contiguous(), False)
        elif pe.dtype != BF16:
            pe = _CastFn.apply(pe.contiguous(), True)
        x = self.embeddings(input_ids, src, pe)
        out = self.encoder(x, add_mask.view(B, 1, 1, S), output_all_encoded_layers=False)[-1]
        return (out[:, :S_out] if extra else out,)


class MTCCMBertForMMTokenClassificationCRF(BertPreTrainedModel):
    """Cross_Modal_Interaction_Module.MTCCMBertForMMTokenClassificationCRF (:887-1057): same constructor arguments,
    sub-module names (= state_dict keys) and forward signature.

    ``embedding``: the text encoder; an icka ``BertModel`` (default: built from ``config``).  ``last_encoder``: a
    ``PromptRobertaModel`` (default: built from ``last_encoder_config``; its hidden size must equal ``config``'s, as
    the reference's blend ``gate*token_embedding + (1-gate)*cross_output_layer`` (:1036) requires).  Foreign CPU/eager
    modules are rejected: there is no fallback path.

    forward(..., mode='train') -> CRF loss; 'dev' -> (tags, loss); 'test' -> tags; mode=None -> emissions."""

    def __init__(self, config, embedding=None, last_encoder=None, layer_num1=1, layer_num2=1, layer_num3=1,
                 num_labels=2, last_encoder_config=None, max_seq_length=128):
        super().__init__(config)
        check_config(config)
        from .crf import CRF
        from .lstm import BiLSTM
        H = config.hidden_size
        self.num_labels = num_labels
        self.max_seq_length = max_seq_length   # the reference hard-codes 128 (:1024)
        if last_encoder is None:
            if last_encoder_config is None:
                raise ValueError("pass last_encoder (a PromptRobertaModel) or last_encoder_config")
            last_encoder = PromptRobertaModel(last_encoder_config)
        if not isinstance(last_encoder, PromptRobertaModel):
            raise TypeError("last_encoder must be an icka_amd.cross_modal.PromptRobertaModel (no eager fallback)")
        if embedding is not None and not isinstance(embedding, BertModel):
            raise TypeError("embedding must be an icka_amd BertModel (no eager fallback)")
        self.last_encoder = last_encoder
        self.bert = embedding if embedding is not None else BertModel(config)
        self.hidden_size = H
        self.self_attention = BertSelfEncoder(config)        # constructed by the reference, unused by its forward
        self.self_attention_v2 = BertSelfEncoder(config)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.vismap2text = nn.Linear(2048, H)
        self.vismapping = nn.Linear(512, H)
        self.txt2img_attention = BertCrossEncoder(config, layer_num1)
        self.cls_layer_Y = nn.ModuleList([BertCrossEncoder(config, layer_num1) for _ in range(2)])
        self.embedding_layer = nn.Embedding(config.vocab_size, H)   # unused by the forward (:951)
        self.lstm = BiLSTM(input_size=H, hidden_size=H, batch_first=True, bidirectional=True)
        self.classifier = nn.Linear(H * 2, num_labels)
        self.crf = CRF(num_tags=num_labels, batch_first=True)
        self.prompt_len = 5
        self.mapping_network_alignment = nn.Sequential(
            nn.Dropout(p=0.3), nn.Linear(H, 756 * self.prompt_len, bias=True), nn.Tanh(), nn.Dropout(p=0.3),
            nn.Linear(756 * self.prompt_len, H * self.prompt_len, bias=True))
        self.mapping_network_vision = nn.Sequential(
            nn.Dropout(p=0.3), nn.Linear(2048, 756 * self.prompt_len, bias=True), nn.Tanh(), nn.Dropout(p=0.3),
            nn.Linear(756 * self.prompt_len, H * self.prompt_len, bias=True))
        self.lastproj = nn.Linear(H, 1024)
        self.cls_layer = cls_layer_both(H, H)
        self.aux_head = nn.Linear(H, 1)
        self.LayerNorm = nn.LayerNorm(H, eps=getattr(config, "layer_norm_eps", 1e-12))   # unused by the forward
        Hr = last_encoder.config.hidden_size
        if Hr != H:
            raise ValueError("last_encoder hidden size %d != config.hidden_size %d (the blend at :1036 adds them)" % (Hr, H))
        if embedding is None:
            own = [m for n, m in self.named_children() if n not in ("last_encoder", "bert")]
            for m in own:
                m.apply(self.init_bert_weights)
            self.bert.apply(self.init_bert_weights)
            self.lstm.reset_parameters()
            self.crf.reset_parameters()

    # ------------------------------------------------------------------------------------------------ pieces
    def _mapping(self, A, seq_mod: nn.Sequential, x2d: torch.Tensor) -> torch.Tensor:
        p = float(seq_mod[0].p) if self.training else 0.0
        if _is_exact(self):
            return X.prompt_mapping(A, x2d, seq_mod[1], seq_mod[4], p)
        return ops.PromptMappingFn.apply(A.anchor, x2d, seq_mod[1], seq_mod[4], A, p)

    def prompts(self, A, clip_aligned: torch.Tensor, visual_embeds_mean: torch.Tensor) -> torch.Tensor:
        """prefix_emb [B, 2*prompt_len, Hr] (:995-1004).  clip_aligned bf16 [B,H]."""
        B = clip_aligned.shape[0]
        H = self.hidden_size
        ex = _is_exact(self)
        align = self._mapping(A, self.mapping_network_alignment, clip_aligned).view(B, self.prompt_len, H)
        vm = visual_embeds_mean.reshape(B, 2048)
        if ex:
            vm = vm.float().contiguous()
        else:
            vm = vm if vm.dtype == BF16 else _CastFn.apply(vm.float().contiguous(), True)
        vision = self._mapping(A, self.mapping_network_vision, vm).view(B, self.prompt_len, H)
        prefix = torch.cat([vision, align], dim=1)           # [B, 10, H]   (concatenation: data movement only)
        if H != 1024:                                         # :1002-1003
            flat = prefix.view(B * 2 * self.prompt_len, H)
            prefix = (X.LinearFn.apply(A.anchor, flat, self.lastproj, A, False) if ex else
                      ops.LinearFn.apply(A.anchor, flat, self.lastproj, A, False, K.EPI_NONE)).view(B, 2 * self.prompt_len, 1024)
        return prefix

    def resolve_offset(self, offsets) -> int:
        """The host integer the reference reads back from the device EVERY step (``offsets.tolist()[0]``, :949; SURVEY section 7
        lists that sync under quirks not to copy).  The offset is a property of the prompt template, the same for every sample
        of a data set (the reference's loop asserts it, My_cross_attention.py:803): an ``int`` or a host tensor is used as it is;
        a device tensor is read back ONCE per model (``set_offset(k)`` states it up front or changes it) -- no per-step sync,
        and the step can be captured (a device tensor with nothing resolved yet raises under capture instead of syncing)."""
        if isinstance(offsets, int):
            return offsets
        if isinstance(offsets, torch.Tensor) and not offsets.is_cuda:
            return int(offsets.reshape(-1)[0])
        off = getattr(self, "_offset_host", None)
        if off is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("offsets is a device tensor and no offset has been resolved yet: pass an int / host tensor, "
                                   "call model.set_offset(k), or run one eager step before capturing")
            off = self._offset_host = int(offsets.reshape(-1)[0].item())
        return off

    def set_offset(self, offset: Optional[int]) -> None:
        """State the prompt template's offset (None: resolve it again from the next device tensor)."""
        self._offset_host = None if offset is None else int(offset)

    def emissions(self, input_ids, segment_ids, input_mask, ori_input_ids, ori_input_mask, ori_segment_ids,
                  added_attention_mask, clip_features, visual_embeds_mean, visual_embeds_att, offsets):
        B, S = ori_input_ids.shape
        H = self.hidden_size
        dev = ori_input_ids.device
        offset = self.resolve_offset(offsets)                                               # :949, without its per-step sync
        # ---- trunk: text encoder -> dropout -> regions -> vismap2text -> text->image cross encoder (:950-969)
        A, _, _, cross, _ = _mner_trunk(self, ori_input_ids, ori_segment_ids, ori_input_mask, added_attention_mask,
                                        visual_embeds_att)
        ex = _is_exact(self)
        cross_k, cross_gate = (X.FanOutFn if ex else ops.FanOutFn).apply(cross, 2)   # alignment K/V source; gate + blend
        # ---- CLIP token -> hidden (:954), then two single-query cross encoders over the text (:981-989)
        cf = clip_features.reshape(B, 512)
        if ex:
            clip = X.LinearFn.apply(A.anchor, cf.float().contiguous(), self.vismapping, A, False).view(B, 1, H)
        else:
            cf = cf if cf.dtype == BF16 else _CastFn.apply(cf.float().contiguous(), True)
            clip = ops.LinearFn.apply(A.anchor, cf, self.vismapping, A, False, K.EPI_NONE).view(B, 1, H)
        text_mask = K.additive_mask(ori_input_mask if ori_input_mask.dtype == torch.int64 else ori_input_mask.long(),
                                    S, torch.empty(B, S, dtype=F32, device=dev)).view(B, 1, 1, S)
        cross3 = cross_k.view(B, S, H)
        for enc in self.cls_layer_Y:
            clip = enc(clip, cross3, text_mask)[-1]
        # ---- prompts (:995-1004) and the prompt-accepting encoder (:1006-1013)
        prefix = self.prompts(A, clip.reshape(B, H), visual_embeds_mean)
        P = prefix.shape[1]
        ones = input_mask[:, :1].repeat(1, P)                                               # :1006-1008
        enc_out = self.last_encoder(input_ids=input_ids, token_type_ids=segment_ids, attention_mask=input_mask,
                                    prompt_embeddings=prefix, input_mask=ones, offset=offset)[0]
        off2 = offset - len(self.last_encoder.mask_positions) + P                           # :1022 (offset - 2 + 10)
        if off2 + self.max_seq_length > enc_out.shape[1] or self.max_seq_length != S:
            raise ValueError("token window [%d, %d) does not fit the encoder output of length %d / text length %d"
                             % (off2, off2 + self.max_seq_length, enc_out.shape[1], S))
        token_embedding = enc_out[:, off2:off2 + self.max_seq_length, :].contiguous()       # :1024 (copy: data movement)
        # ---- scalar gate + blend (:1029-1036), BiLSTM, classifier (:1042-1043)
        if ex:
            from .modeling import scalar_gate_fusion
            result = scalar_gate_fusion(self, cross_gate.view(B, S, H), token_embedding)
        else:
            result = _scalar_gate(self, A, cross_gate, token_embedding.view(B * S, H), B, S)
        x, _ = self.lstm(result.view(B, S, H))
        if ex:
            em = X.LinearFn.apply(A.anchor, x.reshape(B * S, 2 * H), self.classifier, A, False)
        else:
            em = ops.LinearFn.apply(A.anchor, x.reshape(B * S, 2 * H), self.classifier, A, True, K.EPI_NONE)
        return em.view(B, S, self.num_labels)

    def forward(self, input_ids, segment_ids, input_mask, ori_input_ids, ori_input_mask, ori_segment_ids,
                added_attention_mask, clip_features, visual_embeds_mean, visual_embeds_att, offsets, output_mask,
                rela_score=None, temp=None, temp_lamb=None, lamb=None, labels=None, negative_rate=None, mode=None):
        emissions = self.emissions(input_ids, segment_ids, input_mask, ori_input_ids, ori_input_mask, ori_segment_ids,
                                   added_attention_mask, clip_features, visual_embeds_mean, visual_embeds_att, offsets)
        output_mask = (output_mask != 0)
        if mode == "train":
            return -self.crf(emissions, tags=labels, mask=output_mask, reduction="token_mean")
        if mode == "dev":
            pred_tags = self.crf.decode(emissions, mask=output_mask)
            return pred_tags, -self.crf(emissions, tags=labels, mask=output_mask, reduction="token_mean")
        if mode == "test":
            return self.crf.decode(emissions, mask=output_mask)
        return emissions


def _scalar_gate(owner, A, cross2d: torch.Tensor, tok2d: torch.Tensor, B: int, S: int) -> torch.Tensor:
    """:1029-1036 on 2-D [B*S,H] bf16 operands (modeling.scalar_gate_fusion without the arena bookkeeping)."""
    H = cross2d.shape[1]
    c0 = cross2d.view(B, S, H)[:, 0]
    t0 = tok2d.view(B, S, H)[:, 0]
    feat = ops.AddLayerNormFn.apply(A.anchor, c0, t0, owner.cls_layer.proj_norm, A, float(owner.cls_layer.proj_norm.eps))
    related = ops.LinearFn.apply(A.anchor, feat, owner.cls_layer.proj, A, False, K.EPI_NONE)
    logit = ops.LinearFn.apply(A.anchor, related, owner.aux_head, A, True, K.EPI_NONE)        # f32 [B,1]
    return ops.SampleGateFn.apply(tok2d, cross2d, logit.view(B), 0, B, S)
