"""icka_amd -- MI355X-native (gfx950 / CDNA4) implementation of the ICKA multimodal-NER hot path.

Drop-in modules with the reference's names and signatures; every arithmetic step runs in hand-written HIP kernels
behind a C-ABI (include/icka_hip.h, icka_amd/libicka_hip.so).  No CPU path.
"""
from .config import BertConfig
from .modeling import (BertAttention, BertCoAttention, BertCrossAttention, BertCrossAttentionLayer,
                       BertCrossEncoder, BertEmbeddings, BertEncoder, BertIntermediate, BertLayer, BertLayerNorm,
                       BertModel, BertOutput, BertPooler, BertPreTrainedModel, BertSelfAttention, BertSelfEncoder,
                       BertSelfOutput, MTCCMBertForMMTokenClassificationCRF, cls_layer_both, scalar_gate_fusion,
                       resolved_precision, set_precision, token_ce_loss)
from .arena import ParamArena
from .crf import CRF
from .lstm import BiLSTM
from .modeling import MTCCMBertForMMTokenClassificationCRF_gate_1
from .dp import GradReducer
from . import graph
from .graph import DevicePrefetcher, GraphedModule, GraphedStep
from . import cross_modal
from .cross_modal import PromptRobertaModel

__all__ = ["CRF", "BiLSTM", "MTCCMBertForMMTokenClassificationCRF_gate_1", "BertConfig", "BertModel", "BertEmbeddings", "BertEncoder", "BertLayer", "BertLayerNorm", "BertPooler",
           "BertSelfEncoder", "BertCrossEncoder", "BertCrossAttentionLayer", "BertAttention", "BertCrossAttention",
           "BertSelfAttention", "BertCoAttention", "BertSelfOutput", "BertIntermediate", "BertOutput",
           "BertPreTrainedModel", "MTCCMBertForMMTokenClassificationCRF", "cls_layer_both", "scalar_gate_fusion",
           "token_ce_loss", "set_precision", "resolved_precision", "ParamArena", "cross_modal", "PromptRobertaModel",
           "GradReducer", "graph", "DevicePrefetcher", "GraphedModule", "GraphedStep"]
