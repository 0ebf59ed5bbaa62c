"""Plan recognition (posterior) transformer.

Mirrors hulc2.models.plan_encoders.plan_recognition_net.PlanRecognitionTransformersNetwork (reference
plan_recognition_net.py:77-148): learned positions, 2 post-norm encoder layers (8 heads of 16), fc 128->4096,
mean over the sequence, fc_state.  nn.TransformerEncoder is instantiated only as the parameter container that
yields the reference's state_dict keys; the arithmetic is HIP (MFMA GEMM chains, per-(batch,head) attention
waves, fused residual+dropout+LayerNorm).

The sequence mean is taken before the 128->4096 projection: mean_s(W x_s + b) = W mean_s(x_s) + b exactly, which
removes 31/32 of that layer's FLOPs.
"""
from typing import Tuple

import torch
import torch.nn as nn

from hulc2_amd import functional as HF
from hulc2_amd import kernels as kn
from hulc2_amd.utils.distributions import Distribution, State


class PlanRecognitionTransformersNetwork(nn.Module):
    def __init__(self, num_heads: int, num_layers: int, encoder_hidden_size: int, fc_hidden_size: int, plan_features: int,
                 in_features: int, action_space: int, encoder_normalize: bool, positional_normalize: bool,
                 position_embedding: bool, max_position_embeddings: int, dropout_p: float, dist: Distribution):
        super().__init__()
        if in_features % num_heads != 0 or in_features // num_heads != 16:
            raise NotImplementedError("attention kernel is specialised for head_dim 16 without padding (128 / 8)")
        if not position_embedding or encoder_normalize or positional_normalize:
            raise NotImplementedError("configured path: learned position embedding, no extra norms "
                                      "(conf/model/plan_recognition/transformers.yaml)")
        self.in_features, self.plan_features, self.action_space = in_features, plan_features, action_space
        self.padding = False
        self.dist = dist
        self.hidden_size = fc_hidden_size
        self.position_embedding, self.encoder_normalize, self.positional_normalize = True, False, False
        self.num_heads, self.num_layers, self.dropout_p = num_heads, num_layers, float(dropout_p)
        self.position_embeddings = nn.Embedding(max_position_embeddings, in_features)
        layer = nn.TransformerEncoderLayer(in_features, num_heads, dim_feedforward=encoder_hidden_size, dropout=float(dropout_p))
        self.layernorm = nn.LayerNorm(in_features)
        self.dropout = nn.Dropout(p=float(dropout_p))
        self.transformer_encoder = nn.TransformerEncoder(layer, num_layers=num_layers, norm=None, enable_nested_tensor=False)
        self.fc = nn.Linear(in_features, fc_hidden_size)
        self.fc_state = self.dist.build_state(fc_hidden_size, plan_features)

    def _layer_params(self, l: int) -> dict:
        m = self.transformer_encoder.layers[l]
        return {"in_proj_weight": m.self_attn.in_proj_weight, "in_proj_bias": m.self_attn.in_proj_bias,
                "out_proj.weight": m.self_attn.out_proj.weight, "out_proj.bias": m.self_attn.out_proj.bias,
                "linear1.weight": m.linear1.weight, "linear1.bias": m.linear1.bias,
                "linear2.weight": m.linear2.weight, "linear2.bias": m.linear2.bias,
                "norm1.weight": m.norm1.weight, "norm1.bias": m.norm1.bias,
                "norm2.weight": m.norm2.weight, "norm2.bias": m.norm2.bias}

    def frag_operands(self):
        """(weight, hulc_ffn_frag_perm layout) of the feed-forward weights the whole-trunk launch reads fragment-packed: the trainer keeps
        these copies fresh with one gather launch per step"""
        out = []
        for m in self.transformer_encoder.layers:
            out += [(m.linear1.weight, 0), (m.linear2.weight, 1), (m.linear2.weight, 2), (m.linear1.weight, 3)]
        return out

    def lo_operands(self):
        """(weight, layout) of the rounding remainders w - bf16(w) the split-operand forward of the whole-trunk launch reads (selective
        precision site "txl"): kept fresh by the trainer with one residual + one gather launch per step"""
        out = []
        for m in self.transformer_encoder.layers:
            out += [(m.self_attn.in_proj_weight, "lo"), (m.self_attn.out_proj.weight, "lo"), (m.linear1.weight, "ffn_p0_lo"), (m.linear2.weight, "ffn_p1_lo")]
        return out

    def _position_ids(self, S: int, device) -> torch.Tensor:
        """arange(S) (plan_recognition_net.py:133): kept per (length, device) instead of one launch per step"""
        cache = self.__dict__.setdefault("_pos_id_cache", {})
        ids = cache.get((S, device))
        if ids is None:
            ids = cache[(S, device)] = torch.arange(S, dtype=torch.long, device=device)
            ids._hulc_arange = True          # (rows 0..S-1 in order: the table's gradient rows are the batch sum itself)
        return ids

    def forward(self, perceptual_emb: torch.Tensor) -> Tuple[State, torch.Tensor]:
        B, S, E = perceptual_emb.shape
        p = self.dropout_p if self.training else 0.0
        seed = 0x5EED0001           # site id; the per-step stream comes from the device step state (kernels.step_state)
        position_ids = self._position_ids(S, perceptual_emb.device)
        layers = [self._layer_params(l) for l in range(self.num_layers)]
        if HF.txl_block_ok(perceptual_emb, layers, S, self.num_heads):
            # position embedding -> every layer -> sequence mean: one launch per direction, one workgroup per sequence (csrc/txl_block.hip)
            pooled = HF.transformer_trunk_pooled(perceptual_emb, self.position_embeddings.weight, position_ids, layers, self.num_heads, p, seed)
        else:
            x = HF.AddPosFn.apply(perceptual_emb, self.position_embeddings.weight, position_ids, p, seed, True)
            x = x.reshape(B * S, E)
            with kn.site_scope("txl"):
                for l in range(self.num_layers):
                    x = HF.transformer_encoder_layer(x, layers[l], B, S, self.num_heads, p, seed + 100 * (l + 1))
            with kn.site_scope("pool"):
                pooled = HF.SeqMeanFn.apply(x.reshape(B, S, E))
        # selective precision: seq_feat feeds the contrastive head, whose gradient is a cancelling remainder of nearly identical
        # rows — its projection runs exact-fp32 inside a bf16 step (67 MFLOP of the step's 904 GFLOP)
        with kn.site_scope("head"):
            seq_feat = HF.mlp(pooled, [(self.fc.weight, self.fc.bias, False)])
        logits = HF.mlp(seq_feat, [(self.fc_state[0].weight, self.fc_state[0].bias, False)])
        return self.dist.forward_dist(logits), seq_feat
