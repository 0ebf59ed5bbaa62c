"""Goal encoders: 3-layer MLP + LayerNorm on the MFMA GEMM chain.

Mirrors hulc2.models.encoders.goal_encoders.{VisualGoalEncoder, LanguageGoalEncoder} (reference
goal_encoders.py:8-71); keys mlp.{0,2,4} / mlp.{1,3,5} (+ ln) as in the reference's Sequentials.
"""
import torch
import torch.nn as nn

from hulc2_amd import functional as HF


def _check(l2: bool, act: str, who: str):
    if l2 or act != "ReLU":
        raise NotImplementedError(f"hulc2_amd {who}: configured path only (ReLU, no l2-normalise)")


class VisualGoalEncoder(nn.Module):
    def __init__(self, hidden_size: int, latent_goal_features: int, in_features: int, l2_normalize_goal_embeddings: bool,
                 activation_function: str):
        super().__init__()
        _check(l2_normalize_goal_embeddings, activation_function, "VisualGoalEncoder")
        self.l2_normalize_output = l2_normalize_goal_embeddings
        self.act_fn = nn.ReLU()
        self.mlp = nn.Sequential(nn.Linear(in_features, hidden_size), self.act_fn, nn.Linear(hidden_size, hidden_size), self.act_fn,
                                 nn.Linear(hidden_size, latent_goal_features))
        self.ln = nn.LayerNorm(latent_goal_features)

    def mlp_layers(self):
        m = self.mlp
        return [(m[0].weight, m[0].bias, True), (m[2].weight, m[2].bias, True), (m[4].weight, m[4].bias, False)]

    def forward(self, x: torch.Tensor, pre_ln: bool = False) -> torch.Tensor:
        y = HF.mlp(x, self.mlp_layers())
        if pre_ln:                 # Hulc2.training_step stacks the modalities' goals: the LayerNorms write the rows of one tensor
            return y
        return HF.layer_norm(y, self.ln.weight, self.ln.bias, self.ln.eps)


class LanguageGoalEncoder(nn.Module):
    def __init__(self, lang_net, in_features: int, hidden_size: int, latent_goal_features: int, l2_normalize_goal_embeddings: bool,
                 word_dropout_p: float, activation_function: str):
        super().__init__()
        _check(l2_normalize_goal_embeddings, activation_function, "LanguageGoalEncoder")
        if word_dropout_p != 0.0:
            raise NotImplementedError("word_dropout_p != 0 is not on the configured path (conf/model/language_goal/default.yaml)")
        self.lang_net = lang_net
        self.l2_normalize_output = l2_normalize_goal_embeddings
        self.act_fn = nn.ReLU()
        self.mlp = nn.Sequential(nn.Dropout(word_dropout_p), nn.Linear(in_features, hidden_size), self.act_fn,
                                 nn.Linear(hidden_size, hidden_size), self.act_fn, nn.Linear(hidden_size, latent_goal_features))
        self.ln = nn.LayerNorm(latent_goal_features)

    def mlp_layers(self):
        m = self.mlp
        return [(m[1].weight, m[1].bias, True), (m[3].weight, m[3].bias, True), (m[5].weight, m[5].bias, False)]

    def lo_operands(self):
        """rounding remainders the split-operand forward of the paired goal launch reads (precision site "goal"; trainer-maintained)"""
        m = self.mlp
        return [(m[1].weight, "lo"), (m[3].weight, "lo"), (m[5].weight, "lo")]

    def embed(self, x):
        """list[str] -> (B, 384) where the encoder carries its language network; SBERT stays third-party"""
        return self.lang_net(x) if self.lang_net is not None else x

    def forward(self, x, pre_ln: bool = False) -> torch.Tensor:
        y = HF.mlp(self.embed(x), self.mlp_layers(), x3=True)      # (precision site "goal": split-operand chain, the paired launch's arithmetic)
        if pre_ln:
            return y
        return HF.layer_norm(y, self.ln.weight, self.ln.bias, self.ln.eps)
