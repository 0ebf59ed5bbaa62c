"""Plan proposal (prior) network — a 5-GEMM MFMA chain.

Mirrors hulc2.models.plan_encoders.plan_proposal_net.PlanProposalNetwork (reference plan_proposal_net.py:8-47);
keys fc_model.{0,2,4,6}, fc_state.0.
"""
import torch
import torch.nn as nn

from hulc2_amd import functional as HF
from hulc2_amd.utils.distributions import Distribution, State


class PlanProposalNetwork(nn.Module):
    def __init__(self, perceptual_features: int, latent_goal_features: int, plan_features: int, activation_function: str,
                 hidden_size: int, dist: Distribution):
        super().__init__()
        if activation_function != "ReLU":
            raise NotImplementedError("configured path uses ReLU (conf/model/plan_proposal/default.yaml)")
        self.perceptual_features, self.latent_goal_features = perceptual_features, latent_goal_features
        self.plan_features, self.hidden_size = plan_features, hidden_size
        self.in_features = perceptual_features + latent_goal_features
        self.act_fn = nn.ReLU()
        self.dist = dist
        h = hidden_size
        self.fc_model = nn.Sequential(nn.Linear(self.in_features, h), self.act_fn, nn.Linear(h, h), self.act_fn, nn.Linear(h, h),
                                      self.act_fn, nn.Linear(h, h), self.act_fn)
        self.fc_state = self.dist.build_state(h, plan_features)

    def forward(self, initial_percep_emb: torch.Tensor, latent_goal: torch.Tensor) -> State:
        x = torch.cat([initial_percep_emb, latent_goal], dim=-1)
        f = self.fc_model
        layers = [(f[i].weight, f[i].bias, True) for i in (0, 2, 4, 6)] + [(self.fc_state[0].weight, self.fc_state[0].bias, False)]
        return self.dist.forward_dist(HF.mlp(x, layers))
