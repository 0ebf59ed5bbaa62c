"""Latent-plan distribution helper.

Mirrors hulc2.utils.distributions.Distribution (reference hulc2/utils/distributions.py:15-60) for the
configured discrete 32x32 one-hot categorical.  `get_dist` still returns torch.distributions objects for
callers outside the hot path (validation / t-SNE); the training step itself uses the fused HIP kernels
through `rsample_plan` / `kl_balanced`.
"""
from collections import namedtuple
from typing import Optional, Union

import torch
import torch.nn as nn

from hulc2_amd import functional as HF

DiscState = namedtuple("DiscState", ["logit"])
ContState = namedtuple("ContState", ["mean", "std"])
State = Union[DiscState, ContState]


class Distribution:
    def __init__(self, **kwargs):
        self.dist = kwargs.get("dist")
        assert self.dist == "discrete" or self.dist == "continuous"
        if self.dist == "continuous":
            raise NotImplementedError("continuous latent plans are not on the configured path (conf/model/distribution/discrete.yaml)")
        self.category_size = kwargs.get("category_size")
        self.class_size = kwargs.get("class_size")

    # ---- reference API -------------------------------------------------------------------------
    def get_dist(self, state):
        from torch.distributions import Independent, OneHotCategoricalStraightThrough
        shape = state.logit.shape
        logits = torch.reshape(state.logit, shape=(*shape[:-1], self.category_size, self.class_size))
        return Independent(OneHotCategoricalStraightThrough(logits=logits), 1)

    def detach_state(self, state):
        return DiscState(state.logit.detach())

    def sample_latent_plan(self, distribution):
        return torch.flatten(distribution.sample(), start_dim=-2, end_dim=-1)

    def build_state(self, hidden_size, plan_features):
        return nn.Sequential(nn.Linear(hidden_size, plan_features))

    def forward_dist(self, x):
        return DiscState(x)

    # ---- fused hot-path entry points -----------------------------------------------------------
    def rsample_plan(self, state: DiscState, seed: int, idx: Optional[torch.Tensor] = None):
        """pr_dist.rsample() flattened (hulc2.py:235-237): straight-through one-hot; returns (plan, idx)."""
        return HF.PlanSampleFn.apply(state.logit, idx, self.category_size, self.class_size, int(seed))

    def kl_balanced(self, pp_state: DiscState, pr_state: DiscState, kl_beta: float, mix: float) -> torch.Tensor:
        """Hulc2.compute_kl_loss (hulc2.py:444-466) in one kernel pair."""
        return HF.CatKLFn.apply(pp_state.logit, pr_state.logit, self.category_size, self.class_size, float(kl_beta), float(mix))

    def kl_balanced_segments(self, pp_state: DiscState, pr_state: DiscState, kl_beta: float, mix: float, nseg: int) -> torch.Tensor:
        """the same loss for nseg modalities stacked on the batch axis: (nseg,) values, each the mean over its own rows"""
        return HF.CatKLFn.apply(pp_state.logit, pr_state.logit, self.category_size, self.class_size, float(kl_beta), float(mix), int(nseg))

    def rsample_plan_and_kl(self, pp_state: DiscState, pr_state: DiscState, seed: int, idx, kl_beta: float, mix: float, nseg: int = 1):
        """rsample_plan + kl_balanced_segments as one autograd node -> (plan, idx, kl (nseg,))"""
        return HF.PlanSampleKLFn.apply(pp_state.logit, pr_state.logit, idx, self.category_size, self.class_size, int(seed), float(kl_beta),
                                       float(mix), int(nseg))
