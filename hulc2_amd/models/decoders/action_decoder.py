"""ActionDecoder interface (mirrors hulc2/models/decoders/action_decoder.py:7-46)."""
from typing import Optional, Tuple

import torch
from torch import nn


class ActionDecoder(nn.Module):
    def act(self, latent_plan, perceptual_emb, latent_goal, robot_obs: Optional[torch.Tensor] = None) -> torch.Tensor:
        raise NotImplementedError

    def loss(self, latent_plan, perceptual_emb, latent_goal, actions, robot_obs: Optional[torch.Tensor] = None) -> torch.Tensor:
        raise NotImplementedError

    def loss_and_act(self, latent_plan, perceptual_emb, latent_goal, actions,
                     robot_obs: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        raise NotImplementedError

    def _sample(self, *args, **kwargs):
        raise NotImplementedError

    def forward(self, latent_plan, perceptual_emb, latent_goal):
        raise NotImplementedError

    def clear_hidden_state(self) -> None:
        pass
