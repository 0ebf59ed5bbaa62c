"""ConcatEncoders — per-camera encoders concatenated on the feature axis.

Mirrors hulc2.models.perceptual_encoders.concat_encoders.ConcatEncoders (reference concat_encoders.py:11-109)
for the configured cameras: rgb_static + rgb_gripper, proprio/depth/tactile: none.
"""
from typing import Dict, Optional

import torch
import torch.nn as nn

from hulc2_amd.compat import instantiate


class ConcatEncoders(nn.Module):
    def __init__(self, rgb_static, proprio, device, depth_static=None, rgb_gripper=None, depth_gripper=None, tactile=None,
                 state_decoder=None):
        super().__init__()
        for name, cfg in (("depth_static", depth_static), ("depth_gripper", depth_gripper), ("tactile", tactile),
                          ("state_decoder", state_decoder), ("proprio", proprio)):
            if cfg:
                raise NotImplementedError(f"{name} encoders are outside the accelerated path (cfg_low_level uses "
                                          "rgb_static + rgb_gripper only: conf/model/perceptual_encoder/gripper_cam.yaml)")
        self._latent_size = rgb_static["visual_features"] + (rgb_gripper["visual_features"] if rgb_gripper else 0)
        self.rgb_static_encoder = instantiate(rgb_static)
        self.depth_static_encoder = None
        self.rgb_gripper_encoder = instantiate(rgb_gripper) if rgb_gripper else None
        self.depth_gripper_encoder = None
        self.tactile_encoder = None
        self.proprio_encoder = None
        self.state_decoder = None
        self.current_visual_embedding = None
        self.current_state_obs = None

    @property
    def latent_size(self):
        return self._latent_size

    def forward(self, imgs: Dict[str, torch.Tensor], depth_imgs: Dict[str, torch.Tensor], state_obs: torch.Tensor) -> torch.Tensor:
        rgb_static = imgs["rgb_static"]
        b, s, c, h, w = rgb_static.shape
        enc = self.rgb_static_encoder(rgb_static.reshape(-1, c, h, w)).reshape(b, s, -1)
        if "rgb_gripper" in imgs and self.rgb_gripper_encoder is not None:
            g = imgs["rgb_gripper"]
            b, s, c, h, w = g.shape
            enc = torch.cat([enc, self.rgb_gripper_encoder(g.reshape(-1, c, h, w)).reshape(b, s, -1)], dim=-1)
        self.current_visual_embedding = enc.detach()   # detached: holding the graph across steps breaks HIP-graph capture
        self.current_state_obs = state_obs
        return enc
