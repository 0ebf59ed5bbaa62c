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
        if isinstance(imgs, (list, tuple)):             # several modalities at once (Hulc2.training_step): see forward_multi
            return self.forward_multi(imgs)
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

    def forward_multi(self, imgs_list) -> torch.Tensor:
        """Several observation dicts of identical shapes (the modalities of one training step) through the shared encoders
        in one pass: rows of the result are modality-major, (sum B, S, latent).  Same arithmetic per frame as `forward`."""
        st = [im["rgb_static"] for im in imgs_list]
        b, s, c, h, w = st[0].shape
        enc = self.rgb_static_encoder([x.reshape(-1, c, h, w) for x in st]).reshape(len(st) * b, s, -1)
        if self.rgb_gripper_encoder is not None and all("rgb_gripper" in im for im in imgs_list):
            gr = [im["rgb_gripper"] for im in imgs_list]
            _, _, c, h, w = gr[0].shape
            enc = torch.cat([enc, self.rgb_gripper_encoder([x.reshape(-1, c, h, w) for x in gr]).reshape(len(gr) * b, s, -1)], dim=-1)
        self.current_visual_embedding = enc.detach()
        return enc
