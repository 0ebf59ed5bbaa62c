"""Projection heads of the CLIP-style auxiliary loss.

Mirrors hulc2.models.auxiliary_loss_networks.proj_vis_lang.ProjVisLang (reference proj_vis_lang.py:7-27).
"""
from typing import Tuple

import torch
import torch.nn as nn

from hulc2_amd import functional as HF


class ProjVisLang(nn.Module):
    def __init__(self, im_dim: int, lang_dim: int, output_dim: int, proj_lang: bool = True):
        super().__init__()
        self.mlp_im = nn.Sequential(nn.Linear(im_dim, 128), nn.ReLU(), nn.Linear(128, output_dim))
        self.mlp_lang = nn.Sequential(nn.Linear(lang_dim, 128), nn.ReLU(), nn.Linear(128, output_dim)) if proj_lang else None

    def forward(self, vis_emb: torch.Tensor, lang_emb: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        m = self.mlp_im
        vis = HF.mlp(vis_emb, [(m[0].weight, m[0].bias, True), (m[2].weight, m[2].bias, False)])
        if self.mlp_lang is not None:
            m = self.mlp_lang
            lang_emb = HF.mlp(lang_emb, [(m[0].weight, m[0].bias, True), (m[2].weight, m[2].bias, False)])
        return vis, lang_emb
