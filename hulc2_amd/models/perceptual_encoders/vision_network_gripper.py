"""Gripper-camera encoder (nature-CNN) on MI355X kernels.

Mirrors hulc2.models.perceptual_encoders.vision_network_gripper.VisionNetwork (reference
vision_network_gripper.py:11-26,57-89): keys conv_model.{0,2,4,7}, fc1.0, fc2, ln.  The reference's Flatten
runs over NCHW, so the NHWC conv output is re-ordered (a 3136-float copy per frame) before the first Linear.
"""
import os
from typing import Tuple

import torch
import torch.nn as nn

from hulc2_amd import functional as HF
from hulc2_amd import kernels as kn


def nature_cnn(act_fn, num_c):
    return nn.Sequential(nn.Conv2d(num_c, 32, 8, stride=4), act_fn, nn.Conv2d(32, 64, 4, stride=2), act_fn,
                         nn.Conv2d(64, 64, 3, stride=1), act_fn, nn.Flatten(start_dim=1), nn.Linear(64 * 7 * 7, 128), act_fn)


class VisionNetwork(nn.Module):
    def __init__(self, input_width: int, input_height: int, conv_encoder: str, activation_function: str,
                 dropout_vis_fc: float, l2_normalize_output: bool, visual_features: int, num_c: int):
        super().__init__()
        if conv_encoder != "nature_cnn" or activation_function != "ReLU" or l2_normalize_output or dropout_vis_fc != 0.0:
            raise NotImplementedError("hulc2_amd gripper VisionNetwork implements the configured path only: nature_cnn, ReLU, "
                                      "no l2-normalise, dropout 0 (conf/model/perceptual_encoder/rgb_gripper/default.yaml)")
        self.act_fn = nn.ReLU()
        self.conv_model = nature_cnn(self.act_fn, num_c)
        self.fc1 = nn.Sequential(nn.Linear(128, 512), self.act_fn, nn.Dropout(dropout_vis_fc))
        self.fc2 = nn.Linear(512, visual_features)
        self.ln = nn.LayerNorm(visual_features)

    def flatten_linears(self):
        """(weight, (C, H, W)) of Linear layers behind nn.Flatten: the trainer keeps their (h, w, c)-ordered bf16 shadows fresh"""
        return [(self.conv_model[7].weight, (64, 7, 7))]

    def lo_operands(self):
        """rounding remainders the split-operand forward of the fc1 -> fc2 head reads (precision site "encfc"; trainer-maintained)"""
        return [(self.fc1[0].weight, "lo"), (self.fc2.weight, "lo"), (self.conv_model[0].weight, "oihw_flat_lo")]

    def conv_params(self):
        c = self.conv_model
        return (c[0].weight, c[0].bias, c[2].weight, c[2].bias, c[4].weight, c[4].bias)

    def forward(self, x: torch.Tensor, aug_shift=None, aug_pad: int = 0, frame_index=None, pre_ln: bool = False) -> torch.Tensor:
        # (site "encfc": the exact-fp32 flatten-linear below gets the EXACT map — conv3 stores it next to the bf16 one, hulc_conv_desc.y_bf16)
        a3 = HF.conv_stack(x, self.conv_params(), grad_premasked=True, aug_pad=aug_pad, aug_shifts=aug_shift, frame_index=frame_index,
                           exact_out="encfc" in kn.fp32_sites() and not os.environ.get("HULC_A3_NOTWIN"))                                                                               # (N, 7, 7, 64) NHWC
        # nn.Flatten + Linear(3136, 128) + ReLU on the NHWC activation in place: the weight's columns are reordered, not the activations
        c = self.conv_model
        with kn.site_scope("encfc"):         # (selective precision)
            y = HF.flatten_linear_relu(a3, c[7].weight, c[7].bias)
            y = HF.mlp2_rows(y, self.fc1[0].weight, self.fc1[0].bias, self.fc2.weight, self.fc2.bias)
        if pre_ln:
            return y
        return HF.layer_norm(y, self.ln.weight, self.ln.bias, self.ln.eps)

    @staticmethod
    def calc_out_size(w: int, h: int, kernel_size: int, padding: int, stride: int) -> Tuple[int, int]:
        return (w - kernel_size + 2 * padding) // stride + 1, (h - kernel_size + 2 * padding) // stride + 1
