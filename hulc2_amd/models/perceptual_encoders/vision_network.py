"""Static-camera encoder on MI355X kernels.

Mirrors hulc2.models.perceptual_encoders.vision_network.VisionNetwork (reference
hulc2/models/perceptual_encoders/vision_network.py:11-108): same constructor kwargs, same state_dict keys
(conv_model.{0,2,4}, fc1.0, fc2, ln, spatial_softmax.{x_map,y_map,temperature}); the nn layers below only
hold parameters — forward runs the HIP conv stack (NHWC), the lane-per-channel spatial softmax, the MFMA
MLP chain and the wavefront LayerNorm from hulc2_amd.functional.
"""
from typing import Optional, Tuple

import torch
import torch.nn as nn

from hulc2_amd import functional as HF
from hulc2_amd import kernels as kn


class SpatialSoftmax(nn.Module):
    """Buffers follow the reference's quirk (vision_network.py:88-92): x_map varies along rows."""

    def __init__(self, num_rows: int, num_cols: int, temperature: Optional[float] = None):
        super().__init__()
        self.num_rows, self.num_cols = num_rows, num_cols
        a = torch.linspace(-1.0, 1.0, num_cols).reshape(-1, 1).expand(num_cols, num_rows)
        b = torch.linspace(-1.0, 1.0, num_rows).reshape(1, -1).expand(num_cols, num_rows)
        self.register_buffer("x_map", a.reshape(-1).clone())
        self.register_buffer("y_map", b.reshape(-1).clone())
        if temperature:
            self.register_buffer("temperature", torch.ones(1) * temperature)
        else:
            self.temperature = nn.Parameter(torch.ones(1))
        self.coords = None

    def forward(self, a_nhwc: torch.Tensor) -> torch.Tensor:
        """a_nhwc: (N, H, W, C) NHWC conv activations -> (N, 2C) interleaved (ex, ey) per channel."""
        if isinstance(self.temperature, nn.Parameter) and self.temperature.requires_grad:
            raise NotImplementedError("learnable spatial-softmax temperature is not on the configured path "
                                      "(conf/model/perceptual_encoder/rgb_static/default.yaml: spatial_softmax_temp 1.0)")
        out = HF.spatial_softmax(a_nhwc, self.x_map, self.y_map, self.temperature)
        self.coords = out.detach()     # kept for visualisation like the reference; detached so no autograd graph outlives the step
        return out


class VisionNetwork(nn.Module):
    def __init__(self, input_width: int, input_height: int, activation_function: str, dropout_vis_fc: float,
                 l2_normalize_output: bool, visual_features: int, num_c: int, use_sinusoid: bool,
                 spatial_softmax_temp: float):
        super().__init__()
        if activation_function != "ReLU" or use_sinusoid or l2_normalize_output or dropout_vis_fc != 0.0:
            raise NotImplementedError("hulc2_amd VisionNetwork implements the configured path only: ReLU, no sinusoid, "
                                      "no l2-normalise, dropout_vis_fc 0 (conf/model/perceptual_encoder/rgb_static/default.yaml)")
        self.act_fn = nn.ReLU()
        w, h = input_width, input_height
        for k, s in ((8, 4), (4, 2), (3, 1)):
            w, h = self.calc_out_size(w, h, k, 0, s)
        temp = spatial_softmax_temp if isinstance(spatial_softmax_temp, float) else None
        self.spatial_softmax = SpatialSoftmax(num_rows=w, num_cols=h, temperature=temp)
        self.conv_model = nn.Sequential(nn.Conv2d(num_c, 32, 8, stride=4), self.act_fn, nn.Conv2d(32, 64, 4, stride=2),
                                        self.act_fn, nn.Conv2d(64, 64, 3, stride=1), self.act_fn)
        self.fc1 = nn.Sequential(nn.Linear(128, 512), self.act_fn, nn.Dropout(dropout_vis_fc))
        self.fc2 = nn.Linear(512, visual_features)
        self.ln = nn.LayerNorm(visual_features)

    def lo_operands(self):
        """rounding remainders the split-operand forward of the fc1 -> fc2 head reads (precision site "encfc"; trainer-maintained)"""
        return [(self.fc1[0].weight, "lo"), (self.fc2.weight, "lo"), (self.conv_model[0].weight, "oihw_flat_lo")]

    def conv_params(self):
        c = self.conv_model
        return (c[0].weight, c[0].bias, c[2].weight, c[2].bias, c[4].weight, c[4].bias)

    def forward(self, x: torch.Tensor, aug_shift=None, aug_pad: int = 0, frame_index=None, pre_ln: bool = False) -> torch.Tensor:
        """x: (N,3,H,W) fp32 in [-1,1] (or a list of such), or uint8 NHWC frames as stored with the shift augmentation parameters"""
        a3 = HF.conv_stack(x, self.conv_params(), grad_premasked=True, aug_pad=aug_pad, aug_shifts=aug_shift, frame_index=frame_index)      # (N, 21, 21, 64) NHWC
        feat = self.spatial_softmax(a3)                                     # (N, 128)
        with kn.site_scope("encfc"):         # (selective precision, DESIGN §5: the fc tail is one of the sites HULC_FP32_SITES can make exact)
            y = HF.mlp2_rows(feat, self.fc1[0].weight, self.fc1[0].bias, self.fc2.weight, self.fc2.bias)
        if pre_ln:                # ConcatEncoders applies the LayerNorms of both cameras while writing the concatenated embedding
            return y
        return HF.layer_norm(y, self.ln.weight, self.ln.bias, self.ln.eps)

    @staticmethod
    def calc_out_size(w: int, h: int, kernel_size: int, padding: int, stride: int) -> Tuple[int, int]:
        return (w - kernel_size + 2 * padding) // stride + 1, (h - kernel_size + 2 * padding) // stride + 1
