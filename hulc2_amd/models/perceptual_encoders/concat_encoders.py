"""ConcatEncoders — per-camera encoders concatenated on the feature axis.

Mirrors hulc2.models.perceptual_encoders.concat_encoders.ConcatEncoders (reference concat_encoders.py:11-109)
for the configured cameras: rgb_static + rgb_gripper, proprio/depth/tactile: none.
"""
from typing import Dict, Optional

import os

import torch
import torch.nn as nn

from hulc2_amd.compat import instantiate


_side_streams = {}


def _encoder_side_stream(device):
    st = _side_streams.get(device)
    if st is None:
        from hulc2_amd import kernels as kn
        st = _side_streams[device] = kn.capture_stream(device)       # (none of the other cached streams: kernels.capture_stream)
    return st


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
        # concat_encoders.py:34-37: the pretrained trunks are handed the device
        needs_device = any(t in rgb_static["_target_"] for t in ("clip", "r3m"))
        self.rgb_static_encoder = instantiate(rgb_static, device=device) if needs_device else instantiate(rgb_static)
        self.depth_static_encoder = None
        self.rgb_gripper_encoder = instantiate(rgb_gripper) if rgb_gripper else None
        self.depth_gripper_encoder = None
        self.tactile_encoder = None
        self.proprio_encoder = None
        self.state_decoder = None
        self.current_visual_embedding = None
        self.current_state_obs = None
        # RandomShiftsAug pads of conf/datamodule/transforms/rand_shift.yaml:5,12 — used only when uint8 frames arrive together with
        # per-frame shifts (the stored uint8 NHWC frames go straight into conv1)
        self.aug_pad = {"rgb_static": 10, "rgb_gripper": 4}

    @staticmethod
    def _frames(imgs, key):
        """-> (frames flattened over (B, S), shifts flattened or None, store frame numbers or None, B, S).  fp32 (B,S,3,H,W) in [-1,1]
        as the reference's transforms deliver them, or uint8 (B,S,H,W,3) as stored (+ optional imgs[key + "_shift"] (B,S,2) int32
        {sx, sy}), or the whole uint8 episode store (n_frames,H,W,3) with imgs[key + "_index"] (B,S) int32 naming the window frames
        (hulc2_amd.datasets.DeviceEpisodeStore: nothing is gathered, conv1 reads the store in place)."""
        x = imgs[key]
        ix = imgs.get(key + "_index")
        if ix is not None:
            b, s = ix.shape
            sh = imgs.get(key + "_shift")
            return x, (None if sh is None else sh.reshape(b * s, 2)), ix.reshape(b * s), b, s
        b, s = x.shape[0], x.shape[1]
        if x.dtype == torch.uint8:
            sh = imgs.get(key + "_shift")
            return x.reshape(b * s, *x.shape[2:]), (None if sh is None else sh.reshape(b * s, 2)), None, b, s
        return x.reshape(b * s, *x.shape[2:]), None, None, b, s

    @property
    def latent_size(self):
        return self._latent_size

    def forward(self, imgs: Dict[str, torch.Tensor], depth_imgs: Dict[str, torch.Tensor], state_obs: torch.Tensor) -> torch.Tensor:
        if isinstance(imgs, (list, tuple)):             # several modalities at once (Hulc2.training_step): see forward_multi
            return self.forward_multi(imgs)
        x, sh, ix, b, s = self._frames(imgs, "rgb_static")
        enc = self.rgb_static_encoder(x, sh, self.aug_pad["rgb_static"], ix).reshape(b, s, -1)
        if "rgb_gripper" in imgs and self.rgb_gripper_encoder is not None:
            x, sh, ix, b, s = self._frames(imgs, "rgb_gripper")
            enc = torch.cat([enc, self.rgb_gripper_encoder(x, sh, self.aug_pad["rgb_gripper"], ix).reshape(b, s, -1)], dim=-1)
        self.current_visual_embedding = enc.detach()   # detached: holding the graph across steps breaks HIP-graph capture
        self.current_state_obs = state_obs
        return enc

    def forward_multi(self, imgs_list) -> torch.Tensor:
        """Several observation dicts of identical shapes (the modalities of one training step) through the shared encoders
        in one pass: rows of the result are modality-major, (sum B, S, latent).  Same arithmetic per frame as `forward`."""
        fr = [self._frames(im, "rgb_static") for im in imgs_list]
        b, s = fr[0][3], fr[0][4]
        has_gripper = self.rgb_gripper_encoder is not None and all("rgb_gripper" in im for im in imgs_list)
        frg = [self._frames(im, "rgb_gripper") for im in imgs_list] if has_gripper else None
        # both cameras' encoders stop in front of their final LayerNorm when they support it: the two LayerNorms then write the halves of the
        # embedding directly (functional.LayerNormCatFn) — no concatenation copy forward, no strided-gradient copies backward
        import inspect
        fuse_ln = has_gripper and all(hasattr(e, "ln") and "pre_ln" in inspect.signature(e.forward).parameters
                                      for e in (self.rgb_static_encoder, self.rgb_gripper_encoder)) and fr[0][0].is_cuda
        kw = {"pre_ln": True} if fuse_ln else {}
        run_g = lambda: self.rgb_gripper_encoder([f[0] for f in frg], [f[1] for f in frg], self.aug_pad["rgb_gripper"],
                                                 [f[2] for f in frg], **kw).reshape(len(frg) * b, s, -1)
        # The two cameras' encoders are independent chains; the gripper's launches are small (84 x 84 frames: 20 x 20 / 9 x 9 maps, a few hundred
        # workgroups) and leave most of the chip idle.  On a second stream they fill the tails of the static camera's launches — forward here,
        # and backward too (autograd runs a node on the stream of its forward).  The streams join before the embedding is used, so the
        # device-wide-barrier kernels further down (recurrent sweep, MLP chains) still have the GPU to themselves.  HULC_ENC_STREAMS=0: one stream.
        two = has_gripper and fr[0][0].is_cuda and os.environ.get("HULC_ENC_STREAMS", "1") != "0"
        if two:
            cur = torch.cuda.current_stream(fr[0][0].device)
            side = _encoder_side_stream(fr[0][0].device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                g = run_g()
        enc = self.rgb_static_encoder([f[0] for f in fr], [f[1] for f in fr], self.aug_pad["rgb_static"], [f[2] for f in fr], **kw).reshape(len(fr) * b, s, -1)
        if two:
            cur.wait_stream(side)
            g.record_stream(cur)
        elif has_gripper:
            g = run_g()
        if fuse_ln:
            from hulc2_amd import functional as HF
            enc = HF.layer_norm_cat([enc, g], [self.rgb_static_encoder.ln, self.rgb_gripper_encoder.ln], dim=-1).reshape(len(fr) * b, s, -1)
        elif has_gripper:
            enc = torch.cat([enc, g], dim=-1)
        self.current_visual_embedding = enc.detach()
        return enc
