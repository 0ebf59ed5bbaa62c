"""`PixelAffLangDetector` (hulc2/affordance/pixel_aff_lang_detector.py) in the shipped variant of conf/affordance/train_affordance.yaml:
aff_detection = r3m (frozen R3M ResNet-18 trunk, SBERT sentence embedding -> text_fc, language-fused U-Net decoder with BatchNorm, one-channel
segmentation head), depth_dist = gaussian (DepthEstimationGaussian on the trunk's last map), loss = 0.1 cross-entropy over the pixels + 0.9
Gaussian NLL, Adam(lr 1e-4).

Module tree and parameter names follow the reference's state_dict: `model.lang_encoder.text_fc`, `model.aff_stream.decoder.blocks.{i}.
{conv1,conv2}.{0,1}`, `.lang_proj`, `model.aff_stream.segmentation_head`, `model.depth_stream.{fc1,fc2,fc3,depth_mu,depth_sigma}`,
`model.aff_stream.r3m.convnet.*` (the reference additionally registers the trunk's stages a second time as `stem` / `layer1..4`: aliases of
the same tensors, not repeated here).

Trunk modes (`trunk_mode`): "frozen" (default) — the WHOLE trunk is a frozen inference-mode feature extractor (BatchNorm folded into the
convolutions).  "reference" (round 4) — the BatchNorms of the trunk as the reference runs them: r3m_rn18.py:34-38 freezes the PARAMETERS of
layer1..layer4 only and pixel_aff_lang_detector.py:51-53 leaves Lightning's train() on, so in training every BatchNorm2d of the ResNet
normalises with the statistics of the batch and updates its running statistics (eval mode: running statistics, = "frozen"), and (round 5) the
STEM trains: the reference's freeze leaves `conv1.weight`, `bn1.weight`, `bn1.bias` trainable by omission, their gradient is the data gradient
through the whole frozen ResNet (affordance/trunk.py: one autograd node, hulc_nhwc_bn_train_bwd / zero-inserted data-gradient convolutions /
hulc_maxpool_nhwc_bwd / a 7 x 7 weight gradient).  "frozen" stays what it says: no trunk parameter receives a gradient.  Third-party arithmetic (r3m weights, sentence-transformers) is absent: parity of
the trunk and of SBERT is unpinned; everything behind them is pinned on the reference's own modules."""
import os
from typing import Dict, List, Optional, Sequence, Tuple, Union

import torch
import torch.nn as nn

from .. import functional as HF, kernels as kn
from ..models.language_encoders.sbert_lang_encoder import SBertLang
from ..models.perceptual_encoders.vision_r3m import R3M
from .functional import DECODER_CHANNELS, AffDecoderLossFn, DepthNllFn, block_channels

LOSS_WEIGHTS = {"aff": 0.1, "depth": 0.9}                  # conf/affordance/train_affordance.yaml:31-33


class Conv2dReLU(nn.Sequential):
    """unet_decoder.py:6-28: Conv2d(3 x 3, padding 1, no bias) -> BatchNorm2d -> ReLU; only a parameter container here"""

    def __init__(self, cin: int, cout: int):
        super().__init__(nn.Conv2d(cin, cout, 3, padding=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class DecoderBlock(nn.Module):
    def __init__(self, cin: int, cskip: int, cout: int, lang_embed_dim: int = 1024):
        super().__init__()
        self.conv1 = Conv2dReLU(cin + cskip, cout)
        self.conv2 = Conv2dReLU(cout, cout)
        self.lang_proj = nn.Linear(lang_embed_dim, cin)           # (blocks 3 and 4 own one the forward never uses: unet_decoder.py:119-128)


class UnetLangFusionDecoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.blocks = nn.ModuleList([DecoderBlock(ci, cs, co) for ci, cs, co in block_channels()])


class R3MLingunet(nn.Module):
    """visual_lang_encoders/r3m_rn18.py: the trunk, the decoder, the segmentation head"""

    def __init__(self):
        super().__init__()
        self.r3m = R3M("resnet18")
        for p in self.r3m.parameters():
            p.requires_grad = False
        self.decoder = UnetLangFusionDecoder()
        self.segmentation_head = nn.Conv2d(DECODER_CHANNELS[-1], 1, 3, padding=1)


class DepthEstimationGaussian(nn.Module):
    """models/depth/depth_gaussian.py:56-65: parameters only; the forward is `AffDepthLangFusionPixel.depth`"""

    def __init__(self, enc_hw: int, normalized: bool = True):
        super().__init__()
        lin = 512 * enc_hw * enc_hw
        self.fc1 = nn.Linear(lin + 1024, 768)
        self.fc2 = nn.Linear(768 + 1024, 512)
        self.fc3 = nn.Linear(512, 256)
        self.depth_mu = nn.Linear(256, 1)
        self.depth_sigma = nn.Linear(256, 1)
        self.normalized = normalized


class AffDepthLangFusionPixel(nn.Module):
    def __init__(self, img_size: int, sbert: Optional[str]):
        super().__init__()
        self.lang_encoder = SBertLang(sbert) if sbert else _TextFC()
        self.aff_stream = R3MLingunet()
        self.depth_stream = DepthEstimationGaussian(img_size // 32)


class _TextFC(nn.Module):
    """the trainable part of SBertLang when sentence embeddings arrive precomputed ((B, 384) tensors instead of strings)"""

    def __init__(self):
        super().__init__()
        self.text_fc = nn.Linear(384, 1024)


class PixelAffLangDetector(nn.Module):
    """training_step((frame, label), batch_idx) -> loss.  frame["img"]: (B, 3, S, S) fp32, transforms applied (conf/affordance/transforms/
    r3m.yaml), S = 224; frame["lang_goal"]: list of B strings, or the (B, 384) sentence embeddings; label["p0"]: (B, 2) (row, col);
    label["normalized_depth"] (or "depth" with normalize_depth False): (B,)."""

    def __init__(self, img_size: int = 224, normalize_depth: bool = True, loss_weights: Optional[Dict[str, float]] = None, lr: float = 1e-4,
                 sbert: Optional[str] = None, trunk_mode: str = "frozen"):
        super().__init__()
        if img_size % 32:
            raise ValueError("img_size must be a multiple of 32 (ResNet-18 trunk)")
        if trunk_mode not in ("frozen", "reference"):
            raise ValueError("trunk_mode: 'frozen' (inference-mode trunk) or 'reference' (BatchNorm on batch statistics while training)")
        self.trunk_mode = trunk_mode
        self.img_size, self.normalize_depth, self.lr = img_size, normalize_depth, lr
        self.loss_weights = dict(loss_weights or LOSS_WEIGHTS)
        self.model = AffDepthLangFusionPixel(img_size, sbert)
        if trunk_mode == "reference":                  # r3m_rn18.py:34-38 freezes layer1..layer4 only: the stem's three tensors train
            net = self.model.aff_stream.r3m.convnet
            for p in (net.conv1.weight, net.bn1.weight, net.bn1.bias):
                p.requires_grad = True
        self.logged: Dict[str, torch.Tensor] = {}

    # ---- pieces ---------------------------------------------------------------------------------------------------------------------
    def trunk_maps(self, img: torch.Tensor) -> List[torch.Tensor]:
        """R3M.r3m_resnet18 (r3m_rn18.py:71-76): the stem's and the four stages' outputs, NHWC bf16 — from the folded inference-mode trunk, or
        (trunk_mode "reference", training) with every BatchNorm on the statistics of the batch and the stem's three tensors trainable"""
        from ..models.perceptual_encoders.vision_r3m import _trunk_of, trunk_feature_maps
        r3m = self.model.aff_stream.r3m
        net = r3m.convnet
        if (self.trunk_mode == "reference" and self.training and torch.is_grad_enabled() and net.conv1.weight.requires_grad
                and not os.environ.get("HULC_AFF_FROZEN_STEM")):
            from .trunk import TrunkStemFn
            return list(TrunkStemFn.apply(img, net.conv1.weight, net.bn1.weight, net.bn1.bias, _trunk_of(r3m)))
        with torch.no_grad():
            return trunk_feature_maps(r3m, img, batch_stats=self.trunk_mode == "reference" and self.training)

    def text_enc(self, lang_goal) -> torch.Tensor:
        le = self.model.lang_encoder
        emb = lang_goal if torch.is_tensor(lang_goal) else le.encode(list(lang_goal))
        return HF.mlp(emb.float(), [(le.text_fc.weight, le.text_fc.bias, False)])                  # SBertLang.encode_text (:26-29)

    def decoder_params(self) -> List[torch.Tensor]:
        ps = []
        for b in self.model.aff_stream.decoder.blocks:
            ps += [b.conv1[0].weight, b.conv1[1].weight, b.conv1[1].bias, b.conv2[0].weight, b.conv2[1].weight, b.conv2[1].bias]
        head = self.model.aff_stream.segmentation_head
        return ps + [head.weight, head.bias]

    def bn_buffers(self) -> List[torch.Tensor]:
        bufs = []
        for b in self.model.aff_stream.decoder.blocks:
            bufs += [b.conv1[1].running_mean, b.conv1[1].running_var, b.conv2[1].running_mean, b.conv2[1].running_var]
        return bufs

    def bn_step_counters(self) -> List[torch.Tensor]:
        out = []
        for b in self.model.aff_stream.decoder.blocks:
            out += [b.conv1[1].num_batches_tracked, b.conv2[1].num_batches_tracked]
        return out

    def depth_features(self, f4: torch.Tensor, l_enc: torch.Tensor) -> torch.Tensor:
        """DepthEstimationGaussian.forward up to fc3 + ReLU (depth_gaussian.py:77-93); f4 NHWC -> the reference flattens (C, H, W)"""
        d = self.model.depth_stream
        x = torch.cat([f4.permute(0, 3, 1, 2).reshape(f4.shape[0], -1).float(), l_enc], -1)
        x = torch.relu(HF.mlp(x, [(d.fc1.weight, d.fc1.bias, False)]))          # (an MLPFn chain ends with a plain Linear)
        x = torch.cat([x, l_enc], -1)
        return torch.relu(HF.mlp(x, [(d.fc2.weight, d.fc2.bias, True), (d.fc3.weight, d.fc3.bias, False)]))

    # ---- step -----------------------------------------------------------------------------------------------------------------------
    def forward_losses(self, feats: Sequence[torch.Tensor], lang_goal, p0: torch.Tensor, gt_depth: torch.Tensor):
        """everything behind the trunk: feats = trunk maps NHWC bf16 (stem, layer1 .. layer4)"""
        l_enc = self.text_enc(lang_goal)
        blocks = self.model.aff_stream.decoder.blocks
        gs = [HF.mlp(l_enc, [(blocks[i].lang_proj.weight, blocks[i].lang_proj.bias, False)]) for i in range(3)]        # FusionMult's x2_proj(l)
        bufs = self.bn_buffers() if self.training else ("eval", self.bn_buffers())
        aff_loss, logits = AffDecoderLossFn.apply(p0, self.img_size, bufs, *gs, *feats, *self.decoder_params())
        d = self.model.depth_stream
        # depth_mu / depth_sigma, sigma = exp(clamp(., -20, 2)) and nn.GaussianNLLLoss(mu, target, var = sigma) (eps 1e-6): one kernel
        depth_loss, mu, sigma = DepthNllFn.apply(self.depth_features(feats[-1], l_enc), d.depth_mu.weight, d.depth_mu.bias, d.depth_sigma.weight,
                                                 d.depth_sigma.bias, gt_depth)
        loss = self.loss_weights["aff"] * aff_loss + self.loss_weights["depth"] * depth_loss       # pixel_aff_lang_detector.py:165-166
        return loss, {"aff_loss": aff_loss, "depth_loss": depth_loss, "logits": logits, "mu": mu, "sigma": sigma}

    def training_step(self, batch, batch_idx: int = 0) -> torch.Tensor:
        frame, label = batch
        kn.Grid.begin_step()                                   # the previous step's activation / gradient grids are free again
        feats = self.trunk_maps(frame["img"])
        depth_key = "normalized_depth" if self.normalize_depth else "depth"
        loss, info = self.forward_losses(feats, frame["lang_goal"], label["p0"], label[depth_key])
        if self.training:                                      # nn.BatchNorm2d bumps num_batches_tracked on every training forward: state_dicts stay
            torch._foreach_add_(self.bn_step_counters(), 1)   # interchangeable with the reference's (one multi-tensor launch for the ten counters)
        self.logged = {"Training/total_loss": loss.detach(), "Training/aff_loss": info["aff_loss"].detach(), "Training/depth_loss": info["depth_loss"].detach()}
        return loss

    # ---- inference (AffDepthLangFusionPixel.forward / predict, aff_lang_depth_pixel.py:64-129; PixelAffLangDetector.validation_step :71-93) ----
    @torch.no_grad()
    def forward(self, inp: Dict, softmax: bool = True) -> Dict[str, torch.Tensor]:
        """{"aff": (B, H, W, 1) softmax over the pixels — or the (B, H W) logits —, "depth_dist": (mu, sigma)}; BatchNorm by the running
        statistics when the module is in eval mode"""
        B = inp["img"].shape[0]
        kn.Grid.begin_step()                                   # inference between training steps: recycle the grid buffers here too (else the pool grows)
        feats = self.trunk_maps(inp["img"])
        p0 = torch.zeros(B, 2, dtype=torch.int32, device=inp["img"].device)
        _, info = self.forward_losses(feats, inp["lang_goal"], p0, torch.zeros(B, device=inp["img"].device))
        aff = info["logits"]
        if softmax:
            aff = torch.softmax(aff, -1).reshape(B, self.img_size, self.img_size, 1)
        return {"aff": aff, "depth_dist": (info["mu"], info["sigma"])}

    @torch.no_grad()
    def predict_pixels(self, img: torch.Tensor, lang_goal, depth_norm: Tuple[float, float] = (0.0, 1.0), sample: bool = False):
        """-> (p0 (B, 2) int64 (row, col) of the most likely pixel, depth (B,) in metres, sigma (B,)): the arg-max of the heat map and the mean
        of the depth distribution (sample=True: one reparametrised draw, as DepthEstimationGaussian.sample), un-normalised by
        NormalizeVectorInverse(mean, std) (datasets/transforms.py:82-93) when the model predicts normalised depth"""
        out = self.forward({"img": img, "lang_goal": lang_goal}, softmax=False)
        idx = out["aff"].argmax(-1)
        p0 = torch.stack([idx // self.img_size, idx % self.img_size], 1)
        mu, sigma = out["depth_dist"]
        d = (mu + sigma * torch.randn_like(mu) if sample else mu).reshape(-1)
        if self.normalize_depth:
            mean, std = depth_norm
            d = d * (std + 1e-10) + mean              # inverse of (x - mean) / std as NormalizeVectorInverse builds it
        return p0, d, sigma.reshape(-1)

    @torch.no_grad()
    def validation_step(self, batch, batch_idx: int = 0, depth_norm: Tuple[float, float] = (0.0, 1.0)) -> Dict[str, torch.Tensor]:
        """losses as in training plus the two errors the reference logs: summed pixel distance of the arg-max to the label and summed
        absolute depth error (pixel_aff_lang_detector.py:147-160)"""
        frame, label = batch
        kn.Grid.begin_step()
        feats = self.trunk_maps(frame["img"])
        depth_key = "normalized_depth" if self.normalize_depth else "depth"
        loss, info = self.forward_losses(feats, frame["lang_goal"], label["p0"], label[depth_key])
        idx = info["logits"].argmax(-1)
        p0 = torch.stack([idx // self.img_size, idx % self.img_size], 1).float()
        px = (p0 - label["p0"].to(p0.device).float()).norm(dim=1).sum()
        d = info["mu"].reshape(-1)
        if self.normalize_depth:
            d = d * (depth_norm[1] + 1e-10) + depth_norm[0]
        true_depth = label["depth"] if "depth" in label else label[depth_key]
        return {"val_loss": loss, "val_attn_dist_err": px, "val_depth_err": (d - true_depth.to(d.device).float()).abs().sum(), "n_imgs": torch.tensor(p0.shape[0])}

    def configure_optimizers(self):
        return torch.optim.Adam([p for p in self.parameters() if p.requires_grad], lr=self.lr)      # pixel_aff_lang_detector.py:112-114
