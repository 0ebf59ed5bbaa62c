"""CPU ORACLE (test infrastructure only) for the trainable part of the affordance model — SURVEY §8 row f-4, BASELINE configs[4].

NOT part of the product: only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import it.  Plain fp32 torch
functional ops over a flat state_dict with the reference's parameter names.  It restates one training step of
`PixelAffLangDetector` (/root/reference/hulc2/affordance/pixel_aff_lang_detector.py:51-69) in its shipped variant
(conf/affordance/aff_detection/r3m.yaml + conf/affordance/train_affordance.yaml) BEHIND the frozen R3M ResNet-18 trunk:
the trunk is third-party (`r3m`, empty submodule — its forward is `r3m_trunk_features` of oracle/hulc2_oracle.py, parity unpinned),
so the five feature maps it hands to the decoder are inputs here, like the (B, 384) sentence embedding is for SBERT.

Pinning: tests/test_oracle_golden.py::test_affordance_step checks every output and gradient below against
tests/golden/affordance_step_B2_64.npz, produced by the reference's own `UnetLangFusionDecoder`, `FusionMult`,
`cross_entropy_with_logits` and `DepthEstimationGaussian` (oracle/gen_golden.py::gen_affordance).

Deviation from the reference that the product shares: the reference leaves the trunk's stem (conv1, bn1) trainable by
omission (r3m_rn18.py:34-38 freezes layer1..layer4 only) and runs the frozen trunk's BatchNorms in train mode; here the whole trunk is
a frozen feature extractor.
"""
from typing import Dict, List, Sequence

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
DECODER_CHANNELS = (512, 256, 128, 64, 32)          # r3m_rn18.py:54
ENCODER_CHANNELS = (3, 64, 64, 128, 256, 512)       # r3m_rn18.py:59
LOSS_WEIGHTS = {"aff": 0.1, "depth": 0.9}           # conf/affordance/train_affordance.yaml:31-33


def trainable_shapes(enc_hw: int) -> Dict[str, tuple]:
    """names and shapes of the trainable tensors, as the reference's modules register them (state_dict keys of R3M.decoder /
    .segmentation_head, SBertLang.text_fc, DepthEstimationGaussian); enc_hw = side of the trunk's last feature map (7 at 224 x 224)"""
    enc = list(ENCODER_CHANNELS[1:])[::-1]                       # unet_decoder.py:107-108: (512, 256, 128, 64, 64)
    in_ch = [enc[0]] + list(DECODER_CHANNELS[:-1])              # :111-113
    skip_ch = enc[1:] + [0]
    out = {"text_fc.weight": (1024, 384), "text_fc.bias": (1024,)}
    for i, (ci, cs, co) in enumerate(zip(in_ch, skip_ch, DECODER_CHANNELS)):
        b = f"decoder.blocks.{i}."
        out[b + "conv1.0.weight"] = (co, ci + cs, 3, 3)         # Conv2dReLU: bias=False with BatchNorm (unet_decoder.py:16-23)
        out[b + "conv1.1.weight"] = (co,); out[b + "conv1.1.bias"] = (co,)
        out[b + "conv2.0.weight"] = (co, co, 3, 3)
        out[b + "conv2.1.weight"] = (co,); out[b + "conv2.1.bias"] = (co,)
        out[b + "lang_proj.weight"] = (ci, 1024); out[b + "lang_proj.bias"] = (ci,)      # (blocks 3, 4 own one but never use it: :119-128)
    out["segmentation_head.weight"] = (1, DECODER_CHANNELS[-1], 3, 3); out["segmentation_head.bias"] = (1,)
    lin = 512 * enc_hw * enc_hw
    for name, (o, i) in {"fc1": (768, lin + 1024), "fc2": (512, 768 + 1024), "fc3": (256, 512), "depth_mu": (1, 256), "depth_sigma": (1, 256)}.items():
        out[f"depth_stream.{name}.weight"] = (o, i); out[f"depth_stream.{name}.bias"] = (o,)      # depth_gaussian.py:56-65
    return out


def trunk_maps(sd: SD, img: torch.Tensor, p: str = "r3m.convnet.", stages=(2, 2, 2, 2), eps: float = 1e-5, bn_train: bool = False,
               momentum: float = 0.1) -> List[torch.Tensor]:
    """The five maps R3M.r3m_resnet18 hands to the decoder (visual_lang_encoders/r3m_rn18.py:71-76: the ResNet-18's children applied one after
    the other — no / 255 or ImageNet normalisation here, the dataset transforms did that): [stem (conv1, bn1, relu, maxpool), layer1 .. layer4],
    NCHW, BatchNorm on its running statistics (frozen trunk, module docstring).  PARITY UNPINNED like hulc2_oracle.r3m_trunk_features, whose
    restatement of torchvision's resnet18 this repeats with the maps kept; used by bench.py's cpu_baseline leg and the tests.
    bn_train: the trunk AS THE REFERENCE RUNS IT during training — r3m_rn18.py:34-38 freezes the parameters of layer1..layer4, nothing puts the
    ResNet into eval mode and pixel_aff_lang_detector.py:51-53 leaves Lightning's train() on, so every BatchNorm2d normalises with the batch
    statistics and updates sd[... running_mean / running_var / num_batches_tracked] IN PLACE (pass copies).  Pinned by
    tests/golden/r3m_trunk_trainmode.npz (torch's own nn layers in train mode, oracle/gen_golden.py::gen_r3m_trunk_trainmode)."""
    def bn(t, q):
        if bn_train:
            if q + ".num_batches_tracked" in sd:
                sd[q + ".num_batches_tracked"] += 1
            return F.batch_norm(t, sd[q + ".running_mean"], sd[q + ".running_var"], sd[q + ".weight"], sd[q + ".bias"], True, momentum, eps)
        return F.batch_norm(t, sd[q + ".running_mean"], sd[q + ".running_var"], sd[q + ".weight"], sd[q + ".bias"], False, 0.0, eps)

    t = F.relu(bn(F.conv2d(img, sd[p + "conv1.weight"], None, 2, 3), p + "bn1"))
    t = F.max_pool2d(t, 3, 2, 1)
    maps = [t]
    for li, n in enumerate(stages, start=1):
        for b in range(n):
            q = p + f"layer{li}.{b}."
            stride = 2 if (b == 0 and li > 1) else 1
            idn = t
            if q + "downsample.0.weight" in sd:
                idn = bn(F.conv2d(t, sd[q + "downsample.0.weight"], None, stride, 0), q + "downsample.1")
            o = F.relu(bn(F.conv2d(t, sd[q + "conv1.weight"], None, stride, 1), q + "bn1"))
            o = bn(F.conv2d(o, sd[q + "conv2.weight"], None, 1, 1), q + "bn2")
            t = F.relu(o + idn)
        maps.append(t)
    return maps


def conv_bn_relu(x, w, gamma, beta, train: bool, stats=None, running=None):
    """Conv2dReLU = Conv2d(3x3, padding 1, no bias) -> BatchNorm2d -> ReLU (unet_decoder.py:6-28); batch statistics in train mode (appended to
    `stats` as (mean, unbiased variance) for the running-statistics bookkeeping), `running` = (mean, var) in eval mode"""
    y = F.conv2d(x, w, None, padding=1)
    if train:
        mean = y.mean(dim=(0, 2, 3))
        var = y.var(dim=(0, 2, 3), unbiased=False)
        if stats is not None:
            stats.append((mean.detach(), y.var(dim=(0, 2, 3), unbiased=True).detach()))
    else:
        mean, var = running
    y = (y - mean[None, :, None, None]) * torch.rsqrt(var[None, :, None, None] + 1e-5) * gamma[None, :, None, None] + beta[None, :, None, None]
    return F.relu(y)


def decoder_forward(sd: SD, l_enc: torch.Tensor, feats: Sequence[torch.Tensor], out_hw: int, train: bool = True, stats=None, running=None) -> torch.Tensor:
    """UnetLangFusionDecoder.forward (unet_decoder.py:131-146) with DecoderBlock.forward (:60-80): blocks 0..2 multiply the incoming map
    by lang_proj(l_enc) per channel (FusionMult + tile_x2, fusion.py:40-47,64-73), every block up-samples (nearest) to its skip's size —
    block 3's skip (the stem map) has the same size, block 4 has no skip and goes to the input resolution — concatenates [x, skip] and runs
    two Conv2dReLU."""
    rev = list(feats)[::-1]                                         # layer4, layer3, layer2, layer1, stem
    x, skips = rev[0], rev[1:]
    for i in range(len(DECODER_CHANNELS)):
        b = f"decoder.blocks.{i}."
        if i < 3:
            x = x * F.linear(l_enc, sd[b + "lang_proj.weight"], sd[b + "lang_proj.bias"])[:, :, None, None]
        skip = skips[i] if i < len(skips) else None
        scale = (skip.shape[-1] if skip is not None else out_hw) // x.shape[-1]
        x = F.interpolate(x, scale_factor=scale, mode="nearest")
        if skip is not None:
            x = torch.cat([x, skip], dim=1)
        x = conv_bn_relu(x, sd[b + "conv1.0.weight"], sd[b + "conv1.1.weight"], sd[b + "conv1.1.bias"], train, stats, running[2 * i] if running else None)
        x = conv_bn_relu(x, sd[b + "conv2.0.weight"], sd[b + "conv2.1.weight"], sd[b + "conv2.1.bias"], train, stats, running[2 * i + 1] if running else None)
    return x


def depth_forward(sd: SD, f4: torch.Tensor, l_enc: torch.Tensor):
    """DepthEstimationGaussian.forward (depth_gaussian.py:77-102): flatten (C, H, W), two language-conditioned layers, mu and
    sigma = exp(clamp(log_sigma, -20, 2))"""
    p = "depth_stream."
    x = torch.cat([f4.reshape(f4.shape[0], -1), l_enc], -1)
    x = F.relu(F.linear(x, sd[p + "fc1.weight"], sd[p + "fc1.bias"]))
    x = torch.cat([x, l_enc], -1)
    x = F.relu(F.linear(x, sd[p + "fc2.weight"], sd[p + "fc2.bias"]))
    x = F.relu(F.linear(x, sd[p + "fc3.weight"], sd[p + "fc3.bias"]))
    mu = F.linear(x, sd[p + "depth_mu.weight"], sd[p + "depth_mu.bias"])
    sigma = torch.clamp(F.linear(x, sd[p + "depth_sigma.weight"], sd[p + "depth_sigma.bias"]), -20, 2).exp()
    return mu, sigma


def gaussian_nll(mu, target, var, eps: float = 1e-6):
    """nn.GaussianNLLLoss()(mu, target, var) as DepthEstimationGaussian.loss calls it (depth_gaussian.py:67-69: sigma is passed as the
    variance): mean of 0.5 (log(max(var, eps)) + (mu - target)^2 / max(var, eps))"""
    var = torch.clamp(var, min=eps)
    return (0.5 * (torch.log(var) + (mu - target) ** 2 / var)).mean()


def training_step(sd: SD, feats: Sequence[torch.Tensor], emb: torch.Tensor, p0: torch.Tensor, gt_depth: torch.Tensor, out_hw: int, train: bool = True,
                  stats=None, running=None) -> Dict[str, torch.Tensor]:
    """PixelAffLangDetector.training_step -> forward(softmax=False) -> criterion (pixel_aff_lang_detector.py:51-69,116-171) for a square
    input (AffDepthLangFusionPixel pads to a square and crops back: no-ops at 224 x 224, aff_lang_depth_pixel.py:17-30,112-115).
    feats = trunk maps (stem, layer1 .. layer4) NCHW; emb = SBERT sentence embeddings (B, 384); p0 (B, 2) = (row, col) of the labelled pixel.
    train False + running = [(mean, var)] x 10: the inference forward of AffDepthLangFusionPixel.predict (aff_lang_depth_pixel.py:64-96)."""
    B = emb.shape[0]
    l_enc = F.linear(emb, sd["text_fc.weight"], sd["text_fc.bias"])                      # SBertLang.encode_text (sbert_lang_encoder.py:26-29)
    dec = decoder_forward(sd, l_enc, feats, out_hw, train, stats, running)
    aff = F.conv2d(dec, sd["segmentation_head.weight"], sd["segmentation_head.bias"], padding=1)     # r3m_rn18.py:64-69,88
    logits = aff.permute(0, 2, 3, 1).reshape(B, -1)                                     # aff_lang_depth_pixel.py:117-118
    mu, sigma = depth_forward(sd, feats[-1], l_enc)
    logp = F.log_softmax(logits, -1)
    idx = p0[:, 0].long() * out_hw + p0[:, 1].long()
    aff_loss = -(logp[torch.arange(B), idx]).sum() / logits.numel()                    # cross_entropy_with_logits, reduction "mean" over B x H x W (losses.py:6-11)
    depth_loss = gaussian_nll(mu, gt_depth.reshape(B, 1), sigma)
    loss = LOSS_WEIGHTS["aff"] * aff_loss + LOSS_WEIGHTS["depth"] * depth_loss          # pixel_aff_lang_detector.py:165-166
    return {"loss": loss, "aff_loss": aff_loss, "depth_loss": depth_loss, "logits": logits, "mu": mu, "sigma": sigma, "dec": dec, "l_enc": l_enc}
