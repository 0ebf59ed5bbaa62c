"""Static-camera encoder with a frozen R3M ResNet trunk on MI355X kernels.

Mirrors hulc2.models.perceptual_encoders.vision_r3m.VisionR3M (reference hulc2/models/perceptual_encoders/vision_r3m.py:8-32): same
constructor kwargs, same state_dict keys (`r3m.convnet.*` named as torchvision names a ResNet, `fc1.*`, `fc2.*`), frames in [0, 255]
(conf/datamodule/transforms/real_world_r3m.yaml:2-13), trunk under no_grad, two trainable linear layers behind it.

What the trunk computes comes from `r3m`, an un-vendored submodule with downloaded weights (parity unpinned).  Restated from
its public definition: obs / 255 -> Normalize(ImageNet mean, std) -> torchvision resnet18 (or 34) with fc = Identity, i.e. a (N, 512)
feature; `self.r3m(x)` is called with the default obs_shape, so there is no resize / centre crop.  Without network access the constructor
cannot download the weights: the trunk starts from torchvision's initialisation and takes its values from a checkpoint
(`load_state_dict`) or from an r3m `model.pt` (`load_r3m_checkpoint`, or the HULC2_R3M_WEIGHTS environment variable).

The trunk is frozen: BatchNorm uses its running statistics, folded into the convolution weights (scale) and bias (shift) once per
parameter version, so a layer is one `hulc_conv2d_padded_fwd` launch with ReLU and the residual add in its epilogue.  (The reference leaves
the trunk's mode to Lightning's `.train()` call, which would make BatchNorm normalise with batch statistics during training — a side
effect of not calling `.eval()`, not something a frozen backbone is meant to do; this module keeps the frozen semantics in both modes.)
"""
import os
from typing import List, Optional, Sequence, Union

import torch
import torch.nn as nn

from hulc2_amd import functional as HF
from hulc2_amd import kernels as kn
from hulc2_amd import shadow

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
_STAGES = {"resnet18": (2, 2, 2, 2), "resnet34": (3, 4, 6, 3)}


class BasicBlock(nn.Module):
    """Parameter holder named like torchvision.models.resnet.BasicBlock (conv1, bn1, conv2, bn2, downsample.{0,1})."""

    def __init__(self, cin: int, cout: int, stride: int):
        super().__init__()
        self.stride = stride
        self.conv1 = nn.Conv2d(cin, cout, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride, bias=False), nn.BatchNorm2d(cout))


class ResNetTrunk(nn.Module):
    """torchvision resnet18 / resnet34 without the classifier (r3m sets convnet.fc = Identity): parameters only."""

    def __init__(self, name: str = "resnet18"):
        super().__init__()
        if name not in _STAGES:
            raise NotImplementedError(f"VisionR3M trunk {name!r}: only the BasicBlock ResNets (resnet18, resnet34) are built "
                                      "(conf/model/perceptual_encoder/rgb_static/r3m.yaml: resnet18)")
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for li, (n, cout) in enumerate(zip(_STAGES[name], (64, 128, 256, 512)), start=1):
            blocks = []
            for b in range(n):
                blocks.append(BasicBlock(cin, cout, 2 if (b == 0 and li > 1) else 1))
                cin = cout
            setattr(self, f"layer{li}", nn.Sequential(*blocks))
        for m in self.modules():            # torchvision's initialisation
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        self.out_features = 512

    def blocks(self) -> List[BasicBlock]:
        return [b for li in range(1, 5) for b in getattr(self, f"layer{li}")]


class R3M(nn.Module):
    """`load_r3m(...).module` as the reference holds it: the only parameters are the trunk's (the language heads are stripped when r3m
    loads a checkpoint with langweight = 0)."""

    def __init__(self, name: str):
        super().__init__()
        self.convnet = ResNetTrunk(name)
        self.outdim = self.convnet.out_features


def _fold(conv: nn.Conv2d, bn: nn.BatchNorm2d, cin_pad: int, wdtype: torch.dtype):
    """conv + frozen BatchNorm -> (OHWI weight [Cout][KH*KW*Cin'] in wdtype, fp32 bias)."""
    scale = bn.weight.float() * torch.rsqrt(bn.running_var.float() + bn.eps)
    w = conv.weight.float() * scale[:, None, None, None]                       # (O, I, KH, KW)
    if cin_pad > w.shape[1]:
        w = torch.cat([w, w.new_zeros(w.shape[0], cin_pad - w.shape[1], *w.shape[2:])], dim=1)
    w = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous().to(wdtype)
    return w, (bn.bias.float() - bn.running_mean.float() * scale).contiguous()


class VisionR3M(nn.Module):
    def __init__(self, device: Union[str, torch.device, None], visual_features: int, resnet_model: str = "resnet18",
                 freeze_backbone: bool = True):
        super().__init__()
        self.r3m = R3M(resnet_model)
        for p in self.r3m.parameters():
            p.requires_grad = False
        if not freeze_backbone:
            # vision_r3m.py:19-22 re-enables requires_grad on layer4, but the trunk still runs under no_grad (:25-26), so layer4 never
            # receives a gradient; the flags are mirrored for optimizers that enumerate parameters.
            for p in self.r3m.convnet.layer4.parameters():
                p.requires_grad = True
        self.fc1 = nn.Linear(self.r3m.outdim, 256)
        self.fc2 = nn.Linear(256, visual_features)
        self._folded = None
        self._folded_key = None
        path = os.environ.get("HULC2_R3M_WEIGHTS")
        if path:
            self.load_r3m_checkpoint(path)

    # ---- weights -------------------------------------------------------------------------------------------------------------
    def load_r3m_checkpoint(self, path: str) -> None:
        """An r3m release `model.pt`: {"r3m": state_dict} with DataParallel's `module.` prefix; language heads ignored (r3m's
        remove_language_head does the same when it loads the file)."""
        sd = torch.load(path, map_location="cpu")
        sd = sd.get("r3m", sd)
        own = {k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()}
        own = {k: v for k, v in own.items() if k.startswith("convnet.") and not k.startswith("convnet.fc.")}
        self.r3m.load_state_dict(own, strict=True)

    def _trunk_tensors(self):
        return list(self.r3m.convnet.parameters()) + [b for b in self.r3m.convnet.buffers() if b.dtype.is_floating_point]

    def _fold_trunk(self):
        ts = self._trunk_tensors()
        wdtype = torch.bfloat16 if kn.get_compute() == "bf16" else torch.float32
        # a trainable trunk tensor (the affordance model's stem, r3m_rn18.py:34-38) may be moved by an optimizer that writes the parameter
        # arena through raw pointers (no version bump): such optimizers bump shadow's epoch instead
        live = shadow.epoch() if any(t.requires_grad for t in ts) else 0
        key = (wdtype, ts[0].device, tuple(t._version for t in ts), tuple(t.data_ptr() for t in ts), live)
        if key != self._folded_key:
            net = self.r3m.convnet
            with torch.no_grad():
                f = {"stem": _fold(net.conv1, net.bn1, 8, wdtype), "blocks": []}
                if wdtype == torch.bfloat16:
                    # packed stem (hulc_r3m_stem_fwd): [o][kh][kw -> 8][c -> 4], zeros at kw = 7 and c = 3
                    w8, b = _fold(net.conv1, net.bn1, 4, torch.float32)
                    w8 = w8.view(64, 7, 7, 4)
                    w8 = torch.cat([w8, w8.new_zeros(64, 7, 1, 4)], dim=2).reshape(64, 224).contiguous().to(torch.bfloat16)
                    f["stem_packed"] = (w8, b)
                for blk in net.blocks():
                    f["blocks"].append((_fold(blk.conv1, blk.bn1, 0, wdtype), _fold(blk.conv2, blk.bn2, 0, wdtype),
                                        _fold(blk.downsample[0], blk.downsample[1], 0, wdtype) if blk.downsample is not None else None,
                                        blk.stride))
            self._folded, self._folded_key = f, key
        return self._folded

    def _raw_trunk(self):
        """the convolution weights alone (OHWI, compute dtype), no BatchNorm folded in: the operands of the batch-statistics path.  The FROZEN
        layers' operands are cached on their own version counters; the stem's — trainable in the affordance model (r3m_rn18.py:34-38), moved
        by arena optimizers without a version bump — is derived from the live parameter on every call through shadow.weight_operand, which
        re-makes it once per step (inside a captured step too)."""
        net = self.r3m.convnet
        ws = [p for n, p in net.named_parameters() if n != "conv1.weight" and (n.endswith("conv1.weight") or n.endswith("conv2.weight")
              or n.endswith("downsample.0.weight"))]
        wdtype = torch.bfloat16 if kn.get_compute() == "bf16" else torch.float32
        key = (wdtype, ws[0].device, tuple(t._version for t in ws), tuple(t.data_ptr() for t in ws))
        if key != getattr(self, "_raw_key", None):
            def raw(conv):
                return conv.weight.float().permute(0, 2, 3, 1).reshape(conv.weight.shape[0], -1).contiguous().to(wdtype)
            with torch.no_grad():
                self._raw = {"blocks": [(raw(b.conv1), raw(b.conv2), raw(b.downsample[0]) if b.downsample is not None else None, b.stride)
                                        for b in net.blocks()],
                             "zero": torch.zeros(512, dtype=torch.float32, device=ws[0].device)}
            self._raw_key = key
        f = dict(self._raw)
        f["stem"] = shadow.weight_operand(net.conv1.weight, "ohwi_c8")
        return f

    @torch.no_grad()
    def _trunk_maps_batch_stats(self, x: torch.Tensor, mean, std):
        """The trunk with every BatchNorm2d in TRAINING mode (statistics of the batch, running statistics and num_batches_tracked updated): what
        hulc2/affordance/models/visual_lang_encoders/r3m_rn18.py:27-43 + pixel_aff_lang_detector.py:51-53 run during training (the
        parameters of layer1..4 are frozen, the modules stay in train mode).  Per layer: the bias-free convolution into an fp32 map,
        then hulc_nhwc_bn_train_fwd (partial sums in a fixed order, finalize, apply + residual + ReLU).  Returns the five NHWC maps."""
        f = self._raw_trunk()
        net = self.r3m.convnet
        adt = torch.bfloat16 if kn.get_compute() == "bf16" else torch.float32
        n, _, h, w = x.shape
        dev = x.device

        def conv_bn(a, wt, bn, hh, ww, cin, k, stride, pad, relu, add=None):
            cout = wt.shape[0]
            oh, ow = (hh + 2 * pad - k) // stride + 1, (ww + 2 * pad - k) // stride + 1
            z = torch.empty((n, oh, ow, cout), dtype=torch.float32, device=dev)
            kn.conv2d_padded_fwd(a, wt, f["zero"][:cout], z, n, hh, ww, cin, cout, k, k, stride, pad, relu=False)
            y = torch.empty((n, oh, ow, cout), dtype=adt, device=dev)
            kn.nhwc_bn_train_fwd(z, n * oh * ow, cout, bn.weight, bn.bias, bn.eps, 0.1 if bn.momentum is None else bn.momentum,
                                 bn.running_mean, bn.running_var, y, add=add, relu=relu)
            bn.num_batches_tracked += 1
            return y, oh, ow

        a = kn.r3m_normalize(x.contiguous(), torch.empty((n, h, w, 8), dtype=adt, device=dev), mean, std)
        a, h, w = conv_bn(a, f["stem"], net.bn1, h, w, 8, 7, 2, 3, True)
        ph, pw = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
        a = kn.maxpool_nhwc(a, torch.empty((n, ph, pw, 64), dtype=adt, device=dev), n, h, w, 64, 3, 2, 1)
        h, w, c = ph, pw, 64
        maps = [a]
        for bi, (blk, (w1, w2, wd, stride)) in enumerate(zip(net.blocks(), f["blocks"])):
            idn = a if wd is None else conv_bn(a, wd, blk.downsample[1], h, w, c, 1, stride, 0, False)[0]
            o, oh, ow = conv_bn(a, w1, blk.bn1, h, w, c, 3, stride, 1, True)
            c = w1.shape[0]
            a, h, w = conv_bn(o, w2, blk.bn2, oh, ow, c, 3, 1, 1, True, add=idn)
            if bi % 2 == 1:
                maps.append(a)
        return maps

    # ---- forward -------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def trunk_features(self, x: torch.Tensor, want_maps: bool = False, mean=None, std=None, batch_stats: bool = False):
        """x (N, 3, H, W) fp32 in [0, 255] -> (N, 512) fp32: normalise, stem, max pool, the residual stages, global average pool.
        want_maps: return the NHWC maps after the stem (+ max pool) and after each of the four stages instead (the skips of the affordance
        model's U-Net, hulc2/affordance/models/visual_lang_encoders/r3m_rn18.py:71-76); mean / std override the input normalisation."""
        if x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] != 3:
            raise TypeError("VisionR3M expects fp32 (N, 3, H, W) frames in [0, 255] (conf/datamodule/transforms/real_world_r3m.yaml)")
        mean = IMAGENET_MEAN if mean is None else mean
        std = IMAGENET_STD if std is None else std
        if batch_stats:
            if not want_maps:
                raise NotImplementedError("batch-statistics BatchNorm is the affordance model's trunk mode (maps); VisionR3M runs its trunk "
                                          "under no_grad in whatever mode Lightning set — use trunk_feature_maps")
            return self._trunk_maps_batch_stats(x, mean, std)
        f = self._fold_trunk()
        adt = torch.bfloat16 if kn.get_compute() == "bf16" else torch.float32
        n, _, h, w = x.shape
        dev = x.device

        def conv(a, wb, hh, ww, cin, k, stride, pad, relu, add=None):
            cout = wb[0].shape[0]
            oh, ow = (hh + 2 * pad - k) // stride + 1, (ww + 2 * pad - k) // stride + 1
            y = torch.empty((n, oh, ow, cout), dtype=adt, device=dev)
            kn.conv2d_padded_fwd(a, wb[0], wb[1], y, n, hh, ww, cin, cout, k, k, stride, pad, relu=relu, add=add)
            return y, oh, ow

        if "stem_packed" in f:
            xp = kn.r3m_normalize_packed(x.contiguous(), torch.empty((n, h + 6, kn.r3m_packed_width(w), 4), dtype=adt, device=dev), mean, std)
            oh, ow = (h - 1) // 2 + 1, (w - 1) // 2 + 1
            a = kn.r3m_stem_fwd(xp, f["stem_packed"][0], f["stem_packed"][1], torch.empty((n, oh, ow, 64), dtype=adt, device=dev), n, h, w, 64)
            h, w = oh, ow
        else:
            a = kn.r3m_normalize(x.contiguous(), torch.empty((n, h, w, 8), dtype=adt, device=dev), mean, std)
            a, h, w = conv(a, f["stem"], h, w, 8, 7, 2, 3, True)
        ph, pw = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
        a = kn.maxpool_nhwc(a, torch.empty((n, ph, pw, 64), dtype=adt, device=dev), n, h, w, 64, 3, 2, 1)
        h, w, c = ph, pw, 64
        maps = [a]
        # HULC_TRUNK_GRID=1 (bf16): the thirteen stride-1 3 x 3 convolutions on the padded grid (csrc/gridconv.hip with the BasicBlock epilogue);
        # a stage's first block (stride 2 + 1 x 1 shortcut) stays on the gather kernel and its outputs are put on the grid.  OFF by default —
        # measured twice: the gather kernel is the faster one at the trunk's channel counts (1024 frames of 150 x 200: 64 -> 64 at 38 x 50
        # 287 TFLOP/s on the grid against ~450 gathered, whole step 19.4 vs 12.9 ms; 32 images of 224 x 224: 5.03 vs 4.99 ms).
        on_grid = adt == torch.bfloat16 and os.environ.get("HULC_TRUNK_GRID", "0") == "1"
        ag = kn.grid_from_nhwc(a) if on_grid else None
        for bi, (c1, c2, ds, stride) in enumerate(f["blocks"]):
            if on_grid:
                if ds is None and stride == 1:
                    o = kn.gridconv3x3_fused(ag, c1[0], c1[0].shape[0], bias=c1[1], relu=True)
                    ag = kn.gridconv3x3_fused(o, c2[0], c2[0].shape[0], bias=c2[1], add=ag, relu=True)
                else:
                    a = ag.interior().contiguous()
                    idn = kn.grid_from_nhwc(conv(a, ds, h, w, c, 1, stride, 0, False)[0])
                    o, h, w = conv(a, c1, h, w, c, 3, stride, 1, True)
                    c = c1[0].shape[0]
                    ag = kn.gridconv3x3_fused(kn.grid_from_nhwc(o), c2[0], c, bias=c2[1], add=idn, relu=True)
                a = ag.interior()
            else:
                idn = a if ds is None else conv(a, ds, h, w, c, 1, stride, 0, False)[0]
                o, oh, ow = conv(a, c1, h, w, c, 3, stride, 1, True)
                c = c1[0].shape[0]
                a, h, w = conv(o, c2, oh, ow, c, 3, 1, 1, True, add=idn)
            if bi % 2 == 1:                                    # ResNet-18: two BasicBlocks per stage
                maps.append(a)
        if want_maps:
            return maps
        feat = torch.empty((n, c), dtype=torch.float32, device=dev)
        a = a.contiguous()                                     # (a grid view on the bf16 path: 25-49 pixels per frame)
        kn.strided_seq_sum(a, feat, n, h * w, c, h * w * c, c, c, 1.0 / (h * w))
        return feat

    def forward(self, x: Union[torch.Tensor, Sequence[torch.Tensor]], aug_shift=None, aug_pad: int = 0, frame_index=None) -> torch.Tensor:
        """x: (N, 3, H, W) fp32 in [0, 255], or one such tensor per modality of a step (rows of the result modality-major).  The real-world
        transforms apply no shift augmentation to this camera, and the uint8 episode-store path belongs to the from-scratch CNN."""
        xs = list(x) if isinstance(x, (list, tuple)) else [x]
        for opt in (aug_shift, frame_index):
            if opt is not None and any(o is not None for o in (opt if isinstance(opt, (list, tuple)) else [opt])):
                raise NotImplementedError("VisionR3M takes fp32 frames in [0, 255]; shift augmentation / episode-store indices are not part "
                                          "of the real-world static-camera transforms (conf/datamodule/transforms/real_world_r3m.yaml:2-13)")
        feat = [self.trunk_features(t) for t in xs]
        feat = feat[0] if len(feat) == 1 else torch.cat(feat, dim=0)
        return HF.mlp(feat, [(self.fc1.weight, self.fc1.bias, True), (self.fc2.weight, self.fc2.bias, False)])


class _TrunkOnly(VisionR3M):
    """VisionR3M's folded-trunk machinery around an existing R3M module (no heads): the affordance model's encoder"""

    def __init__(self, r3m: "R3M"):
        nn.Module.__init__(self)
        object.__setattr__(self, "_r3m_ref", r3m)
        self._folded = None
        self._folded_key = None

    @property
    def r3m(self):
        return self._r3m_ref


_trunks = {}


def _trunk_of(r3m: "R3M") -> "_TrunkOnly":
    """the (cached) folded-trunk wrapper of an R3M module"""
    t = _trunks.get(id(r3m))
    if t is None or t.r3m is not r3m:
        t = _trunks[id(r3m)] = _TrunkOnly(r3m)
    return t


def trunk_feature_maps(r3m: "R3M", img: torch.Tensor, batch_stats: bool = False):
    """the affordance encoder (hulc2/affordance/models/visual_lang_encoders/r3m_rn18.py:27-32,71-76 uses the ResNet's children directly: no
    / 255, no ImageNet normalisation inside — the dataset transforms did that): img (N, 3, H, W) fp32 -> [stem, layer1, layer2, layer3, layer4]
    NHWC maps in the compute dtype, BatchNorm folded (inference statistics) — or, batch_stats, every BatchNorm in training mode (statistics of the
    batch, running statistics updated): the trunk as the reference runs it while training"""
    t = _trunks.get(id(r3m))
    if t is None or t.r3m is not r3m:
        t = _trunks[id(r3m)] = _TrunkOnly(r3m)
    return t.trunk_features(img, want_maps=True, mean=(0.0, 0.0, 0.0), std=(1.0 / 255.0,) * 3, batch_stats=batch_stats)
