"""Deterministic synthetic parameters and CALVIN-shaped batches.

Used by bench.py (device-resident synthetic play sequences), by the golden-fixture
generator and by the tests: because values are a pure function of (name, shape, seed) the fixtures
under tests/golden/ only need to carry seeds + checksums instead of 47 M parameters.
"""
import math
import zlib
from typing import Dict

import torch


def _gen(seed: int, name: str) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2**63 - 1))
    return g


@torch.no_grad()
def fill_state_dict_(sd: Dict[str, torch.Tensor], seed: int) -> None:
    """In-place, order-independent init of every floating tensor of a state_dict whose keys follow the
    reference's names.  Weights/biases ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (the nn.Linear/Conv default
    scale), LayerNorm gains 1 + 0.1 N(0,1), LayerNorm biases 0.1 N(0,1), embeddings N(0,1).
    Registered buffers (maps, bounds, eye) are left untouched."""
    skip = ("x_map", "y_map", "temperature", "one_hot_embedding_eye", ".ones", "gripper_bounds",
            "action_max_bound", "action_min_bound")
    for name, t in sd.items():
        if not t.is_floating_point() or any(s in name for s in skip):
            continue
        g = _gen(seed, name)
        leaf = name.rsplit(".", 1)[-1]
        if "r3m.convnet." in name:
            # frozen ResNet trunk (VisionR3M): convolutions at torchvision's kaiming fan_out scale, BatchNorm with non-trivial affine
            # parameters and running statistics so the folding is exercised (a variance must stay positive)
            if t.dim() == 4:
                t.copy_(torch.randn(t.shape, generator=g) * math.sqrt(2.0 / (t.shape[0] * t.shape[2] * t.shape[3])))
            elif leaf in ("weight", "running_var"):
                t.copy_(torch.rand(t.shape, generator=g) + 0.5)
            else:
                t.copy_(torch.randn(t.shape, generator=g) * 0.1)
        elif name == "logit_scale":
            t.fill_(math.log(1 / 0.07))
        elif ".ln." in name or ".norm1." in name or ".norm2." in name or ".layernorm." in name:
            r = torch.randn(t.shape, generator=g) * 0.1
            t.copy_(r + 1.0 if leaf == "weight" else r)
        elif "position_embeddings" in name:
            t.copy_(torch.randn(t.shape, generator=g))
        else:
            if t.dim() >= 2:
                bound = 1.0 / math.sqrt(t[0].numel())
            elif "bias_ih" in name or "bias_hh" in name:
                bound = 1.0 / math.sqrt(t.numel())
            else:
                bound = 0.05
            t.copy_((torch.rand(t.shape, generator=g) * 2 - 1) * bound)


def fill_affordance_state_dict_(sd: Dict[str, torch.Tensor], seed: int) -> None:
    """The recipe for the affordance model's trainable part (language-fused U-Net decoder, segmentation head, depth
    head, text_fc): convolution / linear weights U(+-1/sqrt(fan_in)), biases U(+-0.05), BatchNorm gains 1 + 0.1 N(0,1) and shifts
    0.1 N(0,1) (4-D weights mark a convolution, a 1-D `.1.weight` / `.1.bias` inside a Conv2dReLU is its BatchNorm); running statistics
    and counters are left at their initial values."""
    for name, t in sd.items():
        if not t.is_floating_point() or "running_" in name:
            continue
        g = _gen(seed, name)
        leaf = name.rsplit(".", 1)[-1]
        if t.dim() >= 2:
            t.copy_((torch.rand(t.shape, generator=g) * 2 - 1) / math.sqrt(t[0].numel()))
        elif ".conv1.1." in name or ".conv2.1." in name:                      # nn.BatchNorm2d of a Conv2dReLU
            r = torch.randn(t.shape, generator=g) * 0.1
            t.copy_(r + 1.0 if leaf == "weight" else r)
        else:
            t.copy_((torch.rand(t.shape, generator=g) * 2 - 1) * 0.05)


def fill_bert_state_dict_(sd: Dict[str, torch.Tensor], seed: int) -> None:
    """The same recipe for a transformers BertModel state_dict (SURVEY §8 row f-3 parity fixtures): LayerNorm gains 1 + 0.1 N(0,1),
    LayerNorm biases 0.1 N(0,1), embedding tables 0.5 N(0,1), dense weights U(+-2/sqrt(fan_in)) (sharp enough that the attention
    softmax is not uniform), dense biases U(+-0.05).  Integer buffers (position_ids) are left untouched."""
    for name, t in sd.items():
        if not t.is_floating_point():
            continue
        g = _gen(seed, name)
        leaf = name.rsplit(".", 1)[-1]
        if "LayerNorm" in name:
            r = torch.randn(t.shape, generator=g) * 0.1
            t.copy_(r + 1.0 if leaf == "weight" else r)
        elif "embeddings" in name:
            t.copy_(torch.randn(t.shape, generator=g) * 0.5)
        elif t.dim() >= 2:
            t.copy_((torch.rand(t.shape, generator=g) * 2 - 1) * (2.0 / math.sqrt(t.shape[1])))
        else:
            t.copy_((torch.rand(t.shape, generator=g) * 2 - 1) * 0.05)


def checksum(sd: Dict[str, torch.Tensor]) -> Dict[str, float]:
    """Per-tensor (sum, abs-sum) in float64 — stored in fixtures to prove the recipe reproduced."""
    out = {}
    for k, v in sd.items():
        if v.is_floating_point():
            d = v.detach().double()
            out[k] = (float(d.sum()), float(d.abs().sum()))
    return out


def make_modality_batch(seed: int, name: str, B: int, S: int, lang: bool, static_hw=(200, 200),
                        gripper_hw=(84, 84), device="cpu") -> Dict:
    """One modality's batch in the reference's nested-dict contract (hulc2.py:336-361):
    images ~ U(-1,1) (post Normalize(0.5,0.5) range), actions ~ U(-1,1) with a ±1 gripper command,
    state_info.robot_obs (15) with euler angles in columns 3:6, lang ~ 0.05 N(0,1) (B,384)."""
    g = _gen(seed, "batch." + name)

    def u(*shape):
        return torch.rand(shape, generator=g) * 2 - 1

    rgb_static = u(B, S, 3, *static_hw)
    rgb_gripper = u(B, S, 3, *gripper_hw)
    robot_obs = torch.randn(B, S, 8, generator=g)
    state_robot_obs = torch.randn(B, S, 15, generator=g)
    state_robot_obs[..., 3:6] = u(B, S, 3) * (math.pi * 0.5)
    actions = u(B, S, 7)
    actions[..., 6] = (torch.rand(B, S, generator=g) < 0.5).float() * 2 - 1
    plan_idx = torch.randint(0, 32, (B, 32), generator=g)
    batch = {
        "rgb_obs": {"rgb_static": rgb_static, "rgb_gripper": rgb_gripper},
        "depth_obs": {},
        "robot_obs": robot_obs,
        "actions": actions,
        "state_info": {"robot_obs": state_robot_obs},
        "idx": torch.arange(B),
        "lang": torch.empty(0),
        "plan_idx": plan_idx,      # not part of the reference contract: injected categorical sample for parity
    }
    if lang:
        batch["lang"] = torch.randn(B, 384, generator=g) * 0.05
        batch["use_for_aux_lang_loss"] = torch.ones(B, dtype=torch.bool)
    return _to(batch, device)


def make_batch(seed: int, B: int, S: int, device="cpu", **kw) -> Dict[str, Dict]:
    """{'vis': ..., 'lang': ...} as the reference's CombinedLoader yields (hulc2.py:379)."""
    return {
        "vis": make_modality_batch(seed, "vis", B, S, lang=False, device=device, **kw),
        "lang": make_modality_batch(seed, "lang", B, S, lang=True, device=device, **kw),
    }


def _to(x, device):
    if isinstance(x, torch.Tensor):
        return x.to(device)
    if isinstance(x, dict):
        return {k: _to(v, device) for k, v in x.items()}
    return x
