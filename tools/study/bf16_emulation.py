"""CPU study (build container): which bf16 roundings of the benchmarked mode cost gradient fidelity when the contrastive head is on.

The oracle's F.conv2d / F.linear are replaced by autograd Functions that emulate an MFMA kernel with bf16 operands and fp32 accumulation:
forward operands rounded (fo), forward output stored as bf16 (fs), backward operands rounded (bo), data gradient stored as bf16 (bs), or a
split of the ACTIVATION operand into hi + lo bf16 parts (x2: two MFMAs per product, weight single bf16) / both operands (x3).
Sites are chosen by parameter name.  Prints median / max relative-L2 gradient error against the unmodified fp32 oracle.

    python tools/study/bf16_emulation.py B S "name=site:flags;site:flags" ...     flags from {fo,fs,bo,bs,x2,x3}
"""
import sys
import time
from pathlib import Path

import torch
import torch.nn.functional as TF

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from hulc2_amd import param_spec, synthetic as syn  # noqa: E402
from oracle import hulc2_oracle as O  # noqa: E402


import os

# HULC_EMU_HALF=fp16: every rounding of the study is to IEEE half (10-bit mantissa) instead of bf16 (7 bits) — with `rest:fo,fs,bo,bs` an
# emulation of the reference's own `precision: 16` autocast (conf/trainer/play_trainer.yaml:3): conv / linear operands and results in half,
# fp32 accumulation, everything else fp32 (loss scaling changes no rounding; overflow is not modelled)
# HULC_EMU_HALF=fp16fwd: half in the forward direction only, bf16 in the backward products (no loss scaling needed: the small numbers live there)
_MODE = os.environ.get("HULC_EMU_HALF", "")
HALF = torch.float16 if _MODE in ("fp16", "fp16fwd") else torch.bfloat16
HALF_BWD = torch.float16 if _MODE == "fp16" else torch.bfloat16


def r16(t):
    return t.to(HALF).to(torch.float32)


def r16b(t):
    return t.to(HALF_BWD).to(torch.float32)


def split(t):
    hi = r16(t)
    return hi, r16(t - hi)


SITES = {
    "conv": lambda n: "conv_model" in n and not n.endswith("conv_model.7.weight"),
    "conv1": lambda n: "conv_model.0." in n,
    "conv2": lambda n: "conv_model.2." in n,
    "conv3": lambda n: "conv_model.4." in n,
    "encfc": lambda n: n.startswith("perceptual_encoder.") and ("fc1" in n or "fc2" in n or "conv_model.7" in n),
    "encfc1": lambda n: n.startswith("perceptual_encoder.") and "fc1" in n,
    "encfc2": lambda n: n.startswith("perceptual_encoder.") and "fc2" in n,
    "encflat": lambda n: n.startswith("perceptual_encoder.") and "conv_model.7" in n,
    "txlffn": lambda n: n.startswith("plan_recognition.transformer_encoder") and "linear" in n,
    "txlattn": lambda n: n.startswith("plan_recognition.transformer_encoder") and "self_attn" in n,
    "first": lambda n: n in ("language_goal.mlp.1.weight", "visual_goal.mlp.0.weight", "plan_proposal.fc_model.0.weight"),
    "rnn": lambda n: n.startswith("action_decoder."),
    "txl": lambda n: n.startswith("plan_recognition.transformer_encoder"),
    "head": lambda n: n.startswith("plan_recognition.fc.") or n.startswith("proj_vis_lang."),
    "rest": lambda n: True,
}


class Emu:
    def __init__(self, spec, names):
        self.rules = []
        for part in spec.split(";"):
            if part:
                site, flags = part.split(":")
                self.rules.append((SITES[site], set(flags.split(",")) - {""}))
        self.names = names

    def flags(self, w):
        n = self.names.get(id(w), "?")
        for pred, fl in self.rules:
            if pred(n):
                return fl
        return set()


def product(kind, x, w, fl, fwd, **kw):
    """kind: 'conv' / 'lin'; emulated product with the operand treatment of direction fwd/bwd"""
    op = (lambda a, b: TF.conv2d(a, b, None, **kw)) if kind == "conv" else (lambda a, b: TF.linear(a, b))
    key = "fo" if fwd else "bo"
    if "x3" in fl:
        xh, xl = split(x)
        wh, wl = split(w)
        return op(xh, wh) + op(xl, wh) + op(xh, wl)
    if "x2" in fl:
        xh, xl = split(x)
        return op(xh, r16(w)) + op(xl, r16(w))
    if key in fl:
        return op(r16(x), r16(w))
    return op(x, w)


class ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, fl):
        y = product("conv", x, w, fl, True, stride=stride) + b.view(1, -1, 1, 1)
        ctx.save_for_backward(x, w)
        ctx.stride, ctx.fl = stride, fl
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        fl = ctx.fl
        rd = (lambda t: r16b(t)) if ("bo" in fl and "x3" not in fl and "x2" not in fl) else (lambda t: t)
        dyr, xr, wr = rd(dy), rd(x), rd(w)
        dx = torch.nn.grad.conv2d_input(x.shape, wr, dyr, stride=ctx.stride) if ctx.needs_input_grad[0] else None
        dw = torch.nn.grad.conv2d_weight(xr, w.shape, dyr, stride=ctx.stride)
        if dx is not None and "bs" in fl:
            dx = r16b(dx)
        return dx, dw, dy.sum((0, 2, 3)), None, None


class LinFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, fl):
        y = product("lin", x, w, fl, True)
        if b is not None:
            y = y + b
        ctx.save_for_backward(x, w)
        ctx.fl, ctx.has_b = fl, b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        fl = ctx.fl
        rd = (lambda t: r16b(t)) if ("bo" in fl and "x3" not in fl and "x2" not in fl) else (lambda t: t)
        dyr, xr, wr = rd(dy), rd(x), rd(w)
        dx = dyr @ wr
        dw = dyr.reshape(-1, dy.shape[-1]).t() @ xr.reshape(-1, x.shape[-1])
        db = dy.reshape(-1, dy.shape[-1]).sum(0) if ctx.has_b else None
        return dx, dw, db, None


class FProxy:
    def __init__(self, emu):
        self.emu = emu

    def __getattr__(self, k):
        return getattr(TF, k)

    def conv2d(self, x, w, b=None, stride=1, padding=0):
        fl = self.emu.flags(w)
        if not fl or padding != 0:
            return TF.conv2d(x, w, b, stride, padding)
        y = ConvFn.apply(x, w, b, stride, fl)
        return y

    def relu(self, x):
        return TF.relu(x)

    def linear(self, x, w, b=None):
        fl = self.emu.flags(w)
        if not fl:
            return TF.linear(x, w, b)
        return LinFn.apply(x, w, b, fl)


class StoreRound(torch.autograd.Function):
    """activation stored as bf16 after the ReLU (forward), gradient passes"""
    @staticmethod
    def forward(ctx, x):
        return r16(x)

    @staticmethod
    def backward(ctx, g):
        return g


def run(B, S, spec, sd_vals, batch, seed):
    sd = {k: v.clone().requires_grad_(True) for k, v in sd_vals.items()}
    names = {id(v): k for k, v in sd.items()}
    emu = Emu(spec, names)
    old_F, old_stack = O.F, O._conv_stack
    O.F = FProxy(emu)
    conv_fl = emu.flags(sd["perceptual_encoder.rgb_static_encoder.conv_model.0.weight"])
    if "fs" in conv_fl:                         # conv activations stored bf16 (after ReLU) — wrap the oracle's conv stack
        def stack(sd_, p, x):
            F = O.F
            keep = set(os.environ.get("HULC_EMU_FS_LAYERS", "0,2,4").split(","))        # which conv outputs are stored rounded
            for i, st in ((0, 4), (2, 2), (4, 1)):
                x = F.relu(F.conv2d(x, sd_[p + f"conv_model.{i}.weight"], sd_[p + f"conv_model.{i}.bias"], stride=st))
                if str(i) in keep:
                    x = StoreRound.apply(x)
            return x
        O._conv_stack = stack
    try:
        out = O.training_step(sd, batch, dict(gripper_control=True, use_clip_auxiliary_loss=True))
        out["total_loss"].backward()
    finally:
        O.F, O._conv_stack = old_F, old_stack
    return out, {k: v.grad for k, v in sd.items()}


def main():
    B, S = int(sys.argv[1]), int(sys.argv[2])
    specs = sys.argv[3:]
    seed = 321
    torch.set_num_threads(8)
    sd0 = {k: torch.empty(s) for k, s in param_spec.trainable_shapes().items()}
    syn.fill_state_dict_(sd0, seed)
    raw = syn.make_batch(seed, B, S)
    batch = {}
    for m, db in raw.items():
        batch[m] = dict(rgb_static=db["rgb_obs"]["rgb_static"], rgb_gripper=db["rgb_obs"]["rgb_gripper"], actions=db["actions"],
                        robot_obs=db["state_info"]["robot_obs"], plan_idx=db["plan_idx"])
        if m == "lang":
            batch[m].update(lang=db["lang"], use_for_aux_lang_loss=db["use_for_aux_lang_loss"])
    t0 = time.time()
    ref_out, ref = run(B, S, "", sd0, batch, seed)
    print(f"reference {time.time() - t0:.1f}s total {float(ref_out['total_loss']):.6f}", flush=True)
    for item in specs:
        name, _, spec = item.partition("=")
        t0 = time.time()
        out, g = run(B, S, spec, sd0, batch, seed)
        errs = {k: ((g[k] - ref[k]).double().norm() / (ref[k].double().norm() + 1e-30)).item() for k in g if g[k] is not None and ref[k] is not None}
        v = sorted(errs.values())
        worst = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
        emb_err = (out["emb_lang"] - ref_out["emb_lang"]).abs().max().item() / ref_out["emb_lang"].abs().max().item()
        print(f"[{name:>28}] {time.time() - t0:5.1f}s loss {float(out['total_loss']):.6f} emb {emb_err:.1e} median {v[len(v) // 2]:.4f} max {v[-1]:.4f} >5%: "
              f"{sum(e > 0.05 for e in v)}/{len(v)}  worst: " + ", ".join(f"{n.split('.')[0][:6]}..{'.'.join(n.split('.')[-3:])}={e:.3f}" for n, e in worst), flush=True)


if __name__ == "__main__":
    main()
