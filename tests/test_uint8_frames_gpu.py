"""uint8 NHWC frames straight into conv1: RandomShiftsAug / ScaleImageTensor / Normalize applied while the
band kernels stage the frame, against the oracle's restatement of the reference transforms followed by the fp32-input kernels."""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
pytestmark = pytest.mark.gpu

from hulc2_amd import synthetic as syn  # noqa: E402
from hulc2_amd.compat import instantiate  # noqa: E402
from hulc2_amd.config import default_model_config  # noqa: E402
from oracle import hulc2_oracle as O  # noqa: E402  (checker only)

G = ROOT / "tests" / "golden"


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda", 0)


@pytest.mark.parametrize("tag,hw,pad", [("static", 200, 10), ("gripper", 84, 4)])
@pytest.mark.parametrize("with_shift", [True, False])
def test_conv1_from_uint8_frames(dev, tag, hw, pad, with_shift):
    """conv1 forward and weight gradient fed by the stored uint8 frames == the same kernels fed by the transformed fp32 frames"""
    from hulc2_amd import kernels as kn

    kn.set_compute("bf16")
    fx = dict(np.load(G / f"transforms_{tag}.npz"))
    u8 = torch.tensor(fx["frames_u8"])
    shift = torch.tensor(fx["shift"]) if with_shift else None
    n = u8.shape[0]
    x32 = O.frames_u8_to_input(u8, pad, shift)                                  # reference transforms (oracle restatement)
    g = torch.Generator().manual_seed(3)
    w = (torch.randn(32, 192, generator=g) / 192 ** 0.5).to(torch.bfloat16).to(dev)
    b = torch.randn(32, generator=g).to(dev)
    oh = (hw - 8) // 4 + 1
    ys = []
    for x, sh in ((x32.to(dev), None), (u8.to(dev), None if shift is None else shift.to(dev))):
        y = torch.empty(n, oh, oh, 32, device=dev, dtype=torch.bfloat16)
        kn.conv2d_fwd(x, w, b, y, n, hw, hw, 3, 32, 8, 8, 4, True, relu=True, aug_shift=sh, aug_pad=pad)
        ys.append(y.float())
    torch.cuda.synchronize()
    scale = ys[0].abs().max().item()
    assert (ys[0] - ys[1]).abs().max().item() < 2e-2 * scale, "forward"       # inputs agree to 6e-5; both round to bf16
    assert ((ys[0] - ys[1]).norm() / ys[0].norm()).item() < 3e-3
    dy = torch.randn(n, oh, oh, 32, generator=g).to(dev).to(torch.bfloat16)
    gs = []
    for x, sh in ((x32.to(dev), None), (u8.to(dev), None if shift is None else shift.to(dev))):
        dw, db = torch.zeros(32, 192, device=dev), torch.zeros(32, device=dev)
        kn.conv2d_bwd_weight(x, dy, dw, db, n, hw, hw, 3, 32, 8, 8, 4, True, aug_shift=sh, aug_pad=pad)
        gs.append((dw, db))
    torch.cuda.synchronize()
    assert ((gs[0][0] - gs[1][0]).norm() / gs[0][0].norm()).item() < 3e-3, "weight gradient"
    assert torch.equal(gs[0][1], gs[1][1]), "bias gradient does not depend on the frames"


def test_training_step_from_uint8_frames(dev):
    """Hulc2.training_step on uint8 frames + shifts == on the fp32 frames the reference transforms make of them"""
    from hulc2_amd import kernels as kn

    kn.set_compute("bf16")
    m = instantiate(default_model_config(gripper_control=True, dropout_p=0.0)).to(dev)
    syn.fill_state_dict_(m.state_dict(), 9)
    m.train()
    B, S = 2, 4
    batch = syn.make_batch(9, B, S, device=dev)
    g = torch.Generator().manual_seed(4)
    b_u8, b_f32 = {}, {}
    for mod, db in batch.items():
        d8, d32 = dict(db), dict(db)
        d8["rgb_obs"], d32["rgb_obs"] = {}, {}
        for key, hw, pad in (("rgb_static", 200, 10), ("rgb_gripper", 84, 4)):
            u8 = torch.randint(0, 256, (B, S, hw, hw, 3), generator=g, dtype=torch.uint8)
            sh = torch.randint(0, 2 * pad + 1, (B, S, 2), generator=g, dtype=torch.int32)
            d8["rgb_obs"][key], d8["rgb_obs"][key + "_shift"] = u8.to(dev), sh.to(dev)
            d32["rgb_obs"][key] = O.frames_u8_to_input(u8.reshape(B * S, hw, hw, 3), pad, sh.reshape(B * S, 2)).reshape(B, S, 3, hw, hw).to(dev)
        b_u8[mod], b_f32[mod] = d8, d32
    outs = []
    for bt in (b_f32, b_u8):
        for p in m.parameters():
            p.grad = None
        loss = m.training_step(bt, 0)
        loss.backward()
        torch.cuda.synchronize()
        outs.append((loss.detach().clone(), m.perceptual_encoder.rgb_static_encoder.conv_model[0].weight.grad.clone(),
                     m.perceptual_encoder.rgb_gripper_encoder.conv_model[0].weight.grad.clone()))
    assert abs(float(outs[0][0]) - float(outs[1][0])) < 2e-3 * abs(float(outs[0][0]))
    for i in (1, 2):
        # whole-step bf16 gradients: the 6e-5 input differences (grid_sample rounding) flip bf16 roundings / ReLU masks upstream;
        # same bound as the bf16 gradient tolerance of tests/test_parity_gpu.py
        assert ((outs[0][i] - outs[1][i]).norm() / outs[0][i].norm()).item() < 0.15


@pytest.mark.parametrize("form", ["two_tensors", "one_store"])
def test_conv_stack_takes_both_modalities_in_one_conv1_launch(dev, form, monkeypatch):
    """functional.conv_stack on the two modalities' uint8 frames: conv1 as ONE launch per direction (two frame tensors: hulc_conv_desc.x2; two
    windows of one episode store: one index list) against one launch per modality (HULC_CONV1_PER_INPUT=1) — the activations bit for bit,
    conv1's weight and bias gradients to fp32 rounding of another summation order (1e-5 of the largest entry), everything downstream bit for bit."""
    from hulc2_amd import functional as HF, kernels as kn

    kn.set_compute("bf16")
    g = torch.Generator().manual_seed(12)
    hw, pad, Na, Nb = 200, 10, 6, 4
    ws = [((torch.rand(32, 3, 8, 8, generator=g) * 2 - 1) / 192 ** 0.5), torch.zeros(32),
          ((torch.rand(64, 32, 4, 4, generator=g) * 2 - 1) / 512 ** 0.5), torch.zeros(64),
          ((torch.rand(64, 64, 3, 3, generator=g) * 2 - 1) / 576 ** 0.5), torch.zeros(64)]
    sa = torch.randint(0, 2 * pad + 1, (Na, 2), generator=g, dtype=torch.int32).to(dev)
    sb = torch.randint(0, 2 * pad + 1, (Nb, 2), generator=g, dtype=torch.int32).to(dev)
    if form == "two_tensors":
        xs = [torch.randint(0, 256, (n, hw, hw, 3), generator=g, dtype=torch.uint8).to(dev) for n in (Na, Nb)]
        index = None
    else:
        store = torch.randint(0, 256, (16, hw, hw, 3), generator=g, dtype=torch.uint8).to(dev)
        xs = [store, store]
        index = [torch.randint(0, 16, (n,), generator=g, dtype=torch.int32).to(dev) for n in (Na, Nb)]
    outs = []
    for per_input in ("1", ""):
        if per_input:
            monkeypatch.setenv("HULC_CONV1_PER_INPUT", per_input)
        else:
            monkeypatch.delenv("HULC_CONV1_PER_INPUT", raising=False)
        params = [w.clone().to(dev).requires_grad_(True) for w in ws]
        a3 = HF.conv_stack(xs, params, aug_pad=pad, aug_shifts=[sa, sb], frame_index=index)
        (a3.float() * torch.linspace(-1, 1, a3.numel(), device=dev).reshape(a3.shape)).sum().backward()
        torch.cuda.synchronize()
        outs.append((a3.detach().clone(), [q.grad.clone() for q in params]))
    assert torch.equal(outs[0][0], outs[1][0])
    for i, (a, c) in enumerate(zip(outs[0][1], outs[1][1])):
        if i < 2:                                        # conv1's weight and bias gradients: one set of partial sums instead of two
            assert float((a - c).abs().max()) <= 1e-5 * float(a.abs().max()), f"gradient {i}"
        else:
            assert torch.equal(a, c), f"gradient {i}"


def test_conv1_uint8_beyond_the_frame_parameter_table(dev, monkeypatch):
    """a workgroup of the uint8 conv1 kernels reads its frames' shift / index from an LDS table of 256 units and, past it, straight from memory:
    with two workgroups for 120 frames (HULC_CONV1_SLOTS=2: 360 / 780 units each) most units take the second path — forward, sign plane and
    weight gradient equal the default launch (forward bit for bit; the gradient's slabs are summed per workgroup: fp32 rounding)."""
    from hulc2_amd import kernels as kn
    g = torch.Generator().manual_seed(21)
    N, hw, pad = 120, 200, 10
    x = torch.randint(0, 256, (16, hw, hw, 3), generator=g, dtype=torch.uint8).to(dev)
    ix = torch.randint(0, 16, (N,), generator=g, dtype=torch.int32).to(dev)
    sh = torch.randint(0, 2 * pad + 1, (N, 2), generator=g, dtype=torch.int32).to(dev)
    w2d = ((torch.rand(32, 192, generator=g) * 2 - 1) / 192 ** 0.5).to(dev).to(torch.bfloat16)
    b = ((torch.rand(32, generator=g) * 2 - 1) * 0.1).to(dev)
    OH, OW = kn.conv_out_hw(hw, hw, 8, 8, 4)
    dy = torch.randn(N, OH, OW, 32, generator=g).to(dev).to(torch.bfloat16)
    outs = []
    for slots in (None, "2"):
        if slots:
            monkeypatch.setenv("HULC_CONV1_SLOTS", slots)
        y = torch.zeros(N, OH, OW, 32, dtype=torch.bfloat16, device=dev)
        bits = torch.zeros(N * OH * OW, dtype=torch.int32, device=dev)
        dw, db = torch.zeros(32, 192, device=dev), torch.zeros(32, device=dev)
        kn.conv2d_fwd(x, w2d, b, y, N, hw, hw, 3, 32, 8, 8, 4, True, compute=kn.BF16, relu_bits=bits, aug_shift=sh, aug_pad=pad, frame_index=ix)
        kn.conv2d_bwd_weight(x, dy, dw, db, N, hw, hw, 3, 32, 8, 8, 4, True, compute=kn.BF16, aug_shift=sh, aug_pad=pad, frame_index=ix)
        torch.cuda.synchronize()
        outs.append((y, bits, dw, db))
    monkeypatch.delenv("HULC_CONV1_SLOTS", raising=False)
    assert torch.equal(outs[0][0].view(torch.int16), outs[1][0].view(torch.int16)) and torch.equal(outs[0][1], outs[1][1])
    assert float((outs[0][2] - outs[1][2]).abs().max()) <= 1e-5 * float(outs[0][2].abs().max())
    assert float((outs[0][3] - outs[1][3]).abs().max()) <= 1e-5 * float(outs[0][3].abs().max())
