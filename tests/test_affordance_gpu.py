"""Row f-4: one training step of the affordance model's trainable part on the GPU (hulc2_amd/affordance: padded-grid kernels, grouped weight
gradients) against the fixture produced by the reference's own decoder / head / depth modules (tests/golden/affordance_step_B2_64.npz) and
against the CPU oracle on a second shape; bf16 compute, so tolerances are those of the main path's bf16 tests (losses 2e-3, activations
3e-2 of max-abs, gradients relative L2)."""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
pytestmark = pytest.mark.gpu

from hulc2_amd import kernels as kn, synthetic as syn  # noqa: E402

G = ROOT / "tests" / "golden"
NAMES = {"text_fc.": "lang_encoder.text_fc.", "decoder.": "aff_stream.decoder.", "segmentation_head.": "aff_stream.segmentation_head.",
         "depth_stream.": "depth_stream."}


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    kn.set_compute("bf16")
    return torch.device("cuda", 0)


def ref_name(k):
    for a, b in NAMES.items():
        if k.startswith(a):
            return b + k[len(a):]
    raise KeyError(k)


def build(hw, seed, dev):
    from hulc2_amd.affordance import PixelAffLangDetector
    from oracle import affordance_oracle as A
    sd = {k: torch.empty(s) for k, s in A.trainable_shapes(hw // 32).items()}
    syn.fill_affordance_state_dict_(sd, seed)
    m = PixelAffLangDetector(img_size=hw).to(dev)
    own = dict(m.model.named_parameters())
    with torch.no_grad():
        for k, v in sd.items():
            own[ref_name(k)].copy_(v)
    m.train()
    return m, sd, own


def rel(a, b):
    a, b = torch.as_tensor(np.asarray(a)).double().flatten(), torch.as_tensor(np.asarray(b)).double().flatten()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def test_training_step_matches_the_reference_fixture():
    dev = _dev()
    g = dict(np.load(G / "affordance_step_B2_64.npz", allow_pickle=False))
    B, HW = int(g["B"]), int(g["HW"])
    m, sd, own = build(HW, int(g["seed"]), dev)
    feats = [torch.as_tensor(g[f"feat{i}"]).permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev) for i in range(5)]
    loss, info = m.forward_losses(feats, torch.as_tensor(g["emb"]).to(dev), torch.as_tensor(g["p0"]).to(dev), torch.as_tensor(g["gt_depth"]).to(dev))
    loss.backward()
    torch.cuda.synchronize()
    for k in ("loss", "aff_loss", "depth_loss"):
        got = float(loss if k == "loss" else info[k])
        assert abs(got - float(g[k])) <= 2e-3 * abs(float(g[k])) + 1e-6, (k, got, float(g[k]))
    lg = info["logits"][:, ::37].float().cpu()
    assert (lg - torch.as_tensor(g["logits_sub"])).abs().max().item() <= 3e-2 * np.abs(g["logits_sub"]).max()
    assert (info["mu"].detach().cpu() - torch.as_tensor(g["mu"])).abs().max().item() < 2e-2 and rel(info["sigma"].detach().cpu(), g["sigma"]) < 1e-2
    worst = {}
    for k in g:
        if k.startswith("gnorm."):
            p = own[ref_name(k[6:])]
            assert p.grad is not None, k
            e = abs(p.grad.norm().item() - float(g[k])) / (float(g[k]) + 1e-30)
            worst[k] = e
            if "segmentation_head.bias" in k:
                assert p.grad.abs().max().item() <= 1e-6                # analytically zero (the softmax and the one-hot both sum to 1)
                continue
            assert e <= 0.1, (k, e, p.grad.norm().item(), float(g[k]))
        elif k.startswith("grad."):
            p = own[ref_name(k[5:])]
            if "segmentation_head.bias" in k:
                continue
            gr = p.grad.flatten().cpu()
            got = gr if gr.numel() == g[k].shape[0] else gr[::97][:512]
            worst["slice:" + k] = rel(got, g[k])
            # the decoder's gradients are O(1e-7) (loss weight 0.1 / (B H W)) and reach block 0 through ten bf16 convolution + BatchNorm
            # layers: measured 2-16 % relative L2 against the fp32 reference, the deepest BatchNorm shifts worst
            assert rel(got, g[k]) <= 0.25, (k, rel(got, g[k]))
    unused = [n for n, p in own.items() if p.requires_grad and p.grad is None]
    assert sorted(unused) == sorted(f"aff_stream.decoder.blocks.{i}.lang_proj.{n}" for i in (3, 4) for n in ("weight", "bias"))
    blocks = m.model.aff_stream.decoder.blocks
    assert rel(blocks[0].conv1[1].running_mean.cpu(), g["bn_mean.b0c1"]) < 2e-2 and rel(blocks[4].conv2[1].running_var.cpu(), g["bn_var.b4c2"]) < 2e-2
    print("worst gradient errors:", [(k, round(v, 4)) for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:12]])


def test_inference_matches_the_reference_fixture():
    """eval mode (BatchNorm by the running statistics the reference's step produced): logits, the arg-max pixel, the depth distribution"""
    dev = _dev()
    g = dict(np.load(G / "affordance_step_B2_64.npz", allow_pickle=False))
    B, HW = int(g["B"]), int(g["HW"])
    m, sd, own = build(HW, int(g["seed"]), dev)
    blocks = m.model.aff_stream.decoder.blocks
    with torch.no_grad():
        for i in range(5):
            for c in ("conv1", "conv2"):
                bn = getattr(blocks[i], c)[1]
                bn.running_mean.copy_(torch.as_tensor(g[f"run_mean.b{i}{c}"]))
                bn.running_var.copy_(torch.as_tensor(g[f"run_var.b{i}{c}"]))
    m.eval()
    feats = [torch.as_tensor(g[f"feat{i}"]).permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev) for i in range(5)]
    with torch.no_grad():
        _, info = m.forward_losses(feats, torch.as_tensor(g["emb"]).to(dev), torch.as_tensor(g["p0"]).to(dev), torch.as_tensor(g["gt_depth"]).to(dev))
    torch.cuda.synchronize()
    lg = info["logits"].float().cpu()
    assert (lg[:, ::37] - torch.as_tensor(g["eval_logits_sub"])).abs().max().item() <= 3e-2 * np.abs(g["eval_logits_sub"]).max()
    am = lg.argmax(-1)
    ref = torch.as_tensor(g["eval_argmax"]).long()
    for b in range(B):      # bf16 logits: the arg-max may move to a pixel whose reference logit is within the bf16 noise of the maximum
        assert am[b] == ref[b] or float(lg[b, ref[b]]) >= float(lg[b].max()) - 3e-2 * float(lg[b].abs().max()), (b, int(am[b]), int(ref[b]))
    assert (info["mu"].cpu() - torch.as_tensor(g["eval_mu"])).abs().max().item() < 2e-2
    assert rel(torch.softmax(lg, -1).max(-1).values, g["eval_softmax_max"]) < 5e-2
    # repeated inference recycles the grid buffers (no growth of the pool between calls)
    img = torch.randn(B, 3, HW, HW, generator=torch.Generator().manual_seed(0)).to(dev)
    emb = torch.as_tensor(g["emb"]).to(dev)
    m.forward({"img": img, "lang_goal": emb})
    held = sum(len(v) for v in kn.Grid._pool.values())
    for _ in range(3):
        m.forward({"img": img, "lang_goal": emb})
    assert sum(len(v) for v in kn.Grid._pool.values()) == held


def test_training_step_matches_the_oracle_with_sinks_and_trunk():
    """a second shape (B = 3, 96 x 96) end to end through the HIP trunk and the native trainer's gradient arena (deferred grouped weight
    gradients), against the oracle fed with the trunk's own maps"""
    dev = _dev()
    from oracle import affordance_oracle as A
    from hulc2_amd.trainer import ArenaTrainer
    B, HW = 3, 96
    m, sd, own = build(HW, 7, dev)
    syn.fill_state_dict_({"r3m.convnet." + k: v for k, v in m.model.aff_stream.r3m.convnet.state_dict().items()}, 7)
    gen = torch.Generator().manual_seed(11)
    img = torch.randn(B, 3, HW, HW, generator=gen).to(dev)
    emb = (torch.randn(B, 384, generator=gen) * 0.5).to(dev)
    p0 = torch.stack([torch.randint(0, HW, (B,), generator=gen), torch.randint(0, HW, (B,), generator=gen)], 1)
    depth = torch.randn(B, generator=gen)
    feats = m.trunk_maps(img)
    assert [tuple(f.shape) for f in feats] == [(B, 24, 24, 64), (B, 24, 24, 64), (B, 12, 12, 128), (B, 6, 6, 256), (B, 3, 3, 512)]
    # the HIP trunk's maps against the oracle's restatement of the frozen ResNet-18 (bf16 activations through 17 layers)
    tsd = {"r3m.convnet." + k: v.detach().float().cpu() for k, v in m.model.aff_stream.r3m.convnet.state_dict().items()}
    for got, want in zip(feats, A.trunk_maps(tsd, img.cpu())):
        assert (got.float().permute(0, 3, 1, 2).cpu() - want).abs().max().item() <= 4e-2 * want.abs().max().item()
    # oracle on the same maps
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = A.training_step(osd, [f.float().permute(0, 3, 1, 2).cpu() for f in feats], emb.cpu(), p0, depth, HW)
    out["loss"].backward()
    tr = ArenaTrainer(m, lr=1e-4, overlap=False)
    batch = ({"img": img, "lang_goal": emb}, {"p0": p0.to(dev), "normalized_depth": depth.to(dev)})
    loss = tr._forward_backward(batch, 0)
    torch.cuda.synchronize()
    assert abs(float(loss) - float(out["loss"])) <= 2e-3 * abs(float(out["loss"])) + 1e-6
    own = dict(m.model.named_parameters())
    errs = {}
    for k, v in osd.items():
        if v.grad is None or "segmentation_head.bias" in k:
            continue
        errs[k] = rel(own[ref_name(k)].grad.cpu(), v.grad)
    print("gradient errors vs the oracle:", [(k, round(v, 4)) for k, v in sorted(errs.items(), key=lambda kv: -kv[1])])
    # the decoder's gradients are O(1e-7) sums that cancel: block 0's sit at 26-32 % on bf16 activations and move by a few
    # points with the summation order of the frozen trunk's layer1 (round 4: the band kernel's accumulators start from the bias) — the bound
    # per tensor is loose, the median over the decoder is the statement
    for k, e in errs.items():
        assert e <= 0.36, (k, e)
    dec = sorted(e for k, e in errs.items() if k.startswith("decoder."))
    assert dec[len(dec) // 2] <= 0.26 and max(e for k, e in errs.items() if not k.startswith("decoder.")) <= 0.04, (dec[len(dec) // 2], dec[-3:])
    # the public inference entry points on the same model: heat map sums to one, arg-max pixel inside the image, validation errors finite
    m.eval()
    out = m.forward({"img": img, "lang_goal": emb})
    assert tuple(out["aff"].shape) == (B, HW, HW, 1) and abs(float(out["aff"].sum()) - B) < 1e-3 * B
    px, dpt, sig = m.predict_pixels(img, emb, depth_norm=(0.5, 0.1))
    assert tuple(px.shape) == (B, 2) and int(px.min()) >= 0 and int(px.max()) < HW and torch.isfinite(dpt).all() and (sig > 0).all()
    val = m.validation_step(({"img": img, "lang_goal": emb}, {"p0": p0.to(dev), "normalized_depth": depth.to(dev), "depth": depth.to(dev) * 0.1 + 0.5}), 0, (0.5, 0.1))
    assert all(torch.isfinite(torch.as_tensor(v)).all() for v in val.values())


def test_training_loop_tracks_the_oracle_and_replays_as_a_graph():
    """five optimizer steps on one batch (48 x 64 would not be square: 64 x 64, B = 4): the per-step losses of the native trainer follow the
    oracle's Adam trajectory (BatchNorm running statistics included), the loss goes down, and the hipGraph replay continues the descent"""
    dev = _dev()
    from oracle import affordance_oracle as A
    from hulc2_amd.trainer import ArenaTrainer
    B, HW, steps = 4, 64, 5
    m, sd, own = build(HW, 23, dev)
    syn.fill_state_dict_({"r3m.convnet." + k: v for k, v in m.model.aff_stream.r3m.convnet.state_dict().items()}, 23)
    gen = torch.Generator().manual_seed(5)
    img = torch.randn(B, 3, HW, HW, generator=gen).to(dev)
    emb = (torch.randn(B, 384, generator=gen) * 0.5).to(dev)
    p0 = torch.stack([torch.randint(0, HW, (B,), generator=gen), torch.randint(0, HW, (B,), generator=gen)], 1)
    depth = torch.randn(B, generator=gen)
    feats = [f.float().permute(0, 3, 1, 2).cpu() for f in m.trunk_maps(img)]               # the frozen trunk's maps: constant over the steps
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    opt = torch.optim.Adam(list(osd.values()), lr=1e-4)                    # conf/affordance/train_affordance.yaml:29
    want = []
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        loss = A.training_step(osd, feats, emb.cpu(), p0, depth, HW)["loss"]
        loss.backward()
        opt.step()
        want.append(float(loss))
    assert want[-1] < want[0], "the oracle itself must be learning"
    tr = ArenaTrainer(m, lr=1e-4, overlap=False)
    batch = ({"img": img, "lang_goal": emb}, {"p0": p0.to(dev), "normalized_depth": depth.to(dev)})
    got = [float(tr.step(batch, i)) for i in range(steps)]
    # the loss crosses zero (Gaussian NLL): compare against the size of the descent — Adam normalises the bf16 path's gradient noise into
    # small step differences that add up (measured: identical to 3e-4 for two steps, 3 % of the descent after five)
    span = want[0] - want[-1]
    assert all(abs(a - b) <= 0.06 * span for a, b in zip(got, want)), (got, want)
    tr.capture(batch)                                                                    # two more eager steps, then the graphs
    l1 = float(tr.replay())
    l2 = float(tr.replay())
    assert l1 == l1 and l2 < got[0] and l2 <= l1 * 1.05, (got, l1, l2)
    # nn.BatchNorm2d.num_batches_tracked counts every training forward, eager and replayed alike (state_dicts interchange)
    n_fwd = {int(b) for b in m.bn_step_counters()}
    assert len(m.bn_step_counters()) == 10 and len(n_fwd) == 1 and n_fwd.pop() >= steps + 2, [int(b) for b in m.bn_step_counters()]


def test_trunk_on_the_grid_option_matches_the_default_trunk(monkeypatch):
    """HULC_TRUNK_GRID=1 (the stride-1 BasicBlock convolutions through hulc_gridconv3x3_fused; an opt-in, slower than the gather kernel at
    these sizes) hands out the same five maps as strided views"""
    dev = _dev()
    m, sd, own = build(64, 3, dev)
    syn.fill_state_dict_({"r3m.convnet." + k: v for k, v in m.model.aff_stream.r3m.convnet.state_dict().items()}, 3)
    img = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(4)).to(dev)
    base = [f.float().clone() for f in m.trunk_maps(img)]
    monkeypatch.setenv("HULC_TRUNK_GRID", "1")
    alt = m.trunk_maps(img)
    torch.cuda.synchronize()
    assert not alt[-1].is_contiguous()                       # a grid tensor's pixel view
    for a, b in zip(alt, base):
        assert a.shape == b.shape and (a.float() - b).abs().max().item() <= 3e-2 * b.abs().max().item()


def test_full_size_step_properties():
    """BASELINE configs[4] at the size bench.py --affordance runs (224 x 224, B = 32): no oracle finishes that in seconds, so the step is held to
    size-independent properties — finite losses and gradients, bit-identical repeats (fixed summation orders everywhere), gradients of the pixel
    loss sum to zero over an image's logits (softmax minus one-hot), hipGraph replay == eager for the same parameters, BatchNorm statistics move."""
    dev = _dev()
    from hulc2_amd.trainer import ArenaTrainer
    B, HW = 32, 224
    m, sd, own = build(HW, 31, dev)
    syn.fill_state_dict_({"r3m.convnet." + k: v for k, v in m.model.aff_stream.r3m.convnet.state_dict().items()}, 31)
    gen = torch.Generator().manual_seed(9)
    img = torch.randn(B, 3, HW, HW, generator=gen).to(dev)
    emb = (torch.randn(B, 384, generator=gen) * 0.5).to(dev)
    p0 = torch.stack([torch.randint(0, HW, (B,), generator=gen), torch.randint(0, HW, (B,), generator=gen)], 1).to(dev)
    depth = torch.randn(B, generator=gen).to(dev)
    batch = ({"img": img, "lang_goal": emb}, {"p0": p0, "normalized_depth": depth})
    tr = ArenaTrainer(m, lr=0.0, overlap=False)                     # lr 0: the parameters stay put, every step sees the same problem
    bn0 = [b.clone() for b in m.bn_buffers()]
    runs = []
    for i in range(2):
        loss = tr._forward_backward(batch, i)
        torch.cuda.synchronize()
        runs.append((loss.clone(), tr.flat_g.clone()))
    assert torch.isfinite(runs[0][0]) and torch.isfinite(runs[0][1]).all()
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1]), "the step is not bit-reproducible"
    nz = sum(1 for p in tr.params if p.grad is not None and float(p.grad.abs().max()) > 0)
    assert nz >= len(tr.params) - 6, (nz, len(tr.params))           # (lang_proj of the last two blocks is unused by the reference too; head bias is analytically 0)
    assert any(not torch.equal(a, b) for a, b in zip(bn0, m.bn_buffers()))
    # segmentation-head bias gradient = sum over pixels of (softmax - one-hot) = 0 per image
    hb = m.model.aff_stream.segmentation_head.bias.grad
    assert hb is not None and float(hb.abs().max()) <= 1e-5
    # replay == eager (BatchNorm running statistics differ between the runs but do not enter a training-mode step)
    tr.capture(batch)
    l_replay = tr.replay().clone()
    g_replay = tr.flat_g.clone()
    torch.cuda.synchronize()
    assert torch.equal(l_replay, runs[0][0]) and torch.equal(g_replay, runs[0][1]), (float(l_replay), float(runs[0][0]))
    tr.close()


@pytest.mark.parametrize("compute,tol", [("fp32", 2e-4), ("bf16", 5e-2)])
def test_reference_trunk_mode_matches_train_mode_batchnorm_fixture(compute, tol):
    """VERDICT r03 missing #1: `trunk_mode="reference"` — the trunk with every BatchNorm2d in training mode, as the reference runs it
    (r3m_rn18.py:27-43 freezes the parameters of layer1..4 only; pixel_aff_lang_detector.py:51-53 leaves train mode on) — against
    tests/golden/r3m_trunk_trainmode.npz from torch's own nn.Conv2d / nn.BatchNorm2d / nn.MaxPool2d layers in TRAIN mode: the five maps the
    decoder receives, the running statistics after the forward, num_batches_tracked; eval mode falls back to the running statistics; the
    default mode ("frozen") is unchanged and differs."""
    dev = _dev()
    from hulc2_amd.affordance import PixelAffLangDetector
    g = dict(np.load(G / "r3m_trunk_trainmode.npz", allow_pickle=False))
    B, HW, seed = int(g["B"]), int(g["HW"]), int(g["seed"])
    kn.set_compute(compute)
    try:
        m = PixelAffLangDetector(img_size=HW, trunk_mode="reference").to(dev)
        net = m.model.aff_stream.r3m.convnet
        syn.fill_state_dict_({"r3m.convnet." + k: v for k, v in net.state_dict().items()}, seed)
        for b in net.buffers():
            if not b.is_floating_point():
                b.zero_()
        img = torch.randn(B, 3, HW, HW, generator=syn._gen(seed, "x.trunk.train")).to(dev)
        m.train()
        maps = m.trunk_maps(img)
        torch.cuda.synchronize()
        for i, got in enumerate(maps):
            want = torch.as_tensor(g[f"map{i}"])
            got = got.float().permute(0, 3, 1, 2).cpu()
            assert got.shape == want.shape and (got - want).abs().max().item() <= tol * want.abs().max().item(), (i, (got - want).abs().max().item())
        sd = net.state_dict()
        for name, want in zip(g["stat_names"], g["stat_sums"]):
            got = float(sd[str(name)].double().sum())
            assert abs(got - float(want)) <= (1e-4 if compute == "fp32" else 2e-2) * max(1.0, abs(float(want))), (name, got, float(want))
        assert int(sd["bn1.num_batches_tracked"]) == 1 and int(sd["layer3.0.downsample.1.num_batches_tracked"]) == 1
        # eval mode: the running statistics (now the updated ones) — identical to what the frozen mode computes from the same buffers
        m.eval()
        ev = m.trunk_maps(img)
        m2 = PixelAffLangDetector(img_size=HW).to(dev)
        m2.model.aff_stream.r3m.load_state_dict(m.model.aff_stream.r3m.state_dict())
        m2.train()
        fr = m2.trunk_maps(img)
        assert all(torch.equal(a, b) for a, b in zip(ev, fr))
        assert (fr[4].float() - maps[4].float()).abs().max().item() > 0.05 * maps[4].float().abs().max().item()
    finally:
        kn.set_compute("bf16")


@pytest.mark.parametrize("compute,tol", [("fp32", 2e-3), ("bf16", 0.12)])
def test_trainable_stem_gradients_match_the_nn_layer_fixture(compute, tol):
    """Round 5, VERDICT r04 missing #1 / row f-4: the reference's freeze (r3m_rn18.py:34-38) leaves the stem's conv1.weight, bn1.weight and
    bn1.bias trainable, and their gradient is the data gradient through the whole frozen ResNet-18 on train-mode BatchNorms.  TrunkStemFn
    (hulc2_amd/affordance/trunk.py) against tests/golden/r3m_trunk_trainmode.npz: torch's own nn layers, seeded upstream gradients of the five
    maps -> d conv1.weight, d bn1.weight, d bn1.bias (relative L2; fp32 mode: fp32 MFMA + fp32 maps; bf16 mode: bf16 operands and gradient
    maps through twenty layers)."""
    dev = _dev()
    from hulc2_amd.affordance import PixelAffLangDetector
    g = dict(np.load(G / "r3m_trunk_trainmode.npz", allow_pickle=False))
    B, HW, seed = int(g["B"]), int(g["HW"]), int(g["seed"])
    kn.set_compute(compute)
    try:
        m = PixelAffLangDetector(img_size=HW, trunk_mode="reference").to(dev)
        net = m.model.aff_stream.r3m.convnet
        syn.fill_state_dict_({"r3m.convnet." + k: v for k, v in net.state_dict().items()}, seed)
        assert net.conv1.weight.requires_grad and net.bn1.weight.requires_grad and net.bn1.bias.requires_grad
        assert not net.layer1[0].conv1.weight.requires_grad and not net.layer4[1].bn2.weight.requires_grad
        img = torch.randn(B, 3, HW, HW, generator=syn._gen(seed, "x.trunk.train")).to(dev)
        m.train()
        maps = m.trunk_maps(img)
        assert all(t.requires_grad for t in maps)
        loss = 0.0
        for i, t in enumerate(maps):
            up = (torch.randn(B, t.shape[3], t.shape[1], t.shape[2], generator=syn._gen(seed, f"g.trunk.map{i}")) * (0.5 ** i)).permute(0, 2, 3, 1)
            loss = loss + (t.float() * up.to(dev)).sum()
        loss.backward()
        torch.cuda.synchronize()
        for p, name in ((net.conv1.weight, "d_conv1_weight"), (net.bn1.weight, "d_bn1_weight"), (net.bn1.bias, "d_bn1_bias")):
            want = torch.as_tensor(g[name])
            got = p.grad.float().cpu()
            err = (got - want).norm().item() / want.norm().item()
            print(f"[{compute}] {name}: relative L2 {err:.3e}")
            assert got.shape == want.shape and err <= tol, (name, err)
        assert net.layer1[0].conv1.weight.grad is None
    finally:
        kn.set_compute("bf16")


def test_whole_step_trains_the_stem_in_reference_mode():
    """the affordance step in trunk_mode="reference" through the native trainer: the stem's three tensors are in the arena, receive a finite
    non-zero gradient through decoder -> skip connections / depth head -> frozen ResNet, and move under Adam; the frozen layers do not; the
    HULC_AFF_FROZEN_STEM=1 path (round 4's behaviour) leaves them alone.  Against the oracle (its own trunk restatement with train-mode
    BatchNorm, autograd through it) the stem's gradient NORMS agree to the bf16 level of the decoder's own gradients."""
    dev = _dev()
    from oracle import affordance_oracle as A
    from hulc2_amd.affordance import PixelAffLangDetector
    from hulc2_amd.trainer import ArenaTrainer
    B, HW = 4, 64
    kn.set_compute("bf16")
    m = PixelAffLangDetector(img_size=HW, trunk_mode="reference").to(dev)
    syn.fill_affordance_state_dict_({k: v for k, v in m.state_dict().items() if ".r3m." not in k}, 5)
    net = m.model.aff_stream.r3m.convnet
    syn.fill_state_dict_({"r3m.convnet." + k: v for k, v in net.state_dict().items()}, 5)
    m.train()
    gen = torch.Generator().manual_seed(3)
    img = torch.randn(B, 3, HW, HW, generator=gen).to(dev)
    emb = (torch.randn(B, 384, generator=gen) * 0.5).to(dev)
    p0 = torch.stack([torch.randint(0, HW, (B,), generator=gen), torch.randint(0, HW, (B,), generator=gen)], 1)
    depth = torch.randn(B, generator=gen)
    # oracle: trunk (train-mode BatchNorm, stem tensors require grad) -> decoder step
    tsd = {"r3m.convnet." + k: v.detach().float().cpu().clone() for k, v in net.state_dict().items()}
    stem = ["r3m.convnet.conv1.weight", "r3m.convnet.bn1.weight", "r3m.convnet.bn1.bias"]
    for k in stem:
        tsd[k].requires_grad_(True)
    osd = {}
    own = dict(m.model.named_parameters())
    for k, shape in A.trainable_shapes(HW // 32).items():
        osd[k] = own[ref_name(k)].detach().float().cpu().clone().requires_grad_(True)
    omaps = A.trunk_maps(tsd, img.cpu(), bn_train=True)
    out = A.training_step(osd, omaps, emb.cpu(), p0, depth, HW)
    out["loss"].backward()
    tr = ArenaTrainer(m, lr=1e-4, overlap=False)
    names = {id(p): n for n, p in m.named_parameters()}
    in_arena = {names[id(p)] for p in tr.params}
    assert {"model.aff_stream.r3m.convnet.conv1.weight", "model.aff_stream.r3m.convnet.bn1.weight", "model.aff_stream.r3m.convnet.bn1.bias"} <= in_arena
    assert not any("layer" in n and ".r3m." in n for n in in_arena)
    batch = ({"img": img, "lang_goal": emb}, {"p0": p0.to(dev), "normalized_depth": depth.to(dev)})
    before = net.conv1.weight.detach().clone()
    loss = tr._forward_backward(batch, 0)
    torch.cuda.synchronize()
    assert abs(float(loss) - float(out["loss"])) <= 1e-2 * abs(float(out["loss"])) + 1e-6
    for k, p in zip(stem, (net.conv1.weight, net.bn1.weight, net.bn1.bias)):
        got, want = p.grad.float().cpu(), tsd[k].grad
        assert torch.isfinite(got).all() and float(got.abs().max()) > 0.0, k
        ratio = got.norm().item() / want.norm().item()
        err = (got - want).norm().item() / want.norm().item()
        print(f"{k}: |g| ratio {ratio:.3f}, relative L2 {err:.3f}")
        assert 0.6 <= ratio <= 1.6 and err <= 0.6, (k, ratio, err)
    tr.optimizer_step()
    torch.cuda.synchronize()
    assert not torch.equal(net.conv1.weight, before)


@pytest.mark.parametrize("graph", [False, True])
def test_stem_operand_follows_the_parameter_over_several_steps(graph):
    """the stem's kernel-side operand was cached on Parameter._version / data_ptr only, and the arena Adam moves
    conv1.weight through raw pointers (no version bump): from the second step on the trunk convolved with the INITIAL stem weights.  After N
    native-trainer steps (eager, and replayed as a graph) the stem map of a train-mode forward must equal the stem computed by torch from the
    CURRENT conv1.weight / bn1 (batch statistics), and must differ from the one the initial weights give."""
    dev = _dev()
    from hulc2_amd.affordance import PixelAffLangDetector
    from hulc2_amd.trainer import ArenaTrainer
    B, HW = 4, 64
    kn.set_compute("bf16")
    m = PixelAffLangDetector(img_size=HW, trunk_mode="reference").to(dev)
    syn.fill_affordance_state_dict_({k: v for k, v in m.state_dict().items() if ".r3m." not in k}, 5)
    net = m.model.aff_stream.r3m.convnet
    syn.fill_state_dict_({"r3m.convnet." + k: v for k, v in net.state_dict().items()}, 5)
    m.train()
    gen = torch.Generator().manual_seed(3)
    img = torch.randn(B, 3, HW, HW, generator=gen).to(dev)
    emb = (torch.randn(B, 384, generator=gen) * 0.5).to(dev)
    p0 = torch.stack([torch.randint(0, HW, (B,), generator=gen), torch.randint(0, HW, (B,), generator=gen)], 1)
    depth = torch.randn(B, generator=gen)
    batch = ({"img": img, "lang_goal": emb}, {"p0": p0.to(dev), "normalized_depth": depth.to(dev)})
    tr = ArenaTrainer(m, lr=3e-2, overlap=False)           # (a large step: the stem must move visibly within a few steps)
    w0 = net.conv1.weight.detach().clone()
    if graph:
        tr.capture(batch)
        for _ in range(4):
            tr.replay()
    else:
        for i in range(4):
            tr.step(batch, i)
    torch.cuda.synchronize()
    assert (net.conv1.weight - w0).abs().max().item() > 1e-2

    def torch_stem(w):                                       # (train-mode BatchNorm removes whatever scale the trunk gives its input)
        z = torch.nn.functional.conv2d(img, w, stride=2, padding=3)
        mu, var = z.mean((0, 2, 3), keepdim=True), z.var((0, 2, 3), unbiased=False, keepdim=True)
        y = (z - mu) / torch.sqrt(var + net.bn1.eps) * net.bn1.weight.view(1, -1, 1, 1) + net.bn1.bias.view(1, -1, 1, 1)
        return torch.nn.functional.max_pool2d(torch.relu(y), 3, 2, 1)
    with torch.no_grad():
        got = m.trunk_maps(img)[0].float().permute(0, 3, 1, 2)
        want_now, want_old = torch_stem(net.conv1.weight.detach()), torch_stem(w0)
    e_now = (got - want_now).abs().max().item() / want_now.abs().max().item()
    e_old = (got - want_old).abs().max().item() / want_old.abs().max().item()
    print(f"stem map vs current weights {e_now:.3e}, vs initial weights {e_old:.3e}")
    assert e_now <= 3e-2 and e_old > 3 * e_now, (e_now, e_old)
