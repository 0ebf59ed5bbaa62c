"""GPU parity of the frozen R3M trunk (VisionR3M) against the CPU oracle.

The reference's trunk arithmetic lives in the un-vendored `r3m` submodule (parity unpinned); the oracle restates its public
definition (oracle/hulc2_oracle.py::r3m_trunk_features) and is itself pinned against torch's nn layers in tests/test_oracle_golden.py.

Tolerances: fp32 compute 1e-3 of max-abs per tensor (20 chained convolutions; fp32 MFMA sums in a different order than the CPU);
bf16 compute 5e-2 of max-abs for the 512 trunk features (activations are stored in bf16 between the 20 layers); gradients of the two
trainable layers in relative L2 (1e-3 fp32, 0.15 bf16 — the bounds of tests/test_parity_gpu.py).
"""
import sys
from pathlib import Path

import pytest
import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
pytestmark = pytest.mark.gpu

from hulc2_amd import kernels as kn  # noqa: E402
from hulc2_amd import synthetic as syn  # noqa: E402
from hulc2_amd.compat import instantiate  # noqa: E402
from hulc2_amd.config import real_world_model_config  # noqa: E402
from hulc2_amd.models.perceptual_encoders.vision_r3m import VisionR3M  # noqa: E402
from oracle import hulc2_oracle as O  # noqa: E402


@pytest.fixture(params=["fp32", "bf16"])
def mode(request):
    kn.set_compute(request.param)
    yield request.param
    kn.set_compute("bf16")


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape and torch.isfinite(a).all()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


@pytest.mark.parametrize("geom", [
    # N, H, W, Cin, Cout, K, stride, pad, relu, add
    (3, 37, 50, 8, 64, 7, 2, 3, True, False),       # the stem (3 channels padded to 8)
    (2, 38, 50, 64, 64, 3, 1, 1, True, True),        # layer1 second conv: residual + ReLU
    (2, 38, 50, 64, 128, 3, 2, 1, True, False),      # layer2 first conv
    (2, 38, 50, 64, 128, 1, 2, 0, False, False),     # downsample
    (5, 5, 7, 512, 512, 3, 1, 1, True, True),        # layer4: 8 output-channel blocks, K = 4608
    (1, 9, 9, 16, 32, 3, 1, 1, False, False),
])
def test_padded_conv_matches_torch(dev, mode, geom):
    n, h, w, cin, cout, k, s, pad, relu, add = geom
    g = syn._gen(7, f"padconv{geom}")
    adt = torch.bfloat16 if mode == "bf16" else torch.float32
    x = torch.randn(n, cin, h, w, generator=g).to(adt).float()
    wt = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(adt).float()
    b = torch.randn(cout, generator=g) * 0.1
    oh, ow = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
    r = torch.randn(n, cout, oh, ow, generator=g).to(adt).float() if add else None
    ref = F.conv2d(x, wt, b, s, pad)
    if add:
        ref = ref + r
    if relu:
        ref = F.relu(ref)
    y = torch.empty(n, oh, ow, cout, dtype=adt, device=dev)
    kn.conv2d_padded_fwd(x.permute(0, 2, 3, 1).contiguous().to(dev, adt), wt.permute(0, 2, 3, 1).reshape(cout, -1).contiguous().to(dev, adt),
                         b.to(dev), y, n, h, w, cin, cout, k, k, s, pad, relu=relu,
                         add=None if r is None else r.permute(0, 2, 3, 1).contiguous().to(dev, adt))
    assert rel(y.float().permute(0, 3, 1, 2), ref) <= (1e-5 if mode == "fp32" else 1e-2)     # bf16: only the output rounding differs


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_maxpool_is_exact(dev, dtype):
    g = syn._gen(3, "maxpool")
    x = torch.randn(3, 64, 75, 100, generator=g).to(dtype)
    ref = F.max_pool2d(x.float(), 3, 2, 1)
    y = torch.empty(3, ref.shape[2], ref.shape[3], 64, dtype=dtype, device=dev)
    kn.maxpool_nhwc(x.permute(0, 2, 3, 1).contiguous().to(dev), y, 3, 75, 100, 64, 3, 2, 1)
    assert torch.equal(y.float().cpu().permute(0, 3, 1, 2), ref)


def test_normalize(dev):
    g = syn._gen(5, "norm")
    x = torch.rand(2, 3, 30, 44, generator=g) * 255
    y = torch.empty(2, 30, 44, 8, dtype=torch.float32, device=dev)
    kn.r3m_normalize(x.to(dev), y, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225))
    ref = (x / 255 - torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)) / torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    assert rel(y[..., :3].permute(0, 3, 1, 2), ref) < 1e-6
    assert (y[..., 3:] == 0).all()


@pytest.fixture(scope="module")
def net(dev):
    m = VisionR3M(dev, 64).to(dev)
    syn.fill_state_dict_(m.state_dict(), 11)
    return m


def test_trunk_and_head_match_oracle(dev, net, mode):
    """150 x 200 frames (the size SURVEY §8d assigns to configs[3]) and a second, odd size: trunk features, encoder output and the
    gradients of the two trainable layers."""
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    for hw in ((150, 200), (97, 131)):
        x = torch.rand(3, 3, *hw, generator=syn._gen(13, f"frames{hw}")) * 255
        want_f = O.r3m_trunk_features(sd, x)
        got_f = net.trunk_features(x.to(dev))
        assert rel(got_f, want_f) <= (1e-3 if mode == "fp32" else 5e-2), hw
    for p in net.parameters():
        p.grad = None
    out = net(x.to(dev))
    r = torch.randn(out.shape, generator=syn._gen(13, "r"))
    (out * r.to(dev)).sum().backward()
    osd = {k: v.clone().requires_grad_(k.startswith("fc")) for k, v in sd.items()}
    want = O.vision_r3m(osd, "", x)
    (want * r).sum().backward()
    tol = 1e-3 if mode == "fp32" else 5e-2
    assert rel(out, want) <= tol
    gtol = 1e-3 if mode == "fp32" else 0.15         # relative L2, the gradient bound of tests/test_parity_gpu.py (a ReLU unit of fc1 may flip in bf16)
    for name in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"):
        gp = dict(net.named_parameters())[name].grad
        assert gp is not None, name
        want_g = osd[name].grad.double()
        assert ((gp.double().cpu() - want_g).norm() / want_g.norm()).item() <= gtol, name
    assert all(p.grad is None for p in net.r3m.parameters())          # the trunk is frozen (vision_r3m.py:14-16, 25-26)


def test_folded_weights_follow_the_parameters(dev, net):
    x = (torch.rand(2, 3, 64, 64, generator=syn._gen(1, "x")) * 255).to(dev)
    a = net.trunk_features(x)
    with torch.no_grad():
        net.r3m.convnet.bn1.running_mean.add_(0.5)
    b = net.trunk_features(x)
    with torch.no_grad():
        net.r3m.convnet.bn1.running_mean.sub_(0.5)
    c = net.trunk_features(x)
    assert not torch.equal(a, b) and torch.equal(a, c)


# what each mode is held to on configs[3], stated here: loss (relative), median and worst relative-L2 gradient error over the 89 trainable
# tensors = measured on an MI355X (round 4) with <= 2x head-room.  Worst tensor in every bf16-family mode: the gripper camera's conv1 weight,
# whose bf16 BACKWARD product over the frames is 6-12 % off (torch.autocast(bfloat16) on the oracle: the same).
RW_BARS = {("fp32", 2): dict(loss=1e-5, med=1e-3, worst=2e-3),              # measured 1e-7 / 0 / 0
           ("bf16", 2): dict(loss=1e-4, med=0.045, worst=0.25),             # 1.1e-5 / 2.1 % / 12.3 %
           ("mixed", 2): dict(loss=1e-4, med=0.03, worst=0.13),             # 2.2e-6 / 1.4 % / 6.5 %
           ("bf16", 32): dict(loss=1e-4, med=0.016, worst=0.24),            # 8.8e-6 / 0.77 % / 11.7 %
           ("mixed", 32): dict(loss=1e-4, med=0.01, worst=0.105),           # 1.0e-6 / 0.47 % / 5.2 %
           ("bf16+sites", 32): dict(loss=1e-4, med=0.016, worst=0.22)}      # 8.6e-6 / 0.77 % / 10.7 %


_rw_oracle_cache = {}         # (B, S) -> (loss, {name: gradient}): the oracle's frozen trunk at B = 32 takes about a minute on 8 host threads


@pytest.mark.parametrize("B,S,cmode", [(2, 16, "fp32"), (2, 16, "bf16"), (2, 16, "mixed"), (32, 32, "bf16"), (32, 32, "mixed"), (32, 32, "bf16+sites")])
def test_real_world_training_step_matches_oracle(dev, B, S, cmode, monkeypatch):
    """cfg_low_level_rw (BASELINE configs[3]): R3M static camera in [0, 255], whole-embedding decoder input, world-frame actions, no CLIP
    loss — one training_step against the oracle composed the same way: the loss AND the gradient of every trainable parameter, in the exact
    fp32 mode, the benchmarked bf16 mode and the mixed mode at B = 2, and at the benchmark's full size (B = 32 per modality, S = 32) in the
    bf16, mixed and bf16 + every-exact-site modes (the oracle runs once per size).  VERDICT r02 next #9: configs[3] was
    pinned by one scalar."""
    tol_key = cmode
    if cmode == "bf16+sites":
        monkeypatch.setenv("HULC_FP32_SITES", "head,goal,encfc,txl,conv1,a3")
        cmode = "bf16"
    kn.set_compute(cmode)
    try:
        cfg = real_world_model_config(dropout_p=0.0)
        m = instantiate(cfg).to(dev)
        syn.fill_state_dict_(m.state_dict(), 21)
        m.train()
        batch = syn.make_batch(21, B, S, static_hw=(150, 200))
        for mod in batch.values():
            mod["rgb_obs"]["rgb_static"] = (mod["rgb_obs"]["rgb_static"] + 1) * 127.5          # UpScaleImageTensor range
        got = m.training_step(syn._to(batch, dev), 0)
        got.backward()
        torch.cuda.synchronize()
    finally:
        kn.set_compute("bf16")
    trainable = {k for k, p in m.named_parameters() if p.requires_grad}
    sd = {}
    for k, v in m.state_dict().items():
        sd[k] = v.detach().cpu().clone()
        if k in trainable:
            sd[k].requires_grad_(True)
    flat = {}
    for name, db in batch.items():
        flat[name] = dict(rgb_static=db["rgb_obs"]["rgb_static"], rgb_gripper=db["rgb_obs"]["rgb_gripper"], actions=db["actions"],
                          robot_obs=db["state_info"]["robot_obs"], plan_idx=db["plan_idx"])
        if name == "lang":
            flat[name].update(lang=db["lang"], use_for_aux_lang_loss=db["use_for_aux_lang_loss"])
    if (B, S) not in _rw_oracle_cache:
        nthreads = torch.get_num_threads()
        torch.set_num_threads(min(8, nthreads))
        try:
            want = O.training_step(sd, flat, O.real_world_cfg())["total_loss"]
            want.backward()
        finally:
            torch.set_num_threads(nthreads)
        _rw_oracle_cache[(B, S)] = (want.item(), {k: v.grad.clone() for k, v in sd.items() if v.grad is not None})
    want_loss, want_grads = _rw_oracle_cache[(B, S)]
    t = RW_BARS[(tol_key, B)]
    assert abs(got.item() - want_loss) <= t["loss"] * abs(want_loss), (got.item(), want_loss)
    assert all(p.grad is None for p in m.perceptual_encoder.rgb_static_encoder.r3m.parameters())
    errs, checked = {}, 0
    for k, p in m.named_parameters():
        if not p.requires_grad:
            continue
        ref = want_grads.get(k)
        if ref is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        assert p.grad is not None, k
        assert torch.isfinite(p.grad).all(), f"{k}: {int((~torch.isfinite(p.grad)).sum())} non-finite gradient entries of {p.grad.numel()}"
        errs[k] = ((p.grad.double().cpu() - ref.double()).norm() / (ref.double().norm() + 1e-30)).item()
        checked += 1
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:6]
    med = sorted(errs.values())[len(errs) // 2]
    print(f"[{tol_key} B={B}] loss rel {abs(got.item() - want_loss) / abs(want_loss):.2e}; {checked} gradients, median {med:.5f}, worst:", [(k, round(v, 5)) for k, v in worst])
    assert checked >= 60
    bad = {k: v for k, v in errs.items() if v > t["worst"]}
    assert not bad, bad
    assert med <= t["med"], med


def test_shipped_real_world_config_with_sentence_encoder(dev):
    """cfg_low_level_rw exactly as shipped: `language_encoder: sbert` (sentences in the batch, encoded inside the step, no gradient) on top of
    the R3M static camera.  The WordPiece vocabulary is not available offline, so the tokenizer is a deterministic stand-in (word hash ->
    id); everything behind it — MiniLM, language goal encoder, the rest of the step — is checked against the oracle, fp32 compute."""
    import zlib

    from tests.test_oracle_golden import _bert_sd

    def tokenizer(sentences):
        rows = [[101] + [1000 + zlib.crc32(w.encode()) % 20000 for w in s.split()] + [102] for s in sentences]
        n = max(len(r) for r in rows)
        ids = torch.tensor([r + [0] * (n - len(r)) for r in rows])
        mask = torch.tensor([[1] * len(r) + [0] * (n - len(r)) for r in rows])
        return {"input_ids": ids, "attention_mask": mask}

    kn.set_compute("fp32")
    try:
        cfg = real_world_model_config(dropout_p=0.0)
        cfg["language_encoder"] = type(cfg)({"_target_": "hulc2.affordance.models.language_encoders.sbert_lang_encoder.SBertLang",
                                             "nlp_model": "paraphrase-MiniLM-L3-v2", "freeze_backbone": True})
        from hulc2_amd.compat import install_as_hulc2
        install_as_hulc2()
        m = instantiate(cfg).to(dev)
        syn.fill_state_dict_(m.state_dict(), 23)
        bert = _bert_sd(23)
        m.lang_encoder.load_bert_state_dict(bert)
        m.lang_encoder.tokenizer = tokenizer
        assert m.language_goal.lang_net is m.lang_encoder
        m.train()
        batch = syn.make_batch(23, 2, 16, static_hw=(150, 200))
        for mod in batch.values():
            mod["rgb_obs"]["rgb_static"] = (mod["rgb_obs"]["rgb_static"] + 1) * 127.5
        sentences = ["push the red block to the left", "open the drawer"]
        sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        tok = tokenizer(sentences)
        flat = {}
        for name, db in batch.items():
            flat[name] = dict(rgb_static=db["rgb_obs"]["rgb_static"], rgb_gripper=db["rgb_obs"]["rgb_gripper"], actions=db["actions"],
                              robot_obs=db["state_info"]["robot_obs"], plan_idx=db["plan_idx"])
        flat["lang"].update(lang=O.minilm_sentence_embedding(bert, tok["input_ids"], tok["attention_mask"]),
                            use_for_aux_lang_loss=batch["lang"]["use_for_aux_lang_loss"])
        want = O.training_step(sd, flat, O.real_world_cfg())["total_loss"]
        gb = syn._to(batch, dev)
        gb["lang"]["lang"] = sentences
        got = m.training_step(gb, 0)
        assert abs(got.item() - want.item()) <= 1e-3 * abs(want.item())
        got.backward()
        assert all(p.grad is None for p in m.lang_encoder.parameters())           # sbert_lang_encoder.py:41-54: no_grad + detach
        assert m.language_goal.mlp[1].weight.grad is not None
    finally:
        kn.set_compute("bf16")


def test_real_world_validation_and_rollout(dev):
    """cfg_low_level_rw through validation_step (hulc2.py:594-598) and the batch-1 control loop reset / step (hulc2.py:600-628): the R3M
    trunk under torch.no_grad(), frames in [0, 255], world-frame actions (no tcp transform), no CLIP head."""
    m = instantiate(real_world_model_config(dropout_p=0.1)).to(dev)
    syn.fill_state_dict_(m.state_dict(), 31)
    m.eval()
    B, S = 2, 8
    batch = syn.make_batch(5, B, S, device=dev, static_hw=(150, 200))
    for db in batch.values():
        db.pop("plan_idx", None)
        db["rgb_obs"]["rgb_static"] = (db["rgb_obs"]["rgb_static"] + 1) * 127.5
    out = m.validation_step(batch, 0)
    for mod in ("vis", "lang"):
        assert out[f"sampled_plan_pp_{mod}"].shape == (B, 1024)
        assert torch.equal(out[f"sampled_plan_pr_{mod}"].reshape(B, 32, 32).sum(-1), torch.ones(B, 32, device=dev))
    for k in ("val_act/vis_act_loss_pp", "val_act/lang_act_loss_pr", "val_kl/lang_kl_loss", "val_total_mae/vis_total_mae_pr"):
        assert torch.isfinite(torch.as_tensor(m.logged[k])).all(), k
    m.replan_freq = 2
    m.reset()
    vis = batch["vis"]
    goal = {"lang": batch["lang"]["lang"][:1]}
    plans = []
    for s in range(4):
        obs = {"rgb_obs": {k: v[:1, s:s + 1] for k, v in vis["rgb_obs"].items()}, "depth_obs": {},
               "robot_obs": vis["robot_obs"][:1, s:s + 1], "robot_obs_raw": vis["state_info"]["robot_obs"][:1, s:s + 1]}
        a = m.step(obs, goal)
        assert a.shape == (1, 1, 7) and torch.isfinite(a).all()
        plans.append(m.plan.clone())
    assert torch.equal(plans[0], plans[1]) and torch.equal(plans[2], plans[3])
    # the encoder's batch-1 output equals its batched output on the same frame (no batch statistics anywhere in the frozen trunk)
    enc = m.perceptual_encoder.rgb_static_encoder
    x = vis["rgb_obs"]["rgb_static"].reshape(-1, 3, 150, 200)
    with torch.no_grad():
        full, one = enc(x), enc(x[3:4])
    assert (full[3:4] - one).abs().max().item() <= 2e-2 * full.abs().max().item()
