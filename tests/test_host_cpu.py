"""CPU-side checks (no GPU compute): the checkpoint contract, config plumbing, the C-ABI surface and the
"fail loudly, never fall back" rule."""
import ctypes
import re
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from hulc2_amd import param_spec, synthetic as syn  # noqa: E402
from hulc2_amd.compat import Config, instantiate, install_as_hulc2  # noqa: E402
from hulc2_amd.config import default_model_config  # noqa: E402


@pytest.fixture(scope="module")
def model():
    return instantiate(default_model_config())


def test_state_dict_matches_reference_contract(model):
    """parameter names/shapes = SURVEY.md §8b (enumerated from the reference modules)"""
    sd = model.state_dict()
    want = param_spec.trainable_shapes()
    got_params = {k: tuple(v.shape) for k, v in model.named_parameters()}
    assert set(got_params) == set(want), (sorted(set(got_params) ^ set(want)))
    for k, shp in want.items():
        assert got_params[k] == tuple(shp), k
    for k, shp in param_spec.buffer_shapes().items():
        assert k in sd and tuple(sd[k].shape) == tuple(shp), k
    assert sum(p.numel() for p in model.parameters()) == param_spec.num_trainable()


def test_spatial_softmax_buffers_follow_reference_quirk(model):
    ss = model.perceptual_encoder.rgb_static_encoder.spatial_softmax
    lin = torch.linspace(-1, 1, 21)
    assert torch.equal(ss.x_map.view(21, 21)[:, 0], lin)      # x_map varies along rows (vision_network.py:88-92)
    assert torch.equal(ss.y_map.view(21, 21)[0, :], lin)
    assert torch.equal(ss.x_map.view(21, 21)[3], lin[3].expand(21))


def test_setup_input_sizes_and_targets():
    cfg = default_model_config()
    m = instantiate(cfg)
    assert cfg.plan_proposal.perceptual_features == 128 and cfg.action_decoder.plan_features == 1024
    assert type(m).__name__ == "Hulc2" and type(m.action_decoder).__name__ == "LogisticDecoderRNN"
    install_as_hulc2()
    import hulc2.models.hulc2 as ref_path   # the reference's import path now resolves to this package
    assert ref_path.Hulc2 is type(m)


def test_unsupported_configs_fail_loudly():
    cfg = default_model_config()
    cfg.plan_recognition.positional_normalize = True
    with pytest.raises(NotImplementedError):
        instantiate(cfg)
    cfg = default_model_config()
    cfg.action_decoder.rnn_model = "gru_decoder"
    with pytest.raises(NotImplementedError):
        instantiate(cfg)


def test_no_cpu_fallback(model):
    """a CPU batch must raise, not silently run something else"""
    from hulc2_amd.lib import HulcKernelError

    batch = syn.make_batch(0, 1, 2)
    with pytest.raises(HulcKernelError):
        model.training_step(batch, 0)


def test_abi_library_exports_every_declared_symbol():
    from hulc2_amd import build, lib

    build.build(verbose=False)
    header = (ROOT / "include" / "hulc2_amd.h").read_text()
    names = sorted(set(re.findall(r"\b(hulc_[a-z0-9_]+)\s*\(", header)))
    assert len(names) >= 30
    so = ctypes.CDLL(str(lib.lib_path()))
    missing = [n for n in names if not hasattr(so, n)]
    assert not missing, missing
    assert so.hulc_abi_version() == 7


def test_product_never_imports_oracle():
    for p in (ROOT / "hulc2_amd").rglob("*.py"):
        txt = p.read_text()
        assert "import oracle" not in txt and "from oracle" not in txt, p


def test_synthetic_recipe_is_order_independent():
    a = {"x.weight": torch.empty(4, 3), "y.bias": torch.empty(5)}
    b = {"y.bias": torch.empty(5), "x.weight": torch.empty(4, 3)}
    syn.fill_state_dict_(a, 3)
    syn.fill_state_dict_(b, 3)
    assert torch.equal(a["x.weight"], b["x.weight"]) and torch.equal(a["y.bias"], b["y.bias"])


def test_real_world_config_resolves_r3m_encoder():
    """cfg_low_level_rw (BASELINE configs[3]): `hulc2.models.perceptual_encoders.vision_r3m.VisionR3M` resolves here, its state_dict uses
    torchvision's ResNet names under `r3m.convnet.` (what a reference checkpoint holds), the trunk is frozen, the decoder consumes the
    whole perceptual embedding and there is no CLIP head."""
    from hulc2_amd.config import real_world_model_config

    install_as_hulc2()
    import hulc2.models.perceptual_encoders.vision_r3m as ref_path
    import hulc2.affordance.models.language_encoders.sbert_lang_encoder as sbert_path

    m = instantiate(real_world_model_config())
    enc = m.perceptual_encoder.rgb_static_encoder
    assert type(enc) is ref_path.VisionR3M and hasattr(sbert_path, "SBertLang")
    sd = enc.state_dict()
    assert len(sd) == 124                                                    # torchvision resnet18 minus fc (120) + fc1, fc2
    for k, shp in {"r3m.convnet.conv1.weight": (64, 3, 7, 7), "r3m.convnet.bn1.running_var": (64,), "r3m.convnet.layer1.0.conv1.weight": (64, 64, 3, 3),
                   "r3m.convnet.layer2.0.downsample.0.weight": (128, 64, 1, 1), "r3m.convnet.layer2.0.downsample.1.num_batches_tracked": (),
                   "r3m.convnet.layer4.1.bn2.bias": (512,), "fc1.weight": (256, 512), "fc2.weight": (64, 256)}.items():
        assert tuple(sd[k].shape) == shp, k
    assert "r3m.convnet.layer1.0.downsample.0.weight" not in sd and not any(k.startswith("r3m.convnet.fc") for k in sd)
    assert not any(p.requires_grad for p in enc.r3m.parameters())
    assert sum(p.numel() for p in enc.r3m.parameters()) == 11_176_512         # torchvision resnet18 without its classifier
    assert m.action_decoder.perceptual_emb_slice == (0, 128) and not m.action_decoder.gripper_control
    assert not m.use_clip_auxiliary_loss and not hasattr(m, "proj_vis_lang") or m.proj_vis_lang is None
    with pytest.raises(NotImplementedError):
        from hulc2_amd.models.perceptual_encoders.vision_r3m import VisionR3M
        VisionR3M(None, 64, resnet_model="resnet50")


# ---- the C structs of include/hulc2_amd.h against their ctypes mirrors and the binding shown in INTEGRATION.md ----------------------------
_C2CT = {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "unsigned long long": ctypes.c_ulonglong}


def _parse_struct(text: str, name: str):
    """-> [(field, ctypes type)] of `typedef struct name { ... } name;` — handles `const T *a, *b;`, `T a, b;`, `T* a; long b, c;` on one line"""
    end = re.search(r"\}\s*" + name + r"\s*;", text)
    assert end, f"struct {name} not found"
    start = text.rfind("typedef struct", 0, end.start())
    body = re.sub(r"/\*.*?\*/", "", text[text.index("{", start) + 1:end.start()], flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        m2 = re.match(r"(?:const\s+)?(unsigned long long|void|float|int|long|unsigned)\s*(.*)$", decl, re.S)
        assert m2, decl
        base, rest = m2.group(1), m2.group(2)
        for item in rest.split(","):
            item = item.strip()
            ptr = "*" in item
            fields.append((item.replace("*", "").strip(), ctypes.c_void_p if ptr else _C2CT[base]))
    return fields


@pytest.mark.parametrize("cname,pyname", [("hulc_gemm_desc", "GemmDesc"), ("hulc_conv_desc", "ConvDesc"), ("hulc_rnn_wave_desc", "RnnWaveDesc"),
                                          ("hulc_mix_desc", "MixDesc"), ("hulc_txl_attn_desc", "TxlAttnDesc"), ("hulc_wgrad_item", "WgradItem"),
                                          ("hulc_mlp_chain_layer", "MlpChainLayer")])
def test_ctypes_structures_mirror_the_header(cname, pyname):
    """field count, order, names and C types of every descriptor struct == its ctypes.Structure in hulc2_amd/lib.py: a field added on one
    side only makes the kernel read past the caller's struct (the INTEGRATION.md stub was 3 fields short)"""
    from hulc2_amd import lib
    header = (ROOT / "include" / "hulc2_amd.h").read_text()
    want = _parse_struct(header, cname)
    got = [(n, t) for n, t in getattr(lib, pyname)._fields_]
    assert [n for n, _ in got] == [n for n, _ in want], (cname, [n for n, _ in got], [n for n, _ in want])
    for (n, tg), (_, tw) in zip(got, want):
        assert tg is tw, (cname, n, tg, tw)


def test_integration_doc_binding_lists_every_gemm_field():
    """INTEGRATION.md shows the ctypes binding a maintainer would write; its GemmDesc must carry all fields of hulc_gemm_desc in order"""
    from hulc2_amd import lib
    doc = (ROOT / "INTEGRATION.md").read_text()
    m = re.search(r"class GemmDesc\(ctypes\.Structure\):(.*?)\n```", doc, re.S)
    assert m, "INTEGRATION.md no longer shows the GemmDesc binding"
    names = re.findall(r'\("([A-Za-z_0-9]+)",\s*ctypes\.', m.group(1))
    assert names == [n for n, _ in lib.GemmDesc._fields_], (names, [n for n, _ in lib.GemmDesc._fields_])
    assert "raise `NotImplementedError`" not in doc or "validation" not in doc.split("raise `NotImplementedError`")[0][-200:]


def test_fused_heads_follow_the_configured_mixture_count():
    """the fused head layout is derived from out_features x n_mixtures, not assumed to be 6 x 10"""
    from hulc2_amd.models.decoders.logistic_decoder_rnn import LogisticDecoderRNN
    for n_mix, out in ((10, 7), (5, 7), (3, 4)):
        dec = LogisticDecoderRNN(perceptual_features=128, latent_goal_features=32, plan_features=1024, n_mixtures=n_mix, hidden_size=64,
                                 out_features=out, log_scale_min=-7.0, act_max_bound=[1.0] * out, act_min_bound=[-1.0] * out, dataset_dir="",
                                 load_action_bounds=False, num_classes=10, gripper_alpha=1.0, perceptual_emb_slice=[64, 128],
                                 policy_rnn_dropout_p=0.0, num_layers=2, rnn_model="rnn_decoder", gripper_control=False, discrete_gripper=True)
        for g in dec.fused_param_groups():
            members = sum(p.numel() for p in g["params"])
            n = 1
            for d in g["shape"]:
                n *= d
            assert members + g["pad"] == n, (n_mix, out, g["attr"])              # the view spans exactly the members + the zero padding
            assert g["shape"][0] % 8 == 0 and g["shape"][0] - (3 * (out - 1) * n_mix + 2) == (-(3 * (out - 1) * n_mix + 2)) % 8


def test_trainer_optimizer_state_roundtrip_and_reload_hook():
    """ArenaTrainer.state_dict / load_state_dict carry exp_avg / exp_avg_sq / step by parameter NAME, and a model.load_state_dict
    after the trainer exists keeps the parameters inside the arena (the post-hook re-derives the kernel-side shadows on the GPU)"""
    from hulc2_amd.trainer import ArenaTrainer
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(6, 9), torch.nn.ReLU(), torch.nn.Linear(9, 2))
    tr = ArenaTrainer(m)
    tr.exp_avg.normal_()
    tr.exp_avg_sq.uniform_()
    tr.step_count = 17
    sd = tr.state_dict()
    assert set(sd["state"]) == {n for n, _ in m.named_parameters()} and sd["step"] == 17
    m2 = torch.nn.Sequential(torch.nn.Linear(6, 9), torch.nn.ReLU(), torch.nn.Linear(9, 2))
    tr2 = ArenaTrainer(m2)
    tr2.load_state_dict(sd)
    assert tr2.step_count == 17
    for p, off in zip(tr2.params, tr2.offsets):                                  # (alignment padding between parameters carries no state)
        sl = slice(off, off + p.numel())
        assert torch.equal(tr2.exp_avg[sl], tr.exp_avg[sl]) and torch.equal(tr2.exp_avg_sq[sl], tr.exp_avg_sq[sl])
    m2.load_state_dict(m.state_dict())                                           # weights restored AFTER the trainer was built
    for p, off in zip(tr2.params, tr2.offsets):
        assert p.data_ptr() == tr2.flat_p.data_ptr() + 4 * off
    assert all(torch.equal(a, b) for a, b in zip(m2.parameters(), m.parameters()))
    with pytest.raises(KeyError):
        tr2.load_state_dict({**sd, "state": {k: v for k, v in list(sd["state"].items())[:-1]}})


def test_wgrad_group_workspace_counts_the_nine_tap_slabs():
    """hulc_wgrad_group_workspace is host arithmetic (no GPU): a conv_taps_wp item that wgrad_taps.hip takes (bf16, 64-multiples) adds its own slab
    region — tiles x slices x (4 blocks x 9 taps x 1024) floats — behind the grouped kernel's counters; an unsplit one (few rows, many tiles) adds none"""
    import ctypes
    from hulc2_amd import lib as L
    lib = L.load()
    lib.hulc_wgrad_group_workspace.restype = ctypes.c_long
    buf = (ctypes.c_char * 64)()                                   # any 16-byte aligned, non-null address: nothing is dereferenced
    base = ctypes.addressof(buf) // 16 * 16 + 16

    def item(M, N, K, wp, bf16=True):
        it = L.WgradItem()
        it.A, it.B, it.C = base, base, base
        it.M, it.N, it.K, it.lda, it.ldb, it.ldc = M, N, K, M, N, N * 9
        it.a_dtype = it.b_dtype = 1 if bf16 else 0               # HULC_BF16 / HULC_F32 (include/hulc2_amd.h)
        it.col_mul, it.conv_taps_wp = 9, wp
        return it

    counters = 65536 * 4
    one = (L.WgradItem * 1)(item(512, 768, 8192, 16))              # 96 tiles, 128 k-steps: one slice per tile, no slabs
    assert lib.hulc_wgrad_group_workspace(one, 1) == counters
    split = (L.WgradItem * 1)(item(128, 192, 107648, 58))           # 6 tiles, 1682 k-steps -> 27 slices of 63
    assert lib.hulc_wgrad_group_workspace(split, 1) == counters + 6 * 27 * (4 * 9 * 16 * 64) * 4
    f32 = (L.WgradItem * 1)(item(128, 192, 107648, 58, bf16=False))  # fp32 operands stay with the grouped kernel: nine tiles per slice
    assert lib.hulc_wgrad_group_workspace(f32, 1) > counters and lib.hulc_wgrad_group_workspace(f32, 1) != lib.hulc_wgrad_group_workspace(split, 1)


def test_optimizer_state_interchanges_with_torch_adam():
    """a reference Lightning checkpoint's `optimizer_states[0]` is torch.optim.Adam.state_dict() — state keyed by the parameter's
    INDEX in model.parameters() order, per-parameter `step`, `param_groups[0]["params"]`.  to_/from_torch_adam_state_dict convert both ways;
    checked against torch.optim.Adam.load_state_dict / state_dict themselves, with a frozen parameter in the middle of the order."""
    from hulc2_amd.trainer import ArenaTrainer
    torch.manual_seed(1)

    def net():
        m = torch.nn.Sequential(torch.nn.Linear(6, 9), torch.nn.ReLU(), torch.nn.Linear(9, 5), torch.nn.ReLU(), torch.nn.Linear(5, 2))
        m[2].bias.requires_grad_(False)                    # model.parameters() still lists it; torch keeps no state for it
        return m

    # reference side: three torch.optim.Adam steps
    ref = net()
    opt = torch.optim.Adam(ref.parameters(), lr=2e-4)
    for i in range(3):
        opt.zero_grad()
        ref(torch.randn(4, 6)).square().sum().backward()
        opt.step()
    sd = opt.state_dict()

    m = net()
    m.load_state_dict(ref.state_dict())
    tr = ArenaTrainer(m)
    tr.from_torch_adam_state_dict(sd)
    assert tr.step_count == 3 and tr.lr == 2e-4 and tuple(tr.betas) == (0.9, 0.999)
    order = list(m.parameters())
    index = {id(p): off for p, off in zip(tr.params, tr.offsets)}
    for i, p in enumerate(order):
        if not p.requires_grad:
            assert i not in sd["state"]
            continue
        sl = slice(index[id(p)], index[id(p)] + p.numel())
        assert torch.equal(tr.exp_avg[sl].view(p.shape), sd["state"][i]["exp_avg"])
        assert torch.equal(tr.exp_avg_sq[sl].view(p.shape), sd["state"][i]["exp_avg_sq"])

    # and back: torch's own optimizer accepts what the trainer exports, and re-exports the same tensors
    out = tr.to_torch_adam_state_dict()
    assert set(out) == {"state", "param_groups"} and out["param_groups"][0]["params"] == list(range(len(order)))
    assert set(out["state"]) == set(sd["state"])
    opt2 = torch.optim.Adam(net().parameters(), lr=1.0)
    opt2.load_state_dict(out)
    back = opt2.state_dict()
    assert back["param_groups"][0]["lr"] == 2e-4
    for i, rec in sd["state"].items():
        assert float(back["state"][i]["step"]) == float(rec["step"]) == 3.0
        assert torch.equal(back["state"][i]["exp_avg"], rec["exp_avg"]) and torch.equal(back["state"][i]["exp_avg_sq"], rec["exp_avg_sq"])
    # the next torch step from the exported state equals the next torch step of the original optimizer
    x = torch.randn(4, 6)
    m3 = net()
    m3.load_state_dict(ref.state_dict())
    opt3 = torch.optim.Adam(m3.parameters(), lr=2e-4)
    opt3.load_state_dict(out)
    for mod, o in ((ref, opt), (m3, opt3)):
        o.zero_grad()
        mod(x).square().sum().backward()
        o.step()
    assert all(torch.equal(a, b) for a, b in zip(ref.parameters(), m3.parameters()))
    # a fresh trainer exports an empty state (torch creates state lazily), mismatched layouts raise
    assert ArenaTrainer(net()).to_torch_adam_state_dict()["state"] == {}
    with pytest.raises(KeyError):
        tr.from_torch_adam_state_dict({"state": {}, "param_groups": [{**sd["param_groups"][0], "params": [0, 1]}]})
    bad = {"state": {k: dict(v) for k, v in sd["state"].items()}, "param_groups": sd["param_groups"]}
    bad["state"][0]["step"] = torch.tensor(7.0)
    with pytest.raises(ValueError):
        tr.from_torch_adam_state_dict(bad)


def test_a_second_trainer_retires_the_first_ones_load_hook():
    """the load_state_dict post-hook of an earlier trainer kept that trainer (four arenas) alive and re-homed the weights into its
    dead arena on every load.  Building a new trainer closes the old one; close() removes the hook."""
    import gc
    import weakref
    from hulc2_amd.trainer import ArenaTrainer
    m = torch.nn.Sequential(torch.nn.Linear(4, 3))
    t1 = ArenaTrainer(m)
    ref1 = weakref.ref(t1)
    t2 = ArenaTrainer(m)
    assert t1._load_hook is None and t2._load_hook is not None
    assert len(m._load_state_dict_post_hooks) == 1
    del t1
    gc.collect()
    assert ref1() is None                                   # nothing (no hook, no module attribute) holds the first trainer any more
    m.load_state_dict({k: v.clone() + 1 for k, v in m.state_dict().items()})
    for p, off in zip(t2.params, t2.offsets):
        assert p.data_ptr() == t2.flat_p.data_ptr() + 4 * off
    t2.close()
    assert len(m._load_state_dict_post_hooks) == 0


def test_ffn_fragment_permutations():
    """hulc_ffn_frag_perm (host function of the C ABI): each layout is a permutation of the FF x 128 weight elements that moves runs of 4
    consecutive source elements (the trainer's gather launch copies 8-byte chunks), and layout 0 keeps a fragment's 8 k-slots consecutive"""
    import numpy as np
    from hulc2_amd import kernels as kn
    for ff in (128, 2048):
        for n in range(4):
            p = kn.ffn_frag_perm(n, ff).astype(np.int64)
            assert p.shape == (ff * 128,) and np.array_equal(np.sort(p), np.arange(ff * 128))
            q = p.reshape(-1, 4)
            assert (q[:, 0] % 4 == 0).all() and (q[:, 1:] - q[:, :1] == [1, 2, 3]).all()
        a = kn.ffn_frag_perm(0, ff).reshape(-1, 8)
        assert (a[:, 1:] - a[:, :1] == np.arange(1, 8)).all()
        # layout 1, hidden block hb, fragment f = 2 ot + kk, lane (r, hf): W2[32 ot + r][32 hb + 16 kk + 4 hf + {0..3, 8..11}]
        b = kn.ffn_frag_perm(1, ff).reshape(ff // 32, 8, 64, 8)
        hb, f, lane = ff // 32 - 1, 5, 37
        r, hf = lane & 31, lane >> 5
        want = [(32 * (f >> 1) + r) * ff + 32 * hb + 16 * (f & 1) + 4 * hf + (j if j < 4 else j + 4) for j in range(8)]
        assert b[hb, f, lane].tolist() == want


def test_precision_sites_are_validated(monkeypatch):
    """HULC_FP32_SITES (selective precision): the default, 'none', and a misspelt site name"""
    from hulc2_amd import kernels as kn
    monkeypatch.delenv("HULC_FP32_SITES", raising=False)
    assert kn.fp32_sites() == {"head", "goal", "encfc", "txl"}
    monkeypatch.setenv("HULC_FP32_SITES", "none")
    assert kn.fp32_sites() == frozenset()
    monkeypatch.setenv("HULC_FP32_SITES", "head, conv1 ,a3")
    assert kn.fp32_sites() == {"head", "conv1", "a3"}
    monkeypatch.setenv("HULC_FP32_SITES", "head,tlx")
    with pytest.raises(ValueError):
        kn.fp32_sites()
