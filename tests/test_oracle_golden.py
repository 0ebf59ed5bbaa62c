"""Pin the CPU oracle (oracle/hulc2_oracle.py) against fixtures produced by the reference's own leaf
modules (oracle/gen_golden.py, run in the build container).  CPU only, fp32, tolerance 2e-5 rel
(different op order only: explicit transformer / RNN loops vs torch's fused modules)."""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from hulc2_amd import param_spec, synthetic as syn  # noqa: E402
from oracle import hulc2_oracle as O  # noqa: E402

G = ROOT / "tests" / "golden"
RTOL = 2e-5


def load(name):
    return dict(np.load(G / f"{name}.npz", allow_pickle=False))


def close(a, b, rtol=RTOL, what=""):
    a = torch.as_tensor(np.asarray(a)).double() if not isinstance(a, torch.Tensor) else a.detach().double()
    b = torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    scale = b.abs().max().item() + 1e-12
    err = (a - b).abs().max().item()
    assert err <= rtol * scale + 1e-7, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


@pytest.fixture(scope="module")
def sd():
    seed = int(load("vision_static")["seed"])
    d = {k: torch.empty(s) for k, s in param_spec.trainable_shapes().items()}
    syn.fill_state_dict_(d, seed)
    for v in d.values():
        v.requires_grad_(True)
    return d


def check_recipe(sd, fx):
    """the (name, seed) recipe must regenerate exactly the parameters the reference ran with"""
    cs = syn.checksum(sd)
    for k, (s, a) in zip(fx["ck"], fx["cv"]):
        k = str(k)
        if k in cs:
            assert abs(cs[k][0] - s) <= 1e-9 * max(1, abs(s)) and abs(cs[k][1] - a) <= 1e-9 * max(1, a), f"recipe drift in {k}"


def zero(sd):
    for v in sd.values():
        v.grad = None


def test_param_count():
    assert param_spec.num_trainable() == 47_050_727 or abs(param_spec.num_trainable() - 47.05e6) < 0.02e6


@pytest.mark.parametrize("tag,hw", [("vision_static", 200), ("vision_gripper", 84)])
def test_vision(sd, tag, hw):
    fx = load(tag)
    check_recipe(sd, fx)
    zero(sd)
    seed, n = int(fx["seed"]), int(fx["n"])
    x = torch.rand(n, 3, hw, hw, generator=syn._gen(seed, "x." + tag)) * 2 - 1
    p = "perceptual_encoder.rgb_static_encoder." if tag == "vision_static" else "perceptual_encoder.rgb_gripper_encoder."
    out = (O.vision_network_static if tag == "vision_static" else O.vision_network_gripper)(sd, p, x)
    close(out, fx["out"], what="out")
    close(O._conv_stack(sd, p, x)[0], fx["conv3_frame0"], what="conv3")
    if tag == "vision_static":
        close(O.spatial_softmax(O._conv_stack(sd, p, x)), fx["ssm"], what="spatial softmax")
    r = torch.randn(out.shape, generator=syn._gen(seed, "r." + tag))
    (out * r).sum().backward()
    close(sd[p + "conv_model.0.weight"].grad, fx["g_conv0_w"], 1e-4, "g conv0 w")
    close(sd[p + "conv_model.0.bias"].grad, fx["g_conv0_b"], 1e-4, "g conv0 b")
    close(sd[p + "conv_model.2.weight"].grad[::4, ::4], fx["g_conv2_w_s"], 1e-4, "g conv2 w")
    close(sd[p + "conv_model.4.weight"].grad[::4, ::4], fx["g_conv4_w_s"], 1e-4, "g conv4 w")
    close(sd[p + "fc2.weight"].grad, fx["g_fc2_w"], 1e-4, "g fc2 w")
    close(sd[p + "ln.weight"].grad, fx["g_ln_w"], 1e-4, "g ln w")
    if tag == "vision_gripper":
        close(sd[p + "conv_model.7.weight"].grad[::4, ::16], fx["g_fc0_w_s"], 1e-4, "g flatten-fc w")


def test_goal_encoders_and_proposal(sd):
    fx = load("goal_encoders")
    check_recipe(sd, fx)
    zero(sd)
    seed, B = int(fx["seed"]), int(fx["B"])
    xv = torch.randn(B, 128, generator=syn._gen(seed, "x.visual_goal")).requires_grad_()
    xl = (torch.randn(B, 384, generator=syn._gen(seed, "x.language_goal")) * 0.05).requires_grad_()
    ov, ol = O.visual_goal_encoder(sd, "visual_goal.", xv), O.language_goal_encoder(sd, "language_goal.", xl)
    close(ov, fx["out_vis"], what="visual goal")
    close(ol, fx["out_lang"], what="language goal")
    rv = torch.randn(B, 32, generator=syn._gen(seed, "r.visual_goal"))
    rl = torch.randn(B, 32, generator=syn._gen(seed, "r.language_goal"))
    ((ov * rv).sum() + (ol * rl).sum()).backward()
    close(xv.grad, fx["gx_vis"], 1e-4, "gx vis")
    close(xl.grad, fx["gx_lang"], 1e-4, "gx lang")
    close(sd["visual_goal.mlp.4.weight"].grad, fx["g_vis_mlp4_w"], 1e-4, "g mlp4")
    close(sd["language_goal.mlp.1.weight"].grad[::16], fx["g_lang_mlp1_w_s"], 1e-4, "g mlp1")

    fx = load("plan_proposal")
    check_recipe(sd, fx)
    zero(sd)
    e0 = torch.randn(B, 128, generator=syn._gen(seed, "x.plan_proposal.emb")).requires_grad_()
    gl = torch.randn(B, 32, generator=syn._gen(seed, "x.plan_proposal.goal")).requires_grad_()
    lg = O.plan_proposal(sd, "plan_proposal.", e0, gl)
    close(lg, fx["logits"], what="prior logits")
    (lg * torch.randn(B, 1024, generator=syn._gen(seed, "r.plan_proposal"))).sum().backward()
    close(e0.grad, fx["g_emb"], 1e-4, "g emb")
    close(gl.grad, fx["g_goal"], 1e-4, "g goal")
    close(sd["plan_proposal.fc_state.0.weight"].grad[::8, ::8], fx["g_state_w_s"], 1e-4, "g fc_state")


@pytest.mark.parametrize("S", [16, 32])
def test_plan_recognition(sd, S):
    fx = load(f"plan_recognition_S{S}")
    check_recipe(sd, fx)
    zero(sd)
    seed, B = int(fx["seed"]), int(fx["B"])
    x = torch.randn(B, S, 128, generator=syn._gen(seed, f"x.plan_recognition.{S}")).requires_grad_()
    lg, feat = O.plan_recognition(sd, "plan_recognition.", x)
    assert np.array_equal(fx["position_ids"], np.arange(S))          # integer indexing: bit-exact
    close(lg, fx["logits"], what="posterior logits")
    close(feat, fx["seq_feat"], what="seq_feat")
    r1 = torch.randn(B, 1024, generator=syn._gen(seed, f"r1.plan_recognition.{S}"))
    r2 = torch.randn(B, 4096, generator=syn._gen(seed, f"r2.plan_recognition.{S}"))
    ((lg * r1).sum() + (feat * r2).sum()).backward()
    close(x.grad, fx["gx"], 1e-4, "gx")
    p = "plan_recognition."
    close(sd[p + "position_embeddings.weight"].grad, fx["g_pos"], 1e-4, "g pos")
    close(sd[p + "transformer_encoder.layers.0.self_attn.in_proj_weight"].grad, fx["g_inproj_w"], 1e-4, "g in_proj")
    close(sd[p + "transformer_encoder.layers.1.self_attn.out_proj.weight"].grad, fx["g_outproj_w"], 1e-4, "g out_proj")
    close(sd[p + "transformer_encoder.layers.0.norm1.weight"].grad, fx["g_norm1_w"], 1e-4, "g norm1")
    close(sd[p + "fc.bias"].grad, fx["g_fc_b"], 1e-4, "g fc b")


def test_distribution_and_kl():
    fx = load("distribution_kl")
    pp = torch.tensor(fx["pp"]).requires_grad_()
    pr = torch.tensor(fx["pr"]).requires_grad_()
    idx = torch.tensor(fx["idx"])
    kl = O.kl_loss(pp, pr, 0.01, 0.8)
    plan = O.straight_through_sample(pr, idx)
    close(kl, fx["kl"], what="kl")
    assert torch.equal(plan.detach().argmax(-1) if False else plan.detach().reshape(-1, 32, 32).argmax(-1), idx)  # bit-exact indices
    close(plan, fx["plan"], what="plan")
    (kl + (plan * torch.tensor(fx["r"])).sum() * 1e-3).backward()
    close(pp.grad, fx["g_pp"], 1e-4, "g pp")
    close(pr.grad, fx["g_pr"], 1e-4, "g pr")


@pytest.mark.parametrize("S", [16, 32])
def test_decoder(sd, S):
    fx = load(f"decoder_S{S}")
    check_recipe(sd, fx)
    zero(sd)
    seed, B = int(fx["seed"]), int(fx["B"])
    idx = torch.randint(0, 32, (B, 32), generator=syn._gen(seed, f"x.dec.idx.{S}"))
    plan = torch.nn.functional.one_hot(idx, 32).float().flatten(1).requires_grad_()
    emb = torch.randn(B, S, 128, generator=syn._gen(seed, f"x.dec.emb.{S}")).requires_grad_()
    goal = torch.randn(B, 32, generator=syn._gen(seed, f"x.dec.goal.{S}")).requires_grad_()
    acts = torch.tensor(fx["acts"])
    lp, ls, mu, grip = O.decoder_forward(sd, "action_decoder.", plan, emb, goal)
    close(lp, fx["logit_probs"], what="logit_probs")
    close(ls, fx["log_scales"], what="log_scales")
    close(mu, fx["means"], what="means")
    close(grip, fx["grip"], what="grip")
    loss = O.decoder_loss(lp, ls, mu, grip, acts)
    close(loss, fx["loss"], what="loss")
    loss.backward()
    close(plan.grad, fx["g_plan"], 2e-4, "g plan")
    close(emb.grad, fx["g_emb"], 2e-4, "g emb")
    close(goal.grad, fx["g_goal"], 2e-4, "g goal")
    close(sd["action_decoder.rnn.weight_hh_l0"].grad[::16, ::16], fx["g_whh0_s"], 2e-4, "g whh0")
    close(sd["action_decoder.rnn.weight_ih_l1"].grad[::16, ::16], fx["g_wih1_s"], 2e-4, "g wih1")
    close(sd["action_decoder.rnn.bias_ih_l0"].grad, fx["g_bih0"], 2e-4, "g bih0")
    close(sd["action_decoder.gripper_fc.weight"].grad, fx["g_grip_w"], 2e-4, "g gripper_fc")


def test_logistic_mixture_edges():
    fx = load("logistic_mixture_edges")
    lp, ls, mu, grip = (torch.tensor(fx[k]).requires_grad_() for k in ("logit_probs", "log_scales", "means", "grip"))
    acts = torch.tensor(fx["acts"])
    loss = O.decoder_loss(lp, ls, mu, grip, acts)
    close(loss, fx["loss"], what="loss")
    loss.backward()
    for t, k in ((lp, "g_lp"), (ls, "g_ls"), (mu, "g_mu"), (grip, "g_grip")):
        close(t.grad, fx[k], 1e-4, k)
    g = acts[..., 6].clone()
    g[g == -1] = 0
    assert np.array_equal(g.long().numpy(), fx["gripper_labels"])    # integer labels: bit-exact


def test_clip_loss(sd):
    fx = load("clip_loss")
    check_recipe(sd, fx)
    zero(sd)
    seed, B = int(fx["seed"]), int(fx["B"])
    feat = torch.randn(B, 4096, generator=syn._gen(seed, "x.clip.feat")).requires_grad_()
    goal = torch.randn(B, 32, generator=syn._gen(seed, "x.clip.goal")).requires_grad_()
    loss = O.clip_auxiliary_loss(sd, feat, goal, torch.tensor(fx["use"]))
    close(loss, fx["loss"], what="clip loss")
    loss.backward()
    close(feat.grad, fx["g_feat"], 1e-4, "g feat")
    close(goal.grad, fx["g_goal"], 1e-4, "g goal")
    close(sd["logit_scale"].grad, fx["g_logit_scale"], 1e-4, "g logit_scale")


def _oracle_batch(seed, B, S):
    b = syn.make_batch(seed, B, S)
    out = {}
    for m, db in b.items():
        out[m] = dict(rgb_static=db["rgb_obs"]["rgb_static"], rgb_gripper=db["rgb_obs"]["rgb_gripper"],
                      actions=db["actions"], robot_obs=db["state_info"]["robot_obs"], plan_idx=db["plan_idx"])
        if m == "lang":
            out[m]["lang"] = db["lang"]
            out[m]["use_for_aux_lang_loss"] = db["use_for_aux_lang_loss"]
    return out


@pytest.mark.parametrize("B,S", [(2, 16), (2, 32)])
def test_whole_step(sd, B, S):
    fx = load(f"step_B{B}_S{S}")
    check_recipe(sd, fx)
    zero(sd)
    r = O.training_step(sd, _oracle_batch(int(fx["seed"]), B, S), dict(gripper_control=False))
    for k in ("kl_loss", "action_loss", "clip_loss", "total_loss"):
        close(r[k], fx[k], 5e-5, k)
    for m in ("vis", "lang"):
        close(r[f"emb_{m}"], fx[f"emb_{m}"], 5e-5, f"emb {m}")
        close(r[f"goal_{m}"], fx[f"goal_{m}"], 5e-5, f"goal {m}")
        close(r[f"pp_{m}"], fx[f"pp_{m}"], 5e-5, f"pp {m}")
        close(r[f"pr_{m}"], fx[f"pr_{m}"], 5e-5, f"pr {m}")
        close(r[f"seq_feat_{m}"][:, ::8], fx[f"seq_feat_{m}"], 5e-5, f"seq_feat {m}")
    r["total_loss"].backward()
    names = [str(n) for n in fx["grad_names"]]
    for n, ref in zip(names, fx["grad_norms"]):
        if ref < 0:      # parameter the reference never touches (plan_recognition.layernorm: positional_normalize=False)
            assert sd[n].grad is None, f"{n} must stay without gradient"
            continue
        got = sd[n].grad.double().norm().item() if n != "logit_scale" else sd[n].grad.abs().item()
        assert abs(got - ref) <= 2e-4 * max(ref, 1e-6) + 1e-9, f"grad norm {n}: {got} vs {ref}"
    close(sd["perceptual_encoder.rgb_static_encoder.conv_model.0.weight"].grad, fx["g_conv0_w_static"], 2e-4, "g conv0 static")
    close(sd["plan_recognition.position_embeddings.weight"].grad, fx["g_pos"], 2e-4, "g pos")


def test_affordance_trunk_maps_agree_with_the_pooled_restatement():
    """oracle/affordance_oracle.trunk_maps (the five maps the U-Net decoder reads) against hulc2_oracle.r3m_trunk_features (the pooled feature
    of the same ResNet-18 restatement; both parity unpinned: r3m is un-vendored) + the map shapes r3m_rn18.py:59 promises"""
    from oracle import affordance_oracle as A
    from hulc2_amd import synthetic as syn
    sd = {"conv1.weight": (64, 3, 7, 7)}
    def bn(q, c):
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            sd[f"{q}.{leaf}"] = (c,)
    bn("bn1", 64)
    cin = 64
    for li, c in enumerate((64, 128, 256, 512), start=1):
        for b in range(2):
            q = f"layer{li}.{b}."
            sd[q + "conv1.weight"] = (c, cin if b == 0 else c, 3, 3); bn(q + "bn1", c)
            sd[q + "conv2.weight"] = (c, c, 3, 3); bn(q + "bn2", c)
            if b == 0 and li > 1:
                sd[q + "downsample.0.weight"] = (c, cin, 1, 1); bn(q + "downsample.1", c)
        cin = c
    sd = {"r3m.convnet." + k: torch.empty(v) for k, v in sd.items()}
    syn.fill_state_dict_(sd, 5)
    x = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(1)) * 255
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    maps = A.trunk_maps(sd, (x / 255.0 - mean) / std)
    assert [tuple(m.shape[1:]) for m in maps] == [(64, 16, 16), (64, 16, 16), (128, 8, 8), (256, 4, 4), (512, 2, 2)]
    assert torch.allclose(maps[-1].mean(dim=(2, 3)), O.r3m_trunk_features(sd, x), atol=1e-5, rtol=1e-5)
    assert all(torch.isfinite(m).all() for m in maps)


def test_euler_convention_against_scipy():
    """pytorch3d is absent, so the restated euler_angles_to_matrix / matrix_to_euler_angles ('XYZ') are held against an INDEPENDENT
    implementation of the convention pytorch3d documents (R = Rx(a) Ry(b) Rz(c), intrinsic XYZ): scipy.spatial.transform.Rotation.
    Still not the reference's own dependency (row a16 stays "unpinned"), but no longer only self-consistent.  The whole frame change
    (gripper_control.py:16-36) is then rebuilt from scipy objects and compared."""
    from scipy.spatial.transform import Rotation as Rot
    rng = np.random.default_rng(3)
    e = (rng.random((64, 3)) * 2 - 1) * np.array([3.0, 1.4, 3.0])
    R = O.euler_xyz_to_matrix(torch.from_numpy(e)).numpy()
    np.testing.assert_allclose(R, Rot.from_euler("XYZ", e).as_matrix(), atol=1e-12)
    back = O.matrix_to_euler_xyz(torch.from_numpy(R)).numpy()
    np.testing.assert_allclose(back, Rot.from_matrix(R).as_euler("XYZ"), atol=1e-9)
    act = rng.random((4, 16, 7)) * 2 - 1
    obs = rng.standard_normal((4, 16, 15))
    obs[..., 3:6] = (rng.random((4, 16, 3)) * 2 - 1) * np.array([3.0, 1.4, 3.0])
    got = O.world_to_tcp_frame(torch.from_numpy(act).float(), torch.from_numpy(obs).float()).numpy().reshape(-1, 7)
    a2, o2 = act.reshape(-1, 7), obs.reshape(-1, 15)
    w_T_tcp = Rot.from_euler("XYZ", o2[:, 3:6])
    pos = w_T_tcp.inv().apply(a2[:, :3])
    new = Rot.from_euler("XYZ", o2[:, 3:6] + 0.01 * a2[:, 3:6])
    rel = (new.inv() * w_T_tcp).as_euler("XYZ") * 100
    np.testing.assert_allclose(got[:, :3], pos, atol=2e-6)
    np.testing.assert_allclose(got[:, 3:6], rel, atol=2e-2)          # float32 rotation products, x 100
    np.testing.assert_array_equal(got[:, 6], a2[:, 6].astype(np.float32))


def test_world_to_tcp_properties():
    """pytorch3d-backed frame change: parity unpinned -> properties (gripper_control.py:16-63)."""
    g = torch.Generator().manual_seed(0)
    act = torch.rand(3, 5, 7, generator=g) * 2 - 1
    obs = torch.randn(3, 5, 15, generator=g)
    obs[..., 3:6] = (torch.rand(3, 5, 3, generator=g) * 2 - 1) * 1.2
    R = O.euler_xyz_to_matrix(obs[..., 3:6])
    eye = torch.eye(3).expand_as(R)
    assert torch.allclose(R @ R.transpose(-1, -2), eye, atol=1e-5)
    assert torch.allclose(O.matrix_to_euler_xyz(R), obs[..., 3:6], atol=1e-4)
    tcp = O.world_to_tcp_frame(act, obs)
    back = O.tcp_to_world_frame(tcp, obs)
    assert torch.allclose(back, act, atol=2e-3)
    obs0 = obs.clone()
    obs0[..., 3:6] = 0
    z = O.world_to_tcp_frame(act, obs0)
    # at zero tcp rotation positions are unchanged and tcp_new_T_tcp_old = R(0.01*orn)^-1, i.e. the
    # relative euler angles flip sign (to first order in the 0.01 down-scaling)
    assert torch.allclose(z[..., :3], act[..., :3], atol=1e-6)
    assert torch.allclose(z[..., 3:6], -act[..., 3:6], atol=1e-2)
    assert torch.equal(z[..., 6], act[..., 6])


# ---- validation / rollout pieces --------------------------------------------------
def _inference_inputs(seed, tag, B, S):
    idx = torch.randint(0, 32, (B, 32), generator=syn._gen(seed, f"x.{tag}.idx"))
    plan = torch.nn.functional.one_hot(idx, 32).float().flatten(1)
    emb = torch.randn(B, S, 128, generator=syn._gen(seed, f"x.{tag}.emb"))
    goal = torch.randn(B, 32, generator=syn._gen(seed, f"x.{tag}.goal"))
    return plan, emb, goal


def test_decoder_forward_with_carried_state(sd):
    """LogisticDecoderRNN.forward(h_0) (the stateful `act` path, logistic_decoder_rnn.py:105-107,271)"""
    fx = load("decoder_state")
    seed, B, S = int(fx["seed"]), int(fx["B"]), int(fx["S"])
    plan, emb, goal = _inference_inputs(seed, "inf", B, S)
    h0 = torch.randn(2, B, 2048, generator=syn._gen(seed, "x.inf.h0")).abs() * 0.2
    assert abs(float(h0.double().sum()) - float(fx["h0_checksum"])) < 1e-6
    with torch.no_grad():
        lp, ls, mu, grip, h_n = O.decoder_forward(sd, "action_decoder.", plan, emb, goal, h0=h0, return_state=True)
    for a, k in ((lp, "logit_probs"), (ls, "log_scales"), (mu, "means"), (grip, "grip")):
        close(a, fx[k], what=k)
    close(h_n[:, :, ::16], fx["h_n_s"], what="h_n")
    # three single-step calls carrying the state == one 3-step call
    h = h0
    outs = []
    with torch.no_grad():
        for t in range(S):
            o = O.decoder_forward(sd, "action_decoder.", plan, emb[:, t:t + 1], goal, h0=h, return_state=True)
            h = o[4]
            outs.append(o[2])
    close(torch.cat(outs, 1), fx["means"], what="means (stepwise)")


def test_sample_actions_bit_exact_indices():
    """LogisticDecoderRNN._sample with the recorded torch.rand draws: mixture / gripper indices bit-exact, actions to fp32 rounding"""
    fx = load("decoder_sample")
    t = {k: torch.tensor(fx[k]) for k in ("logit_probs", "log_scales", "means", "grip", "u_mix", "u_inv")}
    act, idx = O.sample_actions(t["logit_probs"], t["log_scales"], t["means"], t["grip"], t["u_mix"], t["u_inv"])
    assert np.array_equal(idx.numpy(), fx["mix_idx"])
    assert np.array_equal(t["grip"].argmax(-1).numpy(), fx["gripper_idx"])
    close(act, fx["actions"], 1e-6, "sampled actions")


def test_lmp_val_composition(sd):
    """hulc2.py:283-334 on the oracle: decoder losses, sampled actions (recorded uniforms), MAE, gripper success rate, KL"""
    fx = load("lmp_val")
    seed, B, S = int(fx["seed"]), int(fx["B"]), int(fx["S"])
    emb = torch.randn(B, S, 128, generator=syn._gen(seed, "x.val.emb"))
    goal = torch.randn(B, 32, generator=syn._gen(seed, "x.val.goal"))
    acts = torch.tensor(fx["acts"])
    with torch.no_grad():
        pp = O.plan_proposal(sd, "plan_proposal.", emb[:, 0], goal)
        pr, seq_feat = O.plan_recognition(sd, "plan_recognition.", emb)
        close(seq_feat[:, ::64], fx["seq_feat_s"], what="seq_feat")
        close(O.kl_loss(pp, pr, 0.01, 0.8), fx["kl"], what="kl")
        for tag in ("pp", "pr"):
            plan = torch.nn.functional.one_hot(torch.tensor(fx[f"idx_{tag}"]), 32).float().flatten(1)
            lp, ls, mu, grip = O.decoder_forward(sd, "action_decoder.", plan, emb, goal)
            close(O.decoder_loss(lp, ls, mu, grip, acts), fx[f"loss_{tag}"], what=f"loss {tag}")
            pred, _ = O.sample_actions(lp, ls, mu, grip, torch.tensor(fx[f"u_mix_{tag}"]), torch.tensor(fx[f"u_inv_{tag}"]))
            close(pred, fx[f"pred_{tag}"], 1e-4, f"pred {tag}")
            mae = torch.mean(torch.abs(pred[..., :-1] - acts[..., :-1]), 1)
            close(mae, fx[f"mae_{tag}"], 1e-4, f"mae {tag}")
            sr = torch.mean((acts[..., -1] == torch.where(pred[..., -1] > 0, 1.0, -1.0)).float())
            close(sr, fx[f"grip_sr_{tag}"], what=f"gripper sr {tag}")


def test_tcp_world_round_trip():
    """parity unpinned (pytorch3d): tcp_to_world_frame inverts world_to_tcp_frame (gripper_control.py:16-63)"""
    g = torch.Generator().manual_seed(5)
    a = torch.rand(3, 7, 7, generator=g) * 2 - 1
    obs = torch.randn(3, 7, 15, generator=g)
    obs[..., 3:6] = (torch.rand(3, 7, 3, generator=g) - 0.5) * 3.0
    back = O.tcp_to_world_frame(O.world_to_tcp_frame(a, obs), obs)
    assert (back - a).abs().max().item() < 2e-4


# ---- input transforms -------------------------------------------------------------
@pytest.mark.parametrize("tag", ["static", "gripper"])
def test_input_transforms(tag):
    """uint8 NHWC frames -> model input: the reference's RandomShiftsAug (recorded torch.randint draw) + ScaleImageTensor + Normalize;
    shifts are integers and must select exactly the pixels an integer crop of the replicate-padded frame selects (to grid_sample's
    interpolation rounding)"""
    fx = load(f"transforms_{tag}")
    u8, shift, pad = torch.tensor(fx["frames_u8"]), torch.tensor(fx["shift"]), int(fx["pad"])
    train = O.frames_u8_to_input(u8, pad, shift)
    val = O.frames_u8_to_input(u8)
    close(train[:, :, ::3, ::3], fx["train"], 1e-6, "train transform")
    close(val[:, :, ::3, ::3], fx["val"], 1e-7, "val transform")
    assert abs(float(train.double().sum()) - float(fx["train_sum"])) < 1e-3 and abs(float(val.double().sum()) - float(fx["val_sum"])) < 1e-6
    n, h = u8.shape[0], u8.shape[1]
    idx = torch.arange(h)
    for i in range(n):                                      # integer-crop form used by the HIP kernels
        sx, sy = int(shift[i, 0]), int(shift[i, 1])
        yy, xx = (idx + sy - pad).clamp(0, h - 1), (idx + sx - pad).clamp(0, h - 1)
        crop = (u8[i].permute(2, 0, 1).float()[:, yy][:, :, xx] / 255 - 0.5) / 0.5
        assert (crop - train[i]).abs().max().item() < 1e-4, "grid_sample with integer shifts == integer crop (to interpolation rounding)"


def test_u8_normalise_fma_is_bf16_exact():
    """the kernels stage uint8 frames as bf16(fma(b, 2/255, -1)); the reference computes (b/255 - 0.5)/0.5 in fp32
    (hulc2/utils/transforms.py:8-19 ScaleImageTensor + torchvision Normalize(0.5, 0.5)).  Identical bf16 for all 256 bytes."""
    b = np.arange(256, dtype=np.float32)
    exact = ((b / np.float32(255.0)) - np.float32(0.5)) / np.float32(0.5)
    fma = (b.astype(np.float64) * np.float64(np.float32(2.0 / 255.0)) - 1.0).astype(np.float32)   # one rounding, as v_fma_f32

    def bf16(x):
        u = x.view(np.uint32).astype(np.uint64)
        return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32)

    assert np.array_equal(bf16(exact), bf16(fma))


def test_minilm_oracle_matches_transformers_bert():
    """the oracle's restatement of the sentence encoder (BertModel + mean pooling) against the fixture produced by
    transformers' own BertModel with the seeded weight recipe (oracle/gen_golden.py minilm)"""
    fx = dict(np.load(G / "minilm.npz", allow_pickle=True))
    sd = _bert_sd(int(fx["seed"]))
    for k, want in zip(fx["ck"], fx["cv"]):
        assert abs(float(sd[str(k)].double().sum()) - float(want)) < 1e-6 * max(1.0, abs(float(want))), f"recipe drifted: {k}"
    ids, mask = torch.tensor(fx["input_ids"]), torch.tensor(fx["attention_mask"])
    emb = O.minilm_sentence_embedding(sd, ids, mask)
    want = torch.tensor(fx["sentence_embedding"])
    assert (emb - want).abs().max().item() < 2e-5 * want.abs().max().item() + 1e-6


def _bert_sd(seed):
    D, I = 384, 1536
    shapes = {"embeddings.word_embeddings.weight": (30522, D), "embeddings.position_embeddings.weight": (512, D),
              "embeddings.token_type_embeddings.weight": (2, D), "embeddings.LayerNorm.weight": (D,), "embeddings.LayerNorm.bias": (D,)}
    for l in range(3):
        q = f"encoder.layer.{l}."
        for n in ("query", "key", "value"):
            shapes[q + f"attention.self.{n}.weight"] = (D, D); shapes[q + f"attention.self.{n}.bias"] = (D,)
        shapes[q + "attention.output.dense.weight"] = (D, D); shapes[q + "attention.output.dense.bias"] = (D,)
        shapes[q + "attention.output.LayerNorm.weight"] = (D,); shapes[q + "attention.output.LayerNorm.bias"] = (D,)
        shapes[q + "intermediate.dense.weight"] = (I, D); shapes[q + "intermediate.dense.bias"] = (I,)
        shapes[q + "output.dense.weight"] = (D, I); shapes[q + "output.dense.bias"] = (D,)
        shapes[q + "output.LayerNorm.weight"] = (D,); shapes[q + "output.LayerNorm.bias"] = (D,)
    sd = {k: torch.empty(s) for k, s in shapes.items()}
    syn.fill_bert_state_dict_(sd, seed)
    return sd


def test_r3m_trunk_oracle_matches_nn_layers():
    """the functional restatement of r3m's trunk (normalise -> ResNet-18 without classifier -> global mean)
    against the same network assembled from torch's own nn.Conv2d / nn.BatchNorm2d / nn.MaxPool2d layers in eval mode, and the
    VisionR3M head (vision_r3m.py:28-32).  torchvision and r3m are absent here (parity unpinned); this pins the restatement's wiring —
    strides, paddings, where the residual joins, the downsample branches."""
    import torch.nn as nn

    from hulc2_amd import synthetic as syn
    from hulc2_amd.models.perceptual_encoders.vision_r3m import VisionR3M

    m = VisionR3M(None, 64)
    syn.fill_state_dict_(m.state_dict(), 5)
    m.eval()
    net = m.r3m.convnet
    x = torch.rand(2, 3, 75, 100, generator=syn._gen(5, "r3m.x")) * 255
    with torch.no_grad():
        t = (x / 255 - torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)) / torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
        t = nn.functional.max_pool2d(torch.relu(net.bn1(net.conv1(t))), 3, 2, 1)
        for blk in net.blocks():
            idn = t if blk.downsample is None else blk.downsample(t)
            t = torch.relu(blk.bn2(blk.conv2(torch.relu(blk.bn1(blk.conv1(t))))) + idn)
        want_f = torch.flatten(nn.functional.adaptive_avg_pool2d(t, 1), 1)
        want = m.fc2(torch.relu(m.fc1(want_f)))
        sd = m.state_dict()
        got_f = O.r3m_trunk_features(sd, x)
        got = O.vision_r3m(sd, "", x)
    assert got_f.shape == (2, 512)
    assert (got_f - want_f).abs().max().item() <= 1e-5 * want_f.abs().max().item()
    assert (got - want).abs().max().item() <= 1e-5 * want.abs().max().item()


def test_r3m_trunk_train_mode_oracle_matches_nn_layers():
    """row f-4, VERDICT r03 missing #1: the trunk as the reference runs it during training (BatchNorm on batch statistics, running statistics
    updated) — the oracle's `trunk_maps(bn_train=True)` against tests/golden/r3m_trunk_trainmode.npz, produced by torch's own nn.Conv2d /
    nn.BatchNorm2d / nn.MaxPool2d layers in train mode (oracle/gen_golden.py::gen_r3m_trunk_trainmode)."""
    from oracle import affordance_oracle as A
    g = load("r3m_trunk_trainmode")
    B, HW, seed = int(g["B"]), int(g["HW"]), int(g["seed"])
    from hulc2_amd.models.perceptual_encoders.vision_r3m import R3M
    sd = {"r3m.convnet." + k: torch.empty_like(v) for k, v in R3M("resnet18").convnet.state_dict().items()}
    syn.fill_state_dict_(sd, seed)
    for k, v in sd.items():
        if not v.is_floating_point():
            v.zero_()                                           # (num_batches_tracked: the recipe fills floating-point tensors)
    img = torch.randn(B, 3, HW, HW, generator=syn._gen(seed, "x.trunk.train"))
    with torch.no_grad():
        maps = A.trunk_maps(sd, img, bn_train=True)
    for i, m in enumerate(maps):
        want = torch.as_tensor(g[f"map{i}"])
        assert m.shape == want.shape and (m - want).abs().max().item() <= 2e-5 * want.abs().max().item(), i
    for name, want in zip(g["stat_names"], g["stat_sums"]):
        got = float(sd["r3m.convnet." + str(name)].double().sum())
        assert abs(got - float(want)) <= 1e-5 * max(1.0, abs(float(want))), name
    assert torch.allclose(sd["r3m.convnet.bn1.running_mean"], torch.as_tensor(g["bn1_running_mean"]), atol=1e-6)
    assert torch.allclose(sd["r3m.convnet.layer4.1.bn2.running_var"], torch.as_tensor(g["layer4_1_bn2_running_var"]), rtol=1e-5, atol=1e-7)
    assert int(sd["r3m.convnet.bn1.num_batches_tracked"]) == int(g["tracked"]) == 1
    # (round 5) the trainable stem: the oracle's gradients of conv1.weight / bn1.weight / bn1.bias — autograd through its own functional ops —
    # against the ones torch's nn layers produced for the same seeded upstream gradients of the five maps
    sd3 = {k: torch.empty_like(v) for k, v in sd.items()}
    syn.fill_state_dict_(sd3, seed)
    stem = ["r3m.convnet.conv1.weight", "r3m.convnet.bn1.weight", "r3m.convnet.bn1.bias"]
    for k in stem:
        sd3[k].requires_grad_(True)
    m3 = A.trunk_maps(sd3, img, bn_train=True)
    ups = [torch.randn(m.shape, generator=syn._gen(seed, f"g.trunk.map{i}")) * (0.5 ** i) for i, m in enumerate(m3)]
    sum((m * u).sum() for m, u in zip(m3, ups)).backward()
    for k, name in zip(stem, ("d_conv1_weight", "d_bn1_weight", "d_bn1_bias")):
        want = torch.as_tensor(g[name])
        assert (sd3[k].grad - want).norm().item() <= 1e-4 * want.norm().item(), name
    # and train mode is NOT the frozen inference-mode trunk the default build runs: the maps differ by O(1)
    sd2 = {k: torch.empty_like(v) for k, v in sd.items()}
    syn.fill_state_dict_(sd2, seed)
    with torch.no_grad():
        frozen = A.trunk_maps(sd2, img, bn_train=False)
    assert (frozen[4] - maps[4]).abs().max().item() > 0.05 * maps[4].abs().max().item()


def test_affordance_step():
    """row f-4: the affordance model's trainable part (oracle/affordance_oracle.py) against a step of the reference's own
    UnetLangFusionDecoder + segmentation head + DepthEstimationGaussian + cross_entropy_with_logits (gen_golden.py::gen_affordance):
    losses, logits, depth distribution, decoder output, every gradient norm, gradient slices, BatchNorm running statistics"""
    from oracle import affordance_oracle as A
    g = load("affordance_step_B2_64")
    B, HW = int(g["B"]), int(g["HW"])
    sd = {k: torch.empty(s) for k, s in A.trainable_shapes(HW // 32).items()}
    syn.fill_affordance_state_dict_(sd, int(g["seed"]))
    for v in sd.values():
        v.requires_grad_(True)
    feats = [torch.as_tensor(g[f"feat{i}"]) for i in range(5)]
    stats = []
    out = A.training_step(sd, feats, torch.as_tensor(g["emb"]), torch.as_tensor(g["p0"]), torch.as_tensor(g["gt_depth"]), HW, stats=stats)
    out["loss"].backward()
    for k in ("loss", "aff_loss", "depth_loss", "mu", "sigma"):
        close(out[k], g[k], what=k)
    close(out["logits"][:, ::37], g["logits_sub"], what="logits")
    close(out["dec"][:, :, ::5, ::7], g["dec_sub"], what="decoder output")
    close(out["l_enc"][:, ::16], g["l_enc_sub"], what="l_enc")
    seen = 0
    for k in g:
        if k.startswith("gnorm."):
            close(sd[k[6:]].grad.norm(), g[k], rtol=1e-4, what=k)
            seen += 1
        elif k.startswith("grad."):
            gr = sd[k[5:]].grad.flatten()
            close(gr[::97][:512] if g[k].shape[0] == 512 and gr.numel() > 512 * 97 - 96 else (gr if gr.numel() == g[k].shape[0] else gr[::97][:512]), g[k], rtol=1e-4, what=k)
    assert seen == 50
    unused = [k for k, v in sd.items() if v.grad is None]
    assert sorted(unused) == sorted(f"decoder.blocks.{i}.lang_proj.{n}" for i in (3, 4) for n in ("weight", "bias"))
    # BatchNorm2d bookkeeping (momentum 0.1 from mean 0 / var 1, unbiased batch variance)
    close(0.1 * stats[0][0], g["bn_mean.b0c1"], rtol=1e-4, what="running_mean of block 0 conv1")
    close(0.9 + 0.1 * stats[9][1], g["bn_var.b4c2"], rtol=1e-4, what="running_var of block 4 conv2")
    # inference on the updated running statistics: logits, arg-max pixel, depth distribution
    running = [(0.1 * m, 0.9 + 0.1 * v) for m, v in stats]
    for j, (m, v) in enumerate(running):
        close(m, g[f"run_mean.b{j // 2}conv{j % 2 + 1}"], rtol=1e-4, what=f"running mean {j}")
        close(v, g[f"run_var.b{j // 2}conv{j % 2 + 1}"], rtol=1e-4, what=f"running var {j}")
    with torch.no_grad():
        ev = A.training_step(sd, feats, torch.as_tensor(g["emb"]), torch.as_tensor(g["p0"]), torch.as_tensor(g["gt_depth"]), HW, train=False, running=running)
    close(ev["logits"][:, ::37], g["eval_logits_sub"], rtol=1e-4, what="eval logits")
    assert torch.equal(ev["logits"].argmax(-1).to(torch.int32), torch.as_tensor(g["eval_argmax"]))
    close(ev["mu"], g["eval_mu"], rtol=1e-4, what="eval mu")
