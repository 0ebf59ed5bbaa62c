"""Validation / rollout path on the GPU against the reference fixtures and the CPU oracle:
LogisticDecoderRNN._sample / act / loss_and_act, tcp_to_world_frame, Hulc2.lmp_val / validation_step / step.

Integer results (selected mixture, gripper class, sampled plan classes) are bit-exact; floating point follows
tests/test_parity_gpu.py's tolerances."""
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
TOL = {"fp32": dict(act=1e-4, loss=1e-4), "bf16": dict(act=3e-2, loss=2e-3)}


def load(name):
    return dict(np.load(G / f"{name}.npz", allow_pickle=False))


def close(a, b, rtol, what):
    a = a.detach().double().cpu()
    b = torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    assert torch.isfinite(a).all(), f"{what}: non-finite"
    err, scale = (a - b).abs().max().item(), b.abs().max().item() + 1e-12
    assert err <= rtol * scale + 1e-6, f"{what}: err {err:.3e} > {rtol:g} * {scale:.3e}"


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def model(dev):
    m = instantiate(default_model_config(gripper_control=False, dropout_p=0.0)).to(dev)
    syn.fill_state_dict_(m.state_dict(), int(load("vision_static")["seed"]))
    m.eval()
    return m


@pytest.fixture(params=["fp32", "bf16"])
def mode(request):
    from hulc2_amd import kernels as kn

    kn.set_compute(request.param)
    yield request.param
    kn.set_compute("bf16")


def test_sample_kernel_indices_bit_exact(dev):
    """hulc_mix_sample with the reference's recorded torch.rand draws: selected mixtures / gripper classes bit-exact"""
    from hulc2_amd import functional as HF

    fx = load("decoder_sample")
    t = {k: torch.tensor(fx[k]).to(dev) for k in ("logit_probs", "log_scales", "means", "grip", "u_mix", "u_inv")}
    B, S, A, M = t["means"].shape
    y = torch.cat([t["logit_probs"].reshape(B * S, -1), t["means"].reshape(B * S, -1), t["log_scales"].reshape(B * S, -1),
                   t["grip"].reshape(B * S, 2)], dim=1)
    bounds = torch.tensor([-1.0, 1.0], device=dev)
    act, idx = HF.mix_sample(y, A, M, -7.0, bounds, 1, t["u_mix"].reshape(B * S, A, M), t["u_inv"].reshape(B * S, A), return_idx=True)
    assert np.array_equal(idx.cpu().numpy().reshape(B, S, A), fx["mix_idx"]), "selected mixture indices differ"
    want_grip = np.where(fx["gripper_idx"] == 1, 1.0, -1.0)
    assert np.array_equal(act[:, A].cpu().numpy().reshape(B, S), want_grip), "gripper command differs"
    close(act.reshape(B, S, A + 1), fx["actions"], 1e-5, "sampled actions")
    # without injected uniforms: counter RNG, deterministic per seed, different across seeds, mixtures all reachable
    a1, a2, a3 = (HF.mix_sample(y, A, M, -7.0, bounds, s) for s in (7, 7, 8))
    assert torch.equal(a1, a2) and not torch.equal(a1, a3) and torch.isfinite(a1).all()


def test_tcp_to_world_kernel(dev):
    from hulc2_amd import functional as HF

    g = torch.Generator().manual_seed(5)
    a = torch.rand(3, 9, 7, generator=g) * 2 - 1
    obs = torch.randn(3, 9, 15, generator=g)
    obs[..., 3:6] = (torch.rand(3, 9, 3, generator=g) - 0.5) * 3.0
    want = O.tcp_to_world_frame(a, obs)
    got = HF.tcp_to_world_frame(a.to(dev), obs.to(dev))
    close(got, want.numpy(), 1e-4, "tcp_to_world")
    back = HF.tcp_to_world_frame(HF.world_to_tcp_frame(a.to(dev), obs.to(dev)), obs.to(dev))     # inverse pair (gripper_control.py:16-63)
    assert (back.cpu() - a).abs().max().item() < 3e-4


def test_decoder_carried_state(dev, model, mode):
    """forward(h_0) against the reference fixture, and three single-step `act`-style calls == one 3-step call"""
    fx, t = load("decoder_state"), TOL[mode]
    seed, B, S = int(fx["seed"]), int(fx["B"]), int(fx["S"])
    idx = torch.randint(0, 32, (B, 32), generator=syn._gen(seed, "x.inf.idx"))
    plan = torch.nn.functional.one_hot(idx, 32).float().flatten(1).to(dev)
    emb = torch.randn(B, S, 128, generator=syn._gen(seed, "x.inf.emb")).to(dev)
    goal = torch.randn(B, 32, generator=syn._gen(seed, "x.inf.goal")).to(dev)
    h0 = (torch.randn(2, B, 2048, generator=syn._gen(seed, "x.inf.h0")).abs() * 0.2).to(dev)
    dec = model.action_decoder
    with torch.no_grad():
        lp, ls, mu, grip, h_n = dec(plan, emb, goal, h0)
    close(lp, fx["logit_probs"], t["act"], "logit_probs")
    close(mu, fx["means"], t["act"], "means")
    close(ls, fx["log_scales"], t["act"], "log_scales")
    close(grip, fx["grip"], t["act"], "grip")
    close(h_n[:, :, ::16], fx["h_n_s"], t["act"], "h_n")
    h, outs = h0, []
    with torch.no_grad():
        for s in range(S):
            o = dec(plan, emb[:, s:s + 1], goal, h)
            h = o[4]
            outs.append(o[2])
    close(torch.cat(outs, 1), mu.cpu().numpy(), 1e-6 if mode == "fp32" else 2e-2, "stepwise means vs one call")


def test_lmp_val_against_reference(dev, model, mode):
    """Hulc2.lmp_val (hulc2.py:247-334) with the plan classes and the torch.rand draws of the fixture injected"""
    fx, t = load("lmp_val"), TOL[mode]
    seed, B, S = int(fx["seed"]), int(fx["B"]), int(fx["S"])
    emb = torch.randn(B, S, 128, generator=syn._gen(seed, "x.val.emb")).to(dev)
    goal = torch.randn(B, 32, generator=syn._gen(seed, "x.val.goal")).to(dev)
    acts = torch.tensor(fx["acts"]).to(dev)
    obs = torch.randn(B, S, 15, generator=syn._gen(seed, "x.val.robot")).to(dev)
    dec = model.action_decoder
    dec.injected_uniforms = [(torch.tensor(fx[f"u_mix_{k}"]).to(dev), torch.tensor(fx[f"u_inv_{k}"]).to(dev)) for k in ("pp", "pr")]
    try:
        out = model.lmp_val(emb, goal, acts, obs, torch.tensor(fx["idx_pp"]).to(dev), torch.tensor(fx["idx_pr"]).to(dev))
    finally:
        dec.injected_uniforms = None
    plan_pp, loss_pp, plan_pr, loss_pr, kl, mae_pp, mae_pr, sr_pp, sr_pr, seq_feat = out
    want_pp = torch.nn.functional.one_hot(torch.tensor(fx["idx_pp"]), 32).float().flatten(1)
    assert torch.equal(plan_pp.cpu(), want_pp), "injected plan classes must come back as the one-hot plan (bit-exact)"
    close(loss_pp, fx["loss_pp"], t["loss"], "action loss pp")
    close(loss_pr, fx["loss_pr"], t["loss"], "action loss pr")
    close(kl, fx["kl"], t["loss"] * 5, "kl")
    close(seq_feat[:, ::64], fx["seq_feat_s"], t["act"] * 3, "seq_feat")
    if mode == "fp32":      # in bf16 a near-tie of the Gumbel argmax may select another mixture: compared in fp32 only
        close(mae_pp, fx["mae_pp"], 1e-3, "mae pp")
        close(mae_pr, fx["mae_pr"], 1e-3, "mae pr")
        close(sr_pp, fx["grip_sr_pp"], 1e-6, "gripper success pp")
        close(sr_pr, fx["grip_sr_pr"], 1e-6, "gripper success pr")


def test_validation_step_and_rollout(dev, mode):
    """validation_step output contract (hulc2.py:594-598) and the stateful control loop reset/step (hulc2.py:600-628)"""
    m = instantiate(default_model_config(gripper_control=True, dropout_p=0.1)).to(dev)
    syn.fill_state_dict_(m.state_dict(), 11)
    m.eval()
    B, S = 2, 8
    batch = syn.make_batch(3, B, S, device=dev)
    for db in batch.values():
        db.pop("plan_idx", None)
    out = m.validation_step(batch, 0)
    for mod in ("vis", "lang"):
        assert out[f"sampled_plan_pp_{mod}"].shape == (B, 1024) and out[f"sampled_plan_pr_{mod}"].shape == (B, 1024)
        assert torch.equal(out[f"sampled_plan_pp_{mod}"].reshape(B, 32, 32).sum(-1), torch.ones(B, 32, device=dev)), "one-hot per category"
        assert out[f"idx_{mod}"].shape[0] == B
    for k in ("val_act/vis_act_loss_pp", "val_act/lang_act_loss_pr", "val_kl/vis_kl_loss", "val_total_mae/lang_total_mae_pp",
              "val_grip/vis_grip_sr_pr", "val/val_pred_clip_loss", "val_act/action_loss_pp"):
        assert torch.isfinite(torch.as_tensor(m.logged[k])).all(), k
    # rollout: batch 1, one frame per step, language goal; the plan is re-sampled every replan_freq steps
    m.replan_freq = 2
    m.reset()
    vis = batch["vis"]
    goal = {"lang": batch["lang"]["lang"][:1]}
    acts, plans = [], []
    for s in range(4):
        obs = {"rgb_obs": {k: v[:1, s:s + 1] for k, v in vis["rgb_obs"].items()}, "depth_obs": {},
               "robot_obs": vis["robot_obs"][:1, s:s + 1], "robot_obs_raw": vis["state_info"]["robot_obs"][:1, s:s + 1]}
        a = m.step(obs, goal)
        assert a.shape == (1, 1, 7) and torch.isfinite(a).all()
        assert m.action_decoder.hidden_state is not None and m.action_decoder.hidden_state.shape == (2, 1, 2048)
        acts.append(a)
        plans.append(m.plan.clone())
    assert m.rollout_step_counter == 4
    assert torch.equal(plans[0], plans[1]) and torch.equal(plans[2], plans[3]), "the plan is kept between replans"
    # visual goal variant: current + goal frame as a 2-step sequence
    m.reset()
    gobs = {"rgb_obs": {k: v[:1, -1:] for k, v in vis["rgb_obs"].items()}, "depth_obs": {}, "robot_obs": vis["robot_obs"][:1, -1:]}
    obs = {"rgb_obs": {k: v[:1, :1] for k, v in vis["rgb_obs"].items()}, "depth_obs": {}, "robot_obs": vis["robot_obs"][:1, :1],
           "robot_obs_raw": vis["state_info"]["robot_obs"][:1, :1]}
    a = m.step(obs, gobs)
    assert a.shape == (1, 1, 7) and torch.isfinite(a).all()
