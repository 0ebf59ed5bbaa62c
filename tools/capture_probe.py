"""Bisect HIP-graph capture problems: capture sub-parts of the step in separate child processes."""
import subprocess
import sys

PROBES = ["convs_static", "convs_static_fwd", "ssm_only", "encoder_static", "encoder_gripper", "goal", "proposal", "recognition", "sample_kl", "decoder1", "decoder2", "clip", "step"]


def child(which):
    import torch
    sys.path.insert(0, ".")
    from hulc2_amd import kernels as kn, synthetic as syn
    from hulc2_amd.compat import instantiate
    from hulc2_amd.config import default_model_config
    from hulc2_amd.utils.distributions import DiscState

    dev = torch.device("cuda:0")
    m = instantiate(default_model_config(gripper_control=True, dropout_p=0.1)).to(dev)
    syn.fill_state_dict_(m.state_dict(), 1)
    m.train()
    B, S = 4, 8
    batch = syn.make_batch(1, B, S, device=dev)
    for db in batch.values():
        db.pop("plan_idx", None)
    g = torch.Generator().manual_seed(0)
    emb = torch.randn(B, S, 128, generator=g).to(dev).requires_grad_()
    goal = torch.randn(B, 32, generator=g).to(dev).requires_grad_()
    plan = torch.nn.functional.one_hot(torch.randint(0, 32, (B, 32), generator=g), 32).float().flatten(1).to(dev).requires_grad_()

    def run():
        for p in m.parameters():
            p.grad = None
        if which in ("convs_static", "convs_static_fwd"):
            from hulc2_amd import functional as HF
            x = batch["vis"]["rgb_obs"]["rgb_static"]
            net = m.perceptual_encoder.rgb_static_encoder
            a3 = HF.conv_stack(x.reshape(-1, *x.shape[2:]), net.conv_params(), grad_premasked=False)
            out = a3.float().sum()
            if which == "convs_static_fwd":
                return out.detach()
        elif which in ("convs_ssm", "convs_ssm_mlp", "convs_ssm_fp32", "convs_ssm_nopremask", "convs_ssm_fwd"):
            from hulc2_amd import functional as HF
            if which.endswith("fp32"):
                kn.set_compute("fp32")
            x = batch["vis"]["rgb_obs"]["rgb_static"]
            net = m.perceptual_encoder.rgb_static_encoder
            a3 = HF.conv_stack(x.reshape(-1, *x.shape[2:]), net.conv_params(), grad_premasked=(which != "convs_ssm_nopremask"))
            f = net.spatial_softmax(a3)
            if which == "convs_ssm_fwd":
                return f.sum().detach()
            if which == "convs_ssm_mlp":
                f = HF.mlp(f, [(net.fc1[0].weight, net.fc1[0].bias, True), (net.fc2.weight, net.fc2.bias, False)])
            out = f.sum()
        elif which == "ssm_only":
            net = m.perceptual_encoder.rgb_static_encoder
            a = torch.randn(32, 21, 21, 64, device=dev).to(torch.bfloat16).requires_grad_()
            out = net.spatial_softmax(a).sum()
        elif which == "encoder_static":
            x = batch["vis"]["rgb_obs"]["rgb_static"]
            out = m.perceptual_encoder.rgb_static_encoder(x.reshape(-1, *x.shape[2:])).sum()
        elif which == "encoder_gripper":
            x = batch["vis"]["rgb_obs"]["rgb_gripper"]
            out = m.perceptual_encoder.rgb_gripper_encoder(x.reshape(-1, *x.shape[2:])).sum()
        elif which == "goal":
            out = m.visual_goal(emb[:, -1]).sum() + m.language_goal(batch["lang"]["lang"]).sum()
        elif which == "proposal":
            out = m.plan_proposal(emb[:, 0], goal).logit.sum()
        elif which == "recognition":
            st, f = m.plan_recognition(emb)
            out = st.logit.sum() + f.sum()
        elif which == "sample_kl":
            lg = emb.reshape(B, -1)[:, :1024]
            pl, _ = m.dist.rsample_plan(DiscState(lg), seed=3)
            out = pl.sum() + m.dist.kl_balanced(DiscState(lg * 0.5), DiscState(lg), 0.01, 0.8)
        elif which == "decoder1":
            out = m.action_decoder.loss(plan, emb, goal, batch["vis"]["actions"], batch["vis"]["state_info"]["robot_obs"])
        elif which == "decoder2":
            ls = m.action_decoder.loss_segments([plan, plan], [emb, emb], [goal, goal], [batch["vis"]["actions"], batch["lang"]["actions"]],
                                                [batch["vis"]["state_info"]["robot_obs"], batch["lang"]["state_info"]["robot_obs"]])
            out = ls[0] + ls[1]
        elif which == "clip":
            out = m.clip_auxiliary_loss(torch.randn(B, 4096, device=dev), goal, torch.ones(B, dtype=torch.bool, device=dev))
        else:
            out = m.training_step(batch, 0)
        out.backward()
        return out.detach()

    for _ in range(2):
        run()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        o = run()
    gr.replay()
    torch.cuda.synchronize()
    print(which, "OK", float(o))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for w in PROBES:
            r = subprocess.run([sys.executable, __file__, w], capture_output=True, text=True)
            last = (r.stdout.strip().splitlines() or ["-"])[-1]
            print(f"{w:16s} rc={r.returncode} {last if r.returncode == 0 else r.stderr.strip().splitlines()[-1][:200]}")
