"""debug: Lightning's closure order (training_step -> zero_grad(set_to_none=False) -> backward) against the usual order, per-step checksums"""
import os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from hulc2_amd import kernels as kn, synthetic as syn
from hulc2_amd.compat import instantiate
from hulc2_amd.config import default_model_config

dev = torch.device("cuda:0")


def loop(order, stn, steps=8):
    kn.reset_step_state(dev)
    kn.set_compute("bf16")
    m = instantiate(default_model_config(gripper_control=True, dropout_p=0.1)).to(dev)
    syn.fill_state_dict_(m.state_dict(), 19)
    m.train()
    batch = syn.make_batch(19, 2, 8, device=dev)
    for db in batch.values():
        db.pop("plan_idx", None)
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=2e-4)
    scaler = torch.amp.GradScaler("cuda", init_scale=65536.0)
    out = []
    for i in range(steps):
        if not order:
            opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            loss = m.training_step(batch, i)
        if order:
            opt.zero_grad(set_to_none=stn)
        scaler.scale(loss).backward()
        gs = sum(float(p.grad.double().sum()) for p in m.parameters() if p.grad is not None)
        gn = sum(1 for p in m.parameters() if p.grad is not None)
        scaler.step(opt)
        scaler.update()
        ps = sum(float(p.double().sum()) for p in m.parameters())
        out.append((float(loss), gs, gn, ps, float(scaler.get_scale())))
    node = m.__dict__["_hulc_step_node"]
    print(order, stn, "replays", node.replays, "accum", node.accum_steps, "disabled", node.disabled)
    return out


runs = {"std_a": loop(False, True), "std_b": loop(False, True), "closure_none": loop(True, True), "closure_zero_a": loop(True, False),
        "closure_zero_b": loop(True, False)}
for i in range(8):
    print(i, " | ".join(f"{k}: {v[i][0]:.6f} g{v[i][1]:.6e} n{v[i][2]} p{v[i][3]:.9e} s{v[i][4]:.0f}" for k, v in runs.items()))


def loop2(order, stn, steps=4):
    kn.reset_step_state(dev)
    m = instantiate(default_model_config(gripper_control=True, dropout_p=0.1)).to(dev)
    syn.fill_state_dict_(m.state_dict(), 19)
    m.train()
    batch = syn.make_batch(19, 2, 8, device=dev)
    for db in batch.values():
        db.pop("plan_idx", None)
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=2e-4)
    scaler = torch.amp.GradScaler("cuda", init_scale=65536.0)
    for i in range(steps):
        if not order:
            opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            loss = m.training_step(batch, i)
        if order:
            opt.zero_grad(set_to_none=stn)
        scaler.scale(loss).backward()
        grads = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
        where = {n: ("arena" if m.__dict__["_hulc_step_node"].keeper.flat_g.data_ptr() <= p.grad.data_ptr() < m.__dict__["_hulc_step_node"].keeper.flat_g.data_ptr() + 4 * m.__dict__["_hulc_step_node"].keeper.flat_g.numel() else "other")
                 for n, p in m.named_parameters() if p.grad is not None}
        scaler.step(opt)
        scaler.update()
    return grads, where


ga, wa = loop2(False, True)
gb, wb = loop2(True, False)
for n in ga:
    if not torch.equal(ga[n], gb[n]):
        d = (ga[n] - gb[n]).abs().max().item()
        print(f"DIFF {n}: max|d| {d:.4e} of max {ga[n].abs().max().item():.4e}; std {wa[n]}, closure {wb[n]}; shape {tuple(ga[n].shape)}")
