"""debug: real-world config, mixed mode, B = 2, S = 16: gradients of the plain path vs the step node vs a full ArenaTrainer (sinks)"""
import os, sys
import torch
sys.path.insert(0, ".")
from hulc2_amd import kernels as kn, synthetic as syn, gradsink
from hulc2_amd.compat import instantiate
from hulc2_amd.config import real_world_model_config
from hulc2_amd.trainer import ArenaTrainer

dev = torch.device("cuda", 0)
mode = os.environ.get("MODE", "mixed")


def run(kind):
    global mode
    os.environ.pop("HULC_NO_STEP_NODE", None)
    if kind == "plain":
        os.environ["HULC_NO_STEP_NODE"] = "1"
    kn.set_compute(mode)
    kn.reset_step_state(dev)
    m = instantiate(real_world_model_config(dropout_p=0.0)).to(dev)
    syn.fill_state_dict_(m.state_dict(), 21)
    m.train()
    batch = syn.make_batch(21, 2, 16, static_hw=(150, 200))
    for mod in batch.values():
        mod["rgb_obs"]["rgb_static"] = (mod["rgb_obs"]["rgb_static"] + 1) * 127.5
    batch = syn._to(batch, dev)
    if kind == "trainer":
        tr = ArenaTrainer(m, overlap=False)
        loss = tr._forward_backward(batch, 0)
    else:
        loss = m.training_step(batch, 0)
        loss.backward()
    torch.cuda.synchronize()
    kn.check_faults(dev)
    out = {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in m.named_parameters() if p.requires_grad}
    kn.set_compute("bf16")
    return float(loss.detach()), out


import gc
ref_l, ref = run("plain")
seq = os.environ.get("SEQ", "fp32:node,bf16:node,mixed:node").split(",")
for item in seq:
    md, kind = item.split(":")
    mode = md
    l, g = run(kind)
    gc.collect() if os.environ.get("GC") else None
    if md != os.environ.get("MODE", "mixed"):
        print(f"[{md} {kind}] loss {l:.6f} (warm-up of another mode)")
        continue
    bad = []
    for n in ref:
        a, b = ref[n], g[n]
        if a is None or b is None:
            continue
        e = float((a.double() - b.double()).norm() / (a.double().norm() + 1e-30))
        if not (e <= 1e-4):
            bad.append((n.replace("perceptual_encoder.", "pe."), round(e, 4)))
    print(f"[{md} {kind}] loss {l:.6f} (plain {ref_l:.6f}); tensors off by > 1e-4: {len(bad)}: {bad[:12]}")
