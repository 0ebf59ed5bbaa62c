#!/usr/bin/env python3
"""Which Python lines launch the framework (aten / runtime-copy) kernels of a training step?  Runs eager steps of the benchmark
configuration under torch.profiler with stacks and prints, per kernel-launching aten op, the launches per step and the innermost
hulc2_amd frames that issued them.   usage (GPU box): python tools/glue_trace.py [--steps 2] [--affordance]"""
import collections
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from hulc2_amd import kernels as kn, synthetic as syn  # noqa: E402
from hulc2_amd.compat import instantiate  # noqa: E402
from hulc2_amd.config import default_model_config  # noqa: E402
from hulc2_amd.trainer import ArenaTrainer  # noqa: E402


def main():
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 2
    dev = torch.device("cuda", 0)
    kn.set_compute("bf16")
    if "--affordance" in sys.argv:
        from hulc2_amd.affordance import PixelAffLangDetector
        model = PixelAffLangDetector(img_size=224).to(dev)
        syn.fill_affordance_state_dict_({k: v for k, v in model.state_dict().items() if ".r3m." not in k}, 42)
        syn.fill_state_dict_({"r3m.convnet." + k: v for k, v in model.model.aff_stream.r3m.convnet.state_dict().items()}, 42)
        model.train()
        trainer = ArenaTrainer(model, lr=1e-4, overlap=False)
        g = torch.Generator().manual_seed(42)
        nb = 32
        batch = ({"img": torch.randn(nb, 3, 224, 224, generator=g).to(dev), "lang_goal": (torch.randn(nb, 384, generator=g) * 0.5).to(dev)},
                 {"p0": torch.stack([torch.randint(0, 224, (nb,), generator=g), torch.randint(0, 224, (nb,), generator=g)], 1).to(dev),
                  "normalized_depth": torch.randn(nb, generator=g).to(dev)})
    else:
        model = instantiate(default_model_config(gripper_control=True, dropout_p=0.1)).to(dev)
        syn.fill_state_dict_(model.state_dict(), 42)
        model.train()
        trainer = ArenaTrainer(model, lr=2e-4, overlap=False)
        batch = syn.make_batch(42, 32, 32, device=dev)
        for db in batch.values():
            db.pop("plan_idx", None)
    for i in range(3):
        trainer.step(batch, i)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        for i in range(steps):
            trainer.step(batch, 3 + i)
        torch.cuda.synchronize()
    # CPU-side aten ops that own device kernels: attribute to the innermost repo frame
    rows = collections.Counter()
    for ev in prof.events():
        if not ev.name.startswith("aten::") or not ev.kernels:
            continue
        if any(k.name.startswith("aten::") for k in ev.cpu_children if k.kernels):
            continue                                                        # count the leaf op only
        frames = [f for f in (ev.stack or []) if "hulc2_amd" in f or "bench" in f or "glue_trace" in f]
        where = " <- ".join(f.split("/root/repo/")[-1].split("repo/")[-1] for f in frames[:2]) or "(autograd engine / no repo frame)"
        for k in ev.kernels:
            rows[(ev.name, k.name[:60], where + "  shapes " + str(getattr(ev, "input_shapes", ""))[:120])] += 1
    total = 0
    for (op, kname, where), n in sorted(rows.items(), key=lambda kv: -kv[1]):
        total += n
        print(f"{n / steps:6.1f}/step  {op:28s} {kname:60s} {where}")
    print(f"total {total / steps:.1f} aten-launched kernels per step")
    allk = collections.Counter()
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CUDA and "Memcpy" not in ev.name and "Memset" not in ev.name:
            nm = ev.name.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "")
            allk[nm[:90]] += 1
    print(f"ALL device kernels: {sum(allk.values()) / steps:.1f} per step")
    for name, n in allk.most_common(60):
        print(f"{n / steps:6.1f}/step  {name}")
    # runtime copies / memsets (not kernels of an aten op): name, count, and the CPU op that was running when they were issued
    mem = collections.Counter()
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CUDA and ("Memcpy" in ev.name or "Memset" in ev.name):
            mem[ev.name] += 1
    for name, n in mem.most_common():
        print(f"{n / steps:6.1f}/step  {name}")
    cpu = collections.Counter()
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CPU and any(t in ev.name for t in ("hipMemcpy", "hipMemset", "aten::_to_copy", "aten::to", "aten::item",
                                                                                          "aten::_local_scalar_dense", "aten::scalar_tensor", "aten::tensor")):
            par = ev.cpu_parent.name if ev.cpu_parent is not None else "-"
            cpu[(ev.name, par)] += 1
    for (name, par), n in cpu.most_common(40):
        print(f"{n / steps:6.1f}/step  {name:32s} inside {par}")


if __name__ == "__main__":
    main()
