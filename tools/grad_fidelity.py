"""Gradient fidelity of the benchmarked mode against the CPU oracle at full size (B = 32 per modality, S = 32, CLIP head on), for a list
of selective-precision settings (HULC_FP32_SITES).  One oracle run, one HIP step per setting; prints median / worst relative-L2 error and
the worst tensors.   python tools/grad_fidelity.py [B S] -- settings default: none head head,pool head,pool,txl"""
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from hulc2_amd import kernels as kn, param_spec, synthetic as syn  # noqa: E402
from hulc2_amd.compat import instantiate  # noqa: E402
from hulc2_amd.config import default_model_config  # noqa: E402
from oracle import hulc2_oracle as O  # noqa: E402


def oracle_batch(raw):
    ob = {}
    for m, db in raw.items():
        ob[m] = dict(rgb_static=db["rgb_obs"]["rgb_static"], rgb_gripper=db["rgb_obs"]["rgb_gripper"], actions=db["actions"],
                     robot_obs=db["state_info"]["robot_obs"], plan_idx=db["plan_idx"])
        if m == "lang":
            ob[m].update(lang=db["lang"], use_for_aux_lang_loss=db["use_for_aux_lang_loss"])
    return ob


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    B, S = (int(args[0]), int(args[1])) if len(args) >= 2 else (32, 32)
    settings = os.environ.get("SETTINGS", "none;head;mixed;fp32").split(";")
    seed, dev = 321, torch.device("cuda:0")
    torch.set_num_threads(min(8, torch.get_num_threads()))
    cfg = default_model_config(gripper_control=True, dropout_p=0.0)
    P0 = None
    t0 = time.time()
    sd = {k: torch.empty(s) for k, s in param_spec.trainable_shapes().items()}
    syn.fill_state_dict_(sd, seed)
    for v in sd.values():
        v.requires_grad_(True)
    out = O.training_step(sd, oracle_batch(syn.make_batch(seed, B, S)), dict(gripper_control=True, use_clip_auxiliary_loss=True))
    out["total_loss"].backward()
    print(f"oracle: {time.time() - t0:.1f} s, total {float(out['total_loss']):.6f} clip {float(out['clip_loss']):.6f}", flush=True)
    for st in settings:
        mode = st if st in ("fp32", "mixed") else "bf16"
        os.environ["HULC_FP32_SITES"] = "head" if st in ("fp32", "mixed") else st
        kn.set_compute(mode)
        m = instantiate(default_model_config(gripper_control=True, dropout_p=0.0)).to(dev)
        syn.fill_state_dict_(m.state_dict(), seed)
        m.train()
        batch = syn.make_batch(seed, B, S, device=dev)
        total = m.training_step(batch, 0)
        total.backward()
        torch.cuda.synchronize()
        kn.set_compute("bf16")
        errs = {}
        for n, p in m.named_parameters():
            ref = sd[n].grad
            if ref is None or p.grad is None:
                continue
            g = p.grad.double().cpu()
            errs[n] = ((g - ref.double()).norm() / (ref.double().norm() + 1e-30)).item()
        v = sorted(errs.values())
        worst = sorted(errs.items(), key=lambda kv: -kv[1])[:8]
        over5 = sum(1 for e in v if e > 0.05)
        print(f"[{st:>16}] loss {float(total):.6f} (oracle {float(out['total_loss']):.6f}) clip {float(m.logged['train/lang_clip_loss']) / 3:.6f}  "
              f"median {v[len(v) // 2]:.4f}  max {v[-1]:.4f}  >5%: {over5}/{len(v)}", flush=True)
        for n, e in worst:
            print(f"        {e:.4f}  {n}")
        del m, batch, total
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
