"""headline-mode forward: max-abs error of the perceptual embeddings against the oracle (B = 32, S = 32) per HULC_FP32_SITES setting"""
import os, sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from hulc2_amd import kernels as kn, synthetic as syn
from hulc2_amd.compat import instantiate
from hulc2_amd.config import default_model_config
import test_parity_gpu as T
dev = torch.device("cuda:0")
seed, B, S = 321, 32, 32
cfg = default_model_config(gripper_control=True, dropout_p=0.0)
ref = None
for sites in (None, "head,goal,encfc,txl,a3", "head,goal,encfc,txl,conv1", "head,goal,encfc,txl,conv1,a3"):
    if sites is None:
        os.environ.pop("HULC_FP32_SITES", None)
    else:
        os.environ["HULC_FP32_SITES"] = sites
    kn.set_compute("bf16")
    m = instantiate(cfg).to(dev)
    syn.fill_state_dict_(m.state_dict(), seed)
    m.train()
    batch = syn.make_batch(seed, B, S, device=dev)
    taps = []
    h = m.perceptual_encoder.register_forward_hook(lambda mod, i, o: taps.append(o))
    total = m.training_step(batch, 0)
    h.remove()
    torch.cuda.synchronize()
    if ref is None:
        ref, _ = T._oracle_step(seed, B, S, True, set(dict(m.named_parameters())))
    embs = torch.cat(taps, dim=0).float().cpu()
    ev = (embs[:B] - ref["emb_vis"]).abs().max().item() / ref["emb_vis"].abs().max().item()
    el = (embs[B:] - ref["emb_lang"]).abs().max().item() / ref["emb_lang"].abs().max().item()
    print(f"sites={sites or 'default'}: embedding error vis {ev:.3e} lang {el:.3e} | loss {float(total):.6f} vs {float(ref['total_loss']):.6f}", flush=True)
