"""forward-only time of the benchmark step (B=32/modality, S=32) per compute mode, and of the camera encoders alone"""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from hulc2_amd import kernels as kn, synthetic as syn
from hulc2_amd.compat import instantiate
from hulc2_amd.config import default_model_config

dev = torch.device("cuda:0")
B, S = 32, 32
for mode in ("bf16", "fp32"):
    kn.set_compute(mode)
    m = instantiate(default_model_config(gripper_control=True, dropout_p=0.0)).to(dev)
    syn.fill_state_dict_(m.state_dict(), 1)
    m.train()
    batch = syn.make_batch(1, B, S, device=dev)
    def t(fn, n=10):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3
    with torch.no_grad():
        full = t(lambda: m.training_step(batch, 0))
        enc = t(lambda: m.perceptual_encoder([db["rgb_obs"] for db in batch.values()], None, None))
        st = m.perceptual_encoder.rgb_static_encoder
        xs = [db["rgb_obs"]["rgb_static"].reshape(-1, 3, 200, 200) for db in batch.values()]
        from hulc2_amd import functional as HF
        conv = t(lambda: HF.conv_stack(xs, [p for i in (0, 2, 4) for p in (st.conv_model[i].weight, st.conv_model[i].bias)]))
    def fb():
        for p in m.parameters(): p.grad = None
        m.training_step(batch, 0).backward()
    both = t(fb, 5)
    print(f"{mode}: forward-only step {full:.2f} ms, both encoders {enc:.2f} ms, static conv stack {conv:.2f} ms, fwd+bwd eager {both:.2f} ms", flush=True)
kn.set_compute("bf16")
