"""MFMA utilisation of the plan-recognition transformer block alone (north_star target: fraction of the dense bf16 MFMA roof).
Algorithmic work: 2 x 598 016 MAC per token per layer -> 2.39 MFLOP/token forward for the two layers, x3 for forward+backward.
Measured with hipGraph replay (what the training step uses) at the benchmark size (64 sequences) and at larger batches."""
import sys, torch
sys.path.insert(0, '.')
from hulc2_amd import kernels as kn, synthetic as syn
from hulc2_amd.compat import instantiate
from hulc2_amd.config import default_model_config

dev = torch.device("cuda", 0)
kn.set_compute("bf16")
m = instantiate(default_model_config(gripper_control=True, dropout_p=0.1)).to(dev)
syn.fill_state_dict_(m.state_dict(), 42)
m.train()
net = m.plan_recognition
FLOP_TOKEN = 3 * 2 * 2 * 598016.0          # fwd + bwd (= 3x fwd), 2 layers, 2 FLOP per MAC
import os
for B in ([int(v) for v in os.environ["HULC_TXL_SIZES"].split(",")] if os.environ.get("HULC_TXL_SIZES") else (64, 512, 4096)):
    emb = torch.randn(B, 32, 128, device=dev, requires_grad=True)
    def step():
        for p in net.parameters():
            p.grad = None
        emb.grad = None
        st, seq = net(emb)
        (st.logit.sum() + seq.sum()).backward()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        step()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e-3
    fl = FLOP_TOKEN * B * 32
    print(f"transformer block fwd+bwd, {B:5d} sequences ({B * 32} tokens): {t * 1e3:8.3f} ms  {fl / t / 1e12:7.2f} TFLOP/s  = {fl / t / 2.5e15 * 100:5.2f} % of the 2.5 PFLOP/s bf16 roof")
