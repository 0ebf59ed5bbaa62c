import sys, os, torch
sys.path.insert(0, '.')
from hulc2_amd import kernels as kn, synthetic as syn
from hulc2_amd.compat import instantiate
from hulc2_amd.config import default_model_config
dev = torch.device('cuda')
m = instantiate(default_model_config(gripper_control=True, dropout_p=0.0)).to(dev); syn.fill_state_dict_(m.state_dict(), 1); m.train()
B, S = 64, 32
g = torch.Generator().manual_seed(0)
plan = torch.nn.functional.one_hot(torch.randint(0, 32, (B, 32), generator=g), 32).float().flatten(1).to(dev).requires_grad_()
emb = torch.randn(B, S, 128, generator=g).to(dev).requires_grad_(); goal = torch.randn(B, 32, generator=g).to(dev).requires_grad_()
acts = torch.cat([torch.rand(B, S, 6, generator=g) * 2 - 1, torch.ones(B, S, 1)], -1).to(dev); obs = torch.randn(B, S, 15, generator=g).to(dev)
def run():
    for p in m.parameters(): p.grad = None
    l = m.action_decoder.loss(plan, emb, goal, acts, obs); l.backward(); return l.detach()
for mode in ("eager", "graph"):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): run()
    torch.cuda.synchronize()
    if mode == "graph":
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s): run()
        fn = gr.replay
    else:
        fn = run
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"decoder fwd+bwd B={B} S={S} {mode} wavefront={'off' if os.environ.get('HULC_NO_WAVEFRONT') else 'on'}: {e0.elapsed_time(e1)/5:.3f} ms")
