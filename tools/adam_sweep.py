"""The Adam pass alone at the benchmark's arena size (47.05 M elements, 30 bytes each): HULC_ADAM_NT (non-temporal operand classes, see
csrc/optim.hip) and HULC_ADAM_BLOCKS (grid cap) are read once per process — run once per setting."""
import sys, os, torch
sys.path.insert(0, '.')
from hulc2_amd import kernels as kn
dev = torch.device("cuda")
n = 47_050_000 // 8 * 8
p, g, m, v = (torch.randn(n, device=dev) * 0.01 for _ in range(4))
v.abs_()
sh = torch.zeros(n, dtype=torch.bfloat16, device=dev)
st = kn.step_state(dev); st[1] = 5
def f(): kn.adam_step(p, g, m, v, sh, n, 2e-4, 0.9, 0.999, 1e-8, 0.0, 1, step_state_dev=st)
for _ in range(3): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print(f"NT={os.environ.get('HULC_ADAM_NT')} blocks={os.environ.get('HULC_ADAM_BLOCKS')}: {us:.1f} us  {n * 30 / us / 1e6:.2f} TB/s")
