"""Cost of a kernel node inside a replayed hipGraph: N dependent launches of a one-thread kernel (hulc_step_state_advance on a scratch word pair)
and of a small elementwise kernel, captured and replayed; microseconds per node."""
import sys, torch
sys.path.insert(0, '.')
from hulc2_amd import kernels as kn
dev = torch.device('cuda')
st = torch.zeros(2, dtype=torch.int64, device=dev)
a = torch.zeros(4096, device=dev)
b = torch.zeros(1 << 22, device=dev)


def timeit(fn, n, rep=5):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(n): fn()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep / n * 1e3


for n in (50, 200, 1000):
    print(f"{n:5d} nodes: one-thread kernel {timeit(lambda: kn._call('hulc_step_state_advance', st), n):6.2f} us/node   "
          f"4096-element add {timeit(lambda: a.add_(1.0), n):6.2f} us/node   16 MB add {timeit(lambda: b.add_(1.0), n):6.2f} us/node")
