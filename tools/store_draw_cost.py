"""host + device cost of drawing fresh windows from the episode stores (bench.py --episode-store's draw())"""
import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from hulc2_amd.datasets import DeviceEpisodeStore
dev = torch.device('cuda')
g = torch.Generator().manual_seed(0); n = 4096
rgb = {"rgb_static": torch.randint(0, 256, (n, 200, 200, 3), generator=g, dtype=torch.uint8).to(dev),
       "rgb_gripper": torch.randint(0, 256, (n, 84, 84, 3), generator=g, dtype=torch.uint8).to(dev)}
act = torch.rand(n, 7, generator=g); obs = torch.randn(n, 15, generator=g)
eps = [(a, min(a + 511, n - 1)) for a in range(0, n, 512)]
st = DeviceEpisodeStore(rgb, act, obs, eps, 20, 32, device=dev)
rs = np.random.RandomState(0)
for _ in range(3): st.batch(rs.randint(0, len(st), 32))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): st.batch(rs.randint(0, len(st), 32))
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"batch(32 windows): host {1e3 * (t1 - t0) / 50:.3f} ms per call, drained {1e3 * (t2 - t0) / 50:.3f} ms")
