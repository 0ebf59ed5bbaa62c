"""What HBM rate do plain streaming kernels reach on this box?  Context for the roofline fractions in DESIGN.md (peak 8 TB/s is the
pin rate; this prints what torch's own read / copy / fill kernels achieve on 0.5-2 GB tensors)."""
import torch

dev = torch.device("cuda", 0)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3


for mb in (491, 1024, 2048):
    n = mb * 1024 * 1024 // 4
    x = torch.randn(n, device=dev)
    y = torch.empty_like(x)
    t = timed(lambda: torch.sum(x))
    print(f"read  {mb:5d} MB  sum      {mb / 1024 / t / 1e3 * 1.073741824:6.2f} TB/s")
    t = timed(lambda: y.copy_(x))
    print(f"copy  {mb:5d} MB  copy_    {2 * mb / 1024 / t / 1e3 * 1.073741824:6.2f} TB/s (read + write)")
    t = timed(lambda: y.fill_(1.0))
    print(f"write {mb:5d} MB  fill_    {mb / 1024 / t / 1e3 * 1.073741824:6.2f} TB/s")
    xb = x[: n // 2].view(torch.bfloat16)
    t = timed(lambda: xb.float())
    print(f"cast  {mb // 2:5d} MB  bf16->f32 {(mb / 2 + mb) / 1024 / t / 1e3 * 1.073741824:6.2f} TB/s (read + write)")

import ctypes
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _build  # noqa: E402  (builds the .so from the .hip next to it)
lib = ctypes.CDLL(_build.ensure("read_probe.so"))
out = torch.zeros(65536, device=dev)
n = 1024 * 1024 * 1024 // 4
x = torch.randn(n, device=dev)
for unroll in (1, 4, 8):
    for blocks in (1024, 2048, 4096, 8192, 16384):
        t = timed(lambda: lib.read_probe(ctypes.c_void_p(x.data_ptr()), ctypes.c_long(n // 4), ctypes.c_void_p(out.data_ptr()), blocks, unroll,
                                         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        print(f"read_probe 1 GiB  unroll {unroll}  blocks {blocks:5d}: {n * 4 / t / 1e12:5.2f} TB/s")
