#!/usr/bin/env python3
"""2048^3 bf16 weight-gradient GEMM of the recurrent decoder, both operand layouts: row-major (gemm_tn128.hip) and k-major (gemm_nt128.hip),
back-to-back launches timed with events.  usage (GPU box): python tools/gemm_big_bench.py [M N K]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from hulc2_amd import kernels as kn  # noqa: E402


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (2048, 2048, 2048)
    dev = torch.device("cuda", 0)
    kn.set_compute("bf16")
    C = torch.empty(M, N, device=dev)
    rs = torch.empty(M, device=dev)
    At, Bt = torch.randn(K, M, device=dev).to(torch.bfloat16), torch.randn(K, N, device=dev).to(torch.bfloat16)
    Ak, Bk = At.t().contiguous(), Bt.t().contiguous()
    fl = 2.0 * M * N * K
    t = timed(lambda: kn.gemm(At, Bt, C, M, N, K, M, N, N, a_kmajor=False, b_kmajor=False, rowsum=rs))
    print(f"row-major operands (tn128): {t:7.1f} us  {fl / t / 1e6:7.1f} TFLOP/s")
    t = timed(lambda: kn.gemm(Ak, Bk, C, M, N, K, K, K, N, a_kmajor=True, b_kmajor=True, rowsum=rs))
    print(f"k-major operands   (nt128): {t:7.1f} us  {fl / t / 1e6:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
