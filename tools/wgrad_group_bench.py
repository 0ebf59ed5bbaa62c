#!/usr/bin/env python3
"""Times the grouped weight-gradient launch (csrc/wgrad_group.hip) on the product shapes of one benchmark step, as one call and per shape.
usage (GPU box): python tools/wgrad_group_bench.py [--each]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from hulc2_amd import kernels as kn  # noqa: E402

SHAPES = [(2048, 2048, 64)] * 3 + [(512, 128, 2048)] * 2 + [(384, 128, 2048)] * 2 + [(64, 512, 2048)] * 2 + [(128, 128, 2048)] * 2 + \
         [(2048, 2048, 32)] * 2 + [(32, 128, 32)] * 2 + [(184, 2048, 2048), (128, 3136, 2048), (2048, 64, 2048), (1024, 2048, 64)] + \
         [(32, 2048, 32)] * 2 + [(128, 32, 32), (1024, 4096, 64), (2048, 160, 64), (128, 4096, 32), (2048, 384, 32), (2048, 128, 32),
                                 (4096, 128, 64), (2048, 1024, 64), (2048, 32, 64)]


def timed(fn, n=20):
    if "--once" in sys.argv:                                  # under rocprofv3: one launch per shape, read the durations from the trace
        fn()
        torch.cuda.synchronize()
        return 0.0
    for _ in range(3):
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
    dev = torch.device("cuda", 0)
    kn.set_compute("bf16")
    probs = []
    for M, N, K in SHAPES:
        A, B = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
        probs.append((A, B, torch.empty(M, N, device=dev), torch.empty(M, device=dev), M, N, K, M, N, N, False, False))

    def run(sub):
        kn._wg_pending[dev] = list(sub)
        kn.wgrad_flush(dev)

    print(f"all {len(probs)} products in one call: {timed(lambda: run(probs)):.1f} us")
    if "--each" in sys.argv:
        seen = set()
        for p in probs:
            if p[4:7] in seen:
                continue
            seen.add(p[4:7])
            print(f"  {p[4:7]}: {timed(lambda: run([p])):.1f} us")


if __name__ == "__main__":
    main()
