"""is workgroup -> XCD assignment round-robin on the linear workgroup id?"""
import ctypes
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import _build  # noqa: E402  (builds the .so from the .hip next to it)
lib = ctypes.CDLL(_build.ensure("xcd_probe.so"))
dev = torch.device("cuda")
for (gx, gy, gz, thr) in ((1024, 1, 1, 256), (32, 16, 4, 256), (7, 5, 3, 512), (4096, 1, 1, 64)):
    n = gx * gy * gz
    out = torch.full((n,), -1, dtype=torch.int32, device=dev)
    for rep in range(3):
        lib.xcd_probe(ctypes.c_void_p(out.data_ptr()), gx, gy, gz, thr, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        o = out.cpu()
        ok = bool((o == (torch.arange(n) % 8).int()).all())
        print(f"grid ({gx},{gy},{gz}) x {thr}: xcc_id == linear id % 8: {ok}   first 12: {o[:12].tolist()}")
