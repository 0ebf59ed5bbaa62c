"""what does ds_read_b64_tr_b16 return?  lds[i] = i; lane l = 16 g + i passes byte address base_g + (i // 4) * stride + (i % 4) * 8"""
import ctypes
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import _build  # noqa: E402  (builds the .so from the .hip next to it)
lib = ctypes.CDLL(_build.ensure("tr_probe.so"))
dev = torch.device("cuda")
for stride in (32, 80, 144):
    addr = torch.tensor([(l // 16) * 1024 + ((l % 16) // 4) * stride + (l % 4) * 8 for l in range(64)], dtype=torch.int32, device=dev)
    out = torch.zeros(256, dtype=torch.int16, device=dev)
    lib.tr_probe(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(addr.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    o = out.cpu().view(64, 4).tolist()
    print("stride", stride)
    for l in (0, 1, 2, 3, 4, 5, 15, 16, 17, 33, 63):
        g, i = l // 16, l % 16
        want = [(g * 1024 + k * stride) // 2 + i for k in range(4)]
        print(f"  lane {l:2d}: got {o[l]}  hypothesis(column i of the [4][16] block) {want}  {'OK' if o[l] == want else ''}")
