import sys, torch
sys.path.insert(0, '.')
dev = torch.device('cuda')
N = 47_050_000
flat = torch.randn(N, device=dev)
sizes = [2048 * 2048] * 8 + [4096 * 128] * 6 + [2048] * 40 + [128 * 128 * 3] * 20 + [64 * 3136] * 4 + [184 * 2048] * 2 + [1] * 4 + [1024 * 2048] * 4 + [2048 * 160, 2048 * 128, 2048 * 384]
views, off = [], 0
for n in sizes:
    views.append(flat[off:off + n]); off += (n + 7) // 8 * 8
print(len(views), off)


def t(fn, rep=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep * 1e3


print("foreach_zero of the views  %.1f us" % t(lambda: torch._foreach_zero_(views)))
print("flat.clone()               %.1f us" % t(lambda: flat.clone()))
prev = flat.clone()
print("flat.add_(prev)            %.1f us" % t(lambda: flat.add_(prev)))
srcs = [prev[v.storage_offset():v.storage_offset() + v.numel()] for v in views]
print("foreach_add views          %.1f us" % t(lambda: torch._foreach_add_(views, srcs)))
print("flat.zero_()               %.1f us" % t(lambda: flat.zero_()))
