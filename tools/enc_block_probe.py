"""Which launch of the encoders' backward waits for a co-resident wave?  Eager split step: head, then ONE spinning wave (32 registers, 700 us) on a
second stream, then the encoder backward with per-launch events: the launch that cannot run beside the wave shows its wait as its duration."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe"))
import torch
from _build import ensure
from hulc2_amd import kernels as kn, synthetic as syn
from hulc2_amd.compat import instantiate
from hulc2_amd.config import default_model_config
from hulc2_amd.trainer import ArenaTrainer

occ = ctypes.CDLL(ensure("occupy_probe.so"))
occ.occupy_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda", 0)
kn.set_compute("bf16")
model = instantiate(default_model_config(gripper_control=True, dropout_p=0.1)).to(dev)
syn.fill_state_dict_(model.state_dict(), 42)
model.train()
tr = ArenaTrainer(model, lr=2e-4, overlap=False)
batch = syn.make_batch(42, 32, 32, device=dev)
for db in batch.values():
    db.pop("plan_idx", None)
for i in range(3):
    tr.step(batch, i)
sink = torch.zeros(4, device=dev)
side = torch.cuda.Stream()
cur = torch.cuda.current_stream()
for spin in (0, 70000):
    tr._forward_backward_head(batch, 0)
    torch.cuda.synchronize()
    if spin:
        occ.occupy_launch(int(os.environ.get("NWG", 1)), 32, 0, spin, sink.data_ptr(), side.cuda_stream)
    kn.start_timing()
    tr._backward_encoder()
    rec, kn._timing = kn._timing, None
    torch.cuda.synchronize()
    print(f"--- spinning wave: {spin / 100:.0f} us")
    for key, e0, e1, fl, by in rec:
        d = e0.elapsed_time(e1) * 1e3
        if d > 150 or not spin:
            print(f"{d:8.1f} us  {str(key)[:120]}")
    tr.optimizer_step()
