import os, sys, traceback
os.environ["HULC_FP32_SITES"] = "head,goal,encfc,txl,a3"
sys.path.insert(0, '.')
import torch
from hulc2_amd import kernels as kn, synthetic as syn, functional as HF
from hulc2_amd.compat import instantiate
from hulc2_amd.config import default_model_config
from hulc2_amd.trainer import ArenaTrainer
dev = torch.device("cuda:0")
kn.set_compute("bf16")
m = instantiate(default_model_config(gripper_control=True)).to(dev)
syn.fill_state_dict_(m.state_dict(), 1); m.train()
tr = ArenaTrainer(m)
batch = syn.make_batch(1, 4, 8, device=dev)
tr.step(batch, 0)
orig = kn.cast_f32_to_bf16
def spy(src, dst, n):
    print("cast", tuple(src.shape), "compute", kn.get_compute(), "base", kn.base_mode(), "bwd", kn.backward_compute())
    traceback.print_stack(limit=4)
    return orig(src, dst, n)
kn.cast_f32_to_bf16 = spy
ob = kn.spatial_softmax_bwd
def spy2(x, N, HW, C, xmap, ymap, t, out, stats, dout, dx, relu_mask=True):
    print("ssm bwd x", x.dtype, "dx", dx.dtype, "compute", kn.get_compute(), "base", kn.base_mode())
    return ob(x, N, HW, C, xmap, ymap, t, out, stats, dout, dx, relu_mask=relu_mask)
kn.spatial_softmax_bwd = spy2
tr.step(batch, 1)
torch.cuda.synchronize()
