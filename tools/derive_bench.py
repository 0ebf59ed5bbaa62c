"""Optimizer tail: Adam + the derived weight copies, old launch sequence (round 3: Adam, transposed tiles, fragment gather, residual, remainder
gather, conv repack) against the fused one (round 4: Adam writes the remainders; derive_copies; gather_chunks2).  hipGraph replay of 10 repeats."""
import sys

import torch

sys.path.insert(0, '.')
from hulc2_amd import kernels as kn, synthetic as syn
from hulc2_amd.compat import instantiate
from hulc2_amd.config import default_model_config
from hulc2_amd.trainer import ArenaTrainer

dev = torch.device("cuda")
kn.set_compute("bf16")
m = instantiate(default_model_config(gripper_control=True, dropout_p=0.1)).to(dev)
syn.fill_state_dict_(m.state_dict(), 3)
m.train()
tr = ArenaTrainer(m)
tr.flat_g.normal_()
st = kn.step_state(dev)


def old():
    kn.adam_step(tr.flat_p, tr.flat_g, tr.exp_avg, tr.exp_avg_sq, tr.flat_bf16, tr.total, 2e-4, 0.9, 0.999, 1e-8, 0.0, 1, step_state_dev=st)
    kn.transpose_bf16_tiles(tr.flat_bf16, tr.flat_bf16_t, tr.tiles_t)
    kn.gather_chunks(tr.flat_bf16, tr.flat_bf16_t, tr.frag_shadow, tr.frag_idx)
    kn.residual_bf16(tr.flat_p, tr.flat_bf16, tr.flat_lo, tr.lo_seg)
    kn.gather_chunks(tr.flat_lo, None, tr.lo_frag, tr.lo_frag_idx)
    kn.repack_conv_weights(tr.flat_p, tr.conv_shadow, tr.conv_table)


def new():
    kn.adam_step(tr.flat_p, tr.flat_g, tr.exp_avg, tr.exp_avg_sq, tr.flat_bf16, tr.total, 2e-4, 0.9, 0.999, 1e-8, 0.0, 1, step_state_dev=st,
                 lo=tr.flat_lo, lo_ranges=tr.lo_ranges)
    kn.derive_copies(tr.flat_bf16, tr.flat_bf16_t, tr.tiles_t, tr.flat_p, tr.conv_shadow, tr.conv_table)
    kn.gather_chunks2(tr.flat_bf16, tr.flat_bf16_t, tr.frag_shadow, tr.frag_idx, tr.flat_lo, tr.lo_frag, tr.lo_frag_idx)


def adam_only():
    kn.adam_step(tr.flat_p, tr.flat_g, tr.exp_avg, tr.exp_avg_sq, tr.flat_bf16, tr.total, 2e-4, 0.9, 0.999, 1e-8, 0.0, 1, step_state_dev=st)


def timeit(fn, rep=10):
    fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            for _ in range(rep):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * rep) * 1e3


print(f"params {tr.total / 1e6:.2f} M, lo ranges {tr.lo_ranges}")
print(f"adam alone                     {timeit(adam_only):7.1f} us")
print(f"round-3 sequence (6 launches)  {timeit(old):7.1f} us")
print(f"round-4 sequence (3 launches)  {timeit(new):7.1f} us")
