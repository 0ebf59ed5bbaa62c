#!/usr/bin/env python3
"""Headline benchmark: play-sequences/sec/node of the HULC++ low-level policy training step on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank / GPU)

A "step" = one full optimizer step of `Hulc2.training_step` (forward + backward + gradient all-reduce + Adam) over
one device-resident synthetic CALVIN-shaped batch: 2 modalities x 32 play sequences x 32 timesteps, 200x200 static +
84x84 gripper RGB, random (B,384) language embeddings (BASELINE.json configs[1]/[2]).  Weak scaling:
each rank processes its own 64 sequences per step.

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     — the dominant kernel's algorithmic FLOP/s from HIP events recorded live on the launch stream
  cpu_baseline — the CPU oracle (fp32 torch restatement, kind "port") timed on this box's host cores on a bounded sample
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

SEQ_FLOP_TRAIN = 14.13e9        # 3 x 4.710 GFLOP forward per play sequence
PEAK_BF16 = 2.5e15              # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32 = 157.3e12
PEAK_HBM = 8.0e12               # HBM3E bytes/s (MI355X_MICROARCH.md)


# device kernel behind a timed (entry point, shape): kernel name prefix in the committed --pmc passes.  One kernel instance serves the
# static and the gripper camera (told apart in the passes by LDS size / grid): "max" = its launches with the larger traffic (static frames)
PMC_KERNEL = {
    ("conv2d_bwd_weight", 1024, 200, 200, 3, 32, 8, 4): ("conv1_wgrad_kernel<3, 2, false>", "max"),
    ("conv2d_bwd_weight", 1024, 84, 84, 3, 32, 8, 4): ("conv1_wgrad_kernel<3, 2, false>", "min"),
    ("conv2d_fwd", 1024, 200, 200, 3, 32, 8, 4): ("conv1_band_kernel<3, false>", "max"),
    ("conv2d_fwd", 1024, 84, 84, 3, 32, 8, 4): ("conv1_band_kernel<3, false>", "min"),
    ("conv2d_bwd_data", 2048, 49, 49, 32, 64, 4, 2): ("conv_band_kernel<64, 4, 2, 2, 1, 12, false, false>", "max"),
    ("conv2d_fwd", 2048, 49, 49, 32, 64, 4, 2): ("conv_band_kernel<32, 2, 4, 4, 2, 12, false, false>", "max"),
    ("conv2d_bwd_weight", 2048, 49, 49, 32, 64, 4, 2): ("conv_wgrad_band_kernel<32, 2, 4, 4, 2, false, 10, 5, 1, true>", "max"),
    ("rnn_wavefront", 32, 64, 2048, 1): ("rnn_wavefront_kernel<2048, true>", "max"),
    ("rnn_wavefront", 32, 64, 2048, 0): ("rnn_wavefront_kernel<2048, false>", "max"),
    ("hulc_adam_step",): ("adam_kernel", "max"),
}


def pmc_traffic(key):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this same command
    (profiles/*_pmc_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate passes, FETCH_SIZE doubled per the gfx950 note of
    MI355X_MICROARCH.md).  bench.py cannot run the profiler on itself; None when no committed measurement names the kernel."""
    want = PMC_KERNEL.get(tuple(key))
    files = sorted((ROOT / "profiles").glob("*_pmc_traffic.json"))
    if not want or not files:
        return None
    hits = [rec["traffic_bytes"] for name, rec in json.loads(files[-1].read_text()).get("kernels", {}).items() if name.startswith(want[0])]
    if not hits:
        return None
    return round(max(hits) if want[1] == "max" else min(hits))


def cpu_baseline(seconds_budget=25.0):
    """oracle training step (fwd + bwd + Adam) on the host cores, B=4/modality, S=32, fp32"""
    from hulc2_amd import param_spec, synthetic as syn
    from oracle import hulc2_oracle as O

    B, S = 4, 32
    torch.manual_seed(0)
    sd = {k: torch.empty(s) for k, s in param_spec.trainable_shapes().items()}
    syn.fill_state_dict_(sd, 42)
    for v in sd.values():
        v.requires_grad_(True)
    opt = torch.optim.Adam(list(sd.values()), lr=2e-4)
    raw = syn.make_batch(42, B, S)
    batch = {}
    for m, db in raw.items():
        batch[m] = dict(rgb_static=db["rgb_obs"]["rgb_static"], rgb_gripper=db["rgb_obs"]["rgb_gripper"], actions=db["actions"],
                        robot_obs=db["state_info"]["robot_obs"], plan_idx=db["plan_idx"])
        if m == "lang":
            batch[m].update(lang=db["lang"], use_for_aux_lang_loss=db["use_for_aux_lang_loss"])
    times = []
    t_start = time.time()
    for i in range(6):
        t0 = time.time()
        opt.zero_grad(set_to_none=True)
        out = O.training_step(sd, batch, dict(gripper_control=True))
        out["total_loss"].backward()
        opt.step()
        times.append(time.time() - t0)
        if time.time() - t_start > seconds_budget and i >= 1:
            break
    t = sorted(times[1:] or times)[len(times[1:] or times) // 2]
    return {"value": round(2 * B / t, 3), "unit": "play-sequences/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{len(times)} steps of B={B}/modality S={S} (8 sequences/step), fp32 torch CPU oracle, median of steps after the first"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="play sequences per modality per GPU")
    ap.add_argument("--seq-len", type=int, default=32)
    ap.add_argument("--compute", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying captured HIP graphs")
    ap.add_argument("--breakdown", action="store_true", help="print the per-kernel time table to stderr")
    ap.add_argument("--uint8-frames", action="store_true",
                    help="feed uint8 NHWC frames + shift-augmentation offsets (SURVEY 8 row f-2) instead of transformed fp32 frames; "
                         "a separate data format, not the headline configuration")
    ap.add_argument("--episode-store", action="store_true",
                    help="row f-2 end to end: an HBM-resident uint8 episode store per modality, every step draws new play windows "
                         "(index rows, pad-by-repetition, shifts) and conv1 reads the store in place; separate from the headline configuration")
    ap.add_argument("--store-frames", type=int, default=16384, help="frames in the synthetic episode store (141 KB each)")
    ap.add_argument("--real-world", action="store_true",
                    help="secondary measurement, BASELINE configs[3] (cfg_low_level_rw): static camera 150x200 in [0,255] through the frozen R3M "
                         "ResNet-18 trunk, whole-embedding decoder input, world-frame actions, no CLIP loss")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    # HULC_BENCH_BACKEND=gloo: functional check of the multi-rank control flow on a box with fewer GPUs than ranks (ranks then share devices;
    # run it with HULC_NO_RNN_WAVEFRONT=1 — the device-wide-barrier kernel needs the GPU to itself).  The measured configuration is nccl (= RCCL).
    backend = os.environ.get("HULC_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from hulc2_amd import kernels as kn, synthetic as syn
    from hulc2_amd.compat import instantiate
    from hulc2_amd.config import default_model_config, real_world_model_config
    from hulc2_amd.trainer import ArenaTrainer

    kn.set_compute(args.compute)
    if args.real_world and (args.uint8_frames or args.episode_store):
        raise SystemExit("--real-world takes fp32 frames in [0,255] (conf/datamodule/transforms/real_world_r3m.yaml); the uint8 store feeds the CNN config")
    cfg = real_world_model_config(dropout_p=0.1) if args.real_world else default_model_config(gripper_control=True, dropout_p=0.1)
    model = instantiate(cfg).to(dev)
    syn.fill_state_dict_(model.state_dict(), 42)            # same weights on every rank
    model.train()
    use_graph = not args.no_graph
    trainer = ArenaTrainer(model, lr=2e-4, overlap=not use_graph)
    batch = syn.make_batch(42 + rank, args.batch, args.seq_len, device=dev, **({"static_hw": (150, 200)} if args.real_world else {}))
    for db in batch.values():
        db.pop("plan_idx", None)                            # benchmark samples the latent plan on-device
        if args.real_world:                                 # UpScaleImageTensor: the R3M trunk takes [0, 255]
            db["rgb_obs"]["rgb_static"] = (db["rgb_obs"]["rgb_static"] + 1) * 127.5
    if args.uint8_frames:
        g = torch.Generator().manual_seed(1234 + rank)
        for db in batch.values():
            obs = {}
            for key, hw, pad in (("rgb_static", 200, 10), ("rgb_gripper", 84, 4)):
                obs[key] = torch.randint(0, 256, (args.batch, args.seq_len, hw, hw, 3), generator=g, dtype=torch.uint8).to(dev)
                obs[key + "_shift"] = torch.randint(0, 2 * pad + 1, (args.batch, args.seq_len, 2), generator=g, dtype=torch.int32).to(dev)
            db["rgb_obs"] = obs

    stores, draw = None, None
    if args.episode_store:
        import numpy as np
        from hulc2_amd.datasets import DeviceEpisodeStore
        g = torch.Generator().manual_seed(99 + rank)
        n = args.store_frames
        rgb = {"rgb_static": torch.randint(0, 256, (n, 200, 200, 3), generator=g, dtype=torch.uint8).to(dev),
               "rgb_gripper": torch.randint(0, 256, (n, 84, 84, 3), generator=g, dtype=torch.uint8).to(dev)}
        act = torch.rand(n, 7, generator=g) * 2 - 1
        act[:, 6] = torch.where(act[:, 6] > 0, 1.0, -1.0)
        obs = torch.randn(n, 15, generator=g)
        obs[:, 3:6] = (torch.rand(n, 3, generator=g) * 2 - 1) * 3.14159 * 0.5
        eps = [(a, min(a + 511, n - 1)) for a in range(0, n, 512)]           # 512-frame play episodes
        # vision windows 20..32 (conf/datamodule/datasets/vision_dataset/vision_shm.yaml:5-6); language windows 20..32 over 64-frame
        # annotated spans in the auto_lang_ann.npy layout, aux-loss window 8 (lang_dataset/lang_shm.yaml:5-6,12)
        stores = {"vis": DeviceEpisodeStore(rgb, act, obs, eps, 20, 32, device=dev, seed=rank)}
        spans = [(a + 64 * j, a + 64 * j + 63) for a, b in eps for j in range(0, (b - a + 1) // 64, 2)]
        lang_data = {"language": {"ann": [""] * len(spans), "task": [""] * len(spans),
                                  "emb": (torch.randn(len(spans), 1, 384, generator=g) * 0.05).numpy()}, "info": {"indx": spans}}
        stores["lang"] = DeviceEpisodeStore.from_language_annotations(stores["vis"].rgb, act, obs, lang_data, device=dev, seed=rank)
        sampler = np.random.RandomState(1000 + rank)

        def draw():
            for mod, st in stores.items():
                fresh = st.batch(sampler.randint(0, len(st), args.batch))
                fresh.pop("window_sizes")
                batch[mod] = fresh                                            # same device buffers every step: graph replay sees the new windows
        draw()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    loss = None
    if use_graph:
        for i in range(2):                                  # eager steps: allocator pools + lazy buffers before capture
            loss = trainer.step(batch, i)
        try:
            trainer.capture(batch)
        except Exception as e:                              # noqa: BLE001 - report and fall back to eager launches
            print(f"[bench] graph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            use_graph = False
        if world > 1:                                       # graph and eager modes issue different collectives: all ranks take the same one
            ok = torch.tensor([1 if use_graph else 0], device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            use_graph = bool(ok.item())
    run_step = (lambda i: trainer.replay()) if use_graph else (lambda i: trainer.step(batch, i))
    if draw is not None:
        inner = run_step

        def run_step(i):
            draw()
            return inner(i)
    for i in range(args.warmup):
        loss = run_step(i)
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = run_step(i)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(loss)

    # ---- roofline leg: per-launch HIP events on the launch stream over 3 more eager steps (outside the timed region) ----
    # The dominant kernel = the (entry point, shape) with the largest summed time.  Its bound is whichever roof needs longer for
    # the launch's ALGORITHMIC work (kernels.py annotates flops = 2 x MACs and bytes = every operand / result touched once):
    # bytes / 8 TB/s vs flops / dense MFMA peak.  achieved = that work / the measured average launch duration.
    kn.start_timing()
    for i in range(3):
        trainer.step(batch, i)
    table = kn.stop_timing()
    total_ms = sum(v[1] for v in table.values())
    peak = PEAK_BF16 if args.compute == "bf16" else PEAK_F32
    dom_key, (dom_n, dom_t, dom_flops, dom_bytes) = max(table.items(), key=lambda kv: kv[1][1])
    avg_s = dom_t / dom_n * 1e-3
    hbm_bound = dom_bytes / PEAK_HBM >= dom_flops / peak
    if hbm_bound:
        rl = {"bound": "hbm", "achieved": round(dom_bytes / avg_s / 1e9, 1), "peak": PEAK_HBM / 1e9, "unit": "GB/s",
              "frac": round(dom_bytes / avg_s / PEAK_HBM, 4)}
    else:
        rl = {"bound": "mfma", "achieved": round(dom_flops / avg_s / 1e12, 2), "peak": peak / 1e12, "unit": "TFLOP/s",
              "frac": round(dom_flops / avg_s / peak, 4)}
    rl["traffic"] = pmc_traffic(dom_key)
    if args.breakdown and rank == 0:
        rows = int(os.environ.get("HULC_BREAKDOWN_ROWS", "25"))
        for key, (n, t, f, b) in sorted(table.items(), key=lambda kv: -kv[1][1])[:rows]:
            a = t / n * 1e-3
            print(f"  {t / 3:9.3f} ms/step  {n // 3:4d} launches  {f / a / 1e12:8.1f} TFLOP/s  {b / a / 1e9:8.0f} GB/s  {key}", file=sys.stderr)
        print(f"  sum of kernel time: {total_ms / 3:.3f} ms/step", file=sys.stderr)

    seqs = 2 * args.batch * world * args.steps
    ms_per_step = elapsed / args.steps * 1e3
    value = seqs / elapsed
    # real-world config: frozen trunk forward only (1 167 MMAC per 150x200 frame, by closed form over the 20 convolutions) + 3x the trained part
    seq_flop = (2 * 1167.0e6 * 32 + 3 * 2 * (237.37 + 59.24 + 15.01 + 4.78 + 492.2 + 0.15) * 1e6) if args.real_world else SEQ_FLOP_TRAIN
    workload = ("BASELINE configs[3] (secondary): cfg_low_level_rw — static 150x200 in [0,255] through the frozen R3M ResNet-18 trunk (random "
                "weights), gripper CNN 84x84, decoder on the whole embedding, world-frame actions, no CLIP loss, lang = random (B,384) embeddings"
                if args.real_world else
                "BASELINE configs[1]: synthetic CALVIN-shaped batch, Hulc2.training_step fwd+bwd+allreduce+Adam, "
                "static CNN 200x200 + gripper CNN 84x84, lang = random (B,384) embeddings, dropout 0.1, gripper_control on")
    out = {
        "metric": "play-sequences/sec/node (seq_len=32, real-world cfg: R3M static 150x200)" if args.real_world
                  else "play-sequences/sec/node (seq_len=32, 200x200 RGB)",
        "value": round(value, 2), "unit": "play-sequences/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.compute if args.compute == "bf16" else "f32", "data": "synthetic",
        "config": {"workload": workload,
                   "sequences_per_gpu_step": 2 * args.batch, "seq_len": args.seq_len, "parallelism": f"dp{world}",
                   "launch": "hipGraph replay (fwd+bwd graph, all-reduce, optimizer graph)" if use_graph else "eager launches",
                   "frames": "HBM-resident uint8 episode store, new play windows (20..32 steps, padded by repetition) every step, conv1 reads "
                             "the store through index rows" if args.episode_store
                             else "uint8 NHWC + RandomShiftsAug offsets, scaled/normalised while staging conv1" if args.uint8_frames
                             else "fp32 NCHW, already transformed (the reference's dataloader output)",
                   "final_loss": round(final_loss, 4)},
        "roofline": {**rl, "kernel": "/".join(str(k) for k in dom_key),
                     "algorithmic_bytes_per_launch": dom_bytes, "algorithmic_flops_per_launch": dom_flops,
                     "launches_per_step": dom_n // 3, "avg_launch_ms": round(dom_t / dom_n, 4),
                     "kernel_share_of_step": round(dom_t / max(total_ms, 1e-9), 3),
                     "step_frac_of_mfma_peak": round(value / world * seq_flop / peak, 4),
                     "gpu_kernel_ms_per_step": round(total_ms / 3, 3)},
    }
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
