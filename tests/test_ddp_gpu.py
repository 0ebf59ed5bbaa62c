"""Two data-parallel ranks sharing ONE MI355X (gloo transport on device tensors): the multi-process training paths of
hulc2_amd/trainer.py on the GPU — eager bucketed all-reduce overlapped with backward, and hipGraph replay with the
arena all-reduce between the two graphs (what bench.py runs with --gpus N; there the backend is "nccl" = RCCL, one GPU per
rank).  Both ranks must stay bit-identical replicas.  Sharing a GPU between processes is exactly the situation the
device-wide-barrier RNN kernel must not run in, so those are switched off here (HULC_NO_RNN_WAVEFRONT, HULC_NO_MLP_CHAIN)."""
import datetime
import os
import socket
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
pytestmark = pytest.mark.gpu


def _join_or_end(procs, seconds):
    """wait for the ranks; a rank still alive after `seconds` is ended (this test started exactly these processes) so that neither the assertion
    nor the interpreter's exit waits on it"""
    for p in procs:
        p.join(seconds)
    stuck = [p for p in procs if p.is_alive()]
    for p in stuck:
        p.kill()
        p.join(10)
    assert not stuck, f"{len(stuck)} rank(s) did not finish within {seconds} s"
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, graph, out, real_world=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HULC_NO_RNN_WAVEFRONT="1", HULC_NO_MLP_CHAIN="1")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))     # (default: 30 min of waiting for a rank that died)
    from hulc2_amd import kernels as kn, synthetic as syn
    from hulc2_amd.compat import instantiate
    from hulc2_amd.config import default_model_config, real_world_model_config
    from hulc2_amd.trainer import ArenaTrainer

    kn.set_compute("bf16")
    model = instantiate(real_world_model_config(dropout_p=0.1) if real_world else default_model_config(gripper_control=True, dropout_p=0.1)).to(dev)
    syn.fill_state_dict_(model.state_dict(), 42)
    model.train()
    tr = ArenaTrainer(model, lr=2e-4, overlap=not graph)
    batch = syn.make_batch(100 + rank, 2, 8, device=dev, **({"static_hw": (150, 200)} if real_world else {}))          # different data per rank
    for db in batch.values():
        db.pop("plan_idx", None)
        if real_world:                                                      # frozen R3M trunk: frames in [0, 255], its parameters are not in the arena
            db["rgb_obs"]["rgb_static"] = (db["rgb_obs"]["rgb_static"] + 1) * 127.5
    losses = []
    if graph:
        tr.capture(batch)
        for _ in range(3):
            losses.append(float(tr.replay()))
    else:
        for i in range(3):
            losses.append(float(tr.step(batch, i)))
    torch.cuda.synchronize()
    p = tr.flat_p.double()
    mine = torch.stack([p.sum(), (p * p).sum()]).cpu()
    both = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    ok = all(torch.isfinite(torch.tensor(losses))) and torch.equal(both[0], both[1])
    if rank == 0:
        out.put((bool(ok), losses, [b.tolist() for b in both]))
    dist.destroy_process_group()


@pytest.mark.parametrize("graph,real_world", [(False, False), (True, False), (True, True)])
def test_two_ranks_stay_identical(graph, real_world):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, graph, q, real_world)) for r in range(2)]
    for p in procs:
        p.start()
    _join_or_end(procs, 300)
    ok, losses, sums = q.get(timeout=5)
    assert ok, f"replicas diverged or non-finite loss: losses {losses}, parameter checksums {sums}"


def _bench(args, env_extra, timeout=600):
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_gpus_flag_starts_two_ranks():
    """`python bench.py --gpus 2` from a plain shell = 2 rank processes (gloo transport here: both share this box's one GPU; on a
    multi-GPU node the backend is nccl = RCCL, one GPU per rank).  The line must say n_gpus 2 and come from a 2-rank process group."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    line = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--seq-len", "8", "--no-cpu-baseline", "--no-secondary"],
                  {"HULC_BENCH_BACKEND": "gloo", "HULC_NO_RNN_WAVEFRONT": "1", "HULC_NO_MLP_CHAIN": "1"})   # two ranks share one GPU: no device-wide barrier kernels
    assert line["n_gpus"] == 2 and line["n_ranks_seen"] == 2 and line["config"]["parallelism"] == "dp2"
    assert line["value"] > 0 and line["config"]["final_loss"] == line["config"]["final_loss"]        # finite
    # what an N-GPU number is made of (VERDICT r02 next #3a): algorithm, payload, bytes, exposed communication time
    comm = line["comm"]
    assert comm["algo"] in ("ring", "direct") and comm["payload"] == "fp32" and comm["comm_exposed_ms"] >= 0.0
    assert comm["gradient_bytes"] > 180e6 and comm["bytes_sent_per_rank_per_step"] == comm["gradient_bytes"]       # 2 (W - 1) / W = 1 at W = 2


def test_affordance_bench_runs_on_two_ranks():
    """BASELINE configs[4] is a DDP run: `bench.py --affordance --gpus 2` (gloo transport, both ranks on this box's one GPU) — the same data
    parallel trainer, per-rank BatchNorm statistics as in the reference (no sync_batchnorm in conf/affordance/train_affordance.yaml)"""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    line = _bench(["--affordance", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--no-cpu-baseline", "--no-secondary"],
                  {"HULC_BENCH_BACKEND": "gloo", "HULC_NO_RNN_WAVEFRONT": "1", "HULC_NO_MLP_CHAIN": "1"})
    assert line["n_gpus"] == 2 and line["n_ranks_seen"] == 2 and line["unit"] == "images/s" and line["value"] > 0
    assert line["config"]["final_loss"] == line["config"]["final_loss"]


@pytest.mark.parametrize("algo,payload,graph", [("ring", "fp32", True), ("ring", "fp32", False), ("direct", "bf16", True), ("direct", "fp32", False)])
def test_rccl_path_executes_single_rank(algo, payload, graph):
    """The RCCL code path on real hardware: process group "nccl" (device_id bound), the comm-stream collectives between the split training
    graphs captured with capture_error_mode=thread_local (graph mode) or bucket hooks overlapped with backward (eager mode), with the
    barrier RNN kernel ON in graph mode.  One rank is all a one-GPU box allows: every collective still goes through RCCL."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    args = ["--force-dist", "--steps", "3", "--warmup", "1", "--batch", "4", "--seq-len", "16", "--no-cpu-baseline", "--no-secondary"]
    if not graph:
        args.append("--no-graph")
    line = _bench(args, {"HULC_ALLREDUCE": algo, "HULC_GRAD_PAYLOAD": payload})
    assert line["n_ranks_seen"] == 1 and "nccl" in line["config"]["gradient_allreduce"] and algo in line["config"]["gradient_allreduce"]
    assert ("hipGraph" in line["config"]["launch"]) == graph
    ref = _bench([a for a in args if a != "--force-dist"], {})
    # single rank: the reduced gradients equal the local ones (fp32 exactly; bf16 payload rounds them once), so the loss after 4 optimizer
    # steps tracks the run without a process group
    # (eager mode: gradients go through autograd's AccumulateGrad instead of the sinks and the per-step RNN path replaces the barrier kernel)
    tol = 1e-5 if (payload == "fp32" and graph) else 2e-2
    assert abs(line["config"]["final_loss"] - ref["config"]["final_loss"]) <= tol * abs(ref["config"]["final_loss"]), (line["config"], ref["config"])


def _locked_worker(rank, world, port, lock_path, out, B, S):
    """configs[2]'s real control flow on a one-GPU box: graph mode, split training graphs, the device-wide-barrier kernels ON.  The two ranks share
    the GPU, so every stretch of GPU work runs under a file lock that synchronises the device before it is released (ArenaTrainer.gpu_section):
    the ranks take turns and a barrier kernel never meets a foreign kernel; the collectives (gloo) run outside the lock."""
    import contextlib
    import fcntl
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.pop("HULC_NO_RNN_WAVEFRONT", None)
    os.environ.pop("HULC_NO_MLP_CHAIN", None)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
    from hulc2_amd import functional as HF, kernels as kn, synthetic as syn
    from hulc2_amd.compat import instantiate
    from hulc2_amd.config import default_model_config
    from hulc2_amd.trainer import ArenaTrainer

    fd = os.open(lock_path, os.O_RDWR | os.O_CREAT)

    @contextlib.contextmanager
    def turn():
        fcntl.flock(fd, fcntl.LOCK_EX)
        try:
            yield
            torch.cuda.synchronize()
        finally:
            fcntl.flock(fd, fcntl.LOCK_UN)

    calls = {"rnn": 0, "chain": 0}
    rnn0, chain0 = kn.rnn_wavefront, kn.mlp_chain
    kn.rnn_wavefront = lambda *a, **k: (calls.__setitem__("rnn", calls["rnn"] + 1), rnn0(*a, **k))[1]
    kn.mlp_chain = lambda *a, **k: (calls.__setitem__("chain", calls["chain"] + 1), chain0(*a, **k))[1]
    kn.set_compute("bf16")
    with turn():
        model = instantiate(default_model_config(gripper_control=True, dropout_p=0.1)).to(dev)
        syn.fill_state_dict_(model.state_dict(), 42)
        model.train()
        tr = ArenaTrainer(model, lr=2e-4, overlap=False)
        batch = syn.make_batch(100 + rank, B, S, device=dev)
        for db in batch.values():
            db.pop("plan_idx", None)
    tr.gpu_section = turn
    tr._fault_every = 1
    losses = [float(tr.step(batch, 0))]
    tr.capture(batch)
    assert tr.graph_enc is not None                                          # the split graphs of world > 1
    for _ in range(3):
        losses.append(float(tr.replay()))
    with turn():
        kn.check_faults(dev)
        p = tr.flat_p.double()
        mine = torch.stack([p.sum(), (p * p).sum()]).cpu()
    both = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    ok = all(torch.isfinite(torch.tensor(losses))) and torch.equal(both[0], both[1])
    if rank == 0:
        out.put((bool(ok), losses, [b.tolist() for b in both], dict(calls)))
    dist.destroy_process_group()


def test_two_ranks_graph_mode_with_barrier_kernels_at_full_batch(tmp_path):
    """VERDICT r02 next #3d: 2 ranks, B = 32 per modality, S = 32 (BASELINE configs[2]'s per-GPU workload), hipGraph replay with the split
    training graphs and the all-reduce between them, recurrent + MLP-chain barrier kernels ON — replicas stay bit-identical, no fault."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from hulc2_amd import kernels as kn
    if kn.device_cu_count(torch.device("cuda:0")) < 256:
        pytest.skip("the barrier kernels are gated off on this device")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_locked_worker, args=(r, 2, port, str(tmp_path / "gpu.lock"), q, 32, 32)) for r in range(2)]
    for p in procs:
        p.start()
    _join_or_end(procs, 600)
    ok, losses, sums, calls = q.get(timeout=5)
    assert ok, f"replicas diverged or non-finite loss: losses {losses}, parameter checksums {sums}"
    assert calls["rnn"] >= 2 and calls["chain"] >= 2, calls                 # the barrier kernels really were on the path
    assert losses[-1] < losses[0] * 1.5


def _torch_ddp_worker(port, out):
    """one process = one rank of an RCCL group on the GPU: Hulc2 under torch's own DistributedDataParallel with the parked comm hook"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    for k in ("HULC_NO_RNN_WAVEFRONT", "HULC_NO_MLP_CHAIN"):
        os.environ.pop(k, None)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=300))
    from torch.nn.parallel import DistributedDataParallel as DDP
    from hulc2_amd import kernels as kn, synthetic as syn
    from hulc2_amd.compat import instantiate
    from hulc2_amd.config import default_model_config
    from hulc2_amd.ddp import register_parked_comm_hook

    kn.set_compute("bf16")
    kn.set_concurrent_streams(False)                       # the cooperative kernels stay ON: that is the point of the hook
    calls = {"rnn": 0, "chain": 0}
    rnn0, chain0 = kn.rnn_wavefront, kn.mlp_chain
    kn.rnn_wavefront = lambda *a, **k: (calls.__setitem__("rnn", calls["rnn"] + 1), rnn0(*a, **k))[1]
    kn.mlp_chain = lambda *a, **k: (calls.__setitem__("chain", calls["chain"] + 1), chain0(*a, **k))[1]
    model = instantiate(default_model_config(gripper_control=True, dropout_p=0.0)).to(dev)
    syn.fill_state_dict_(model.state_dict(), 42)
    model.train()
    batch = syn.make_batch(7, 8, 32, device=dev)
    for db in batch.values():
        db.pop("plan_idx", None)

    class Step(torch.nn.Module):                           # Lightning's module wrapper: forward = training_step
        def __init__(self, m):
            super().__init__()
            self.module = m

        def forward(self, b, i):
            return self.module.training_step(b, i)

    kn.reset_step_state(dev)
    model.training_step(batch, 0).backward()               # the plain call: reference bits
    ref = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    for p in model.parameters():
        p.grad = None
    ddp = DDP(Step(model), device_ids=[0], static_graph=True, find_unused_parameters=False)
    st = register_parked_comm_hook(ddp)
    st.keep_log = True
    same, logs = True, []
    for it in range(4):
        kn.reset_step_state(dev)
        st.log.clear()
        for p in model.parameters():
            p.grad = None
        ddp(batch, 0).backward()
        torch.cuda.synchronize()
        logs.append(list(st.log))
        got = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
        same = same and set(got) == set(ref) and all(torch.equal(got[n], ref[n]) for n in ref)
    kn.check_faults(dev)                                    # no barrier kernel timed out next to the collectives
    out.put({"same": bool(same), "logs": logs, "calls": calls})
    dist.destroy_process_group()


def test_torch_ddp_with_parked_comm_hook_keeps_the_cooperative_kernels():
    """Hulc2 under torch.nn.parallel.DistributedDataParallel (what Lightning's DDPStrategy builds, hulc2/training.py:72-75) with
    hulc2_amd.ddp.register_parked_comm_hook on a one-rank RCCL group — the cooperative kernels (recurrent sweep, MLP chains) run, no fault, the
    averaged gradients are the bits of the plain backward, the bucket all-reduces of everything behind the camera encoders are parked until the
    encoder output's gradient exists and the encoders' own bucket goes out after it."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_torch_ddp_worker, args=(_free_port(), q))
    p.start()
    _join_or_end([p], 300)
    res = q.get(timeout=5)
    assert res["same"], "gradients under DDP + parked hook differ from the plain backward"
    assert res["calls"]["rnn"] >= 2 and res["calls"]["chain"] >= 2, res["calls"]
    print(res["logs"])
    # (static_graph: the reducer delays every all-reduce of the first two passes to their end while it learns the graph and rebuilds its
    # buckets — nothing to park there)
    for log in res["logs"][2:]:
        assert log.count("release") == 1 and log.count("park") >= 2 and "send" not in log[:log.index("release")], log
        # (round 4's loop: the encoders' bucket becomes ready after the release and is sent without parking — send > park; with the step node
        #  (round 5) every gradient of the pass reaches the reducer at once, behind the whole captured backward: all buckets are parked and
        #  released at the last one — send == park)
        assert log.count("send") >= log.count("park"), log
