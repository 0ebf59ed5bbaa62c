"""Data-parallel path on CPU: world_size-2 gloo run of the gradient arena + bucketed all-reduce of
hulc2_amd/trainer.py (the RCCL path uses the same code with backend 'nccl')."""
import datetime
import os
import socket
import sys
from pathlib import Path

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


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


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))     # (default: 30 min of waiting for a rank that died)
    from hulc2_amd.trainer import ArenaTrainer

    torch.manual_seed(0)                                   # identical replicas
    model = torch.nn.Sequential(torch.nn.Linear(37, 64), torch.nn.ReLU(), torch.nn.Linear(64, 50), torch.nn.ReLU(),
                                torch.nn.Linear(50, 3))
    unused = torch.nn.Parameter(torch.ones(5))             # a parameter that never receives a gradient
    model.register_parameter("unused", unused)
    import copy
    ref_model = copy.deepcopy(model)                       # hook-free replica: gives this rank's local gradients
    tr = ArenaTrainer(model, bucket_mb=0)                  # bucket_mb=0 -> one bucket per parameter: exercises the hook logic
    assert len(tr.buckets.buckets) == len(tr.params)
    ok = True
    for step in range(3):
        g = torch.Generator().manual_seed(100 * step + rank)
        x = torch.randn(8, 37, generator=g)
        tr.zero_grad()
        loss = model(x).pow(2).mean()
        loss.backward()                                    # hooks launch the bucket all-reduces while autograd runs
        tr.buckets.finish()
        ref_model.zero_grad(set_to_none=True)
        ref_model(x).pow(2).mean().backward()
        local = torch.zeros_like(tr.flat_g)
        for p, off in zip(ref_model.parameters(), tr.offsets):
            if p.grad is not None:
                local[off:off + p.numel()] = p.grad.reshape(-1)
        gathered = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        want = sum(gathered)
        ok = ok and torch.allclose(tr.flat_g, want, atol=1e-6)
        for p, off in zip(tr.params, tr.offsets):          # .grad stays a view of the arena (no copies)
            ok = ok and p.grad.data_ptr() == tr.flat_g.data_ptr() + 4 * off
        ok = ok and p.data.data_ptr() >= tr.flat_p.data_ptr()
    if rank == 0:
        out.put(bool(ok))
    dist.destroy_process_group()


def test_gradient_buckets_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    _join_or_end(procs, 120)
    assert q.get(timeout=5) is True


def test_bucket_layout_is_contiguous_and_covers_arena():
    from hulc2_amd.trainer import ArenaTrainer

    model = torch.nn.Sequential(torch.nn.Linear(10, 20), torch.nn.Linear(20, 30), torch.nn.Linear(30, 7))
    tr = ArenaTrainer(model, bucket_mb=1)
    assert tr.total % 8 == 0 and all(o % 8 == 0 for o in tr.offsets)
    spans = sorted((b["lo"], b["hi"]) for b in tr.buckets.buckets)
    assert spans[0][0] == 0
    for (a, b), (c, d) in zip(spans, spans[1:]):
        assert b <= c
    covered = sum(b - a for a, b in spans)
    assert covered >= sum(p.numel() for p in tr.params)


def _comm_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))     # (default: 30 min of waiting for a rank that died)
    from hulc2_amd.trainer import GradComm

    n = 1003                                               # not a multiple of 8 * world: exercises the zero-padded tail chunk
    res = {}
    for algo in ("ring", "direct"):
        for payload in ("fp32", "bf16"):
            g = torch.Generator().manual_seed(7)
            base = [torch.randn(n, generator=g) for _ in range(world)]          # every rank can rebuild every rank's gradients
            arena = base[rank].clone()
            comm = GradComm(arena, None, algo, payload)
            comm.reserve()
            comm.reduce(5, n - 3)                          # an interior slice: the elements outside must stay untouched
            want = arena.clone()
            if payload == "fp32":
                want[5:n - 3] = sum(b[5:n - 3] for b in base)
                tol = 1e-6
            else:                                          # one bf16 rounding of the inputs (+ one of the sum for "direct")
                want[5:n - 3] = sum(b[5:n - 3].bfloat16().float() for b in base)
                tol = 4e-2
            inside = torch.allclose(arena[5:n - 3], want[5:n - 3], atol=tol, rtol=tol)
            outside = torch.equal(arena[:5], base[rank][:5]) and torch.equal(arena[n - 3:], base[rank][n - 3:])
            gathered = [torch.empty_like(arena) for _ in range(world)]
            dist.all_gather(gathered, arena)
            same = all(torch.equal(gathered[0][5:n - 3], t[5:n - 3]) for t in gathered)     # replicas bit-identical after the reduce
            res[(algo, payload)] = (inside, outside, same)
    # defaults and the description bench.py prints: "auto" resolves to ring on CPU (the timing probe is for GPUs on a real fabric), the payload
    # is fp32 at EVERY world size like the reference's DDP (bf16 is opt-in: HULC_GRAD_PAYLOAD / grad_payload), bytes on the wire = 2 (W - 1) / W x payload
    for k in ("HULC_ALLREDUCE", "HULC_GRAD_PAYLOAD"):
        os.environ.pop(k, None)
    c = GradComm(torch.zeros(n), None)
    d = c.describe()
    res["defaults"] = (c.algo == "ring", c.payload == "fp32", d["gradient_bytes"] == 4 * n and d["bytes_sent_per_rank_per_step"] == 4 * n * (world - 1) * 2 // world)
    res["rule"] = (GradComm.default_payload(8) == "fp32", GradComm.default_payload(4) == "fp32", GradComm.default_payload(2) == "fp32")
    # an opted-in bf16 payload with algo "auto" takes the fp32-accumulating direct exchange (one rounding), never a bf16 ring chosen by a timer;
    # the choice is recorded and can be re-applied (ArenaTrainer.load_state_dict -> pin)
    cb = GradComm(torch.zeros(n), None, None, "bf16")
    res["bf16_auto"] = (cb.algo == "direct", cb.describe()["chosen_by"].startswith("rule"), d["chosen_by"].startswith("rule"))
    c.pin("direct")
    res["pin"] = (c.algo == "direct", c.describe()["chosen_by"] == "checkpoint")
    # the timing probe itself (on a real fabric it runs when the trainer is built; here on the wall clock): both algorithms timed, every rank
    # draws the same conclusion, the gradients are untouched, no staging buffer survives
    g = torch.Generator().manual_seed(3 + rank)
    arena = torch.randn(4099, generator=g)
    before = arena.clone()
    cp = GradComm(arena, None, "ring", "fp32")
    cp.PROBE_ELEMS = 1024
    best = cp._probe()
    votes = [None] * world
    dist.all_gather_object(votes, best)
    res["probe"] = (best in ("ring", "direct"), len(set(votes)) == 1, set(cp.probe_ms) == {"ring", "direct"}, torch.equal(arena, before), cp._bufs == {})
    if rank == 0:
        out.put(res)
    dist.destroy_process_group()


def test_gradient_allreduce_algorithms_world2():
    """ring / direct (all-to-all + fixed-order local sum + all-gather) x fp32 / bf16 payload: same sums, untouched neighbours, identical replicas"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_comm_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    _join_or_end(procs, 120)
    res = q.get(timeout=5)
    assert len(res) == 9
    for key, flags in res.items():
        assert all(flags), f"{key}: (sum correct, neighbours untouched, replicas identical) = {flags}"


def test_bench_launcher_starts_n_ranks():
    """`python bench.py --gpus N` from a plain shell starts N rank processes; --dry-run = rendezvous only, no GPU"""
    import json
    import subprocess
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True, timeout=120,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")})
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                       # ONE JSON line on stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["n_ranks_seen"] == 2 and line["rank_sum"] == 3.0


def test_bench_launcher_propagates_rank_failure():
    """a failing rank ends the job with a non-zero exit code (here: no GPU in this container -> every rank refuses to run)"""
    import subprocess
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a machine without a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["HULC_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.strip().startswith("{")]


class _Toy(torch.nn.Module):
    """stands in for Hulc2 under torch DDP: `perceptual_encoder` first (its gradients are the LAST of a backward), a head of two large layers"""

    def __init__(self):
        super().__init__()
        self.perceptual_encoder = torch.nn.Linear(48, 64)
        self.head = torch.nn.Sequential(torch.nn.Linear(64, 4096), torch.nn.ReLU(), torch.nn.Linear(4096, 300))

    def forward(self, x):
        return self.head(self.perceptual_encoder(x)).pow(2).mean()


def _parked_hook_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    import copy
    from torch.nn.parallel import DistributedDataParallel as DDP
    from hulc2_amd.ddp import register_parked_comm_hook

    res = {}
    for frozen in (False, True):
        torch.manual_seed(0)
        model = _Toy()
        if frozen:                                        # the cut tensor never receives a gradient: the last bucket releases
            for p in model.perceptual_encoder.parameters():
                p.requires_grad_(False)
        ref = copy.deepcopy(model)
        ddp = DDP(model, bucket_cap_mb=1, static_graph=True)          # (the reference: DDPStrategy(static_graph=True), hulc2/training.py:72-75)
        st = register_parked_comm_hook(ddp)
        st.keep_log = True
        ok, logs = True, []
        for step in range(3):
            xs = [torch.randn(8, 48, generator=torch.Generator().manual_seed(10 * step + r)) for r in range(world)]
            st.log.clear()
            for p in model.parameters():
                p.grad = None
            ddp(xs[rank]).backward()
            logs.append(list(st.log))
            want = {}
            for r in range(world):                        # every rank can rebuild every rank's local gradients
                ref.zero_grad(set_to_none=True)
                ref(xs[r]).backward()
                for n, p in ref.named_parameters():
                    if p.grad is not None:
                        want[n] = want.get(n, 0) + p.grad / world
            for n, p in model.named_parameters():
                if p.requires_grad:
                    ok = ok and p.grad is not None and torch.allclose(p.grad, want[n], atol=1e-6, rtol=1e-5)
        # order: nothing is sent before the release; every parked bucket is sent right behind it; buckets after the cut go straight out
        order_ok = all(l.count("release") == 1 and "send" not in l[:l.index("release")] and l.count("send") >= l.count("park")
                       and l[l.index("release") + 1:l.index("release") + 1 + l.count("park")] == ["send"] * l.count("park") for l in logs)
        res["frozen" if frozen else "trainable"] = (ok, order_ok, logs[-1])
    if rank == 0:
        out.put(res)
    dist.destroy_process_group()


def test_ddp_parked_comm_hook_world2():
    """hulc2_amd.ddp.register_parked_comm_hook under torch DDP (gloo, 2 ranks): averaged gradients equal the ranks' mean, no all-reduce is issued
    before the cut (the encoder output's gradient) has been reached, the parked buckets follow it in order, later buckets go straight out; with
    a frozen encoder the last bucket is the release point."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_parked_hook_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    _join_or_end(procs, 120)
    res = q.get(timeout=5)
    for key, (ok, order_ok, last) in res.items():
        assert ok and order_ok, (key, ok, order_ok, last)
    # trainable encoder: at least one bucket parked AND at least one sent after the release without parking (the encoder's own bucket)
    last = res["trainable"][2]
    assert last.count("park") >= 1 and last.count("send") > last.count("park"), last
    assert res["frozen"][2].count("send") == res["frozen"][2].count("park"), res["frozen"][2]
