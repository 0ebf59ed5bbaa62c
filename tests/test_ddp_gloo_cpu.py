"""Data-parallel path on CPU: world_size-2 gloo run of the gradient arena + bucketed all-reduce of
hulc2_amd/trainer.py (the RCCL path uses the same code with backend 'nccl')."""
import os
import socket
import sys
from pathlib import Path

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
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
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
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
