"""run-to-run spread of the mixed-mode step (exact-fp32 forward): the same graph-replayed step rebuilt several times in one process with the
allocator perturbed in between; HULC_EAGER=1 adds the per-kernel table of an eager step for each instance"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hulc2_amd import kernels as kn, synthetic as syn
from hulc2_amd.compat import instantiate
from hulc2_amd.config import default_model_config
from hulc2_amd.trainer import ArenaTrainer

dev = torch.device("cuda", 0)
junk = []
for rep in range(6):
    kn.set_compute("mixed")
    model = instantiate(default_model_config(gripper_control=True, dropout_p=0.1)).to(dev)
    syn.fill_state_dict_(model.state_dict(), 42)
    model.train()
    tr = ArenaTrainer(model, lr=2e-4, overlap=False)
    batch = syn.make_batch(42, 32, 32, device=dev)
    for db in batch.values():
        db.pop("plan_idx", None)
    for i in range(2):
        tr.step(batch, i)
    if os.environ.get("HULC_EAGER"):
        kn.start_timing()
        for i in range(3):
            tr.step(batch, i)
        rec = kn.stop_timing()
        top = sorted(rec.items(), key=lambda kv: -kv[1][1])[:8]
        print("  eager:", [(str(k)[:60], round(v[1] / 3, 3)) for k, v in top], flush=True)
    tr.capture(batch)
    for _ in range(2):
        tr.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        tr.replay()
    torch.cuda.synchronize()
    print(f"instance {rep}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step", flush=True)
    tr.close()
    del tr, model, batch
    gc.collect()
    torch.cuda.empty_cache()
    junk.append(torch.empty((37 + 61 * rep) << 20, dtype=torch.uint8, device=dev))      # shifts where the next instance's buffers land
