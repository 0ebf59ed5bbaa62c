"""Turn the outputs of tools/profile_round.sh (gpurun_out/*) into the committed files profiles/r<NN>_<tag>_* (round = $HULC_ROUND, default 02): kernel stats, the FETCH_SIZE and
WRITE_SIZE passes and the per-launch HBM traffic json bench.py reads.  usage: python tools/collect_profiles.py <tag>"""
import json, re, subprocess, sys
tag = sys.argv[1]
import os
RN = os.environ.get('HULC_ROUND', '02')
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
bench = [l for l in open('gpurun_out/bench_default.log').read().splitlines() if l.startswith('{"metric"')][-1]
eager = [l for l in open('gpurun_out/pk.log').read().splitlines() if l.startswith('{"metric"')][-1]
hdr = f"# round {int(RN)}, commit {commit}, 1x MI355X, bf16 compute, 64 play-sequences per step\n"
open(f'profiles/r{RN}_{tag}_kernel_stats.txt', 'w').write(
    hdr + "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph   (10 profiled steps incl. warmup + 3 "
    "roofline-leg steps; eager launches: rocprofv3 cannot trace through hipGraph capture on this image)\n# bench line of the profiled (eager) run: "
    + eager[:330] + "\n# default bench line (hipGraph replay) of the same build: " + bench + "\n" + open('gpurun_out/k_stats.txt').read())
for t, ctr in (('f', 'FETCH_SIZE'), ('w', 'WRITE_SIZE')):
    open(f'profiles/r{RN}_{tag}_pmc_{ctr.lower()}.txt', 'w').write(
        hdr + f"# rocprofv3 --pmc {ctr} --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph  (own pass; unit KB; launches of "
        "one kernel instance are told apart by LDS size and grid)\n" + open(f'gpurun_out/{t}_pmc.txt').read())


if os.path.exists('gpurun_out/m_mfma.txt'):
    open(f'profiles/r{RN}_{tag}_pmc_mfma.txt', 'w').write(
        hdr + "# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python3 bench.py --steps 2 "
        "--warmup 1 --no-cpu-baseline --no-secondary --no-graph   (own pass, SQ + GRBM counters only; sums over all launches of a kernel)\n"
        + open('gpurun_out/m_mfma.txt').read())


def parse(path):
    d = {}
    for l in open(path).read().splitlines()[1:]:
        m = re.match(r"\s*(\d+)\s+([\d.]+)\s+(\w+)\s+(\d+)\s+(\d+)\s+(.*)", l)
        if m:
            d[(m.group(6).strip(), int(m.group(4)), int(m.group(5)))] = (int(m.group(1)), float(m.group(2)))
    return d


f, w = parse('gpurun_out/f_pmc.txt'), parse('gpurun_out/w_pmc.txt')
out = {"note": f"per-launch averages from the two --pmc passes in this directory (r{RN}_{tag}_pmc_*.txt); key = kernel @lds=<LDS bytes> @grid=<work-items>; "
               "traffic_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 FETCH_SIZE halving, MI355X_MICROARCH.md HBM section)", "kernels": {}}
for k, (n, fv) in f.items():
    wv = w.get(k, (0, 0.0))[1]
    if fv + wv >= 20000:
        out["kernels"][f"{k[0]} @lds={k[1]} @grid={k[2]}"] = {"launches": n, "fetch_size_kb": fv, "write_size_kb": wv, "traffic_bytes": (2 * fv + wv) * 1024}
json.dump(out, open(f'profiles/r{RN}_{tag}_pmc_traffic.json', 'w'), indent=1)
print(json.loads(bench)["roofline"])
for k, v in list(out["kernels"].items())[:8]:
    print(k[:100], round(v["traffic_bytes"] / 1e6))
