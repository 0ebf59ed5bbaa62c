"""gpurun_out/{aff_stats,rw_stats}.txt + logs (tools/profile_secondary.sh) -> profiles/r<NN>_aff_kernel_stats.txt, profiles/r<NN>_rw_kernel_stats.txt"""
import os, subprocess
RN = os.environ.get('HULC_ROUND', '03')
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
for tag, flag, what, log, blog, stats in (
        ("aff", "--affordance --batch 32", "BASELINE configs[4] (secondary): affordance model, 32 images of 224 x 224 per step", "pa.log", "bench_aff_default.log", "aff_stats.txt"),
        ("rw", "--real-world", "BASELINE configs[3] (secondary): cfg_low_level_rw, frozen R3M ResNet-18 static encoder on 150 x 200 frames, 64 play-sequences per step", "prw.log",
         "bench_rw_default.log", "rw_stats.txt")):
    eager = [l for l in open(f'gpurun_out/{log}').read().splitlines() if l.startswith('{"metric"')][-1]
    bench = [l for l in open(f'gpurun_out/{blog}').read().splitlines() if l.startswith('{"metric"')][-1]
    open(f'profiles/r{RN}_{tag}_kernel_stats.txt', 'w').write(
        f"# round {int(RN)}, commit {commit}, 1x MI355X, bf16 compute, {what}\n"
        f"# rocprofv3 --kernel-trace --stats -- python3 bench.py {flag} --steps 5 --warmup 2 --no-cpu-baseline --no-graph   (10 profiled steps incl. warmup "
        "+ 3 roofline-leg steps; eager launches)\n# bench line of the profiled (eager) run: " + eager[:330] + "\n# default bench line (hipGraph replay) of the same build: "
        + bench + "\n" + open(f'gpurun_out/{stats}').read())
    print(tag, "ok")
