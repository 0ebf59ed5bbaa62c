"""gpurun_out/{aff_stats.txt, pa.log, bench_aff_default.log} (tools/profile_affordance.sh + `python3 bench.py --affordance`) -> profiles/r<NN>_aff_kernel_stats.txt"""
import os, subprocess
RN = os.environ.get('HULC_ROUND', '02')
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
eager = [l for l in open('gpurun_out/pa.log').read().splitlines() if l.startswith('{"metric"')][-1]
bench = [l for l in open('gpurun_out/bench_aff_default.log').read().splitlines() if l.startswith('{"metric"')][-1]
open(f'profiles/r{RN}_aff_kernel_stats.txt', 'w').write(
    f"# round {int(RN)}, commit {commit}, 1x MI355X, bf16 compute, BASELINE configs[4] (secondary): affordance model, 32 images of 224 x 224 per step\n"
    "# rocprofv3 --kernel-trace --stats -- python3 bench.py --affordance --batch 32 --steps 5 --warmup 2 --no-cpu-baseline --no-graph   (10 profiled steps incl. warmup "
    "+ 3 roofline-leg steps; eager launches)\n# bench line of the profiled (eager) run: " + eager[:330] + "\n# default bench line (hipGraph replay) of the same build: "
    + bench + "\n" + open('gpurun_out/aff_stats.txt').read())
