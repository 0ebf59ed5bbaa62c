cd $GRAFT_REPO_ROOT
t0=$(date +%s)
python3 bench.py > gpurun_out/bench_full.log 2> gpurun_out/bench_full.err
t1=$(date +%s)
echo "wall $((t1-t0)) s"
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_full.log') if l.startswith('{"metric"')][-1])
print(d['ms_per_step'], d['value'], json.dumps(d['cpu_baseline'])[:500])
PY
