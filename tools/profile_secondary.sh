# kernel stats of the two secondary configurations (BASELINE configs[3] and configs[4]) on the GPU box: eager runs under rocprofv3 + the default
# (hipGraph) bench lines.   usage: bash tools/profile_secondary.sh ; then HULC_ROUND=NN python tools/collect_secondary_profiles.py
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/pa gpurun_out/prw
rocprofv3 --kernel-trace --stats -d gpurun_out/pa -o k -- python3 bench.py --affordance --batch 32 --steps 5 --warmup 2 --no-cpu-baseline --no-graph > gpurun_out/pa.log 2>&1
python3 tools/rocpd_stats.py $(find gpurun_out/pa -name "*.db" | head -1) > gpurun_out/aff_stats.txt
rocprofv3 --kernel-trace --stats -d gpurun_out/prw -o k -- python3 bench.py --real-world --steps 5 --warmup 2 --no-cpu-baseline --no-graph > gpurun_out/prw.log 2>&1
python3 tools/rocpd_stats.py $(find gpurun_out/prw -name "*.db" | head -1) > gpurun_out/rw_stats.txt
rm -rf gpurun_out/pa gpurun_out/prw
python3 bench.py --affordance --no-cpu-baseline > gpurun_out/bench_aff_default.log 2>&1
python3 bench.py --real-world --no-cpu-baseline > gpurun_out/bench_rw_default.log 2>&1
head -8 gpurun_out/rw_stats.txt | cut -c1-150
