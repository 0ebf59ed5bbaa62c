cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prw
rocprofv3 --kernel-trace --stats -d gpurun_out/prw -o k -- python3 bench.py --real-world --steps 5 --warmup 2 --no-cpu-baseline --no-graph > gpurun_out/prw.log 2>&1
python3 tools/rocpd_stats.py $(find gpurun_out/prw -name "*.db" | head -1) > gpurun_out/rw_k_stats.txt
python3 bench.py --real-world --no-cpu-baseline > gpurun_out/bench_rw.log 2>&1
rm -rf gpurun_out/prw
tail -1 gpurun_out/bench_rw.log | cut -c1-300
head -30 gpurun_out/rw_k_stats.txt
