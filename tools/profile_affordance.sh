cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/pa
rocprofv3 --kernel-trace --stats -d gpurun_out/pa -o k -- python3 bench.py --affordance --batch 32 --steps 5 --warmup 2 --no-cpu-baseline --no-graph > gpurun_out/pa.log 2>&1
python3 tools/rocpd_stats.py $(find gpurun_out/pa -name "*.db" | head -1) > gpurun_out/aff_stats.txt
rm -rf gpurun_out/pa
head -30 gpurun_out/aff_stats.txt | cut -c1-150
