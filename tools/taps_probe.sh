cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for d in 0 1 2; do
  mkdir -p gpurun_out/pp$d
  HULC_TAPS_DBG=$d rocprofv3 --kernel-trace --stats -d gpurun_out/pp$d -o k -- python3 bench.py --affordance --batch 32 --steps 3 --warmup 1 --no-cpu-baseline --no-graph > /dev/null 2>&1
  python3 tools/rocpd_stats.py $(find gpurun_out/pp$d -name "*.db" | head -1) | grep "wgrad_taps_kernel" | cut -c1-100
  rm -rf gpurun_out/pp$d
done
