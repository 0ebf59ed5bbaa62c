cd $GRAFT_REPO_ROOT
python3 bench.py --no-cpu-baseline --no-secondary > gpurun_out/bench_a.log 2>&1
tail -1 gpurun_out/bench_a.log | cut -c1-260
python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu 2>&1 | tail -3
