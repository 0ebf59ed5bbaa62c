cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_mlp_chain_gpu.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2 3; do
HULC_LIB=hulc2_amd/libhulc2_amd_base.so python3 bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | cut -c1-230 | sed 's/.*"ms_per_step"/base ms_per_step/'
python3 bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | cut -c1-230 | sed 's/.*"ms_per_step"/warm ms_per_step/'
done
