cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_conv_gpu.py tests/test_parity_gpu.py tests/test_uint8_frames_gpu.py -x -q -m gpu 2>&1 | tail -4
for i in 1 2; do
HULC_CONV1_PER_INPUT=1 python3 bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | cut -c1-230 | sed 's/.*"ms_per_step"/per-input ms_per_step/'
python3 bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | cut -c1-230 | sed 's/.*"ms_per_step"/paired    ms_per_step/'
done
