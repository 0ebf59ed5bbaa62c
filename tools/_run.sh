cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_parity_gpu.py tests/test_trainer_gpu.py tests/test_r3m_gpu.py tests/test_uint8_frames_gpu.py -q -m gpu 2>&1 | tail -3
for i in 1 2 3; do
HULC_BAND_GLDS=0 python3 bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | cut -c1-230 | sed 's/.*"ms_per_step"/regs ms_per_step/'
python3 bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | cut -c1-230 | sed 's/.*"ms_per_step"/glds ms_per_step/'
done
