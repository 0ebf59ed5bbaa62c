cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_affordance_gpu.py tests/test_r3m_gpu.py -x -q -m gpu 2>&1 | tail -12
