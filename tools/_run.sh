cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu -s 2>&1 | grep -v "^$" | grep "gradient error\|passed\|failed\|Error\|assert" | head -30
python3 -m pytest tests/test_r3m_gpu.py -x -q -m gpu -s -k real_world_training_step 2>&1 | grep "gradients, median\|passed\|failed\|Error\|assert" | cut -c1-400 | head -20
