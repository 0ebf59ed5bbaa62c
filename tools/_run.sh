cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_parity_gpu.py -q -m gpu -s -k "arrangements" 2>&1 | grep "gradient error\|passed\|failed\|Error\|assert" | head -30
