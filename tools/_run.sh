cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_trainer_gpu.py -q -m gpu -k "external_optimizer" 2>&1 | tail -12
