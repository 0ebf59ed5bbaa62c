cd $GRAFT_REPO_ROOT
python3 -m pytest tests/ -q -m gpu 2>&1 | tail -4
