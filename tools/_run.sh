cd $GRAFT_REPO_ROOT
HULC_BREAKDOWN_ROWS=70 python3 bench.py --no-cpu-baseline --no-secondary --breakdown --steps 20 2>&1 | grep "ms/step" | cut -c1-200
