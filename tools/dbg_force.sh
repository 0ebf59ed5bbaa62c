cd $GRAFT_REPO_ROOT
A="--force-dist --steps 3 --warmup 1 --batch 4 --seq-len 16 --no-cpu-baseline --no-secondary"
for v in "HULC_BENCH_BACKEND=gloo" "HULC_NO_SPLIT_GRAPH=1" "HULC_CAPTURE_MODE=global" "HULC_RNN_DBG=0"; do
  echo "== variant: $v"; env $v python bench.py $A 2>&1 | grep -o '"final_loss": [^}]*\|NaN.*\|Error.*' | head -3
done
