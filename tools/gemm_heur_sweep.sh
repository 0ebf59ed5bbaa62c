# A/B sweep of the tiled-GEMM launch heuristics (HULC_TILE_TARGET = workgroups wanted before split-K stops, HULC_TILE_MINKT = k-tiles
# kept per slice, HULC_TILE_SMALLK = reductions up to this length use 64x64 tiles): bash tools/gemm_heur_sweep.sh "768 4 256" ...
for cfg in "$@"; do
  set -- $cfg
  echo "== target=$1 minkt=$2 smallk=$3"
  HULC_TILE_TARGET=$1 HULC_TILE_MINKT=$2 HULC_TILE_SMALLK=$3 timeout 300 python bench.py --no-cpu-baseline --breakdown 2>&1 | grep -E "sum of kernel|ms_per_step" | sed -E 's/.*("ms_per_step": [0-9.]+).*/\1/'
done
