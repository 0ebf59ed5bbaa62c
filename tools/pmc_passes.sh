# usage (GPU box): bash tools/pmc_passes.sh <tag> "<counter set 1>" "<counter set 2>" ... -- python3 script.py args   : one rocprofv3 --pmc run per set,
# per-kernel averages into gpurun_out/pmc_<tag>_<i>.txt
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=$1; shift
sets=()
while [ "$1" != "--" ]; do sets+=("$1"); shift; done
shift
i=0
for set in "${sets[@]}"; do
  rm -rf gpurun_out/pp; mkdir -p gpurun_out/pp
  rocprofv3 --pmc $set --kernel-trace -d gpurun_out/pp -o p -- "$@" > gpurun_out/pp.log 2>&1
  python3 tools/rocpd_stats.py --pmc $(find gpurun_out/pp -name "*.db" | head -1) > gpurun_out/pmc_${tag}_$i.txt
  i=$((i+1))
done
rm -rf gpurun_out/pp
