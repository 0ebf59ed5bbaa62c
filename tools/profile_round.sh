cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
# per-kernel durations and counters are taken on ONE stream (the default two-stream encoders overlap launches: their traced durations would
# include each other's share of the chip); the default bench line at the end runs the shipped two-stream configuration
export HULC_ENC_STREAMS=0
export HULC_FORK=0            # (round 6: the forked branches overlap launches too; the default bench line below runs the shipped forked graph)
mkdir -p gpurun_out/pk gpurun_out/pf gpurun_out/pw gpurun_out/pm
rocprofv3 --kernel-trace --stats -d gpurun_out/pk -o k -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-graph > gpurun_out/pk.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pf -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-graph > gpurun_out/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pw -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-graph > gpurun_out/pw.log 2>&1
# MFMA utilisation pass (north_star: "rocprof HBM GB/s and MFMA utilisation against chip peak"): SQ + GRBM counters only, own run
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/pm -o m -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-graph > gpurun_out/pm.log 2>&1
unset HULC_ENC_STREAMS HULC_FORK
python3 bench.py > gpurun_out/bench_default.log 2>&1
for d in pk pf pw; do f=$(find gpurun_out/$d -name "*.db" | head -1); echo $d $f; done
python3 tools/rocpd_stats.py $(find gpurun_out/pk -name "*.db" | head -1) > gpurun_out/k_stats.txt
python3 tools/rocpd_stats.py --pmc $(find gpurun_out/pf -name "*.db" | head -1) > gpurun_out/f_pmc.txt
python3 tools/rocpd_stats.py --pmc $(find gpurun_out/pw -name "*.db" | head -1) > gpurun_out/w_pmc.txt
python3 tools/rocpd_stats.py --mfma $(find gpurun_out/pm -name "*.db" | head -1) > gpurun_out/m_mfma.txt
rm -rf gpurun_out/pk gpurun_out/pf gpurun_out/pw gpurun_out/pm
tail -1 gpurun_out/bench_default.log | cut -c1-400
