set -x
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
mkdir -p gpurun_out/r01i
timeout 600 python bench.py > gpurun_out/r01i/bench.json 2> gpurun_out/r01i/bench.err; tail -c 3000 gpurun_out/r01i/bench.json
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r01i/stats -- python $R/bench.py --no-cpu-baseline --steps 10 > $R/gpurun_out/r01i/stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r01i/pmc_fetch -- python $R/bench.py --no-cpu-baseline --no-kernel-events --steps 2 --warmup 1 > $R/gpurun_out/r01i/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r01i/pmc_write -- python $R/bench.py --no-cpu-baseline --no-kernel-events --steps 2 --warmup 1 > $R/gpurun_out/r01i/pmc_write.log 2>&1
cd $R
DB=$(find gpurun_out/r01i/stats -name "*.db" | head -1); python tools/rocpd_summary.py $DB > gpurun_out/r01i/kernel_stats.md; head -12 gpurun_out/r01i/kernel_stats.md
python tools/pmc_traffic.py gpurun_out/r01i/pmc_fetch gpurun_out/r01i/pmc_write gpurun_out/r01i/pmc_traffic.json r01i | tail -30
find gpurun_out/r01i -name "*.csv" -size +2M -delete; du -sh gpurun_out/r01i
