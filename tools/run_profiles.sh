set -x
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
mkdir -p gpurun_out/r01h
timeout 600 python bench.py > gpurun_out/r01h/bench.json 2> gpurun_out/r01h/bench.err; tail -c 3000 gpurun_out/r01h/bench.json
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r01h/stats -- python $R/bench.py --no-cpu-baseline --steps 10 > $R/gpurun_out/r01h/stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r01h/pmc_fetch -- python $R/bench.py --no-cpu-baseline --no-kernel-events --steps 2 --warmup 1 > $R/gpurun_out/r01h/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r01h/pmc_write -- python $R/bench.py --no-cpu-baseline --no-kernel-events --steps 2 --warmup 1 > $R/gpurun_out/r01h/pmc_write.log 2>&1
cd $R
DB=$(find gpurun_out/r01h/stats -name "*.db" | head -1); python tools/rocpd_summary.py $DB > gpurun_out/r01h/kernel_stats.md; head -12 gpurun_out/r01h/kernel_stats.md
python tools/pmc_traffic.py gpurun_out/r01h/pmc_fetch gpurun_out/r01h/pmc_write gpurun_out/r01h/pmc_traffic.json r01h | tail -30
find gpurun_out/r01h -name "*.csv" -size +2M -delete; du -sh gpurun_out/r01h
