#!/bin/bash
# round 5, call i: u's high | low split moved from layer 3's wave to waves 0 | 1 (chain kernel), XCD-aware tile order of the low-resolution ESA kernels
O=$GRAFT_REPO_ROOT/gpurun_out/r05i; mkdir -p $O; cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_esa_models.py -q -x 2>&1 | tail -5 | tee $O/t.txt
for g in 2 3; do ESR_CHAIN_G=$g timeout 100 python tools/r05/chain_trace.py run 32 256 256 2>&1 | grep -v amdgpu.ids | head -7 | tee -a $O/trace.txt; done
for mode in "" "--sizes div2k --streams 1"; do
  timeout 300 python bench.py --model team04_rlfn --compute bf16 $mode --no-cpu-baseline --no-other-configs 2> $O/err.txt | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode', j['value'], j['ms_per_step'], [(k['kernel'][:28], k['avg_ms']) for k in j['roofline']['kernels'][:6]])" | tee -a $O/sum.txt
done
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python $R/bench.py --model team04_rlfn --compute bf16 --no-cpu-baseline --no-kernel-events --steps 2 --warmup 1 > $O/pmc_$c.log 2>&1
done
cd $R
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_traffic.json r05i --model team04_rlfn --compute bf16 | tail -60 > $O/pmc.txt
find $O -name "*.csv" -size +1M -delete
grep -A 8 s2pool $O/pmc.txt
