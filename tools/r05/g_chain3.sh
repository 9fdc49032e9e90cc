#!/bin/bash
# round 5, call g: strip width G = 3 of the chain kernel -- tests for both widths, step trace, RLFN bf16 batch 32 / DIV2K mode with each width forced and with the cost model's choice
O=$GRAFT_REPO_ROOT/gpurun_out/r05g; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_chain.py -q -x 2>&1 | tail -5 | tee $O/t.txt
for g in 2 3; do ESR_CHAIN_G=$g timeout 100 python tools/r05/chain_trace.py run 32 256 256 2>&1 | grep -v amdgpu.ids | head -7 | tee -a $O/trace.txt; done
for g in 2 3 auto; do
  if [ $g = auto ]; then unset ESR_CHAIN_G; else export ESR_CHAIN_G=$g; fi
  for mode in "" "--sizes div2k --streams 1"; do
  timeout 300 python bench.py --model team04_rlfn --compute bf16 $mode --no-cpu-baseline --no-other-configs 2> $O/err.txt | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('G=$g', '$mode', j['value'], j['ms_per_step'], [(k['kernel'][:28], k['avg_ms']) for k in j['roofline']['kernels'][:2]])" | tee -a $O/sum.txt
  done
done
