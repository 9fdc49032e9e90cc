#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r05q; mkdir -p $O; cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
for v in "" c64abl "" c64abl; do
  if [ -n "$v" ]; then export ESR_HIP_LIB=$R/tools/r05/libesr_$v.so; else unset ESR_HIP_LIB; fi
  timeout 300 python bench.py --model rfdn_baseline --compute bf16 --no-cpu-baseline --no-other-configs 2> $O/err.txt | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=$v', j['value'], j['ms_per_step'], [(k['kernel'][:30], k['avg_ms']) for k in j['roofline']['kernels'][:6] if 'conv64' in k['kernel']])" | tee -a $O/sum.txt
done
