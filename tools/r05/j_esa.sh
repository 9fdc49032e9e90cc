#!/bin/bash
# round 5, call j: esa_apply prologue with every load in flight at once; grid cap A/B; XCD-aware low-resolution ESA tiles
O=$GRAFT_REPO_ROOT/gpurun_out/r05j; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_esa_models.py tests/test_gpu_bsrn.py tests/test_gpu_h16.py tests/test_gpu_chain.py -q -x 2>&1 | tail -5 | tee $O/t.txt
for cap in 4096 2048 1024 768 512; do
  export ESR_ESA_GRID_CAP=$cap
  for mc in "team04_rlfn bf16" "rfdn_baseline bf16" "team18_bsrn f16 --tile 270x480"; do
  for mode in "" "--sizes div2k --streams 1"; do
  set -- $mc
  if [ "$1" = team18_bsrn ] && [ -n "$mode" ]; then m2="--sizes div2k --streams 1"; extra=""; else m2="$mode"; extra="$3 $4"; fi
  timeout 300 python bench.py --model $1 --compute $2 $extra $m2 --no-cpu-baseline --no-other-configs 2> $O/err.txt | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cap=$cap', '$1', '$m2', j['value'], j['ms_per_step'], [(k['kernel'][:34], k['avg_ms']) for k in j['roofline']['kernels'] if 'esa_apply' in k['kernel']])" | tee -a $O/sum.txt
  done
  done
done
