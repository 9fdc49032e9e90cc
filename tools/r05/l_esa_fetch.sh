#!/bin/bash
# round 5, call l: esa_apply with unconditional fetch loads (both groups of an iteration in flight together); occupancy targets 1 (none) / 3 / 4 waves per SIMD
O=$GRAFT_REPO_ROOT/gpurun_out/r05l; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_esa_models.py tests/test_gpu_bsrn.py tests/test_gpu_multi.py -q -x 2>&1 | tail -3 | tee $O/t.txt
for v in "" w3 w4; do
  if [ -n "$v" ]; then export ESR_HIP_LIB=$GRAFT_REPO_ROOT/tools/r05/libesr_$v.so; else unset ESR_HIP_LIB; fi
  for mc in "team04_rlfn bf16" "rfdn_baseline bf16" "team18_bsrn f16 --tile 270x480"; do
  for mode in "" "--sizes div2k --streams 1"; do
  set -- $mc
  if [ -n "$mode" ]; then extra=""; else extra="$3 $4"; fi
  timeout 300 python bench.py --model $1 --compute $2 $extra $mode --no-cpu-baseline --no-other-configs 2> $O/err.txt | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=$v', '$1', '$mode', j['value'], j['ms_per_step'], [(k['kernel'][:34], k['avg_ms']) for k in j['roofline']['kernels'] if 'esa_apply' in k['kernel']])" | tee -a $O/sum.txt
  done
  done
done
