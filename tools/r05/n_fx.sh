#!/bin/bash
# round 5, call n: conv48r / conv48rq with their switches compiled in (FX) + the GELU's two Horner chains interleaved (no s_nop between dependent v_pk_fma_f32)
O=$GRAFT_REPO_ROOT/gpurun_out/r05n; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee $O/t.txt
for mc in "team18_bsrn f16 --tile 270x480" "team18_bsrn bf16 --tile 270x480" "rfdn_baseline bf16" "team04_rlfn bf16"; do
  for mode in "" "--sizes div2k --streams 1"; do
  set -- $mc
  if [ -n "$mode" ]; then extra=""; else extra="$3 $4"; fi
  timeout 300 python bench.py --model $1 --compute $2 $extra $mode --no-cpu-baseline --no-other-configs 2> $O/err.txt | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', '$mode', j['value'], j['ms_per_step'], [(k['kernel'][:44], k['avg_ms']) for k in j['roofline']['kernels'][:7]])" | tee -a $O/sum.txt
  done
done
