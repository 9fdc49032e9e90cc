#!/bin/bash
# round 5, call h: esr_s16.hip built with -fno-slp-vectorize (no compiler-made v_pk_*_f32 beside MFMAs) and with the scalar C++ GELU, against the product
O=$GRAFT_REPO_ROOT/gpurun_out/r05h; mkdir -p $O; cd $GRAFT_REPO_ROOT
for v in product noslp gelucxx product noslp gelucxx; do
  if [ $v = product ]; then unset ESR_HIP_LIB; else export ESR_HIP_LIB=$GRAFT_REPO_ROOT/tools/r05/libesr_$v.so; fi
  for cfg in "team18_bsrn f16 --tile 270x480" "rfdn_baseline bf16" "team04_rlfn bf16"; do set -- $cfg
  timeout 300 python bench.py --model $1 --compute $2 $3 $4 --no-cpu-baseline --no-other-configs 2> $O/err.txt | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '$1', j['value'], j['ms_per_step'], [(k['kernel'][:26], k['avg_ms']) for k in j['roofline']['kernels'][:3]])" | tee -a $O/sum.txt
  done
done
unset ESR_HIP_LIB
