#!/bin/bash
# round 5, call f: GELU as plain VALU (product) against packed fp32 (research build): BSRN fp16 batch 32x270x480 and DIV2K mode, bit-equality
O=$GRAFT_REPO_ROOT/gpurun_out/r05f; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_bsrn.py tests/test_gpu_h16.py -m gpu -q -x 2>&1 | tail -3 | tee $O/t.txt
for v in plain packed plain packed; do
  if [ $v = packed ]; then export ESR_HIP_LIB=$GRAFT_REPO_ROOT/tools/r05/libesr_gelupk.so; else unset ESR_HIP_LIB; fi
  timeout 300 python bench.py --model team18_bsrn --compute f16 --tile 270x480 --no-cpu-baseline --no-other-configs 2> $O/err.txt | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'bsrn f16 32x270x480', j['value'], j['ms_per_step'], [(k['kernel'][:28], k['avg_ms']) for k in j['roofline']['kernels'][:4]])" | tee -a $O/sum.txt
done
unset ESR_HIP_LIB
