#!/bin/bash
# round 5, call s: esa_apply's post-chain GELU as packed interleaved arithmetic (28 instead of 52 VALU per fragment)
O=$GRAFT_REPO_ROOT/gpurun_out/r05s; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bsrn.py tests/test_gpu_esa_models.py tests/test_gpu_multi.py tests/test_gpu_erratum.py -q -x 2>&1 | tail -2 | tee $O/t.txt
for i in 1 2; do
for mode in "--tile 270x480" "--sizes div2k --streams 1" "--sizes div2k"; do
  timeout 300 python bench.py --model team18_bsrn --compute f16 $mode --no-cpu-baseline --no-other-configs 2> $O/err.txt | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode', j['value'], j['ms_per_step'], [(k['kernel'][:36], k['avg_ms']) for k in j['roofline']['kernels'] if 'esa_apply' in k['kernel']])" | tee -a $O/sum.txt
done; done
