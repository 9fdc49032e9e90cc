#!/bin/bash
# round 5, call k: prologues without dependent round trips (esa_apply images, s2pool16 patch, conv_s16 / conv48r biases + border table by LDS-DMA, pack_input)
O=$GRAFT_REPO_ROOT/gpurun_out/r05k; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 | tee $O/t.txt
for mc in "imdn_baseline f32" "team04_rlfn bf16" "rfdn_baseline bf16" "team18_bsrn f16 --tile 270x480"; do
  for mode in "" "--sizes div2k --streams 1" "--sizes div2k"; do
  set -- $mc
  if [ -n "$mode" ]; then extra=""; else extra="$3 $4"; fi
  timeout 300 python bench.py --model $1 --compute $2 $extra $mode --no-cpu-baseline --no-other-configs 2> $O/err.txt | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$mode', j['value'], j['ms_per_step'], [(k['kernel'][:40], k['avg_ms']) for k in j['roofline']['kernels'][:8]])" | tee -a $O/sum.txt
  done
done
