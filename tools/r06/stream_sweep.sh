#!/bin/bash
# a batch-32 step as 1 / 2 / 4 sub-batches on as many HIP streams: do one stream's launch ramps fill the other's?   -> gpurun_out/ssweep.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for mc in "imdn_baseline f32 256x256" "rfdn_baseline bf16 256x256" "team04_rlfn bf16 256x256" "team18_bsrn f16 270x480"; do set -- $mc
  for s in 1 2 4; do
    timeout 120 python bench.py --model $1 --compute $2 --tile $3 --streams $s --no-cpu-baseline --no-other-configs --no-kernel-events --steps 30 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1 $2 streams $s', d['value'], 'images/s', d['ms_per_step'], 'ms/step')"
  done
done 2>&1 | tee gpurun_out/ssweep.txt
