#!/bin/bash
# images/s of one forward per step at batch 2 .. 32 (tensors of a small batch stay in the 256 MB Infinity Cache between launches; a large batch amortises
# prologues and tile quantisation): is a batch-32 step better served as cache-sized slices?   -> gpurun_out/bsweep.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for mc in "imdn_baseline f32 256x256" "rfdn_baseline bf16 256x256" "team04_rlfn bf16 256x256" "team18_bsrn f16 270x480"; do set -- $mc
  for b in 2 4 8 16 32; do
    timeout 120 python bench.py --model $1 --compute $2 --tile $3 --batch $b --no-cpu-baseline --no-other-configs --no-kernel-events --steps 30 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1 $2 batch $b', d['value'], 'images/s', d['ms_per_step'], 'ms/step')"
  done
done 2>&1 | tee gpurun_out/bsweep.txt
