#!/bin/bash
# RFDN bf16 with its nf-wide tensors at pitch 56 (tight, default) against pitch 64 (--no-tight-pitch): batch 32 and DIV2K-shaped single images, twice round-robin
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2; do for f in "" "--no-tight-pitch"; do
  for mode in "" "--sizes div2k --streams 1" "--sizes div2k"; do
    timeout 200 python bench.py --model rfdn_baseline --compute bf16 $mode $f --no-cpu-baseline --no-other-configs --no-kernel-events 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('tight' if '$f' == '' else 'pitch64', '[$mode]', d['value'], 'images/s', d['ms_per_step'], 'ms/step')"
  done; done; done 2>&1 | tee gpurun_out/tight_ab.txt
