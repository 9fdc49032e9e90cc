#!/bin/bash
# conv64m / rfdb_tail: out-of-pixel DMA parts masked (library) against requested (tools/abl/libesr_nomask.so): RFDN bf16 per-op + bench, round-robin
# (libesr_nomask.so = ntire2022_esr_amd/libesr_hip.so built from commit c1eec5e, the one before the mask; A/B libraries are not tracked)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for lib in "" "$PWD/tools/abl/libesr_nomask.so"; do
  ESR_HIP_LIB=$lib timeout 200 python bench.py --model rfdn_baseline --compute bf16 --no-cpu-baseline --no-other-configs --steps 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('masked' if '$lib' == '' else 'nomask', d['value'], [(k['kernel'][:44], k['avg_ms']) for k in d['roofline']['kernels'][:6]])"
done; done 2>&1 | tee gpurun_out/mask_ab.txt
