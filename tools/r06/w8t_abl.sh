#!/bin/bash
# every tools/abl/libesr_w8t_*.so: per-kernel event time of the tail inside the IMDN headline step (one gpurun call = one box)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/w8t
for so in $(ls tools/abl/libesr_w8t_*.so | sort -V); do
  ESR_HIP_LIB=$PWD/$so timeout 200 python bench.py --no-cpu-baseline --no-other-configs --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = [x for x in d['roofline']['kernels'] if 'tail' in x['kernel'] or 'wino8_f32' in x['kernel']]
print('$(basename $so)', d['value'], [(x['kernel'][:24], x['avg_ms']) for x in k])"
done 2>&1 | tee gpurun_out/w8t/abl.txt
