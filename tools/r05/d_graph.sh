#!/bin/bash
# round 5, call d: HIP-graph forwards -- bit-equality tests, host time per forward (idle queue), DIV2K-mode lines at 1 / 8 / 10 streams
O=$GRAFT_REPO_ROOT/gpurun_out/r05d; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_graph.py -q -x 2>&1 | tail -12 | tee $O/t.txt
for a in "4 bf16 339x510" "0 bf16 339x510" "18 f16 270x480" "-1 f32 256x256"; do timeout 200 python tools/b1_latency.py $a 2>&1 | grep -v amdgpu.ids | tee -a $O/lat.txt; done
for st in 1 8 10; do
timeout 300 python bench.py --model team04_rlfn --compute bf16 --sizes div2k --streams $st --no-cpu-baseline --no-other-configs 2> $O/err_$st.txt | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rlfn div2k streams', $st, j['value'], j['ms_per_step'])" | tee -a $O/sum.txt
done
