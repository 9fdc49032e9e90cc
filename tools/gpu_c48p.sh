#!/bin/bash
# conv48rp_kernel: RLFN tests, overlapped-forward check, benches
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_h16.py tests/test_gpu_big.py tests/test_gpu_multi.py tests/test_gpu_esa_models.py -m gpu -x -q 2>&1 | tail -5
for c in bf16 f16; do python tools/dbg/out_hash.py team04_rlfn $c 2>/dev/null | tail -1; done
python tools/dbg/streams_race.py team04_rlfn bf16 200 2>&1 | grep -E "mismatching|serial"
python tools/dbg/streams_race.py team04_rlfn f16 100 2>&1 | grep -E "mismatching|serial"
for m in "team04_rlfn bf16" "team04_rlfn bf16 --sizes div2k --streams 1" "team04_rlfn bf16 --sizes div2k"; do
  set -- $m; python bench.py --model $1 --compute $2 $3 $4 $5 $6 --no-cpu-baseline --steps 30 > /tmp/b.json 2>/dev/null; python tools/show_bench.py /tmp/b.json | head -4 | cut -c1-170; done
python bench.py --model team04_rlfn --compute bf16 --tile 339x510 --batch 1 --b1-latency --no-cpu-baseline --no-kernel-events --steps 50 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=1 339x510 latency ms', j.get('b1_latency_ms'))"
