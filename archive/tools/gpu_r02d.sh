#!/bin/bash
O=gpurun_out/r02d; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_h16.py -m gpu -q -x -s > $O/pytest_s16.log 2>&1; echo "rc=$?" >> $O/pytest_s16.log
tail -15 $O/pytest_s16.log
timeout 900 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep -n "passed\|failed\|rc=\|FAILED\|Error" $O/pytest.log | tail -30
for cfg in "rfdn_baseline bf16" "team04_rlfn bf16" "imdn_baseline bf16" "team18_bsrn f16"; do set -- $cfg
  timeout 300 python bench.py --model $1 --compute $2 --no-cpu-baseline > $O/bench_$1_$2_b32.json 2> $O/bench_$1_$2.err
done
for f in $O/bench_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=j["roofline"]; print(j["value"], j["unit"], j["ms_per_step"], r["bound"], r["kernel"], r["frac"], r["avg_launch_ms"])
    for k in r["kernels"][:8]: print("   ", k)
except Exception as e: print("ERR", e)
PY
done
tail -n 3 $O/*.err
