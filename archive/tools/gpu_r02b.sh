#!/bin/bash
# s16 bring-up: unit tests first (short timeout: a hung kernel must not hold the box), then the whole suite, then benches
O=gpurun_out/r02b; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_h16.py -m gpu -q -x -s > $O/pytest_s16.log 2>&1; echo "rc=$?" >> $O/pytest_s16.log
tail -30 $O/pytest_s16.log
timeout 900 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep -n "passed\|failed\|rc=\|FAILED\|Error" $O/pytest.log | tail -40
for cfg in "rfdn_baseline bf16" "team04_rlfn bf16" "team04_rlfn f16" "imdn_baseline bf16"; do set -- $cfg
  timeout 300 python bench.py --model $1 --compute $2 --no-cpu-baseline > $O/bench_$1_$2_b32.json 2> $O/bench_$1_$2.err
done
timeout 300 python bench.py --model rfdn_baseline --compute bf16 --sizes div2k --no-cpu-baseline > $O/bench_c2_div2k.json 2>> $O/bench_c2.err
timeout 300 python bench.py --model team04_rlfn --compute bf16 --sizes div2k --no-cpu-baseline > $O/bench_c3_div2k.json 2>> $O/bench_c3.err
for f in $O/bench_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=j["roofline"]; print(j["value"], j["unit"], j["ms_per_step"], r["bound"], r["kernel"], r["frac"], r["avg_launch_ms"])
    for k in r["kernels"][:7]: print("   ", k)
except Exception as e: print("ERR", e)
PY
done
tail -3 $O/*.err
