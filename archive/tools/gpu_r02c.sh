#!/bin/bash
O=gpurun_out/r02c; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep -n "passed\|failed\|rc=\|FAILED\|Error" $O/pytest.log | tail -40
grep -n "bsrn.*PSNR\|RFDNFINAL.*PSNR\|bsrn.*dPSNR" $O/pytest.log | head -20
timeout 300 python bench.py --model team18_bsrn --compute f16 --tile 270x480 --no-cpu-baseline > $O/bench_c4_270x480.json 2> $O/bench_c4.err
timeout 300 python bench.py --model team18_bsrn --compute f16 --no-cpu-baseline > $O/bench_bsrn_f16_b32.json 2>> $O/bench_c4.err
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
