#!/bin/bash
# first GPU pass of round 2: full -m gpu suite (incl. the stated-size parity cases) + bench lines of every BASELINE config
set -x
O=gpurun_out/r02a; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x -s > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_c1.json 2> $O/bench_c1.err
timeout 300 python bench.py --model rfdn_baseline --compute bf16 --sizes div2k --no-cpu-baseline > $O/bench_c2_div2k.json 2> $O/bench_c2.err
timeout 300 python bench.py --model team04_rlfn --compute bf16 --sizes div2k --no-cpu-baseline > $O/bench_c3_div2k.json 2> $O/bench_c3.err
timeout 300 python bench.py --model team18_bsrn --compute f16 --tile 270x480 --no-cpu-baseline > $O/bench_c4_270x480.json 2> $O/bench_c4.err
timeout 300 python bench.py --model rfdn_baseline --compute bf16 --no-cpu-baseline > $O/bench_rfdn_bf16_b32.json 2>> $O/bench_c2.err
timeout 300 python bench.py --model team04_rlfn --compute bf16 --no-cpu-baseline > $O/bench_rlfn_bf16_b32.json 2>> $O/bench_c3.err
timeout 300 python bench.py --model team18_bsrn --compute f16 --no-cpu-baseline > $O/bench_bsrn_f16_b32.json 2>> $O/bench_c4.err
for f in $O/bench_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=j["roofline"]; print(j["value"], j["unit"], j["ms_per_step"], r["bound"], r["kernel"], r["frac"], r["avg_launch_ms"])
    for k in r["kernels"][:6]: print("   ", k)
except Exception as e: print("ERR", e)
PY
done
