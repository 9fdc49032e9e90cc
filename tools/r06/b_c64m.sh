#!/bin/bash
# round 6: conv64m_kernel -- correctness (fp64 reference), then A/B against conv64r / conv64rq (ESR_C64M=0) on RFDN bf16
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r06c; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_c64m.py -q -x 2>&1 | tail -25 > $O/c64m_tests.txt
cat $O/c64m_tests.txt
for v in 1 0; do
  ESR_C64M=$v timeout 300 python bench.py --model rfdn_baseline --compute bf16 --no-cpu-baseline --no-other-configs > $O/bench_rfdn_b32_c64m$v.json 2>/dev/null
  ESR_C64M=$v timeout 300 python bench.py --model rfdn_baseline --compute bf16 --sizes div2k --streams 1 --no-cpu-baseline --no-other-configs > $O/bench_rfdn_div2k_c64m$v.json 2>/dev/null
done
python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06c"
for f in sorted(os.listdir(O)):
    if f.startswith("bench_") and f.endswith(".json"):
        try:
            d = json.loads(open(os.path.join(O, f)).read().strip().splitlines()[-1])
        except Exception as e:
            print(f, "ERR", e); continue
        print(f, d["value"], d["ms_per_step"])
        for k in d["roofline"]["kernels"][:8]:
            print("    ", k["kernel"][:60], k["share"], k["avg_ms"], k["gbs"])
PY
