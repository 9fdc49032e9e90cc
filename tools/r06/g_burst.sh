#!/bin/bash
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r06x; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_c64m.py tests/test_gpu_h16.py -q -x 2>&1 | tail -3
timeout 300 python bench.py --model rfdn_baseline --compute bf16 --no-cpu-baseline --no-other-configs > $O/bench_rfdn_b32.json 2>/dev/null
timeout 300 python bench.py --model rfdn_baseline --compute bf16 --sizes div2k --streams 1 --no-cpu-baseline --no-other-configs > $O/bench_rfdn_div2k.json 2>/dev/null
timeout 300 python bench.py --model rfdn_baseline --compute bf16 --sizes div2k --no-cpu-baseline --no-other-configs > $O/bench_rfdn_div2k_8s.json 2>/dev/null
python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06x"
for f in sorted(os.listdir(O)):
    if f.startswith("bench_") and f.endswith(".json"):
        d = json.loads(open(os.path.join(O, f)).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], [(k["kernel"][:28], k["avg_ms"]) for k in d["roofline"]["kernels"] if "conv64" in k["kernel"]])
PY
