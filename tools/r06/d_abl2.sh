#!/bin/bash
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r06i; mkdir -p $O; cd $R
for a in sk2 sk4 sk8; do
  ESR_HIP_LIB=$R/tools/r06/libesr_$a.so timeout 300 python bench.py --model rfdn_baseline --compute bf16 --no-cpu-baseline --no-other-configs > $O/bench_rfdn_b32_abl$a.json 2>/dev/null
done
python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06i"
for f in sorted(os.listdir(O)):
    if f.startswith("bench_") and f.endswith(".json"):
        try:
            d = json.loads(open(os.path.join(O, f)).read().strip().splitlines()[-1])
        except Exception as e:
            print(f, "ERR", e); continue
        print(f, d["value"], d["ms_per_step"], [(k["kernel"][:28], k["avg_ms"]) for k in d["roofline"]["kernels"] if "conv64m" in k["kernel"]])
PY
