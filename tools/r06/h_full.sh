#!/bin/bash
# round 6: the whole GPU suite + the default bench line after conv64m_kernel
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r06o; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/gputests.txt
cat $O/gputests.txt
timeout 300 python bench.py --model rfdn_baseline --compute bf16 --no-cpu-baseline --no-other-configs > $O/bench_rfdn_b32.json 2>/dev/null
timeout 600 python bench.py --no-cpu-baseline > $O/bench_default.json 2>/dev/null
python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06o"
d = json.loads(open(O + "/bench_rfdn_b32.json").read().strip().splitlines()[-1])
print("rfdn b32", d["value"], [(k["kernel"][:28], k["avg_ms"]) for k in d["roofline"]["kernels"] if "conv64" in k["kernel"]])
d = json.loads(open(O + "/bench_default.json").read().strip().splitlines()[-1])
print("imdn", d["value"], d["ms_per_step"])
for o in d["other_configs"]: print(o["workload"][:70], o["value"], o["ms_per_image"])
PY
