#!/bin/bash
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r04g; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q -rP -x 2>&1 | tail -400 > $O/gputests.txt
grep -E "passed|failed|error" $O/gputests.txt | tail -3
( time python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r04g/bench_default.json").read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], j["roofline"]["kernel"], j["roofline"]["frac"], j["roofline"].get("frac_algorithmic"), j["roofline"]["traffic"], (j["roofline"]["traffic_source"] or "")[:90])
for o in j.get("other_configs", []): print("  ", o["workload"][:70], o["dtype"], o["value"], o["ms_per_step"], o["roofline"]["kernel"], o["roofline"]["bound"], o["roofline"]["frac"])
print("cpu", j.get("cpu_baseline", {}).get("value"))
PY
