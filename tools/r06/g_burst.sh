#!/bin/bash
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r06n; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_c64m.py -q -x 2>&1 | tail -3
ESR_HIP_LIB=$R/tools/r06/libesr_tr0.so timeout 200 python tools/r06/trace_c64m.py 2>&1 | grep -v amdgpu.ids | tee $O/trace_phase.txt
timeout 300 python bench.py --model rfdn_baseline --compute bf16 --no-cpu-baseline --no-other-configs > $O/bench_rfdn_b32_prod.json 2>/dev/null
for v in ph2 ph3 ph4sk; do ESR_HIP_LIB=$R/tools/r06/libesr_$v.so timeout 300 python bench.py --model rfdn_baseline --compute bf16 --no-cpu-baseline --no-other-configs > $O/bench_rfdn_b32_$v.json 2>/dev/null; done
timeout 300 python bench.py --model rfdn_baseline --compute bf16 --sizes div2k --streams 1 --no-cpu-baseline --no-other-configs > $O/bench_rfdn_div2k.json 2>/dev/null
python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06n"
for f in sorted(os.listdir(O)):
    if f.startswith("bench_") and f.endswith(".json"):
        d = json.loads(open(os.path.join(O, f)).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], [(k["kernel"][:28], k["avg_ms"]) for k in d["roofline"]["kernels"] if "conv64" in k["kernel"]])
PY
