#!/bin/bash
# round 6: conv64m_kernel after the fast DMA pieces -- tests, per-kernel times of the product build and of the ablation builds
# (C64M_ABL bit 0 no stores, bit 1 no DMA behind the first tile, bit 2 no epilogue)
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r06d; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_c64m.py -q -x 2>&1 | tail -5 > $O/c64m_tests.txt
cat $O/c64m_tests.txt
run() {  # tag, env
  env $2 timeout 300 python bench.py --model rfdn_baseline --compute bf16 --no-cpu-baseline --no-other-configs > $O/bench_rfdn_b32_$1.json 2>/dev/null
}
run prod "A=1"
for a in 1 2 4 7; do run abl$a "ESR_HIP_LIB=$R/tools/r06/libesr_abl$a.so"; done
env timeout 300 python bench.py --model rfdn_baseline --compute bf16 --sizes div2k --streams 1 --no-cpu-baseline --no-other-configs > $O/bench_rfdn_div2k_prod.json 2>/dev/null
python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06d"
for f in sorted(os.listdir(O)):
    if f.startswith("bench_") and f.endswith(".json"):
        try:
            d = json.loads(open(os.path.join(O, f)).read().strip().splitlines()[-1])
        except Exception as e:
            print(f, "ERR", e); continue
        print(f, d["value"], d["ms_per_step"], [(k["kernel"][:28], k["avg_ms"]) for k in d["roofline"]["kernels"] if "conv64m" in k["kernel"]])
PY
