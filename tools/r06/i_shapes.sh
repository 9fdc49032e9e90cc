#!/bin/bash
# round 6: is conv64m_kernel's memory cost tied to the power-of-two row pitch of 256-wide tiles (channel camping)?  per-pixel kernel times by tile shape
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r06q; mkdir -p $O; cd $R
for t in 256x256 256x272 256x240 240x256 272x272 128x512 512x128; do
  timeout 300 python bench.py --model rfdn_baseline --compute bf16 --tile $t --no-cpu-baseline --no-other-configs > $O/bench_$t.json 2>/dev/null
done
python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06q"
for f in sorted(os.listdir(O)):
    if f.startswith("bench_") and f.endswith(".json"):
        d = json.loads(open(os.path.join(O, f)).read().strip().splitlines()[-1])
        h, w = (int(v) for v in f[6:-5].split("x"))
        px = 32 * h * w
        print(f, round(d["value"] * h * w / 65536, 1), "img/s (256^2-equivalent)", [(k["kernel"][:28], round(k["avg_ms"] * 1e6 / px, 4)) for k in d["roofline"]["kernels"][:6]], "ns per pixel")
PY
