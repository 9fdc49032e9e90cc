#!/bin/bash
# round 5, call e: new parity tests (crops, full-tensor checksums, graphs), host latency with the direct eager path, bench line with calibration
O=$GRAFT_REPO_ROOT/gpurun_out/r05e; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_crops.py tests/test_gpu_big.py tests/test_gpu_graph.py tests/test_shim_and_op.py -m gpu -q -x -s 2>&1 | grep -v "^$" | tail -30 | tee $O/t.txt
timeout 200 python tools/b1_latency.py 4 bf16 339x510 2>&1 | grep -v amdgpu.ids | tee $O/lat.txt
timeout 400 python bench.py --model team04_rlfn --compute bf16 --sizes div2k --streams 1 --no-cpu-baseline --no-other-configs > $O/div2k.json 2> $O/div2k.err
python - <<'PY'
import json,os
j=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r05e/div2k.json").read().strip().splitlines()[-1]); r=j["roofline"]
print(j["value"], {k: r[k] for k in ("kernel","bound","frac","peak","avg_launch_ms","avg_launch_ms_with_event_pair","event_pair_ms","frac_of_hbm_peak","frac_algorithmic")}, r.get("peak_source"))
for k in r["kernels"][:8]: print(k)
PY
