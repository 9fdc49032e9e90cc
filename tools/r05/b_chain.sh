#!/bin/bash
# round 5, call b: the 8-wave chain kernel -- tests, step trace, RLFN bf16 A/B (B = 1 on one stream, batch 32)
O=$GRAFT_REPO_ROOT/gpurun_out/r05b; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 240 python -m pytest tests/test_gpu_chain.py -q -x 2>&1 | tail -15 > $O/t.txt
cat $O/t.txt
timeout 100 python tools/r05/chain_trace.py run 32 256 256 2>&1 | grep -v amdgpu.ids | head -14 | tee $O/trace.txt
timeout 100 python tools/r05/chain_trace.py run 1 339 510 2>&1 | grep -v amdgpu.ids | head -3 | tee -a $O/trace.txt
if grep -q "passed" $O/t.txt && ! grep -q "failed" $O/t.txt; then
for fc in 1 0; do
if [ $fc = 0 ]; then FC=--no-fuse-chain; else FC=; fi
timeout 300 python bench.py $FC --model team04_rlfn --compute bf16 --sizes div2k --streams 1 --no-cpu-baseline --no-other-configs > $O/div2k_fc$fc.json 2> $O/div2k_fc$fc.err
timeout 300 python bench.py $FC --model team04_rlfn --compute bf16 --no-cpu-baseline --no-other-configs > $O/b32_fc$fc.json 2> $O/b32_fc$fc.err
done
python - <<'PY' | tee $O/summary.txt
import json,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r05b/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); r=j["roofline"]
        print(os.path.basename(f), j["value"], j["ms_per_step"], [(k["kernel"],k["avg_ms"]) for k in r["kernels"][:6]])
    except Exception as e: print(f, "ERR", e)
PY
fi
