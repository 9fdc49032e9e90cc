#!/bin/bash
# round 4: conv48rp micro-steps -- kernel test, RLFN model tests, benches
O=$GRAFT_REPO_ROOT/gpurun_out/r04m; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_h16.py -q -x -k "conv48rp or post_chain or batch" 2>&1 | tail -3 > $O/t.txt
timeout 900 python -m pytest tests/test_gpu_esa_models.py tests/test_gpu_multi.py tests/test_gpu_big.py -q -x -k "rlfn" 2>&1 | tail -3 >> $O/t.txt
for rep in 1 2; do
timeout 300 python bench.py --model team04_rlfn --compute bf16 --no-cpu-baseline --no-other-configs > $O/b32_rlfn_$rep.json 2> $O/b32.err
timeout 300 python bench.py --model team04_rlfn --compute bf16 --sizes div2k --streams 1 --no-cpu-baseline --no-other-configs > $O/div2k_rlfn_$rep.json 2> $O/div2k.err
done
python - <<'PY' > $O/summary.txt
import json,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r04m/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); r=j["roofline"]
        print(os.path.basename(f), j["value"], j["ms_per_step"], [(k["kernel"],k["avg_ms"]) for k in r["kernels"][:4]])
    except Exception as e: print(f, "ERR", e)
PY
cat $O/t.txt $O/summary.txt
