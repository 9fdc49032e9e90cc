#!/bin/bash
# round 4: conv64r_kernel first light -- bit-identity tests, RFDN / IMDN bf16 model tests, benches
O=$GRAFT_REPO_ROOT/gpurun_out/r04j; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_h16.py -q -x -k "conv64r" 2>&1 | tail -15 > $O/t_c64.txt
timeout 1500 python -m pytest tests/test_gpu_h16.py tests/test_gpu_multi.py tests/test_gpu_big.py tests/test_gpu_esa_models.py tests/test_gpu_imdn.py -q -x 2>&1 | tail -15 > $O/t_models.txt
for m in rfdn_baseline imdn_baseline; do
  timeout 300 python bench.py --model $m --compute bf16 --no-cpu-baseline --no-other-configs > $O/b32_${m}.json 2> $O/b32_${m}.err
  timeout 300 python bench.py --model $m --compute bf16 --sizes div2k --streams 1 --no-cpu-baseline --no-other-configs > $O/div2k_${m}.json 2> $O/div2k_${m}.err
done
python - <<'PY' > $O/summary.txt
import json,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r04j/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); r=j["roofline"]
        print(os.path.basename(f), j["value"], j["ms_per_step"], [(k["kernel"],k["avg_ms"]) for k in r["kernels"][:10]])
    except Exception as e: print(f, "ERR", e)
PY
cat $O/t_c64.txt $O/t_models.txt $O/summary.txt
