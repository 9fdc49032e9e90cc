#!/bin/bash
# round 4: epilogue micro-steps in conv48r / conv64r -- kernel tests, 16-bit model tests, benches of the three 16-bit configs (+ RFDN / RLFN DIV2K mode)
O=$GRAFT_REPO_ROOT/gpurun_out/r04k; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_h16.py tests/test_gpu_multi.py tests/test_gpu_big.py tests/test_gpu_esa_models.py tests/test_gpu_bsrn.py -q -x 2>&1 | tail -8 > $O/t_models.txt
for m in "team04_rlfn bf16 256x256" "rfdn_baseline bf16 256x256" "team18_bsrn f16 270x480"; do set -- $m
  timeout 300 python bench.py --model $1 --compute $2 --tile $3 --no-cpu-baseline --no-other-configs > $O/b32_$1.json 2> $O/b32_$1.err
  timeout 300 python bench.py --model $1 --compute $2 --sizes div2k --streams 1 --no-cpu-baseline --no-other-configs > $O/div2k_$1.json 2> $O/div2k_$1.err
done
python - <<'PY' > $O/summary.txt
import json,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r04k/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); r=j["roofline"]
        print(os.path.basename(f), j["value"], j["ms_per_step"], [(k["kernel"],k["avg_ms"]) for k in r["kernels"][:9]])
    except Exception as e: print(f, "ERR", e)
PY
cat $O/t_models.txt $O/summary.txt
