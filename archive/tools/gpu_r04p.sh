#!/bin/bash
# round 4: the hi + lo head with block 1's distillation 1x1 in its epilogue (RFDN, BSRN bf16) -- tests + benches
O=$GRAFT_REPO_ROOT/gpurun_out/r04p; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_h16.py tests/test_gpu_esa_models.py tests/test_gpu_multi.py tests/test_gpu_big.py tests/test_gpu_bsrn.py -q -x 2>&1 | tail -4 > $O/t.txt
python tools/per_op.py 0 bf16 2>&1 | grep -E "kernels per forward|head|B1.c1_d|LR_conv|upsampler" >> $O/t.txt
for rep in 1 2; do
timeout 300 python bench.py --model rfdn_baseline --compute bf16 --no-cpu-baseline --no-other-configs > $O/b32_rfdn_$rep.json 2> $O/b32.err
timeout 300 python bench.py --model rfdn_baseline --compute bf16 --sizes div2k --streams 1 --no-cpu-baseline --no-other-configs > $O/div2k_rfdn_$rep.json 2> $O/div2k.err
done
python - <<'PY' > $O/summary.txt
import json,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r04p/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); r=j["roofline"]
        print(os.path.basename(f), j["value"], j["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
cat $O/t.txt $O/summary.txt
