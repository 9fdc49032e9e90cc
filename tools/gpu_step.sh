#!/bin/bash
# one development step on the GPU: selected tests, then bench lines of the 16-bit configs   ->  gpurun_out/step/
#   bash tools/gpu_step.sh "<pytest selection>" [full]
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/step; mkdir -p $O; cd $R
timeout 900 python -m pytest $1 -m gpu -x -q 2>&1 | tail -8
for m in "team04_rlfn bf16" "rfdn_baseline bf16"; do set -- $m
  timeout 200 python bench.py --model $1 --compute $2 --no-cpu-baseline > $O/b_$1_$2.json 2>/dev/null; done
timeout 200 python bench.py --model team18_bsrn --compute f16 --tile 270x480 --no-cpu-baseline > $O/b_bsrn_f16_270x480.json 2>/dev/null
for m in "team04_rlfn bf16" "rfdn_baseline bf16"; do set -- $m
  timeout 200 python bench.py --model $1 --compute $2 --sizes div2k --streams 1 --no-cpu-baseline > $O/b_$1_$2_div2k_s1.json 2>/dev/null; done
python tools/show_bench.py $O/b_*.json
