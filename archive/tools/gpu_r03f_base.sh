#!/bin/bash
# baseline of the restored build: the whole GPU suite + one bench line per config   ->  gpurun_out/r03f_base/
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03f_base; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/gputests.txt; cat $O/gputests.txt
python bench.py --no-cpu-baseline > $O/b_imdn_f32.json 2>/dev/null
for m in "team04_rlfn bf16" "rfdn_baseline bf16" "imdn_baseline bf16"; do set -- $m
  timeout 200 python bench.py --model $1 --compute $2 --no-cpu-baseline > $O/b_$1_$2.json 2>/dev/null; done
timeout 200 python bench.py --model team18_bsrn --compute f16 --tile 270x480 --no-cpu-baseline > $O/b_bsrn_f16_270x480.json 2>/dev/null
for m in "team04_rlfn bf16" "rfdn_baseline bf16" "team18_bsrn f16"; do set -- $m
  timeout 200 python bench.py --model $1 --compute $2 --sizes div2k --streams 1 --no-cpu-baseline > $O/b_$1_$2_div2k_s1.json 2>/dev/null; done
python tools/show_bench.py $O/b_*.json
