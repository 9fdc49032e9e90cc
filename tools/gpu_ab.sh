#!/bin/bash
# A/B in ONE gpurun call: the committed baseline worktree (tools/abl/base, `git worktree add tools/abl/base <rev>` + build) against
# the working tree, alternating:   bash tools/gpu_ab.sh ["model compute [bench options]" ...]
R=$GRAFT_REPO_ROOT
if [ $# -eq 0 ]; then set -- "rfdn_baseline bf16" "team04_rlfn bf16" "team18_bsrn f16" "imdn_baseline bf16"; fi
for rep in 1 2; do
for m in "$@"; do read -r model compute extra <<< "$m"
  for side in base new; do
    if [ $side = base ]; then cd $R/tools/abl/base; else cd $R; fi
    timeout 300 python bench.py --model $model --compute $compute --no-cpu-baseline --steps 30 $extra > /tmp/b_$side.json 2>/tmp/b.err || tail -2 /tmp/b.err
    python $R/tools/show_bench.py /tmp/b_$side.json | head -1 | sed "s/^/$side /" | cut -c1-90
  done
done
done
