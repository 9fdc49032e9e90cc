#!/bin/bash
# A/B in ONE gpurun call: the committed baseline worktree (tools/abl/base) against the working tree, alternating
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for m in "rfdn_baseline bf16" "team04_rlfn bf16" "team18_bsrn f16"; do set -- $m
  for side in base new; do
    if [ $side = base ]; then cd $R/tools/abl/base; else cd $R; fi
    timeout 300 python bench.py --model $1 --compute $2 --no-cpu-baseline --steps 30 > /tmp/b_$side.json 2>/tmp/b.err || tail -2 /tmp/b.err
    python $R/tools/show_bench.py /tmp/b_$side.json | head -1 | sed "s/^/$side /" | cut -c1-80
  done
done
done
cd $R; python tools/show_bench.py /tmp/b_new.json | head -12
