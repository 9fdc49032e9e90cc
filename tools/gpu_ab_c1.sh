#!/bin/bash
# A/B of the headline config in ONE gpurun call: tools/abl/base (a built worktree of the baseline commit) against the working tree, alternating
R=$GRAFT_REPO_ROOT; N=${1:-3}
for rep in $(seq $N); do
  for side in base new; do
    if [ $side = base ]; then cd $R/tools/abl/base; X=""; else cd $R; X="--no-other-configs"; fi
    python bench.py --help 2>/dev/null | grep -q no-other-configs && X="--no-other-configs"
    timeout 300 python bench.py --no-cpu-baseline $X --steps 30 > /tmp/b_$side.json 2>/tmp/b.err || tail -2 /tmp/b.err
    python $R/tools/show_bench.py /tmp/b_$side.json | head -4 | sed "s/^/$side /" | cut -c1-200
  done
done
