#!/bin/bash
# every tools/abl/libesr_w8_*.so through tools/wino/w8_ab.py, twice round-robin (one gpurun call = one box)
R=$GRAFT_REPO_ROOT; cd $R
true
for rep in 1 2; do for so in tools/abl/libesr_w8_*.so; do echo "== $(basename $so)"; python tools/wino/w8_ab.py 1 $so | grep "wino8=1" | cut -c1-130; done; done
