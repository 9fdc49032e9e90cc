#!/bin/bash
# round 5, call c: the whole GPU suite on the current tree + the default bench line
O=$GRAFT_REPO_ROOT/gpurun_out/r05c; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee $O/t.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
