#!/bin/bash
# the product library: overlapped against serial forwards for the three ESA networks (16-bit) and RLFN fp32   bash tools/dbg/race_all.sh [rounds]
R=$GRAFT_REPO_ROOT; N=${1:-300}
for m in "team04_rlfn bf16" "rfdn_baseline bf16" "team18_bsrn f16" "team04_rlfn f16" "team04_rlfn f32"; do echo "== $m"; python $R/tools/dbg/streams_race.py $m $N 2>&1 | grep -E "mismatching|serial"; done
