#!/bin/bash
# race hunt: streams_race.py for the product library and every tools/abl/libesr_r_*.so     bash tools/dbg/race_run.sh [rounds] [model compute]
R=$GRAFT_REPO_ROOT; N=${1:-150}; M=${2:-team04_rlfn}; C=${3:-bf16}
echo "== product"; python $R/tools/dbg/streams_race.py $M $C $N 2>&1 | grep -E "mismatching|serial|s2pool16" 
echo "== product, synchronize after every forward"; python $R/tools/dbg/streams_race.py $M $C $N - sync 2>&1 | grep -E "mismatching|serial|s2pool16|round" | head -12
for so in $R/tools/abl/libesr_r_*.so; do echo "== $(basename $so)"; python $R/tools/dbg/streams_race.py $M $C $N $so 2>&1 | grep -E "mismatching|serial|s2pool16"; done
