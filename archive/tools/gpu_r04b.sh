#!/bin/bash
# round 4, call b: which instruction pair of the old apply loop goes wrong (tools/dbg/race_dump.py experiments)
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r04b; mkdir -p $O
cd $R
for v in old old_nopC old_ldswait old_zeroC old_biaslate; do
  echo "== plain $v" >> $O/race.txt
  timeout 300 python tools/dbg/streams_race.py team04_rlfn bf16 150 $R/tools/abl/libesr_l_$v.so 2>&1 | grep -E "mismatching|serial" >> $O/race.txt
done
echo "== old with dumps (C input)" >> $O/race.txt
timeout 600 python tools/dbg/race_dump.py run old 100 2>&1 | grep -E "^round|group|  s\.|  C\.|mismatching" | cut -c1-300 | head -60 >> $O/race.txt
cat $O/race.txt
