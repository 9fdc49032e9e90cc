#!/bin/bash
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r04c; mkdir -p $O
cd $R
timeout 600 python tools/dbg/race_dump.py run old 40 2>&1 | grep -E "^round|group|mismatching" | cut -c1-200 | head -20 > $O/race.txt
ls -la $R/gpurun_out/race_dump >> $O/race.txt
cat $O/race.txt
