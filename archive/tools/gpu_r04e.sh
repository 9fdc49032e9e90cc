#!/bin/bash
# round 4, call e: the op_sel erratum -- isolated probe, the round-3 loop with and without esr_lone, then the whole GPU suite
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r04e; mkdir -p $O
cd $R
timeout 200 ./tools/dbg/pk_opsel_probe 0.4 quick > $O/pk_opsel_probe_quick.txt 2>&1
cat $O/pk_opsel_probe_quick.txt | cut -c1-260
for v in old_nolone old; do
  echo "== round-3 loop, $v (old = with esr_lone)" >> $O/race.txt
  timeout 400 python tools/dbg/streams_race.py team04_rlfn bf16 250 $R/tools/abl/libesr_l_$v.so 2>&1 | grep -E "mismatching|serial" >> $O/race.txt
done
cat $O/race.txt
timeout 2400 python -m pytest tests -m gpu -x -q -rP 2>&1 | tail -150 > $O/gputests_tail.txt
grep -E "passed|failed|error" $O/gputests_tail.txt | tail -5
