#!/bin/bash
# round 5, call p: esa_chain_kernel with its weights requested one layer ahead; occupancy targets none (216 registers) / 3 waves per SIMD (168 + 84 B scratch)
O=$GRAFT_REPO_ROOT/gpurun_out/r05p; mkdir -p $O; cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in "" cw3; do
  if [ -n "$v" ]; then export ESR_HIP_LIB=$R/tools/r05/libesr_$v.so; else unset ESR_HIP_LIB; fi
  for mc in "team04_rlfn --compute bf16" "team04_rlfn --compute bf16 --sizes div2k --streams 1" "team18_bsrn --compute f16 --tile 270x480" "rfdn_baseline --compute bf16"; do
    cd /tmp
    timeout 300 rocprofv3 --kernel-trace --stats -d $O/st_$v -- python $R/bench.py --model $mc --no-cpu-baseline --no-other-configs --no-kernel-events --steps 10 > $O/st.log 2>&1
    cd $R
    DB=$(find $O/st_$v -name "*.db" | head -1)
    echo "lib=$v $mc: $(tail -1 $O/st.log | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])" 2>/dev/null)  $(python tools/rocpd_summary.py $DB | grep "esa_chain_kernel\|s2pool16" | awk -F'|' '{print $2, $5}' | tr '\n' ' ')" | tee -a $O/sum.txt
    rm -rf $O/st_$v
  done
done
