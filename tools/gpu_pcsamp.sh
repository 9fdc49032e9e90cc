#!/bin/bash
# PC sampling of one conv_s16 launch config: where the waves stall (research tooling)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pcsamp; rm -rf $O; mkdir -p $O
ARGS=${ARGS:-"48 48 3 0"}
cd /tmp
for m in stochastic host_trap; do
  if [ $m = stochastic ]; then U="--pc-sampling-unit cycles --pc-sampling-interval 16384"; else U="--pc-sampling-unit time --pc-sampling-interval 1"; fi
  timeout 150 rocprofv3 --pc-sampling-beta-enabled 1 --pc-sampling-method $m $U --kernel-trace --output-format csv -d $O/$m -- python $R/tools/abl/probe_one.py $ARGS > $O/$m.log 2>&1
  echo "$m rc=$?"; tail -3 $O/$m.log; find $O/$m -type f | head; 
done
for f in $(find $O -name "*pc_sampling*.csv"); do echo $f; wc -l $f; head -3 $f; done
# keep the merge small
for f in $(find $O -name "*pc_sampling*.csv"); do gzip -9 $f; done
du -sh $O
