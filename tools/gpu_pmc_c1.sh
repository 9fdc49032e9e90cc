#!/bin/bash
# PMC traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the headline config only -> gpurun_out/<tag>/pmc_c1_imdn_f32.txt
TAG=${1:-pmc_c1}
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/$TAG; mkdir -p $O
cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json 2>/dev/null
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python $R/bench.py --no-cpu-baseline --no-kernel-events --steps 2 --warmup 1 > $O/pmc_$c.log 2>&1
done
cd $R
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_traffic.json $TAG | tail -40 > $O/pmc_c1_imdn_f32.txt
find $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE -name "*.csv" -size +1M -delete
cat $O/pmc_c1_imdn_f32.txt
