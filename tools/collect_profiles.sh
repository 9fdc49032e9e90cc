#!/bin/bash
# copy the summaries of a tools/profile_r06.sh run (gpurun_out/<tag>/) into profiles/ under the names DESIGN.md cites
TAG=${1:-r06}; O=gpurun_out/$TAG
cd "$(dirname "$0")/.."
for f in $O/${TAG}_*_kernel_stats.md $O/${TAG}_host_latency.txt $O/${TAG}_per_op.txt $O/${TAG}_copy_roof_summary.txt $O/${TAG}_gputests.txt; do [ -s "$f" ] && cp "$f" profiles/; done
for f in $O/bench_*.json $O/b1_*.json; do [ -s "$f" ] && tail -1 "$f" > profiles/${TAG}_$(basename "$f"); done
for f in $O/pmc_*.txt; do [ -s "$f" ] && cp "$f" profiles/${TAG}_$(basename "$f"); done
[ -s $O/pmc_traffic.json ] && cp $O/pmc_traffic.json profiles/pmc_traffic.json
[ -s $O/sq_counters.txt ] && cp $O/sq_counters.txt profiles/${TAG}_sq_counters.txt
git status --short profiles | head -60
