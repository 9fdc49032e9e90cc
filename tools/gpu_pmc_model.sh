#!/bin/bash
# per-kernel SQ counters of one bench configuration: bash tools/gpu_pmc_model.sh <model> <compute> <tile> <kernel substring>
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_model; mkdir -p $O
M=$1; C=$2; T=$3; K=$4
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_WAVES" "SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-30); rm -rf $O/$tag
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/$tag -- python $R/bench.py --model $M --compute $C --tile $T --no-cpu-baseline --no-kernel-events --steps 2 --warmup 1 > $O/$tag.log 2>&1
  f=$(find $O/$tag -name "*counter_collection.csv" | head -1)
  python - "$f" "$K" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    k=row.get("Kernel_Name","")
    if sys.argv[2] in k: acc[k[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,d in acc.items():
    print(k, {c: round(sum(v)/len(v)) for c,v in d.items()})
PY
done
