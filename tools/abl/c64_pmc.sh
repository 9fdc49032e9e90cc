#!/bin/bash
# SQ counters of conv64r_kernel (64 -> 64, batch 32), separate passes; run from the repo root on the GPU box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c64pmc; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/$tag -- python $R/tools/abl/c64_abl.py run1 > $O/$tag.log 2>&1
  f=$(find $O/$tag -name "*counter_collection.csv" | head -1)
  python - "$f" >> $O/counters.txt <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
try:
    for row in csv.DictReader(open(sys.argv[1])):
        acc[row["Kernel_Name"][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k,d in acc.items():
        if "conv64r" in k or "conv48r" in k:
            print(k, {c: round(sum(v)/len(v)) for c,v in d.items()}, "launches", len(next(iter(d.values()))))
except Exception as e: print("ERR", e, sys.argv[1])
PY
  find $O/$tag -name "*.csv" -size +1M -delete
done
cat $O/counters.txt
