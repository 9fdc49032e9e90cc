#!/bin/bash
# where the wave-cycles of the 48->48 3x3 conv_s16 launch go (SQ activity / wait counters)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_s16b; mkdir -p $O
LIBP=${1:-$R/ntire2022_esr_amd/libesr_hip.so}
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $O/avail.txt; wc -l $O/avail.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_THREAD_CYCLES_VALU" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SMEM SQ_WAIT_INST_ANY"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40); rm -rf $O/$tag
  timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/$tag -- python $R/tools/abl/probe_one.py 48 48 3 0 $LIBP > $O/$tag.log 2>&1
  f=$(find $O/$tag -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { echo "no csv for $set"; tail -3 $O/$tag.log; continue; }
  python - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if "conv_s16" in row.get("Kernel_Name",""): acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,v in acc.items(): print(f"{k:36s} {sum(v)/len(v):16.0f}   per wave-tile {sum(v)/len(v)/2048/16:10.1f}")
PY
  rm -rf $O/$tag
done
