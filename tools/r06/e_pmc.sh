#!/bin/bash
# round 6: SQ / TA / TCP counters of conv64m_kernel (separate passes, --kernel-trace only) -- where do the waves wait?
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r06j; mkdir -p $O
cd /tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
for lib in prod abl3; do
  [ $lib = prod ] && unset ESR_HIP_LIB || export ESR_HIP_LIB=$R/tools/r06/libesr_$lib.so
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum" "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_${lib}_$i -- python $R/bench.py --model rfdn_baseline --compute bf16 --no-cpu-baseline --no-other-configs --no-kernel-events --steps 2 --warmup 1 > $O/pmc_${lib}_$i.log 2>&1
    f=$(find $O/pmc_${lib}_$i -name "*counter_collection.csv" | head -1)
    python - "$f" "$lib" >> $O/pmc_summary.txt <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
try:
    for row in csv.DictReader(open(sys.argv[1])):
        acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k,d in acc.items():
        if "conv64m" in k:
            print(sys.argv[2], k[20:75], {c: round(sum(v)/len(v)) for c,v in d.items()}, "launches", len(next(iter(d.values()))))
except Exception as e: print("ERR", e, sys.argv[1:])
PY
    find $O/pmc_${lib}_$i -name "*.csv" -size +1M -delete
  done
done
cat $O/pmc_summary.txt
grep -c . $O/counters_list.txt
