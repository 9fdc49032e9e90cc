#!/bin/bash
# round 5, call r: EXECUTED instruction mix per kernel (SQ counters, one pass per set) for the three 16-bit configs at batch 32
O=$GRAFT_REPO_ROOT/gpurun_out/r05r; mkdir -p $O; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
for mc in "team04_rlfn --compute bf16" "rfdn_baseline --compute bf16" "team18_bsrn --compute f16 --tile 270x480"; do
  set -- $mc; name=$1
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD"; do
    tag=$(echo $set | cut -d' ' -f1)
    timeout 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/${name}_$tag -- python $R/bench.py --model $mc --no-cpu-baseline --no-other-configs --no-kernel-events --steps 2 --warmup 1 > $O/${name}_$tag.log 2>&1
    f=$(find $O/${name}_$tag -name "*counter_collection.csv" | head -1)
    python - "$f" "$name" >> $O/sq_mix.txt <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
try:
    for row in csv.DictReader(open(sys.argv[1])):
        acc[row["Kernel_Name"].replace("void (anonymous namespace)::","")[:56]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k,d in acc.items():
        if any(s in k for s in ("Fill","copyBuffer","probe")): continue
        v={c: sum(x)/len(x) for c,x in d.items()}
        extra=""
        if "SQ_INSTS_MFMA" in v and v["SQ_INSTS_MFMA"]>0:
            extra=" others/MFMA %.2f (VALU-MFMA %.2f, SALU %.2f, LDS %.2f)"%((v["SQ_INSTS_VALU"]-v["SQ_INSTS_MFMA"]+v["SQ_INSTS_SALU"]+v["SQ_INSTS_LDS"])/v["SQ_INSTS_MFMA"], (v["SQ_INSTS_VALU"]-v["SQ_INSTS_MFMA"])/v["SQ_INSTS_MFMA"], v["SQ_INSTS_SALU"]/v["SQ_INSTS_MFMA"], v["SQ_INSTS_LDS"]/v["SQ_INSTS_MFMA"])
        print(sys.argv[2], k, {c: round(x) for c,x in v.items()}, "launches", len(next(iter(d.values()))), extra)
except Exception as e: print("ERR", e, sys.argv[1])
PY
    find $O/${name}_$tag -name "*.csv" -size +1M -delete
  done
done
