#!/bin/bash
# round 4, call a: the overlapped-forward hunt with per-group dumps (tools/dbg/race_dump.py) + SQ counters of the headline config
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r04a; mkdir -p $O
cd $R
echo "== control: old loop, 4 streams" > $O/race.txt
timeout 300 python tools/dbg/streams_race.py team04_rlfn bf16 200 $R/tools/abl/libesr_l_old.so 2>&1 | grep -E "mismatching|serial" >> $O/race.txt
echo "== old loop with dumps" >> $O/race.txt
timeout 600 python tools/dbg/race_dump.py run old 300 >> $O/race.txt 2>&1
echo "== old loop, GPU_MAX_HW_QUEUES=1" >> $O/race.txt
GPU_MAX_HW_QUEUES=1 timeout 300 python tools/dbg/streams_race.py team04_rlfn bf16 200 $R/tools/abl/libesr_l_old.so 2>&1 | grep -E "mismatching|serial" >> $O/race.txt
echo "== nopref with dumps (control)" >> $O/race.txt
timeout 300 python tools/dbg/race_dump.py run nopref 100 2>&1 | tail -3 >> $O/race.txt
echo "== pingpong with dumps" >> $O/race.txt
timeout 600 python tools/dbg/race_dump.py run pingpong 200 >> $O/race.txt 2>&1
cat $O/race.txt | cut -c1-400
# SQ counters, headline config
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE GRBM_COUNT SQ_WAVES SQ_INSTS_VMEM_RD"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$tag -- python $R/bench.py --no-cpu-baseline --no-kernel-events --steps 2 --warmup 1 > $O/pmc_$tag.log 2>&1
  f=$(find $O/pmc_$tag -name "*counter_collection.csv" | head -1)
  python - "$f" >> $O/sq_counters.txt <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
try:
    for row in csv.DictReader(open(sys.argv[1])):
        acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k,d in acc.items():
        if "wino" in k or "imdb" in k:
            print(k, {c: round(sum(v)/len(v)) for c,v in d.items()}, "launches", len(next(iter(d.values()))))
except Exception as e: print("ERR", e, sys.argv[1])
PY
  find $O/pmc_$tag -name "*.csv" -size +1M -delete
done
cat $O/sq_counters.txt
cd $R; python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_c1.json 2>$O/bench_c1.err; python tools/show_bench.py $O/bench_c1.json | head -12
