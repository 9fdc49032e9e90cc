#!/bin/bash
# round-6 profiles: for every BASELINE config a bench line, rocprofv3 --kernel-trace --stats and the two PMC traffic passes
#   bash tools/profile_r06.sh <tag>      -> gpurun_out/<tag>/ ; copy the .md / .json summaries into profiles/
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
run_cfg() {   # name, bench options...
  local name=$1; shift
  timeout 400 python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  cd /tmp
  timeout 400 rocprofv3 --kernel-trace --stats -d $O/stats_$name -- python $R/bench.py --no-cpu-baseline --no-other-configs --steps 10 "$@" > $O/stats_$name.log 2>&1
  cd $R
  DB=$(find $O/stats_$name -name "*.db" | head -1)
  { echo "# $TAG $name -- rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-other-configs --steps 10 $*"; echo;
    echo "un-profiled bench line of the same build, same gpurun call:"; echo '```'; tail -1 $O/bench_$name.json; echo '```'; echo;
    python tools/rocpd_summary.py $DB | head -16; } > $O/${TAG}_${name}_kernel_stats.md
  rm -rf $O/stats_$name
}
pmc_cfg() {   # name, bench options (tile or div2k mode; div2k passes run on ONE stream so that launches do not overlap)
  local name=$1; shift
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${name}_$c -- python $R/bench.py --no-cpu-baseline --no-kernel-events --steps 2 --warmup 1 "$@" > $O/pmc_${name}_$c.log 2>&1
  done
  cd $R
  python tools/pmc_traffic.py $O/pmc_${name}_FETCH_SIZE $O/pmc_${name}_WRITE_SIZE $O/pmc_traffic.json $TAG "$@" | tail -40 > $O/pmc_${name}.txt
  find $O/pmc_${name}_FETCH_SIZE $O/pmc_${name}_WRITE_SIZE -name "*.csv" -size +1M -delete
}
cp profiles/pmc_traffic.json $O/pmc_traffic.json 2>/dev/null
run_cfg c1_imdn_f32
pmc_cfg c1_imdn_f32
run_cfg c2_rfdn_bf16_div2k --model rfdn_baseline --compute bf16 --sizes div2k
run_cfg c2_rfdn_bf16_div2k_s1 --model rfdn_baseline --compute bf16 --sizes div2k --streams 1 --no-cpu-baseline
pmc_cfg c2_rfdn_bf16_div2k --model rfdn_baseline --compute bf16 --sizes div2k --streams 1
run_cfg c2_rfdn_bf16_b32 --model rfdn_baseline --compute bf16 --no-cpu-baseline
pmc_cfg c2_rfdn_bf16_b32 --model rfdn_baseline --compute bf16
run_cfg c3_rlfn_bf16_div2k --model team04_rlfn --compute bf16 --sizes div2k
run_cfg c3_rlfn_bf16_div2k_s1 --model team04_rlfn --compute bf16 --sizes div2k --streams 1 --no-cpu-baseline
pmc_cfg c3_rlfn_bf16_div2k --model team04_rlfn --compute bf16 --sizes div2k --streams 1
run_cfg c3_rlfn_bf16_b32 --model team04_rlfn --compute bf16 --no-cpu-baseline
pmc_cfg c3_rlfn_bf16_b32 --model team04_rlfn --compute bf16
run_cfg c4_bsrn_f16_270x480 --model team18_bsrn --compute f16 --tile 270x480
pmc_cfg c4_bsrn_f16_270x480 --model team18_bsrn --compute f16 --tile 270x480
run_cfg x_imdn_bf16_b32 --model imdn_baseline --compute bf16 --no-cpu-baseline
for a in "4 bf16 339x510" "0 bf16 339x510" "18 f16 270x480" "-1 f32 256x256"; do timeout 200 python tools/b1_latency.py $a 2>&1 | grep -v amdgpu.ids >> $O/${TAG}_host_latency.txt; done
timeout 300 bash -c 'for a in "4 bf16 32 256x256" "4 bf16 1 339x510" "0 bf16 32 256x256" "0 bf16 1 339x510" "18 f16 32 270x480" "18 f16 1 339x510" "-1 f32 32 256x256"; do python tools/per_op.py $a 2>&1 | grep -v amdgpu.ids; done' > $O/${TAG}_per_op.txt
ESR_BW_PROBE_VERBOSE=1 timeout 200 python -c "
import ctypes, torch
from ntire2022_esr_amd import _lib as L
lib = L.lib(); st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for mb, reps in ((16, 64), (256, 4), (1024, 1)):
    buf = torch.empty(2 * (mb << 20), dtype=torch.uint8, device='cuda'); g = ctypes.c_double(0)
    for _ in range(2): L.check(lib.esr_bw_probe(ctypes.c_void_p(buf.data_ptr()), mb << 20, reps, st, ctypes.byref(g)), 'bw')
    print(f'== 2 x {mb} MiB x {reps}: best {g.value:.1f} GB/s', flush=True); del buf
" 2>&1 | grep "==" > $O/${TAG}_copy_roof_summary.txt
python bench.py --b1-latency --no-cpu-baseline --no-kernel-events > $O/b1_imdn_f32.json 2>/dev/null
python bench.py --streams 2 --no-cpu-baseline --no-kernel-events > $O/bench_c1_imdn_f32_2streams.json 2>/dev/null
python bench.py --sizes div2k --no-cpu-baseline > $O/bench_x_imdn_f32_div2k.json 2>/dev/null
python bench.py --model team18_bsrn --compute f16 --sizes div2k --no-cpu-baseline > $O/bench_x_bsrn_f16_div2k.json 2>/dev/null
for mc in "rfdn_baseline bf16" "team04_rlfn bf16" "team18_bsrn f16" "imdn_baseline f32"; do set -- $mc
  python bench.py --model $1 --compute $2 --tile 339x510 --batch 1 --b1-latency --no-cpu-baseline --no-kernel-events --steps 50 > $O/b1_$1_$2_339x510.json 2>/dev/null
done
# SQ counters of the headline config (separate passes)
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE GRBM_COUNT SQ_WAVES SQ_INSTS_VMEM_RD"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/sq_$tag -- python $R/bench.py --no-cpu-baseline --no-kernel-events --steps 2 --warmup 1 > $O/sq_$tag.log 2>&1
  f=$(find $O/sq_$tag -name "*counter_collection.csv" | head -1)
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
  find $O/sq_$tag -name "*.csv" -size +1M -delete
done
cd $R
timeout 2400 python -m pytest tests -m gpu -q -rP 2>&1 | grep -vE "^$" | tail -300 > $O/${TAG}_gputests.txt
ls $O; du -sh $O
