#!/bin/bash
# round-2 profiles: for every BASELINE config a bench line, rocprofv3 --kernel-trace --stats and the two PMC traffic passes
#   bash tools/profile_r02.sh <tag>      -> gpurun_out/<tag>/ ; copy the .md / .json summaries into profiles/
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
run_cfg() {   # name, bench options...
  local name=$1; shift
  timeout 400 python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  cd /tmp
  timeout 400 rocprofv3 --kernel-trace --stats -d $O/stats_$name -- python $R/bench.py --no-cpu-baseline --steps 10 "$@" > $O/stats_$name.log 2>&1
  cd $R
  DB=$(find $O/stats_$name -name "*.db" | head -1)
  { echo "# $TAG $name -- rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --steps 10 $*"; echo;
    echo "un-profiled bench line of the same build, same gpurun call:"; echo '```'; tail -1 $O/bench_$name.json; echo '```'; echo;
    python tools/rocpd_summary.py $DB | head -16; } > $O/${TAG}_${name}_kernel_stats.md
  rm -rf $O/stats_$name
}
pmc_cfg() {   # name, bench options (tile mode only)
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
run_cfg c2_rfdn_bf16_div2k --model rfdn_baseline --compute bf16 --sizes div2k --no-cpu-baseline
run_cfg c2_rfdn_bf16_div2k_s1 --model rfdn_baseline --compute bf16 --sizes div2k --streams 1 --no-cpu-baseline
run_cfg c2_rfdn_bf16_b32 --model rfdn_baseline --compute bf16 --no-cpu-baseline
pmc_cfg c2_rfdn_bf16_b32 --model rfdn_baseline --compute bf16
run_cfg c3_rlfn_bf16_div2k --model team04_rlfn --compute bf16 --sizes div2k --no-cpu-baseline
run_cfg c3_rlfn_bf16_div2k_s1 --model team04_rlfn --compute bf16 --sizes div2k --streams 1 --no-cpu-baseline
run_cfg c3_rlfn_bf16_b32 --model team04_rlfn --compute bf16 --no-cpu-baseline
pmc_cfg c3_rlfn_bf16_b32 --model team04_rlfn --compute bf16
run_cfg c4_bsrn_f16_270x480 --model team18_bsrn --compute f16 --tile 270x480 --no-cpu-baseline
pmc_cfg c4_bsrn_f16_270x480 --model team18_bsrn --compute f16 --tile 270x480
run_cfg x_imdn_bf16_b32 --model imdn_baseline --compute bf16 --no-cpu-baseline
python bench.py --b1-latency --no-cpu-baseline --no-kernel-events > $O/b1_imdn_f32.json 2>/dev/null
python bench.py --streams 2 --no-cpu-baseline --no-kernel-events > $O/bench_c1_imdn_f32_2streams.json 2>/dev/null
python bench.py --sizes div2k --no-cpu-baseline > $O/bench_x_imdn_f32_div2k.json 2>/dev/null
python bench.py --model team18_bsrn --compute f16 --sizes div2k --no-cpu-baseline > $O/bench_x_bsrn_f16_div2k.json 2>/dev/null
for mc in "rfdn_baseline bf16" "team04_rlfn bf16" "team18_bsrn f16" "imdn_baseline f32"; do set -- $mc
  python bench.py --model $1 --compute $2 --tile 339x510 --batch 1 --b1-latency --no-cpu-baseline --no-kernel-events --steps 50 > $O/b1_$1_$2_339x510.json 2>/dev/null
done
ls $O; du -sh $O
