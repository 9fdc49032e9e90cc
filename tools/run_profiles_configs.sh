# rocprofv3 kernel stats for the secondary BASELINE configs (RFDN bf16, RLFN bf16, BSRN fp16, IMDN bf16) -> gpurun_out/r01g/
set -x
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r01g
for mc in "imdn_baseline bf16" "rfdn_baseline bf16" "team04_rlfn bf16" "team18_bsrn f16"; do
  set -- $mc; M=$1; C=$2
  timeout 300 python bench.py --no-cpu-baseline --steps 10 --model $M --compute $C | tail -1 > gpurun_out/r01g/${M}_${C}.json
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r01g/stats_${M}_${C} -- python $R/bench.py --no-cpu-baseline --steps 10 --model $M --compute $C > $R/gpurun_out/r01g/${M}_${C}.log 2>&1
  cd $R
  DB=$(find gpurun_out/r01g/stats_${M}_${C} -name "*.db" | head -1)
  python tools/rocpd_summary.py $DB > gpurun_out/r01g/${M}_${C}_kernel_stats.md
  rm -rf gpurun_out/r01g/stats_${M}_${C}
done
ls -la gpurun_out/r01g
