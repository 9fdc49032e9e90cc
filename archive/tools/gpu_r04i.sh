#!/bin/bash
# round 4, hi + lo skip in all four networks: the whole GPU suite, then A/B bench lines (bf16, with / without the pairs)
O=$GRAFT_REPO_ROOT/gpurun_out/r04i; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q -rP 2>&1 | grep -vE "^$" | tail -400 > $O/gputests.txt
for m in imdn_baseline team18_bsrn team04_rlfn rfdn_baseline; do
  for hl in 1 0; do
    timeout 300 python bench.py --model $m --compute bf16 --no-cpu-baseline --no-other-configs $([ $hl = 0 ] && echo --no-hilo-skip) > $O/b32_${m}_hl$hl.json 2> $O/b32_${m}_hl$hl.err
  done
done
python - <<'PY' > $O/summary.txt
import json,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r04i/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); r=j["roofline"]
        print(os.path.basename(f), j["value"], j["ms_per_step"], [(k["kernel"],k["avg_ms"]) for k in r["kernels"][:8]])
    except Exception as e: print(f, "ERR", e)
PY
grep -E "passed|failed|smooth|hi \+ lo" $O/gputests.txt | tail -20; cat $O/summary.txt
