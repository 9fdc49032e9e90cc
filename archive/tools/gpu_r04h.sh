#!/bin/bash
# round 4, hi + lo skip: kernel tests, network-level PSNR tests, benches of the two bf16 configs with and without the pairs
O=$GRAFT_REPO_ROOT/gpurun_out/r04h; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_h16.py -q -x -k "hilo" 2>&1 | tail -15 > $O/t_hilo.txt
timeout 1500 python -m pytest tests/test_gpu_multi.py tests/test_gpu_h16.py tests/test_gpu_big.py -q -rP -k "bf16 or hilo" 2>&1 | grep -E "dPSNR|PSNR|passed|failed|Error|assert" | tail -60 > $O/t_net.txt
for m in team04_rlfn rfdn_baseline; do
  for hl in 1 0; do
    timeout 300 python bench.py --model $m --compute bf16 --no-cpu-baseline --no-other-configs $([ $hl = 0 ] && echo --no-hilo-skip) > $O/b32_${m}_hl$hl.json 2> $O/b32_${m}_hl$hl.err
    timeout 300 python bench.py --model $m --compute bf16 --sizes div2k --streams 1 --no-cpu-baseline --no-other-configs $([ $hl = 0 ] && echo --no-hilo-skip) > $O/div2k_${m}_hl$hl.json 2> $O/div2k_${m}_hl$hl.err
  done
done
python - <<'PY' > $O/summary.txt
import json,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r04h/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); r=j["roofline"]
        print(os.path.basename(f), j["value"], j["ms_per_step"], [(k["kernel"],k["avg_ms"]) for k in r["kernels"][:8]])
    except Exception as e: print(f, "ERR", e)
PY
cat $O/t_hilo.txt $O/t_net.txt $O/summary.txt
