#!/bin/bash
O=gpurun_out/r02e; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep -n "passed\|failed\|rc=\|FAILED\|Error" $O/pytest.log | tail -30
# wall-clock of the evaluation loop on a DIV2K-shaped synthetic set (PNG decode / encode included)
cd $O && timeout 600 python -m ntire2022_esr_amd.harness --model_id 4 --synthetic 40 --save_dir /tmp/syn_out > harness_syn.log 2>&1; tail -4 harness_syn.log; cat pipeline.json; cd - > /dev/null
python - <<'PY'
import time,sys,types,logging,torch,os
sys.path.insert(0,os.getcwd())
from ntire2022_esr_amd import harness as H
from ntire2022_esr_amd.registry import select_model
dev=torch.device("cuda:0")
model,name,dr,tile=select_model(4,dev)
pairs=H.make_synthetic_dataset("/tmp/syn_out/_synthetic",40)
log=logging.getLogger("x")
for label,kw in (("serial",dict(device_metrics=False)),("pipeline w4",dict(io_workers=4,inflight=3)),("pipeline w8",dict(io_workers=8,inflight=4)),("pipeline w16",dict(io_workers=16,inflight=6))):
    a=types.SimpleNamespace(save_dir="/tmp/syn_out/"+label.replace(" ","_"),rank=0,world=1,**kw)
    t0=time.perf_counter(); r=H.run(model,name,dr,tile,log,dev,a,mode="valid",pairs=pairs); t=time.perf_counter()-t0
    print(f"{label}: {40/t:.2f} images/s wall, mean forward {r['valid_ave_runtime']:.2f} ms", flush=True)
PY
