#!/bin/bash
# quick loop: s16 correctness, one-launch timings, instruction counters of the 48->48 3x3, model benches
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/quick; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_h16.py tests/test_gpu_big.py tests/test_gpu_esa_models.py -m gpu -q -x 2>&1 | tail -3
python - <<'PY'
import ctypes, os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from ntire2022_esr_amd import _lib as L
from ntire2022_esr_amd.engine import pack_conv_s16
lib = L.lib(); dev = "cuda:0"
for (cin, cout, k, res) in ((48, 48, 3, 0), (64, 64, 3, 0), (64, 64, 3, 1), (48, 48, 3, 2), (48, 48, 1, 0), (128, 64, 1, 0), (64, 32, 1, 0)):
    x = torch.randn(32, 256, 256, cin, device=dev).to(torch.bfloat16); y = torch.zeros(32, 256, 256, cout, device=dev, dtype=torch.bfloat16)
    r = x if res == 1 else torch.randn(32, 256, 256, cout, device=dev).to(torch.bfloat16)
    blob = pack_conv_s16(torch.randn(cout, cin, k, k) * 0.1, torch.randn(cout), "bf16").to(dev)
    d = L.ConvDesc(); d.n, d.h, d.w, d.cin, d.cout, d.ksize = 32, 256, 256, cin, cout, k
    d.act, d.slope, d.storage, d.compute = 1, 0.05, 1, 1
    d.inp = L.View(x.data_ptr(), cin, 0); d.out0 = L.View(y.data_ptr(), cout, 0)
    if res: d.res_mode, d.res = (1 if res == 1 else 2), L.View(r.data_ptr(), cout, 0)
    d.wpacked = blob.data_ptr(); st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3): assert lib.esr_conv2d_f32(ctypes.byref(d), st) == 0
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(20): lib.esr_conv2d_f32(ctypes.byref(d), st)
    e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 20
    gb = 32 * 65536 * (cin + cout + (cout if res == 2 else 0)) * 2 / 1e9
    print(f"{cin:3d}->{cout:3d} k{k} res{res}: {ms:.4f} ms  {gb / ms:.2f} TB/s  {2 * 32 * 65536 * cin * cout * k * k / ms / 1e9:.0f} TFLOP/s", flush=True)
PY
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-30); rm -rf $O/$tag
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/$tag -- python $R/tools/abl/probe_one.py 48 48 3 0 > $O/$tag.log 2>&1
  f=$(find $O/$tag -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if "conv_s16" in row.get("Kernel_Name",""): acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,v in acc.items(): print(k, round(sum(v)/len(v)/2048/16), "per wave-tile")
PY
done
cd $R
for m in "team04_rlfn bf16" "rfdn_baseline bf16" "imdn_baseline bf16"; do set -- $m; timeout 200 python bench.py --model $1 --compute $2 --no-cpu-baseline > $O/b_$1.json 2>/dev/null; python -c "
import json,sys; j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], j['value']); [print('  ',k) for k in j['roofline']['kernels'][:5]]" $O/b_$1.json; done
