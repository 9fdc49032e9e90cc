#!/usr/bin/env python3
"""time of the ESA apply launch inside a whole-model bench for the product library and every tools/abl/libesr_r_abl_*.so (ABLATIONS: their
results are wrong on purpose)   apply_abl.py [model compute]"""
import json, os, subprocess, sys, glob
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
model, compute = (sys.argv[1:3] + ["team04_rlfn", "bf16"])[:2] if len(sys.argv) > 2 else ("team04_rlfn", "bf16")
for so in [os.path.join(REPO, "ntire2022_esr_amd", "libesr_hip.so")] + sorted(glob.glob(os.path.join(REPO, "tools", "abl", "libesr_r_abl_*.so"))):
    code = (f"import sys; sys.path.insert(0, {REPO!r}); import ntire2022_esr_amd._lib as L; L.SO_PATH = {so!r}; import runpy; "
            f"sys.argv = ['bench.py', '--model', {model!r}, '--compute', {compute!r}, '--no-cpu-baseline', '--steps', '30']; "
            f"runpy.run_path({os.path.join(REPO, 'bench.py')!r}, run_name='__main__')")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=REPO)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        k = [x for x in d["roofline"]["kernels"] if "esa_apply" in x["kernel"]][0]
        print(f"{os.path.basename(so):32s} {d['value']:9.1f} img/s   {k['kernel']}: {k['avg_ms']:.4f} ms", flush=True)
    except Exception as e:
        print(os.path.basename(so), "FAILED", e, out.stderr[-300:])
