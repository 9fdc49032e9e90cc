#!/usr/bin/env python3
"""print the headline value and the kernel table of bench.py JSON lines: show_bench.py file [file ...]"""
import json, sys
for f in sys.argv[1:]:
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        r = j["roofline"]
        print(f"{f}: {j['value']} {j['unit']}  {j['ms_per_step']} ms/step  [{j['config'].get('workload', '')}]  {r['bound']} {r['kernel']} frac {r['frac']}")
        for k in r["kernels"][:7]:
            print("    ", k)
    except Exception as e:
        print(f, "ERR", e)
