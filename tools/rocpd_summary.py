#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (ROCm 7.2 default output of --kernel-trace --stats) into the
per-kernel summary committed under profiles/.   usage: rocpd_summary.py results.db > summary.md"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
print("| kernel | calls | total [us] | avg [us] | % |")
print("|---|---|---|---|---|")
for name, calls, tot, avg, pct in rows:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"| `{name}` | {calls} | {tot:.1f} | {avg:.3f} | {pct:.2f} |")
