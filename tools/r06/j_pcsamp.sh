#!/bin/bash
# round 6: PC sampling (stochastic) of conv64m_kernel -- which instructions do the waves sit on, and why
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06r; rm -rf $O; mkdir -p $O
cd /tmp
for m in stochastic host_trap; do
  if [ $m = stochastic ]; then U="--pc-sampling-unit cycles --pc-sampling-interval 8192"; else U="--pc-sampling-unit time --pc-sampling-interval 1"; fi
  (cd $R && timeout 200 rocprofv3 --pc-sampling-beta-enabled 1 --pc-sampling-method $m $U --kernel-trace --output-format csv -d $O/$m -- python tools/r06/run_c64m.py 100 1 > $O/$m.log 2>&1)
  echo "$m rc=$?"; tail -3 $O/$m.log; find $O/$m -type f | head
done
for f in $(find $O -name "*pc_sampling*.csv"); do echo $f; wc -l $f; head -3 $f; done
cd $R
python - <<'PY'
import csv, glob, collections, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r06r"
for f in glob.glob(O + "/**/*pc_sampling*.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    if not rows: continue
    print(f, len(rows), list(rows[0].keys()))
    keys = rows[0].keys()
    ins = next((k for k in keys if "nstruction" in k and "omment" not in k), None)
    c = collections.Counter()
    for r in rows:
        c[(r.get(ins, "")[:60], r.get("Stall_Reason", r.get("Wave_Issued", "")))] += 1
    for (i, s), n in c.most_common(40):
        print(f"   {n:7d}  {s:28s} {i}")
PY
for f in $(find $O -name "*pc_sampling*.csv"); do gzip -9 $f; done
du -sh $O
