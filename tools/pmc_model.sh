cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; M=$1
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$M/$c -- python $R/bench.py --model $M --no-cpu-baseline --no-kernel-events --steps 2 --warmup 1 > $R/gpurun_out/pmc_$M/$c.log 2>&1
done
cd $R
python - <<PY
import csv, glob, collections
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/pmc_$M/%s/**/*counter_collection.csv" % c, recursive=True)[0]
    acc = collections.defaultdict(lambda: [0.0, 0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        a = acc[r["Kernel_Name"][:60]]; a[0] += float(r["Counter_Value"]); a[1] += 1; a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    res[c] = acc
for k in res["FETCH_SIZE"]:
    f = res["FETCH_SIZE"][k]; w = res["WRITE_SIZE"].get(k, [0, 1, 0])
    if f[1] < 3: continue
    print(f"{k:62s} n={f[1]:4d} avg_us={f[2]/f[1]:8.1f} read_MB={2*f[0]/f[1]/1024:9.1f} write_MB={w[0]/w[1]/1024:9.1f} -> {(2*f[0]/f[1]+w[0]/w[1])/1024/1e3/(f[2]/f[1]/1e6)/1e3:6.2f} TB/s")
PY
find gpurun_out/pmc_$M -name "*.csv" -size +1M -delete
