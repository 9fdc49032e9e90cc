#!/usr/bin/env python3
"""profiles/pmc_traffic.json from two rocprofv3 PMC passes (run on the GPU box).

usage: pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json> [round tag]
Each pass:  rocprofv3 --kernel-trace --pmc <COUNTER> --output-format csv -d <dir> -- \
            python bench.py --no-cpu-baseline --no-kernel-events --steps 2 --warmup 1
Counter units are KB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies 128-byte requests at
64 B, so read bytes = 2 x FETCH_SIZE; WRITE_SIZE is exact.  Infinity-Cache hits are counted, so this is fabric traffic,
an upper bound on HBM traffic.  The algorithmic bytes come from the op list of the model (engine.collect_profile)."""
import collections, csv, glob, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_kernel(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    assert files, f"no counter_collection.csv under {d}"
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            a = acc[r["Kernel_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}


def label(sym):
    m = re.search(r"conv_f32_kernel<(\d+), (\d+), (true|false), (\d+), (\d+), (\d+)>", sym)
    if not m:
        return None
    base = f"conv_f32_kernel<NT={m.group(1)},KS={m.group(2)},NCHW_IN={int(m.group(3) == 'true')},NW={m.group(4)}"
    if m.group(5) != "0":
        base += f",TAIL={m.group(5)}"
    if m.group(6) != "0":
        base += f",POST={m.group(6)}"
    return base + ">"


def main():
    fdir, wdir, out = sys.argv[1:4]
    tag = sys.argv[4] if len(sys.argv) > 4 else "r01"
    import torch
    from ntire2022_esr_amd.registry import select_model
    m, _, dr, _ = select_model(-1, torch.device("cuda:0"))
    x = torch.rand(32, 3, 256, 256, device="cuda:0") * dr
    m.enable_profiling(1); m(x); torch.cuda.synchronize()
    algo = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for o in m.collect_profile():
        if "read_bytes" in o:
            a = algo[o["kernel"]]; a[0] += o["read_bytes"]; a[1] += o["write_bytes"]; a[2] += 1
    fetch, write = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE")
    res = {"_how": __doc__.split("usage:")[1].strip(), "_round": tag}
    for sym, fkb in fetch.items():
        lab = label(sym)
        if lab is None or lab not in algo:
            continue
        wkb = write.get(sym, 0.0)
        ar, aw, n = algo[lab]
        hbm = 2.0 * fkb * 1024 + wkb * 1024
        res[lab] = {"FETCH_SIZE_KB": round(fkb, 1), "WRITE_SIZE_KB": round(wkb, 1),
                    "algorithmic_read_KB": round(ar / n / 1024), "algorithmic_write_KB": round(aw / n / 1024),
                    "hbm_bytes_per_launch": int(hbm), "algorithmic_bytes_per_launch": int((ar + aw) / n),
                    "traffic_over_algorithmic": round(hbm / ((ar + aw) / n), 3)}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
