#!/usr/bin/env python3
"""profiles/pmc_traffic.json from two rocprofv3 PMC passes of a bench configuration (run on the GPU box).

usage: pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json> <round tag> [bench.py options]
Each pass:  rocprofv3 --kernel-trace --pmc <COUNTER> --output-format csv -d <dir> --  \
            python bench.py --no-cpu-baseline --no-kernel-events --steps 2 --warmup 1 [bench.py options]
Counter units are KB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies 128-byte requests at
64 B, so read bytes = 2 x FETCH_SIZE; WRITE_SIZE is exact.  Infinity-Cache hits are counted, so this is fabric traffic,
an upper bound on HBM traffic.  The algorithmic bytes come from the op list of the model (engine.op_costs).
The file is keyed by workload ("<model>:<compute>:<batch>x<H>x<W>"), then by kernel label; an existing file is updated."""
import argparse
import collections
import csv
import glob
import json
import os
import re
import sys

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
    return {k: (v[0], v[1]) for k, v in acc.items()}        # total, launches


def label(sym, compute):
    """rocprofv3 kernel symbol -> the kernel label of engine.op_costs / bench.py"""
    m = re.search(r"conv_f32_kernel<(\d+), (\d+), (true|false), (\d+), (\d+), (\d+)(?:, (true|false))?>", sym)
    if m:
        base = f"conv_f32_kernel<NT={m.group(1)},KS={m.group(2)},NCHW_IN={int(m.group(3) == 'true')},NW={m.group(4)}"
        if m.group(5) != "0":
            base += f",TAIL={m.group(5)}"
        if m.group(6) != "0":
            base += f",POST={m.group(6)}"
        if m.group(7) == "true":
            base += ",BLK"                  # split store into a channel-blocked out1 (esr_conv_desc.blocked8)
        return base + ">"
    m = re.search(r"imdb_tail_kernel<(true|false)>", sym)
    if m:
        return f"imdb_tail_kernel<FOLD={int(m.group(1) == 'true')}>"
    m = re.search(r"conv_s16_kernel<(\d+), (\d+), (\d+), (true|false), (true|false), (\d+), (\d+)>", sym)
    if m:
        base = f"conv_s16_kernel<NT={m.group(1)},KS={m.group(2)},NW={m.group(3)},{compute}"
        if m.group(6) != "0":
            base += f",POST={m.group(6)}" + (f"+{m.group(7)}" if m.group(7) != "0" else "")
        return base + ">"
    m = re.search(r"bsconv_kernel<(\d+), (\d+), (\d+)>", sym)
    if m:
        return f"bsconv_kernel<NTP={m.group(1)},NTD={m.group(2)}>"
    for k in ("esa_apply", "dwconv3x3_kernel", "conv3x3s2_kernel", "maxpool7s3_kernel"):
        if k in sym:
            return "esa_apply_kernel" if k == "esa_apply" else k
    return None


def main():
    fdir, wdir, out, tag = sys.argv[1:5]
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="imdn_baseline")
    ap.add_argument("--compute", default="f32")
    ap.add_argument("--tile", default="256x256")
    ap.add_argument("--batch", type=int, default=32)
    a, _ = ap.parse_known_args(sys.argv[5:])
    import torch
    import bench
    from ntire2022_esr_amd.registry import select_model
    m, _, dr, _ = select_model(bench.MODELS[a.model][0], torch.device("cuda:0"))
    m.set_compute(a.compute)
    h, w = (int(v) for v in a.tile.split("x"))
    ent = m.prepare((a.batch, 3, h, w), "cuda:0")
    algo = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for o in m.op_costs(ent.plan, ent.arr):
        x = algo[o["kernel"]]
        x[0] += o["read_bytes"]; x[1] += o["write_bytes"]; x[2] += 1
    fetch, write = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE")
    # several device symbols can share a label (the residual / no-residual variants of conv_s16_kernel): totals, then per launch
    tot = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for sym, (fkb, n) in fetch.items():
        lab = label(sym, a.compute)
        if lab is None or lab not in algo:
            continue
        t = tot[lab]
        t[0] += fkb; t[1] += write.get(sym, (0.0, 0))[0]; t[2] += n
    res = json.load(open(out)) if os.path.exists(out) else {}
    res["_how"] = __doc__.split("usage:")[1].strip()
    key = f"{a.model}:{a.compute}:{a.batch}x{h}x{w}"
    cur = {"_round": tag}
    for lab, (fkb, wkb, n) in tot.items():
        ar, aw, na = algo[lab]
        hbm = (2.0 * fkb + wkb) * 1024 / n
        alg = (ar + aw) / na
        cur[lab] = {"FETCH_SIZE_KB": round(fkb / n, 1), "WRITE_SIZE_KB": round(wkb / n, 1),
                    "algorithmic_read_KB": round(ar / na / 1024), "algorithmic_write_KB": round(aw / na / 1024),
                    "hbm_bytes_per_launch": int(hbm), "algorithmic_bytes_per_launch": int(alg),
                    "traffic_over_algorithmic": round(hbm / alg, 3), "launches_sampled": n}
    res[key] = cur
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({key: cur}, indent=1))


if __name__ == "__main__":
    main()
