#!/usr/bin/env python3
"""profiles/pmc_traffic.json from two rocprofv3 PMC passes of a bench configuration (run on the GPU box).

usage: pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json> <round tag> [bench.py options]
Each pass:  rocprofv3 --kernel-trace --pmc <COUNTER> --output-format csv -d <dir> --  \
            python bench.py --no-cpu-baseline --no-kernel-events --steps 2 --warmup 1 [bench.py options]
Counter units are KB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies 128-byte requests at
64 B, so read bytes = 2 x FETCH_SIZE; WRITE_SIZE is exact.  Infinity-Cache hits are counted, so this is fabric traffic,
an upper bound on HBM traffic.  The algorithmic bytes come from the op list of the model (engine.op_costs).
The file is keyed by workload ("<model>:<compute>:<batch>x<H>x<W>", or "<model>:<compute>:div2k" for --sizes div2k), then by
device symbol (the spelling of esr_prof_kernel_symbol = rocprofv3's, namespace and argument list dropped); an existing file is updated."""
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


def normalize(sym):
    """rocprofv3 kernel name -> the spelling of esr_prof_kernel_symbol: 'void (anonymous namespace)::wino_f32_kernel<1, 0, 0>((anonymous
    namespace)::WinoK)' -> 'wino_f32_kernel<1, 0, 0>'"""
    m = re.search(r"([A-Za-z_0-9]+_kernel)(<[^()]*?>)?\(", sym)
    if not m:
        m = re.search(r"([A-Za-z_0-9]+_kernel)(<[^()]*?>)?", sym)
    return (m.group(1) + (m.group(2) or "")) if m else None


def main():
    fdir, wdir, out, tag = sys.argv[1:5]
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="imdn_baseline")
    ap.add_argument("--compute", default="f32")
    ap.add_argument("--tile", default="256x256")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--sizes", default="tile")
    a, _ = ap.parse_known_args(sys.argv[5:])
    import torch
    import bench
    from ntire2022_esr_amd.registry import select_model
    m, _, dr, _ = select_model(bench.MODELS[a.model][0], torch.device("cuda:0"))
    m.set_compute(a.compute)
    h, w = (int(v) for v in a.tile.split("x"))
    if a.sizes == "div2k":
        batch, shapes, key = a.batch or 1, bench.DIV2K_LR_SHAPES, f"{a.model}:{a.compute}:div2k"
    else:
        batch, shapes = a.batch or 32, [(h, w)]
        key = f"{a.model}:{a.compute}:{batch}x{h}x{w}"
    # algorithmic bytes per device symbol: one profiled forward per shape of the step (the library names the symbol of every op)
    m.enable_profiling(1)
    with torch.no_grad():
        for sh in shapes:
            m(torch.rand(batch, 3, sh[0], sh[1], device="cuda:0") * dr)
    torch.cuda.synchronize()
    algo = collections.defaultdict(lambda: [0.0, 0.0, 0, 0.0])
    for o in m.collect_profile():
        for sym in o["kernel"].split(" + "):               # (an op lowered to two launches: bytes attributed to the first)
            x = algo[sym]
            x[0] += o["read_bytes"] * o["passes"]; x[1] += o["write_bytes"] * o["passes"]; x[2] += o["passes"]
            x[3] += o.get("stored_bytes", o["read_bytes"] + o["write_bytes"]) * o["passes"]
            break
    m.disable_profiling()
    fetch, write = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE")
    tot = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for sym, (fkb, n) in fetch.items():
        lab = normalize(sym)
        if lab is None or lab not in algo:
            continue
        t = tot[lab]
        t[0] += fkb; t[1] += write.get(sym, (0.0, 0))[0]; t[2] += n
    res = json.load(open(out)) if os.path.exists(out) else {}
    res["_how"] = __doc__.split("usage:")[1].strip()
    cur = {"_round": tag, "_build": bench.library_build()}       # esr_source_hash(): bench.py replays an entry only for the build it was recorded on
    for lab, (fkb, wkb, n) in tot.items():
        ar, aw, na, st = algo[lab]
        hbm = (2.0 * fkb + wkb) * 1024 / n
        alg = (ar + aw) / na
        cur[lab] = {"FETCH_SIZE_KB": round(fkb / n, 1), "WRITE_SIZE_KB": round(wkb / n, 1),
                    "algorithmic_read_KB": round(ar / na / 1024), "algorithmic_write_KB": round(aw / na / 1024),
                    "hbm_bytes_per_launch": int(hbm), "algorithmic_bytes_per_launch": int(alg),
                    "traffic_over_algorithmic": round(hbm / alg, 3),
                    # stored = with the tensors' pad channels (nf = 50 at pitch 64 ...): what the launch has to move as the tensors are laid out
                    "stored_bytes_per_launch": int(st / na), "traffic_over_stored": round(hbm / (st / na), 3), "launches_sampled": n}
    res[key] = cur
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({key: cur}, indent=1))


if __name__ == "__main__":
    main()
