#!/usr/bin/env python3
"""bench.py -- images/s of the x4 SR forward path on N MI355X (default: IMDN fp32, 32 x 256x256 -> 1024x1024).

  python bench.py --gpus N --steps K --warmup W
      N > 1 works both ways: under `python -m torch.distributed.run ... bench.py --gpus N ...` (RANK / LOCAL_RANK /
      WORLD_SIZE / MASTER_* from the environment) and as a plain `python bench.py --gpus N`, which re-executes itself
      under torch.distributed.run on 127.0.0.1 with N ranks.

Workloads (BASELINE.json configs):
  [1] headline          python bench.py                                         IMDN x4 fp32, 32 x 3x256x256 per GPU
  [2] RFDN bf16 DIV2K   python bench.py --model rfdn_baseline --compute bf16 --sizes div2k      B = 1, DIV2K-val LR shapes
  [3] RLFN bf16 DIV2K   python bench.py --model team04_rlfn  --compute bf16 --sizes div2k
  [4] BSRN fp16 tiles   python bench.py --model team18_bsrn  --compute f16 --tile 270x480

A "step" is one pass of the hot path (test_demo.py forward(), :364-367) over one batch of synthetic LR input already
resident in HBM (`--sizes div2k`: one pass over a fixed list of 10 DIV2K-val-shaped LR images, one image per forward like
the reference's loop, test_demo.py:416-433, the forwards spread round-robin over `--streams` HIP streams -- default 8 in
this mode, `--streams 1` = the strictly serial loop; the engine keeps one workspace per stream).  Image-level data parallelism: every rank holds a full replica
and its own inputs, there is no collective inside the timed region ("scaling": "weak"); the only communication is the
MAX-reduction of the elapsed time (and a gather of the ranks that took part).

Prints ONE JSON line on rank 0 with the contract fields plus
  roofline      the kernel symbol with the largest share of the timed kernel time: ALGORITHMIC flops and HBM bytes per
                launch / average launch duration from HIP events recorded on the launch stream during the timed steps
                (DIV2K mode / several streams: during a replay of the same steps on one stream, see `roofline.events`),
                against the dense MFMA peak of its operand type and the 8 TB/s HBM peak; `bound` names the binding one
  cpu_baseline  the reference's CPU path restated (oracle/torch_port.py: the same ATen op sequence) timed on this box's
                host cores on a bounded sample (rank 0, N=1 only)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0                  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (about 6.3 TB/s achievable)
MODELS = {
    # name -> (registry id, data_range, algorithmic GFLOP per 256x256 image: BASELINE.md section 2)
    "imdn_baseline": (-1, 1.0, 116.86),       # BASELINE.json configs[1]: the headline workload
    "rfdn_baseline": (0, 255.0, 54.07),
    "team04_rlfn": (4, 255.0, 39.32),
    "team18_bsrn": (18, 1.0, 18.86),
}
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2516.0, "f16": 2516.0}   # dense MFMA peaks, MI355X_MICROARCH.md
# LR shapes (H, W) of a DIV2K-validation-like mix (x4 bicubic of 2040-wide / 2040-tall 2K photographs): mostly 339x510
DIV2K_LR_SHAPES = [(339, 510), (339, 510), (384, 510), (339, 510), (510, 339), (294, 510), (339, 510), (345, 510),
                   (510, 384), (339, 510)]


NO_HILO_SKIP = False           # --no-hilo-skip: bf16 plans keep the long skip in single bf16 numbers (A/B of the hi + lo pairs)
NO_TIGHT_PITCH = False         # --no-tight-pitch: RFDN's nf-wide 16-bit tensors at pitch 64 instead of 56 (A/B; model.tight_pitch = False)
NO_FUSE_TAIL = False           # --no-fuse-tail: RFDB's c4 / c5 + esa.conv1 as separate launches (A/B of rfdb_tail_kernel; model.fuse_tail = False)
NO_FUSE_CHAIN = False          # --no-fuse-chain: a block's 3x3 chain as separate launches (A/B of esr_conv_chain_s16; model.fuse_chain = False)


def build_model(name, device, compute):
    """Registry model with the exported reference checkpoint (weights/<name>.safetensors)."""
    from ntire2022_esr_amd.registry import select_model
    m, _, _, _ = select_model(MODELS[name][0], device)
    m.set_compute(compute)
    m.hilo_skip = not NO_HILO_SKIP
    m.fuse_chain = not NO_FUSE_CHAIN
    m.fuse_tail = not NO_FUSE_TAIL
    m.tight_pitch = not NO_TIGHT_PITCH
    return m, "checkpoint"


def cpu_baseline(name, shape, budget_s=16.0):
    """The reference's PyTorch CPU path, restated op for op (oracle/torch_port.py), batch 1, fp32, one LR image of the
    bench workload's shape.  SURVEY 8d asks for P = 1 and P = all physical cores; oneDNN
    over-threads badly on one small image (all cores is NOT the fastest setting on a 2x64-core host), so a thread-count
    sweep is timed as well and the reported `value` is the FASTEST setting (the strongest baseline), `cores` its thread
    count; every point of the sweep is in `points_ms`."""
    import torch
    from safetensors.torch import load_file
    from oracle import torch_port as TP
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or os.cpu_count()
    except Exception:
        phys = os.cpu_count()
    sd = load_file(os.path.join(REPO, "weights", name + ".safetensors"))
    dr = MODELS[name][1]
    x = torch.rand(1, 3, shape[0], shape[1], generator=torch.Generator().manual_seed(0)) * dr
    fwd = TP.FORWARD[name]
    cands = sorted({c for c in (8, 16, 32, 64, phys) if c <= phys})
    sweep = {}
    t_start = time.perf_counter()
    with torch.no_grad():
        for c in cands:
            torch.set_num_threads(c)
            fwd(sd, x)
            ts = []
            for _ in range(2):
                t0 = time.perf_counter()
                fwd(sd, x)
                ts.append(time.perf_counter() - t0)
            sweep[c] = min(ts)
        best = min(sweep, key=sweep.get)
        torch.set_num_threads(best)
        fwd(sd, x)
        times = []
        t_end = time.perf_counter() + max(3.0, budget_s * 0.6 - (time.perf_counter() - t_start))
        while (time.perf_counter() < t_end and len(times) < 200) or len(times) < 5:
            t0 = time.perf_counter()
            fwd(sd, x)
            times.append(time.perf_counter() - t0)
        torch.set_num_threads(1)                      # the P = 1 point: one timed forward (seconds of scalar-core work)
        t0 = time.perf_counter()
        fwd(sd, x)
        p1 = time.perf_counter() - t0
    times.sort()
    med = min(times[len(times) // 2], sweep[best])        # the sweep's own point at this thread count counts too: the STRONGEST baseline
    pts = {"1": round(p1 * 1e3, 1)}
    pts.update({str(c): round(v * 1e3, 1) for c, v in sweep.items()})
    return {"value": round(1.0 / med, 3), "unit": "images/s", "cores": int(best), "kind": "port",
            "points_ms": pts, "physical_cores": int(phys),
            "p1_images_per_s": round(1.0 / p1, 3), "pall_images_per_s": round(1.0 / sweep[max(sweep)], 3),
            "sample": f"{len(times)} forwards of 1x3x{shape[0]}x{shape[1]} fp32 (value = min(median of these, sweep point) = {med * 1e3:.1f} ms) with "
                      f"torch.set_num_threads({best}) = fastest of the sweep in points_ms (1 = one core, {max(sweep)} = all "
                      f"{phys} physical cores); oracle/torch_port.py = the reference's ATen op sequence on oneDNN"}


def library_build():
    """esr_source_hash() of the libesr_hip.so this process runs (include/esr_hip.h, ABI v9)"""
    from ntire2022_esr_amd import _lib as L
    return L.lib().esr_source_hash().decode()


_CAL = {}


def calibration(device):
    """Measured on this device, once per process: (a) what per-launch HIP event brackets add to a launch (esr_event_pair_ms: a probe kernel
    launched n times inside one pair and n times with a pair each) -- esr_run_ops_profiled brackets every launch, which adds ~2.5 us
    to a 13 us single-image kernel: the per-launch averages below have it subtracted; (b) the
    read + write rate of a plain copy kernel at a 2 x 16 MiB working set (esr_bw_probe) -- what a streaming kernel can reach on tensors the
    previous launch left in the 256 MB Infinity Cache; single-image launches are priced against THIS, not against 8 TB/s of HBM."""
    import ctypes
    import torch
    from ntire2022_esr_amd import _lib as L
    key = str(device)
    if key not in _CAL:
        lib = L.lib()
        st = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        pair = ctypes.c_double(0.0)
        L.check(lib.esr_event_pair_ms(st, 100, ctypes.byref(pair)), "esr_event_pair_ms")
        buf = torch.empty(2 * (16 << 20), dtype=torch.uint8, device=device)
        gbs = ctypes.c_double(0.0)
        L.check(lib.esr_bw_probe(ctypes.c_void_p(buf.data_ptr()), 16 << 20, 64, st, ctypes.byref(gbs)), "esr_bw_probe")
        # (c) the same copy kernel at a 2 x 1 GiB working set: what "HBM-bound" can mean on this device (4.7-5.0 TB/s of the nominal 8)
        big = ctypes.c_double(0.0)
        try:
            buf2 = torch.empty(2 * (1 << 30), dtype=torch.uint8, device=device)
            for _ in range(2):
                L.check(lib.esr_bw_probe(ctypes.c_void_p(buf2.data_ptr()), 1 << 30, 1, st, ctypes.byref(big)), "esr_bw_probe")
            del buf2
        except Exception:
            big = ctypes.c_double(0.0)
        _CAL[key] = {"event_pair_ms": pair.value, "l3_copy_gbs": gbs.value, "hbm_copy_gbs": big.value}
    return _CAL[key]


L3_RESIDENT_BYTES = 128e6          # a launch whose algorithmic traffic is below this works on Infinity-Cache-resident tensors (256 MB MALL)


def roofline_from_profile(prof, peak, traffic_key, events_desc, cal=None):
    """The roofline object of a JSON line from per-op event times (HipSRModel.collect_profile): dominant kernel symbol = largest share
    of the timed kernel time; ALGORITHMIC flops / bytes per launch over its average launch duration against the dense MFMA peak of its
    operand type and the HBM peak.  Returns (roofline dict, executed flops per step)."""
    by_kernel = {}
    for o in prof:
        by_kernel[o["kernel"]] = by_kernel.get(o["kernel"], 0.0) + o["ms_sum"]
    dom_name = max(by_kernel, key=by_kernel.get)
    dom = [o for o in prof if o["kernel"] == dom_name]
    launches = sum(o["passes"] for o in dom)
    ms = sum(o["ms_sum"] for o in dom)
    flops = sum(o["flops"] * o["passes"] for o in dom) / launches                 # algorithmic = direct-convolution flops
    flops_exec = sum(o["flops_exec"] * o["passes"] for o in dom) / launches       # what the matrix cores execute (Winograd: 16/36)
    nbytes = sum((o["read_bytes"] + o["write_bytes"]) * o["passes"] for o in dom) / launches
    stored = sum(o.get("stored_bytes", o["read_bytes"] + o["write_bytes"]) * o["passes"] for o in dom) / launches
    # the event pair's own time is not the kernel's (15.5 us by events against rocprofv3's 12.8 us)
    pair_ms = cal["event_pair_ms"] if cal else 0.0
    raw_ms = ms
    ms = max(ms - pair_ms * launches, 0.25 * ms)
    total_ms = sum(max(o["ms_sum"] - pair_ms * o["passes"], 0.25 * o["ms_sum"]) for o in prof)
    avg_s = ms / launches * 1e-3
    tflops, gbs = flops_exec / avg_s / 1e12, nbytes / avg_s / 1e9
    tflops_direct = flops / avg_s / 1e12
    # 16-bit modes: conv_s16 / bsconv / esa kernels multiply on v_mfma_f32_16x16x32; the NCHW head and the ESA low-resolution
    # convs (conv_f32_kernel) stay on the fp32 MFMA in every mode
    kpeak = peak if not dom_name.startswith(("conv_f32", "wino_f32", "wino8_f32")) else PEAK_TFLOPS["f32"]
    # the memory roof: HBM (8 TB/s) for launches that stream more than the Infinity Cache keeps; for single-image launches the measured
    # rate of a copy kernel at a cache-resident working set (calibration)
    mem_peak, mem_name = HBM_PEAK_GBS, "hbm"
    if cal and nbytes < L3_RESIDENT_BYTES and cal.get("l3_copy_gbs", 0) > HBM_PEAK_GBS * 0.5:
        mem_peak, mem_name = round(cal["l3_copy_gbs"], 1), "l3"
    f_mfma, f_hbm = tflops / kpeak, gbs / mem_peak
    # PMC traffic: replayed from profiles/pmc_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, tools/pmc_traffic.py)
    # ONLY when it was recorded for the library build that is running (esr_source_hash); a stale file gives traffic = null
    traffic, traffic_src = None, None
    tpath = os.path.join(REPO, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            ent = json.load(open(tpath)).get(traffic_key, {})
            have, want = ent.get("_build"), library_build()
            if ent and have != want:
                traffic_src = (f"profiles/pmc_traffic.json holds {traffic_key} for library build {str(have)[:12]}, this run is build {want[:12]}: "
                               "not replayed (re-collect: tools/profile_r04.sh)")
            else:
                traffic = ent.get(dom_name, {}).get("hbm_bytes_per_launch")
                if traffic is not None:
                    traffic_src = (f"profiles/pmc_traffic.json, {ent.get('_round', '?')}, library build {want[:12]} (= this run's): separate "
                                   "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on an EARLIER run (tools/pmc_traffic.py), "
                                   "not measured in this run")
        except Exception:
            traffic = None
    if f_mfma >= f_hbm:
        roofline = {"bound": "mfma", "kernel": dom_name, "achieved": round(tflops, 2), "peak": kpeak, "unit": "TFLOP/s",
                    "frac": round(f_mfma, 4), "traffic": traffic}
    else:
        roofline = {"bound": mem_name, "kernel": dom_name, "achieved": round(gbs, 1), "peak": mem_peak, "unit": "GB/s",
                    "frac": round(f_hbm, 4), "traffic": traffic}
        if mem_name == "l3":
            roofline["peak_source"] = ("esr_bw_probe: read + write rate of a copy kernel over 2 x 16 MiB repeated inside one launch on this device "
                                       "(tensors of a single-image launch are Infinity-Cache resident; against the 8 TB/s HBM peak the fraction "
                                       f"would be {gbs / HBM_PEAK_GBS:.4f})")
    roofline.update({"traffic_source": traffic_src, "launches": launches, "avg_launch_ms": round(ms / launches, 4),
                     "avg_launch_ms_with_event_pair": round(raw_ms / launches, 4), "event_pair_ms": round(pair_ms, 5),
                     "algorithmic_gflop_per_launch": round(flops / 1e9, 3),
                     "executed_gflop_per_launch": round(flops_exec / 1e9, 3),
                     "direct_equivalent_tflops": round(tflops_direct, 2),
                     # both definitions side by side: frac prices EXECUTED flops (<= 1 by construction), frac_algorithmic
                     # prices SURVEY 8d's algorithmic (direct-convolution) flops over the same time -- above 1 for a Winograd kernel
                     "frac_algorithmic": round(tflops_direct / kpeak, 4),
                     "flops_accounting": ("achieved / frac price the flops the matrix cores EXECUTE (Winograd F(2x2,3x3): 16 products per 2x2 "
                                          "outputs and (cin, cout) where the direct convolution has 36); direct_equivalent_tflops / "
                                          "frac_algorithmic = the algorithmic (direct) flops of SURVEY 8d over the same time"),
                     "algorithmic_mb_per_launch": round(nbytes / 1e6, 2),
                     # the bytes the launch moves with its tensors' pad channels (nf = 50 at pitch 64, ...): stored / algorithmic = padding waste
                     "stored_mb_per_launch": round(stored / 1e6, 2),
                     "frac_of_mfma_peak": round(f_mfma, 4), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4),
                     # the measured read + write rate of a plain copy kernel over 2 x 1 GiB on this device: the practical memory roof
                     "hbm_copy_kernel_gbs": round((cal or {}).get("hbm_copy_gbs", 0.0), 1) or None,
                     "share_of_kernel_time": round(ms / total_ms, 4)})
    # every kernel symbol: share of the step, achieved TFLOP/s and GB/s on algorithmic work
    table = []
    for kn, kms in sorted(by_kernel.items(), key=lambda kv: -kv[1]):
        ops = [o for o in prof if o["kernel"] == kn]
        n = sum(o["passes"] for o in ops)
        fl = sum(o["flops_exec"] * o["passes"] for o in ops)
        fd = sum(o["flops"] * o["passes"] for o in ops)
        by = sum((o["read_bytes"] + o["write_bytes"]) * o["passes"] for o in ops)
        sb = sum(o.get("stored_bytes", o["read_bytes"] + o["write_bytes"]) * o["passes"] for o in ops)
        kms = max(kms - pair_ms * n, 0.25 * kms)
        table.append({"kernel": kn, "share": round(kms / total_ms, 4), "avg_ms": round(kms / n, 4),
                      "tflops": round(fl / (kms * 1e-3) / 1e12, 2), "tflops_direct_equivalent": round(fd / (kms * 1e-3) / 1e12, 2),
                      "gbs": round(by / (kms * 1e-3) / 1e9, 1), "stored_over_algorithmic": round(sb / by, 3) if by else None})
    exec_per_step = sum(o["flops_exec"] * o["passes"] for o in prof) / max(1, max(o["passes"] for o in prof))
    roofline["kernels"] = table
    roofline["events"] = events_desc
    roofline["library_build"] = library_build()
    return roofline, exec_per_step


# What the driver's fixed command also reports: the other BASELINE.json configs, measured right behind the headline's
# timed region in the same process, a second or two each.  (model, compute, sizes, tile, batch, streams)
# HIP streams of the DIV2K mode: 8 = the device's hardware queues (measured 2 / 4 / 6 / 8 / 10 streams: RLFN 3290 / 3400 / 3415 / 3580 / 3245
# images/s, RFDN 1567 / 1577 / 1585 / 1671 / 1561)
DIV2K_STREAMS = 8
OTHER_CONFIGS = [
    ("rfdn_baseline", "bf16", "div2k", None, 1, 1),          # config [2], the reference's strictly serial loop
    ("rfdn_baseline", "bf16", "div2k", None, 1, DIV2K_STREAMS),      # config [2], the forwards spread over HIP streams
    ("team04_rlfn", "bf16", "div2k", None, 1, 1),            # config [3], serial
    ("team04_rlfn", "bf16", "div2k", None, 1, DIV2K_STREAMS),        # config [3], streams
    ("team18_bsrn", "f16", "tile", (270, 480), 32, 1),       # config [4]
]


def measure_other_config(spec, device, budget_s=1.2):
    """One entry of `other_configs`: un-instrumented timed steps (the number that fits ~budget_s after 2 warm-up steps), then a replay of
    a few steps on one stream with a HIP event pair around every launch for the dominant kernel's roofline."""
    import torch
    name, compute, sizes, tile, B, nstreams = spec
    _, dr, gflop256 = MODELS[name]
    model, _ = build_model(name, device, compute)
    shapes = DIV2K_LR_SHAPES if sizes == "div2k" else [tile]
    gen = torch.Generator().manual_seed(0)
    xs = [(torch.rand(B, 3, h, w, generator=gen) * dr).to(device) for h, w in shapes]
    streams = [torch.cuda.Stream(device) for _ in range(nstreams)] if nstreams > 1 else None

    def step(spread=True):
        if streams is None or not spread:
            for x in xs:
                model(x)
        else:
            for k, x in enumerate(xs):
                with torch.cuda.stream(streams[k % nstreams]):
                    model(x)

    with torch.no_grad():
        for _ in range(2):
            step()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize(device)
        one = max(time.perf_counter() - t0, 1e-4)
        steps = int(min(200, max(5, budget_s / one)))
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize(device)
        elapsed = time.perf_counter() - t0
        psteps = 3
        step(False)
        model.enable_profiling(psteps)
        step(False)
        torch.cuda.synchronize(device)
        model.collect_profile()
        for _ in range(psteps):
            step(False)
        torch.cuda.synchronize(device)
        prof = model.collect_profile()
        model.disable_profiling()
    imgs = B * len(xs) * steps
    if sizes == "div2k":
        wl = f"{name} x4 {compute}, DIV2K-val-shaped LR images (10 per step, B = {B} per forward, {nstreams} HIP stream(s))"
        tkey = f"{name}:{compute}:div2k"
    else:
        wl = f"{name} x4 {compute}, {B}x3x{tile[0]}x{tile[1]} LR batch -> {B}x3x{4 * tile[0]}x{4 * tile[1]}"
        tkey = f"{name}:{compute}:{B}x{tile[0]}x{tile[1]}"
    r, _ = roofline_from_profile(prof, PEAK_TFLOPS[compute], tkey, f"replay of {psteps} steps on one stream with a HIP event pair around every launch",
                                 cal=calibration(device))
    del model
    return {"workload": wl, "dtype": compute, "value": round(imgs / elapsed, 2), "unit": "images/s", "steps": steps,
            "ms_per_step": round(elapsed / steps * 1e3, 3), "ms_per_image": round(elapsed / imgs * 1e3, 4), "streams": nstreams,
            "roofline": {k: r[k] for k in ("kernel", "bound", "frac", "frac_algorithmic", "avg_launch_ms", "avg_launch_ms_with_event_pair", "achieved",
                                           "peak", "unit", "traffic", "algorithmic_mb_per_launch", "stored_mb_per_launch", "share_of_kernel_time", "frac_of_mfma_peak",
                                           "frac_of_hbm_peak", "hbm_copy_kernel_gbs") if k in r}}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="LR tiles per GPU per step (default 32; 1 with --sizes div2k)")
    ap.add_argument("--model", default="imdn_baseline", choices=sorted(MODELS))
    ap.add_argument("--compute", default="f32", choices=["f32", "bf16", "f16"],
                    help="f32 (headline) | bf16 | f16 arithmetic of the 16-bit configs (BASELINE.json configs [2]-[4])")
    ap.add_argument("--tile", default="256x256", help="LR tile HxW (config [4]: 270x480)")
    ap.add_argument("--sizes", default="tile", choices=["tile", "div2k"],
                    help="div2k: one step = the 10 DIV2K-val-shaped LR images of DIV2K_LR_SHAPES, one image per forward")
    ap.add_argument("--streams", type=int, default=None,
                    help="HIP streams the forwards of a step are spread over, round-robin (default: 1 for tiles, 8 with --sizes "
                         "div2k).  One image per forward leaves most of the chip idle (352 tiles on 256 CUs, a third of the "
                         "launches latency-bound low-resolution kernels); every stream has its own workspace in the engine, so "
                         "independent images overlap.  --streams 1 is the reference's strictly serial loop")
    ap.add_argument("--dataset", type=int, default=0, metavar="N",
                    help="with --sizes div2k: one step = a synthetic N-image DIV2K-shaped list (shapes cycle through DIV2K_LR_SHAPES) "
                         "SHARDED round-robin over the ranks (image i -> rank i mod W, test_demo.py:416 loop -> dist.shard), per-image "
                         "runtimes gathered once after the timed region (dist.gather_rows): BASELINE.json config [3] with N = 200 "
                         "(DIV2K valid + test).  The total work is fixed, so the line says \"scaling\": \"strong\"")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hilo-skip", action="store_true", help="bf16: single-bf16 long skip instead of hi + lo pairs (A/B; model.hilo_skip = False)")
    ap.add_argument("--no-tight-pitch", action="store_true", help="16-bit RFDN plans: nf-wide tensors at whole K chunks (pitch 64) instead of round_up(nf, 8) = 56 (A/B; model.tight_pitch = False)")
    ap.add_argument("--no-fuse-tail", action="store_true", help="16-bit RFDN plans: c4 and c5 + esa.conv1 as separate launches (A/B; model.fuse_tail = False)")
    ap.add_argument("--no-fuse-chain", action="store_true", help="16-bit plans: the 3x3 chain of a block as separate launches (A/B; model.fuse_chain = False)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default headline run only: skip the `other_configs` leg (BASELINE.json configs [2]-[4] measured behind the timed region)")
    ap.add_argument("--b1-latency", action="store_true",
                    help="also report the latency of a single-image forward (extra launches after the timed region: "
                         "keep it off when the run is profiled, the B=1 launches would enter the per-kernel averages)")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not record per-kernel HIP events during the timed steps")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="rendezvous + timing reduction only (gloo on CPU, NO forward, not a measurement): lets the N > 1 "
                         "launch logic be tested on a box without GPUs")
    return ap.parse_args()


def main():
    args = parse_args()
    global NO_HILO_SKIP, NO_FUSE_CHAIN, NO_FUSE_TAIL, NO_TIGHT_PITCH
    NO_FUSE_TAIL = bool(args.no_fuse_tail)
    NO_TIGHT_PITCH = bool(args.no_tight_pitch)
    NO_HILO_SKIP = bool(args.no_hilo_skip)
    NO_FUSE_CHAIN = bool(args.no_fuse_chain)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: one process per GPU under torch.distributed.run, rendezvous on 127.0.0.1
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.call(cmd, env=env))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    selftest = args.launcher_selftest
    if not selftest:
        assert torch.cuda.is_available(), "bench.py needs an MI355X; the engine has no CPU fallback"
        torch.cuda.set_device(local_rank)
    device = torch.device("cpu") if selftest else torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if selftest:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    def barrier():
        if dist is not None:
            dist.barrier()
        if not selftest:
            torch.cuda.synchronize(device)

    def device_identity():
        """what tells two GPUs of a node apart: UUID (when torch exposes it) and PCI address"""
        if selftest:
            return f"cpu:{socket.gethostname()}:{os.getpid()}"
        pr = torch.cuda.get_device_properties(device)
        uuid = str(getattr(pr, "uuid", ""))
        pci = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))
        return f"{pr.name}|{uuid}|{pci}"

    def reduce_elapsed(elapsed, images=0):
        """MAX over ranks of the elapsed time (all_reduce) + one record per rank gathered over the same process group (RCCL on GPUs):
        rank, its own elapsed time, its images, the identity of its device -- a multi-GPU line proves by itself that N distinct
        devices took part and how evenly they ran."""
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        recs = [{"rank": rank, "elapsed_s": round(elapsed, 6), "images": images, "device": device_identity()}]
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            recs = [None] * world
            dist.all_gather_object(recs, {"rank": rank, "elapsed_s": round(elapsed, 6), "images": images, "device": device_identity()})
            recs.sort(key=lambda r: r["rank"])
        for r in recs:
            r["images_per_s"] = round(r["images"] / r["elapsed_s"], 2) if r["elapsed_s"] > 0 else None
        devs = sorted({r["device"] for r in recs})
        if len(devs) != world:
            raise SystemExit(f"{world} ranks but {len(devs)} distinct devices: {devs}")
        return float(t.item()), recs

    if selftest:
        barrier()
        t0 = time.perf_counter()
        time.sleep(0.01 * (rank + 1))
        barrier()
        elapsed, recs = reduce_elapsed(time.perf_counter() - t0)
        seen = [r["rank"] for r in recs]
        if rank == 0:
            print(json.dumps({"launcher_selftest": True, "n_gpus": world, "ranks_seen": seen, "ranks": recs,
                              "max_elapsed_s": round(elapsed, 4), "note": "no forward was run; not a measurement"}), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    _, dr, gflop256 = MODELS[args.model]
    model, weights = build_model(args.model, device, args.compute)
    peak = PEAK_TFLOPS[args.compute]
    th, tw = (int(v) for v in args.tile.lower().split("x"))
    if args.sizes == "div2k":
        B = args.batch or 1
        shapes = DIV2K_LR_SHAPES
    else:
        B = args.batch or 32
        shapes = [(th, tw)]
    sharded = args.sizes == "div2k" and args.dataset > 0
    gen = torch.Generator().manual_seed(0 if sharded else rank)       # sharded: every rank holds the same image list
    xs = [(torch.rand(B, 3, h, w, generator=gen) * dr).to(device) for h, w in shapes]
    if sharded:
        from ntire2022_esr_amd import dist as D
        my_items = D.shard(args.dataset, rank, world)                 # image i -> rank i mod W
        imgs_per_step = B * len(my_items)
        gflop_per_step = sum(B * gflop256 * (shapes[i % len(shapes)][0] * shapes[i % len(shapes)][1]) / 65536.0 for i in my_items)
    else:
        my_items = list(range(len(xs)))
        imgs_per_step = B * len(xs)
        gflop_per_step = sum(B * gflop256 * (h * w) / 65536.0 for h, w in shapes)
    nstreams = max(1, args.streams if args.streams is not None else (DIV2K_STREAMS if args.sizes == "div2k" else 1))
    streams = [torch.cuda.Stream(device) for _ in range(nstreams)] if nstreams > 1 else None
    if args.sizes == "tile" and nstreams > 1:
        # the batch of a step as `nstreams` sub-batches, one forward each
        if B % nstreams:
            raise SystemExit(f"--streams {nstreams} does not divide the batch {B}")
        xs = [c.contiguous() for c in xs[0].chunk(nstreams)]
        my_items = list(range(len(xs)))           # (imgs_per_step stays B: the same batch, in sub-batches)
    # Per-kernel HIP events bracket every launch of the roofline leg.  Since round 6 the timed region carries NO event pairs in any
    # mode (86 event records per instrumented step were work inside the number); the SAME steps are replayed on one
    # stream with events right behind the timed region (one image per forward with several streams would time overlapping kernels anyway).
    events_after = not args.no_kernel_events

    def step(spread=True):
        y = None
        if streams is None or not spread:
            for i in my_items:
                y = model(xs[i % len(xs)])
        else:
            for k, i in enumerate(my_items):
                with torch.cuda.stream(streams[k % nstreams]):
                    y = model(xs[i % len(xs)])
        return y

    with torch.no_grad():
        for _ in range(args.warmup):
            y = step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            y = step()
        barrier()
        elapsed = time.perf_counter() - t0
        instrumented_ms = None
        if events_after:
            step(False)                    # the default stream's context: plans, workspace
            model.enable_profiling(args.steps)
            step(False)
            torch.cuda.synchronize(device)
            model.collect_profile()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step(False)
            torch.cuda.synchronize(device)
            instrumented_ms = (time.perf_counter() - t1) / args.steps * 1e3
    if my_items:
        hl, wl_ = shapes[my_items[-1] % len(shapes)]
        assert tuple(y.shape) == (xs[my_items[-1] % len(xs)].shape[0], 3, 4 * hl, 4 * wl_)
    elapsed, recs = reduce_elapsed(elapsed, imgs_per_step * args.steps)
    seen = [r["rank"] for r in recs]
    per_image = None
    if sharded:
        # the reference's per-image runtime (test_demo.py:429-433: an event pair around each forward), one extra pass over this
        # rank's shard on ONE stream; rows gathered ONCE over the process group and averaged in index order (dist.gather_rows)
        rows = []
        with torch.no_grad():
            for i in my_items:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                model(xs[i % len(xs)])
                e1.record()
                rows.append((i, e0, e1))
            torch.cuda.synchronize(device)
        rows = [(i, e0.elapsed_time(e1), float("nan"), float("nan")) for i, e0, e1 in rows]
        allrows = D.gather_rows(rows, args.dataset, rank, world, device)
        per_image = {"images": int(args.dataset), "ave_runtime_ms": round(D.ordered_mean(allrows[:, 1]), 4),
                     "max_runtime_ms": round(float(allrows[:, 1].max()), 4), "gathered_rows": int(len(allrows)),
                     "how": "one event pair per forward in an extra pass on one stream (test_demo.py:429-433), rows gathered with one "
                            "all_gather_into_tensor and averaged in index order in float64 (ntire2022_esr_amd/dist.py)"}

    roofline, exec_per_step = None, 0.0
    if not args.no_kernel_events:
        prof = model.collect_profile()
        model.disable_profiling()
        tkey = f"{args.model}:{args.compute}:div2k" if args.sizes == "div2k" else f"{args.model}:{args.compute}:{B}x{th}x{tw}"
        events_desc = (f"HIP event pair around every launch in a replay of the same {args.steps} steps on ONE stream after the "
                       f"timed region ({instrumented_ms:.3f} ms/step with the events; the timed region carries none)")
        roofline, exec_per_step = roofline_from_profile(prof, peak, tkey, events_desc, cal=calibration(device))

    if rank == 0:
        imgs = sum(r["images"] for r in recs)
        value = imgs / elapsed
        if args.sizes == "div2k":
            wl = (f"{args.model} x4 {args.compute}, DIV2K-val-shaped LR images {sorted(set(shapes))} "
                  f"({args.dataset if sharded else len(shapes)} per step{' SHARDED round-robin over the ranks' if sharded else ''}, "
                  f"B = {B} per forward, forwards spread over {nstreams} HIP stream(s))")
            metric = "images/sec (DIV2K-val-shaped LR ~339x510 -> x4)"
        else:
            wl = f"{args.model} x4 {args.compute}, {B}x3x{th}x{tw} LR batch per GPU -> {B}x3x{4 * th}x{4 * tw}"
            if nstreams > 1:
                wl += f" ({nstreams} sub-batches of {B // nstreams}, one HIP stream each)"
            metric = f"images/sec ({th}x{tw}->{4 * th}x{4 * tw} x4)"
        gflop_all = gflop_per_step * world if not sharded else sum(
            B * gflop256 * (shapes[i % len(shapes)][0] * shapes[i % len(shapes)][1]) / 65536.0 for i in range(args.dataset))
        model_tflops = gflop_all * args.steps / elapsed / 1e3       # algorithmic (direct-convolution) flops: SURVEY 8d
        out = {
            "metric": metric,
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None, "dtype": args.compute,
            "data": f"synthetic (uniform [0,{dr:g}) LR tiles resident in HBM; weights: {weights})",
            "config": {"workload": wl, "batch_per_gpu": B, "streams_per_gpu": nstreams,
                       "parallelism": f"image-parallel replicas x{world}",
                       "algorithmic_gflop_per_step_per_gpu": round(gflop_per_step, 2)},
            "ranks_seen": seen,
            "ranks": recs,                      # per rank: its own elapsed time, images, images/s, device identity (N distinct, checked)
            "per_image": per_image,
            "model_tflops": round(model_tflops, 2),                   # direct-equivalent (algorithmic flops / time)
            "model_executed_tflops": None if roofline is None else round(world * exec_per_step / (elapsed / args.steps) / 1e12, 2),
            "model_frac_of_mfma_peak": None if roofline is None else round(exec_per_step / (elapsed / args.steps) / 1e12 / peak, 4),
            "roofline": roofline,
        }
        if world == 1 and args.b1_latency:
            # the reference's own semantics (test_demo.py:416-433: one image per forward), outside the timed region
            with torch.no_grad():
                x1 = xs[0][:1].contiguous()
                for _ in range(5):
                    model(x1)
                torch.cuda.synchronize(device)
                t1 = time.perf_counter()
                for _ in range(50):
                    model(x1)
                torch.cuda.synchronize(device)
            out["b1_latency_ms"] = round((time.perf_counter() - t1) / 50 * 1e3, 3)
        headline = (args.model == "imdn_baseline" and args.compute == "f32" and args.sizes == "tile" and (th, tw) == (256, 256)
                    and args.batch is None and nstreams == 1)
        if world == 1 and headline and not args.no_other_configs and not args.no_kernel_events:
            # BASELINE.json configs [2]-[4] in front of the driver: same process, behind the headline's timed region (not part of `value`)
            del model, xs
            torch.cuda.empty_cache()
            out["other_configs"] = [measure_other_config(spec, device) for spec in OTHER_CONFIGS]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.model, shapes[0])
            out["speedup_vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
