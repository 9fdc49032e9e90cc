"""DIV2K mode (one image per forward) with S model replicas on S HIP streams: images/s for S = 1..4.

A 339x510 image is 352 tiles of 16x32 on 256 CUs and a third of the launches are latency-bound low-resolution
kernels, so a single stream leaves the chip half idle; independent images on other streams fill it.
Usage (GPU box): python tools/dbg/streams_probe.py [model compute]...
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402


def run(name, compute, nstreams, steps=6):
    dev = torch.device("cuda", 0)
    dr = bench.MODELS[name][1]
    models = [bench.build_model(name, dev, compute)[0] for _ in range(nstreams)]
    streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
    gen = torch.Generator().manual_seed(0)
    xs = [(torch.rand(1, 3, h, w, generator=gen) * dr).to(dev) for h, w in bench.DIV2K_LR_SHAPES]
    torch.cuda.synchronize()

    def step():
        for i, x in enumerate(xs):
            s = i % nstreams
            with torch.cuda.stream(streams[s]):
                y = models[s](x)
        return y

    with torch.no_grad():
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        # host-only cost of enqueueing (GPU idle at the start, nothing waited for)
        t1 = time.perf_counter()
        step()
        host = time.perf_counter() - t1
        torch.cuda.synchronize()
    return steps * len(xs) / el, host / len(xs) * 1e3


if __name__ == "__main__":
    pairs = sys.argv[1:] or ["team04_rlfn", "bf16", "rfdn_baseline", "bf16", "team18_bsrn", "f16", "imdn_baseline", "f32"]
    for name, compute in zip(pairs[0::2], pairs[1::2]):
        for s in (1, 2, 4, 6, 8):
            v, host = run(name, compute, s)
            print(f"{name:14s} {compute:5s} streams={s}: {v:8.1f} images/s   host enqueue {host:.3f} ms/image", flush=True)
