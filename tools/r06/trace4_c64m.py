"""per-block loop times of conv64m_kernel (C64M_TRACE4 builds)"""
import ctypes, sys, torch
sys.path.insert(0, ".")
from ntire2022_esr_amd.registry import select_model
from ntire2022_esr_amd import _lib as L
m = select_model(0, torch.device("cuda:0"))[0]
m.set_compute("bf16")
x = (torch.rand(32, 3, 256, 256) * 255.0).cuda()
for _ in range(3):
    m(x)
torch.cuda.synchronize()
m.enable_profiling(20)
m(x); torch.cuda.synchronize(); m.collect_profile()
for _ in range(20):
    m(x)
torch.cuda.synchronize()
prof = m.collect_profile(); m.disable_profiling()
kms = {}
for o in prof:
    if "conv64m" in o["kernel"]:
        a = kms.setdefault(o["kernel"], [0.0, 0]); a[0] += o["ms_sum"]; a[1] += o["passes"]
buf = (ctypes.c_ulonglong * (2 * 256 * 4))()
assert L.lib().esr_c64m_trace4_read(buf) == 0
for k, kn in ((0, "plain"), (1, "post")):
    rows = [(buf[(k * 256 + b) * 4], buf[(k * 256 + b) * 4 + 1], buf[(k * 256 + b) * 4 + 2]) for b in range(256)]
    tot = max(buf[(k * 256 + b) * 4 + 3] for b in range(256))
    for kn2, (ms, n) in kms.items():
        if ("true>" in kn2) == (k == 1):
            print(f"{kn2}: {ms / n * 1e3:.1f} us per launch by events (incl. ~2 us event pair); slowest block entry -> loop end {tot} ticks => {tot / (ms / n * 1e3) / 1e3:.2f} GHz if the launch were only that")
    loop = sorted(r[0] for r in rows)
    print(kn, "loop ticks per block: min", loop[0], "median", loop[128], "max", loop[-1], "| tiles", rows[0][2], "| tile-end wait+barrier per tile: median", sorted(r[1] / max(r[2], 1) for r in rows)[128], "max", max(r[1] / max(r[2], 1) for r in rows))
    # by tile position (block b -> tile (b & 7) * 32 + (b >> 3) at 256 tiles per image, 16 per row)
    def pos(b):
        t = (b & 7) * 32 + (b >> 3)
        return t // 16, t % 16
    edge = [rows[b][0] for b in range(256) if pos(b)[0] in (0, 15) or pos(b)[1] in (0, 15)]
    inner = [rows[b][0] for b in range(256) if not (pos(b)[0] in (0, 15) or pos(b)[1] in (0, 15))]
    print("    border-tile blocks: mean", sum(edge) // len(edge), "interior blocks: mean", sum(inner) // len(inner))
    for xcd in range(8):
        v = [rows[b][0] for b in range(256) if b % 8 == xcd]
        print("    XCD", xcd, "mean", sum(v) // len(v), "max", max(v))
