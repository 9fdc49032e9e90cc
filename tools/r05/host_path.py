"""host cost of the pieces of one forward (idle queue): python tools/r05/host_path.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ntire2022_esr_amd.registry import select_model
from ntire2022_esr_amd import _lib as L
m, name, dr, _ = select_model(4, torch.device("cuda:0"))
m.set_compute("bf16")
x = torch.rand(1, 3, 339, 510, device="cuda:0") * dr
for _ in range(4): y = m(x)
torch.cuda.synchronize()
def t(f, n=300):
    tot = 0.0
    for _ in range(n):
        t0 = time.perf_counter(); f(); tot += time.perf_counter() - t0
        torch.cuda.synchronize()
    return tot / n * 1e6
lib = L.lib()
key = (1, 3, 339, 510, x.device)
ent = m._plans[key]
st = torch.cuda.current_stream(x.device).cuda_stream
print("model(x)                         %.1f us" % t(lambda: m(x)))
print("m._forward_impl(x)               %.1f us" % t(lambda: m._forward_impl(x)))
print("torch.empty(y)                   %.1f us" % t(lambda: torch.empty((1, 3, 1356, 2040), dtype=torch.float32, device=x.device)))
print("current_stream().cuda_stream     %.1f us" % t(lambda: torch.cuda.current_stream(x.device).cuda_stream))
print("esr_graph_launch (same x, y)     %.1f us" % t(lambda: lib.esr_graph_launch(ent.graph, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), ctypes.c_void_p(st))))
y2 = torch.empty_like(y)
flip = [0]
def alt():
    flip[0] ^= 1
    lib.esr_graph_launch(ent.graph, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p((y2 if flip[0] else y).data_ptr()), ctypes.c_void_p(st))
print("esr_graph_launch (y alternates)  %.1f us" % t(alt))
print("esr_run_ops                      %.1f us" % t(lambda: lib.esr_run_ops(ent.arr, len(ent.arr), ctypes.c_void_p(st))))
print("m._ctx + m._entry                %.1f us" % t(lambda: m._entry(key, m._ctx(x.device))))
print("x.contiguous() + shape           %.1f us" % t(lambda: (x.contiguous(), x.shape)))
