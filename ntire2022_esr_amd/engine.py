"""Host-side engine shared by the four networks: weight packing, workspace, op lists.

A network's forward is a flat `esr_op` list (include/esr_hip.h) built once per
(N, H, W, device) and replayed by ONE C call (`esr_run_ops`) on the caller's
current HIP stream -- so the reference's `start.record(); forward(); end.record()`
bracket (test_demo.py:429-432) times exactly the device work, and nothing here
synchronises the host.  PyTorch only provides device memory (caching allocator)
and the stream handle.
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def pack_conv(weight, bias, cin_map=None, cin_phys=None):
    """OIHW (or [out,in] for nn.Linear) fp32 weights + bias -> packed blob (CPU float32 tensor).

    Host-side, no GPU needed.  `cin_map[s]` = logical input channel carried by physical
    slot s, or -1 for a zero pad slot (padded concat buffers)."""
    lib = L.lib()
    w = weight.detach().to("cpu", torch.float32).contiguous()
    if w.dim() == 2:
        w = w[:, :, None, None].contiguous()
    cout, cin, k, k2 = w.shape
    assert k == k2 and k in (1, 3), "only 1x1 / 3x3 kernels are on the hot path"
    b = None if bias is None else bias.detach().to("cpu", torch.float32).contiguous()
    if cin_map is not None:
        cm = np.ascontiguousarray(np.asarray(cin_map, dtype=np.int32))
        cin_phys = len(cm)
        cm_p = cm.ctypes.data_as(ctypes.c_void_p)
    else:
        cm, cm_p = None, None
        cin_phys = cin if cin_phys is None else cin_phys
    nbytes = lib.esr_packed_conv_bytes(cin_phys, cout, k)
    if nbytes == 0:
        raise L.EsrError(f"esr_packed_conv_bytes rejected cin_phys={cin_phys} cout={cout} k={k}")
    out = torch.empty(nbytes // 4, dtype=torch.float32)
    rc = lib.esr_pack_conv_f32(_ptr(w), _ptr(b) if b is not None else None, cin, cout, k,
                               cm_p, cin_phys, _ptr(out), nbytes)
    L.check(rc, "esr_pack_conv_f32")
    return out


def unpack_conv(blob, cin, cout, k, cin_map=None, cin_phys=None):
    lib = L.lib()
    blob = blob.detach().to("cpu", torch.float32).contiguous()
    if cin_map is not None:
        cm = np.ascontiguousarray(np.asarray(cin_map, dtype=np.int32))
        cin_phys, cm_p = len(cm), cm.ctypes.data_as(ctypes.c_void_p)
    else:
        cm_p = None
        cin_phys = cin if cin_phys is None else cin_phys
    w = torch.empty(cout, cin, k, k)
    b = torch.empty(cout)
    rc = lib.esr_unpack_conv_f32(_ptr(blob), blob.numel() * 4, cin, cout, k, cm_p, cin_phys, _ptr(w), _ptr(b))
    L.check(rc, "esr_unpack_conv_f32")
    return w, b


class Buffer:
    """An NHWC fp32 activation buffer inside the workspace: [N*H*W][pitch]."""

    def __init__(self, name, pitch, offset_floats):
        self.name, self.pitch, self.offset = name, pitch, offset_floats

    def __getitem__(self, sl):
        """buf[a:b] -> channel slice view (coff=a, channels=b-a)."""
        a = 0 if sl.start is None else sl.start
        b = self.pitch if sl.stop is None else sl.stop
        return (self, a, b - a)


INPUT = "__input__"
OUTPUT = "__output__"


class Plan:
    """Builds the op list for one (N, H, W); see HipSRModel._build_plan in each network."""

    def __init__(self, n, h, w):
        self.n, self.h, self.w = n, h, w
        self.npix = n * h * w
        self.total = 0
        self.buffers = []
        self.ops = []          # python dicts until finalize()

    def buffer(self, name, pitch):
        assert pitch % 4 == 0
        b = Buffer(name, pitch, self.total)
        self.total += self.npix * pitch
        self.buffers.append(b)
        return b

    def conv(self, wname, src, dst, cin, cout, k=3, act=L.ACT_NONE, slope=0.05,
             res=None, res_mode=L.RES_NONE, dst1=None, split=0):
        """src/dst/res: INPUT | OUTPUT | Buffer | (Buffer, coff, channels)."""
        self.ops.append(dict(w=wname, src=src, dst=dst, dst1=dst1, cin=cin, cout=cout, k=k, act=act,
                             slope=slope, res=res, res_mode=res_mode, split=split))

    @staticmethod
    def _view(v, base_ptr):
        if isinstance(v, Buffer):
            v = (v, 0, v.pitch)
        buf, coff, _ = v
        return L.View(ctypes.c_void_p(base_ptr + buf.offset * 4), buf.pitch, coff)

    def finalize(self, workspace, weights):
        """weights: name -> device blob tensor.  Returns (Op array, input op indices, output op indices)."""
        arr = (L.Op * len(self.ops))()
        in_idx, out_idx = [], []
        base = workspace.data_ptr() if workspace is not None else 0
        for i, o in enumerate(self.ops):
            op = arr[i]
            op.kind = L.OP_CONV
            d = op.conv
            d.n, d.h, d.w = self.n, self.h, self.w
            d.cin, d.cout, d.ksize = o["cin"], o["cout"], o["k"]
            d.act, d.slope, d.res_mode = o["act"], o["slope"], o["res_mode"]
            d.split = o["split"]
            if o["src"] is INPUT:
                d.in_layout = L.NCHW_IN
                in_idx.append(i)
            else:
                d.in_layout = L.NHWC
                d.inp = self._view(o["src"], base)
            if o["dst"] is OUTPUT:
                d.out_layout = L.NCHW_SHUFFLE4
                out_idx.append(i)
            else:
                d.out_layout = L.NHWC
                d.out0 = self._view(o["dst"], base)
            if o["dst1"] is not None:
                d.out1 = self._view(o["dst1"], base)
            if o["res"] is not None:
                d.res = self._view(o["res"], base)
            d.wpacked = ctypes.c_void_p(weights[o["w"]].data_ptr())
        return arr, in_idx, out_idx


class HipSRModel(nn.Module):
    """Base of the drop-in nn.Modules.  Subclasses register reference-compatible parameters
    with `_add_conv` and describe their forward with `_build_plan`."""

    def __init__(self):
        super().__init__()
        self._conv_specs = {}      # path -> (cin, cout, k, cin_map)
        self._packed = None        # path -> device blob
        self._packed_sig = None
        self._plans = {}
        self._prof_passes = 0      # >0: record HIP events around every op (bench roofline leg)
        self._profs = {}

    # -- parameter registration: same key names as the reference state_dict -------------------
    def _add_conv(self, path, cin, cout, k, cin_map=None, linear=False):
        """Create nested containers so that `path + '.weight'` / `path + '.bias'` are the
        state_dict keys (e.g. 'model.1.sub.0.conv1.0').  The leaf is an nn.Conv2d / nn.Linear
        used purely as a parameter holder with the reference's shapes and default init."""
        parts = path.split(".")
        mod = self
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, nn.Module())
            mod = mod._modules[p]
        leaf = nn.Linear(cin, cout) if linear else nn.Conv2d(cin, cout, k, 1, (k - 1) // 2)
        mod.add_module(parts[-1], leaf)
        self._conv_specs[path] = (cin, cout, k, cin_map)

    def _leaf(self, path):
        mod = self
        for p in path.split("."):
            mod = mod._modules[p]
        return mod

    # -- packing ------------------------------------------------------------------------------
    def _signature(self):
        return tuple((p.data_ptr(), p._version, p.device) for p in self.parameters())

    def repack(self, device):
        packed = {}
        for path, (cin, cout, k, cin_map) in self._conv_specs.items():
            leaf = self._leaf(path)
            packed[path] = pack_conv(leaf.weight, leaf.bias, cin_map=cin_map).to(device)
        self._extra_pack(packed, device)
        self._packed = packed
        self._packed_sig = self._signature()
        self._plans.clear()

    def _extra_pack(self, packed, device):
        pass

    # -- forward ------------------------------------------------------------------------------
    def _build_plan(self, plan):
        raise NotImplementedError

    def forward(self, x):
        if not x.is_cuda:
            raise L.EsrError(f"{type(self).__name__}: input is on {x.device}; this engine only runs on an "
                             "MI355X through libesr_hip.so and has no CPU fallback (use oracle/ for CPU checks)")
        if x.dtype != torch.float32 or x.dim() != 4:
            raise L.EsrError("expected a 4-D float32 NCHW tensor (uint2tensor4 output)")
        lib = L.lib()
        x = x.contiguous()
        if self._packed is None or self._packed_sig != self._signature() or \
                next(iter(self._packed.values())).device != x.device:
            self.repack(x.device)
        n, c, h, w = x.shape
        key = (n, c, h, w, x.device)
        ent = self._plans.get(key)
        if ent is None:
            plan = Plan(n, h, w)
            self._build_plan(plan, c)
            ws = torch.empty(max(plan.total, 4), dtype=torch.float32, device=x.device)
            arr, in_idx, out_idx = plan.finalize(ws, self._packed)
            ent = (arr, in_idx, out_idx, ws, plan)
            if len(self._plans) > 8:
                self._plans.clear()
            self._plans[key] = ent
        arr, in_idx, out_idx, ws, plan = ent
        y = torch.empty((n, self.out_nc, h * self.upscale, w * self.upscale), dtype=torch.float32, device=x.device)
        for i in in_idx:
            arr[i].conv.inp.ptr = x.data_ptr()
        for i in out_idx:
            arr[i].conv.out0.ptr = y.data_ptr()
        stream = torch.cuda.current_stream(x.device).cuda_stream
        if self._prof_passes > 0:
            prof = self._profs.get(key)
            if prof is None:
                prof = ctypes.c_void_p()
                L.check(lib.esr_prof_create(len(arr), self._prof_passes, ctypes.byref(prof)), "esr_prof_create")
                self._profs[key] = prof
            rc = lib.esr_run_ops_profiled(arr, len(arr), ctypes.c_void_p(stream), prof)
        else:
            rc = lib.esr_run_ops(arr, len(arr), ctypes.c_void_p(stream))
        L.check(rc, f"{type(self).__name__}.forward")
        return y

    # -- per-kernel timing (HIP events on the launch stream; see esr_run_ops_profiled) ---------
    def enable_profiling(self, max_passes):
        self.disable_profiling()
        self._prof_passes = int(max_passes)

    def disable_profiling(self):
        for prof in self._profs.values():
            L.lib().esr_prof_destroy(prof)
        self._profs = {}
        self._prof_passes = 0

    def collect_profile(self):
        """After a device synchronise: list of dicts {name, kernel, flops, ms_sum, passes} per op, summed
        over the recorded passes of every cached shape."""
        out = []
        for key, prof in self._profs.items():
            arr, _, _, _, plan = self._plans[key]
            n = len(arr)
            ms = (ctypes.c_double * n)()
            passes = ctypes.c_int(0)
            L.check(L.lib().esr_prof_collect(prof, ms, n, ctypes.byref(passes)), "esr_prof_collect")
            for i, o in enumerate(plan.ops):
                nt = (o["cout"] + 15) // 16
                kern = f"conv_f32_kernel<NT={nt},KS={o['k']},NCHW_IN={int(o['src'] is INPUT)}>"
                out.append(dict(name=o["w"], kernel=kern, cin=o["cin"], cout=o["cout"], k=o["k"],
                                flops=2.0 * plan.npix * o["cin"] * o["cout"] * o["k"] * o["k"],
                                ms_sum=ms[i], passes=passes.value))
        return out

    def workspace_bytes(self, n, h, w, c=3):
        plan = Plan(n, h, w)
        self._build_plan(plan, c)
        return plan.total * 4
