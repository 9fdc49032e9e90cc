"""-m gpu: esr_graph_create / esr_graph_launch (ABI v11) -- a forward's launches replayed as one HIP graph with the network input / output
pointers patched per call.  Same kernels, same order: outputs must be bit-identical to esr_run_ops, also when many forwards with
DIFFERENT inputs are enqueued back to back without a host synchronisation (a parameter update must not reach a launch already queued)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(name, compute):
    from ntire2022_esr_amd.registry import select_model
    mid = {"imdn_baseline": -1, "rfdn_baseline": 0, "team04_rlfn": 4, "team18_bsrn": 18}[name]
    m = select_model(mid, torch.device(DEV))[0]
    m.set_compute(compute)
    return m


@pytest.mark.parametrize("name,compute,shape", [("team04_rlfn", "bf16", (1, 3, 48, 64)), ("imdn_baseline", "f32", (1, 3, 40, 56)),
                                                ("rfdn_baseline", "bf16", (2, 3, 33, 47)), ("team18_bsrn", "f16", (1, 3, 36, 60))])
def test_graph_forward_equals_run_ops(name, compute, shape):
    m = _model(name, compute)
    dr = 255.0 if name in ("rfdn_baseline", "team04_rlfn") else 1.0
    g = torch.Generator().manual_seed(5)
    xs = [(torch.rand(*shape, generator=g) * dr).to(DEV) for _ in range(6)]
    m.use_graphs = False
    ref = [m(x).clone() for x in xs]
    torch.cuda.synchronize()
    m.use_graphs = True
    ys = [m(x) for x in xs]                   # back to back, no synchronisation: forwards 2 .. 6 are graph launches with new x / y each
    torch.cuda.synchronize()
    ent = m._plans[(shape[0], shape[1], shape[2], shape[3], torch.device(DEV))]
    assert ent.graph is not None, "the second forward of a shape should have captured a graph"
    from ntire2022_esr_amd import _lib as L
    assert L.lib().esr_graph_nodes(ent.graph) >= len(ent.arr)
    for y, r in zip(ys, ref):
        assert torch.equal(y, r), float((y - r).abs().max())


def test_graphs_on_several_streams_and_after_workspace_growth():
    """one graph per (stream, shape); a larger shape on the same stream grows the workspace: the old graph is dropped and rebuilt"""
    m = _model("team04_rlfn", "bf16")
    g = torch.Generator().manual_seed(9)
    small = [(torch.rand(1, 3, 32, 40, generator=g) * 255.0).to(DEV) for _ in range(4)]
    big = (torch.rand(1, 3, 80, 96, generator=g) * 255.0).to(DEV)
    m.use_graphs = False
    ref_s = [m(x).clone() for x in small]
    ref_b = m(big).clone()
    torch.cuda.synchronize()
    m.use_graphs = True
    streams = [torch.cuda.Stream(DEV) for _ in range(3)]
    outs = []
    for rep in range(3):
        for si, st in enumerate(streams):
            with torch.cuda.stream(st):
                outs.append((si, [m(x) for x in small]))
    with torch.cuda.stream(streams[0]):
        yb1 = m(big)                          # grows stream 0's workspace
        yb2 = m(big)
        ys_again = [m(x) for x in small]      # the small shape again: re-finalised against the new base, graph rebuilt
        ys_again2 = [m(x) for x in small]
    torch.cuda.synchronize()
    for _, ys in outs:
        for y, r in zip(ys, ref_s):
            assert torch.equal(y, r)
    assert torch.equal(yb1, ref_b) and torch.equal(yb2, ref_b)
    for y, r in zip(ys_again + ys_again2, ref_s + ref_s):
        assert torch.equal(y, r)


def test_many_graph_launches_rotate_instances():
    """ESR_GRAPH_EXECS = 4 executable instances per captured graph, used round-robin with x / y re-patched per launch: 14 forwards back to back"""
    m = _model("team04_rlfn", "bf16")
    g = torch.Generator().manual_seed(11)
    xs = [(torch.rand(1, 3, 40, 52, generator=g) * 255.0).to(DEV) for _ in range(14)]
    m.use_graphs = False
    ref = [m(x).clone() for x in xs]
    torch.cuda.synchronize()
    m.use_graphs = True
    ys = [m(x) for x in xs]
    ys2 = [m(x) for x in xs[::-1]]
    torch.cuda.synchronize()
    for y, r in zip(ys + ys2, ref + ref[::-1]):
        assert torch.equal(y, r)


def test_tracers_and_dispatch_modes_see_the_operator():
    """the eager fast path (straight to the C ABI) is only for plain eager calls; torch.jit.trace, make_fx(real) and any active
    TorchDispatchMode must go through esr::sr_forward, or a traced graph bakes the output in as an unwritten constant"""
    from torch.fx.experimental.proxy_tensor import make_fx
    from torch.utils._python_dispatch import TorchDispatchMode
    m = _model("imdn_baseline", "f32")
    x = torch.rand(1, 3, 24, 32, device=DEV)
    want = m(x).clone()

    seen = []

    class Log(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            seen.append(str(func))
            return func(*args, **(kwargs or {}))

    with Log():
        y = m(x)
    assert any("sr_forward" in s for s in seen), seen
    assert torch.equal(y, want)
    gm = make_fx(lambda t: m(t), tracing_mode="real")(x)
    assert any("sr_forward" in str(n.target) for n in gm.graph.nodes), gm.graph
    x2 = torch.rand(1, 3, 24, 32, device=DEV)
    assert torch.equal(gm(x2), m(x2))
    tr = torch.jit.trace(m, x, check_trace=False)
    assert "sr_forward" in str(tr.graph)
    assert torch.equal(tr(x2), m(x2))
