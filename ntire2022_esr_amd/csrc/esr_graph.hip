// esr_graph.hip -- a forward's op list as ONE HIP graph launch (ABI v11: esr_graph_create / esr_graph_launch / esr_graph_destroy).
//
// The reference runs one image per forward (test_demo.py:416-433) and esr_run_ops enqueues 20-35 kernels for it: 5-8 us of host time
// per launch, 0.11-0.27 ms per forward from one thread -- as long as an image's GPU time once several streams share the chip.  The
// kernel launches of an op list depend on nothing but the list itself (descriptors, workspace addresses, weights), so they are captured
// ONCE into a graph; a forward then costs two kernel-node parameter updates (the network input x and output y are the only pointers
// that change between calls: they are patched in the captured argument blocks) and one hipGraphLaunch on the caller's stream.
//
// How x / y are found (round 6; ADVICE r05: the round-5 version read 8 bytes at +0 and +32 of EVERY captured node's first argument and
// compared bit patterns -- past the end of an 8-byte scalar argument, and blind to a pointer at any other offset): while esr_graph_create
// captures, every launcher whose kernel can read the network input or write the network output (conv_f32 / wino / conv_s16 families:
// one parameter struct; pack_input_kernel: the pointer is its first argument) reports the launch it has just enqueued through
// esr_graph_note_io(stream, in pointer, its byte offset in the argument block, out pointer, its offset).  The recorder asks the capturing
// stream for the node it has just added (hipStreamGetCaptureInfo_v2: the capture's current dependency set is exactly that node) and keeps
// {node, offset} for the pointers that equal x / y.  The number of recorded places must equal the number of ops that hold x / y
// (counted from the op list), and no op may hold x / y in any other pointer field -- else ESR_ERR_UNSUPPORTED and the caller stays on
// esr_run_ops.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <vector>

#include "esr_internal.h"

// One captured graph, ESR_GRAPH_EXECS executable instances used round-robin.  A parameter patch of an instance and its destruction wait for
// the event recorded behind that instance's last launch (the runtime does not promise that a queued launch has snapshotted its
// kernel arguments); with several instances the host still runs that many forwards ahead of the GPU before such a wait can block.
#define ESR_GRAPH_EXECS 4
struct esr_graph {
    hipGraph_t graph;
    struct Patch { hipGraphNode_t node; hipKernelNodeParams params; size_t offset; bool is_y; };
    std::vector<Patch> patches;
    struct Exec { hipGraphExec_t exec; hipEvent_t last; bool launched; const void* x; void* y; };
    Exec execs[ESR_GRAPH_EXECS];
    int next;
    int n_nodes;
};

namespace {
struct Recorder {
    const void* x;
    const void* y;
    struct Place { hipGraphNode_t node; size_t offset; bool is_y; };
    std::vector<Place> places;
    bool failed;
};
thread_local Recorder* g_rec = nullptr;
}  // namespace

void esr_graph_note_io(hipStream_t st, const void* in_ptr, size_t in_off, const void* out_ptr, size_t out_off)
{
    Recorder* r = g_rec;
    if (!r) return;
    const bool hx = in_ptr && in_ptr == r->x, hy = out_ptr && out_ptr == r->y;
    if (!hx && !hy) return;
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    hipGraph_t graph = nullptr;
    const hipGraphNode_t* deps = nullptr;
    size_t ndeps = 0;
    if (hipStreamGetCaptureInfo_v2(st, &status, &id, &graph, &deps, &ndeps) != hipSuccess || status != hipStreamCaptureStatusActive ||
        ndeps != 1 || !deps) { r->failed = true; return; }
    hipGraphNodeType ty;
    if (hipGraphNodeGetType(deps[0], &ty) != hipSuccess || ty != hipGraphNodeTypeKernel) { r->failed = true; return; }
    if (hx) r->places.push_back({deps[0], in_off, false});
    if (hy) r->places.push_back({deps[0], out_off, true});
}

// instance `slot` of the graph, created on first use.  Which x / y a fresh instance holds is not assumed (x = y = null here): its first
// launch sets every recorded place explicitly.
static int esr_graph_instance(esr_graph* g, int slot)
{
    esr_graph::Exec& ex = g->execs[slot];
    if (ex.exec) return ESR_OK;
    hipError_t e = hipGraphInstantiate(&ex.exec, g->graph, nullptr, nullptr, 0);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ex.last, hipEventDisableTiming);
    if (e != hipSuccess) {
        esr_set_err("hipGraphInstantiate", e);
        if (ex.exec) (void)hipGraphExecDestroy(ex.exec);
        ex.exec = nullptr;
        return ESR_ERR_LAUNCH;
    }
    ex.x = nullptr; ex.y = nullptr; ex.launched = false;
    return ESR_OK;
}

extern "C" {

int esr_graph_create(const esr_op* ops, int n_ops, const void* x, void* y, esr_graph** out)
{
    if (!ops || n_ops <= 0 || !x || !y || !out) return ESR_ERR_BAD_ARG;
    *out = nullptr;
    // the ops that hold x / y where a launcher reports them (conv.in of a convolution / the input packer, conv.out0 of a convolution);
    // x / y anywhere else in the list cannot be patched
    int want_x = 0, want_y = 0;
    for (int i = 0; i < n_ops; ++i) {
        const esr_op& o = ops[i];
        const bool conv = o.kind == ESR_OP_CONV, pack = o.kind == ESR_OP_PACK_INPUT;
        if (conv || pack) {
            want_x += o.conv.in.ptr == x;
            want_y += conv && o.conv.out0.ptr == y;
            const void* others[] = {o.conv.res.ptr, o.conv.out1.ptr, o.conv.tail_cat.ptr, o.conv.post_out.ptr, o.conv.post2_out.ptr, pack ? o.conv.out0.ptr : nullptr};
            for (const void* q : others)
                if (q && (q == x || q == y)) {
                    esr_set_err("esr_graph_create: the network input / output is also a residual / second output of an op", hipErrorInvalidValue);
                    return ESR_ERR_UNSUPPORTED;
                }
        } else if (o.kind == ESR_OP_DWCONV && (o.conv.in.ptr == x || o.conv.out0.ptr == y)) {
            esr_set_err("esr_graph_create: a depthwise op holds the network input / output", hipErrorInvalidValue);
            return ESR_ERR_UNSUPPORTED;
        }
    }
    if (want_x == 0 || want_y == 0) {
        esr_set_err("esr_graph_create: no op of the list holds the network input / output pointer", hipErrorInvalidValue);
        return ESR_ERR_UNSUPPORTED;
    }
    hipStream_t cs = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
    if (e != hipSuccess) { esr_set_err("hipStreamCreateWithFlags (graph capture)", e); return ESR_ERR_LAUNCH; }
    e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { esr_set_err("hipStreamBeginCapture", e); (void)hipStreamDestroy(cs); return ESR_ERR_LAUNCH; }
    Recorder rec{x, y, {}, false};
    g_rec = &rec;
    const int rc = esr_run_ops(ops, n_ops, cs);
    g_rec = nullptr;
    hipGraph_t graph = nullptr;
    e = hipStreamEndCapture(cs, &graph);
    (void)hipStreamDestroy(cs);
    if (rc != ESR_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess || !graph) { esr_set_err("hipStreamEndCapture", e); return ESR_ERR_LAUNCH; }
    esr_graph* g = new esr_graph();
    g->graph = graph; g->n_nodes = 0; g->next = 0;
    for (auto& ex : g->execs) ex = {nullptr, nullptr, false, nullptr, nullptr};
    size_t n = 0;
    (void)hipGraphGetNodes(graph, nullptr, &n);
    g->n_nodes = (int)n;
    int got_x = 0, got_y = 0;
    bool ok = !rec.failed;
    for (const auto& pl : rec.places) {
        hipKernelNodeParams kp;
        memset(&kp, 0, sizeof(kp));
        if (hipGraphKernelNodeGetParams(pl.node, &kp) != hipSuccess || !kp.kernelParams || !kp.kernelParams[0]) { ok = false; break; }
        // the launcher named the offset inside its own argument type, so the read stays inside the block; the value must be the pointer
        const void* v;
        memcpy(&v, static_cast<char*>(kp.kernelParams[0]) + pl.offset, sizeof(v));
        if (v != (pl.is_y ? (const void*)y : x)) { ok = false; break; }
        g->patches.push_back({pl.node, kp, pl.offset, pl.is_y});
        (pl.is_y ? got_y : got_x)++;
    }
    if (!ok || got_x != want_x || got_y != want_y) {
        esr_set_err("esr_graph_create: the captured launches do not account for every op that holds the network input / output", hipErrorInvalidValue);
        (void)hipGraphDestroy(graph);
        delete g;
        return ESR_ERR_UNSUPPORTED;
    }
    const int irc = esr_graph_instance(g, 0);
    if (irc != ESR_OK) { (void)hipGraphDestroy(graph); delete g; return irc; }
    *out = g;
    return ESR_OK;
}

int esr_graph_launch(esr_graph* g, const void* x, void* y, void* hip_stream)
{
    if (!g || !x || !y) return ESR_ERR_BAD_ARG;
    const int slot = g->next;
    g->next = (slot + 1) % ESR_GRAPH_EXECS;
    const int irc = esr_graph_instance(g, slot);
    if (irc != ESR_OK) return irc;
    esr_graph::Exec& ex = g->execs[slot];
    if (x != ex.x || y != ex.y) {
        // this instance's previous launch (ESR_GRAPH_EXECS forwards ago) may still be queued: its kernel arguments must not change under it
        if (ex.launched && hipEventQuery(ex.last) != hipSuccess) (void)hipEventSynchronize(ex.last);
        // the host-side argument blocks are shared by the instances: write BOTH pointers everywhere first (a node that holds x and y
        // must not keep another instance's other pointer), then hand every patched node to this instance
        for (auto& pt : g->patches) {
            const void* want = pt.is_y ? y : x;
            memcpy(static_cast<char*>(pt.params.kernelParams[0]) + pt.offset, &want, sizeof(want));
        }
        for (auto& pt : g->patches) {
            const hipError_t e = hipGraphExecKernelNodeSetParams(ex.exec, pt.node, &pt.params);
            if (e != hipSuccess) { esr_set_err("hipGraphExecKernelNodeSetParams", e); return ESR_ERR_LAUNCH; }
        }
        ex.x = x; ex.y = y;
    }
    const hipError_t e = hipGraphLaunch(ex.exec, static_cast<hipStream_t>(hip_stream));
    if (e != hipSuccess) { esr_set_err("hipGraphLaunch", e); return ESR_ERR_LAUNCH; }
    (void)hipEventRecord(ex.last, static_cast<hipStream_t>(hip_stream));
    ex.launched = true;
    return ESR_OK;
}

int esr_graph_nodes(const esr_graph* g) { return g ? g->n_nodes : 0; }

void esr_graph_destroy(esr_graph* g)
{
    if (!g) return;
    for (auto& ex : g->execs) {
        if (ex.launched) (void)hipEventSynchronize(ex.last);      // launches of this instance still in a queue finish first
        if (ex.last) (void)hipEventDestroy(ex.last);
        if (ex.exec) (void)hipGraphExecDestroy(ex.exec);
    }
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
}

}  // extern "C"
