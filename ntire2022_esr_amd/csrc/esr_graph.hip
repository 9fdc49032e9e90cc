// esr_graph.hip -- a forward's op list as ONE HIP graph launch (ABI v11: esr_graph_create / esr_graph_launch / esr_graph_destroy).
//
// The reference runs one image per forward (test_demo.py:416-433) and esr_run_ops enqueues 20-35 kernels for it: 5-8 us of host time
// per launch, 0.11-0.27 ms per forward from one thread -- as long as an image's GPU time once several streams share the chip.  The
// kernel launches of an op list depend on nothing but the list itself (descriptors, workspace addresses, weights), so they are captured
// ONCE into a graph; a forward then costs two kernel-node parameter updates (the network input x and output y are the only pointers
// that change between calls: they are patched in the captured argument blocks) and one hipGraphLaunch on the caller's stream.
//
// How x / y are found: every kernel that touches the network input or output takes either a parameter block whose first member is the
// input pointer and whose member at byte 32 is the first output pointer (ConvK, WinoK, S16K: esr_hip.hip, esr_wino.hip, esr_s16.hip), or
// -- pack_input_kernel -- the input pointer as its first scalar argument.  After the capture the nodes' argument blocks are scanned for the
// capture-time values of x and y at exactly those places.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <vector>

#include "esr_internal.h"

struct esr_graph {
    hipGraph_t graph;
    hipGraphExec_t exec;
    struct Patch { hipGraphNode_t node; hipKernelNodeParams params; size_t offset; bool is_y; };
    std::vector<Patch> patches;
    const void* x;          // the values the argument blocks hold now
    void* y;
    int n_nodes;
};

extern "C" {

int esr_graph_create(const esr_op* ops, int n_ops, const void* x, void* y, esr_graph** out)
{
    if (!ops || n_ops <= 0 || !x || !y || !out) return ESR_ERR_BAD_ARG;
    *out = nullptr;
    hipStream_t cs = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
    if (e != hipSuccess) { esr_set_err("hipStreamCreateWithFlags (graph capture)", e); return ESR_ERR_LAUNCH; }
    e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { esr_set_err("hipStreamBeginCapture", e); (void)hipStreamDestroy(cs); return ESR_ERR_LAUNCH; }
    const int rc = esr_run_ops(ops, n_ops, cs);
    hipGraph_t graph = nullptr;
    e = hipStreamEndCapture(cs, &graph);
    (void)hipStreamDestroy(cs);
    if (rc != ESR_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess || !graph) { esr_set_err("hipStreamEndCapture", e); return ESR_ERR_LAUNCH; }
    esr_graph* g = new esr_graph();
    g->graph = graph; g->exec = nullptr; g->x = x; g->y = y; g->n_nodes = 0;
    size_t n = 0;
    (void)hipGraphGetNodes(graph, nullptr, &n);
    std::vector<hipGraphNode_t> nodes(n);
    if (n) (void)hipGraphGetNodes(graph, nodes.data(), &n);
    g->n_nodes = (int)n;
    bool have_x = false, have_y = false;
    for (size_t i = 0; i < n; ++i) {
        hipGraphNodeType ty;
        if (hipGraphNodeGetType(nodes[i], &ty) != hipSuccess || ty != hipGraphNodeTypeKernel) continue;
        hipKernelNodeParams kp;
        memset(&kp, 0, sizeof(kp));
        if (hipGraphKernelNodeGetParams(nodes[i], &kp) != hipSuccess || !kp.kernelParams || !kp.kernelParams[0]) continue;
        // (every kernel of the library takes at least 40 bytes of arguments, laid out contiguously behind the first)
        char* a0 = static_cast<char*>(kp.kernelParams[0]);
        const void *v0, *v32;
        memcpy(&v0, a0, sizeof(v0));
        memcpy(&v32, a0 + 32, sizeof(v32));
        if (v0 == x) { g->patches.push_back({nodes[i], kp, 0, false}); have_x = true; }
        if (v32 == y) { g->patches.push_back({nodes[i], kp, 32, true}); have_y = true; }
    }
    if (!have_x || !have_y) {
        esr_set_err("esr_graph_create: network input / output pointer not found in the captured launches", hipErrorInvalidValue);
        (void)hipGraphDestroy(graph);
        delete g;
        return ESR_ERR_UNSUPPORTED;
    }
    e = hipGraphInstantiate(&g->exec, graph, nullptr, nullptr, 0);
    if (e != hipSuccess) { esr_set_err("hipGraphInstantiate", e); (void)hipGraphDestroy(graph); delete g; return ESR_ERR_LAUNCH; }
    *out = g;
    return ESR_OK;
}

int esr_graph_launch(esr_graph* g, const void* x, void* y, void* hip_stream)
{
    if (!g || !x || !y) return ESR_ERR_BAD_ARG;
    if (x != g->x || y != g->y) {
        for (auto& pt : g->patches) {
            const void* cur = pt.is_y ? g->y : g->x;
            const void* want = pt.is_y ? y : x;
            if (cur == want) continue;
            memcpy(static_cast<char*>(pt.params.kernelParams[0]) + pt.offset, &want, sizeof(want));
            const hipError_t e = hipGraphExecKernelNodeSetParams(g->exec, pt.node, &pt.params);
            if (e != hipSuccess) { esr_set_err("hipGraphExecKernelNodeSetParams", e); return ESR_ERR_LAUNCH; }
        }
        g->x = x; g->y = y;
    }
    const hipError_t e = hipGraphLaunch(g->exec, static_cast<hipStream_t>(hip_stream));
    if (e != hipSuccess) { esr_set_err("hipGraphLaunch", e); return ESR_ERR_LAUNCH; }
    return ESR_OK;
}

int esr_graph_nodes(const esr_graph* g) { return g ? g->n_nodes : 0; }

void esr_graph_destroy(esr_graph* g)
{
    if (!g) return;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
}

}  // extern "C"
