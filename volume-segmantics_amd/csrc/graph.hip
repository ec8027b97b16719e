// hipGraph capture / replay of sequences of library calls (gfx950).
//
// A training step of this network is ~430 kernel launches of 3-60 us on two streams with fork / join events: enqueued one by
// one the host needs 2.5-3 ms per 5 ms step, and any hiccup on the host (or a slower host) shows up as idle GPU time
// (round-1 driver run: 8.7 ms per step against 4.9 ms of GPU critical path).  Captured once and replayed, a step costs the
// host two API calls.
//
//   void* cs = vs_capture_begin();                 // the library's capture stream: pass it as `stream` to the calls to record
//   vs_unet_forward(..., cs); vs_dice_loss_fwd(..., cs); vs_dice_loss_bwd(..., cs); vs_unet_backward_adamw(..., cs);
//   vs_capture_end(&graph);                        // nothing has executed yet
//   vs_graph_launch(graph, stream);                // every replay = that sequence, ordered on `stream`
//
// Recorded calls keep their pointer arguments, so every buffer they touch must stay where it is; the per-step scalars of the
// optimiser live in device memory (vs_adamw_args.hyper, written by vs_train_hyper_set before each replay).  The side stream
// the backward pass forks onto joins the capture through its fork / join events and becomes parallel branches of the graph.
// Replaces the host loop of VolSeg2dTrainer._train_one_batch (vol_seg_2d_trainer.py:419-432).
#include "common.h"
#include "prof.h"

struct vs_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    size_t nodes = 0;
};

namespace {
hipStream_t g_capture_stream = nullptr;
bool g_capturing = false;
}  // namespace

extern "C" void* vs_capture_begin(void) {
    if (g_capturing) { vs_set_error("capture_begin: a capture is already in progress"); return nullptr; }
    if (prof_on()) { vs_set_error("capture_begin: the event profiler is on (vs_profile_enable(0) first)"); return nullptr; }
    if (!g_capture_stream && hipStreamCreateWithFlags(&g_capture_stream, hipStreamNonBlocking) != hipSuccess) {
        vs_set_error("capture_begin: cannot create the capture stream");
        return nullptr;
    }
    // relaxed: other threads of the process (data loaders, the caching allocator) may keep calling the runtime meanwhile
    const hipError_t e = hipStreamBeginCapture(g_capture_stream, hipStreamCaptureModeRelaxed);
    if (e != hipSuccess) { vs_set_error("capture_begin: hipStreamBeginCapture -> %s", hipGetErrorString(e)); return nullptr; }
    g_capturing = true;
    return (void*)g_capture_stream;
}

extern "C" int vs_capture_end(vs_graph_t** out) {
    VS_REQUIRE(out, "capture_end: null out pointer");
    VS_REQUIRE(g_capturing, "capture_end: no capture in progress");
    g_capturing = false;
    hipGraph_t graph = nullptr;
    VS_CHECK_HIP(hipStreamEndCapture(g_capture_stream, &graph));
    VS_REQUIRE(graph, "capture_end: the capture was invalidated (an unjoined stream or an illegal call during capture)");
    vs_graph* g = new vs_graph();
    g->graph = graph;
    hipError_t e = hipGraphGetNodes(graph, nullptr, &g->nodes);
    if (e == hipSuccess) e = hipGraphInstantiate(&g->exec, graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGraphDestroy(graph);
        delete g;
        vs_set_error("capture_end: hipGraphInstantiate -> %s", hipGetErrorString(e));
        return VS_ERR_HIP;
    }
    *out = g;
    return VS_OK;
}

// abandon a capture after a failed call inside it (the partial graph is dropped)
extern "C" int vs_capture_abort(void) {
    if (!g_capturing) return VS_OK;
    g_capturing = false;
    hipGraph_t graph = nullptr;
    (void)hipStreamEndCapture(g_capture_stream, &graph);
    if (graph) (void)hipGraphDestroy(graph);
    (void)hipGetLastError();
    return VS_OK;
}

extern "C" int vs_graph_launch(vs_graph_t* g, void* stream) {
    VS_REQUIRE(g && g->exec, "graph_launch: null graph");
    VS_CHECK_HIP(hipGraphLaunch(g->exec, (hipStream_t)stream));
    return VS_OK;
}

extern "C" int64_t vs_graph_num_nodes(const vs_graph_t* g) { return g ? (int64_t)g->nodes : -1; }

extern "C" void vs_graph_destroy(vs_graph_t* g) {
    if (!g) return;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
}
