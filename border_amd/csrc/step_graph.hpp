// Replaying a launch-bound opt step from a hipGraph.
//
// The SAC and Mlp-DQN steps are chains of 20-70 kernels of 2-8 us: the host needs 4-5 us per launch (hipLaunchKernel + argument
// marshalling), which is what bounds those configurations (LAB.md 6).  The step's launch sequence is the same every time - only
// a handful of by-value arguments change (Adam's bias corrections, the RNG counters, the replay stream position) - so it is
// captured once into a hipGraph and replayed with ONE hipGraphLaunch; the varying arguments are patched into their kernel nodes
// (hipGraphExecKernelNodeSetParams) right before the launch.
//
// Every launch of such a step goes through bdr::step_launch().  The agent runs its normal C++ enqueue sequence in one of
// three modes:
//   EAGER    plain launches (parity entry points, profiling, prioritized replay, anything the graph does not cover)
//   CAPTURE  the same launches inside hipStreamBeginCapture / EndCapture; the function of every kernel node is remembered
//   REPLAY   NO launches: the sequence only redoes its host bookkeeping (counters, stream positions) and, for launches marked
//            `varying`, patches the freshly built arguments into node #cursor
// so graph and eager execution are the same code path and produce the same bits; a sequence that launches a different number of
// kernels than the captured one (batch size changed, buffers re-allocated...) is detected and re-captured.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <tuple>
#include <vector>

#include "common.hpp"

namespace bdr {

struct StepGraph {
    enum Mode { EAGER = 0, CAPTURE = 1, REPLAY = 2 };
    Mode mode = EAGER;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    std::vector<hipGraphNode_t> nodes;   // kernel nodes in launch order
    std::vector<void*> funcs;            // their functions, as launched during capture
    size_t cursor = 0;
    bool broken = false;                 // a graph API failed once: this agent stays eager (results are identical)
    int32_t status = BDR_OK;             // first error inside a REPLAY / CAPTURE pass
    long break_at = -2, replays = 0;     // test hook BDR_STEP_GRAPH_BREAK_AT=n: REPLAY pass number n is treated as diverged (-2: env not read yet)
    // what the captured sequence depended on
    uint64_t key_a = 0, key_b = 0, key_n = 0;

    void reset()
    {
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        exec = nullptr; graph = nullptr; nodes.clear(); funcs.clear(); cursor = 0;
    }
    ~StepGraph() { reset(); }
};

// When to replay from the graph.  mode 0: never (BDR_NO_STEP_GRAPH=1 / BDR_STEP_GRAPH=0), 1: always (BDR_STEP_GRAPH=1), 2 (default):
// whichever is faster HERE, by measurement.  A graph replay costs the host a few us per opt instead of ~5 us per launch, but ~5 us
// of device time that eager launches do not; which one wins depends on what else the host does between opts (a bare opt loop:
// eager; a trainer loop that steps environments and pushes: the graph).  Both modes are the same enqueue code and the same bits, so
// the policy times them: 64 opts each (device time between two events on the agent's stream, read back only when ready - no
// synchronisation), then 4 096 opts in the faster one, then again.
struct StepGraphPolicy {
    int mode = 2;
    // adaptive state
    int cur = 0;                       // 0 eager, 1 graph
    int phase = 0;                     // 0 settle, 1 window running, 2 waiting for the end event, 3 exploiting
    int count = 0, measured = 0;       // measured: bit m set once mode m has a time
    float t_ms[2] = {0.f, 0.f};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    static constexpr int SETTLE = 8, WINDOW = 64, EXPLOIT = 4096;

    void from_env()
    {
        { const char* e = getenv("BDR_NO_STEP_GRAPH"); if (e && e[0] == '1') mode = 0; }
        { const char* e = getenv("BDR_STEP_GRAPH"); if (e && (e[0] == '0' || e[0] == '1')) mode = e[0] - '0'; }
    }
    ~StepGraphPolicy() { if (ev0) (void)hipEventDestroy(ev0); if (ev1) (void)hipEventDestroy(ev1); }
    // called on entry of every graphable opt(); -> 1 replay from the graph, 0 launch eagerly, <0 error
    int want(hipStream_t st)
    {
        if (mode != 2) return mode;
        if (!ev0) { if (hipEventCreate(&ev0) != hipSuccess || hipEventCreate(&ev1) != hipSuccess) return -1; }
        switch (phase) {
        case 0:   // let the mode settle (first graph use captures and instantiates)
            if (++count >= SETTLE) { if (hipEventRecord(ev0, st) != hipSuccess) return -1; phase = 1; count = 0; }
            break;
        case 1:
            if (++count >= WINDOW) { if (hipEventRecord(ev1, st) != hipSuccess) return -1; phase = 2; count = 0; }
            break;
        case 2: {
            const hipError_t q = hipEventQuery(ev1);
            if (q == hipErrorNotReady) break;
            if (q != hipSuccess) return -1;
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ev0, ev1) != hipSuccess) return -1;
            t_ms[cur] = ms; measured |= 1 << cur;
            if (measured != 3) { cur ^= 1; phase = 0; count = 0; }                       // now time the other mode
            else { cur = t_ms[1] < t_ms[0] ? 1 : 0; phase = 3; count = 0; }
            break;
        }
        default:
            if (++count >= EXPLOIT) { measured = 0; phase = 0; count = 0; }             // conditions change (the host's other work): measure again
            break;
        }
        return cur;
    }
};

// the pass a thread is currently running (nullptr: eager)
inline StepGraph*& step_graph_current()
{
    static thread_local StepGraph* g = nullptr;
    return g;
}

template <class... KArgs, class... Args>
inline hipError_t step_launch(hipStream_t st, bool varying, void (*kernel)(KArgs...), dim3 grid, dim3 block, Args... args)
{
    StepGraph* g = step_graph_current();
    if (!g || g->mode == StepGraph::EAGER) {
        hipLaunchKernelGGL(kernel, grid, block, 0, st, args...);
        return hipGetLastError();
    }
    if (g->mode == StepGraph::CAPTURE) {
        hipLaunchKernelGGL(kernel, grid, block, 0, st, args...);
        g->funcs.push_back(reinterpret_cast<void*>(kernel));
        g->cursor += 1;
        return hipGetLastError();
    }
    // REPLAY
    const size_t k = g->cursor++;
    if (k >= g->nodes.size() || g->funcs[k] != reinterpret_cast<void*>(kernel)) { g->status = BDR_ERR_INVALID; return hipSuccess; }   // sequence changed: re-capture
    if (!varying) return hipSuccess;
    std::tuple<KArgs...> vals(static_cast<KArgs>(args)...);   // arguments with the kernel's own parameter types
    void* params[sizeof...(KArgs) ? sizeof...(KArgs) : 1];
    size_t i = 0;
    std::apply([&](auto&... v) { ((params[i++] = (void*)&v), ...); }, vals);
    hipKernelNodeParams np{};
    np.func = reinterpret_cast<void*>(kernel);
    np.gridDim = grid; np.blockDim = block; np.sharedMemBytes = 0; np.kernelParams = params; np.extra = nullptr;
    return hipGraphExecKernelNodeSetParams(g->exec, g->nodes[k], &np);
}

// kernel nodes of a captured linear chain, in execution order
inline int32_t step_graph_collect(StepGraph* g)
{
    size_t n = 0;
    BDR_HIP(hipGraphGetNodes(g->graph, nullptr, &n));
    std::vector<hipGraphNode_t> all(n);
    BDR_HIP(hipGraphGetNodes(g->graph, all.data(), &n));
    // walk from the root along the single-successor chain a stream capture produces
    size_t nr = 0;
    BDR_HIP(hipGraphGetRootNodes(g->graph, nullptr, &nr));
    if (nr != 1) return fail(BDR_ERR_HIP, "captured step graph has %zu roots", nr);
    hipGraphNode_t cur = nullptr;
    BDR_HIP(hipGraphGetRootNodes(g->graph, &cur, &nr));
    g->nodes.clear();
    for (size_t step = 0; step < n && cur; ++step) {
        hipGraphNodeType ty;
        BDR_HIP(hipGraphNodeGetType(cur, &ty));
        if (ty == hipGraphNodeTypeKernel) g->nodes.push_back(cur);
        size_t nd = 0;
        BDR_HIP(hipGraphNodeGetDependentNodes(cur, nullptr, &nd));
        if (nd == 0) break;
        if (nd != 1) return fail(BDR_ERR_HIP, "captured step graph is not a chain");
        hipGraphNode_t nxt = nullptr;
        BDR_HIP(hipGraphNodeGetDependentNodes(cur, &nxt, &nd));
        cur = nxt;
    }
    if (g->nodes.size() != g->funcs.size()) return fail(BDR_ERR_HIP, "captured step graph: %zu kernel nodes for %zu launches", g->nodes.size(), g->funcs.size());
    for (size_t k = 0; k < g->nodes.size(); ++k) {
        hipKernelNodeParams np{};
        BDR_HIP(hipGraphKernelNodeGetParams(g->nodes[k], &np));
        if (np.func != g->funcs[k]) return fail(BDR_ERR_HIP, "captured step graph: node %zu is not launch %zu", k, k);
    }
    return BDR_OK;
}

// Runs `enqueue` (the agent's launch sequence on `st`) through the graph: capture + instantiate when there is no graph for this
// key yet, else a REPLAY pass + one hipGraphLaunch.  The key must change whenever a buffer the sequence touches is re-allocated:
// generation counters, not pointers (an allocator hands the same address out again).  `enqueue` must route every launch through
// step_launch and do all allocation / cross-stream waiting BEFORE it is called.
//
// A step is never dropped: `enqueue` advances host state (Adam step counters, RNG stream positions, the replay stream position)
// while it builds its launches, so when a REPLAY pass finds that the sequence no longer matches the captured one - or the capture /
// instantiation fails after the pass has run - `restore` puts that host state back to what it was on entry (the caller snapshots
// it before the call) and the step is enqueued again with plain launches.  The graph is marked broken from then on (same code,
// same bits, eager launches).
template <class F, class R>
inline int32_t step_graph_run(StepGraph* g, hipStream_t st, uint64_t key_a, uint64_t key_b, uint64_t key_n, F&& enqueue, R&& restore)
{
    if (g->broken) return enqueue();
    if (g->exec && (g->key_a != key_a || g->key_b != key_b || g->key_n != key_n)) g->reset();
    StepGraph*& cur = step_graph_current();
    auto eager_again = [&](const char* why) -> int32_t {
        if (getenv("BDR_STEP_GRAPH_DEBUG")) fprintf(stderr, "[border_amd] step graph: %s; this step and all later ones run eagerly\n", why);
        g->reset(); g->broken = true;
        restore();
        return enqueue();
    };
    if (g->exec) {
        if (g->break_at == -2) { const char* e = getenv("BDR_STEP_GRAPH_BREAK_AT"); g->break_at = e ? atol(e) : -1; }
        const bool forced = g->break_at >= 0 && g->replays++ == g->break_at;
        g->mode = StepGraph::REPLAY; g->cursor = 0; g->status = BDR_OK;
        cur = g;
        const int32_t s = enqueue();
        cur = nullptr; g->mode = StepGraph::EAGER;
        BDR_TRY(s);
        if (g->status == BDR_OK && g->cursor == g->nodes.size() && !forced) {
            BDR_HIP(hipGraphLaunch(g->exec, st));
            return BDR_OK;
        }
        // The sequence no longer matches the captured one; nothing of this step has been launched yet.
        return eager_again("the launch sequence changed under a captured graph");
    }
    // capture
    g->reset();
    g->key_a = key_a; g->key_b = key_b; g->key_n = key_n;
    hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { (void)hipGetLastError(); g->broken = true; return enqueue(); }
    g->mode = StepGraph::CAPTURE; g->cursor = 0;
    cur = g;
    const int32_t s = enqueue();
    cur = nullptr; g->mode = StepGraph::EAGER;
    e = hipStreamEndCapture(st, &g->graph);
    if (s != BDR_OK) { g->reset(); restore(); return s; }   // the step itself is in error: nothing was launched, nothing advanced
    if (e != hipSuccess || !g->graph) { (void)hipGetLastError(); return eager_again("hipStreamEndCapture failed"); }
    e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) { (void)hipGetLastError(); g->exec = nullptr; return eager_again("hipGraphInstantiate failed"); }
    const int32_t c = step_graph_collect(g);
    if (getenv("BDR_STEP_GRAPH_DEBUG")) fprintf(stderr, "[border_amd] step graph: %zu kernel nodes captured, collect=%d\n", g->nodes.size(), (int)c);
    if (c != BDR_OK) { g->broken = true; }   // the graph is still valid for THIS step (nothing varies yet); later steps run eagerly
    BDR_HIP(hipGraphLaunch(g->exec, st));
    if (c != BDR_OK) g->reset();
    return BDR_OK;
}

}  // namespace bdr
