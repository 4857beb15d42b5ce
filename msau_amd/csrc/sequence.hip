// Launch-sequence executor: enqueues a pre-built list of launches with one call (include/msau_hip.h).
// This is the native side of msau_amd/plan.py: the plan builds the op records once per shape, a training
// step is then a handful of C calls instead of ~450 Python -> ctypes round trips.
#include "msau_common.h"

#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

static int run_one_raw(void* stream, const msau_op& o, int i);

// ---- MSAU_OP_PROBE: event pairs around selected launches ------------------------------------------------------
namespace {
struct ProbePair { hipEvent_t a, b; };
thread_local std::vector<ProbePair> g_probe_pool;        // timing-enabled events, reused
thread_local size_t g_probe_used = 0;
}  // namespace

static int run_one(void* stream, const msau_op& o_in, int i) {
    if (!(o_in.kind & MSAU_OP_PROBE)) return run_one_raw(stream, o_in, i);
    msau_op o = o_in;
    o.kind &= ~MSAU_OP_PROBE;
    if (g_probe_used == g_probe_pool.size()) {
        ProbePair p;
        if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess)
            return msau_set_error(MSAU_ERR_HIP, "run_ops: cannot create probe events");
        g_probe_pool.push_back(p);
    }
    ProbePair& p = g_probe_pool[g_probe_used++];
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipEventRecord(p.a, s) != hipSuccess) return msau_set_error(MSAU_ERR_HIP, "run_ops: probe record failed");
    int rc = run_one_raw(stream, o, i);
    if (rc) return rc;
    if (hipEventRecord(p.b, s) != hipSuccess) return msau_set_error(MSAU_ERR_HIP, "run_ops: probe record failed");
    return 0;
}

extern "C" int msau_probe_overhead(void* stream, int reps, float* us) {
    MSAU_CHECK_ARG(us && reps > 0 && reps <= 4096, "probe_overhead: bad args");
    hipStream_t s = static_cast<hipStream_t>(stream);
    std::vector<ProbePair> ev(reps);
    bool ok = true;
    for (auto& p : ev) ok = ok && hipEventCreate(&p.a) == hipSuccess && hipEventCreate(&p.b) == hipSuccess;
    for (auto& p : ev) ok = ok && hipEventRecord(p.a, s) == hipSuccess && hipEventRecord(p.b, s) == hipSuccess;
    double sum = 0.0;
    for (auto& p : ev) {
        float ms = 0.f;
        ok = ok && hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess;
        sum += ms;
    }
    for (auto& p : ev) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    if (!ok) return msau_set_error(MSAU_ERR_HIP, "probe_overhead: HIP event call failed");
    *us = (float)(sum * 1000.0 / reps);
    return 0;
}

extern "C" int msau_probe_read(float* us, int cap, int* n) {
    MSAU_CHECK_ARG(n && (us || cap == 0), "probe_read: null pointer");
    int k = 0;
    for (size_t i = 0; i < g_probe_used; ++i) {
        float ms = 0.f;
        hipError_t e = hipEventSynchronize(g_probe_pool[i].b);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, g_probe_pool[i].a, g_probe_pool[i].b);
        if (e != hipSuccess) { g_probe_used = 0; return msau_set_error(MSAU_ERR_HIP, "probe_read: %s", hipGetErrorString(e)); }
        if (k < cap) us[k++] = ms * 1000.f;
    }
    *n = k;
    g_probe_used = 0;
    return 0;
}

// ---- a second host thread for the side stream's launches -----------------------------------------------------------------
// A sweep of small images (the reference trains batch 1, a shape per document: tools/funsd_loop.py) is bound by the HOST: ~320
// launches at ~4.4 us of hipLaunchKernel each, one thread.  The side stream's ~90 launches need nothing from the calling thread
// but the event they wait for, so a worker thread (one per calling thread, started on first use) enqueues them while the caller
// goes on with the main stream.  Results cannot change: stream order and event dependencies are what they were.  The caller
// drains the worker wherever it needs the side stream's tail (join, comm fork, end of the call).  MSAU_SIDE_THREAD=1 turns it on
// (off by default: see the measurement at its use).
namespace {
struct SideWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv, cv_done;
    std::deque<std::function<int()>> q;
    bool stop = false;
    int pending = 0, rc = 0, dev = 0;
    char err[512] = "";
    void run() {
        (void)hipSetDevice(dev);
        for (;;) {
            std::function<int()> f;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) return;
                f = std::move(q.front());
                q.pop_front();
            }
            const int r = f();
            std::lock_guard<std::mutex> lk(m);
            if (r && !rc) { rc = r; snprintf(err, sizeof(err), "%s", msau_last_error()); }
            if (--pending == 0) cv_done.notify_all();
        }
    }
    void push(std::function<int()> f) {
        std::lock_guard<std::mutex> lk(m);
        q.push_back(std::move(f));
        ++pending;
        cv.notify_one();
    }
    int drain() {
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return pending == 0; });
        const int r = rc;
        rc = 0;
        if (r) return msau_set_error(r, "%s", err);
        return 0;
    }
    ~SideWorker() {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
            cv.notify_one();
        }
        if (th.joinable()) th.join();
    }
};
SideWorker* side_worker() {
    static thread_local SideWorker* w = nullptr;                // leaked on purpose at process exit: no HIP calls from static destructors
    if (!w) {
        w = new SideWorker();
        (void)hipGetDevice(&w->dev);
        w->th = std::thread([p = w] { p->run(); });
    }
    return w;
}
}  // namespace

extern "C" int msau_run_ops(void* stream, const msau_op* ops, int n) {
    MSAU_CHECK_ARG(ops || n == 0, "run_ops: null list");
    for (int i = 0; i < n; ++i) {
        msau_op o = ops[i];
        o.kind &= ~(MSAU_OP_SIDE | MSAU_OP_JOIN | MSAU_OP_COMM);       // one stream: nothing to fork or join
        int rc = run_one(stream, o, i);
        if (rc) return rc;              // msau_last_error() holds the failing launch's message
    }
    return 0;
}

extern "C" int msau_run_ops_overlap(void* stream, void* side_stream, const msau_op* ops, int n, int join) {
    return msau_run_ops_dp(stream, side_stream, nullptr, ops, n, join);
}

extern "C" int msau_run_ops_dp(void* stream, void* side_stream, void* comm_stream, const msau_op* ops, int n, int join) {
    MSAU_CHECK_ARG((ops || n == 0) && side_stream && side_stream != stream && (!comm_stream || (comm_stream != stream && comm_stream != side_stream)),
                   "run_ops_overlap: bad args");
    static thread_local std::vector<hipEvent_t> pool;          // timing-disabled events, reused across calls
    hipStream_t ms = static_cast<hipStream_t>(stream), ss = static_cast<hipStream_t>(side_stream);
    size_t used = 0;
    // Events recorded while the stream is being CAPTURED become part of that graph.  Diagnostic switch for the round-3 finding
    // "destroy a captured graph, capture another: crash" (tools/repro/): MSAU_CAPTURE_FRESH_EVENTS=1 gives every fork / join of a
    // captured sweep an event of its own that is never reused (nor destroyed) instead of one from the pool that eager sweeps and
    // earlier captures have used.
    static const bool fresh_in_capture = std::getenv("MSAU_CAPTURE_FRESH_EVENTS") && std::getenv("MSAU_CAPTURE_FRESH_EVENTS")[0] == '1';
    hipStreamCaptureStatus cap0 = hipStreamCaptureStatusNone;
    if (fresh_in_capture) (void)hipStreamIsCapturing(static_cast<hipStream_t>(stream), &cap0);
    auto next_event = [&](hipEvent_t* ev) -> int {
        if (cap0 != hipStreamCaptureStatusNone) {
            hipError_t err = hipEventCreateWithFlags(ev, hipEventDisableTiming);
            if (err != hipSuccess) return msau_set_error(MSAU_ERR_HIP, "run_ops_overlap: event: %s", hipGetErrorString(err));
            return 0;
        }
        if (used == pool.size()) {
            hipEvent_t e;
            hipError_t err = hipEventCreateWithFlags(&e, hipEventDisableTiming);
            if (err != hipSuccess) return msau_set_error(MSAU_ERR_HIP, "run_ops_overlap: event: %s", hipGetErrorString(err));
            pool.push_back(e);
        }
        *ev = pool[used++];
        return 0;
    };
    // A fork (event record on `stream`, wait on the side stream) is expensive on the main queue: 4.9 us per record on
    // MI355X / ROCm 7.2 against 2 us for a whole tiny kernel (tools/launch_floor.py).  Side ops only need inputs that
    // stay valid for the rest of the sweep, so they are held back and released in batches behind ONE fork: when
    // `fork_every` of them are pending, before a slab reduction (it consumes them), before a join, and at the end.
    static const int fork_every = std::getenv("MSAU_FORK_EVERY") ? atoi(std::getenv("MSAU_FORK_EVERY")) : 6;   // (round 3, after the fused launches: 1: +8 %, 2: +2 %, 3-4: 0, 5-6: -0.7 %, 8: 0, 12: +1 %, 24: +4 %)
    std::vector<std::pair<msau_op, int>> pending;
    bool any_side = false;
    // A second side queue (MSAU_SIDE2, owned by the library): weight gradients are mutually independent and most of their
    // grids (64..384 workgroups) do not fill 256 CUs, so two of them side by side finish sooner than one after the other.
    // Launches of a released batch alternate between the two queues; whatever consumes them (slab reduction, comm fork,
    // join, the end of this call) first orders the second queue into the first, so callers still see ONE side stream.
    static const int side2_on = std::getenv("MSAU_SIDE2") ? atoi(std::getenv("MSAU_SIDE2")) : 0;
    static thread_local hipStream_t side2 = nullptr;
    if (side2_on && !side2 && hipStreamCreateWithFlags(&side2, hipStreamNonBlocking) != hipSuccess)
        return msau_set_error(MSAU_ERR_HIP, "run_ops_overlap: second side stream");
    hipStream_t s2 = side2_on ? side2 : nullptr;
    bool s2_open = false;                                      // work on s2 that ss has not been ordered behind yet
    unsigned turn = 0;
    // measured 2026-10-04: NO gain -- tools/funsd_loop.py 661.9 -> 659.3 docs/s with identical host time per step (1.377 ms), the
    // batch-16 step 3.234 -> 3.241 ms: two threads launching into two streams of one device take turns in the runtime.  Off by default.
    static const int thread_on = std::getenv("MSAU_SIDE_THREAD") ? atoi(std::getenv("MSAU_SIDE_THREAD")) : 0;
    bool any_probe = false;
    for (int i = 0; i < n; ++i) any_probe = any_probe || (ops[i].kind & MSAU_OP_PROBE);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(ms, &cap);                      // a sweep being captured into a graph stays on one thread
    SideWorker* worker = (thread_on && !s2 && !any_probe && cap == hipStreamCaptureStatusNone) ? side_worker() : nullptr;     // (probe events live in the caller's thread)
    auto drain = [&]() -> int { return worker ? worker->drain() : 0; };
    struct DrainGuard { SideWorker* w; ~DrainGuard() { if (w) (void)w->drain(); } } drain_guard{worker};     // every return path: the op list is the caller's again
    auto close_s2 = [&]() -> int {
        if (!s2_open) return 0;
        hipEvent_t ev;
        int rc = next_event(&ev);
        if (rc) return rc;
        if (hipEventRecord(ev, s2) != hipSuccess || hipStreamWaitEvent(ss, ev, 0) != hipSuccess)
            return msau_set_error(MSAU_ERR_HIP, "run_ops_overlap: side join failed");
        s2_open = false;
        return 0;
    };
    auto flush = [&]() -> int {
        if (pending.empty()) return 0;
        hipEvent_t ev;
        int rc = next_event(&ev);
        if (rc) return rc;
        if (hipEventRecord(ev, ms) != hipSuccess) return msau_set_error(MSAU_ERR_HIP, "run_ops_overlap: fork failed");
        // weight gradients of one shape released together share a grid (msau_conv2d_wgrad_group)
        static const bool group_off = std::getenv("MSAU_WGRAD_GROUP") && std::getenv("MSAU_WGRAD_GROUP")[0] == '0';
        if (worker) {
            // the worker thread waits for the fork and enqueues the batch; this thread goes on with the main stream
            worker->push([ss, side_stream, ev, batch = std::move(pending)]() -> int {
                if (hipStreamWaitEvent(ss, ev, 0) != hipSuccess) return msau_set_error(MSAU_ERR_HIP, "run_ops_overlap: fork failed");
                std::vector<char> done(batch.size(), 0);
                for (size_t i = 0; i < batch.size(); ++i) {
                    if (done[i]) continue;
                    const msau_op& o = batch[i].first;
                    if (!group_off && o.kind == MSAU_OP_WGRAD) {
                        const msau_wgrad_desc* ds[4] = {static_cast<const msau_wgrad_desc*>(o.args), nullptr, nullptr, nullptr};
                        int n = 1;
                        for (size_t j = i + 1; j < batch.size() && n < 4; ++j) {
                            const msau_op& p = batch[j].first;
                            if (done[j] || p.kind != MSAU_OP_WGRAD || p.dtype != o.dtype) continue;
                            if (!msau_conv2d_wgrad_groupable(o.dtype, ds[0], static_cast<const msau_wgrad_desc*>(p.args))) continue;
                            ds[n++] = static_cast<const msau_wgrad_desc*>(p.args);
                            done[j] = 1;
                        }
                        if (n > 1) {
                            const int rc = msau_conv2d_wgrad_group(side_stream, o.dtype, ds, n);
                            if (rc) return rc;
                            continue;
                        }
                    }
                    const int rc = run_one(side_stream, o, batch[i].second);
                    if (rc) return rc;
                }
                return 0;
            });
            pending.clear();
            any_side = true;
            return 0;
        }
        if (hipStreamWaitEvent(ss, ev, 0) != hipSuccess || (s2 && hipStreamWaitEvent(s2, ev, 0) != hipSuccess))
            return msau_set_error(MSAU_ERR_HIP, "run_ops_overlap: fork failed");
        std::vector<char> done(pending.size(), 0);
        for (size_t i = 0; i < pending.size(); ++i) {
            if (done[i]) continue;
            const msau_op& o = pending[i].first;
            if (!group_off && o.kind == MSAU_OP_WGRAD) {
                const msau_wgrad_desc* ds[4] = {static_cast<const msau_wgrad_desc*>(o.args), nullptr, nullptr, nullptr};
                int n = 1;
                for (size_t j = i + 1; j < pending.size() && n < 4; ++j) {
                    const msau_op& p = pending[j].first;
                    if (done[j] || p.kind != MSAU_OP_WGRAD || p.dtype != o.dtype) continue;
                    if (!msau_conv2d_wgrad_groupable(o.dtype, ds[0], static_cast<const msau_wgrad_desc*>(p.args))) continue;
                    ds[n++] = static_cast<const msau_wgrad_desc*>(p.args);
                    done[j] = 1;
                }
                void* q = side_stream;
                if (s2 && (turn++ & 1)) { q = s2; s2_open = true; }
                if (n > 1) {
                    rc = msau_conv2d_wgrad_group(q, o.dtype, ds, n);
                    if (rc) return rc;
                    continue;
                }
                rc = run_one(q, o, pending[i].second);
                if (rc) return rc;
                continue;
            }
            // the slab reduction consumes both queues: on the first, behind the second; channel sums alternate like the rest
            void* q = side_stream;
            if ((o.kind & 0xff) == MSAU_OP_WGRAD_REDUCE) { rc = close_s2(); if (rc) return rc; }
            else if (s2 && (turn++ & 1)) { q = s2; s2_open = true; }
            rc = run_one(q, o, pending[i].second);
            if (rc) return rc;
        }
        pending.clear();
        any_side = true;
        return 0;
    };
    auto join_side = [&]() -> int {
        hipEvent_t ev;
        int rc = drain();                                        // everything released so far is enqueued on the side stream
        if (rc) return rc;
        rc = close_s2();
        if (rc) return rc;
        rc = next_event(&ev);
        if (rc) return rc;
        if (hipEventRecord(ev, ss) != hipSuccess || hipStreamWaitEvent(ms, ev, 0) != hipSuccess)
            return msau_set_error(MSAU_ERR_HIP, "run_ops_overlap: join failed");
        return 0;
    };
    hipStream_t cs = static_cast<hipStream_t>(comm_stream);
    bool any_comm = false;
    // MSAU_SIDE_DEFER=1: hold side launches back while the main stream runs bandwidth-bound launches (the level-0 / level-1
    // layers) and release them when it reaches the latency-bound ones (levels 2-3): two bandwidth-bound kernels side by side
    // only slow each other down, a bandwidth-bound kernel beside a latency-bound chain is free.  `heavy` = pixels of the launch.
    static const int defer = std::getenv("MSAU_SIDE_DEFER") ? atoi(std::getenv("MSAU_SIDE_DEFER")) : 0;
    static const long long heavy_px = std::getenv("MSAU_SIDE_HEAVY_PX") ? atoll(std::getenv("MSAU_SIDE_HEAVY_PX")) : 200000;
    auto pixels_of = [](const msau_op& o) -> long long {
        switch (o.kind & 0xff) {
            case MSAU_OP_CONV2D: { const msau_conv_desc* d = static_cast<const msau_conv_desc*>(o.args); return (long long)d->B * d->Hout * d->Wout; }
            case MSAU_OP_CONV_PAIR: { const msau_conv_pair_desc* d = static_cast<const msau_conv_pair_desc*>(o.args); return (long long)d->B * d->H * d->W; }
            case MSAU_OP_LRN_FWD: case MSAU_OP_LRN_BWD: return static_cast<const msau_lrn_args*>(o.args)->npix;
            case MSAU_OP_POOL_FWD: case MSAU_OP_POOL_BWD: { const msau_pool_args* d = static_cast<const msau_pool_args*>(o.args); return (long long)d->B * d->H * d->W; }
            default: return 0;
        }
    };
    bool main_heavy = false;
    for (int i = 0; i < n; ++i) {
        msau_op o = ops[i];
        const bool side = o.kind & MSAU_OP_SIDE;
        const bool join_first = o.kind & MSAU_OP_JOIN;
        const bool comm = o.kind & MSAU_OP_COMM;
        o.kind &= ~(MSAU_OP_SIDE | MSAU_OP_JOIN | MSAU_OP_COMM);
        if (comm) {
            // the exchange of a finished bucket: behind everything released so far on the side stream (the slab reduction
            // that completed it) and on the main stream, on a stream of its own; the sweep goes on
            if (!cs) return msau_set_error(MSAU_ERR_ARG, "run_ops_dp: op %d needs the comm stream", i);
            int rc = flush();
            if (rc) return rc;
            rc = drain();
            if (rc) return rc;
            rc = close_s2();
            if (rc) return rc;
            hipEvent_t ev;
            rc = next_event(&ev);
            if (rc) return rc;
            if (hipEventRecord(ev, any_side ? ss : ms) != hipSuccess || hipStreamWaitEvent(cs, ev, 0) != hipSuccess)
                return msau_set_error(MSAU_ERR_HIP, "run_ops_dp: comm fork failed");
            rc = run_one(comm_stream, o, i);
            if (rc) return rc;
            any_comm = true;
            continue;
        }
        if (side) {
            pending.emplace_back(o, i);
            const bool hold = defer && main_heavy && (int)pending.size() < 96;
            if (!hold && ((int)pending.size() >= fork_every || (o.kind & 0xff) == MSAU_OP_WGRAD_REDUCE)) {
                int rc = flush();
                if (rc) return rc;
            }
            continue;
        }
        if (defer) main_heavy = pixels_of(o) >= heavy_px;
        if (join_first) {
            int rc = flush();
            if (rc) return rc;
            if (any_side) { rc = join_side(); if (rc) return rc; }
        }
        if ((o.kind & 0xff) == MSAU_OP_WGRAD && !pending.empty()) {
            // a weight gradient kept on the main stream (the net's first conv: the tail of the sweep, ~100 us): release the
            // held-back side launches first so that they run beside it instead of after it
            int rc = flush();
            if (rc) return rc;
        }
        int rc = run_one(stream, o, i);
        if (rc) return rc;
    }
    {
        int rc = flush();
        if (rc) return rc;
        rc = drain();                                            // the op list and the descriptors belong to the caller again after this call
        if (rc) return rc;
        rc = close_s2();
        if (rc) return rc;
    }
    if (any_comm && join) {
        hipEvent_t ev;
        int rc = next_event(&ev);
        if (rc) return rc;
        if (hipEventRecord(ev, cs) != hipSuccess || hipStreamWaitEvent(ms, ev, 0) != hipSuccess)
            return msau_set_error(MSAU_ERR_HIP, "run_ops_dp: comm join failed");
    }
    if (any_side && join) return join_side();
    return 0;
}

static int run_one_raw(void* stream, const msau_op& o, int i) {
    {
        int rc;
        switch (o.kind) {
            case MSAU_OP_CONV2D: rc = msau_conv2d(stream, o.dtype, static_cast<const msau_conv_desc*>(o.args)); break;
            case MSAU_OP_WGRAD: rc = msau_conv2d_wgrad(stream, o.dtype, static_cast<const msau_wgrad_desc*>(o.args)); break;
            case MSAU_OP_CONV_PAIR: rc = msau_conv_pair(stream, o.dtype, static_cast<const msau_conv_pair_desc*>(o.args)); break;
            case MSAU_OP_BOX_FWD: rc = msau_box_fwd(stream, o.dtype, static_cast<const msau_box_args*>(o.args)); break;
            case MSAU_OP_BOX_BWD: rc = msau_box_bwd(stream, o.dtype, static_cast<const msau_box_args*>(o.args)); break;
            case MSAU_OP_LRN_FWD: {
                const msau_lrn_args* a = static_cast<const msau_lrn_args*>(o.args);
                rc = msau_lrn_fwd(stream, o.dtype, a->a, a->out, a->npix, a->C, a->Cs, a->n, a->alpha, a->beta, a->k);
                break;
            }
            case MSAU_OP_LRN_BWD: {
                const msau_lrn_args* a = static_cast<const msau_lrn_args*>(o.args);
                rc = msau_lrn_bwd(stream, o.dtype, a->a, a->dy, a->out, a->npix, a->C, a->Cs, a->n, a->alpha, a->beta, a->k);
                break;
            }
            case MSAU_OP_POOL_FWD: {
                const msau_pool_args* a = static_cast<const msau_pool_args*>(o.args);
                rc = msau_maxpool2x2_fwd(stream, o.dtype, a->x_or_dy, a->y_or_dx, a->idx, a->B, a->H, a->W, a->Cs);
                break;
            }
            case MSAU_OP_POOL_BWD: {
                const msau_pool_args* a = static_cast<const msau_pool_args*>(o.args);
                rc = msau_maxpool2x2_bwd(stream, o.dtype, a->x_or_dy, a->idx, a->y_or_dx, a->mask, a->B, a->H, a->W, a->Cs, a->accumulate);
                break;
            }
            case MSAU_OP_ATTN_FWD: {
                const msau_attn_args* a = static_cast<const msau_attn_args*>(o.args);
                rc = msau_selfattn_fwd(stream, o.dtype, a->f, a->g, a->h, a->x_or_dy, a->y, a->stats, a->B, a->N, a->Ds, a->Cs);
                break;
            }
            case MSAU_OP_ATTN_BWD: {
                const msau_attn_args* a = static_cast<const msau_attn_args*>(o.args);
                rc = msau_selfattn_bwd(stream, o.dtype, a->f, a->g, a->h, a->x_or_dy, a->stats, a->df, a->dg, a->dh, a->ws,
                                       a->B, a->N, a->Ds, a->Cs);
                break;
            }
            case MSAU_OP_CHANNEL_SUM: {
                const msau_csum_args* a = static_cast<const msau_csum_args*>(o.args);
                rc = msau_channel_sum(stream, o.dtype, a->g, a->npix, a->Cs, a->partials, a->nblk);
                break;
            }
            case MSAU_OP_WGRAD_REDUCE: {
                const msau_reduce_args* a = static_cast<const msau_reduce_args*>(o.args);
                rc = msau_wgrad_reduce(stream, a->slab_arena, a->flat_grads, a->table_dev, a->n_entries, a->max_elems);
                break;
            }
            case MSAU_OP_ALLREDUCE: {
                const msau_allreduce_args* a = static_cast<const msau_allreduce_args*>(o.args);
                rc = msau_allreduce_bucket(stream, a->comm, a->buf, a->count);
                break;
            }
            default: return msau_set_error(MSAU_ERR_ARG, "run_ops: op %d has unknown kind %d", i, o.kind);
        }
        return rc;
    }
}
