// Launch-sequence executor: enqueues a pre-built list of launches with one call (include/msau_hip.h).
// This is the native side of msau_amd/plan.py: the plan builds the op records once per shape, a training
// step is then a handful of C calls instead of ~450 Python -> ctypes round trips.
#include "msau_common.h"

#include <cstdlib>
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
    // Two pools of timing-disabled events, reused across calls: with and without the system-scope fence HIP attaches to an event
    // record (a write-back + invalidate for the host and for peers; HIP documents hipEventDisableSystemFence for timing-only events).
    //   FORK events (main -> side, ~20 per backward sweep: the side launches read what main-stream launches wrote) are created WITHOUT
    //   it: the consumer's visibility of the producer's writes then rests on the agent-scope release / acquire of the dispatch packets
    //   themselves -- the same mechanism that makes two consecutive launches of ONE stream see each other's data across the 8 XCDs --
    //   and it is checked, not assumed: msau_fork_visibility_check (tests/test_host... -m gpu: 0 stale words in thousands of forks
    //   at 64 KB .. 256 MB), the bit-reproducibility tests of the whole step beside the side stream, tools/det_check.py.  Cost of the
    //   fence: 2.99 -> 2.956 ms/step (round 5, interleaved A/B).
    //   JOIN events (side -> main: the slabs before clip + Adam, the forward's attention branch) and everything on the comm stream
    //   (RCCL's kernels feed the fabric) keep the default, fenced form: there are three or four per step.
    //   MSAU_EVENT_FENCE=1 fences the forks as well.
    static thread_local std::vector<hipEvent_t> pool, pool_sys;
    hipStream_t ms = static_cast<hipStream_t>(stream), ss = static_cast<hipStream_t>(side_stream);
    size_t used = 0, used_sys = 0;
    static const bool fence_all = std::getenv("MSAU_EVENT_FENCE") && std::getenv("MSAU_EVENT_FENCE")[0] == '1';
    auto next_event_of = [&](hipEvent_t* ev, bool sys) -> int {
        std::vector<hipEvent_t>& pl = sys ? pool_sys : pool;
        size_t& u = sys ? used_sys : used;
        if (u == pl.size()) {
            hipEvent_t e;
            hipError_t err = hipEventCreateWithFlags(&e, hipEventDisableTiming | (sys || fence_all ? 0u : hipEventDisableSystemFence));
            if (err != hipSuccess) return msau_set_error(MSAU_ERR_HIP, "run_ops_overlap: event: %s", hipGetErrorString(err));
            pl.push_back(e);
        }
        *ev = pl[u++];
        return 0;
    };
    auto next_event = [&](hipEvent_t* ev) -> int { return next_event_of(ev, false); };
    // A fork (event record on `stream`, wait on the side stream) is expensive on the main queue: 4.9 us per record on
    // MI355X / ROCm 7.2 against 2 us for a whole tiny kernel (tools/launch_floor.py).  Side ops only need inputs that
    // stay valid for the rest of the sweep, so they are held back and released in batches behind ONE fork: when
    // `fork_every` of them are pending, before a slab reduction (it consumes them), before a join, and at the end.
    // (Measured and removed, profiles/HISTORY_r03_r04.md: a second side queue +5-9 %, side launches from a worker thread 0 %,
    //  holding side launches back while the main stream runs bandwidth-bound launches +3 %.)
    static const int fork_every = std::getenv("MSAU_FORK_EVERY") ? atoi(std::getenv("MSAU_FORK_EVERY")) : 6;   // (round 3, after the fused launches: 1: +8 %, 2: +2 %, 3-4: 0, 5-6: -0.7 %, 8: 0, 12: +1 %, 24: +4 %)
    std::vector<std::pair<msau_op, int>> pending;
    bool any_side = false;
    auto flush = [&]() -> int {
        if (pending.empty()) return 0;
        hipEvent_t ev;
        int rc = next_event(&ev);
        if (rc) return rc;
        if (hipEventRecord(ev, ms) != hipSuccess || hipStreamWaitEvent(ss, ev, 0) != hipSuccess)
            return msau_set_error(MSAU_ERR_HIP, "run_ops_overlap: fork failed");
        // weight gradients of one shape released together share a grid (msau_conv2d_wgrad_group)
        static const bool group_off = std::getenv("MSAU_WGRAD_GROUP") && std::getenv("MSAU_WGRAD_GROUP")[0] == '0';
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
                if (n > 1) {
                    rc = msau_conv2d_wgrad_group(side_stream, o.dtype, ds, n);
                    if (rc) return rc;
                    continue;
                }
            }
            rc = run_one(side_stream, o, pending[i].second);
            if (rc) return rc;
        }
        pending.clear();
        any_side = true;
        return 0;
    };
    auto join_side = [&]() -> int {
        hipEvent_t ev;
        int rc = next_event_of(&ev, true);                                 // (joins keep the system-scope fence)
        if (rc) return rc;
        if (hipEventRecord(ev, ss) != hipSuccess || hipStreamWaitEvent(ms, ev, 0) != hipSuccess)
            return msau_set_error(MSAU_ERR_HIP, "run_ops_overlap: join failed");
        return 0;
    };
    hipStream_t cs = static_cast<hipStream_t>(comm_stream);
    bool any_comm = false;
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
            hipEvent_t ev;
            rc = next_event_of(&ev, true);
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
            // (a side-stream attention core closes the forward sweep's attention branch -- f, g, h, core: released at once, it has the
            //  whole decoder of its stage to run beside)
            if ((int)pending.size() >= fork_every || (o.kind & 0xff) == MSAU_OP_WGRAD_REDUCE || (o.kind & 0xff) == MSAU_OP_ATTN_FWD) {
                int rc = flush();
                if (rc) return rc;
            }
            continue;
        }
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
    }
    if (any_comm && join) {
        hipEvent_t ev;
        int rc = next_event_of(&ev, true);
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
            case MSAU_OP_ATTN_PROJ_BWD: rc = msau_attn_proj_bwd(stream, o.dtype, static_cast<const msau_attn_proj_bwd_args*>(o.args)); break;
            case MSAU_OP_DGRAD2_1X1: rc = msau_dgrad2_1x1(stream, o.dtype, static_cast<const msau_dgrad2_args*>(o.args)); break;
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
