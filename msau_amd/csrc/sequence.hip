// Launch-sequence executor: enqueues a pre-built list of launches with one call (include/msau_hip.h).
// This is the native side of msau_amd/plan.py: the plan builds the op records once per shape, a training
// step is then a handful of C calls instead of ~450 Python -> ctypes round trips.
#include "msau_common.h"

#include <vector>

static int run_one(void* stream, const msau_op& o, int i);

extern "C" int msau_run_ops(void* stream, const msau_op* ops, int n) {
    MSAU_CHECK_ARG(ops || n == 0, "run_ops: null list");
    for (int i = 0; i < n; ++i) {
        int rc = run_one(stream, ops[i], i);
        if (rc) return rc;              // msau_last_error() holds the failing launch's message
    }
    return 0;
}

extern "C" int msau_run_ops_overlap(void* stream, void* side_stream, const msau_op* ops, int n, int join) {
    MSAU_CHECK_ARG((ops || n == 0) && side_stream && side_stream != stream, "run_ops_overlap: bad args");
    static thread_local std::vector<hipEvent_t> pool;          // timing-disabled events, reused across calls
    hipStream_t ms = static_cast<hipStream_t>(stream), ss = static_cast<hipStream_t>(side_stream);
    size_t used = 0;
    auto next_event = [&](hipEvent_t* ev) -> int {
        if (used == pool.size()) {
            hipEvent_t e;
            hipError_t err = hipEventCreateWithFlags(&e, hipEventDisableTiming);
            if (err != hipSuccess) return msau_set_error(MSAU_ERR_HIP, "run_ops_overlap: event: %s", hipGetErrorString(err));
            pool.push_back(e);
        }
        *ev = pool[used++];
        return 0;
    };
    bool main_dirty = true, any_side = false;                   // main has work the side stream has not waited for yet
    for (int i = 0; i < n; ++i) {
        msau_op o = ops[i];
        const bool side = o.kind & MSAU_OP_SIDE;
        o.kind &= ~MSAU_OP_SIDE;
        if (side) {
            if (main_dirty) {
                hipEvent_t ev;
                int rc = next_event(&ev);
                if (rc) return rc;
                if (hipEventRecord(ev, ms) != hipSuccess || hipStreamWaitEvent(ss, ev, 0) != hipSuccess)
                    return msau_set_error(MSAU_ERR_HIP, "run_ops_overlap: fork failed");
                main_dirty = false;
            }
            any_side = true;
        } else {
            main_dirty = true;
        }
        int rc = run_one(side ? side_stream : stream, o, i);
        if (rc) return rc;
    }
    if (any_side && join) {
        hipEvent_t ev;
        int rc = next_event(&ev);
        if (rc) return rc;
        if (hipEventRecord(ev, ss) != hipSuccess || hipStreamWaitEvent(ms, ev, 0) != hipSuccess)
            return msau_set_error(MSAU_ERR_HIP, "run_ops_overlap: join failed");
    }
    return 0;
}

static int run_one(void* stream, const msau_op& o, int i) {
    {
        int rc;
        switch (o.kind) {
            case MSAU_OP_CONV2D: rc = msau_conv2d(stream, o.dtype, static_cast<const msau_conv_desc*>(o.args)); break;
            case MSAU_OP_WGRAD: rc = msau_conv2d_wgrad(stream, o.dtype, static_cast<const msau_wgrad_desc*>(o.args)); break;
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
            default: return msau_set_error(MSAU_ERR_ARG, "run_ops: op %d has unknown kind %d", i, o.kind);
        }
        return rc;
    }
}
