// Pointwise (1x1) data gradients of the 64-channel level as single launches: one wave = one 16-pixel tile, weight fragments as 16-byte
// loads straight from the packed data-gradient images msau_conv2d reads (pack.hip), no LDS, no barrier, every load of the tile issued
// before the first MFMA.  The tensors of this level are 2.75 MB at the bench shape: a launch here is its launch latency plus ONE memory
// round trip, so what counts is the number of launches on the dependent chain, not their bytes.
#include "msau_common.h"

// =============================================================================================
// The data gradients of the three 1x1 projections f, g (C -> C/8) and h (C -> C) of the attention block
// (model/layers/attention.py:152-154) in ONE launch: dx = [mask] (Wf^T df + Wg^T dg + Wh^T dh [+ add] [+ dx_old]).
// They were three launches of 5-9 us on a 2.75 MB tensor (two generic 8 -> 64 launches + one lean 64 -> 64), each waiting for
// the previous one's rounded partial sum.  Here: K = 8 + 8 + 64 (+ 16 of zeros) = three k-steps of ONE MFMA chain per
// 16-pixel tile; a wave owns a tile (rows = the 64 channels of x in msau_conv2d's row order: slot ct*16 + 4q + j <-> channel
// 16q + 4ct + j, so a lane ends up with 16 consecutive channels of its pixel), the A fragments are 16-byte loads straight from the
// three packed data-gradient images msau_conv2d reads (pack.hip: [row][kchunk], kchunk 32 / 32 / 64 -- the SAME rounded weights as
// the three launches), the B fragments 16-byte loads of df / dg / dh.  No LDS, no barrier; every load of the tile (12 weight
// fragments, 3 gradient fragments, up to 6 epilogue operands) is issued before the first MFMA: one memory round trip per wave.
// =============================================================================================
namespace {
template <bool ADD, bool ACC, bool MASK>
__global__ __launch_bounds__(256) void attn_proj_bwd_kernel(const bf16_t* __restrict__ df, const bf16_t* __restrict__ dg,
                                                            const bf16_t* __restrict__ dh, const bf16_t* __restrict__ wf,
                                                            const bf16_t* __restrict__ wg, const bf16_t* __restrict__ wh,
                                                            const bf16_t* __restrict__ add, const bf16_t* __restrict__ mask,
                                                            bf16_t* __restrict__ dx, long long npix) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const long long ntiles = (npix + 15) >> 4;
    const long long wstride = (long long)gridDim.x * 4;
    long long tile = (long long)blockIdx.x * 4 + wave;
    if (tile >= ntiles) return;
    // weight fragments: k-step 0 = {f co 0-7, g co 0-7, h co 0-15}, 1 = h co 16-47, 2 = {h co 48-63, zeros}
    bf16x8 A[4][3];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const int row = ct * 16 + lr;
        const bf16_t* p0 = lg == 0 ? wf + row * 32 : lg == 1 ? wg + row * 32 : wh + row * 64 + (lg - 2) * 8;     // (one load, per-lane address)
        A[ct][0] = load8<bf16_t>(p0);
        A[ct][1] = load8<bf16_t>(wh + row * 64 + 16 + lg * 8);
        A[ct][2] = load8<bf16_t>(wh + row * 64 + 48 + (lg & 1) * 8);
        if (lg >= 2) A[ct][2] = zero8<bf16_t>();
    }
    for (; tile < ntiles; tile += wstride) {
        const long long p = tile * 16 + lr;
        const bool live = p < npix;
        const long long pc = live ? p : 0;                       // (dead columns of the last tile read pixel 0 and store nothing)
        const bf16_t* q0 = lg == 0 ? df + pc * 8 : lg == 1 ? dg + pc * 8 : dh + pc * 64 + (lg - 2) * 8;
        bf16x8 b0 = load8<bf16_t>(q0);
        bf16x8 b1 = load8<bf16_t>(dh + pc * 64 + 16 + lg * 8);
        bf16x8 b2 = load8<bf16_t>(dh + pc * 64 + 48 + (lg & 1) * 8);
        if (lg >= 2) b2 = zero8<bf16_t>();
        const long long eo = pc * 64 + lg * 16;                  // the lane's 16 channels of its pixel
        bf16x8 ea[2], ey[2], em[2];
        if (ADD) { ea[0] = load8<bf16_t>(add + eo); ea[1] = load8<bf16_t>(add + eo + 8); }
        if (ACC) { ey[0] = load8<bf16_t>(dx + eo); ey[1] = load8<bf16_t>(dx + eo + 8); }
        if (MASK) { em[0] = load8<bf16_t>(mask + eo); em[1] = load8<bf16_t>(mask + eo + 8); }
        f32x4 acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            acc[ct] = mma8(A[ct][0], b0, f32x4{0.f, 0.f, 0.f, 0.f});
            acc[ct] = mma8(A[ct][1], b1, acc[ct]);
            acc[ct] = mma8(A[ct][2], b2, acc[ct]);
        }
        bf16x8 o[2];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = ct * 4 + j;                        // channel 16 lg + c
                float v = acc[ct][j];
                if (ADD) v += (float)ea[c >> 3][c & 7];
                if (ACC) v += (float)ey[c >> 3][c & 7];
                if (MASK) v = ((float)em[c >> 3][c & 7] > 0.f) ? v : 0.f;
                o[c >> 3][c & 7] = (bf16_t)v;
            }
        if (live) {
            store8<bf16_t>(dx + eo, o[0]);
            store8<bf16_t>(dx + eo + 8, o[1]);
        }
    }
}
}  // namespace

extern "C" int msau_attn_proj_bwd(void* stream, int dtype, const msau_attn_proj_bwd_args* a) {
    MSAU_CHECK_ARG(a && a->df && a->dg && a->dh && a->wf_pack && a->wg_pack && a->wh_pack && a->dx && a->npix > 0,
                   "attn_proj_bwd: null pointer / empty tensor");
    MSAU_CHECK_ARG(dtype == MSAU_BF16 && a->C == 64, "attn_proj_bwd: bf16, C = 64 only (dtype %d, C %d)", dtype, a->C);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long ntiles = (a->npix + 15) >> 4;
    const int grid = (int)(cdiv64(ntiles, 4) < 1024 ? cdiv64(ntiles, 4) : 1024);
    const bf16_t *df = static_cast<const bf16_t*>(a->df), *dg = static_cast<const bf16_t*>(a->dg), *dh = static_cast<const bf16_t*>(a->dh);
    const bf16_t *wf = static_cast<const bf16_t*>(a->wf_pack), *wg = static_cast<const bf16_t*>(a->wg_pack), *wh = static_cast<const bf16_t*>(a->wh_pack);
    const bf16_t *add = static_cast<const bf16_t*>(a->add), *mask = static_cast<const bf16_t*>(a->mask_b);
    bf16_t* dx = static_cast<bf16_t*>(a->dx);
#define APB(ADD_, ACC_, MASK_) hipLaunchKernelGGL((attn_proj_bwd_kernel<ADD_, ACC_, MASK_>), dim3(grid), dim3(256), 0, s, df, dg, dh, wf, wg, wh, add, mask, dx, (long long)a->npix)
    const int key = (add ? 4 : 0) | (a->accumulate ? 2 : 0) | (mask ? 1 : 0);
    switch (key) {
        case 0: APB(false, false, false); break;
        case 1: APB(false, false, true); break;
        case 2: APB(false, true, false); break;
        case 3: APB(false, true, true); break;
        case 4: APB(true, false, false); break;
        case 5: APB(true, false, true); break;
        case 6: APB(true, true, false); break;
        default: APB(true, true, true); break;
    }
#undef APB
    MSAU_CHECK_LAUNCH("attn_proj_bwd_kernel");
    return 0;
}

// =============================================================================================
// The two data gradients of a 1x1 conv over concat(x1, x2) (the coupling conv of a coupled stage, model/model.py:143-148) at 64 + 64
// channels in ONE launch: dx1 = [mask1] (W1^T g [+ dx1_old]), dx2 = [mask2] (W2^T g [+ dx2_old]).  (8, 16 and 32 channels have
// MSAU_CONV_DOUT instances of the tile / row kernels; 128 output rows were two launches of conv_lean<CIN64,CT4,K1>: 10.5 + 11.2 us.)
// w1_pack / w2_pack: the two packed data-gradient images (64 rows x kchunk 64, msau_conv2d's row order).
// =============================================================================================
namespace {
__global__ __launch_bounds__(256) void dgrad2_1x1_kernel(const bf16_t* __restrict__ g, const bf16_t* __restrict__ w1, const bf16_t* __restrict__ w2,
                                                         bf16_t* __restrict__ dx1, bf16_t* __restrict__ dx2, const bf16_t* __restrict__ mask1,
                                                         const bf16_t* __restrict__ mask2, int acc1, int acc2, long long npix) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const long long ntiles = (npix + 15) >> 4;
    const long long wstride = (long long)gridDim.x * 4;
    long long tile = (long long)blockIdx.x * 4 + wave;
    if (tile >= ntiles) return;
    bf16x8 A[2][4][2];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            A[0][ct][ks] = load8<bf16_t>(w1 + (ct * 16 + lr) * 64 + ks * 32 + lg * 8);
            A[1][ct][ks] = load8<bf16_t>(w2 + (ct * 16 + lr) * 64 + ks * 32 + lg * 8);
        }
    for (; tile < ntiles; tile += wstride) {
        const long long p = tile * 16 + lr;
        const bool live = p < npix;
        const long long pc = live ? p : 0;
        const bf16x8 b0 = load8<bf16_t>(g + pc * 64 + lg * 8), b1 = load8<bf16_t>(g + pc * 64 + 32 + lg * 8);
        const long long eo = pc * 64 + lg * 16;
        bf16x8 ey[2][2], em[2][2];
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            bf16_t* dx = o ? dx2 : dx1;
            const bf16_t* mk = o ? mask2 : mask1;
            if (o ? acc2 : acc1) { ey[o][0] = load8<bf16_t>(dx + eo); ey[o][1] = load8<bf16_t>(dx + eo + 8); }
            else { ey[o][0] = zero8<bf16_t>(); ey[o][1] = zero8<bf16_t>(); }
            if (mk) { em[o][0] = load8<bf16_t>(mk + eo); em[o][1] = load8<bf16_t>(mk + eo + 8); }
        }
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const bool masked = (o ? mask2 : mask1) != nullptr;
            bf16x8 out[2];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                f32x4 acc = mma8(A[o][ct][0], b0, f32x4{0.f, 0.f, 0.f, 0.f});
                acc = mma8(A[o][ct][1], b1, acc);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = ct * 4 + j;
                    float v = acc[j] + (float)ey[o][c >> 3][c & 7];
                    if (masked) v = ((float)em[o][c >> 3][c & 7] > 0.f) ? v : 0.f;
                    out[c >> 3][c & 7] = (bf16_t)v;
                }
            }
            if (live) {
                bf16_t* dx = o ? dx2 : dx1;
                store8<bf16_t>(dx + eo, out[0]);
                store8<bf16_t>(dx + eo + 8, out[1]);
            }
        }
    }
}
}  // namespace

extern "C" int msau_dgrad2_1x1(void* stream, int dtype, const msau_dgrad2_args* a) {
    MSAU_CHECK_ARG(a && a->g && a->w1_pack && a->w2_pack && a->dx1 && a->dx2 && a->npix > 0, "dgrad2_1x1: null pointer / empty tensor");
    MSAU_CHECK_ARG(dtype == MSAU_BF16 && a->C == 64, "dgrad2_1x1: bf16, C = 64 only (dtype %d, C %d)", dtype, a->C);
    const long long ntiles = (a->npix + 15) >> 4;
    const int grid = (int)(cdiv64(ntiles, 4) < 1024 ? cdiv64(ntiles, 4) : 1024);
    hipLaunchKernelGGL(dgrad2_1x1_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const bf16_t*>(a->g),
                       static_cast<const bf16_t*>(a->w1_pack), static_cast<const bf16_t*>(a->w2_pack), static_cast<bf16_t*>(a->dx1),
                       static_cast<bf16_t*>(a->dx2), static_cast<const bf16_t*>(a->mask1), static_cast<const bf16_t*>(a->mask2),
                       a->accumulate1, a->accumulate2, (long long)a->npix);
    MSAU_CHECK_LAUNCH("dgrad2_1x1_kernel");
    return 0;
}
