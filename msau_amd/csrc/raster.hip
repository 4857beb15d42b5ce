// Chargrid rasteriser on device: paints the one-hot character grid and the label mask
// straight into the NHWC activation layout from compact box lists, instead of the reference's Python
// double loop over a dense float64 [C,H,W] array (data_generator_funsd_bert.py:149-186) followed by a
// host->device copy of mostly zeros.  Painter semantics are the reference's: boxes are applied in order,
// a later box overwrites an earlier one (also with "nothing", for characters outside the charset), and
// boxes are clipped to the grid.  Order is resolved with an integer atomicMax of the box index
// (deterministic).
#include "msau_common.h"

namespace {

__global__ void raster_owner_kernel(const int32_t* __restrict__ boxes, int n, int32_t* __restrict__ owner, int B, int H, int W) {
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const int32_t* bx = boxes + (size_t)i * 6;
        const int b = bx[0];
        int y0 = max(bx[1], 0), y1 = min(bx[2], H), x0 = max(bx[3], 0), x1 = min(bx[4], W);
        if (b < 0 || b >= B || y1 <= y0 || x1 <= x0) continue;
        const int w = x1 - x0, area = w * (y1 - y0);
        for (int t = threadIdx.x; t < area; t += blockDim.x) {
            int y = y0 + t / w, x = x0 + t % w;
            atomicMax(&owner[((size_t)b * H + y) * W + x], i);
        }
    }
}

template <typename T>
__global__ void raster_onehot_kernel(const int32_t* __restrict__ boxes, const int32_t* __restrict__ owner,
                                     T* __restrict__ grid, int64_t npix, int C, int Cs) {
    const int cgs = Cs >> 3;
    const int64_t total = npix * cgs;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / cgs;
        const int cg = (int)(i - p * cgs);
        const int o = owner[p];
        const int v = o >= 0 ? boxes[(size_t)o * 6 + 5] : -1;
        typename Vec8<T>::type out;
#pragma unroll
        for (int j = 0; j < 8; ++j) out[j] = (T)((cg * 8 + j == v && v < C) ? 1.0f : 0.0f);
        store8<T>(grid + i * 8, out);
    }
}

__global__ void raster_label_kernel(const int32_t* __restrict__ boxes, const int32_t* __restrict__ owner,
                                    int64_t* __restrict__ labels, int64_t npix) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
        const int o = owner[p];
        labels[p] = o >= 0 ? (int64_t)boxes[(size_t)o * 6 + 5] : 0;
    }
}

// dense painter (data_generator_funsd_bert.py:64-93 get_box_mask_box_label): the owning box's feature vector
// feats[value][0..C) (fp32, row stride C) at every covered pixel, zeros elsewhere and in the padded channels.
// A WAVE paints a pixel: owner and box value are wave-uniform (scalar loads, one dependent chain per pixel instead of one per
// 16 bytes), lane l writes the 8-channel groups l, l + 64, ...: two 16-byte reads of the (cache-resident) feature row, one
// 16-byte store, 128 contiguous bytes per 8 lanes.  Empty pixels (most of a chargrid) skip the reads.  Four pixels per trip
// keep four stores and their reads in flight.  (First version: a thread per 16 bytes with a 64-bit division, three dependent
// loads and eight scalar reads each -- 0.8 ms for the 2.1 GB grid of cfg 4; this one runs at the store bandwidth.)
template <typename T>
__global__ __launch_bounds__(256) void raster_dense_kernel(const int32_t* __restrict__ boxes, const int32_t* __restrict__ owner,
                                                           const float* __restrict__ feats, T* __restrict__ grid, int64_t npix, int C, int Cs) {
    const int cgs = Cs >> 3;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const bool vec = (C & 3) == 0;                                    // rows of feats are 16-byte aligned
    for (int64_t p0 = wave * 4; p0 < npix; p0 += nwaves * 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t p = p0 + q;
            if (p >= npix) break;                                     // wave-uniform
            const int o = owner[p];
            const int v = o >= 0 ? boxes[(size_t)o * 6 + 5] : -1;
            T* dst = grid + p * Cs;
            if (v < 0) {
                for (int cg = lane; cg < cgs; cg += 64) store8<T>(dst + cg * 8, zero8<T>());
                continue;
            }
            const float* row = feats + (size_t)v * C;
            for (int cg = lane; cg < cgs; cg += 64) {
                typename Vec8<T>::type out;
                const int c = cg * 8;
                if (vec && c + 8 <= C) {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(row + c), hi = *reinterpret_cast<const f32x4*>(row + c + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { out[j] = (T)lo[j]; out[4 + j] = (T)hi[j]; }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) out[j] = (T)(c + j < C ? row[c + j] : 0.0f);
                }
                store8<T>(dst + c, out);
            }
        }
    }
}

}  // namespace

extern "C" int msau_raster_dense(void* stream, int dtype, const int32_t* boxes, const int32_t* owner, const float* feats,
                                 void* grid, int B, int H, int W, int C, int Cs) {
    MSAU_CHECK_ARG(owner && grid && feats && B > 0 && H > 0 && W > 0 && C > 0 && Cs >= C && Cs % 8 == 0, "raster_dense: bad args");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t npix = (int64_t)B * H * W;
    int64_t blocks = cdiv64(npix, 16);                                // a wave takes 4 pixels per trip
    if (blocks > 8192) blocks = 8192;
    if (dtype == MSAU_F32)
        hipLaunchKernelGGL(raster_dense_kernel<float>, dim3((int)blocks), dim3(256), 0, s, boxes, owner, feats, static_cast<float*>(grid), npix, C, Cs);
    else if (dtype == MSAU_BF16)
        hipLaunchKernelGGL(raster_dense_kernel<bf16_t>, dim3((int)blocks), dim3(256), 0, s, boxes, owner, feats, static_cast<bf16_t*>(grid), npix, C, Cs);
    else return msau_set_error(MSAU_ERR_ARG, "raster_dense: bad dtype");
    MSAU_CHECK_LAUNCH("raster_dense");
    return 0;
}

extern "C" int msau_raster_owner(void* stream, const int32_t* boxes, int n, int32_t* owner, int B, int H, int W) {
    MSAU_CHECK_ARG(owner && B > 0 && H > 0 && W > 0 && n >= 0 && (n == 0 || boxes), "raster_owner: bad args");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(owner, 0xFF, sizeof(int32_t) * (size_t)B * H * W, s);      // -1
    if (e != hipSuccess) return msau_set_error(MSAU_ERR_HIP, "raster_owner: memset: %s", hipGetErrorString(e));
    if (n == 0) return 0;
    hipLaunchKernelGGL(raster_owner_kernel, dim3(n < 4096 ? n : 4096), dim3(64), 0, s, boxes, n, owner, B, H, W);
    MSAU_CHECK_LAUNCH("raster_owner");
    return 0;
}

extern "C" int msau_raster_onehot(void* stream, int dtype, const int32_t* boxes, const int32_t* owner, void* grid,
                                  int B, int H, int W, int C, int Cs) {
    MSAU_CHECK_ARG(owner && grid && B > 0 && H > 0 && W > 0 && C > 0 && Cs >= C && Cs % 8 == 0, "raster_onehot: bad args");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t npix = (int64_t)B * H * W;
    int64_t blocks = cdiv64(npix * (Cs / 8), 256);
    if (blocks > 8192) blocks = 8192;
    if (dtype == MSAU_F32)
        hipLaunchKernelGGL(raster_onehot_kernel<float>, dim3((int)blocks), dim3(256), 0, s, boxes, owner, static_cast<float*>(grid), npix, C, Cs);
    else if (dtype == MSAU_BF16)
        hipLaunchKernelGGL(raster_onehot_kernel<bf16_t>, dim3((int)blocks), dim3(256), 0, s, boxes, owner, static_cast<bf16_t*>(grid), npix, C, Cs);
    else return msau_set_error(MSAU_ERR_ARG, "raster_onehot: bad dtype");
    MSAU_CHECK_LAUNCH("raster_onehot");
    return 0;
}

extern "C" int msau_raster_labels(void* stream, const int32_t* boxes, const int32_t* owner, int64_t* labels, int B, int H, int W) {
    MSAU_CHECK_ARG(owner && labels && B > 0 && H > 0 && W > 0, "raster_labels: bad args");
    const int64_t npix = (int64_t)B * H * W;
    int64_t blocks = cdiv64(npix, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(raster_label_kernel, dim3((int)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), boxes, owner, labels, npix);
    MSAU_CHECK_LAUNCH("raster_labels");
    return 0;
}
