// The net's first conv fed with BOX LISTS instead of a painted tensor (MSAU_CONV_OWNER).
//
// The BERT chargrid of the reference (data_generator_funsd_bert.py:64-93 get_box_mask_box_label, :240) is piecewise constant:
// every pixel of a text-line box carries that line's feature vector, feats[value(box)] (768 floats), everything else is zero.
// Painted, it is 1536 bytes per pixel -- 2.1 GB per batch of 16 tiles 336 x 256 -- written once by the painter and read twice,
// by the first conv and by its weight gradient: 6.3 GB of the step's traffic for a tensor whose information is a 4-byte box
// index per pixel plus a table.  So the tensor is never painted:
//
//   forward   y[p][co] = b[co] + sum_tap T[v(p + tap)][tap][co],      T[v][tap][co] = sum_c W[co][c][tap] * r(feats[v][c])
//   backward  dW[co][c][tap] = sum_boxes r(feats[v(box)][c]) * S[box][tap][co],
//             S[box][tap][co] = sum over the pixels q the box owns of g[q - tap][co];            db[co] = sum_p g[p][co]
//
// v(q) = value of the box that owns pixel q (msau_raster_owner: the last box painted over it, as the reference's painter
// leaves it), r() = rounding to the storage type (what the painted tensor would have held), W rounded the same way (what
// the packed weight image holds).  T is 288 bytes per feature row, S 288 bytes per box: both stay in L2.  Every sum runs in
// a fixed order (no atomics): results are reproducible, and equal to the painted path up to the order of the fp32 sums.
// The weight gradient is written as a few slabs (the box range split 32 ways) in the layout msau_wgrad_reduce expects.
#include "msau_common.h"

namespace {

template <typename T> __device__ __forceinline__ float rstore(float v) { return (float)(T)v; }

// ---- Wt[c][j] = r(W[co][c][tap]), j = tap * 8 + co: the weight in the order the two small GEMMs below read it (72 contiguous floats per channel)
template <typename T>
__global__ __launch_bounds__(256) void owner_wt_kernel(const float* __restrict__ w, float* __restrict__ wt, int C) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= C * 72) return;
    const int c = i / 72, j = i - c * 72, tap = j >> 3, co = j & 7;
    wt[i] = rstore<T>(w[((size_t)co * C + c) * 9 + tap]);
}

// ---- T[v][j] = sum_c r(feats[v][c]) * Wt[c][j]: a wave takes 64 feature rows (lane = row) and 18 of the 72 outputs; the 18 weights
// of a channel are wave-uniform (scalar loads), the lane's own feature row streams through the vector cache
constexpr int JW = 18;
template <typename T>
__global__ __launch_bounds__(64) void owner_table_kernel(const float* __restrict__ wt, const float* __restrict__ feats, float* __restrict__ table,
                                                         int n_vec, int C) {
    const int v = blockIdx.x * 64 + threadIdx.x, j0 = blockIdx.y * JW;
    const float* row = feats + (size_t)(v < n_vec ? v : n_vec - 1) * C;
    float acc[JW];
#pragma unroll
    for (int t = 0; t < JW; ++t) acc[t] = 0.f;
    for (int c = 0; c < C; ++c) {
        const float f = rstore<T>(row[c]);
        const float* wr = wt + (size_t)c * 72 + j0;                       // wave-uniform
#pragma unroll
        for (int t = 0; t < JW; ++t) acc[t] += f * wr[t];
    }
    if (v < n_vec) {
#pragma unroll
        for (int t = 0; t < JW; ++t) table[(size_t)v * 72 + j0 + t] = acc[t];
    }
}

// ---- bf16 storage: the same product on the matrix cores (both operands ARE bf16 values, so the products are exact and only the
// order of the fp32 sums differs).  Wj[j][c] (j padded to 80) is the weight in bf16, B-fragment order; a wave takes 16 feature rows:
// A[row][k] = r(feats[row][c0 + k]) (32 bytes of the lane's fp32 row -> 8 bf16), five 16-column tiles of j, K = C in steps of 32.
__global__ __launch_bounds__(256) void owner_wj_kernel(const float* __restrict__ w, bf16_t* __restrict__ wj, int C, int Cp) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 80 * Cp) return;
    const int j = i / Cp, c = i - j * Cp, tap = j >> 3, co = j & 7;
    wj[i] = (bf16_t)((j < 72 && c < C) ? w[((size_t)co * C + c) * 9 + tap] : 0.f);
}

__global__ __launch_bounds__(64) void owner_table_mfma_kernel(const bf16_t* __restrict__ wj, const float* __restrict__ feats, float* __restrict__ table,
                                                              int n_vec, int C, int Cp) {
    const int lane = threadIdx.x, lr = lane & 15, lg = lane >> 4;
    const int v0 = blockIdx.x * 16, t0 = blockIdx.y;              // 16 feature rows x one 16-column tile of j per wave (five tiles: 163 -> 815 waves at cfg 4)
    const int vr = v0 + lr < n_vec ? v0 + lr : n_vec - 1;
    const float* row = feats + (size_t)vr * C;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const bool vec = (C & 3) == 0;
    for (int c0 = 0; c0 < Cp; c0 += 32) {
        const int c = c0 + lg * 8;
        bf16x8 a;
        if (vec && c + 8 <= C) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(row + c), hi = *reinterpret_cast<const f32x4*>(row + c + 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) { a[k] = (bf16_t)lo[k]; a[4 + k] = (bf16_t)hi[k]; }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = (bf16_t)(c + k < C ? row[c + k] : 0.f);
        }
        const bf16x8 b = load8<bf16_t>(wj + (size_t)(t0 * 16 + lr) * Cp + c);
        acc = mma8(a, b, acc);
    }
    // D: column = lane & 15 (j within the tile), row = 4 (lane >> 4) + reg (feature row within the 16)
    const int j = t0 * 16 + lr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int v = v0 + 4 * lg + r;
        if (j < 72 && v < n_vec) table[(size_t)v * 72 + j] = acc[r];
    }
}

// ---- forward: a thread per pixel, taps in (ky, kx) order
template <typename T>
__global__ __launch_bounds__(256) void owner_fwd_kernel(const int32_t* __restrict__ owner, const int32_t* __restrict__ boxes, const float* __restrict__ table,
                                                        const float* __restrict__ bias, T* __restrict__ y, int B, int H, int W, int n_vec, bool relu_out) {
    const int64_t npix = (int64_t)B * H * W;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(p % W);
        const int64_t r = p / W;
        const int yy = (int)(r % H);
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
        if (bias) { a0 = *reinterpret_cast<const f32x4*>(bias); a1 = *reinterpret_cast<const f32x4*>(bias + 4); }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int qy = yy + ky - 1;
            if ((unsigned)qy >= (unsigned)H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int qx = x + kx - 1;
                if ((unsigned)qx >= (unsigned)W) continue;
                const int o = owner[p + (int64_t)(ky - 1) * W + (kx - 1)];
                if (o < 0) continue;
                const int v = boxes[(size_t)o * 6 + 5];
                if ((unsigned)v >= (unsigned)n_vec) continue;
                const float* t = table + (size_t)v * 72 + (ky * 3 + kx) * 8;
                a0 += *reinterpret_cast<const f32x4*>(t);
                a1 += *reinterpret_cast<const f32x4*>(t + 4);
            }
        }
        typename Vec8<T>::type out;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            out[c] = (T)(relu_out ? fmaxf(a0[c], 0.f) : a0[c]);
            out[4 + c] = (T)(relu_out ? fmaxf(a1[c], 0.f) : a1[c]);
        }
        store8<T>(y + p * 8, out);
    }
}

// ---- S[box][tap * 8 + co]: a wave per box; lanes stride over the box's rectangle, then 72 threads add the 64 lanes up in order
template <typename T>
__global__ __launch_bounds__(64) void owner_sums_kernel(const int32_t* __restrict__ owner, const int32_t* __restrict__ boxes, const T* __restrict__ g,
                                                        float* __restrict__ sums, int n_boxes, int B, int H, int W) {
    __shared__ float part[64][73];
    const int lane = threadIdx.x;
    for (int i = blockIdx.x; i < n_boxes; i += gridDim.x) {
        const int32_t* bx = boxes + (size_t)i * 6;
        const int b = bx[0];
        const int y0 = max(bx[1], 0), y1 = min(bx[2], H), x0 = max(bx[3], 0), x1 = min(bx[4], W);
        float acc[72];
#pragma unroll
        for (int k = 0; k < 72; ++k) acc[k] = 0.f;
        if (b >= 0 && b < B && y1 > y0 && x1 > x0) {                       // wave-uniform
            const int w = x1 - x0, area = w * (y1 - y0);
            for (int t = lane; t < area; t += 64) {
                const int qy = y0 + t / w, qx = x0 + t % w;
                const int64_t q = ((int64_t)b * H + qy) * W + qx;
                if (owner[q] != i) continue;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int py = qy - (ky - 1);
                    if ((unsigned)py >= (unsigned)H) continue;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int px = qx - (kx - 1);
                        if ((unsigned)px >= (unsigned)W) continue;
                        const typename Vec8<T>::type gv = load8<T>(g + (((int64_t)b * H + py) * W + px) * 8);
#pragma unroll
                        for (int c = 0; c < 8; ++c) acc[(ky * 3 + kx) * 8 + c] += (float)gv[c];
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 72; ++k) part[lane][k] = acc[k];
        __syncthreads();
        for (int k = lane; k < 72; k += 64) {
            float s = 0.f;
            for (int l = 0; l < 64; ++l) s += part[l][k];
            sums[(size_t)i * 72 + k] = s;
        }
    }
}

// ---- dW[c][j] = sum_boxes r(feats[v(box)][c]) * S[box][j] as `ksplit` partial sums = slabs in the layout msau_wgrad_reduce expects
// ([chunk][co][tap * cch + c % cch]; it adds the slabs up in slab order): a wave takes 64 stored channels (lane = channel), 18 of
// the 72 outputs and one contiguous range of boxes; the box's feature-row index and its 18 sums are wave-uniform
template <typename T>
__global__ __launch_bounds__(64) void owner_wgrad_kernel(const int32_t* __restrict__ boxes, const float* __restrict__ feats, const float* __restrict__ sums,
                                                         float* __restrict__ slabs, int n_boxes, int n_vec, int C, int cch, int kext, int nchunks, int ksplit) {
    const int c = blockIdx.x * 64 + threadIdx.x, j0 = blockIdx.y * JW, ks = blockIdx.z;
    const int per = (n_boxes + ksplit - 1) / ksplit;
    const int b0 = ks * per, b1 = min(n_boxes, b0 + per);
    float acc[JW];
#pragma unroll
    for (int t = 0; t < JW; ++t) acc[t] = 0.f;
    const int cl = c < C ? c : C - 1;
    int i = b0;
    for (; i + 3 < b1; i += 4) {                                            // four feature loads in flight; ONE running order of the sums
        int v[4];
        float f[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = boxes[(size_t)(i + q) * 6 + 5];  // wave-uniform
#pragma unroll
        for (int q = 0; q < 4; ++q) f[q] = (unsigned)v[q] < (unsigned)n_vec ? rstore<T>(feats[(size_t)v[q] * C + cl]) : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float* sr = sums + (size_t)(i + q) * 72 + j0;
#pragma unroll
            for (int t = 0; t < JW; ++t) acc[t] += f[q] * sr[t];
        }
    }
    for (; i < b1; ++i) {
        const int v = boxes[(size_t)i * 6 + 5];
        const float f = (unsigned)v < (unsigned)n_vec ? rstore<T>(feats[(size_t)v * C + cl]) : 0.f;
        const float* sr = sums + (size_t)i * 72 + j0;
#pragma unroll
        for (int t = 0; t < JW; ++t) acc[t] += f * sr[t];
    }
    if (c < C) {
        const int chunk = c / cch, cc = c - chunk * cch;
        float* slab = slabs + (size_t)ks * nchunks * 8 * kext;
#pragma unroll
        for (int t = 0; t < JW; ++t) {
            const int j = j0 + t, tap = j >> 3, co = j & 7;
            slab[((size_t)chunk * 8 + co) * kext + tap * cch + cc] = acc[t];
        }
    }
}

// the ones column (bias gradient) of slab 0, chunk 0, from the channel-sum partials, in block order
__global__ __launch_bounds__(64) void owner_bias_kernel(const float* __restrict__ csum, int csum_blocks, float* __restrict__ slab, int cch, int kext) {
    __shared__ float part[8][8];
    const int co = threadIdx.x & 7, sl = threadIdx.x >> 3;                  // eight interleaved slices of the blocks, then the slices in order
    float s = 0.f;
    for (int k = sl; k < csum_blocks; k += 8) s += csum[(size_t)k * 8 + co];
    part[sl][co] = s;
    __syncthreads();
    if (threadIdx.x < 8) {
        float t = 0.f;
        for (int k = 0; k < 8; ++k) t += part[k][co];
        slab[(size_t)co * kext + 9 * cch] = t;
    }
}

}  // namespace

int msau_ownerconv_takes(int dtype, const msau_conv_desc* d) {
    return (d->flags & MSAU_CONV_OWNER) && !(d->flags & ~(MSAU_CONV_OWNER | MSAU_CONV_RELU_OUT)) && d->Cout == 8 && d->C2 == 0 && d->KH == 3 && d->KW == 3 &&
           d->dil == 1 && d->stride == 1 && d->ups == 1 && d->pad_t == 1 && d->pad_l == 1 && d->Hin == d->Hout && d->Win == d->Wout &&
           (dtype == MSAU_F32 || dtype == MSAU_BF16);
}

int msau_ownerconv_fwd(hipStream_t s, int dtype, const msau_conv_desc* d) {
    const msau_owner_ctx* c = static_cast<const msau_owner_ctx*>(d->x1);
    MSAU_CHECK_ARG(c && c->owner && c->feats && c->w && c->wt && c->table && c->n_vec >= 0 && c->n_boxes >= 0 && (c->n_boxes == 0 || c->boxes) &&
                   c->C > 0 && c->C <= d->C1, "conv2d: bad MSAU_CONV_OWNER context");
    // the re-ordered weight: 72 * C floats (fp32 storage) or 80 * roundup(C, 32) bf16 values -- the caller says what it allocated
    const int64_t wt_need = dtype == MSAU_F32 ? (int64_t)72 * c->C : (int64_t)40 * roundup(c->C, 32);
    MSAU_CHECK_ARG(c->wt_floats >= wt_need, "conv2d: MSAU_CONV_OWNER workspace wt holds %d floats, %lld needed (72 * roundup(C, 32) always suffices)",
                   c->wt_floats, (long long)wt_need);
    const int64_t npix = (int64_t)d->B * d->Hout * d->Wout;
    int64_t blocks = cdiv64(npix, 256);
    if (blocks > 16384) blocks = 16384;
    const bool relu = d->flags & MSAU_CONV_RELU_OUT;
    if (c->n_vec > 0) {
        if (dtype == MSAU_F32) {
            const dim3 gw(cdiv(c->C * 72, 256)), gt(cdiv(c->n_vec, 64), 72 / JW);
            hipLaunchKernelGGL(owner_wt_kernel<float>, gw, dim3(256), 0, s, c->w, c->wt, c->C);
            hipLaunchKernelGGL(owner_table_kernel<float>, gt, dim3(64), 0, s, c->wt, c->feats, c->table, c->n_vec, c->C);
        } else {
            const int Cp = roundup(c->C, 32);                              // (80 * Cp bf16 values: checked against wt_floats above)
            hipLaunchKernelGGL(owner_wj_kernel, dim3(cdiv(80 * Cp, 256)), dim3(256), 0, s, c->w, reinterpret_cast<bf16_t*>(c->wt), c->C, Cp);
            hipLaunchKernelGGL(owner_table_mfma_kernel, dim3(cdiv(c->n_vec, 16), 5), dim3(64), 0, s, reinterpret_cast<const bf16_t*>(c->wt), c->feats, c->table, c->n_vec, c->C, Cp);
        }
        MSAU_CHECK_LAUNCH("owner_table");
    }
    if (dtype == MSAU_F32)
        hipLaunchKernelGGL(owner_fwd_kernel<float>, dim3((int)blocks), dim3(256), 0, s, c->owner, c->boxes, c->table, d->bias, static_cast<float*>(d->y),
                           d->B, d->Hout, d->Wout, c->n_vec, relu);
    else
        hipLaunchKernelGGL(owner_fwd_kernel<bf16_t>, dim3((int)blocks), dim3(256), 0, s, c->owner, c->boxes, c->table, d->bias, static_cast<bf16_t*>(d->y),
                           d->B, d->Hout, d->Wout, c->n_vec, relu);
    MSAU_CHECK_LAUNCH("owner_fwd");
    return 0;
}

// slabs an MSAU_CONV_OWNER weight gradient writes (= ways the box range is split): what the slab reduction must be told
int msau_ownerconv_slabs(const msau_wgrad_desc* d) {
    int k = 96;                                                     // (32: 50 us for the 2605 boxes of the bench's cfg-4 batch; each part is a dependent chain of loads)
    if (k > d->nslabs) k = d->nslabs;
    return k < 1 ? 1 : k;
}

int msau_ownerconv_wgrad(hipStream_t s, int dtype, const msau_wgrad_desc* d, int cch, int nchunks, int kext) {
    const msau_owner_ctx* c = static_cast<const msau_owner_ctx*>(d->x1);
    MSAU_CHECK_ARG(c && c->owner && c->feats && c->sums && c->csum && c->csum_blocks >= 1 && c->n_boxes >= 0 && (c->n_boxes == 0 || c->boxes) &&
                   c->C > 0 && c->C <= d->C1, "wgrad: bad MSAU_CONV_OWNER context");
    MSAU_CHECK_ARG(d->Cout == 8 && d->C2 == 0 && d->KH == 3 && d->KW == 3 && d->dil == 1 && d->stride == 1 && d->pad_t == 1 && d->pad_l == 1 &&
                   kext >= 9 * cch + 1 && cch * nchunks >= d->C1 && !(d->flags & MSAU_CONV_RELU_IN), "wgrad: MSAU_CONV_OWNER is the 3x3 C -> 8 conv of the net's input");
    const int64_t npix = (int64_t)d->B * d->Hout * d->Wout;
    int rc = msau_channel_sum(s, dtype, d->g, npix, 8, c->csum, c->csum_blocks);
    if (rc) return rc;
    // the box range is split `ksplit` ways: each part writes one slab, the slab reduction (told nslabs = msau_ownerconv_slabs) adds them up
    const int ksplit = msau_ownerconv_slabs(d);
    const size_t slab_elems = (size_t)nchunks * 8 * kext;
    hipError_t e = hipMemsetAsync(d->slabs, 0, (size_t)ksplit * slab_elems * 4, s);      // padded channels, ones / padding columns
    if (e != hipSuccess) return msau_set_error(MSAU_ERR_HIP, "wgrad: memset: %s", hipGetErrorString(e));
    if (c->n_boxes > 0) {
        const int nb = c->n_boxes < 8192 ? c->n_boxes : 8192;
        const dim3 grid(cdiv(c->C, 64), 72 / JW, ksplit);
        if (dtype == MSAU_F32) {
            hipLaunchKernelGGL(owner_sums_kernel<float>, dim3(nb), dim3(64), 0, s, c->owner, c->boxes, static_cast<const float*>(d->g), c->sums, c->n_boxes, d->B, d->Hout, d->Wout);
            hipLaunchKernelGGL(owner_wgrad_kernel<float>, grid, dim3(64), 0, s, c->boxes, c->feats, c->sums, d->slabs, c->n_boxes, c->n_vec, c->C, cch, kext, nchunks, ksplit);
        } else {
            hipLaunchKernelGGL(owner_sums_kernel<bf16_t>, dim3(nb), dim3(64), 0, s, c->owner, c->boxes, static_cast<const bf16_t*>(d->g), c->sums, c->n_boxes, d->B, d->Hout, d->Wout);
            hipLaunchKernelGGL(owner_wgrad_kernel<bf16_t>, grid, dim3(64), 0, s, c->boxes, c->feats, c->sums, d->slabs, c->n_boxes, c->n_vec, c->C, cch, kext, nchunks, ksplit);
        }
        MSAU_CHECK_LAUNCH("owner_wgrad");
    }
    hipLaunchKernelGGL(owner_bias_kernel, dim3(1), dim3(64), 0, s, c->csum, c->csum_blocks, d->slabs, cch, kext);
    MSAU_CHECK_LAUNCH("owner_bias");
    return 0;
}
