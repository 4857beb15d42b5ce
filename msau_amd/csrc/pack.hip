// Parameter packing / gradient un-packing between the reference's parameter layouts (fp32 OIHW for
// Conv2d, IOHW for ConvTranspose2d; model/layers/layers.py:56-58,127-129,221-226) and the images
// the MFMA kernels read / the slabs they write.  Also the error plumbing of the C ABI.
#include "msau_common.h"
#include <cstdlib>

static thread_local char g_err[512] = "";

int msau_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* msau_last_error(void) { return g_err; }
extern "C" int msau_version(void) { return 10; }

extern "C" int msau_lds_pixel_stride(int raw_bytes, int esz, int c8_per_chunk, int read_stride) { return lds_pixel_stride(raw_bytes, esz, c8_per_chunk, read_stride); }
extern "C" int msau_lds_wrow_stride(int nks, int esz) { return lds_wrow_stride(nks, esz); }

// sizeof() of every struct that crosses the ABI by pointer, so that a binding can check its mirror (tests/test_host_cpu.py)
extern "C" int msau_sizeof(int which) {
    switch (which) {
        case 0: return (int)sizeof(msau_conv_desc);
        case 1: return (int)sizeof(msau_wgrad_desc);
        case 2: return (int)sizeof(msau_pack_entry);
        case 3: return (int)sizeof(msau_unpack_entry);
        case 4: return (int)sizeof(msau_op);
        case 5: return (int)sizeof(msau_lrn_args);
        case 6: return (int)sizeof(msau_pool_args);
        case 7: return (int)sizeof(msau_attn_args);
        case 8: return (int)sizeof(msau_csum_args);
        case 9: return (int)sizeof(msau_reduce_args);
        case 10: return (int)sizeof(msau_conv_pack_geom);
        case 11: return (int)sizeof(msau_wgrad_geom);
        case 12: return (int)sizeof(msau_conv_pair_desc);
        case 13: return (int)sizeof(msau_box_args);
        case 14: return (int)sizeof(msau_allreduce_args);
        case 15: return (int)sizeof(msau_owner_ctx);
        case 16: return (int)sizeof(msau_attn_proj_bwd_args);
        case 17: return (int)sizeof(msau_dgrad2_args);
        default: return -1;
    }
}

namespace {

// stored K channel -> real channel of the (concatenated) parameter dim, or -1 for padding
__device__ __forceinline__ int real_channel(int cs, int k1_real, int k1_store, int k2_real, int k2_store) {
    if (cs < k1_store) return cs < k1_real ? cs : -1;
    int c2 = cs - k1_store;
    return c2 < k2_real ? k1_real + c2 : -1;
}

__global__ void pack_kernel(const float* __restrict__ params, unsigned char* __restrict__ arena,
                            const msau_pack_entry* __restrict__ table) {
    const msau_pack_entry e = table[blockIdx.y];
    const float* src = params + e.src_off;
    if (e.kind == 1) {                                  // bias: fp32, zero padded
        float* dst = reinterpret_cast<float*>(arena + e.dst_off);
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < e.rows_pad; i += gridDim.x * blockDim.x)
            dst[i] = i < e.rows_real ? src[e.row_off + i] : 0.f;
        return;
    }
    const int taps = e.KH * e.KW;
    // more than 128 rows: slices of 128 rows, image [slice][chunk][row][k], the row permutation is per slice (conv.hip)
    const int srows = e.rows_pad > 128 ? 128 : e.rows_pad;
    const int CT = srows >> 4;
    const int64_t total = (int64_t)e.nchunks * e.rows_pad * e.kchunk;
    // a thread writes 8 consecutive k of one row (kchunk is a multiple of 32): one 16-byte (bf16) / two 16-byte (fp32) stores.  With
    // one element per thread the launch was 18.5 k workgroups for 2.5 M elements -- 16 us of workgroup dispatch at the head of every step.
    for (int64_t i8 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i8 < total; i8 += (int64_t)gridDim.x * blockDim.x * 8) {
        const int k0 = (int)(i8 % e.kchunk);
        const int64_t r = i8 / e.kchunk;
        const int slot = (int)(r % srows);
        const int chunk = (int)((r / srows) % e.nchunks);
        const int slice = (int)(r / ((int64_t)srows * e.nchunks));
        // row slot (ct*16 + 4q + j)  <->  output channel q*CT*4 + ct*4 + j   (epilogue layout of conv.hip)
        const int ct = slot >> 4, q = (slot >> 2) & 3, j = slot & 3;
        const int row = slice * 128 + q * (CT * 4) + ct * 4 + j;
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + u;
            v[u] = 0.f;
            if (row < e.rows_real && k < taps * e.cch) {
                int tap = k / e.cch, c = k - tap * e.cch;
                int kc = real_channel(chunk * e.cch + c, e.k1_real, e.k1_store, e.k2_real, e.k2_store);
                if (kc >= 0) {
                    int ky = tap / e.KW, kx = tap - ky * e.KW;
                    if (e.flip) { ky = e.KH - 1 - ky; kx = e.KW - 1 - kx; }
                    int prow = e.row_off + row;
                    int i0 = e.row_is_dim0 ? prow : kc;
                    int i1 = e.row_is_dim0 ? kc : prow;
                    v[u] = src[(((int64_t)i0 * e.dim1 + i1) * e.KH + ky) * e.KW + kx];
                }
            }
        }
        if (e.dtype == MSAU_F32) {
            float* dst = reinterpret_cast<float*>(arena + e.dst_off) + i8;
            *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
        } else {
            bf16x8 o;
#pragma unroll
            for (int u = 0; u < 8; ++u) o[u] = (bf16_t)v[u];
            *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(arena + e.dst_off) + i8) = o;
        }
    }
}

// Slab reduction.  A slab is treated as the flat array it is ([chunk][row][kext], `slab_elems` floats, 256-byte
// aligned): a workgroup owns runs of 64 consecutive floats, 16 threads x float4 cover a run (whole 128-byte lines, each
// read exactly once), the 16 thread rows sum slabs t, t+16, ... with four independent loads in flight, and the 16
// partial sums are combined in a fixed order (bit-reproducible).  Only then is the flat position decoded into the
// parameter's OIHW / IOHW element.  The "ones" column of a conv's slab (bias gradient) is part of the same pass.
// (The first version gave each workgroup 32 columns of ONE row: rows are 64-byte aligned, so most 128-byte groups
// straddled two lines, every line was fetched twice from different XCDs and the reduction ran at 0.9 TB/s -- 150 us per
// stage on the side stream, during which the data-gradient kernels of the main stream stalled.)
__global__ __launch_bounds__(256) void unpack_kernel(const float* __restrict__ slabs, float* __restrict__ grads,
                                                     const msau_unpack_entry* __restrict__ table) {
    constexpr int NC = 64, NS = 16;
    __shared__ float red[NS][NC + 4];
    const msau_unpack_entry e = table[blockIdx.y];
    const int taps = e.KH * e.KW;
    const int kreal = taps * e.cch;                             // real K columns per chunk (the ones column follows)
    const int rows_store = e.slab_elems / (e.kext * e.nchunks);
    const int ngroups = (e.slab_elems + NC - 1) / NC;
    const float* s0 = slabs + e.slab_off;
    const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const bool conv_bias = e.b_off >= 0 && e.b_elem_stride == e.kext;     // bias = ones column of these slabs
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int f0 = grp * NC + cl * 4;                       // flat position of this thread's 4 floats
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
        if (f0 < e.slab_elems) {
            const float* p = s0 + f0;
            int s = sl;
            for (; s + 3 * NS < e.nslabs; s += 4 * NS) {
                a0 += *reinterpret_cast<const f32x4*>(p + (int64_t)s * e.slab_elems);
                a1 += *reinterpret_cast<const f32x4*>(p + (int64_t)(s + NS) * e.slab_elems);
                a2 += *reinterpret_cast<const f32x4*>(p + (int64_t)(s + 2 * NS) * e.slab_elems);
                a3 += *reinterpret_cast<const f32x4*>(p + (int64_t)(s + 3 * NS) * e.slab_elems);
            }
            for (; s < e.nslabs; s += NS) a0 += *reinterpret_cast<const f32x4*>(p + (int64_t)s * e.slab_elems);
        }
        const f32x4 sum = (a0 + a1) + (a2 + a3);
        __syncthreads();
        *reinterpret_cast<f32x4*>(&red[sl][cl * 4]) = sum;
        __syncthreads();
        if (threadIdx.x < NC) {
            const int f = grp * NC + threadIdx.x;
            if (f < e.slab_elems) {
                float t = 0.f;
#pragma unroll
                for (int i = 0; i < NS; ++i) t += red[i][threadIdx.x];
                const int k = f % e.kext;
                const int rr = f / e.kext;
                const int r = rr % rows_store, chunk = rr / rows_store;
                if (r < e.rows_real) {
                    if (k < kreal) {
                        const int tap = k / e.cch, c = k - tap * e.cch;
                        const int kc = real_channel(chunk * e.cch + c, e.k1_real, e.k1_store, e.k2_real, e.k2_store);
                        if (kc >= 0) {
                            const int ky = tap / e.KW, kx = tap - ky * e.KW;
                            const int i0 = e.row_is_dim0 ? r : kc;
                            const int i1 = e.row_is_dim0 ? kc : r;
                            float* dst = grads + e.w_off + (((int64_t)i0 * e.dim1 + i1) * e.KH + ky) * e.KW + kx;
                            *dst = e.accumulate ? *dst + t : t;
                        }
                    } else if (k == kreal && chunk == 0 && conv_bias && r < e.b_count) {
                        float* dst = grads + e.b_off + r;
                        *dst = e.accumulate ? *dst + t : t;
                    }
                }
            }
        }
    }
    if (e.b_off >= 0 && !conv_bias) {
        // bias gradient from channel-sum partials (transposed conv): [b_nslabs][b_slab_stride], element stride 1.  A column (= bias
        // element) is summed by 256 / ncol threads with four independent loads in flight each, then in a fixed order.  (Rounds 1-4 gave a
        // column 4 threads and one dependent 4-byte load per iteration: 64 L2 round trips in a row -- for the level-0 / level-1 layers'
        // 256 partials that serial chain, not the 70 MB of slabs, was the duration of the whole reduction launch: 44 us.)
        const int ncol = e.b_count <= 16 ? 16 : 64, nrow = 256 / ncol;
        const int col = threadIdx.x % ncol, row = threadIdx.x / ncol;
        const int bgroups = (e.b_count + ncol - 1) / ncol;
        for (int grp = blockIdx.x; grp < bgroups; grp += gridDim.x) {
            const int r = grp * ncol + col;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            if (r < e.b_count) {
                const float* p = slabs + e.b_src_off + (int64_t)r * e.b_elem_stride;
                int s = row;
                for (; s + 3 * nrow < e.b_nslabs; s += 4 * nrow) {
                    a0 += p[(int64_t)s * e.b_slab_stride];
                    a1 += p[(int64_t)(s + nrow) * e.b_slab_stride];
                    a2 += p[(int64_t)(s + 2 * nrow) * e.b_slab_stride];
                    a3 += p[(int64_t)(s + 3 * nrow) * e.b_slab_stride];
                }
                for (; s < e.b_nslabs; s += nrow) a0 += p[(int64_t)s * e.b_slab_stride];
            }
            __syncthreads();
            red[row][col] = (a0 + a1) + (a2 + a3);
            __syncthreads();
            if (row == 0 && r < e.b_count) {
                float t = 0.f;
                for (int i = 0; i < nrow; ++i) t += red[i][col];
                float* dst = grads + e.b_off + r;
                *dst = e.accumulate ? *dst + t : t;
            }
        }
    }
}

}  // namespace

extern "C" int msau_pack_params(void* stream, const float* flat_params, void* pack_arena,
                                const msau_pack_entry* table_dev, int n_entries, int max_elems_per_entry) {
    MSAU_CHECK_ARG(flat_params && pack_arena && table_dev && n_entries > 0, "pack_params: bad args");
    int bx = cdiv(max_elems_per_entry, 256 * 8);             // (8 elements per thread)
    if (bx > 64) bx = 64;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(pack_kernel, dim3(bx, n_entries), dim3(256), 0, static_cast<hipStream_t>(stream),
                       flat_params, static_cast<unsigned char*>(pack_arena), table_dev);
    MSAU_CHECK_LAUNCH("pack_kernel");
    return 0;
}

extern "C" int msau_wgrad_reduce(void* stream, const float* slab_arena, float* flat_grads,
                                 const msau_unpack_entry* table_dev, int n_entries, int max_elems_per_entry) {
    MSAU_CHECK_ARG(slab_arena && flat_grads && table_dev && n_entries > 0, "wgrad_reduce: bad args");
    int bx = cdiv(max_elems_per_entry, 64);
    if (bx > 128) bx = 128;                                   // (64 ... 592 measured: 0.0 % of the step)
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(unpack_kernel, dim3(bx, n_entries), dim3(256), 0, static_cast<hipStream_t>(stream),
                       slab_arena, flat_grads, table_dev);
    MSAU_CHECK_LAUNCH("unpack_kernel");
    return 0;
}
