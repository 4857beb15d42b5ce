// Parameter packing / gradient un-packing between the reference's parameter layouts (fp32 OIHW for
// Conv2d, IOHW for ConvTranspose2d; model/layers/layers.py:56-58,127-129,221-226) and the images
// the MFMA kernels read / the slabs they write.  Also the error plumbing of the C ABI.
#include "msau_common.h"

static thread_local char g_err[512] = "";

int msau_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* msau_last_error(void) { return g_err; }
extern "C" int msau_version(void) { return 1; }

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
    const int CT = e.rows_pad >> 4;
    const int64_t total = (int64_t)e.nchunks * e.rows_pad * e.kchunk;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int k = (int)(i % e.kchunk);
        int64_t r = i / e.kchunk;
        int slot = (int)(r % e.rows_pad);
        int chunk = (int)(r / e.rows_pad);
        // row slot (ct*16 + 4q + j)  <->  output channel q*CT*4 + ct*4 + j   (epilogue layout of conv.hip)
        int ct = slot >> 4, q = (slot >> 2) & 3, j = slot & 3;
        int row = q * (CT * 4) + ct * 4 + j;
        float v = 0.f;
        if (row < e.rows_real && k < taps * e.cch) {
            int tap = k / e.cch, c = k - tap * e.cch;
            int kc = real_channel(chunk * e.cch + c, e.k1_real, e.k1_store, e.k2_real, e.k2_store);
            if (kc >= 0) {
                int ky = tap / e.KW, kx = tap - ky * e.KW;
                if (e.flip) { ky = e.KH - 1 - ky; kx = e.KW - 1 - kx; }
                int prow = e.row_off + row;
                int i0 = e.row_is_dim0 ? prow : kc;
                int i1 = e.row_is_dim0 ? kc : prow;
                v = src[(((int64_t)i0 * e.dim1 + i1) * e.KH + ky) * e.KW + kx];
            }
        }
        if (e.dtype == MSAU_F32) reinterpret_cast<float*>(arena + e.dst_off)[i] = v;
        else reinterpret_cast<bf16_t*>(arena + e.dst_off)[i] = (bf16_t)v;
    }
}

// 256 threads = 32 consecutive K columns (one 128-B line per slab) x 8 slab lanes: slab lane t sums slabs
// t, t+8, ... with independent loads in flight, then the 8 partial sums are combined in a fixed order.
__global__ __launch_bounds__(256) void unpack_kernel(const float* __restrict__ slabs, float* __restrict__ grads,
                                                     const msau_unpack_entry* __restrict__ table) {
    constexpr int NC = 32, NS = 8;
    __shared__ float red[NS][NC + 1];
    const msau_unpack_entry e = table[blockIdx.y];
    const int taps = e.KH * e.KW;
    const int kcols = e.nchunks * taps * e.cch;                 // stored K columns
    const int kc16 = (kcols + NC - 1) / NC;
    const int rows_store = e.slab_elems / (e.kext * e.nchunks);
    const int ngroups = e.rows_real * kc16;                     // groups of 16 consecutive columns of one row
    const float* s0 = slabs + e.slab_off;
    const int col = threadIdx.x & (NC - 1), sl = threadIdx.x / NC;
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int r = grp / kc16;
        const int kk = (grp - r * kc16) * NC + col;
        float sum = 0.f;
        int chunk = 0, k = 0, tap = 0, c = 0, kc = -1;
        if (kk < kcols) {
            chunk = kk / (taps * e.cch);
            k = kk - chunk * taps * e.cch;
            tap = k / e.cch; c = k - tap * e.cch;
            kc = real_channel(chunk * e.cch + c, e.k1_real, e.k1_store, e.k2_real, e.k2_store);
        }
        if (kc >= 0) {
            const float* p = s0 + ((int64_t)chunk * rows_store + r) * e.kext + k;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            int s = sl;
            for (; s + 3 * NS < e.nslabs; s += 4 * NS) {
                a0 += p[(int64_t)s * e.slab_elems];
                a1 += p[(int64_t)(s + NS) * e.slab_elems];
                a2 += p[(int64_t)(s + 2 * NS) * e.slab_elems];
                a3 += p[(int64_t)(s + 3 * NS) * e.slab_elems];
            }
            for (; s < e.nslabs; s += NS) a0 += p[(int64_t)s * e.slab_elems];
            sum = (a0 + a1) + (a2 + a3);
        }
        __syncthreads();
        red[sl][col] = sum;
        __syncthreads();
        if (sl == 0 && kc >= 0) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < NS; ++i) t += red[i][col];
            int ky = tap / e.KW, kx = tap - ky * e.KW;
            int i0 = e.row_is_dim0 ? r : kc;
            int i1 = e.row_is_dim0 ? kc : r;
            float* dst = grads + e.w_off + (((int64_t)i0 * e.dim1 + i1) * e.KH + ky) * e.KW + kx;
            *dst = e.accumulate ? *dst + t : t;
        }
    }
    if (e.b_off >= 0) {
        // bias gradient: the "ones" column of the wgrad slabs, or channel-sum partials; same 16 x 16 scheme
        const int bgroups = (e.b_count + NC - 1) / NC;
        for (int grp = blockIdx.x; grp < bgroups; grp += gridDim.x) {
            const int r = grp * NC + col;
            float a0 = 0.f, a1 = 0.f;
            if (r < e.b_count) {
                const float* p = slabs + e.b_src_off + (int64_t)r * e.b_elem_stride;
                int s = sl;
                for (; s + NS < e.b_nslabs; s += 2 * NS) {
                    a0 += p[(int64_t)s * e.b_slab_stride];
                    a1 += p[(int64_t)(s + NS) * e.b_slab_stride];
                }
                for (; s < e.b_nslabs; s += NS) a0 += p[(int64_t)s * e.b_slab_stride];
            }
            __syncthreads();
            red[sl][col] = a0 + a1;
            __syncthreads();
            if (sl == 0 && r < e.b_count) {
                float t = 0.f;
#pragma unroll
                for (int i = 0; i < NS; ++i) t += red[i][col];
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
    int bx = cdiv(max_elems_per_entry, 256);
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
    int bx = cdiv(max_elems_per_entry, 32);
    if (bx > 128) bx = 128;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(unpack_kernel, dim3(bx, n_entries), dim3(256), 0, static_cast<hipStream_t>(stream),
                       slab_arena, flat_grads, table_dev);
    MSAU_CHECK_LAUNCH("unpack_kernel");
    return 0;
}
