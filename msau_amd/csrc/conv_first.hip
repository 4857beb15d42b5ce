// The net's first conv fed with the API's input tensor itself (MSAU_CONV_NCHW): fp32 NCHW [B][C][H][W] in, 3x3 SAME, 64 stored
// input channels -> 8 output channels (Conv2dBnLrnDrop of model/model.py:136-140 on the chargrid), bf16 storage.
//
// Before: msau_nchw_to_nhwc (read 4 B, write 2 B per element) and then the conv (read the 2 B again): at the bench size
// 528 MB + 198 MB in two launches, 81 + 44 us.  Here a workgroup streams down a band of image rows: per row it loads the 64
// channel planes' row (fp32, whole 128-byte lines), rounds to bf16, writes the pixels into a three-row ring in LDS and -- once --
// the NHWC copy the weight gradient of the backward reads (y2), and computes the output row from the ring with the MFMA
// sequence of conv_lean.hip's 64 -> 8 instance (k = [tap][channel], 18 steps of 32, bias in the epilogue): same products, same
// order, same roundings -- bit-identical to the two launches (tests/test_fused_gpu.py).  352 + 176 + 22 MB, one launch.
//
// Layout.  Ring slot = (roundup(W, 16) + 2) pixels x 160 bytes (128 of channels + 32: the stride that keeps the fragment
// reads of the real 16-lane groups conflict-free, msau_common.h); pixel p of a row sits at entry p + 1, entries 0 and > W stay zero
// (the SAME padding).  Weights: 16 rows x (18 x 64 + 32) bytes.  3 x 258 x 160 + 18.5 KB = 139 KB at W = 256: one workgroup per CU.
// Loads: a wave-item is 32 pixels (8 quads) x 64 channels; lane = (channel group cg = lane & 7, quad lane >> 3) reads the float4 of
// its quad from the 8 planes of its group (8 lanes of a plane = one 128-byte line), transposes 8 x 4 in registers and owns four
// 16-byte pixel pieces: the 8 lanes of a quad write whole pixels (128 contiguous bytes) to LDS and to y2.  Two rows of loads are
// in flight per workgroup (2 x 64 KB at W = 256) while the previous row is computed.
#include "msau_common.h"
#include <cstdlib>

#ifndef MSAU_FIRST_DEPTH
#define MSAU_FIRST_DEPTH 2                     // (3 measured the same 103 us: the walk is not bound by the loads in flight)
#endif

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOOB = 0x80000000u;
constexpr int PS = 160;                           // bytes per pixel entry of a ring slot
constexpr int WS = 18 * 64 + 32;                  // bytes per weight row
constexpr int W_BYTES = 16 * WS;

struct FirstArgs {
    const float* x;                               // [B][C][H][W]
    unsigned char* xn;                            // [B][H][W][64] bf16 or NULL
    const bf16_t* wpack;                          // [16][576]
    const float* bias;                            // [>= 8] or NULL
    unsigned char* y;                             // [B][H][W][8] bf16
    int B, C, H, W;
    int nseg, SH;                                 // row bands per image, rows per band
    int slot_px;                                  // pixel entries per ring slot
    int relu_out;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// NI: wave-items (32-pixel blocks) per wave and row: W <= 128 * NI.  NTW: 16-pixel column tiles per wave: W <= 64 * NTW.
template <int NI, int NTW>
__global__ __launch_bounds__(256) void first_conv_nchw_kernel(const FirstArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int b = blockIdx.x / a.nseg, seg = blockIdx.x - b * a.nseg;
    const int H = a.H, W = a.W;
    const int y0 = seg * a.SH, y1 = min(H, y0 + a.SH);
    if (y0 >= y1) return;                                                  // workgroup-uniform
    const int slot_bytes = a.slot_px * PS;
    unsigned char* wl = smem + 3 * slot_bytes;

    // ---- weights -> LDS (through registers, all loads at one point), ring cleared (halo entries and columns beyond W stay zero)
    {
        constexpr int NWI = 16 * 72, NITW = (NWI + 255) / 256;             // 16-byte pieces of the packed image
        bf16x8 wr[NITW];
#pragma unroll
        for (int it = 0; it < NITW; ++it) {
            const int idx = min(tid + it * 256, NWI - 1);
            const int r = idx / 72, g8 = idx - r * 72;
            wr[it] = load8<bf16_t>(a.wpack + r * 576 + g8 * 8);
        }
        for (int i = tid; i < 3 * slot_bytes / 16; i += 256) *reinterpret_cast<u32x4*>(smem + i * 16) = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int it = 0; it < NITW; ++it) {
            const int idx = tid + it * 256;
            const int r = idx / 72, g8 = idx - r * 72;
            if (idx < NWI) *reinterpret_cast<bf16x8*>(wl + r * WS + g8 * 16) = wr[it];
        }
    }

    const unsigned plane = (unsigned)H * (unsigned)W * 4u;                 // bytes of one channel plane
    const float* xb = a.x + (size_t)b * a.C * H * W;
    const unsigned img_n = (unsigned)H * (unsigned)W * 128u;
    unsigned char* nb = a.xn ? a.xn + (size_t)b * img_n : nullptr;
    const unsigned img_y = (unsigned)H * (unsigned)W * 16u;
    const __amdgpu_buffer_rsrc_t ry = rsrc_of(a.y + (size_t)b * img_y, img_y);

    // ---- per-lane constants of the row loads: item it of this wave = pixel block (wave + 4 * it), quad qs of it, channel group cg
    const int cg = lane & 7, qs = lane >> 3;
    unsigned ld_off[NI];                                                   // byte offset of (channel cg * 8, row 0, quad) or out of range
    unsigned st_off[NI];                                                   // ... of the quad's first pixel, channel group cg, in an NHWC row
    int lds_px[NI];                                                        // ... in a ring slot (0: a quad beyond W)
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        const int px0 = ((wave + 4 * it) * 8 + qs) * 4;                    // first pixel of the quad
        ld_off[it] = px0 < W ? (unsigned)(cg * 8) * plane + (unsigned)px0 * 4u : kOOB;
        st_off[it] = px0 < W ? (unsigned)px0 * 128u + (unsigned)cg * 16u : kOOB;
        lds_px[it] = px0 < W ? (px0 + 1) * PS : 0;
    }
    constexpr int DEPTH = MSAU_FIRST_DEPTH;                                // rows of loads in flight
    f32x4 rr[DEPTH][NI][8];
    // Every load of a row is issued unconditionally, in straight-line code: a row outside the image gets a buffer of ZERO records
    // (a scalar select), a quad beyond W an offset beyond any buffer, channels beyond the real C fall behind the image's C planes --
    // all of them return 0 in hardware.  (With the conditions folded into the offsets the compiler branched around every load and
    // waited with vmcnt(0) between them: 163 us for the bench's input instead of the two launches' 125.)
    auto load_row = [&](int r, f32x4 (&v)[NI][8]) {
        const bool ok = r >= 0 && r < H;
        const __amdgpu_buffer_rsrc_t rx = rsrc_of(xb, ok ? (unsigned)a.C * plane : 0u);
        const unsigned ro = ok ? (unsigned)r * (unsigned)W * 4u : 0u;
#pragma unroll
        for (int it = 0; it < NI; ++it)
#pragma unroll
            for (int k = 0; k < 8; ++k)
                v[it][k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ld_off[it] + (unsigned)k * plane + ro, 0, 0));
    };
    // row r (registers v) -> ring slot (r + 1) % 3 and, for the band's own rows, the NHWC copy
    auto put_row = [&](int r, const f32x4 (&v)[NI][8]) {
        unsigned char* slot = smem + ((r + 3) % 3) * slot_bytes;          // (r >= -1)
        const bool own = r >= y0 && r < y1;
        const __amdgpu_buffer_rsrc_t rn = rsrc_of(nb, own && nb ? img_n : 0u);      // (halo rows: zero records, nothing is stored)
        const unsigned no = own ? (unsigned)r * (unsigned)W * 128u : 0u;
#pragma unroll
        for (int it = 0; it < NI; ++it) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                bf16x8 o;
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] = (bf16_t)v[it][k][p];
                // (a quad beyond W holds zeros: it writes them to entry 0 of the slot, the left halo, which is zero)
                *reinterpret_cast<bf16x8*>(slot + lds_px[it] + (lds_px[it] ? p * PS : 0) + cg * 16) = o;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rn, st_off[it] + (unsigned)p * 128u + no, 0, 0);
            }
        }
    };

    // ---- B-fragment offsets: k-step ks = tap ks / 2, channel groups (ks & 1) * 4 + lg; slot of input row o - 1 + ky
    const int col0 = wave * 16 + lr;                                       // this lane's pixel in column tile t: col0 + 64 * t
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.bias && lg < 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = a.bias[lg * 4 + j];
    }
    auto compute_row = [&](int o) {
        f32x4 acc[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 18; ++ks) {
            const int tap = ks >> 1, ky = tap / 3, kx = tap - ky * 3;
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(wl + lr * WS + ks * 64 + lg * 16);
            const unsigned char* rowp = smem + ((o + ky + 2) % 3) * slot_bytes + (col0 + kx) * PS + ((ks & 1) * 4 + lg) * 16;
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                // (tiles beyond W read zeros of the ring or of the weights behind it -- finite either way -- and are not stored)
                const bf16x8 bf = *reinterpret_cast<const bf16x8*>(rowp + t * 64 * PS);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc[t], 0, 0, 0);
            }
        }
        const unsigned yo = (unsigned)o * (unsigned)W * 16u;
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int col = col0 + 64 * t;
            bf16x4 ov;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = acc[t][j] + bv[j];
                if (a.relu_out) v = fmaxf(v, 0.f);
                ov[j] = (bf16_t)v;
            }
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(unsigned __attribute__((ext_vector_type(2))), ov), ry,
                                                  lg < 2 && col < W ? yo + (unsigned)col * 16u + (unsigned)lg * 8u : kOOB, 0, 0);
        }
    };

    // ---- the walk: rows y0 - 1 .. y1 go through the ring; output row r - 1 is computed when input row r has arrived
#pragma unroll
    for (int dd = 0; dd < DEPTH; ++dd) load_row(y0 - 1 + dd, rr[dd]);
    __syncthreads();                                                       // ring cleared, weights staged
    for (int r = y0 - 1; r <= y1; r += DEPTH) {
#pragma unroll
        for (int dd = 0; dd < DEPTH; ++dd) {
            const int row = r + dd;                                        // workgroup-uniform, as is everything tested below
            if (row <= y1) {
                put_row(row, rr[dd]);
                load_row(row + DEPTH <= y1 ? row + DEPTH : -1, rr[dd]);
                __syncthreads();
                if (row - 1 >= y0) compute_row(row - 1);
                __syncthreads();
            }
        }
    }
}

// (a wave's column tiles beyond W read past the slot: the next slot or the weight image behind the ring -- finite values inside the
//  allocation for every W the launch accepts; their results are not stored)
int slot_pixels(int W) { return roundup(W, 16) + 2; }

}  // namespace

// 1 if msau_conv2d takes `d` with MSAU_CONV_NCHW set
int msau_firstconv_takes(int dtype, const msau_conv_desc* d) {
    static const bool off = std::getenv("MSAU_FIRST_NCHW") && std::getenv("MSAU_FIRST_NCHW")[0] == '0';
    if (off || dtype != MSAU_BF16 || !d) return 0;
    if (d->flags & ~(MSAU_CONV_NCHW | MSAU_CONV_RELU_OUT)) return 0;
    if (d->C1 != 64 || d->C2 || d->Cout != 8 || d->KH != 3 || d->KW != 3 || d->dil != 1 || d->stride != 1 || d->ups != 1) return 0;
    if (d->pad_t != 1 || d->pad_l != 1 || d->Hin != d->Hout || d->Win != d->Wout) return 0;
    if (d->Win % 4 || d->Win > 320 || d->Win < 16 || d->Hin < 2) return 0;
    if ((int64_t)d->Hin * d->Win * 64 * 4 >= (1ll << 31)) return 0;         // 32-bit offsets inside an image
    if (3 * slot_pixels(d->Win) * PS + W_BYTES + 256 > MSAU_LDS_LIMIT) return 0;
    return 1;
}

int msau_firstconv_launch(hipStream_t s, int dtype, const msau_conv_desc* d, int real_channels) {
    (void)dtype;
    FirstArgs a;
    a.x = static_cast<const float*>(d->x1);
    a.xn = static_cast<unsigned char*>(d->y2);
    a.wpack = static_cast<const bf16_t*>(d->wpack);
    a.bias = d->bias;
    a.y = static_cast<unsigned char*>(d->y);
    a.B = d->B; a.C = real_channels; a.H = d->Hin; a.W = d->Win;
    // bands: about one workgroup per CU (the ring takes most of a CU's LDS), at least 8 rows each (2 halo rows per band)
    int nseg = cdiv(256, d->B);
    if (nseg > d->Hin / 8) nseg = d->Hin / 8;
    if (nseg < 1) nseg = 1;
    a.SH = cdiv(d->Hin, nseg);
    a.nseg = cdiv(d->Hin, a.SH);
    a.slot_px = slot_pixels(d->Win);
    a.relu_out = (d->flags & MSAU_CONV_RELU_OUT) ? 1 : 0;
    const int lds = 3 * a.slot_px * PS + W_BYTES;
    const int ni = cdiv(d->Win, 128), ntw = cdiv(d->Win, 64);
#define FIRST_CASE(NIV, NTWV)                                                                                                  \
    if (ni <= NIV && ntw <= NTWV) {                                                                                            \
        static bool attr_set = false;                                                                                          \
        if (!attr_set) {                                                                                                       \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&first_conv_nchw_kernel<NIV, NTWV>),              \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, MSAU_LDS_LIMIT);                    \
            if (e != hipSuccess) return msau_set_error(MSAU_ERR_HIP, "first_conv: hipFuncSetAttribute: %s", hipGetErrorString(e)); \
            attr_set = true;                                                                                                   \
        }                                                                                                                      \
        hipLaunchKernelGGL((first_conv_nchw_kernel<NIV, NTWV>), dim3(a.B * a.nseg), dim3(256), lds, s, a);                    \
        MSAU_CHECK_LAUNCH("first_conv_nchw_kernel");                                                                           \
        return 0;                                                                                                              \
    }
    FIRST_CASE(1, 2) FIRST_CASE(2, 4) FIRST_CASE(3, 5)
#undef FIRST_CASE
    return msau_set_error(MSAU_ERR_ARG, "first_conv: no instance for width %d", d->Win);
}
