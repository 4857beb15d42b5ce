// Weight / bias gradient of the implicit-GEMM convolution for gfx950.
//
//   dW[co][k] = sum_pix g[pix][co] * xin[pix (+) tap(k)][c(k)]        k = [tap][channel-in-chunk]
//   db[co]    = sum_pix g[pix][co]                                     (the extra "ones" column)
//
// GEMM with the PIXELS on the MFMA reduction dimension: rows = output channels, columns = k.
// Both operands are stored pixel-major (NHWC) in LDS, i.e. transposed w.r.t. what the MFMA wants;
// bf16 uses ds_read_b64_tr_b16 (hardware transpose read, per-lane addresses make the tap shift a
// plain address offset), fp32 uses ds_read_b32 (one element per lane per 16x16x4 MFMA).
// A workgroup walks a strided list of 16x16 pixel tiles with its accumulators in registers and
// writes ONE slab at the end: no atomics, results are bitwise reproducible for a fixed nslabs.
//
// Replaces autograd's conv2d weight/bias gradient reached from train_chargrid_funsd_msau.py:57.
#include "msau_common.h"

namespace {

struct WGeom {
    int esz, taps, cch, nchunks, kreal, kextc, NKT, NKW, CTN, compact, slice;
    int TIH, TIW, PSx, PSg, x_bytes, g_bytes, tab_bytes, total;
};

inline int pix_stride(int c, int esz) {
    int ps = c * esz;
    if (((ps >> 4) & 1) == 0) ps += 16;
    return ps;
}

int wgrad_geom(int dtype, const msau_wgrad_desc* d, WGeom* out) {
    MSAU_CHECK_ARG(dtype == MSAU_F32 || dtype == MSAU_BF16, "wgrad: bad dtype");
    const int Cin = d->C1 + d->C2;
    MSAU_CHECK_ARG(Cin > 0 && Cin % 8 == 0 && d->Cout > 0 && d->Cout % 8 == 0 && d->Cout <= 1024, "wgrad: bad channels");
    MSAU_CHECK_ARG(d->KH >= 1 && d->KW >= 1 && d->KH <= 7 && d->KW <= 7 && d->dil >= 1 && (d->stride == 1 || d->stride == 2), "wgrad: bad kernel");
    WGeom g;
    g.esz = dtype == MSAU_F32 ? 4 : 2;
    g.taps = d->KH * d->KW;
    // wide outputs are launched in slices: 128 channels (bf16) / 64 (fp32: the 256-pixel gradient tile in LDS is 4 bytes wide)
    g.compact = d->stride == 1 && d->dil >= 16 && g.taps > 1;  // large dilation: one 16 x 16 block per tap (see conv.hip)
    g.slice = g.esz == 4 ? (g.compact ? 32 : 64) : 128;        // (fp32 + per-tap blocks: 9 x 256 input pixels leave room for 32)
    const int cout_s = d->Cout > g.slice ? g.slice : d->Cout;
    g.CTN = cdiv(cout_s, 16);
    g.TIH = 15 * d->stride + (d->KH - 1) * d->dil + 1;
    g.TIW = 15 * d->stride + (d->KW - 1) * d->dil + 1;
    if (g.compact) { g.TIH = g.taps * 16; g.TIW = 16; }
    g.PSg = pix_stride(cout_s, g.esz);
    g.g_bytes = roundup(256 * g.PSg, 16);
    int best = 0;
    for (int c = (Cin < 64 ? Cin : 64); c >= 8; c -= 8) {
        if (Cin % c) continue;
        if (d->C2 && d->C1 % c) continue;                    // a chunk never straddles the two sources
        int kext = roundup(g.taps * c + 8, 16);
        int nkw = cdiv(kext / 16, 4);
        if (nkw * g.CTN > 40 || nkw > 10) continue;           // accumulator registers per wave
        int xb = roundup(g.TIH * g.TIW * pix_stride(c, g.esz), 16);
        if (xb + g.g_bytes + 640 + 64 > 150 * 1024) continue;   // 640: the column table at its largest (below)
        best = c; break;
    }
    if (!best) return msau_set_error(MSAU_ERR_LDS, "wgrad: no chunk fits (Cin %d Cout %d k %dx%d dil %d)", Cin, d->Cout, d->KH, d->KW, d->dil);
    g.cch = best;
    g.nchunks = Cin / best;
    g.kreal = g.taps * best;
    g.kextc = roundup(g.kreal + 8, 16);
    g.NKT = g.kextc / 16;
    int nkw = cdiv(g.NKT, 4);
    g.NKW = nkw <= 1 ? 1 : nkw <= 2 ? 2 : nkw <= 3 ? 3 : nkw <= 5 ? 5 : 10;
    g.PSx = pix_stride(best, g.esz);
    g.x_bytes = roundup(g.TIH * g.TIW * g.PSx, 16);
    g.tab_bytes = 64 * g.NKW;                                  // 4 waves x NKW rounds of k-tiles, 4 column groups each (>= kextc / 4 entries)
    g.total = g.x_bytes + g.g_bytes + g.tab_bytes + 64;
    *out = g;
    return 0;
}

struct WArgs {
    msau_wgrad_desc d;                           // d.Cout = channels of THIS slice
    int g_stride, co0, cout_total;               // channels per pixel of g, first channel of the slice, slab rows per chunk
    int compact;
    int cch, nchunks, kreal, kextc, NKT;
    int TIH, TIW, PSx, PSg, x_bytes, g_bytes, tab_bytes;
    int tiles_x, tiles_y, ntiles;
};

#define TAB_ABS 0x40000000

template <typename T, int CTN, int NKW>
__global__ __launch_bounds__(256) void wgrad_kernel(const WArgs a) {
    typedef typename Vec8<T>::type V8;
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* lds_x = smem;
    unsigned char* lds_g = smem + a.x_bytes;
    int* tab = reinterpret_cast<int*>(smem + a.x_bytes + a.g_bytes);          // per 4-column group
    unsigned char* lds_ones = smem + a.x_bytes + a.g_bytes + a.tab_bytes;     // {1,0,0,0,0,0,0,0}
    const int ones_off = a.x_bytes + a.g_bytes + a.tab_bytes - 0;             // relative to lds_x base (= smem)

    const msau_wgrad_desc& d = a.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int chunk = blockIdx.y;
    const int cg_per_chunk = a.cch >> 3;

    // column-group table: byte offset (relative to the pixel's slot in lds_x) of 4 consecutive k.  It covers all 4 * NKW
    // k-tiles of the instance, the ones past kextc as zero columns: a wave issues every round's MFMAs and NO branch
    // surrounds an MFMA of the tile loop (wgrad_lean.hip has the reason: a missed MFMA -> accvgpr_read wait on a branch edge)
    for (int c4 = tid; c4 < 16 * NKW; c4 += 256) {
        int k = c4 * 4, off;
        if (k < a.kreal) {
            int tap = k / a.cch, c = k - tap * a.cch;
            int ky = tap / d.KW, kx = tap - ky * d.KW;
            off = ((ky * d.dil) * a.TIW + kx * d.dil) * a.PSx + c * (int)sizeof(T);
            if (a.compact) off = (tap * 256) * a.PSx + c * (int)sizeof(T);
        } else if (k == a.kreal) {
            off = TAB_ABS | ones_off;                                         // column kreal = ones (bias)
        } else {
            off = TAB_ABS | (ones_off + 4 * (int)sizeof(T));                  // padding columns: zeros
        }
        tab[c4] = off;
    }
    if (tid < 8) reinterpret_cast<T*>(lds_ones)[tid] = (T)(tid == 0 ? 1.0f : 0.0f);

    f32x4 acc[NKW][CTN];
#pragma unroll
    for (int i = 0; i < NKW; ++i)
#pragma unroll
        for (int ct = 0; ct < CTN; ++ct) acc[i][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

    const T* x1 = static_cast<const T*>(d.x1);
    const T* x2 = static_cast<const T*>(d.x2);
    const T* gp = static_cast<const T*>(d.g);
    const bool relu_in = d.flags & MSAU_CONV_RELU_IN;
    const int npix_in = a.TIH * a.TIW;
    const int cog = d.Cout >> 3;

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        int t = tile;
        const int txi = t % a.tiles_x; t /= a.tiles_x;
        const int tyi = t % a.tiles_y; t /= a.tiles_y;
        const int b = t;
        const int oy0 = tyi * 16, ox0 = txi * 16;
        const int vy0 = oy0 * d.stride - d.pad_t, vx0 = ox0 * d.stride - d.pad_l;
        __syncthreads();                       // previous tile's reads done (also orders the table writes)
        for (int idx = tid; idx < npix_in * cg_per_chunk; idx += 256) {
            int pix = idx / cg_per_chunk, cg = idx - pix * cg_per_chunk;
            int iy = pix / a.TIW, ix = pix - iy * a.TIW;
            int ry = vy0 + iy, rx = vx0 + ix;
            if (a.compact) {                                   // pix = (tap, row, column) of that tap's 16 x 16 block
                const int tap = iy >> 4, py = iy & 15;
                const int ky = tap / d.KW, kx = tap - ky * d.KW;
                ry = vy0 + py + ky * d.dil;
                rx = vx0 + ix + kx * d.dil;
            }
            V8 v = zero8<T>();
            if (ry >= 0 && rx >= 0 && ry < d.Hin && rx < d.Win) {
                int cs = chunk * a.cch + cg * 8;
                size_t p = ((size_t)b * d.Hin + ry) * d.Win + rx;
                const T* src = cs < d.C1 ? x1 + p * d.C1 + cs : x2 + p * d.C2 + (cs - d.C1);
                v = load8<T>(src);
                if (relu_in) v = relu8<T>(v);
            }
            *reinterpret_cast<V8*>(lds_x + pix * a.PSx + cg * 8 * (int)sizeof(T)) = v;
        }
        for (int idx = tid; idx < 256 * cog; idx += 256) {
            int m = idx / cog, cg = idx - m * cog;
            int oy = oy0 + (m >> 4), ox = ox0 + (m & 15);
            V8 v = zero8<T>();                                   // pixels outside the image contribute 0
            if (oy < d.Hout && ox < d.Wout)
                v = load8<T>(gp + (((size_t)b * d.Hout + oy) * d.Wout + ox) * a.g_stride + a.co0 + cg * 8);
            *reinterpret_cast<V8*>(lds_g + m * a.PSg + cg * 8 * (int)sizeof(T)) = v;
        }
        __syncthreads();

        if constexpr (sizeof(T) == 2) {
            // ---- bf16: 8 blocks of 32 pixels; lane (column li, pixel group lg) needs pixels 8*lg..8*lg+7.
            // tr-read: lane 16*lg + 4*q + p supplies the address of row (pixel) q, columns 4p..4p+3 and
            // receives column li of the 4 rows.  EXEC stays all-ones through this section.
            typedef __attribute__((address_space(3))) bf16x4* lds_v4;
            const int q = li >> 2, p = li & 3;
            for (int blk = 0; blk < 8; ++blk) {
                const int m0 = blk * 32 + lg * 8 + q;             // first-read pixel of this lane's address
                const int m1 = m0 + 4;
                bf16x8 afrag[CTN];
#pragma unroll
                for (int ct = 0; ct < CTN; ++ct) {
                    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(lds_g + m0 * a.PSg + (ct * 16 + 4 * p) * 2));
                    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(lds_g + m1 * a.PSg + (ct * 16 + 4 * p) * 2));
                    afrag[ct] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
                const int pb0 = (((m0 >> 4) * d.stride) * a.TIW + (m0 & 15) * d.stride) * a.PSx;
                const int pb1 = (((m1 >> 4) * d.stride) * a.TIW + (m1 & 15) * d.stride) * a.PSx;
#pragma unroll
                for (int i = 0; i < NKW; ++i) {
                    const int e = tab[(wave + 4 * i) * 4 + p];        // k-tiles past NKT: zero columns
                    const int o0 = (e & TAB_ABS) ? (e & ~TAB_ABS) : pb0 + e;
                    const int o1 = (e & TAB_ABS) ? (e & ~TAB_ABS) : pb1 + e;
                    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(smem + o0));
                    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(smem + o1));
                    bf16x8 bfrag = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                    for (int ct = 0; ct < CTN; ++ct)
                        acc[i][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[ct], bfrag, acc[i][ct], 0, 0, 0);
                }
            }
        } else {
            // ---- fp32: 64 steps of 4 pixels; lane (li, lg) supplies pixel 4*step + lg
            int colofs[NKW];
#pragma unroll
            for (int i = 0; i < NKW; ++i) colofs[i] = tab[(wave + 4 * i) * 4 + (li >> 2)] + (li & 3) * 4;
            for (int step = 0; step < 64; ++step) {
                const int m = step * 4 + lg;
                float afrag[CTN];
#pragma unroll
                for (int ct = 0; ct < CTN; ++ct)
                    afrag[ct] = *reinterpret_cast<const float*>(lds_g + m * a.PSg + (ct * 16 + li) * 4);
                const int pb = (((m >> 4) * d.stride) * a.TIW + (m & 15) * d.stride) * a.PSx;
#pragma unroll
                for (int i = 0; i < NKW; ++i) {
                    const int e = colofs[i];
                    const int o = (e & TAB_ABS) ? (e & ~TAB_ABS) : pb + e;
                    const float bv = *reinterpret_cast<const float*>(smem + o);
#pragma unroll
                    for (int ct = 0; ct < CTN; ++ct)
                        acc[i][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[ct], bv, acc[i][ct], 0, 0, 0);
                }
            }
        }
    }

    // ---- write this workgroup's slab: slabs[blockIdx.x][chunk][co][k]
    float* slab = d.slabs + (((size_t)blockIdx.x * a.nchunks + chunk) * a.cout_total + a.co0) * a.kextc;
#pragma unroll
    for (int i = 0; i < NKW; ++i) {
        const int nkt = wave + 4 * i;
        if (nkt >= a.NKT) continue;
#pragma unroll
        for (int ct = 0; ct < CTN; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = ct * 16 + lg * 4 + j;
                if (co < d.Cout) slab[(size_t)co * a.kextc + nkt * 16 + li] = acc[i][ct][j];
            }
    }
}

template <typename T, int CTN, int NKW>
int launch_wgrad(hipStream_t s, const WArgs& a, dim3 grid, int lds) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<T, CTN, NKW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, MSAU_LDS_LIMIT);
        if (e != hipSuccess) return msau_set_error(MSAU_ERR_HIP, "wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL((wgrad_kernel<T, CTN, NKW>), grid, dim3(256), lds, s, a);
    MSAU_CHECK_LAUNCH("wgrad_kernel");
    return 0;
}

template <typename T, int CTN>
int launch_wgrad_nkw(hipStream_t s, const WArgs& a, int NKW, dim3 grid, int lds) {
    if constexpr (CTN == 8) {
        switch (NKW) {
            case 1: return launch_wgrad<T, CTN, 1>(s, a, grid, lds);
            case 2: return launch_wgrad<T, CTN, 2>(s, a, grid, lds);
            case 3: return launch_wgrad<T, CTN, 3>(s, a, grid, lds);
            default: return launch_wgrad<T, CTN, 5>(s, a, grid, lds);
        }
    } else {
        switch (NKW) {
            case 1: return launch_wgrad<T, CTN, 1>(s, a, grid, lds);
            case 2: return launch_wgrad<T, CTN, 2>(s, a, grid, lds);
            case 3: return launch_wgrad<T, CTN, 3>(s, a, grid, lds);
            case 5: return launch_wgrad<T, CTN, 5>(s, a, grid, lds);
            default: return launch_wgrad<T, CTN, 10>(s, a, grid, lds);
        }
    }
}

template <typename T>
int launch_wgrad_ct(hipStream_t s, const WArgs& a, int CTN, int NKW, dim3 grid, int lds) {
    switch (CTN) {
        case 1: return launch_wgrad_nkw<T, 1>(s, a, NKW, grid, lds);
        case 2: return launch_wgrad_nkw<T, 2>(s, a, NKW, grid, lds);
        case 3: case 4: return launch_wgrad_nkw<T, 4>(s, a, NKW, grid, lds);
        default: return launch_wgrad_nkw<T, 8>(s, a, NKW, grid, lds);
    }
}

}  // namespace

// wgrad_lean.hip: compile-time-specialised instances
int msau_wgrad_lean_applicable(int dtype, const msau_wgrad_desc* d, int cch);
int msau_wgrad_lean_try(hipStream_t s, int dtype, const msau_wgrad_desc* d, int cch, int nchunks, int kextc);
int msau_wgrad_lean_group(hipStream_t s, int dtype, const msau_wgrad_desc* const* ds, int n, int cch, int nchunks, int kextc);
int msau_wgrad_lean_groupable(int dtype, const msau_wgrad_desc* d, int cch);

extern "C" int msau_wgrad_geometry(int dtype, const msau_wgrad_desc* d, msau_wgrad_geom* out) {
    WGeom g;
    int rc = wgrad_geom(dtype, d, &g);
    if (rc) return rc;
    out->cch = g.cch; out->nchunks = g.nchunks; out->kext = g.kextc;
    out->max_slabs = d->B * cdiv(d->Hout, 16) * cdiv(d->Wout, 16);
    out->slab_bytes = (int64_t)g.nchunks * d->Cout * g.kextc * 4;
    out->lean = msau_wgrad_lean_applicable(dtype, d, g.cch);
    if (d->nslabs >= 1 && msau_rowwgrad_takes(dtype, d, g.cch, g.nchunks, g.kextc)) out->lean = 2;     // row-streaming instance
    out->reserved = 0;
    return 0;
}

extern "C" int msau_conv2d_wgrad(void* stream, int dtype, const msau_wgrad_desc* d) {
    MSAU_CHECK_ARG(d && d->x1 && d->g && d->slabs, "wgrad: null pointer");
    MSAU_CHECK_ARG(d->C2 == 0 || d->x2, "wgrad: x2 missing");
    WGeom g;
    int rc = wgrad_geom(dtype, d, &g);
    if (rc) return rc;
    if (d->flags & MSAU_CONV_OWNER)                                            // ownerconv.hip: per-box sums of g instead of a painted input tensor
        return msau_ownerconv_wgrad(static_cast<hipStream_t>(stream), dtype, d, g.cch, g.nchunks, g.kextc);
    const int tiles_x = cdiv(d->Wout, 16), tiles_y = cdiv(d->Hout, 16), ntiles = d->B * tiles_x * tiles_y;
    MSAU_CHECK_ARG(d->nslabs >= 1 && d->nslabs <= ntiles, "wgrad: nslabs %d not in [1,%d]", d->nslabs, ntiles);
    if (msau_rowwgrad_takes(dtype, d, g.cch, g.nchunks, g.kextc))              // conv_rows.hip: every row of x and g read once
        return msau_rowwgrad_launch(static_cast<hipStream_t>(stream), d);
    if (d->Cout <= g.slice) {
        rc = msau_wgrad_lean_try(static_cast<hipStream_t>(stream), dtype, d, g.cch, g.nchunks, g.kextc);
        if (rc != 0) return rc < 0 ? rc : 0;
    }
    MSAU_CHECK_ARG(!(d->flags & MSAU_CONV_IDS), "wgrad: MSAU_CONV_IDS is implemented by the bf16 64 -> 8 3x3 instance only");
    // CTN as instantiated (3 -> 4)
    int CTN = g.CTN == 3 ? 4 : (g.CTN > 4 ? 8 : g.CTN);
    MSAU_CHECK_ARG(!(CTN == 8 && g.NKW > 5), "wgrad: Cout %d with K %d unsupported", d->Cout, g.kextc);
    dim3 grid(d->nslabs, g.nchunks);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // wide outputs: one launch per slice; every slice writes its rows of the same slabs
    for (int co0 = 0; co0 < d->Cout; co0 += g.slice) {
        WArgs a;
        a.d = *d;
        a.d.Cout = d->Cout - co0 < g.slice ? d->Cout - co0 : g.slice;
        a.g_stride = d->Cout; a.co0 = co0; a.cout_total = d->Cout; a.compact = g.compact;
        a.cch = g.cch; a.nchunks = g.nchunks; a.kreal = g.kreal; a.kextc = g.kextc; a.NKT = g.NKT;
        a.TIH = g.TIH; a.TIW = g.TIW; a.PSx = g.PSx; a.PSg = g.PSg;
        a.x_bytes = g.x_bytes; a.g_bytes = g.g_bytes; a.tab_bytes = g.tab_bytes;
        a.tiles_x = tiles_x; a.tiles_y = tiles_y; a.ntiles = ntiles;
        rc = dtype == MSAU_F32 ? launch_wgrad_ct<float>(s, a, CTN, g.NKW, grid, g.total) : launch_wgrad_ct<bf16_t>(s, a, CTN, g.NKW, grid, g.total);
        if (rc) return rc;
    }
    return 0;
}

// ---- grouped launch: up to 4 weight gradients of one shape in one grid (include/msau_hip.h)
extern "C" int msau_owner_slabs(const msau_wgrad_desc* d) { return d ? msau_ownerconv_slabs(d) : 0; }

extern "C" int msau_conv2d_wgrad_groupable(int dtype, const msau_wgrad_desc* a, const msau_wgrad_desc* b) {
    if (!a || !b) return 0;
    WGeom g;
    if (wgrad_geom(dtype, a, &g)) return 0;
    if (msau_rowwgrad_takes(dtype, a, g.cch, g.nchunks, g.kextc)) return 0;        // the row kernel runs one layer per launch
    if (a->Cout > g.slice || !msau_wgrad_lean_groupable(dtype, a, g.cch)) return 0;
    return a->B == b->B && a->Hin == b->Hin && a->Win == b->Win && a->Hout == b->Hout && a->Wout == b->Wout && a->C1 == b->C1 &&
           a->C2 == b->C2 && a->Cout == b->Cout && a->KH == b->KH && a->KW == b->KW && a->dil == b->dil && a->pad_t == b->pad_t &&
           a->pad_l == b->pad_l && a->stride == b->stride && a->nslabs == b->nslabs;
}

extern "C" int msau_conv2d_wgrad_group(void* stream, int dtype, const msau_wgrad_desc* const* ds, int n) {
    MSAU_CHECK_ARG(ds && n >= 1 && n <= 4, "wgrad_group: 1..4 launches");
    for (int i = 0; i < n; ++i) {
        MSAU_CHECK_ARG(ds[i] && ds[i]->x1 && ds[i]->g && ds[i]->slabs && (ds[i]->C2 == 0 || ds[i]->x2), "wgrad_group: null pointer");
        MSAU_CHECK_ARG(i == 0 || msau_conv2d_wgrad_groupable(dtype, ds[0], ds[i]), "wgrad_group: launch %d differs in shape from launch 0", i);
    }
    if (n == 1) return msau_conv2d_wgrad(stream, dtype, ds[0]);
    WGeom g;
    int rc = wgrad_geom(dtype, ds[0], &g);
    if (rc) return rc;
    const int ntiles = ds[0]->B * cdiv(ds[0]->Wout, 16) * cdiv(ds[0]->Hout, 16);
    MSAU_CHECK_ARG(ds[0]->nslabs >= 1 && ds[0]->nslabs <= ntiles, "wgrad: nslabs %d not in [1,%d]", ds[0]->nslabs, ntiles);
    rc = msau_wgrad_lean_group(static_cast<hipStream_t>(stream), dtype, ds, n, g.cch, g.nchunks, g.kextc);
    if (rc == 1) return 0;
    if (rc < 0) return rc;
    return msau_set_error(MSAU_ERR_ARG, "wgrad_group: no shared-grid instance for this shape");
}
