// Implicit-GEMM convolution for gfx950: NHWC activations, a (4*PT) x 16 output-pixel tile per
// 256-thread workgroup, the input halo tile and the weight chunk staged once in LDS, MFMA
// 16x16x32 (bf16) / 16x16x4 (fp32) with output channels on the MFMA rows and pixels on the
// columns, so every lane ends up with 4 consecutive output channels of one pixel and the
// epilogue (bias, masks, residual, accumulate, ReLU) is vector loads/stores along C.
//
// Replaces torch.nn.Conv2d + utils.pad_2d (model/layers/layers.py:82-102,152-164), the concat
// feeding it (model/model.py:147,242,251), ConvTranspose2d (layers.py:249-250) and their data
// gradients.  See include/msau_hip.h for the exact arithmetic.
#include "msau_common.h"

namespace {

struct ConvGeom {
    int esz, taps, cch, nchunks, kchunk, rows, CT, ngroups;   // ngroups = kchunk / 8; rows / CT are those of ONE slice
    int nslices;                                              // output-channel slices of <= 128 rows (Cout > 128)
};

__host__ __device__ inline int pix_stride_bytes(int cch, int esz) {
    int ps = cch * esz;
    if (((ps >> 4) & 1) == 0) ps += 16;      // odd number of 16-B slots per pixel: conflict-free b128 reads
    return ps;
}

struct TileGeom { int TH, TIH, TIW, PS, WS, in_bytes, w_bytes, tab_bytes, total; };

inline bool conv_compact(int KH, int KW, int dil, int stride, int ups) { return stride == 1 && ups == 1 && dil >= 16 && KH * KW > 1; }

inline TileGeom tile_geom(const ConvGeom& g, int PT, int KH, int KW, int dil, int stride, int ups = 1) {
    TileGeom t;
    t.TH = 4 * PT;
    t.TIH = (t.TH - 1) * stride + (KH - 1) * dil + 1;
    t.TIW = 15 * stride + (KW - 1) * dil + 1;
    if (conv_compact(KH, KW, dil, stride, ups)) { t.TIH = KH * KW * t.TH; t.TIW = 16; }     // one TH x 16 block per tap
    t.PS = pix_stride_bytes(g.cch, g.esz);
    t.WS = g.kchunk * g.esz + 16;
    t.in_bytes = roundup(t.TIH * t.TIW * t.PS, 16);
    t.w_bytes = roundup(g.rows * t.WS, 16);
    t.tab_bytes = roundup(g.ngroups * 4, 16);
    t.total = t.in_bytes + t.w_bytes + t.tab_bytes;
    return t;
}

int conv_geom(int dtype, int C1, int C2, int Cout, int KH, int KW, int dil, int stride, int ups, ConvGeom* out) {
    const int Cin = C1 + C2;
    MSAU_CHECK_ARG(dtype == MSAU_F32 || dtype == MSAU_BF16, "conv: bad dtype %d", dtype);
    MSAU_CHECK_ARG(C1 > 0 && C1 % 8 == 0 && C2 >= 0 && C2 % 8 == 0, "conv: bad source channels (%d,%d)", C1, C2);
    MSAU_CHECK_ARG(Cin > 0 && Cin % 8 == 0 && Cout > 0 && Cout % 8 == 0, "conv: channels must be multiples of 8 (%d,%d)", Cin, Cout);
    MSAU_CHECK_ARG(KH >= 1 && KW >= 1 && KH <= 7 && KW <= 7 && dil >= 1, "conv: bad kernel %dx%d dil %d", KH, KW, dil);
    MSAU_CHECK_ARG((stride == 1 || stride == 2) && (ups == 1 || ups == 2) && !(stride == 2 && ups == 2), "conv: bad stride/ups");
    MSAU_CHECK_ARG(Cout <= 1024, "conv: Cout %d > 1024 unsupported", Cout);
    ConvGeom g;
    g.esz = dtype == MSAU_F32 ? 4 : 2;
    g.taps = KH * KW;
    // more than 128 output channels (8 accumulator tiles per wave): the launch is cut into slices of 128 channels, each
    // with its own 128-row sub-image [slice][chunk][row][k]; msau_conv2d runs one generic launch per slice, writing its
    // channels into the full-width output (the 256-channel levels of the reference's constructor defaults; the 3c-wide
    // data gradient of the box variant's 1x1 convs)
    g.nslices = Cout > 128 ? cdiv(Cout, 128) : 1;
    int ct = cdiv(g.nslices > 1 ? 128 : Cout, 16);
    g.CT = ct <= 1 ? 1 : ct <= 2 ? 2 : ct <= 4 ? 4 : 8;
    g.rows = g.CT * 16;
    int best = 0;
    // 64 -> 64 3x3 (level 3) in bf16: one K chunk, so that the compile-time-specialised instances (conv_lean.hip: one
    // chunk per source; channel-split over blockIdx.y for small images) take it.  The generic kernel still fits with
    // its smallest tile when the launch is too small for the lean path.
    if (g.esz == 2 && C2 == 0 && Cin == 64 && g.CT == 4 && KH == 3 && KW == 3 && dil == 1 && stride == 1 && ups == 1) {
        g.cch = 64;
        g.kchunk = roundup(g.taps * 64, 32);
        g.ngroups = g.kchunk / 8;
        if (tile_geom(g, 1, KH, KW, dil, stride, ups).total <= 150 * 1024) best = 64;
    }
    // 32 -> 64 3x3 with dilation 8 (the level-3 entry conv) in bf16: one K chunk as well -- its lean instance
    // (conv_lean.hip, lean_dil<8>: a 32 x 32 pixel tile, 120 KB of LDS) takes the launch instead of the generic kernel
    if (!best && g.esz == 2 && C2 == 0 && Cin == 32 && g.CT == 4 && KH == 3 && KW == 3 && dil == 8 && stride == 1 && ups == 1) {
        g.cch = 32;
        g.kchunk = roundup(g.taps * 32, 32);
        g.ngroups = g.kchunk / 8;
        if (tile_geom(g, 1, KH, KW, dil, stride, ups).total <= 150 * 1024) best = 32;
    }
    // the level-3 -> level-2 transposed conv 64 -> 32 (ups = 2) and its data gradient 32 -> 64 (stride = 2), bf16: one K chunk
    // too, for the strided / zero-stuffed lean instances (84 / 125 KB of LDS)
    if (!best && g.esz == 2 && C2 == 0 && KH == 3 && KW == 3 && dil == 1 &&
        ((Cin == 64 && g.CT == 2 && stride == 1 && ups == 2) || (Cin == 32 && g.CT == 4 && stride == 2 && ups == 1))) {
        g.cch = Cin;
        g.kchunk = roundup(g.taps * Cin, 32);
        g.ngroups = g.kchunk / 8;
        if (tile_geom(g, 1, KH, KW, dil, stride, ups).total <= 150 * 1024) best = Cin;
    }
    // 64 -> 32 3x3 with dilation 8 (the DATA GRADIENT of the level-3 entry conv), bf16: two chunks of 32 channels for the chunked
    // instance of conv_lean.hip (round 4; the generic choice was four chunks of 16 for the generic kernel: 0.2 TB/s)
    if (!best && g.esz == 2 && C2 == 0 && Cin == 64 && g.CT == 2 && KH == 3 && KW == 3 && dil == 8 && stride == 1 && ups == 1) {
        g.cch = 32;
        g.kchunk = roundup(g.taps * 32, 32);
        g.ngroups = g.kchunk / 8;
        if (tile_geom(g, 1, KH, KW, dil, stride, ups).total <= 150 * 1024) best = 32;
    }
    for (int pass = 0; pass < 2 && !best; ++pass) {
        for (int c = (Cin < 128 ? Cin : 128); c >= 8; c -= 8) {
            if (Cin % c) continue;
            if (C2 && C1 % c) continue;                      // a chunk never straddles the two sources
            g.cch = c;
            g.kchunk = roundup(g.taps * c, 32);
            g.ngroups = g.kchunk / 8;
            TileGeom t = tile_geom(g, pass == 0 ? 4 : 1, KH, KW, dil, stride, ups);
            if (t.total <= (pass == 0 ? 72 * 1024 : 150 * 1024)) { best = c; break; }
        }
    }
    if (!best) return msau_set_error(MSAU_ERR_LDS, "conv: no channel chunk fits LDS (Cin %d Cout %d k %dx%d dil %d)", Cin, Cout, KH, KW, dil);
    g.cch = best;
    g.nchunks = Cin / best;
    g.kchunk = roundup(g.taps * best, 32);
    g.ngroups = g.kchunk / 8;
    *out = g;
    return 0;
}

struct ConvArgs {
    msau_conv_desc d;
    int ystride;                                 // channels per pixel of the output tensor (>= d.Cout when the launch is a slice)
    int compact;                                 // 1: large dilation -- the LDS tile holds one TH x 16 pixel block per TAP instead of
                                                 //    the (mostly unused) halo rectangle: TIH = taps * TH, TIW = 16
    int cch, nchunks, kchunk, ngroups;
    int TIH, TIW, PS, WS;
    int in_bytes, w_bytes;
    int tiles_x, tiles_y;
    int ct_total, rows_total;                    // channel tiles / packed rows of the whole (slice of the) conv; a workgroup takes CT of
                                                 // them, blockIdx.y picks which (row split: see msau_conv2d)
};

template <typename T, int CT, int PT>
__global__ __launch_bounds__(256) void conv_kernel(const ConvArgs a) {
    typedef typename Vec8<T>::type V8;
    typedef typename Vec4<T>::type V4;
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* lds_in = smem;
    unsigned char* lds_w = smem + a.in_bytes;
    int* koff_tab = reinterpret_cast<int*>(smem + a.in_bytes + a.w_bytes);

    const msau_conv_desc& d = a.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    constexpr int TH = 4 * PT;
    const int ct0 = blockIdx.y * CT, CTT = a.ct_total;           // this workgroup's channel tiles [ct0, ct0 + CT) of CTT

    int bid = blockIdx.x;
    const int txi = bid % a.tiles_x; bid /= a.tiles_x;
    const int tyi = bid % a.tiles_y; bid /= a.tiles_y;
    const int b = bid;
    const int oy0 = tyi * TH, ox0 = txi * 16;
    const int vy0 = oy0 * d.stride - d.pad_t, vx0 = ox0 * d.stride - d.pad_l;
    const int cg_per_chunk = a.cch >> 3;
    const int taps = d.KH * d.KW;

    // k-group -> byte offset (relative to the lane's pixel) of its 8 channels in the LDS input tile
    for (int G = tid; G < a.ngroups; G += 256) {
        int off = 0;
        if (G < taps * cg_per_chunk) {
            int tap = G / cg_per_chunk, cg = G - tap * cg_per_chunk;
            int ky = tap / d.KW, kx = tap - ky * d.KW;
            off = ((ky * d.dil) * a.TIW + kx * d.dil) * a.PS + cg * 8 * (int)sizeof(T);
            if (a.compact) off = (tap * TH * 16) * a.PS + cg * 8 * (int)sizeof(T);
        }
        koff_tab[G] = off;
    }

    int pixoff[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        int ty = wave * PT + pt;
        pixoff[pt] = ((ty * d.stride) * a.TIW + lr * d.stride) * a.PS;
    }

    f32x4 acc[CT][PT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const T* x1 = static_cast<const T*>(d.x1);
    const T* x2 = static_cast<const T*>(d.x2);
    const T* wp = static_cast<const T*>(d.wpack);
    const bool relu_in = d.flags & MSAU_CONV_RELU_IN;
    const int npix_in = a.TIH * a.TIW;
    const int rows = CT * 16;
    const int wg_per_row = a.kchunk >> 3;
    const int nks = a.kchunk >> 5;

    for (int chunk = 0; chunk < a.nchunks; ++chunk) {
        if (chunk) __syncthreads();
        // ---- stage the input halo tile for this channel chunk (zero padding / zero stuffing here)
        for (int idx = tid; idx < npix_in * cg_per_chunk; idx += 256) {
            int pix = idx / cg_per_chunk, cg = idx - pix * cg_per_chunk;
            int iy = pix / a.TIW, ix = pix - iy * a.TIW;
            int vy = vy0 + iy, vx = vx0 + ix;
            if (a.compact) {                                   // pix = (tap, row in tile, column): gather that tap's pixel
                const int tap = iy / TH, py = iy - tap * TH;
                const int ky = tap / d.KW, kx = tap - ky * d.KW;
                vy = vy0 + py + ky * d.dil;
                vx = vx0 + ix + kx * d.dil;
            }
            bool ok = vy >= 0 && vx >= 0;
            int ry = vy, rx = vx;
            if (d.ups == 2) { ok = ok && !((vy | vx) & 1); ry = vy >> 1; rx = vx >> 1; }
            ok = ok && ry < d.Hin && rx < d.Win;
            V8 v = zero8<T>();
            if (ok) {
                int cs = chunk * a.cch + cg * 8;
                size_t p = ((size_t)b * d.Hin + ry) * d.Win + rx;
                const T* src = cs < d.C1 ? x1 + p * d.C1 + cs : x2 + p * d.C2 + (cs - d.C1);
                v = load8<T>(src);
                if (relu_in) v = relu8<T>(v);
            }
            *reinterpret_cast<V8*>(lds_in + pix * a.PS + cg * 8 * (int)sizeof(T)) = v;
        }
        // ---- stage this chunk's packed weights: global [chunk][row][kchunk] -> LDS rows of WS bytes
        const T* wsrc = wp + ((size_t)chunk * a.rows_total + ct0 * 16) * a.kchunk;
        for (int idx = tid; idx < rows * wg_per_row; idx += 256) {
            int r = idx / wg_per_row, g8 = idx - r * wg_per_row;
            *reinterpret_cast<V8*>(lds_w + r * a.WS + g8 * 8 * (int)sizeof(T)) = load8<T>(wsrc + (size_t)r * a.kchunk + g8 * 8);
        }
        __syncthreads();
        // ---- MFMA over this chunk's K
        for (int ks = 0; ks < nks; ++ks) {
            const int G = ks * 4 + lg;
            const int koff = koff_tab[G];
            V8 bfrag[PT];
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) bfrag[pt] = *reinterpret_cast<const V8*>(lds_in + pixoff[pt] + koff);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                V8 afrag = *reinterpret_cast<const V8*>(lds_w + (ct * 16 + lr) * a.WS + G * 8 * (int)sizeof(T));
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = mma8(afrag, bfrag[pt], acc[ct][pt]);
            }
        }
    }

    // ---- epilogue: lane (pixel lr of row ty, q = lg) owns channels q*CT*4 + ct*4 + {0..3}
    const int Cout = d.Cout;
    T* y = static_cast<T*>(d.y);
    const T* add = static_cast<const T*>(d.add);
    const T* mask_a = static_cast<const T*>(d.mask_a);
    const T* mask_b = static_cast<const T*>(d.mask_b);
    const int flags = d.flags;
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int oy = oy0 + wave * PT + pt, ox = ox0 + lr;
        if (oy >= d.Hout || ox >= d.Wout) continue;
        const size_t pbase = (((size_t)b * d.Hout + oy) * d.Wout + ox) * a.ystride;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int co = lg * (CTT * 4) + (ct0 + ct) * 4;
            if (co >= Cout) continue;
            f32x4 v = acc[ct][pt];
            if (d.bias) {
                f32x4 bv = *reinterpret_cast<const f32x4*>(d.bias + co);
                v += bv;
            }
            if (flags & MSAU_CONV_MASK_A) {
                V4 m = load4<T>(mask_a + pbase + co);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = ((float)m[j] > 0.f) ? v[j] : 0.f;
            }
            if (flags & MSAU_CONV_ADD) {
                V4 r = load4<T>(add + pbase + co);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += (float)r[j];
            }
            if (flags & MSAU_CONV_ACCUM) {
                V4 r = load4<T>(y + pbase + co);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += (float)r[j];
            }
            if (flags & MSAU_CONV_RELU_OUT) {
                if (flags & MSAU_CONV_ELU) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : expm1f(v[j]);          // torch.nn.ELU(alpha = 1)
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                }
            }
            if (flags & MSAU_CONV_MASK_B) {
                V4 m = load4<T>(mask_b + pbase + co);
                if (flags & MSAU_CONV_ELU) {                                  // d ELU / dz at the stored y = ELU(z): z > 0 ? 1 : exp(z) = y + 1
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = ((float)m[j] > 0.f) ? v[j] : v[j] * ((float)m[j] + 1.f);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = ((float)m[j] > 0.f) ? v[j] : 0.f;
                }
            }
            V4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (T)v[j];
            store4<T>(y + pbase + co, o);
        }
    }
}

template <typename T, int CT, int PT>
int launch_conv(hipStream_t s, const ConvArgs& a, int nblocks, int nsplit, int lds) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_kernel<T, CT, PT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, MSAU_LDS_LIMIT);
        if (e != hipSuccess) return msau_set_error(MSAU_ERR_HIP, "conv: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_kernel<T, CT, PT>), dim3(nblocks, nsplit), dim3(256), lds, s, a);
    MSAU_CHECK_LAUNCH("conv_kernel");
    return 0;
}

template <typename T, int CT>
int launch_conv_pt(hipStream_t s, const ConvArgs& a, int PT, int nblocks, int nsplit, int lds) {
    switch (PT) {
        case 4: return launch_conv<T, CT, 4>(s, a, nblocks, nsplit, lds);
        case 2: return launch_conv<T, CT, 2>(s, a, nblocks, nsplit, lds);
        default: return launch_conv<T, CT, 1>(s, a, nblocks, nsplit, lds);
    }
}

template <typename T>
int launch_conv_ct(hipStream_t s, const ConvArgs& a, int CT, int PT, int nblocks, int nsplit, int lds) {
    switch (CT) {
        case 1: return launch_conv_pt<T, 1>(s, a, PT, nblocks, nsplit, lds);
        case 2: return launch_conv_pt<T, 2>(s, a, PT, nblocks, nsplit, lds);
        case 4: return launch_conv_pt<T, 4>(s, a, PT, nblocks, nsplit, lds);
        default: return launch_conv_pt<T, 8>(s, a, PT, nblocks, nsplit, lds);
    }
}

}  // namespace

extern "C" int msau_conv_pack_geometry(int dtype, int C1, int C2, int Cout, int KH, int KW, int dil, int stride, int ups,
                                       msau_conv_pack_geom* out) {
    ConvGeom g;
    int rc = conv_geom(dtype, C1, C2, Cout, KH, KW, dil, stride, ups, &g);
    if (rc) return rc;
    out->cch = g.cch; out->nchunks = g.nchunks; out->kchunk = g.kchunk; out->rows = g.rows * g.nslices;
    out->bytes = (int64_t)g.nslices * g.nchunks * g.rows * g.kchunk * g.esz;
    return 0;
}

static int conv_plan(int dtype, const msau_conv_desc* d, ConvGeom* gout, TileGeom* tout, int* PTout, int64_t* nbout);
// conv_lean.hip: compile-time-specialised instances for the hot layer shapes
int msau_conv_lean_applicable(int dtype, const msau_conv_desc* d, int nchunks, int CT);
int msau_conv_lean_try(hipStream_t s, int dtype, const msau_conv_desc* d, int kchunk, int nchunks, int CT);
int msau_conv_lean_head_capable(int dtype, const msau_conv_desc* d, int nchunks, int CT);
int msau_conv_lean_dout_capable(int dtype, const msau_conv_desc* d, int nchunks, int CT);
int msau_conv_lean_lrn_capable(int dtype, const msau_conv_desc* d, int nchunks, int CT);
int msau_conv_lean_pool_capable(int dtype, const msau_conv_desc* d, int nchunks, int CT);
int msau_conv_lean_ids_capable(int dtype, const msau_conv_desc* d, int nchunks, int CT);
int msau_conv_chunked_try(hipStream_t s, int dtype, const msau_conv_desc* d, int cch, int kchunk, int nchunks, int CT);
int msau_conv_chunked_capable(int dtype, const msau_conv_desc* d, int cch, int nchunks, int CT);
int msau_firstconv_takes(int dtype, const msau_conv_desc* d);
int msau_firstconv_launch(hipStream_t s, int dtype, const msau_conv_desc* d, int real_channels);

extern "C" int msau_conv2d_launch_info(int dtype, const msau_conv_desc* d, int32_t* info) {
    MSAU_CHECK_ARG(d && info, "conv2d_launch_info: null pointer");
    ConvGeom g; TileGeom t; int PT; int64_t nb;
    int rc = conv_plan(dtype, d, &g, &t, &PT, &nb);
    if (rc) return rc;
    info[0] = g.CT; info[1] = PT; info[2] = t.total; info[3] = (int32_t)nb; info[4] = g.cch; info[5] = g.nchunks;
    info[6] = msau_conv_lean_applicable(dtype, d, g.nchunks, g.CT);
    if (g.nslices == 1 && msau_rowconv_takes(dtype, d)) info[6] = 3;                         // rowconv8_kernel (conv_rows.hip)
    if (g.nslices == 1 && !(d->flags & (MSAU_CONV_HEAD | MSAU_CONV_DOUT | MSAU_CONV_LRN | MSAU_CONV_POOL | MSAU_CONV_IDS)) &&
        msau_conv_chunked_capable(dtype, d, g.cch, g.nchunks, g.CT)) info[6] = 2;       // conv_chunked_kernel (conv_lean.hip)
    info[7] = msau_conv_lean_head_capable(dtype, d, g.nchunks, g.CT) | (msau_conv_lean_dout_capable(dtype, d, g.nchunks, g.CT) << 1) |
              (((d->flags & MSAU_CONV_DOUT) && g.nslices == 1 && msau_rowconv_takes(dtype, d)) << 1) |
              (msau_conv_lean_lrn_capable(dtype, d, g.nchunks, g.CT) << 2) | (msau_conv_lean_pool_capable(dtype, d, g.nchunks, g.CT) << 3) |
              (msau_conv_lean_ids_capable(dtype, d, g.nchunks, g.CT) << 4);
    {
        msau_conv_desc p = *d;
        p.flags |= MSAU_CONV_OWNER;
        if (msau_ownerconv_takes(dtype, &p)) info[7] |= 32;
    }
    {
        msau_conv_desc p = *d;
        p.flags |= MSAU_CONV_NCHW;
        if (msau_firstconv_takes(dtype, &p)) info[7] |= 64;
    }
    if (g.nslices == 1 && !(info[7] & 4)) {              // would a row-streaming instance take this launch with MSAU_CONV_LRN added?
        msau_conv_desc p = *d;
        p.flags |= MSAU_CONV_LRN;
        if (!p.y2) p.y2 = p.y;
        if (!(p.lrn_k > 0.f)) p.lrn_k = 1.f;
        if (msau_rowconv_takes(dtype, &p)) info[7] |= 4;
    }
    return 0;
}

static int conv_plan(int dtype, const msau_conv_desc* d, ConvGeom* gout, TileGeom* tout, int* PTout, int64_t* nbout) {
    ConvGeom g;
    int rc = conv_geom(dtype, d->C1, d->C2, d->Cout, d->KH, d->KW, d->dil, d->stride, d->ups, &g);
    if (rc) return rc;
    int PT = 4;
    auto ntiles = [&](int pt) { return (int64_t)d->B * cdiv(d->Hout, 4 * pt) * cdiv(d->Wout, 16); };
    if (ntiles(4) < 512) PT = ntiles(2) >= 384 ? 2 : 1;
    TileGeom t = tile_geom(g, PT, d->KH, d->KW, d->dil, d->stride, d->ups);
    while (t.total > 150 * 1024 && PT > 1) { PT >>= 1; t = tile_geom(g, PT, d->KH, d->KW, d->dil, d->stride, d->ups); }
    if (t.total > 150 * 1024) return msau_set_error(MSAU_ERR_LDS, "conv2d: tile needs %d B of LDS", t.total);
    int64_t nb = ntiles(PT);
    MSAU_CHECK_ARG(nb < (1ll << 31), "conv2d: grid too large");
    *gout = g; *tout = t; *PTout = PT; *nbout = nb;
    return 0;
}

extern "C" int msau_conv2d(void* stream, int dtype, const msau_conv_desc* d) {
    MSAU_CHECK_ARG(d && d->x1 && d->wpack && d->y, "conv2d: null pointer");
    MSAU_CHECK_ARG(d->B > 0 && d->Hin > 0 && d->Win > 0 && d->Hout > 0 && d->Wout > 0, "conv2d: bad dims");
    MSAU_CHECK_ARG(d->C1 % 8 == 0 && d->C2 % 8 == 0 && (d->C2 == 0 || d->x2), "conv2d: bad sources");
    MSAU_CHECK_ARG(!(d->flags & MSAU_CONV_ADD) || d->add, "conv2d: ADD without pointer");
    MSAU_CHECK_ARG(!(d->flags & MSAU_CONV_MASK_A) || d->mask_a, "conv2d: MASK_A without pointer");
    MSAU_CHECK_ARG(!(d->flags & MSAU_CONV_MASK_B) || d->mask_b, "conv2d: MASK_B without pointer");
    if (d->flags & MSAU_CONV_WGRAD)                                            // a rider only the row-streaming coupling instance carries
        MSAU_CHECK_ARG(msau_conv2d_rider_slabs(dtype, d) > 0, "conv2d: MSAU_CONV_WGRAD is not implemented for this launch (msau_conv2d_rider_slabs says 0)");
    if (d->flags & MSAU_CONV_OWNER) {                                          // ownerconv.hip: box lists instead of a painted input tensor
        MSAU_CHECK_ARG(msau_ownerconv_takes(dtype, d), "conv2d: MSAU_CONV_OWNER is the 3x3 stride-1 C -> 8 conv, no other flag but RELU_OUT");
        return msau_ownerconv_fwd(static_cast<hipStream_t>(stream), dtype, d);
    }
    if (d->flags & MSAU_CONV_NCHW) {                                           // conv_first.hip: the fp32 NCHW input tensor itself
        MSAU_CHECK_ARG(msau_firstconv_takes(dtype, d) && d->head_classes > 0 && d->head_classes <= d->C1,
                       "conv2d: MSAU_CONV_NCHW is the bf16 3x3 64 -> 8 conv (W %% 4 == 0, W <= 288), head_classes = real input channels");
        return msau_firstconv_launch(static_cast<hipStream_t>(stream), dtype, d, d->head_classes);
    }
    ConvGeom g; TileGeom t; int PT; int64_t nb;
    int rc = conv_plan(dtype, d, &g, &t, &PT, &nb);
    if (rc) return rc;
    if (d->flags & MSAU_CONV_HEAD) {
        MSAU_CHECK_ARG(d->head_probs && d->head_argmax && d->head_classes > 0 && d->head_classes <= 16 &&
                       d->head_classes <= d->Cout && d->flags == MSAU_CONV_HEAD, "conv2d: bad HEAD arguments");
        if (!msau_conv_lean_head_capable(dtype, d, g.nchunks, g.CT))
            return msau_set_error(MSAU_ERR_ARG, "conv2d: MSAU_CONV_HEAD is not implemented for this launch (see "
                                  "msau_conv2d_launch_info info[7]); run msau_softmax_argmax_nhwc on y instead");
    }
    if (d->flags & MSAU_CONV_DOUT) {
        const int okf = MSAU_CONV_DOUT | MSAU_CONV_ADD | MSAU_CONV_ACCUM | MSAU_CONV_MASK_B | MSAU_CONV_WGRAD;   // (WGRAD: checked above)
        MSAU_CHECK_ARG(d->y2 && !(d->flags & ~okf) && !(d->flags2 & ~(MSAU_CONV_ACCUM | MSAU_CONV_MASK_B)) &&
                       (!(d->flags2 & MSAU_CONV_MASK_B) || d->mask_b2), "conv2d: bad DOUT arguments");
        if (!msau_conv_lean_dout_capable(dtype, d, g.nchunks, g.CT) && !(g.nslices == 1 && msau_rowconv_takes(dtype, d)))
            return msau_set_error(MSAU_ERR_ARG, "conv2d: MSAU_CONV_DOUT is not implemented for this launch (see "
                                  "msau_conv2d_launch_info info[7]); issue one launch per output instead");
    }
    if (d->flags & MSAU_CONV_LRN) {
        MSAU_CHECK_ARG(d->y2 && !(d->flags & ~(MSAU_CONV_LRN | MSAU_CONV_RELU_IN)) && d->lrn_k > 0.f, "conv2d: bad LRN arguments");
        if (!msau_conv_lean_lrn_capable(dtype, d, g.nchunks, g.CT) && !(g.nslices == 1 && msau_rowconv_takes(dtype, d)))
            return msau_set_error(MSAU_ERR_ARG, "conv2d: MSAU_CONV_LRN is not implemented for this launch (see "
                                  "msau_conv2d_launch_info info[7]); run msau_lrn_fwd on y instead");
    }
    if (d->flags & MSAU_CONV_IDS) {
        if (!msau_conv_lean_ids_capable(dtype, d, g.nchunks, g.CT))
            return msau_set_error(MSAU_ERR_ARG, "conv2d: MSAU_CONV_IDS is not implemented for this launch (see msau_conv2d_launch_info "
                                  "info[7] & 16); paint the one-hot input with msau_onehot_ids instead");
    }
    if (d->flags & MSAU_CONV_POOL) {
        MSAU_CHECK_ARG(d->pool_y && !(d->flags & (MSAU_CONV_DOUT | MSAU_CONV_HEAD | MSAU_CONV_LRN)), "conv2d: bad POOL arguments");
        if (!msau_conv_lean_pool_capable(dtype, d, g.nchunks, g.CT))
            return msau_set_error(MSAU_ERR_ARG, "conv2d: MSAU_CONV_POOL is not implemented for this launch (see "
                                  "msau_conv2d_launch_info info[7]); run msau_maxpool2x2_fwd on y instead");
    }
    if (g.nslices == 1 && msau_rowconv_takes(dtype, d))                       // conv_rows.hip: row-streaming instances (level 0)
        return msau_rowconv_launch(static_cast<hipStream_t>(stream), dtype, d, g.kchunk, g.rows);
    if (g.nslices == 1 && !(d->flags & (MSAU_CONV_HEAD | MSAU_CONV_DOUT | MSAU_CONV_LRN | MSAU_CONV_POOL | MSAU_CONV_IDS))) {
        rc = msau_conv_chunked_try(static_cast<hipStream_t>(stream), dtype, d, g.cch, g.kchunk, g.nchunks, g.CT);
        if (rc != 0) return rc < 0 ? rc : 0;
    }
    if (g.nslices == 1) {
        rc = msau_conv_lean_try(static_cast<hipStream_t>(stream), dtype, d, g.kchunk, g.nchunks, g.CT);
        if (rc != 0) return rc < 0 ? rc : 0;
    }
    if (d->flags & MSAU_CONV_DOUT) return msau_set_error(MSAU_ERR_ARG, "conv2d: DOUT launch was not taken by a lean instance");
    MSAU_CHECK_ARG(g.nslices == 1 || !(d->flags & MSAU_CONV_HEAD), "conv2d: HEAD with more than 128 output channels");
    hipStream_t s = static_cast<hipStream_t>(stream);
    // row split: with few pixel tiles (the 21 x 16 and 11 x 8 pixel levels of the reference's constructor defaults: 48-96
    // tiles for 256 CUs) a workgroup takes only `cts` of the CT channel tiles and blockIdx.y the rest -- every workgroup
    // staged all 128 rows of every K chunk (147 KB per chunk at 64 channels x 3x3) for 16 pixels of work per wave: 60 us per
    // launch for a 0.3 MB tensor.  The packed image and the channel order do not change.
    static const int split_wgs = getenv("MSAU_CONV_ROWSPLIT") ? atoi(getenv("MSAU_CONV_ROWSPLIT")) : 768;    // workgroups aimed for; 0 = off
    int cts = g.CT;
    while (cts > 1 && nb * g.nslices * (g.CT / cts) < split_wgs) cts >>= 1;
    const int w_bytes = roundup(cts * 16 * t.WS, 16), lds_total = t.in_bytes + w_bytes + t.tab_bytes;
    for (int sl = 0; sl < g.nslices; ++sl) {
        ConvArgs a;
        a.d = *d;
        a.ystride = d->Cout;
        a.compact = conv_compact(d->KH, d->KW, d->dil, d->stride, d->ups);
        if (g.nslices > 1) {                                  // this slice's rows / channels [128*sl, 128*sl + n)
            const size_t co0 = (size_t)sl * 128, esz = g.esz;
            a.d.Cout = d->Cout - (int)co0 < 128 ? d->Cout - (int)co0 : 128;
            a.d.wpack = static_cast<const char*>(d->wpack) + (size_t)sl * g.nchunks * g.rows * g.kchunk * esz;
            if (d->bias) a.d.bias = d->bias + co0;
            a.d.y = static_cast<char*>(d->y) + co0 * esz;
            if (d->add) a.d.add = static_cast<const char*>(d->add) + co0 * esz;
            if (d->mask_a) a.d.mask_a = static_cast<const char*>(d->mask_a) + co0 * esz;
            if (d->mask_b) a.d.mask_b = static_cast<const char*>(d->mask_b) + co0 * esz;
        }
        a.cch = g.cch; a.nchunks = g.nchunks; a.kchunk = g.kchunk; a.ngroups = g.ngroups;
        a.TIH = t.TIH; a.TIW = t.TIW; a.PS = t.PS; a.WS = t.WS; a.in_bytes = t.in_bytes; a.w_bytes = w_bytes;
        a.tiles_x = cdiv(d->Wout, 16); a.tiles_y = cdiv(d->Hout, 4 * PT);
        a.ct_total = g.CT; a.rows_total = g.rows;
        rc = dtype == MSAU_F32 ? launch_conv_ct<float>(s, a, cts, PT, (int)nb, g.CT / cts, lds_total)
                               : launch_conv_ct<bf16_t>(s, a, cts, PT, (int)nb, g.CT / cts, lds_total);
        if (rc) return rc;
    }
    return 0;
}
