// HBM-bound kernels of the MSAU train path: boundary layout conversion, cross-channel LRN,
// 2x2 max pool, masked cross entropy, clip+Adam.  All vectorised 16 B per lane along the
// channel (NHWC) dimension; accumulation in fp32.
#include "msau_common.h"
#include <cstdlib>

namespace {

constexpr int kThreads = 256;

inline int grid_for(int64_t work, int cap = 256 * 16) {
    int64_t b = cdiv64(work, kThreads);
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

// =============================================================================================
// layout: NCHW fp32 (reference API, train_chargrid_funsd_msau.py:50-53) <-> NHWC T
// =============================================================================================
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int B, int C, int Cs, int64_t HW) {
    const int cgs = Cs >> 3;
    const int64_t total = (int64_t)B * cgs * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t p = i % HW;                      // pixel fastest: coalesced plane reads
        int64_t r = i / HW;
        int cg = (int)(r % cgs);
        int b = (int)(r / cgs);
        typename Vec8<T>::type v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int c = cg * 8 + j;
            v[j] = (T)(c < C ? src[((int64_t)b * C + c) * HW + p] : 0.f);
        }
        store8<T>(dst + ((int64_t)b * HW + p) * Cs + cg * 8, v);
    }
}

// Tiled form for wide inputs: a workgroup moves 64 pixels x all channels through LDS, so that the plane reads are
// 256-byte runs and every NHWC pixel (Cs * sizeof(T) bytes) is written as whole lines by neighbouring lanes.
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_tiled_kernel(const float* __restrict__ src, T* __restrict__ dst,
                                                                 int C, int Cs, int64_t HW, int tiles_per_img) {
    extern __shared__ __align__(16) unsigned char tsm[];
    T* tile = reinterpret_cast<T*>(tsm);                       // [64][Cs + 8]
    const int PS = Cs + 8;
    const int b = blockIdx.x / tiles_per_img;
    const int64_t p0 = (int64_t)(blockIdx.x - b * tiles_per_img) * 64;
    const int px = threadIdx.x & 63, c0 = threadIdx.x >> 6;
    const bool pv = p0 + px < HW;
    for (int c = c0; c < Cs; c += 4) {
        float v = 0.f;
        if (pv && c < C) v = src[((int64_t)b * C + c) * HW + p0 + px];
        tile[px * PS + c] = (T)v;
    }
    __syncthreads();
    const int cgs = Cs >> 3;
    for (int i = threadIdx.x; i < 64 * cgs; i += 256) {
        const int q = i / cgs, cg = i - q * cgs;
        if (p0 + q < HW)
            store8<T>(dst + ((int64_t)b * HW + p0 + q) * Cs + cg * 8, *reinterpret_cast<const typename Vec8<T>::type*>(tile + q * PS + cg * 8));
    }
}

// Register form (HW % 4 == 0): a thread owns 4 consecutive pixels x 8 consecutive channels -- eight 16-byte plane loads in
// flight, an 8 x 4 transpose in registers, four 16-byte (bf16) pixel stores; the Cs/8 lanes of a pixel quad write whole
// pixels (Cs * sizeof(T) contiguous bytes), lanes Cs/8 apart read 16-byte neighbours of one plane.  No LDS, no barrier.
template <typename T, bool WIDE>
__global__ __launch_bounds__(256) void nchw_to_nhwc_reg_kernel(const float* __restrict__ src, T* __restrict__ dst,
                                                               int C, int Cs, int64_t HW, int64_t total) {
    const int cgs = Cs >> 3;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;        // (image, pixel quad, channel group), group fastest
    if (i >= total) return;
    const int64_t quads = HW >> 2;
    int cg;
    int64_t r;
    if (WIDE) {
        // many channels (cgs % 8 == 0, quads % 8 == 0): a wave owns 8 quads x 8 groups, so that its eight lanes of one
        // plane read a whole 128-byte line; with the group fastest across all of Cs those eight quads sit in eight
        // different waves (three blocks at 768 channels) and every XCD's L2 fetches the line for itself
        const int64_t w = i >> 6;
        const int lane = (int)(i & 63);
        const int cgb = cgs >> 3;
        cg = (int)(w % cgb) * 8 + (lane >> 3);
        r = (w / cgb) * 8 + (lane & 7);
    } else {
        cg = (int)(i % cgs);
        r = i / cgs;
    }
    const int64_t b = r / quads, q = r - b * quads;
    f32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cg * 8 + j;
        v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < C) v[j] = *reinterpret_cast<const f32x4*>(src + ((int64_t)b * C + c) * HW + q * 4);
    }
    T* out = dst + ((int64_t)b * HW + q * 4) * Cs + cg * 8;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        typename Vec8<T>::type o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (T)v[j][p];
        store8<T>(out + (int64_t)p * Cs, o);
    }
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst, int B, int C, int Cs, int64_t HW) {
    const int cgs = Cs >> 3;
    const int64_t total = (int64_t)B * cgs * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t p = i % HW;
        int64_t r = i / HW;
        int cg = (int)(r % cgs);
        int b = (int)(r / cgs);
        typename Vec8<T>::type v = load8<T>(src + ((int64_t)b * HW + p) * Cs + cg * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int c = cg * 8 + j;
            if (c < C) dst[((int64_t)b * C + c) * HW + p] = (float)v[j];
        }
    }
}

template <typename T>
__global__ void nchw_grad_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int B, int C, int Cs,
                                         int64_t HW, int accumulate) {
    const int cgs = Cs >> 3;
    const int64_t total = (int64_t)B * cgs * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t p = i % HW;
        int64_t r = i / HW;
        int cg = (int)(r % cgs);
        int b = (int)(r / cgs);
        T* q = dst + ((int64_t)b * HW + p) * Cs + cg * 8;
        typename Vec8<T>::type v = accumulate ? load8<T>(q) : zero8<T>();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int c = cg * 8 + j;
            float a = accumulate ? (float)v[j] : 0.f;
            v[j] = (T)(c < C ? a + src[((int64_t)b * C + c) * HW + p] : 0.f);
        }
        store8<T>(q, v);
    }
}

// =============================================================================================
// LRN across channels.  Fast path: C == Cs == n == 8*G, G a power of two: G lanes per pixel,
// each holding 8 channels; window sums are differences of prefix sums, fetched from the lane
// G/2 away with one xor-shuffle per channel.  Generic path: one thread per pixel.
//   forward window of c : [c - n/2, c + (n-1)/2]         (torch.nn.LocalResponseNorm)
//   adjoint window of j : [j - (n-1)/2, j + n/2]
// =============================================================================================
__device__ __forceinline__ float pow_neg_beta(float d, float beta, bool beta075) {
    if (beta075) { float r = rsqrtf(d); return r * sqrtf(r); }       // d^-0.75, ~2 ulp, 3 VALU ops
    return __expf(-beta * __logf(d));
}

// F075: beta == 0.75 with the raw rsq / sqrt path compiled in alone (the reference's only setting); otherwise `powmode`
// picks at run time (0 exp/log, 1 rsq/sqrt)
template <typename T, int G, bool BWD, bool F075>
__global__ void lrn_fast_kernel(const T* __restrict__ a, const T* __restrict__ dy, T* __restrict__ out,
                                int64_t npix, float alpha_over_n, float beta, float k, int powmode_rt) {
    const int powmode = F075 ? 1 : powmode_rt;
    const int64_t total = npix * G;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // all lanes of a G-group run the same number of iterations (total % G == 0, stride % G == 0)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int l = (int)(i % G);
        typename Vec8<T>::type av = load8<T>(a + i * 8);
        float x[8], sq[8], P[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { x[j] = (float)av[j]; sq[j] = x[j] * x[j]; }
        float win[8];
        if constexpr (G == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < 8; ++c) if (c >= j - 4 && c <= j + 3) s += sq[c];
                win[j] = s;
            }
        } else {
            float run = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { run += sq[j]; P[j] = run; }
            float E = 0.f, tot;
            {   // exclusive scan of the lane totals across the G lanes of the pixel
                float t = run, inc = run;
#pragma unroll
                for (int o = 1; o < G; o <<= 1) { float n = __shfl_up(inc, o, G); if (l >= o) inc += n; }
                E = inc - t;
                tot = __shfl(inc, G - 1, G);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float X = E + (j ? P[j - 1] : 0.f);            // exclusive prefix at channel 8l+j
                float Xo = __shfl_xor(X, G / 2, G);
                win[j] = (l < G / 2) ? Xo : tot - Xo;
            }
        }
        // d^-beta and (backward) 1/d without an IEEE division: v_div_scale / v_div_fmas are VCC-dependent multi-instruction
        // sequences; rsq / rcp are single transcendental ops (d >= k = 1, so no range problems)
        float dnb[8], invd[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float d = k + alpha_over_n * win[j];
            // raw v_rsq_f32 / v_sqrt_f32 (1 ulp each, d >= k > 0: no denormal or range fix-ups needed) instead of the
            // library forms, whose correction sequences are a dozen VCC-dependent instructions per call
            if (powmode == 1) { const float r = __builtin_amdgcn_rsqf(d); dnb[j] = r * __builtin_amdgcn_sqrtf(r); invd[j] = r * r; }
            else { dnb[j] = __expf(-beta * __logf(d)); invd[j] = __builtin_amdgcn_rcpf(d); }
        }
        typename Vec8<T>::type ov;
        if constexpr (!BWD) {
#pragma unroll
            for (int j = 0; j < 8; ++j) ov[j] = (T)(x[j] * dnb[j]);
        } else {
            typename Vec8<T>::type gv = load8<T>(dy + i * 8);
            float g[8], qv[8], adj[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { g[j] = (float)gv[j]; qv[j] = g[j] * x[j] * dnb[j] * invd[j]; }
            if constexpr (G == 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float s = 0.f;
#pragma unroll
                    for (int c = 0; c < 8; ++c) if (c >= j - 3 && c <= j + 4) s += qv[c];
                    adj[j] = s;
                }
            } else {
                float run = 0.f, Q[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { run += qv[j]; Q[j] = run; }
                float t = run, inc = run;
#pragma unroll
                for (int o = 1; o < G; o <<= 1) { float n = __shfl_up(inc, o, G); if (l >= o) inc += n; }
                float E = inc - t;
                float tot = __shfl(inc, G - 1, G);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float I = E + Q[j];                         // inclusive prefix at channel 8l+j
                    float Io = __shfl_xor(I, G / 2, G);
                    adj[j] = (l < G / 2) ? Io : tot - Io;
                }
            }
            const float c2 = 2.f * beta * alpha_over_n;
#pragma unroll
            for (int j = 0; j < 8; ++j) ov[j] = (T)(g[j] * dnb[j] - c2 * x[j] * adj[j]);
        }
        store8<T>(out + i * 8, ov);
    }
}

template <typename T, bool BWD>
__global__ void lrn_generic_kernel(const T* __restrict__ a, const T* __restrict__ dy, T* __restrict__ out,
                                   int64_t npix, int C, int Cs, int n, float alpha_over_n, float beta, float k, bool beta075) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
        float x[128], q[128];
        for (int c = 0; c < Cs; c += 8) {
            typename Vec8<T>::type v = load8<T>(a + p * Cs + c);
            for (int j = 0; j < 8; ++j) x[c + j] = (float)v[j];
        }
        if (BWD)
            for (int c = 0; c < Cs; c += 8) {
                typename Vec8<T>::type v = load8<T>(dy + p * Cs + c);
                for (int j = 0; j < 8; ++j) q[c + j] = (float)v[j];      // q holds dy for now
            }
        float dnb[128], dd[128];
        for (int c = 0; c < C; ++c) {
            float s = 0.f;
            int lo = c - n / 2, hi = c + (n - 1) / 2;
            for (int cc = (lo < 0 ? 0 : lo); cc <= hi && cc < C; ++cc) s += x[cc] * x[cc];
            dd[c] = k + alpha_over_n * s;
            dnb[c] = pow_neg_beta(dd[c], beta, beta075);
        }
        for (int c0 = 0; c0 < Cs; c0 += 8) {
            typename Vec8<T>::type ov;
            for (int j = 0; j < 8; ++j) {
                int c = c0 + j;
                float o = 0.f;
                if (c < C) {
                    if (!BWD) o = x[c] * dnb[c];
                    else {
                        float s = 0.f;
                        int lo = c - (n - 1) / 2, hi = c + n / 2;
                        for (int cc = (lo < 0 ? 0 : lo); cc <= hi && cc < C; ++cc) s += q[cc] * x[cc] * dnb[cc] / dd[cc];
                        o = q[c] * dnb[c] - 2.f * beta * alpha_over_n * x[c] * s;
                    }
                }
                ov[j] = (T)o;
            }
            store8<T>(out + p * Cs + c0, ov);
        }
    }
}

template <typename T, bool BWD>
int lrn_dispatch(hipStream_t s, const void* a, const void* dy, void* out, int64_t npix, int C, int Cs, int n,
                 float alpha, float beta, float k) {
    MSAU_CHECK_ARG(a && out && (!BWD || dy), "lrn: null pointer");
    MSAU_CHECK_ARG(npix > 0 && C > 0 && C <= Cs && Cs % 8 == 0 && Cs <= 256 && n >= 1, "lrn: bad dims C=%d Cs=%d n=%d", C, Cs, n);
    const float aon = alpha / (float)n;
    const bool b075 = beta == 0.75f;
    const int powmode = b075 ? 1 : 0;
    const T* ap = static_cast<const T*>(a);
    const T* gp = static_cast<const T*>(dy);
    T* op = static_cast<T*>(out);
    const int G = Cs / 8;
    const bool fast = (C == Cs) && (n == C) && ((G & (G - 1)) == 0) && G <= 32;
    MSAU_CHECK_ARG(fast || Cs <= 128, "lrn: the generic path holds at most 128 channels per thread (C=%d n=%d)", C, n);
    if (fast) {
        int grid = grid_for(npix * G);
        switch (G) {
            case 1: if (powmode == 1) hipLaunchKernelGGL((lrn_fast_kernel<T, 1, BWD, true>), dim3(grid), dim3(kThreads), 0, s, ap, gp, op, npix, aon, beta, k, powmode);
                    else hipLaunchKernelGGL((lrn_fast_kernel<T, 1, BWD, false>), dim3(grid), dim3(kThreads), 0, s, ap, gp, op, npix, aon, beta, k, powmode); break;
            case 2: if (powmode == 1) hipLaunchKernelGGL((lrn_fast_kernel<T, 2, BWD, true>), dim3(grid), dim3(kThreads), 0, s, ap, gp, op, npix, aon, beta, k, powmode);
                    else hipLaunchKernelGGL((lrn_fast_kernel<T, 2, BWD, false>), dim3(grid), dim3(kThreads), 0, s, ap, gp, op, npix, aon, beta, k, powmode); break;
            case 4: if (powmode == 1) hipLaunchKernelGGL((lrn_fast_kernel<T, 4, BWD, true>), dim3(grid), dim3(kThreads), 0, s, ap, gp, op, npix, aon, beta, k, powmode);
                    else hipLaunchKernelGGL((lrn_fast_kernel<T, 4, BWD, false>), dim3(grid), dim3(kThreads), 0, s, ap, gp, op, npix, aon, beta, k, powmode); break;
            case 8: if (powmode == 1) hipLaunchKernelGGL((lrn_fast_kernel<T, 8, BWD, true>), dim3(grid), dim3(kThreads), 0, s, ap, gp, op, npix, aon, beta, k, powmode);
                    else hipLaunchKernelGGL((lrn_fast_kernel<T, 8, BWD, false>), dim3(grid), dim3(kThreads), 0, s, ap, gp, op, npix, aon, beta, k, powmode); break;
            case 16: if (powmode == 1) hipLaunchKernelGGL((lrn_fast_kernel<T, 16, BWD, true>), dim3(grid), dim3(kThreads), 0, s, ap, gp, op, npix, aon, beta, k, powmode);
                    else hipLaunchKernelGGL((lrn_fast_kernel<T, 16, BWD, false>), dim3(grid), dim3(kThreads), 0, s, ap, gp, op, npix, aon, beta, k, powmode); break;
            default: if (powmode == 1) hipLaunchKernelGGL((lrn_fast_kernel<T, 32, BWD, true>), dim3(grid), dim3(kThreads), 0, s, ap, gp, op, npix, aon, beta, k, powmode);
                     else hipLaunchKernelGGL((lrn_fast_kernel<T, 32, BWD, false>), dim3(grid), dim3(kThreads), 0, s, ap, gp, op, npix, aon, beta, k, powmode); break;
        }
    } else {
        hipLaunchKernelGGL((lrn_generic_kernel<T, BWD>), dim3(grid_for(npix)), dim3(kThreads), 0, s, ap, gp, op, npix, C, Cs, n, aon, beta, k, b075);
    }
    MSAU_CHECK_LAUNCH("lrn_kernel");
    return 0;
}

// =============================================================================================
// 2x2/2 max pool after zero SAME padding (bottom/right only, for odd sizes): model/model.py:158-160
// =============================================================================================
template <typename T>
__global__ void pool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ idx,
                                int B, int H, int W, int Ho, int Wo, int Cs) {
    const int cgs = Cs >> 3;
    const int64_t total = (int64_t)B * Ho * Wo * cgs;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int cg = (int)(i % cgs);
        int64_t p = i / cgs;
        int ox = (int)(p % Wo); p /= Wo;
        int oy = (int)(p % Ho);
        int b = (int)(p / Ho);
        float best[8]; int bi[8];
#pragma unroll
        for (int pos = 0; pos < 4; ++pos) {
            int iy = 2 * oy + (pos >> 1), ix = 2 * ox + (pos & 1);
            typename Vec8<T>::type v = zero8<T>();                // the pad value is 0, not -inf
            if (iy < H && ix < W) v = load8<T>(x + (((int64_t)b * H + iy) * W + ix) * Cs + cg * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float f = (float)v[j];
                if (pos == 0 || f > best[j]) { best[j] = f; bi[j] = pos; }     // strict >: first maximum wins
            }
        }
        typename Vec8<T>::type o;
        uint64_t packed = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) { o[j] = (T)best[j]; packed |= (uint64_t)bi[j] << (8 * j); }
        store8<T>(y + i * 8, o);
        if (idx) *reinterpret_cast<uint64_t*>(idx + i * 8) = packed;      // forward-only plans keep no indices
    }
}

// One POOLED row per blockIdx.y (+ 65535 * blockIdx.z), threads along (pooled x, channel group); a thread owns the whole 2x2
// window: dy and the positions are read ONCE (rounds 1-4 ran a thread per input pixel -- the four pixels of a window sat in two
// workgroup rows and each fetched the window's 16 + 8 bytes again: PMC x1.27 the algorithmic bytes, profiles/r04_traffic.json),
// the two pixels of a window row are 32 contiguous bytes per lane.  No 64-bit divisions.
template <typename T>
__global__ __launch_bounds__(256) void pool_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ idx, T* __restrict__ dx,
                                                       const T* __restrict__ mask, int orows, int H, int W, int Ho, int Wo, int Cs, int accumulate) {
    const int cgs = Cs >> 3;
    // flat index over (pooled row, pooled x, channel group): every thread of every workgroup but the last has a window (with one
    // pooled ROW per workgroup half of the 256 threads had none at the net's sizes: Wo * cgs = 128); 32-bit divisions only
    const unsigned wc = (unsigned)(Wo * cgs);
    const unsigned flat = blockIdx.x * 256u + threadIdx.x;
    const int orow = (int)(flat / wc);                               // b * Ho + oy
    const int t = (int)(flat - (unsigned)orow * wc);                 // ox * cgs + cg
    if (orow >= orows) return;
    const int b = orow / Ho, oy = orow - b * Ho;
    const int ox = t / cgs, cg = t - ox * cgs;
    const int64_t o = ((int64_t)orow * Wo * cgs + t) * 8;
    const typename Vec8<T>::type g = load8<T>(dy + o);
    const uint64_t packed = *reinterpret_cast<const uint64_t*>(idx + o);
    typedef typename Vec8<T>::type V8;
    const bool elu = accumulate & 2;                                 // bit 1: `mask` is the output of an ELU, not of a ReLU
    accumulate &= 1;
    V8 old[4], m[4];
    bool ok[4];
#pragma unroll
    for (int pos = 0; pos < 4; ++pos) {
        const int iy = 2 * oy + (pos >> 1), ix = 2 * ox + (pos & 1);
        ok[pos] = iy < H && ix < W;
        const int64_t i = (((int64_t)b * H + iy) * W + ix) * cgs + cg;
        old[pos] = accumulate && ok[pos] ? load8<T>(dx + i * 8) : zero8<T>();
        if (mask && ok[pos]) m[pos] = load8<T>(mask + i * 8);
    }
#pragma unroll
    for (int pos = 0; pos < 4; ++pos) {
        if (!ok[pos]) continue;
        const int iy = 2 * oy + (pos >> 1), ix = 2 * ox + (pos & 1);
        const int64_t i = (((int64_t)b * H + iy) * W + ix) * cgs + cg;
        V8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = (float)old[pos][j] + ((int)((packed >> (8 * j)) & 0xff) == pos ? (float)g[j] : 0.f);
            if (mask && !((float)m[pos][j] > 0.f)) v = elu ? v * ((float)m[pos][j] + 1.f) : 0.f;       // (ELU: y <= 0 -> d/dz = y + 1)
            r[j] = (T)v;
        }
        store8<T>(dx + i * 8, r);
    }
}

// =============================================================================================
// masked cross entropy (model/model.py:446-459), batch rule of SURVEY 8(e)
// =============================================================================================
// One 1024-thread workgroup per sample, 16-byte loads (two labels), eight in flight per thread, no atomics and no
// memset: counts[b] is written, not accumulated.  (The first version spread a sample over 64 workgroups and let every
// wave atomicAdd into counts[b]: 4096 same-address atomics took 51 us for 11 MB of labels.)
__global__ __launch_bounds__(1024) void label_count_kernel(const int64_t* __restrict__ labels, int32_t* __restrict__ counts, int64_t hw) {
    __shared__ int red[16];
    const int b = blockIdx.x;
    const int64_t* base = labels + (int64_t)b * hw;
    int local = 0;
    typedef long long ll2 __attribute__((ext_vector_type(2)));
    const int64_t npair = hw >> 1;                                   // base is 16-byte aligned when hw is even or b == 0
    const bool aligned = ((reinterpret_cast<uintptr_t>(base) & 15) == 0);
    if (aligned) {
        const ll2* p2 = reinterpret_cast<const ll2*>(base);
        int64_t i = threadIdx.x;
        for (; i + 7 * 1024 < npair; i += 8 * 1024) {
            ll2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p2[i + u * 1024];
#pragma unroll
            for (int u = 0; u < 8; ++u) local += (v[u][0] > 0) + (v[u][1] > 0);
        }
        for (; i < npair; i += 1024) { ll2 v = p2[i]; local += (v[0] > 0) + (v[1] > 0); }
        if ((hw & 1) && threadIdx.x == 0) local += base[hw - 1] > 0;
    } else {
        for (int64_t i = threadIdx.x; i < hw; i += 1024) local += base[i] > 0;
    }
    for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int i = 0; i < 16; ++i) t += red[i];
        counts[b] = t;
    }
}

// The same count with a sample spread over K workgroups (grid K x B): partial[b * K + k] is written, the consumer
// (msau_masked_ce_multi with counts_k = K) adds the K integers up -- exact in any order, no atomics, no follow-up launch.  One
// workgroup per sample left 240 of 256 CUs idle: 15 us for the bench's 11 MB of labels, on the main queue between the sweeps.
__global__ __launch_bounds__(256) void label_count_split_kernel(const int64_t* __restrict__ labels, int32_t* __restrict__ partial,
                                                                int64_t hw, int K) {
    __shared__ int red[4];
    const int b = blockIdx.y, k = blockIdx.x;
    const int64_t chunk = (hw + K - 1) / K;
    const int64_t lo = (int64_t)k * chunk, hi = lo + chunk < hw ? lo + chunk : hw;
    const int64_t* base = labels + (int64_t)b * hw;
    int local = 0;
    int64_t i = lo + threadIdx.x;
    for (; i + 7 * 256 < hi; i += 8 * 256) {
        int64_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = base[i + u * 256];
#pragma unroll
        for (int u = 0; u < 8; ++u) local += v[u] > 0;
    }
    for (; i < hi; i += 256) local += base[i] > 0;
    for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) partial[b * K + k] = red[0] + red[1] + red[2] + red[3];
}

// the block's sum of class weights (second reduction of the weighted CE), same order as the loss partials
__device__ __forceinline__ void ce_weight_sum(float local_w, float* red, float* out) {
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) local_w += __shfl_down(local_w, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local_w;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < kThreads / 64; ++i) s += red[i];
        out[blockIdx.x] = s;
    }
}

template <typename T, bool ALL>
__global__ void masked_ce_kernel(const T* __restrict__ logits, const int64_t* __restrict__ labels,
                                 const int32_t* __restrict__ counts, T* __restrict__ dlogits, float* __restrict__ partials,
                                 int B, int64_t hw, int C, int Cs, float scale, const float* __restrict__ cw = nullptr) {
    // cw (ALL only): per-class weights of torch.nn.CrossEntropyLoss(weight) (model/training/cost.py:24-31): every pixel's term and
    // gradient are multiplied by cw[label]; the sum of those weights -- the loss's denominator -- goes to the second half of partials
    __shared__ float red[kThreads / 64];
    float local = 0.f, local_w = 0.f;
    const int64_t total = (int64_t)B * hw;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(p / hw);
        const int64_t lab = labels[p];
        const bool on = ALL ? (lab >= 0 && lab < C) : (lab > 0 && lab < C);
        float w = on ? (ALL ? scale : scale / (float)max(counts[b], 1)) : 0.f;
        if (ALL && cw && on) { const float c = cw[(int)lab]; w *= c; local_w += c; }
        float x[16];
        float mx = -INFINITY;
        for (int c0 = 0; c0 < Cs && c0 < 16; c0 += 8) {
            typename Vec8<T>::type v = load8<T>(logits + p * Cs + c0);
#pragma unroll
            for (int j = 0; j < 8; ++j) { x[c0 + j] = (float)v[j]; if (c0 + j < C) mx = fmaxf(mx, x[c0 + j]); }
        }
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += __expf(x[c] - mx);
        const float lse = mx + __logf(se);
        if (on) local += w * (lse - x[(int)lab]);
        for (int c0 = 0; c0 < Cs; c0 += 8) {
            typename Vec8<T>::type o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int c = c0 + j;
                float g = 0.f;
                if (on && c < C) g = w * (__expf(x[c] - lse) - (c == (int)lab ? 1.f : 0.f));
                o[j] = (T)g;
            }
            store8<T>(dlogits + p * Cs + c0, o);
        }
    }
    for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < kThreads / 64; ++i) s += red[i];
        partials[blockIdx.x] = s;
    }
    if (ALL && cw) ce_weight_sum(local_w, red, partials + gridDim.x);
}

// Any class count (Cs > 16: e.g. the 17-class key-value head): the logits of a pixel are read three times from the
// cache instead of being held in registers.  Same arithmetic and summation order as masked_ce_kernel.
template <typename T, bool ALL>
__global__ void masked_ce_wide_kernel(const T* __restrict__ logits, const int64_t* __restrict__ labels,
                                      const int32_t* __restrict__ counts, T* __restrict__ dlogits, float* __restrict__ partials,
                                      int B, int64_t hw, int C, int Cs, float scale, const float* __restrict__ cw = nullptr) {
    // cw (ALL only): per-class weights of torch.nn.CrossEntropyLoss(weight) (model/training/cost.py:24-31): every pixel's term and
    // gradient are multiplied by cw[label]; the sum of those weights -- the loss's denominator -- goes to the second half of partials
    __shared__ float red[kThreads / 64];
    float local = 0.f, local_w = 0.f;
    const int64_t total = (int64_t)B * hw;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(p / hw);
        const int64_t lab = labels[p];
        const bool on = ALL ? (lab >= 0 && lab < C) : (lab != 0 && lab > 0 && lab < C);
        float w = on ? (ALL ? scale : scale / (float)max(counts[b], 1)) : 0.f;
        if (ALL && cw && on) { const float c = cw[(int)lab]; w *= c; local_w += c; }
        const T* l = logits + p * Cs;
        float mx = -INFINITY;
        for (int c = 0; c < C; ++c) mx = fmaxf(mx, (float)l[c]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += __expf((float)l[c] - mx);
        const float lse = mx + __logf(se);
        if (on) local += w * (lse - (float)l[lab]);
        for (int c = 0; c < Cs; ++c) {
            float g = 0.f;
            if (on && c < C) g = w * (__expf((float)l[c] - lse) - (c == (int)lab ? 1.f : 0.f));
            dlogits[p * Cs + c] = (T)g;
        }
    }
    for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < kThreads / 64; ++i) s += red[i];
        partials[blockIdx.x] = s;
    }
    if (ALL && cw) ce_weight_sum(local_w, red, partials + gridDim.x);
}

constexpr int kCeMaxB = 1024;
// Final + auxiliary masked CE in ONE launch (model/model.py:455-458: CE(out) + CE(aux) over the same labelled pixels):
// the label and the per-sample weight are read once, both gradients are written, block partial sums go to ws and the
// one-wave follow-up kernel adds them up in index order -> loss[0], reproducibly.  (A "last block sums" ticket
// instead of the second launch was tried: thousands of same-address atomics cost 200 us.)
template <typename T, int NL, int CS8>
__global__ void masked_ce_multi_kernel(const T* __restrict__ l0, const T* __restrict__ l1, const int64_t* __restrict__ labels,
                                       const int32_t* __restrict__ counts, T* __restrict__ d0, T* __restrict__ d1,
                                       float* __restrict__ ws, float* __restrict__ loss, int B, int hw, int C, float scale, int K) {
    constexpr int Cs = CS8 * 8;
    __shared__ float red[kThreads / 64];
    __shared__ int cnt[kCeMaxB];                                       // per-sample counts: the K partial counts added up (B <= kCeMaxB)
    const bool in_lds = B <= kCeMaxB;
    if (in_lds) {
        for (int b = threadIdx.x; b < B; b += blockDim.x) {
            int t = 0;
            for (int k = 0; k < K; ++k) t += counts[b * K + k];
            cnt[b] = t;
        }
        __syncthreads();
    }
    float local = 0.f;
    const int64_t total = (int64_t)B * hw;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)((unsigned)p / (unsigned)hw);               // total < 2^31 (checked by the host)
        const int lab = (int)labels[p];
        const bool on = lab != 0 && lab < C && lab > 0;
        const float w = on ? scale / (float)max(in_lds ? cnt[b] : counts[b], 1) : 0.f;
#pragma unroll
        for (int t = 0; t < NL; ++t) {
            const T* lg = t ? l1 : l0;
            T* dg = t ? d1 : d0;
            float x[Cs];                                               // every index below is a compile-time constant:
            float mx = -INFINITY, xl = 0.f;                            // the array stays in registers
#pragma unroll
            for (int c0 = 0; c0 < Cs; c0 += 8) {
                typename Vec8<T>::type v = load8<T>(lg + p * Cs + c0);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    x[c0 + j] = (float)v[j];
                    if (c0 + j < C) mx = fmaxf(mx, x[c0 + j]);
                    xl = (c0 + j == lab) ? x[c0 + j] : xl;
                }
            }
            float se = 0.f;
#pragma unroll
            for (int c = 0; c < Cs; ++c) se += c < C ? __expf(x[c] - mx) : 0.f;
            const float lse = mx + __logf(se);
            if (on) local += w * (lse - xl);
#pragma unroll
            for (int c0 = 0; c0 < Cs; c0 += 8) {
                typename Vec8<T>::type o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = c0 + j;
                    float g = 0.f;
                    if (on && c < C) g = w * (__expf(x[c] - lse) - (c == lab ? 1.f : 0.f));
                    o[j] = (T)g;
                }
                store8<T>(dg + p * Cs + c0, o);
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < kThreads / 64; ++i) s += red[i];
        ws[blockIdx.x] = s;
    }
}

// 256 threads, fixed association order (thread t adds elements t, t+256, ...; lanes, then waves, combine in index
// order) -> reproducible.  SET overwrites out[0], otherwise out[0] += sum.
template <bool SET>
__global__ __launch_bounds__(256) void ordered_sum_kernel_t(const float* __restrict__ partials, int n, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += partials[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = (red[0] + red[1]) + (red[2] + red[3]);
        if (SET) out[0] = t; else out[0] += t;
    }
}

// =============================================================================================
// per-channel sums (bias gradient of the transposed conv)
// =============================================================================================
template <typename T>
__global__ void channel_sum_kernel(const T* __restrict__ g, int64_t npix, int Cs, float* __restrict__ partials) {
    // thread t of the block owns channel group (t % cgs); rows of pixels strided by blockDim/cgs
    extern __shared__ float sm[];
    const int cgs = Cs >> 3;
    const int cg = threadIdx.x % cgs, lane_p = threadIdx.x / cgs, ppb = blockDim.x / cgs;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (lane_p < ppb) {
        // four 16-byte loads in flight per thread (rounds 1-4: one, i.e. ~20 dependent round trips per thread for the level-0 tensor:
        // 11 us for 22 MB); the order of the sum is fixed by the grid, as before
        const int64_t stride = (int64_t)gridDim.x * ppb;
        int64_t p = (int64_t)blockIdx.x * ppb + lane_p;
        float a1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, a2[8] = {0, 0, 0, 0, 0, 0, 0, 0}, a3[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (; p + 3 * stride < npix; p += 4 * stride) {
            const typename Vec8<T>::type v0 = load8<T>(g + p * Cs + cg * 8), v1 = load8<T>(g + (p + stride) * Cs + cg * 8),
                                         v2 = load8<T>(g + (p + 2 * stride) * Cs + cg * 8), v3 = load8<T>(g + (p + 3 * stride) * Cs + cg * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) { acc[j] += (float)v0[j]; a1[j] += (float)v1[j]; a2[j] += (float)v2[j]; a3[j] += (float)v3[j]; }
        }
        for (; p < npix; p += stride) {
            const typename Vec8<T>::type v = load8<T>(g + p * Cs + cg * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = (acc[j] + a1[j]) + (a2[j] + a3[j]);
    }
    float* mine = sm + threadIdx.x * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) mine[j] = acc[j];
    __syncthreads();
    for (int c = threadIdx.x; c < Cs; c += blockDim.x) {
        int cgc = c >> 3, j = c & 7;
        float s = 0.f;
        for (int r = 0; r < ppb; ++r) s += sm[(r * cgs + cgc) * 8 + j];
        partials[(int64_t)blockIdx.x * Cs + c] = s;
    }
}

// =============================================================================================
// global-norm clip + Adam (train_chargrid_funsd_msau.py:24-26,58-59)
// =============================================================================================
__global__ void sqsum_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ partials, float* __restrict__ state, float beta1, float beta2) {
    __shared__ float red[kThreads / 64];
    // four independent loads in flight per thread (a fixed order: the same bits on every run)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const float v0 = g[i], v1 = g[i + stride], v2 = g[i + 2 * stride], v3 = g[i + 3 * stride];
        s0 += v0 * v0; s1 += v1 * v1; s2 += v2 * v2; s3 += v3 * v3;
    }
    for (; i < n; i += stride) { const float v = g[i]; s0 += v * v; }
    float s = (s0 + s1) + (s2 + s3);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i2 = 0; i2 < kThreads / 64; ++i2) t += red[i2];
        partials[blockIdx.x] = t;
        if (blockIdx.x == 0 && state) {
            // the step counter and Adam's bias corrections, once: every workgroup of adam_kernel reads them (rounds 1-4: each of
            // its 512 workgroups evaluated two double-precision pow() before touching a parameter)
            const float step = state[0] + 1.f;
            state[0] = step;
            state[3] = (float)(1.0 - pow((double)beta1, (double)step));
            state[4] = (float)(1.0 - pow((double)beta2, (double)step));
        }
    }
}

// clip coefficient and bias corrections are derived by every workgroup from the sqsum partials (fixed order: the same bits
// everywhere) instead of by a one-wave kernel in between: one dependent launch less at the tail of the step
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            float* __restrict__ state, const float* __restrict__ partials, int npart, int64_t n, float lr,
                            float beta1, float beta2, float eps, float max_norm, float grad_scale) {
    __shared__ float sh[4];
    if (threadIdx.x < 64) {
        float s = 0.f;
        for (int i = threadIdx.x; i < npart; i += 64) s += partials[i];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if (threadIdx.x == 0) {
            const float norm = sqrtf(s) * grad_scale;
            const float coef = max_norm / (norm + 1e-6f);               // torch.nn.utils.clip_grad_norm_
            sh[0] = coef < 1.f ? coef : 1.f;
            sh[1] = state[3];                                            // bias corrections: sqsum_kernel's workgroup 0
            sh[2] = state[4];
            if (blockIdx.x == 0) { state[1] = norm; state[2] = sh[0]; }
        }
    }
    __syncthreads();
    const float gs = sh[0] * grad_scale;
    const float bc1 = sh[1], bc2s = sqrtf(sh[2]);
    const float step_size = lr / bc1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gi = g[i] * gs;
        float mi = beta1 * m[i] + (1.f - beta1) * gi;
        float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi; v[i] = vi;
        float denom = sqrtf(vi) / bc2s + eps;
        p[i] -= step_size * (mi / denom);
    }
}

__global__ void softmax_nchw_kernel(const float* __restrict__ logits, float* __restrict__ pred, int B, int C, int64_t hw) {
    const int64_t total = (int64_t)B * hw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t b = i / hw, p = i % hw;
        const float* src = logits + b * C * hw + p;
        float mx = -INFINITY;
        for (int c = 0; c < C; ++c) mx = fmaxf(mx, src[c * hw]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += __expf(src[c * hw] - mx);
        float inv = 1.f / se;
        for (int c = 0; c < C; ++c) pred[b * C * hw + c * hw + p] = __expf(src[c * hw] - mx) * inv;
    }
}

// softmax + first-maximum index over the C real channels of NHWC logits, any C <= 255.  Operation for operation
// msau_head_softmax() of msau_common.h (max, sum and product in channel order, __expf), which the conv epilogue
// (MSAU_CONV_HEAD) uses for C <= 16: the two produce identical bits.
template <typename T>
__global__ void softmax_argmax_nhwc_kernel(const T* __restrict__ logits, float* __restrict__ probs,
                                           uint8_t* __restrict__ amax, int64_t npix, int C, int Cs) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
        const T* l = logits + p * Cs;
        float mx = -INFINITY;
        for (int c = 0; c < C; ++c) mx = fmaxf(mx, (float)l[c]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += __expf((float)l[c] - mx);
        const float inv = 1.f / se;
        int best = 0;
        float bp = -1.f;
        for (int c = 0; c < C; ++c) {
            const float pr = __expf((float)l[c] - mx) * inv;
            probs[p * C + c] = pr;
            if (pr > bp) { bp = pr; best = c; }
        }
        amax[p] = (uint8_t)best;
    }
}

template <typename T>
__global__ void onehot_ids_kernel(const int32_t* __restrict__ ids, T* __restrict__ grid, int64_t npix, int C, int Cs) {
    const int cgs = Cs >> 3;
    const int64_t total = npix * cgs;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cg = (int)(i % cgs);
        const int id = ids[i / cgs];
        typename Vec8<T>::type o = zero8<T>();
        const int j = id - cg * 8;
        if (id >= 0 && id < C && j >= 0 && j < 8) o[j] = (T)1.0f;
        store8<T>(grid + i * 8, o);
    }
}

}  // namespace

#define DISPATCH_T(dtype, CALL_F32, CALL_BF16)                          \
    do {                                                                \
        if ((dtype) == MSAU_F32) { CALL_F32; }                          \
        else if ((dtype) == MSAU_BF16) { CALL_BF16; }                   \
        else return msau_set_error(MSAU_ERR_ARG, "bad dtype %d", (int)(dtype)); \
    } while (0)

extern "C" int msau_nchw_to_nhwc(void* stream, int dtype, const float* src, void* dst, int B, int C, int Cs, int H, int W) {
    MSAU_CHECK_ARG(src && dst && B > 0 && C > 0 && Cs >= C && Cs % 8 == 0 && H > 0 && W > 0, "nchw_to_nhwc: bad args");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int64_t HW = (int64_t)H * W;
    static const bool reg_off = std::getenv("MSAU_NCHW_REG") && std::getenv("MSAU_NCHW_REG")[0] == '0';
    if (!reg_off && HW % 4 == 0 && Cs >= 16 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const int64_t total = (int64_t)B * (HW / 4) * (Cs / 8);
        MSAU_CHECK_ARG(cdiv64(total, 256) < (1ll << 31), "nchw_to_nhwc: grid too large");
        const int grid = (int)cdiv64(total, 256);
        static const bool wide_off = getenv("MSAU_NCHW_WIDE") && atoi(getenv("MSAU_NCHW_WIDE")) == 0;
        if (!wide_off && Cs > 64 && (Cs / 8) % 8 == 0 && (HW / 4) % 8 == 0) {
            DISPATCH_T(dtype,
                       hipLaunchKernelGGL((nchw_to_nhwc_reg_kernel<float, true>), dim3(grid), dim3(256), 0, s, src, static_cast<float*>(dst), C, Cs, HW, total),
                       hipLaunchKernelGGL((nchw_to_nhwc_reg_kernel<bf16_t, true>), dim3(grid), dim3(256), 0, s, src, static_cast<bf16_t*>(dst), C, Cs, HW, total));
        } else {
            DISPATCH_T(dtype,
                       hipLaunchKernelGGL((nchw_to_nhwc_reg_kernel<float, false>), dim3(grid), dim3(256), 0, s, src, static_cast<float*>(dst), C, Cs, HW, total),
                       hipLaunchKernelGGL((nchw_to_nhwc_reg_kernel<bf16_t, false>), dim3(grid), dim3(256), 0, s, src, static_cast<bf16_t*>(dst), C, Cs, HW, total));
        }
        MSAU_CHECK_LAUNCH("nchw_to_nhwc_reg");
        return 0;
    }
    if (Cs >= 32 && Cs <= 1024) {
        const int tiles = (int)cdiv64(HW, 64);
        MSAU_CHECK_ARG((int64_t)B * tiles < (1ll << 31), "nchw_to_nhwc: grid too large");
        const size_t esz = dtype == MSAU_F32 ? 4 : 2;
        const size_t lds = 64 * (size_t)(Cs + 8) * esz;
        if (lds <= 64 * 1024) {
            DISPATCH_T(dtype,
                       hipLaunchKernelGGL(nchw_to_nhwc_tiled_kernel<float>, dim3(B * tiles), dim3(256), lds, s, src, static_cast<float*>(dst), C, Cs, HW, tiles),
                       hipLaunchKernelGGL(nchw_to_nhwc_tiled_kernel<bf16_t>, dim3(B * tiles), dim3(256), lds, s, src, static_cast<bf16_t*>(dst), C, Cs, HW, tiles));
            MSAU_CHECK_LAUNCH("nchw_to_nhwc_tiled");
            return 0;
        }
    }
    int grid = grid_for((int64_t)B * (Cs / 8) * HW);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(grid), dim3(kThreads), 0, s, src, static_cast<float*>(dst), B, C, Cs, HW),
               hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(grid), dim3(kThreads), 0, s, src, static_cast<bf16_t*>(dst), B, C, Cs, HW));
    MSAU_CHECK_LAUNCH("nchw_to_nhwc");
    return 0;
}

extern "C" int msau_nhwc_to_nchw(void* stream, int dtype, const void* src, float* dst, int B, int C, int Cs, int H, int W) {
    MSAU_CHECK_ARG(src && dst && B > 0 && C > 0 && Cs >= C && Cs % 8 == 0 && H > 0 && W > 0, "nhwc_to_nchw: bad args");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int64_t HW = (int64_t)H * W;
    int grid = grid_for((int64_t)B * (Cs / 8) * HW);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(grid), dim3(kThreads), 0, s, static_cast<const float*>(src), dst, B, C, Cs, HW),
               hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3(grid), dim3(kThreads), 0, s, static_cast<const bf16_t*>(src), dst, B, C, Cs, HW));
    MSAU_CHECK_LAUNCH("nhwc_to_nchw");
    return 0;
}

extern "C" int msau_nchw_grad_to_nhwc(void* stream, int dtype, const float* src, void* dst, int B, int C, int Cs, int H, int W,
                                      int accumulate) {
    MSAU_CHECK_ARG(src && dst && B > 0 && C > 0 && Cs >= C && Cs % 8 == 0 && H > 0 && W > 0, "nchw_grad_to_nhwc: bad args");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int64_t HW = (int64_t)H * W;
    int grid = grid_for((int64_t)B * (Cs / 8) * HW);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(nchw_grad_to_nhwc_kernel<float>, dim3(grid), dim3(kThreads), 0, s, src, static_cast<float*>(dst), B, C, Cs, HW, accumulate),
               hipLaunchKernelGGL(nchw_grad_to_nhwc_kernel<bf16_t>, dim3(grid), dim3(kThreads), 0, s, src, static_cast<bf16_t*>(dst), B, C, Cs, HW, accumulate));
    MSAU_CHECK_LAUNCH("nchw_grad_to_nhwc");
    return 0;
}

extern "C" int msau_lrn_fwd(void* stream, int dtype, const void* a, void* y, int64_t npix, int C, int Cs, int n,
                            float alpha, float beta, float k) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == MSAU_F32) return lrn_dispatch<float, false>(s, a, nullptr, y, npix, C, Cs, n, alpha, beta, k);
    if (dtype == MSAU_BF16) return lrn_dispatch<bf16_t, false>(s, a, nullptr, y, npix, C, Cs, n, alpha, beta, k);
    return msau_set_error(MSAU_ERR_ARG, "lrn_fwd: bad dtype");
}

extern "C" int msau_lrn_bwd(void* stream, int dtype, const void* a, const void* dy, void* da, int64_t npix, int C, int Cs,
                            int n, float alpha, float beta, float k) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == MSAU_F32) return lrn_dispatch<float, true>(s, a, dy, da, npix, C, Cs, n, alpha, beta, k);
    if (dtype == MSAU_BF16) return lrn_dispatch<bf16_t, true>(s, a, dy, da, npix, C, Cs, n, alpha, beta, k);
    return msau_set_error(MSAU_ERR_ARG, "lrn_bwd: bad dtype");
}

extern "C" int msau_maxpool2x2_fwd(void* stream, int dtype, const void* x, void* y, uint8_t* idx, int B, int H, int W, int Cs) {
    MSAU_CHECK_ARG(x && y && B > 0 && H > 0 && W > 0 && Cs % 8 == 0, "maxpool_fwd: bad args");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    int grid = grid_for((int64_t)B * Ho * Wo * (Cs / 8));
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(pool_fwd_kernel<float>, dim3(grid), dim3(kThreads), 0, s, static_cast<const float*>(x), static_cast<float*>(y), idx, B, H, W, Ho, Wo, Cs),
               hipLaunchKernelGGL(pool_fwd_kernel<bf16_t>, dim3(grid), dim3(kThreads), 0, s, static_cast<const bf16_t*>(x), static_cast<bf16_t*>(y), idx, B, H, W, Ho, Wo, Cs));
    MSAU_CHECK_LAUNCH("pool_fwd");
    return 0;
}

extern "C" int msau_maxpool2x2_bwd(void* stream, int dtype, const void* dy, const uint8_t* idx, void* dx, const void* mask,
                                   int B, int H, int W, int Cs, int accumulate) {
    MSAU_CHECK_ARG(dy && dx && idx && B > 0 && H > 0 && W > 0 && Cs % 8 == 0, "maxpool_bwd: bad args");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    MSAU_CHECK_ARG((int64_t)B * Ho * Wo * (Cs / 8) < (1ll << 31) - 256, "maxpool_bwd: image too large");
    const int rows = B * Ho;                                         // pooled rows: a thread owns a 2x2 window
    const dim3 grid((unsigned)(((int64_t)rows * Wo * (Cs / 8) + 255) / 256));
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(pool_bwd_kernel<float>, grid, dim3(256), 0, s, static_cast<const float*>(dy), idx, static_cast<float*>(dx), static_cast<const float*>(mask), rows, H, W, Ho, Wo, Cs, accumulate),
               hipLaunchKernelGGL(pool_bwd_kernel<bf16_t>, grid, dim3(256), 0, s, static_cast<const bf16_t*>(dy), idx, static_cast<bf16_t*>(dx), static_cast<const bf16_t*>(mask), rows, H, W, Ho, Wo, Cs, accumulate));
    MSAU_CHECK_LAUNCH("pool_bwd");
    return 0;
}

extern "C" int msau_label_counts(void* stream, const int64_t* labels, int32_t* counts, int B, int64_t hw) {
    MSAU_CHECK_ARG(labels && counts && B > 0 && hw > 0, "label_counts: bad args");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(label_count_kernel, dim3(B), dim3(1024), 0, s, labels, counts, hw);
    MSAU_CHECK_LAUNCH("label_count");
    return 0;
}

extern "C" int msau_label_counts_split(void* stream, const int64_t* labels, int32_t* partial, int B, int64_t hw, int K) {
    MSAU_CHECK_ARG(labels && partial && B > 0 && B <= 65535 && hw > 0 && K >= 1 && K <= 64, "label_counts_split: bad args");
    hipLaunchKernelGGL(label_count_split_kernel, dim3(K, B), dim3(256), 0, static_cast<hipStream_t>(stream), labels, partial, hw, K);
    MSAU_CHECK_LAUNCH("label_count_split");
    return 0;
}

static int ce_blocks(int64_t npix) { return grid_for(npix, 1024); }
extern "C" int64_t msau_ce_ws_floats(int64_t npix_total) { return 2 * ce_blocks(npix_total); }      // (second half: msau_softmax_ce_weighted's weight sums)

extern "C" int msau_masked_ce(void* stream, int dtype, const void* logits, const int64_t* labels, const int32_t* counts,
                              void* dlogits, float* loss_accum, float* ws, int B, int64_t hw, int C, int Cs, float scale) {
    MSAU_CHECK_ARG(logits && labels && counts && dlogits && loss_accum && ws, "masked_ce: null pointer");
    MSAU_CHECK_ARG(B > 0 && hw > 0 && C > 0 && C <= Cs && Cs % 8 == 0 && Cs <= 256, "masked_ce: bad dims (n_class <= 256)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int nb = ce_blocks((int64_t)B * hw);
    if (Cs > 16) {
        DISPATCH_T(dtype,
                   hipLaunchKernelGGL((masked_ce_wide_kernel<float, false>), dim3(nb), dim3(kThreads), 0, s, static_cast<const float*>(logits), labels, counts, static_cast<float*>(dlogits), ws, B, hw, C, Cs, scale),
                   hipLaunchKernelGGL((masked_ce_wide_kernel<bf16_t, false>), dim3(nb), dim3(kThreads), 0, s, static_cast<const bf16_t*>(logits), labels, counts, static_cast<bf16_t*>(dlogits), ws, B, hw, C, Cs, scale));
        MSAU_CHECK_LAUNCH("masked_ce_wide");
        hipLaunchKernelGGL(ordered_sum_kernel_t<false>, dim3(1), dim3(256), 0, s, ws, nb, loss_accum);
        MSAU_CHECK_LAUNCH("ordered_sum");
        return 0;
    }
    DISPATCH_T(dtype,
               hipLaunchKernelGGL((masked_ce_kernel<float, false>), dim3(nb), dim3(kThreads), 0, s, static_cast<const float*>(logits), labels, counts, static_cast<float*>(dlogits), ws, B, hw, C, Cs, scale),
               hipLaunchKernelGGL((masked_ce_kernel<bf16_t, false>), dim3(nb), dim3(kThreads), 0, s, static_cast<const bf16_t*>(logits), labels, counts, static_cast<bf16_t*>(dlogits), ws, B, hw, C, Cs, scale));
    MSAU_CHECK_LAUNCH("masked_ce");
    hipLaunchKernelGGL(ordered_sum_kernel_t<false>, dim3(1), dim3(256), 0, s, ws, nb, loss_accum);
    MSAU_CHECK_LAUNCH("ordered_sum");
    return 0;
}

static int ce_multi_blocks(int64_t npix) { return grid_for(npix, 2048); }
extern "C" int64_t msau_ce_multi_ws_floats(int64_t npix_total) { return ce_multi_blocks(npix_total) + 1; }

extern "C" int msau_masked_ce_multi(void* stream, int dtype, const void* logits, const void* aux, const int64_t* labels,
                                    const int32_t* counts, void* dlogits, void* daux, float* loss, float* ws,
                                    int B, int64_t hw, int C, int Cs, float scale, int counts_k) {
    MSAU_CHECK_ARG(logits && labels && counts && dlogits && loss && ws && (!aux || daux), "masked_ce_multi: null pointer");
    MSAU_CHECK_ARG(counts_k == 1 || (counts_k > 1 && counts_k <= 64 && B <= kCeMaxB), "masked_ce_multi: counts_k > 1 needs B <= 1024");
    MSAU_CHECK_ARG(B > 0 && hw > 0 && C > 0 && C <= Cs && Cs % 8 == 0 && Cs <= 16 && (int64_t)B * hw < (1ll << 31),
                   "masked_ce_multi: bad dims (n_class <= 16, B*H*W < 2^31)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nb = ce_multi_blocks((int64_t)B * hw);
#define CE_MULTI(T, NL, C8) hipLaunchKernelGGL((masked_ce_multi_kernel<T, NL, C8>), dim3(nb), dim3(kThreads), 0, s, static_cast<const T*>(logits), \
        static_cast<const T*>(aux), labels, counts, static_cast<T*>(dlogits), static_cast<T*>(daux), ws, loss, B, (int)hw, C, scale, counts_k)
    if (aux && Cs == 8) { DISPATCH_T(dtype, CE_MULTI(float, 2, 1), CE_MULTI(bf16_t, 2, 1)); }
    else if (aux) { DISPATCH_T(dtype, CE_MULTI(float, 2, 2), CE_MULTI(bf16_t, 2, 2)); }
    else if (Cs == 8) { DISPATCH_T(dtype, CE_MULTI(float, 1, 1), CE_MULTI(bf16_t, 1, 1)); }
    else { DISPATCH_T(dtype, CE_MULTI(float, 1, 2), CE_MULTI(bf16_t, 1, 2)); }
#undef CE_MULTI
    MSAU_CHECK_LAUNCH("masked_ce_multi");
    hipLaunchKernelGGL(ordered_sum_kernel_t<true>, dim3(1), dim3(256), 0, s, ws, nb, loss);
    MSAU_CHECK_LAUNCH("ordered_sum_set");
    return 0;
}

extern "C" int msau_softmax_ce(void* stream, int dtype, const void* logits, const int64_t* labels, void* dlogits,
                               float* loss_accum, float* ws, int B, int64_t hw, int C, int Cs, float scale) {
    MSAU_CHECK_ARG(logits && labels && dlogits && loss_accum && ws, "softmax_ce: null pointer");
    MSAU_CHECK_ARG(B > 0 && hw > 0 && C > 0 && C <= Cs && Cs % 8 == 0 && Cs <= 256, "softmax_ce: bad dims (n_class <= 256)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int nb = ce_blocks((int64_t)B * hw);
    if (Cs > 16) {
        DISPATCH_T(dtype,
                   hipLaunchKernelGGL((masked_ce_wide_kernel<float, true>), dim3(nb), dim3(kThreads), 0, s, static_cast<const float*>(logits), labels, nullptr, static_cast<float*>(dlogits), ws, B, hw, C, Cs, scale),
                   hipLaunchKernelGGL((masked_ce_wide_kernel<bf16_t, true>), dim3(nb), dim3(kThreads), 0, s, static_cast<const bf16_t*>(logits), labels, nullptr, static_cast<bf16_t*>(dlogits), ws, B, hw, C, Cs, scale));
        MSAU_CHECK_LAUNCH("softmax_ce_wide");
        hipLaunchKernelGGL(ordered_sum_kernel_t<false>, dim3(1), dim3(256), 0, s, ws, nb, loss_accum);
        MSAU_CHECK_LAUNCH("ordered_sum");
        return 0;
    }
    DISPATCH_T(dtype,
               hipLaunchKernelGGL((masked_ce_kernel<float, true>), dim3(nb), dim3(kThreads), 0, s, static_cast<const float*>(logits), labels, nullptr, static_cast<float*>(dlogits), ws, B, hw, C, Cs, scale),
               hipLaunchKernelGGL((masked_ce_kernel<bf16_t, true>), dim3(nb), dim3(kThreads), 0, s, static_cast<const bf16_t*>(logits), labels, nullptr, static_cast<bf16_t*>(dlogits), ws, B, hw, C, Cs, scale));
    MSAU_CHECK_LAUNCH("softmax_ce");
    hipLaunchKernelGGL(ordered_sum_kernel_t<false>, dim3(1), dim3(256), 0, s, ws, nb, loss_accum);
    MSAU_CHECK_LAUNCH("ordered_sum");
    return 0;
}

extern "C" int msau_softmax_ce_weighted(void* stream, int dtype, const void* logits, const int64_t* labels, const float* class_w,
                                        void* dlogits, float* sums, float* ws, int B, int64_t hw, int C, int Cs) {
    MSAU_CHECK_ARG(logits && labels && class_w && dlogits && sums && ws, "softmax_ce_weighted: null pointer");
    MSAU_CHECK_ARG(B > 0 && hw > 0 && C > 0 && C <= Cs && Cs % 8 == 0 && Cs <= 256, "softmax_ce_weighted: bad dims (n_class <= 256)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nb = ce_blocks((int64_t)B * hw);
    if (Cs > 16) {
        DISPATCH_T(dtype,
                   hipLaunchKernelGGL((masked_ce_wide_kernel<float, true>), dim3(nb), dim3(kThreads), 0, s, static_cast<const float*>(logits), labels, nullptr, static_cast<float*>(dlogits), ws, B, hw, C, Cs, 1.f, class_w),
                   hipLaunchKernelGGL((masked_ce_wide_kernel<bf16_t, true>), dim3(nb), dim3(kThreads), 0, s, static_cast<const bf16_t*>(logits), labels, nullptr, static_cast<bf16_t*>(dlogits), ws, B, hw, C, Cs, 1.f, class_w));
    } else {
        DISPATCH_T(dtype,
                   hipLaunchKernelGGL((masked_ce_kernel<float, true>), dim3(nb), dim3(kThreads), 0, s, static_cast<const float*>(logits), labels, nullptr, static_cast<float*>(dlogits), ws, B, hw, C, Cs, 1.f, class_w),
                   hipLaunchKernelGGL((masked_ce_kernel<bf16_t, true>), dim3(nb), dim3(kThreads), 0, s, static_cast<const bf16_t*>(logits), labels, nullptr, static_cast<bf16_t*>(dlogits), ws, B, hw, C, Cs, 1.f, class_w));
    }
    MSAU_CHECK_LAUNCH("softmax_ce_weighted");
    hipLaunchKernelGGL(ordered_sum_kernel_t<false>, dim3(1), dim3(256), 0, s, ws, nb, sums);
    hipLaunchKernelGGL(ordered_sum_kernel_t<false>, dim3(1), dim3(256), 0, s, ws + nb, nb, sums + 1);
    MSAU_CHECK_LAUNCH("ordered_sum");
    return 0;
}

extern "C" int msau_channel_sum(void* stream, int dtype, const void* g, int64_t npix, int Cs, float* partials, int nblk) {
    MSAU_CHECK_ARG(g && partials && npix > 0 && Cs % 8 == 0 && Cs <= 256 && nblk > 0, "channel_sum: bad args");
    MSAU_CHECK_ARG(kThreads % (Cs / 8) == 0, "channel_sum: Cs/8 must divide %d", kThreads);
    hipStream_t s = static_cast<hipStream_t>(stream);
    size_t lds = kThreads * 8 * sizeof(float);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(channel_sum_kernel<float>, dim3(nblk), dim3(kThreads), lds, s, static_cast<const float*>(g), npix, Cs, partials),
               hipLaunchKernelGGL(channel_sum_kernel<bf16_t>, dim3(nblk), dim3(kThreads), lds, s, static_cast<const bf16_t*>(g), npix, Cs, partials));
    MSAU_CHECK_LAUNCH("channel_sum");
    return 0;
}

static int adam_blocks(int64_t n) { return grid_for(n, 512); }
extern "C" int64_t msau_adam_ws_floats(int64_t n) { return adam_blocks(n); }

extern "C" int msau_clip_adam_step(void* stream, float* params, const float* grads, float* m, float* v, float* state,
                                   float* ws, int64_t n, float lr, float beta1, float beta2, float eps, float max_norm,
                                   float grad_scale) {
    MSAU_CHECK_ARG(params && grads && m && v && state && ws && n > 0, "clip_adam: bad args");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int nb = adam_blocks(n);
    // 128 partial sums of squares (two per lane of the wave that adds them up again in EVERY workgroup of adam_kernel; with one
    // partial per adam workgroup, 512, that prologue was eight dependent loads in front of 5 elements per thread)
    const int nsq = nb < 128 ? nb : 128;
    hipLaunchKernelGGL(sqsum_kernel, dim3(nsq), dim3(kThreads), 0, s, grads, n, ws, state, beta1, beta2);
    MSAU_CHECK_LAUNCH("sqsum");
    hipLaunchKernelGGL(adam_kernel, dim3(nb), dim3(kThreads), 0, s, params, grads, m, v, state, ws, nsq, n, lr, beta1, beta2, eps, max_norm, grad_scale);
    MSAU_CHECK_LAUNCH("adam");
    return 0;
}

// Busy-wait kernel (one wave): keeps the stream occupied for ~`microseconds` so that launches issued behind
// it queue up on the device; bench.py uses it to time kernels with HIP events without host launch gaps.
__global__ void spin_kernel(long long cycles) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(32);
}

// ---- stress check of the fork events (main -> side) created WITHOUT the system-scope fence (sequence.hip: msau_run_ops_dp) ----
// A producer kernel on `stream` rewrites a buffer with a new pattern, a fence-less event forks, a consumer kernel on `side_stream`
// counts the words that do not hold the new pattern; a default (fenced) event joins before the next rewrite, exactly as the executor's
// sweeps do.  The consumer of iteration i - 1 has left the OLD pattern in the L2s of the XCDs its workgroups ran on: a consumer that is
// not made to see the producer's writes reads stale lines, and this count is not zero.
__global__ void vis_fill_kernel(unsigned* __restrict__ a, long long words, unsigned pattern) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (long long)gridDim.x * blockDim.x)
        a[i] = pattern ^ (unsigned)i;
}
__global__ void vis_check_kernel(const unsigned* __restrict__ a, long long words, unsigned pattern, unsigned long long* __restrict__ bad) {
    unsigned long long local = 0;
    // (another thread -> word mapping than the producer's: the consumer's workgroups do not sit where the producer's did)
    for (long long i = words - 1 - ((long long)blockIdx.x * blockDim.x + threadIdx.x); i >= 0; i -= (long long)gridDim.x * blockDim.x)
        local += a[i] != (pattern ^ (unsigned)i);
    if (local) atomicAdd(bad, local);
}

extern "C" int msau_fork_visibility_check(void* stream, void* side_stream, int iters, int64_t words, int system_fence, int64_t* mismatches) {
    MSAU_CHECK_ARG(side_stream && stream != side_stream && iters > 0 && iters <= 100000 && words > 0 && words <= (1ll << 28) && mismatches,
                   "fork_visibility_check: bad args");
    hipStream_t ms = static_cast<hipStream_t>(stream), ss = static_cast<hipStream_t>(side_stream);
    unsigned* buf = nullptr;
    unsigned long long* bad = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    bool ok = hipMalloc(&buf, (size_t)words * 4) == hipSuccess && hipMalloc(&bad, 8) == hipSuccess &&
              hipMemsetAsync(bad, 0, 8, ms) == hipSuccess &&
              hipEventCreateWithFlags(&fork, hipEventDisableTiming | (system_fence ? 0u : hipEventDisableSystemFence)) == hipSuccess &&
              hipEventCreateWithFlags(&join, hipEventDisableTiming) == hipSuccess;
    const int grid = (int)((words + 4 * 256 - 1) / (4 * 256) < 2048 ? (words + 4 * 256 - 1) / (4 * 256) : 2048);
    for (int it = 0; ok && it < iters; ++it) {
        const unsigned pattern = 0x9E3779B9u * (unsigned)(it + 1);
        hipLaunchKernelGGL(vis_fill_kernel, dim3(grid), dim3(256), 0, ms, buf, (long long)words, pattern);
        ok = ok && hipEventRecord(fork, ms) == hipSuccess && hipStreamWaitEvent(ss, fork, 0) == hipSuccess;
        hipLaunchKernelGGL(vis_check_kernel, dim3(grid), dim3(256), 0, ss, buf, (long long)words, pattern, bad);
        ok = ok && hipEventRecord(join, ss) == hipSuccess && hipStreamWaitEvent(ms, join, 0) == hipSuccess;
    }
    unsigned long long host = ~0ull;
    ok = ok && hipStreamSynchronize(ms) == hipSuccess && hipMemcpy(&host, bad, 8, hipMemcpyDeviceToHost) == hipSuccess;
    if (fork) (void)hipEventDestroy(fork);
    if (join) (void)hipEventDestroy(join);
    if (buf) (void)hipFree(buf);
    if (bad) (void)hipFree(bad);
    if (!ok) return msau_set_error(MSAU_ERR_HIP, "fork_visibility_check: a HIP call failed (%s)", hipGetErrorString(hipGetLastError()));
    *mismatches = (int64_t)host;
    return 0;
}

extern "C" int msau_spin(void* stream, int microseconds) {
    MSAU_CHECK_ARG(microseconds >= 0 && microseconds <= 200000, "spin: 0..200000 us");
    // wall_clock64 ticks at 100 MHz on gfx950 (s_memrealtime)
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), (long long)microseconds * 100);
    MSAU_CHECK_LAUNCH("spin_kernel");
    return 0;
}

// A stream with an explicit queue priority: -1 = the device's highest, 0 = default, +1 = the device's lowest.  The
// weight-gradient side stream is created with +1 so that, when both queues have workgroups ready, the dispatcher serves
// the data-gradient chain (the critical path) first and the side launches fill what is left.
extern "C" int msau_stream_create(int priority, void** stream_out) {
    MSAU_CHECK_ARG(stream_out && priority >= -1 && priority <= 1, "stream_create: priority is -1, 0 or +1");
    int least = 0, greatest = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (e != hipSuccess) return msau_set_error(MSAU_ERR_HIP, "stream_create: %s", hipGetErrorString(e));
    hipStream_t s = nullptr;
    e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority < 0 ? greatest : priority > 0 ? least : 0);
    if (e != hipSuccess) return msau_set_error(MSAU_ERR_HIP, "stream_create: %s", hipGetErrorString(e));
    *stream_out = s;
    return 0;
}

extern "C" int msau_stream_destroy(void* stream) {
    MSAU_CHECK_ARG(stream, "stream_destroy: null");
    hipError_t e = hipStreamDestroy(static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return msau_set_error(MSAU_ERR_HIP, "stream_destroy: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int msau_fill_zero(void* stream, void* p, int64_t bytes) {
    MSAU_CHECK_ARG(p && bytes >= 0, "fill_zero: bad args");
    hipError_t e = hipMemsetAsync(p, 0, (size_t)bytes, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return msau_set_error(MSAU_ERR_HIP, "fill_zero: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int msau_softmax_channels_nchw(void* stream, const float* logits, float* pred, int B, int C, int64_t hw) {
    MSAU_CHECK_ARG(logits && pred && B > 0 && C > 0 && hw > 0, "softmax: bad args");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(softmax_nchw_kernel, dim3(grid_for((int64_t)B * hw)), dim3(kThreads), 0, s, logits, pred, B, C, hw);
    MSAU_CHECK_LAUNCH("softmax_nchw");
    return 0;
}

extern "C" int msau_softmax_argmax_nhwc(void* stream, int dtype, const void* logits, float* probs, uint8_t* argmax,
                                        int64_t npix, int C, int Cs) {
    MSAU_CHECK_ARG(logits && probs && argmax && npix > 0 && C > 0 && C <= 255 && C <= Cs, "softmax_argmax: bad args");
    hipStream_t s = static_cast<hipStream_t>(stream);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(softmax_argmax_nhwc_kernel<float>, dim3(grid_for(npix)), dim3(kThreads), 0, s, static_cast<const float*>(logits), probs, argmax, npix, C, Cs),
               hipLaunchKernelGGL(softmax_argmax_nhwc_kernel<bf16_t>, dim3(grid_for(npix)), dim3(kThreads), 0, s, static_cast<const bf16_t*>(logits), probs, argmax, npix, C, Cs));
    MSAU_CHECK_LAUNCH("softmax_argmax_nhwc");
    return 0;
}

extern "C" int msau_onehot_ids(void* stream, int dtype, const int32_t* ids, void* grid, int64_t npix, int C, int Cs) {
    MSAU_CHECK_ARG(ids && grid && npix > 0 && C > 0 && Cs % 8 == 0 && C <= Cs, "onehot_ids: bad args");
    hipStream_t s = static_cast<hipStream_t>(stream);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(onehot_ids_kernel<float>, dim3(grid_for(npix * (Cs / 8))), dim3(kThreads), 0, s, ids, static_cast<float*>(grid), npix, C, Cs),
               hipLaunchKernelGGL(onehot_ids_kernel<bf16_t>, dim3(grid_for(npix * (Cs / 8))), dim3(kThreads), 0, s, ids, static_cast<bf16_t*>(grid), npix, C, Cs));
    MSAU_CHECK_LAUNCH("onehot_ids");
    return 0;
}
