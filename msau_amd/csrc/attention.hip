// Bottleneck self-attention core (model/layers/attention.py:156-162), without the N x N matrix:
//     s[i][j] = g_i . f_j ;  beta = softmax_j(s) ;  o[:, j] = sum_i h[:, i] beta[i][j] ;  y = x + o
// The normalisation runs along j (rows of s) but the sum runs along i, so the row statistics
// (m_i, Z_i) come first (stats kernel) and every other pass recomputes exp(s - m_i)/Z_i on the fly.
// Round-1 implementation: fp32 VALU, one "own" row per lane, the "other" side staged through LDS and
// read as broadcasts, rows of the other side split over the 4 waves of a workgroup.
#include "msau_common.h"

namespace {

constexpr int TI = 128;          // rows of the other side staged per step (32 per wave)

enum { MODE_FWD_OUT = 0, MODE_BWD_DH = 1, MODE_BWD_DG = 2, MODE_BWD_DF = 3 };

template <typename T, int DS>
__global__ __launch_bounds__(256) void attn_stats_kernel(const T* __restrict__ f, const T* __restrict__ g,
                                                         float* __restrict__ stats, int N) {
    __shared__ float fs[256 * DS];
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    float gi[DS];
#pragma unroll
    for (int d = 0; d < DS; ++d) gi[d] = 0.f;
    if (i < N) {
#pragma unroll
        for (int d0 = 0; d0 < DS; d0 += 8) {
            typename Vec8<T>::type v = load8<T>(g + ((size_t)b * N + i) * DS + d0);
#pragma unroll
            for (int d = 0; d < 8; ++d) gi[d0 + d] = (float)v[d];
        }
    }
    float m = -INFINITY, Z = 0.f;
    for (int j0 = 0; j0 < N; j0 += 256) {
        __syncthreads();
        {
            int j = j0 + threadIdx.x;
#pragma unroll
            for (int d0 = 0; d0 < DS; d0 += 8) {
                typename Vec8<T>::type v = zero8<T>();
                if (j < N) v = load8<T>(f + ((size_t)b * N + j) * DS + d0);
#pragma unroll
                for (int d = 0; d < 8; ++d) fs[threadIdx.x * DS + d0 + d] = (float)v[d];
            }
        }
        __syncthreads();
        const int nj = min(256, N - j0);
        for (int jj = 0; jj < nj; ++jj) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < DS; ++d) s += gi[d] * fs[jj * DS + d];
            float mn = fmaxf(m, s);
            Z = Z * __expf(m - mn) + __expf(s - mn);
            m = mn;
        }
    }
    if (i < N) { stats[((size_t)b * N + i) * 2] = m; stats[((size_t)b * N + i) * 2 + 1] = Z; }
}

// LDS row of the "other" side: [DS] vector | m, 1/Z, delta, pad | [CS] vector   (all fp32)
template <typename T, int DS, int CS, int MODE>
__global__ __launch_bounds__(256) void attn_pass_kernel(const T* __restrict__ f, const T* __restrict__ g, const T* __restrict__ h,
                                                        const T* __restrict__ xdy,       // x (FWD_OUT) or dy (BWD_*)
                                                        const float* __restrict__ stats, float* __restrict__ delta,
                                                        T* __restrict__ out, int N) {
    constexpr bool OWN_IS_I = (MODE == MODE_BWD_DH || MODE == MODE_BWD_DG);   // own row index is i (a row of s)
    constexpr bool ACC_CS = (MODE == MODE_FWD_OUT || MODE == MODE_BWD_DH);    // accumulate a CS vector (else a DS vector)
    constexpr int ROW = DS + 4 + CS;
    constexpr int NACC = ACC_CS ? CS : DS;
    extern __shared__ __align__(16) float sm[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = blockIdx.y, own = blockIdx.x * 64 + lane;
    const bool valid = own < N;
    const size_t ob = (size_t)b * N + (valid ? own : 0);

    // ---- own-side registers
    float ov[DS];                         // f_j (own = j) or g_i (own = i)
    {
        const T* src = (OWN_IS_I ? g : f) + ob * DS;
#pragma unroll
        for (int d0 = 0; d0 < DS; d0 += 8) {
            typename Vec8<T>::type v = load8<T>(src + d0);
#pragma unroll
            for (int d = 0; d < 8; ++d) ov[d0 + d] = valid ? (float)v[d] : 0.f;
        }
    }
    float om = 0.f, oinvz = 0.f, odelta = 0.f;
    if (OWN_IS_I) { om = stats[ob * 2]; oinvz = 1.f / stats[ob * 2 + 1]; if (MODE == MODE_BWD_DG) odelta = delta[ob]; }
    float oc[(MODE == MODE_BWD_DG || MODE == MODE_BWD_DF) ? CS : 1];   // h_i (DG) or dy_j (DF)
    if constexpr (MODE == MODE_BWD_DG || MODE == MODE_BWD_DF) {
        const T* src = (MODE == MODE_BWD_DG ? h : xdy) + ob * CS;
#pragma unroll
        for (int c0 = 0; c0 < CS; c0 += 8) {
            typename Vec8<T>::type v = load8<T>(src + c0);
#pragma unroll
            for (int c = 0; c < 8; ++c) oc[c0 + c] = (float)v[c];
        }
    }
    float acc[NACC];
#pragma unroll
    for (int c = 0; c < NACC; ++c) acc[c] = 0.f;

    // ---- sweep the other side
    for (int r0 = 0; r0 < N; r0 += TI) {
        __syncthreads();
        // stage TI rows: other-side DS vector (+ stats if the other side is i) + CS vector
        for (int idx = threadIdx.x; idx < TI * (DS / 8); idx += 256) {
            int r = idx / (DS / 8), d0 = (idx % (DS / 8)) * 8;
            typename Vec8<T>::type v = zero8<T>();
            if (r0 + r < N) v = load8<T>((OWN_IS_I ? f : g) + ((size_t)b * N + r0 + r) * DS + d0);
#pragma unroll
            for (int d = 0; d < 8; ++d) sm[r * ROW + d0 + d] = (float)v[d];
        }
        if (!OWN_IS_I)
            for (int r = threadIdx.x; r < TI; r += 256) {
                float m = 0.f, iz = 0.f, dl = 0.f;
                if (r0 + r < N) {
                    size_t q = (size_t)b * N + r0 + r;
                    m = stats[q * 2]; iz = 1.f / stats[q * 2 + 1];
                    if (MODE == MODE_BWD_DF) dl = delta[q];
                }
                sm[r * ROW + DS] = m; sm[r * ROW + DS + 1] = iz; sm[r * ROW + DS + 2] = dl;
            }
        {
            const T* src = (MODE == MODE_FWD_OUT || MODE == MODE_BWD_DF) ? h : xdy;   // h_i for own=j, dy_j for own=i
            for (int idx = threadIdx.x; idx < TI * (CS / 8); idx += 256) {
                int r = idx / (CS / 8), c0 = (idx % (CS / 8)) * 8;
                typename Vec8<T>::type v = zero8<T>();
                if (r0 + r < N) v = load8<T>(src + ((size_t)b * N + r0 + r) * CS + c0);
#pragma unroll
                for (int c = 0; c < 8; ++c) sm[r * ROW + DS + 4 + c0 + c] = (float)v[c];
            }
        }
        __syncthreads();
        const int rend = min(TI, N - r0);
        for (int r = w * (TI / 4); r < (w + 1) * (TI / 4) && r < rend; ++r) {
            const float* row = sm + r * ROW;
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < DS; ++d) s += ov[d] * row[d];
            const float m = OWN_IS_I ? om : row[DS];
            const float iz = OWN_IS_I ? oinvz : row[DS + 1];
            const float p = __expf(s - m) * iz;                         // beta[i][j]
            if constexpr (ACC_CS) {
#pragma unroll
                for (int c = 0; c < CS; ++c) acc[c] += row[DS + 4 + c] * p;
            } else {
                float dbeta = 0.f;
#pragma unroll
                for (int c = 0; c < CS; ++c) dbeta += oc[c] * row[DS + 4 + c];
                const float dl = OWN_IS_I ? odelta : row[DS + 2];
                const float ds = p * (dbeta - dl);
#pragma unroll
                for (int d = 0; d < DS; ++d) acc[d] += ds * row[d];
            }
        }
    }

    // ---- combine the 4 waves' partial sums through LDS (fixed order), then the epilogue
    __syncthreads();
    float* red = sm;                                     // [4][64][NACC+1]
#pragma unroll
    for (int c = 0; c < NACC; ++c) red[(w * 64 + lane) * (NACC + 1) + c] = acc[c];
    __syncthreads();
    constexpr int PER = NACC / 4;                        // channels finished by each wave
    float fin[PER];
#pragma unroll
    for (int c = 0; c < PER; ++c) {
        float s = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) s += red[(ww * 64 + lane) * (NACC + 1) + w * PER + c];
        fin[c] = s;
    }
    if constexpr (MODE == MODE_BWD_DH) {
        // delta_i = sum_c h[i][c] * dH[i][c]
        float part = 0.f;
        if (valid) {
#pragma unroll
            for (int c = 0; c < PER; ++c) part += (float)h[ob * CS + w * PER + c] * fin[c];
        }
        __syncthreads();
        red[w * 64 + lane] = part;
        __syncthreads();
        if (w == 0 && valid) delta[ob] = red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane];
    }
    if (valid) {
#pragma unroll
        for (int c = 0; c < PER; ++c) {
            float v = fin[c];
            if (MODE == MODE_FWD_OUT) v += (float)xdy[ob * CS + w * PER + c];
            out[ob * NACC + w * PER + c] = (T)v;
        }
    }
}

template <typename T, int DS, int CS, int MODE>
int launch_pass(hipStream_t s, const T* f, const T* g, const T* h, const T* xdy, const float* stats, float* delta, T* out,
                int B, int N) {
    constexpr int ROW = DS + 4 + CS;
    constexpr int NACC = (MODE == MODE_FWD_OUT || MODE == MODE_BWD_DH) ? CS : DS;
    size_t lds = sizeof(float) * (size_t)max(TI * ROW, 256 * (NACC + 1));
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_pass_kernel<T, DS, CS, MODE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, MSAU_LDS_LIMIT);
        if (e != hipSuccess) return msau_set_error(MSAU_ERR_HIP, "attn: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL((attn_pass_kernel<T, DS, CS, MODE>), dim3(cdiv(N, 64), B), dim3(256), lds, s, f, g, h, xdy, stats, delta, out, N);
    MSAU_CHECK_LAUNCH("attn_pass_kernel");
    return 0;
}

template <typename T, int DS, int CS>
int attn_fwd_t(hipStream_t s, const void* f, const void* g, const void* h, const void* x, void* y, float* stats, int B, int N) {
    const T* fp = static_cast<const T*>(f); const T* gp = static_cast<const T*>(g);
    hipLaunchKernelGGL((attn_stats_kernel<T, DS>), dim3(cdiv(N, 256), B), dim3(256), 0, s, fp, gp, stats, N);
    MSAU_CHECK_LAUNCH("attn_stats_kernel");
    return launch_pass<T, DS, CS, MODE_FWD_OUT>(s, fp, gp, static_cast<const T*>(h), static_cast<const T*>(x), stats, nullptr,
                                                static_cast<T*>(y), B, N);
}

template <typename T, int DS, int CS>
int attn_bwd_t(hipStream_t s, const void* f, const void* g, const void* h, const void* dy, const float* stats,
               void* df, void* dg, void* dh, float* ws, int B, int N) {
    const T* fp = static_cast<const T*>(f); const T* gp = static_cast<const T*>(g);
    const T* hp = static_cast<const T*>(h); const T* dyp = static_cast<const T*>(dy);
    int rc = launch_pass<T, DS, CS, MODE_BWD_DH>(s, fp, gp, hp, dyp, stats, ws, static_cast<T*>(dh), B, N);
    if (rc) return rc;
    rc = launch_pass<T, DS, CS, MODE_BWD_DG>(s, fp, gp, hp, dyp, stats, ws, static_cast<T*>(dg), B, N);
    if (rc) return rc;
    return launch_pass<T, DS, CS, MODE_BWD_DF>(s, fp, gp, hp, dyp, stats, ws, static_cast<T*>(df), B, N);
}

// ---- any (Ds, Cs): one workgroup per score row / column, run-time channel counts.  O(N^2 (Ds + Cs)) like the tiled
// kernels but without their register / LDS tiling -- the path for widths that have no instance (the 256-channel bottleneck
// of the reference's constructor defaults, where N is a handful of positions).  Same two-pass softmax arithmetic.
constexpr int kAnyChunk = 2048;            // score-vector chunk held in LDS

template <typename T>
__global__ __launch_bounds__(256) void attn_any_stats_kernel(const T* __restrict__ f, const T* __restrict__ g, float* __restrict__ stats, int N, int Ds) {
    __shared__ float red[256];
    const int b = blockIdx.y, i = blockIdx.x;
    const T* gi = g + ((size_t)b * N + i) * Ds;
    float m = -INFINITY;
    for (int j = threadIdx.x; j < N; j += 256) {
        const T* fj = f + ((size_t)b * N + j) * Ds;
        float sc = 0.f;
        for (int d = 0; d < Ds; ++d) sc += (float)gi[d] * (float)fj[d];
        m = fmaxf(m, sc);
    }
    red[threadIdx.x] = m; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
    m = red[0]; __syncthreads();
    float Z = 0.f;
    for (int j = threadIdx.x; j < N; j += 256) {
        const T* fj = f + ((size_t)b * N + j) * Ds;
        float sc = 0.f;
        for (int d = 0; d < Ds; ++d) sc += (float)gi[d] * (float)fj[d];
        Z += __expf(sc - m);
    }
    red[threadIdx.x] = Z; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) { stats[((size_t)b * N + i) * 2] = m; stats[((size_t)b * N + i) * 2 + 1] = red[0]; }
}

// MODE 0: y[j] = x[j] + sum_i beta[i,j] h[i]      (block = column j)
// MODE 1: dh[i] = sum_j beta[i,j] dy[j], delta[i] = h[i] . dh[i]   (block = row i)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void attn_any_mix_kernel(const T* __restrict__ f, const T* __restrict__ g, const T* __restrict__ h,
                                                           const T* __restrict__ xdy, const float* __restrict__ stats, float* __restrict__ delta,
                                                           T* __restrict__ out, int N, int Ds, int Cs) {
    __shared__ float beta[kAnyChunk];
    __shared__ float red[256];
    const int b = blockIdx.y, me = blockIdx.x;
    const size_t base = (size_t)b * N;
    const T* mine = (MODE == 0 ? f : g) + (base + me) * Ds;          // f_j (column) or g_i (row)
    float acc[4] = {0.f, 0.f, 0.f, 0.f};                             // channels tid, tid + 256, ... (Cs <= 1024)
    for (int o0 = 0; o0 < N; o0 += kAnyChunk) {
        const int no = min(kAnyChunk, N - o0);
        __syncthreads();
        for (int t = threadIdx.x; t < no; t += 256) {
            const int o = o0 + t;
            const int row = MODE == 0 ? o : me;                       // softmax row of this score
            const T* other = (MODE == 0 ? g : f) + (base + o) * Ds;
            float sc = 0.f;
            for (int d = 0; d < Ds; ++d) sc += (float)mine[d] * (float)other[d];
            beta[t] = __expf(sc - stats[(base + row) * 2]) / stats[(base + row) * 2 + 1];
        }
        __syncthreads();
        const T* src = MODE == 0 ? h : xdy;                            // h[i] or dy[j]
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = threadIdx.x + 256 * k;
            if (c < Cs) {
                float a = acc[k];
                for (int t = 0; t < no; ++t) a += beta[t] * (float)src[(base + o0 + t) * Cs + c];
                acc[k] = a;
            }
        }
    }
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = threadIdx.x + 256 * k;
        if (c < Cs) {
            if (MODE == 0) out[(base + me) * Cs + c] = (T)((float)xdy[(base + me) * Cs + c] + acc[k]);
            else { out[(base + me) * Cs + c] = (T)acc[k]; dot += (float)h[(base + me) * Cs + c] * acc[k]; }
        }
    }
    if (MODE == 1) {
        red[threadIdx.x] = dot; __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
        if (threadIdx.x == 0) delta[base + me] = red[0];
    }
}

// MODE 2: dg[i] = sum_j dS[i,j] f[j]   (block = row i)      MODE 3: df[j] = sum_i dS[i,j] g[i]   (block = column j)
//   dS[i,j] = beta[i,j] (h[i] . dy[j] - delta[i])
template <typename T, int MODE>
__global__ __launch_bounds__(256) void attn_any_dscore_kernel(const T* __restrict__ f, const T* __restrict__ g, const T* __restrict__ h,
                                                              const T* __restrict__ dy, const float* __restrict__ stats, const float* __restrict__ delta,
                                                              T* __restrict__ out, int N, int Ds, int Cs) {
    __shared__ float ds[kAnyChunk];
    const int b = blockIdx.y, me = blockIdx.x;
    const size_t base = (size_t)b * N;
    const T* mine = (MODE == 2 ? g : f) + (base + me) * Ds;
    const T* hv = (MODE == 2 ? h : dy) + (base + me) * Cs;            // h[i] (row) or dy[j] (column)
    float acc = 0.f;                                                  // output channel tid (Ds <= 256)
    for (int o0 = 0; o0 < N; o0 += kAnyChunk) {
        const int no = min(kAnyChunk, N - o0);
        __syncthreads();
        for (int t = threadIdx.x; t < no; t += 256) {
            const int o = o0 + t;
            const int row = MODE == 2 ? me : o;
            const T* other = (MODE == 2 ? f : g) + (base + o) * Ds;
            const T* ov = (MODE == 2 ? dy : h) + (base + o) * Cs;
            float sc = 0.f, hd = 0.f;
            for (int d = 0; d < Ds; ++d) sc += (float)mine[d] * (float)other[d];
            for (int c = 0; c < Cs; ++c) hd += (float)hv[c] * (float)ov[c];
            const float bt = __expf(sc - stats[(base + row) * 2]) / stats[(base + row) * 2 + 1];
            ds[t] = bt * (hd - delta[base + row]);
        }
        __syncthreads();
        if ((int)threadIdx.x < Ds) {
            const T* src = (MODE == 2 ? f : g);
            for (int t = 0; t < no; ++t) acc += ds[t] * (float)src[(base + o0 + t) * Ds + threadIdx.x];
        }
    }
    if ((int)threadIdx.x < Ds) out[(base + me) * Ds + threadIdx.x] = (T)acc;
}

template <typename T>
int attn_any_fwd(hipStream_t s, const void* f, const void* g, const void* h, const void* x, void* y, float* stats, int B, int N, int Ds, int Cs) {
    MSAU_CHECK_ARG(Ds % 8 == 0 && Cs % 8 == 0 && Ds <= 256 && Cs <= 1024 && N <= 65535 && B <= 65535, "selfattn: unsupported (Ds,Cs,N)=(%d,%d,%d)", Ds, Cs, N);
    const T* fp = static_cast<const T*>(f); const T* gp = static_cast<const T*>(g);
    hipLaunchKernelGGL(attn_any_stats_kernel<T>, dim3(N, B), dim3(256), 0, s, fp, gp, stats, N, Ds);
    hipLaunchKernelGGL((attn_any_mix_kernel<T, 0>), dim3(N, B), dim3(256), 0, s, fp, gp, static_cast<const T*>(h), static_cast<const T*>(x), stats,
                       nullptr, static_cast<T*>(y), N, Ds, Cs);
    MSAU_CHECK_LAUNCH("attn_any_fwd");
    return 0;
}

template <typename T>
int attn_any_bwd(hipStream_t s, const void* f, const void* g, const void* h, const void* dy, const float* stats, void* df, void* dg, void* dh,
                 float* ws, int B, int N, int Ds, int Cs) {
    MSAU_CHECK_ARG(Ds % 8 == 0 && Cs % 8 == 0 && Ds <= 256 && Cs <= 1024 && N <= 65535 && B <= 65535, "selfattn: unsupported (Ds,Cs,N)=(%d,%d,%d)", Ds, Cs, N);
    const T* fp = static_cast<const T*>(f); const T* gp = static_cast<const T*>(g);
    const T* hp = static_cast<const T*>(h); const T* dyp = static_cast<const T*>(dy);
    hipLaunchKernelGGL((attn_any_mix_kernel<T, 1>), dim3(N, B), dim3(256), 0, s, fp, gp, hp, dyp, stats, ws, static_cast<T*>(dh), N, Ds, Cs);
    hipLaunchKernelGGL((attn_any_dscore_kernel<T, 2>), dim3(N, B), dim3(256), 0, s, fp, gp, hp, dyp, stats, ws, static_cast<T*>(dg), N, Ds, Cs);
    hipLaunchKernelGGL((attn_any_dscore_kernel<T, 3>), dim3(N, B), dim3(256), 0, s, fp, gp, hp, dyp, stats, ws, static_cast<T*>(df), N, Ds, Cs);
    MSAU_CHECK_LAUNCH("attn_any_bwd");
    return 0;
}

#define ATTN_DISPATCH(FN, ...)                                                                      \
    do {                                                                                            \
        if (Ds == 8 && Cs == 8) return FN<T, 8, 8>(__VA_ARGS__);                                    \
        if (Ds == 8 && Cs == 16) return FN<T, 8, 16>(__VA_ARGS__);                                  \
        if (Ds == 8 && Cs == 32) return FN<T, 8, 32>(__VA_ARGS__);                                  \
        if (Ds == 8 && Cs == 64) return FN<T, 8, 64>(__VA_ARGS__);                                  \
        if (Ds == 16 && Cs == 128) return FN<T, 16, 128>(__VA_ARGS__);                              \
    } while (0)

template <typename T>
int attn_fwd_d(hipStream_t s, const void* f, const void* g, const void* h, const void* x, void* y, float* stats,
               int B, int N, int Ds, int Cs) {
    ATTN_DISPATCH(attn_fwd_t, s, f, g, h, x, y, stats, B, N);
    return attn_any_fwd<T>(s, f, g, h, x, y, stats, B, N, Ds, Cs);
}
template <typename T>
int attn_bwd_d(hipStream_t s, const void* f, const void* g, const void* h, const void* dy, const float* stats,
               void* df, void* dg, void* dh, float* ws, int B, int N, int Ds, int Cs) {
    ATTN_DISPATCH(attn_bwd_t, s, f, g, h, dy, stats, df, dg, dh, ws, B, N);
    return attn_any_bwd<T>(s, f, g, h, dy, stats, df, dg, dh, ws, B, N, Ds, Cs);
}

}  // namespace

// bf16 instances on the matrix cores (attention_mfma.hip)
int msau_attn_mfma_supported(int Ds, int Cs, int N);
int msau_attn_mfma_fwd(hipStream_t s, const void* f, const void* g, const void* h, const void* x, void* y, float* stats,
                       int B, int N, int Ds, int Cs);
int msau_attn_mfma_bwd(hipStream_t s, const void* f, const void* g, const void* h, const void* dy, const float* stats,
                       void* df, void* dg, void* dh, float* ws, int B, int N, int Ds, int Cs);

extern "C" int msau_selfattn_fwd(void* stream, int dtype, const void* f, const void* g, const void* h, const void* x, void* y,
                                 float* stats, int B, int N, int Ds, int Cs) {
    MSAU_CHECK_ARG(f && g && h && x && y && stats && B > 0 && N > 0, "selfattn_fwd: bad args");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == MSAU_F32) return attn_fwd_d<float>(s, f, g, h, x, y, stats, B, N, Ds, Cs);
    if (dtype == MSAU_BF16 && msau_attn_mfma_supported(Ds, Cs, N)) return msau_attn_mfma_fwd(s, f, g, h, x, y, stats, B, N, Ds, Cs);
    if (dtype == MSAU_BF16) return attn_fwd_d<bf16_t>(s, f, g, h, x, y, stats, B, N, Ds, Cs);
    return msau_set_error(MSAU_ERR_ARG, "selfattn_fwd: bad dtype");
}

extern "C" int msau_selfattn_bwd(void* stream, int dtype, const void* f, const void* g, const void* h, const void* dy,
                                 const float* stats, void* df, void* dg, void* dh, float* ws, int B, int N, int Ds, int Cs) {
    MSAU_CHECK_ARG(f && g && h && dy && stats && df && dg && dh && ws && B > 0 && N > 0, "selfattn_bwd: bad args");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == MSAU_F32) return attn_bwd_d<float>(s, f, g, h, dy, stats, df, dg, dh, ws, B, N, Ds, Cs);
    if (dtype == MSAU_BF16 && msau_attn_mfma_supported(Ds, Cs, N))
        return msau_attn_mfma_bwd(s, f, g, h, dy, stats, df, dg, dh, ws, B, N, Ds, Cs);
    if (dtype == MSAU_BF16) return attn_bwd_d<bf16_t>(s, f, g, h, dy, stats, df, dg, dh, ws, B, N, Ds, Cs);
    return msau_set_error(MSAU_ERR_ARG, "selfattn_bwd: bad dtype");
}
