// Box convolution (Burkov & Lempitsky, "Deep Neural Networks with Box Convolutions", NeurIPS 2018) for the
// model/model_box.py variant of MSAU (MultiBoxConvBlock, model_box.py:9-59: BoxConv2d(c, 3, 28, 28) -> 1x1 conv).
//
// PARITY UNPINNED: the reference imports `BoxConv2d` from the third-party package `box_convolution`
// (github shrubb/box-convolutions), which is neither vendored nor installed here and has no fixtures in the reference
// tree.  The arithmetic below follows the published definition and is checked against this repository's own CPU
// restatement (oracle/box_oracle.py) only -- "self-consistent", not "reference-identical".
//
// Definition used (pixels are unit squares [i, i+1) x [j, j+1), the image is zero outside):
//   out[b, y, x, c*F + f] = 1/A * integral over rows [y + hmin, y + hmax + 1) x columns [x + wmin, x + wmax + 1) of in[b, ., ., c]
//   A = (hmax - hmin + 1) * (wmax - wmin + 1)              (normalised box filter; a box with min = max = 0 is one pixel)
//   (hmin, hmax, wmin, wmax)[c][f] = stored parameter * reparametrisation (max box size): real-valued, learnable.
// The integral of a piecewise-constant image is piecewise bilinear in the corner position, so it is the bilinear
// interpolation of the integral image II (fp32, channel-planar [B][C][H+1][W+1], II[i][j] = sum in[0..i) x [0..j)):
// four corners x four taps.  Gradients:
//   d in : the same box filter applied to the output gradient with the reflected box (-hmax, -hmin, -wmax, -wmin),
//          summed over the F filters of a channel (msau_box_filter with sum_filters = 1 on II of the gradient);
//   d box: line integrals of the image along the box edges (also read from II), reduced over all pixels in a fixed
//          order (per-workgroup partials, then one ordered pass): msau_box_param_grad.
// Layout: activations NHWC like everything else; II planar so that a wave's 64 consecutive columns of one channel are
// one 256-byte run.  Everything here is HBM / L2 bound gather work -- no MFMA.
#include "msau_common.h"

namespace {

// ---- integral image, pass A: vertical running sums, NHWC T -> planar fp32 [B][C][H+1][W+1] (row 0 / column 0 zero).
// A block owns 64 columns x all channels x one SEGMENT of rows (kSegRows): a single block per column strip would walk
// all H rows serially (96 blocks for a 512 x 384 image on 256 CUs).  The segment's starting sums come from a first
// kernel that only adds up each segment's column totals (the bf16 input is small; reading it twice is cheap).
constexpr int kSegRows = 32;

template <typename T>
__global__ __launch_bounds__(256) void box_segment_sums_kernel(const T* __restrict__ in, float* __restrict__ segsum, int H, int W, int Cs, int C,
                                                               int relu_in) {
    // segsum[b][seg][c][x] = sum over the rows of the segment; thread = (x, c) pairs of a 64-column strip
    const int b = blockIdx.y, seg = blockIdx.z, x0 = blockIdx.x * 64;
    const int nx = min(64, W - x0);
    const int y0 = seg * kSegRows, y1 = min(H, y0 + kSegRows);
    const int nseg = gridDim.z;
    for (int i = threadIdx.x; i < nx * Cs; i += 256) {
        const int xx = i / Cs, c = i - xx * Cs;
        if (c >= C) continue;
        float acc = 0.f;
        for (int y = y0; y < y1; ++y) {
            const float v = (float)in[(((int64_t)b * H + y) * W + x0) * Cs + i];
            acc += relu_in ? fmaxf(v, 0.f) : v;
        }
        segsum[(((int64_t)b * nseg + seg) * C + c) * W + x0 + xx] = acc;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void box_integral_cols_kernel(const T* __restrict__ in, const float* __restrict__ segsum, float* __restrict__ ii,
                                                                int H, int W, int Cs, int C, int relu_in) {
    extern __shared__ float tile[];                         // [C][65] row tile, then [C][64] running sums
    const int b = blockIdx.y, seg = blockIdx.z, x0 = blockIdx.x * 64;
    const int nseg = gridDim.z;
    const int nx = min(64, W - x0);
    const int y0 = seg * kSegRows, y1 = min(H, y0 + kSegRows);
    const int64_t plane = (int64_t)(H + 1) * (W + 1);
    float* out_b = ii + (int64_t)b * C * plane;
    float* acc = tile + C * 65;
    for (int i = threadIdx.x; i < C * 64; i += 256) {
        const int c = i >> 6, xx = i & 63;
        float a = 0.f;
        if (xx < nx)
            for (int s2 = 0; s2 < seg; ++s2) a += segsum[(((int64_t)b * nseg + s2) * C + c) * W + x0 + xx];      // fixed order
        acc[i] = a;
    }
    if (seg == 0) {                                          // row 0 and column 0 of II are zero
        for (int i = threadIdx.x; i < C * nx; i += 256) {
            const int c = i / nx, xx = i - c * nx;
            out_b[(int64_t)c * plane + x0 + xx + 1] = 0.f;
        }
        if (x0 == 0)
            for (int i = threadIdx.x; i < C * (H + 1); i += 256) {
                const int c = i / (H + 1), y = i - c * (H + 1);
                out_b[(int64_t)c * plane + (int64_t)y * (W + 1)] = 0.f;
            }
    }
    __syncthreads();
    for (int y = y0; y < y1; ++y) {
        const T* row = in + (((int64_t)b * H + y) * W + x0) * Cs;
        for (int i = threadIdx.x; i < nx * Cs; i += 256) {
            const int xx = i / Cs, c = i - xx * Cs;
            if (c < C) tile[c * 65 + xx] = relu_in ? fmaxf((float)row[i], 0.f) : (float)row[i];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < C * nx; i += 256) {
            const int c = i / nx, xx = i - c * nx;
            const float v = acc[c * 64 + xx] + tile[c * 65 + xx];
            acc[c * 64 + xx] = v;
            out_b[(int64_t)c * plane + (int64_t)(y + 1) * (W + 1) + x0 + xx + 1] = v;
        }
        __syncthreads();
    }
}

// ---- pass B: horizontal inclusive scan of every row of every plane, in place; one wave per row
__global__ __launch_bounds__(256) void box_integral_rows_kernel(float* __restrict__ ii, int64_t nrows, int W1) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    float* p = ii + row * W1;
    float carry = 0.f;
    for (int x0 = 0; x0 < W1; x0 += 64) {
        const int x = x0 + lane;
        float v = x < W1 ? p[x] : 0.f;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float n = __shfl_up(v, o, 64);
            if (lane >= o) v += n;
        }
        v += carry;
        if (x < W1) p[x] = v;
        carry = __shfl(v, 63, 64);
    }
}

__device__ __forceinline__ float lerp_ii(const float* __restrict__ plane, int W1, int i, float fi, int j, float fj, int H, int W) {
    // bilinear interpolation of II at (i + fi, j + fj), i in [0, H], j in [0, W]; taps beyond the last row / column have
    // weight 0 there (fi = 0 when i == H), so clamp the index instead of branching
    const int i1 = min(i + 1, H), j1 = min(j + 1, W);
    const float a = plane[(int64_t)i * W1 + j], b = plane[(int64_t)i * W1 + j1];
    const float c = plane[(int64_t)i1 * W1 + j], d = plane[(int64_t)i1 * W1 + j1];
    const float top = a + fj * (b - a), bot = c + fj * (d - c);
    return top + fi * (bot - top);
}

__device__ __forceinline__ void split(float v, float hi, int& i, float& f) {
    v = fminf(fmaxf(v, 0.f), hi);
    const float fl = floorf(v);
    i = (int)fl;
    f = v - fl;
}

// ---- the box filter itself.  grid (x tiles of 64, y, b); a wave takes (channel, filter) pairs round-robin, lane = column.
// params: fp32 [4][C][F] = hmin, hmax, wmin, wmax in PIXELS (already multiplied by the reparametrisation; for the input
// gradient the caller passes the reflected boxes).  SUMF: out[c] = sum_f (input gradient), else out[c*F + f].
template <typename T, bool SUMF>
__global__ __launch_bounds__(256) void box_filter_kernel(const float* __restrict__ ii, const float* __restrict__ params, T* __restrict__ out,
                                                         int H, int W, int C, int F, int Cs_out, int accumulate,
                                                         const T* __restrict__ mask_a, const T* __restrict__ add, const T* __restrict__ mask_b) {
    extern __shared__ float stile[];                        // [64][NOUT + 1]
    const int NOUT = SUMF ? C : C * F;
    const int b = blockIdx.z, y = blockIdx.y, x0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = x0 + lane;
    const int W1 = W + 1;
    const int64_t plane = (int64_t)(H + 1) * W1;
    const int CF = C * F;
    for (int c = wave; c < C; c += 4) {
        float sumf = 0.f;
        for (int f = 0; f < F; ++f) {
            const int p = c * F + f;
            // forward: the integral image of input channel c; input gradient (SUMF): of output-gradient channel c*F + f
            const float* pl = ii + ((int64_t)b * (SUMF ? CF : C) + (SUMF ? p : c)) * plane;
            const float hmin = params[p], hmax = params[CF + p], wmin = params[2 * CF + p], wmax = params[3 * CF + p];
            const float inv_area = 1.f / ((hmax - hmin + 1.f) * (wmax - wmin + 1.f));
            int i1, i2, j1, j2;
            float fi1, fi2, fj1, fj2;
            split((float)y + hmin, (float)H, i1, fi1);
            split((float)y + hmax + 1.f, (float)H, i2, fi2);
            split((float)x + wmin, (float)W, j1, fj1);
            split((float)x + wmax + 1.f, (float)W, j2, fj2);
            const float s = lerp_ii(pl, W1, i2, fi2, j2, fj2, H, W) - lerp_ii(pl, W1, i1, fi1, j2, fj2, H, W)
                          - lerp_ii(pl, W1, i2, fi2, j1, fj1, H, W) + lerp_ii(pl, W1, i1, fi1, j1, fj1, H, W);
            const float v = s * inv_area;
            if (SUMF) sumf += v;
            else stile[lane * (NOUT + 1) + p] = v;
        }
        if (SUMF) stile[lane * (NOUT + 1) + c] = sumf;
    }
    __syncthreads();
    // coalesced NHWC store (padded channels written as zero)
    const int nx = min(64, W - x0);
    T* orow = out + (((int64_t)b * H + y) * W + x0) * Cs_out;
    // epilogue in the order of msau_conv2d: v *= (mask_a > 0); v += add; v += old (accumulate); v *= (mask_b > 0)
    const int64_t rowoff = (((int64_t)b * H + y) * W + x0) * Cs_out;
    for (int i = threadIdx.x; i < nx * Cs_out; i += 256) {
        const int xx = i / Cs_out, ch = i - xx * Cs_out;
        float v = ch < NOUT ? stile[xx * (NOUT + 1) + ch] : 0.f;
        if (mask_a) v = (float)mask_a[rowoff + i] > 0.f ? v : 0.f;
        if (add) v += (float)add[rowoff + i];
        if (accumulate) v += (float)orow[i];
        if (mask_b) v = (float)mask_b[rowoff + i] > 0.f ? v : 0.f;
        orow[i] = (T)v;
    }
}

// ---- gradient w.r.t. the box parameters.  Same tiling; every workgroup covers ROWS consecutive rows and writes one
// partial per (pair, parameter): partials[block][4][C*F] (pixel units; the caller scales by the reparametrisation).
//   S   = box integral, O = S / A
//   dS/dhmax =  row integral of row floor(y + hmax + 1) over the box columns      dS/dhmin = -row integral of row floor(y + hmin)
//   dS/dwmax =  column integral of column floor(x + wmax + 1) over the box rows   dS/dwmin = -column integral of column floor(x + wmin)
//   dO/dhmax = (dS/dhmax - O * (wmax - wmin + 1)) / A   ... (A depends on the parameters through the normalisation)
constexpr int kPgRows = 8;
template <typename T>
__global__ __launch_bounds__(256) void box_pgrad_kernel(const float* __restrict__ ii, const float* __restrict__ params, const T* __restrict__ gout,
                                                        float* __restrict__ partials, int H, int W, int C, int F, int Cs_out) {
    extern __shared__ float gtile[];                        // [64][CF + 1] gradient tile, then [4][CF] block sums
    const int CF = C * F;
    const int b = blockIdx.z, y0 = blockIdx.y * kPgRows, x0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = x0 + lane;
    const int W1 = W + 1;
    const int64_t plane = (int64_t)(H + 1) * W1;
    float* bsum = gtile + 64 * (CF + 1);
    for (int i = threadIdx.x; i < 4 * CF; i += 256) bsum[i] = 0.f;
    const int nx = min(64, W - x0);
    for (int y = y0; y < min(y0 + kPgRows, H); ++y) {
        __syncthreads();
        const T* grow = gout + (((int64_t)b * H + y) * W + x0) * Cs_out;
        for (int i = threadIdx.x; i < 64 * Cs_out; i += 256) {
            const int xx = i / Cs_out, ch = i - xx * Cs_out;
            if (ch < CF) gtile[xx * (CF + 1) + ch] = xx < nx ? (float)grow[i] : 0.f;
        }
        __syncthreads();
        for (int p = wave; p < CF; p += 4) {
            const int c = p / F;
            const float* pl = ii + ((int64_t)b * C + c) * plane;
            const float hmin = params[p], hmax = params[CF + p], wmin = params[2 * CF + p], wmax = params[3 * CF + p];
            const float hh = hmax - hmin + 1.f, ww = wmax - wmin + 1.f;
            const float inv_area = 1.f / (hh * ww);
            const float g = gtile[lane * (CF + 1) + p];
            int i1, i2, j1, j2;
            float fi1, fi2, fj1, fj2;
            const float r1 = (float)y + hmin, r2 = (float)y + hmax + 1.f, c1 = (float)x + wmin, c2 = (float)x + wmax + 1.f;
            split(r1, (float)H, i1, fi1);
            split(r2, (float)H, i2, fi2);
            split(c1, (float)W, j1, fj1);
            split(c2, (float)W, j2, fj2);
            // the 4 x 4 integral-image values around the four corners, loaded once: the box integral AND the four edge
            // (line) integrals are combinations of them
            const int rr[4] = {i1, min(i1 + 1, H), i2, min(i2 + 1, H)};
            const int cc[4] = {j1, min(j1 + 1, W), j2, min(j2 + 1, W)};
            float v[4][4];
#pragma unroll
            for (int a2 = 0; a2 < 4; ++a2)
#pragma unroll
                for (int b2 = 0; b2 < 4; ++b2) v[a2][b2] = pl[(int64_t)rr[a2] * W1 + cc[b2]];
            float A[4], Bc[4], dA[4], dB[4];                    // per row: value at column c1 / c2, column derivative there
#pragma unroll
            for (int a2 = 0; a2 < 4; ++a2) {
                dA[a2] = v[a2][1] - v[a2][0];
                dB[a2] = v[a2][3] - v[a2][2];
                A[a2] = v[a2][0] + fj1 * dA[a2];
                Bc[a2] = v[a2][2] + fj2 * dB[a2];
            }
            const float F11 = A[0] + fi1 * (A[1] - A[0]), F21 = A[2] + fi2 * (A[3] - A[2]);
            const float F12 = Bc[0] + fi1 * (Bc[1] - Bc[0]), F22 = Bc[2] + fi2 * (Bc[3] - Bc[2]);
            const float O = (F22 - F12 - F21 + F11) * inv_area;
            // an edge that is clamped to the image border does not move the integral
            const bool r1in = r1 > 0.f && r1 < (float)H, r2in = r2 > 0.f && r2 < (float)H;
            const bool c1in = c1 > 0.f && c1 < (float)W, c2in = c2 > 0.f && c2 < (float)W;
            const float dS_hmax = r2in ? (Bc[3] - Bc[2]) - (A[3] - A[2]) : 0.f;              // row integral along the bottom edge
            const float dS_hmin = r1in ? -((Bc[1] - Bc[0]) - (A[1] - A[0])) : 0.f;
            const float dS_wmax = c2in ? (dB[2] + fi2 * (dB[3] - dB[2])) - (dB[0] + fi1 * (dB[1] - dB[0])) : 0.f;   // column integral, right edge
            const float dS_wmin = c1in ? -((dA[2] + fi2 * (dA[3] - dA[2])) - (dA[0] + fi1 * (dA[1] - dA[0]))) : 0.f;
            float d4[4];
            d4[0] = g * (dS_hmin + O * ww) * inv_area;          // dA/dhmin = -ww
            d4[1] = g * (dS_hmax - O * ww) * inv_area;
            d4[2] = g * (dS_wmin + O * hh) * inv_area;
            d4[3] = g * (dS_wmax - O * hh) * inv_area;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float v = lane < nx ? d4[k] : 0.f;
                for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
                if (lane == 0) bsum[k * CF + p] += v;           // pair p belongs to this wave only: no race
            }
        }
    }
    __syncthreads();
    const int64_t blk = ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    for (int i = threadIdx.x; i < 4 * CF; i += 256) partials[blk * 4 * CF + i] = bsum[i];
}

// ordered sum of the partials -> flat gradient: grads[off[k] + p] (+)= scale[k] * sum_blk partials[blk][k][p]
__global__ __launch_bounds__(256) void box_pgrad_reduce_kernel(const float* __restrict__ partials, int64_t nblk, int CF, float* __restrict__ grads,
                                                               int64_t off0, int64_t off1, int64_t off2, int64_t off3, float scale_h, float scale_w) {
    __shared__ float red[4];
    const int e = blockIdx.x;                                  // element k*CF + p
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < nblk; i += 256) s += partials[i * 4 * CF + e];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = (red[0] + red[1]) + (red[2] + red[3]);
        const int k = e / CF, p = e - k * CF;
        const int64_t off = k == 0 ? off0 : k == 1 ? off1 : k == 2 ? off2 : off3;
        grads[off + p] = t * (k < 2 ? scale_h : scale_w);
    }
}

// stored parameters -> pixel units (and the reflected boxes for the input gradient), with the box kept valid:
//   |edge| <= max size, hmax >= hmin, wmax >= wmin (extent of at least one pixel)
__global__ void box_params_kernel(const float* __restrict__ flat, int64_t off0, int64_t off1, int64_t off2, int64_t off3, int CF,
                                  float rh, float rw, float* __restrict__ fwd, float* __restrict__ refl) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= CF) return;
    float hmin = flat[off0 + p] * rh, hmax = flat[off1 + p] * rh, wmin = flat[off2 + p] * rw, wmax = flat[off3 + p] * rw;
    hmin = fminf(fmaxf(hmin, -rh), rh); hmax = fminf(fmaxf(hmax, -rh), rh);
    wmin = fminf(fmaxf(wmin, -rw), rw); wmax = fminf(fmaxf(wmax, -rw), rw);
    hmax = fmaxf(hmax, hmin); wmax = fmaxf(wmax, wmin);
    fwd[p] = hmin; fwd[CF + p] = hmax; fwd[2 * CF + p] = wmin; fwd[3 * CF + p] = wmax;
    refl[p] = -hmax; refl[CF + p] = -hmin; refl[2 * CF + p] = -wmax; refl[3 * CF + p] = -wmin;
}

}  // namespace

extern "C" int64_t msau_box_integral_ws_floats(int B, int H, int W, int C) { return (int64_t)B * cdiv(H, kSegRows) * C * W; }

extern "C" int msau_box_integral(void* stream, int dtype, const void* in, float* ii, float* ws, int B, int H, int W, int C, int Cs, int relu_in) {
    MSAU_CHECK_ARG(in && ii && ws && B > 0 && H > 0 && W > 0 && C > 0 && C <= Cs && Cs % 8 == 0 && C <= 512, "box_integral: bad args");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t lds = (size_t)C * (65 + 64) * sizeof(float);
    MSAU_CHECK_ARG(lds <= 150 * 1024, "box_integral: %d channels exceed the LDS tile", C);
    const int nseg = cdiv(H, kSegRows);
    MSAU_CHECK_ARG(nseg <= 65535 && B <= 65535, "box_integral: image too tall / batch too large");
    dim3 grid(cdiv(W, 64), B, nseg);
    if (dtype == MSAU_F32) {
        hipLaunchKernelGGL(box_segment_sums_kernel<float>, grid, dim3(256), 0, s, static_cast<const float*>(in), ws, H, W, Cs, C, relu_in);
        if (lds > 60 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&box_integral_cols_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, MSAU_LDS_LIMIT);
        hipLaunchKernelGGL(box_integral_cols_kernel<float>, grid, dim3(256), lds, s, static_cast<const float*>(in), ws, ii, H, W, Cs, C, relu_in);
    } else if (dtype == MSAU_BF16) {
        hipLaunchKernelGGL(box_segment_sums_kernel<bf16_t>, grid, dim3(256), 0, s, static_cast<const bf16_t*>(in), ws, H, W, Cs, C, relu_in);
        if (lds > 60 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&box_integral_cols_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, MSAU_LDS_LIMIT);
        hipLaunchKernelGGL(box_integral_cols_kernel<bf16_t>, grid, dim3(256), lds, s, static_cast<const bf16_t*>(in), ws, ii, H, W, Cs, C, relu_in);
    } else return msau_set_error(MSAU_ERR_ARG, "box_integral: bad dtype");
    MSAU_CHECK_LAUNCH("box_integral_cols");
    const int64_t nrows = (int64_t)B * C * (H + 1);
    MSAU_CHECK_ARG(cdiv64(nrows, 4) < (1ll << 31), "box_integral: too many rows");
    hipLaunchKernelGGL(box_integral_rows_kernel, dim3((unsigned)cdiv64(nrows, 4)), dim3(256), 0, s, ii, nrows, W + 1);
    MSAU_CHECK_LAUNCH("box_integral_rows");
    return 0;
}

extern "C" int msau_box_params(void* stream, const float* flat_params, int64_t off_hmin, int64_t off_hmax, int64_t off_wmin, int64_t off_wmax,
                               int C, int F, float max_h, float max_w, float* params_fwd, float* params_refl) {
    MSAU_CHECK_ARG(flat_params && params_fwd && params_refl && C > 0 && F > 0 && max_h > 0 && max_w > 0, "box_params: bad args");
    hipLaunchKernelGGL(box_params_kernel, dim3(cdiv(C * F, 64)), dim3(64), 0, static_cast<hipStream_t>(stream), flat_params,
                       off_hmin, off_hmax, off_wmin, off_wmax, C * F, max_h, max_w, params_fwd, params_refl);
    MSAU_CHECK_LAUNCH("box_params");
    return 0;
}

extern "C" int msau_box_filter(void* stream, int dtype, const float* ii, const float* params, void* out, int B, int H, int W, int C, int F,
                               int Cs_out, int sum_filters, int accumulate, const void* mask_a, const void* add, const void* mask_b) {
    MSAU_CHECK_ARG(ii && params && out && B > 0 && H > 0 && W > 0 && C > 0 && F > 0 && Cs_out % 8 == 0, "box_filter: bad args");
    const int nout = sum_filters ? C : C * F;
    MSAU_CHECK_ARG(nout <= Cs_out, "box_filter: %d output channels do not fit the stored %d", nout, Cs_out);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t lds = (size_t)64 * (nout + 1) * sizeof(float);
    MSAU_CHECK_ARG(lds <= 60 * 1024, "box_filter: too many output channels (%d)", nout);
    dim3 grid(cdiv(W, 64), H, B);
#define BOXF(T, S) hipLaunchKernelGGL((box_filter_kernel<T, S>), grid, dim3(256), lds, s, ii, params, static_cast<T*>(out), H, W, C, F, Cs_out, accumulate, \
                                   static_cast<const T*>(mask_a), static_cast<const T*>(add), static_cast<const T*>(mask_b))
    if (dtype == MSAU_F32) { if (sum_filters) BOXF(float, true); else BOXF(float, false); }
    else if (dtype == MSAU_BF16) { if (sum_filters) BOXF(bf16_t, true); else BOXF(bf16_t, false); }
    else return msau_set_error(MSAU_ERR_ARG, "box_filter: bad dtype");
#undef BOXF
    MSAU_CHECK_LAUNCH("box_filter");
    return 0;
}

extern "C" int64_t msau_box_pgrad_ws_floats(int B, int H, int W, int C, int F) {
    return (int64_t)B * cdiv(H, kPgRows) * cdiv(W, 64) * 4 * C * F;
}

extern "C" int msau_box_param_grad(void* stream, int dtype, const float* ii, const float* params, const void* gout, float* ws, float* flat_grads,
                                   int64_t off_hmin, int64_t off_hmax, int64_t off_wmin, int64_t off_wmax, int B, int H, int W, int C, int F,
                                   int Cs_out, float max_h, float max_w) {
    MSAU_CHECK_ARG(ii && params && gout && ws && flat_grads && B > 0 && H > 0 && W > 0 && C > 0 && F > 0 && C * F <= Cs_out, "box_param_grad: bad args");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int CF = C * F;
    const size_t lds = ((size_t)64 * (CF + 1) + 4 * CF) * sizeof(float);
    MSAU_CHECK_ARG(lds <= 60 * 1024, "box_param_grad: too many channels (%d)", CF);
    dim3 grid(cdiv(W, 64), cdiv(H, kPgRows), B);
    if (dtype == MSAU_F32)
        hipLaunchKernelGGL(box_pgrad_kernel<float>, grid, dim3(256), lds, s, ii, params, static_cast<const float*>(gout), ws, H, W, C, F, Cs_out);
    else if (dtype == MSAU_BF16)
        hipLaunchKernelGGL(box_pgrad_kernel<bf16_t>, grid, dim3(256), lds, s, ii, params, static_cast<const bf16_t*>(gout), ws, H, W, C, F, Cs_out);
    else return msau_set_error(MSAU_ERR_ARG, "box_param_grad: bad dtype");
    MSAU_CHECK_LAUNCH("box_pgrad");
    const int64_t nblk = (int64_t)grid.x * grid.y * grid.z;
    hipLaunchKernelGGL(box_pgrad_reduce_kernel, dim3(4 * CF), dim3(256), 0, s, ws, nblk, CF, flat_grads, off_hmin, off_hmax, off_wmin, off_wmax, max_h, max_w);
    MSAU_CHECK_LAUNCH("box_pgrad_reduce");
    return 0;
}

// ---- launch-sequence records (msau_run_ops): one box conv forward / backward sweep step
extern "C" int msau_box_fwd(void* stream, int dtype, const msau_box_args* a) {
    MSAU_CHECK_ARG(a && a->in && a->ii && a->ws_ii && a->params_fwd && a->out, "box_fwd: null pointer");
    int rc = msau_box_integral(stream, dtype, a->in, a->ii, a->ws_ii, a->B, a->H, a->W, a->C, a->Cs_in, a->relu_in);
    if (rc) return rc;
    return msau_box_filter(stream, dtype, a->ii, a->params_fwd, a->out, a->B, a->H, a->W, a->C, a->F, a->Cs_out, 0, 0, nullptr, nullptr, nullptr);
}

extern "C" int msau_box_bwd(void* stream, int dtype, const msau_box_args* a) {
    MSAU_CHECK_ARG(a && a->ii && a->params_fwd && a->params_refl && a->gout && a->ws && a->flat_grads && a->ii_g, "box_bwd: null pointer");
    // (1) box parameters: needs the forward integral image (kept) and the output gradient
    int rc = msau_box_param_grad(stream, dtype, a->ii, a->params_fwd, a->gout, a->ws, a->flat_grads, a->off_hmin, a->off_hmax, a->off_wmin,
                                 a->off_wmax, a->B, a->H, a->W, a->C, a->F, a->Cs_out, a->max_h, a->max_w);
    if (rc || !a->gin) return rc;
    // (2) input: the reflected boxes over the integral image of the output gradient, summed over the filters
    rc = msau_box_integral(stream, dtype, a->gout, a->ii_g, a->ws_ii, a->B, a->H, a->W, a->C * a->F, a->Cs_out, 0);
    if (rc) return rc;
    return msau_box_filter(stream, dtype, a->ii_g, a->params_refl, a->gin, a->B, a->H, a->W, a->C, a->F, a->Cs_in, 1, a->accumulate,
                           a->mask_a, a->add, a->mask_b);
}
