// Bottleneck self-attention core on the matrix cores (bf16 storage), flash-style: no N x N matrix.
//   s[i][j] = g_i . f_j ;  P[i][j] = exp(s[i][j] - m_i) / Z_i ;  y[j] = x[j] + sum_i h[i] P[i][j]
// (model/layers/attention.py:156-162 -- rows of s are normalised, but the output sums over ROWS.)
//
// One kernel template, four modes.  A wave owns 16 "own" positions as MFMA COLUMNS and sweeps the
// other side in steps of 32 rows; a 16x16 score tile (MFMA, K = d padded to 32) leaves each lane with
// 4 consecutive rows of its own column, so two tiles are exactly the 8 k-values (rows
// 4q..4q+3 and 16+4q..16+4q+3) of the B operand of the next MFMA -- the accumulator tile feeds the
// second product with no LDS round trip; the A operand of that product (h^T, dy^T, f^T, g^T) is read
// from the pixel-major LDS tile with ds_read_b64_tr_b16 in the same permuted k order.
//
//   mode O  (own j, sweep i): acc[c][j] += h^T[c][i]  P[i][j]                 -> y = x + acc
//   mode DH (own i, sweep j): acc[c][i] += dy^T[c][j] P[i][j]                 -> dh ; delta_i = h_i . dh_i
//   mode DG (own i, sweep j): ds = P (dy_j . h_i - delta_i) ; acc[d][i] += f^T[d][j] ds   -> dg
//   mode DF (own j, sweep i): ds = P (h_i . dy_j - delta_i) ; acc[d][j] += g^T[d][i] ds   -> df
// plus a statistics kernel (row max m_i and row sum Z_i of exp).
#include "msau_common.h"

namespace {

enum { M_O = 0, M_DH = 1, M_DG = 2, M_DF = 3 };
constexpr int CHUNK = 256;                // sweep rows staged in LDS per barrier phase (128: -0.3 %)
constexpr float LOG2E = 1.4426950408889634f;
constexpr int kSweepWays = 2;             // waves sharing one group of own positions in the sweeps (see attn_sweep_body)

typedef __attribute__((address_space(3))) bf16x4* lds_v4;

__device__ __forceinline__ bf16x8 tr_pair(const unsigned char* lo, const unsigned char* hi) {
    bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)lo);
    bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)hi);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r[i] = (bf16_t)a[i]; r[4 + i] = (bf16_t)b[i]; }
    return r;
}

// ---------------------------------------------------------------------------------------------
// statistics: m_i = max_j s[i][j], Z_i = sum_j exp(s[i][j] - m_i).  Wave = 16 rows i (as MFMA rows),
// f of the whole sample in LDS.
// ---------------------------------------------------------------------------------------------
template <int DS>
__global__ __launch_bounds__(256) void attn_stats_mfma(const bf16_t* __restrict__ f, const bf16_t* __restrict__ g,
                                                       float* __restrict__ stats, int N) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int Npad = (N + 15) & ~15;
    constexpr int RB = DS * 2;                                   // bytes per f / g row
    // stage f[b] (zero rows beyond N) + one zero slot
    for (int r = tid; r < Npad; r += 256) {
#pragma unroll
        for (int c = 0; c < DS / 8; ++c) {
            bf16x8 v = zero8<bf16_t>();
            if (r < N) v = load8<bf16_t>(f + ((size_t)b * N + r) * DS + c * 8);
            *reinterpret_cast<bf16x8*>(smem + r * RB + c * 16) = v;
        }
    }
    const int zoff = Npad * RB;
    if (tid < 2) *reinterpret_cast<bf16x8*>(smem + zoff + tid * 16) = zero8<bf16_t>();
    __syncthreads();
    const int i0 = (blockIdx.x * 4 + wave) * 16;
    if (i0 >= N) return;
    bf16x8 afrag = zero8<bf16_t>();                               // A: G rows, k = d (lanes g < DS/8)
    if (lg < DS / 8 && i0 + lr < N) afrag = load8<bf16_t>(g + ((size_t)b * N + i0 + lr) * DS + lg * 8);
    // Two passes over the columns instead of one online-softmax sweep: the running (m, Z) update is a serial chain of two
    // exponentials per score tile, and with one or two waves per SIMD nothing hides it (84 tiles x ~450 cycles).  Pass 1
    // takes the row maximum (MFMA + max only), pass 2 sums exp(s - m) with the final m: twice the (cheap) MFMAs, half the
    // exponentials, and the tiles of a pass are independent -- four in flight.
    float m[4], Z[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { m[r] = -1e30f; Z[r] = 0.f; }
    const int boff = lg < DS / 8 ? lg * 16 : -1;
    auto score = [&](int j0) {
        bf16x8 bfrag = *reinterpret_cast<const bf16x8*>(smem + (boff < 0 ? zoff : (j0 + lr) * RB + boff));
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, bfrag, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    };
    const int Nq = Npad & ~63;                                     // whole groups of four 16-column tiles
    for (int j0 = 0; j0 < Nq; j0 += 64) {
        f32x4 s4[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) s4[t] = score(j0 + t * 16);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bool valid = j0 + t * 16 + lr < N;
#pragma unroll
            for (int r = 0; r < 4; ++r) m[r] = valid ? fmaxf(m[r], s4[t][r]) : m[r];
        }
    }
    for (int j0 = Nq; j0 < Npad; j0 += 16) {
        const f32x4 s = score(j0);
        const bool valid = j0 + lr < N;
#pragma unroll
        for (int r = 0; r < 4; ++r) m[r] = valid ? fmaxf(m[r], s[r]) : m[r];
    }
    // the row maximum over all columns: merge the 16 column-lanes of each row group
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) m[r] = fmaxf(m[r], __shfl_xor(m[r], o, 64));
    }
    float ml[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) ml[r] = m[r] * LOG2E;
    for (int j0 = 0; j0 < Nq; j0 += 64) {
        f32x4 s4[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) s4[t] = score(j0 + t * 16);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bool valid = j0 + t * 16 + lr < N;
#pragma unroll
            for (int r = 0; r < 4; ++r) Z[r] += valid ? __builtin_amdgcn_exp2f(s4[t][r] * LOG2E - ml[r]) : 0.f;
        }
    }
    for (int j0 = Nq; j0 < Npad; j0 += 16) {
        const f32x4 s = score(j0);
        const bool valid = j0 + lr < N;
#pragma unroll
        for (int r = 0; r < 4; ++r) Z[r] += valid ? __builtin_amdgcn_exp2f(s[r] * LOG2E - ml[r]) : 0.f;
    }
    // sum over the 16 column-lanes
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Z[r] += __shfl_xor(Z[r], o, 64);
    }
    if (lr == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int i = i0 + lg * 4 + r;
            if (i < N) { stats[((size_t)b * N + i) * 2] = m[r]; stats[((size_t)b * N + i) * 2 + 1] = Z[r]; }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// the four sweep modes
// ---------------------------------------------------------------------------------------------
template <int DS, int CS, int MODE, int SW>
__device__ __forceinline__ void attn_sweep_body(const bf16_t* __restrict__ f, const bf16_t* __restrict__ g,
                                                const bf16_t* __restrict__ h, const bf16_t* __restrict__ xdy,
                                                const float* __restrict__ stats, float* __restrict__ delta,
                                                bf16_t* __restrict__ out, int N) {
    constexpr bool OWN_I = (MODE == M_DH || MODE == M_DG);        // own positions are rows of s
    constexpr bool ACC_C = (MODE == M_O || MODE == M_DH);         // second product over the C tensor
    constexpr int CTC = CS / 16, KSC = CS / 32;
    constexpr int RB = DS * 2;                                    // bytes per d-vector row
    constexpr int TS = CS * 2 + 16;                               // padded row stride of the C tensor tile
    constexpr int VS_BYTES = CHUNK * RB + 64;                     // + slack for tr-reads past the last row / zero slot
    constexpr int TS_BYTES = CHUNK * TS;
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* vs = smem;                                     // sweep-side d-vectors  [CHUNK][DS]
    unsigned char* ts = smem + VS_BYTES;                          // sweep-side C tensor   [CHUNK][CS] (padded)
    float* st_m = reinterpret_cast<float*>(smem + VS_BYTES + TS_BYTES);      // [CHUNK] (own = j modes)
    float* st_z = st_m + CHUNK;
    float* st_d = st_z + CHUNK;

    // SW waves share a group of 16 own positions and split the swept rows between them (wave w takes the 32-row steps
    // w/4, w/4 + SW, ...): the sweep is one dependent chain per wave (fragment read -> MFMA -> exp -> MFMA), and with 336
    // workgroups of four waves a SIMD held one or two of them -- nothing to overlap the chain with.  The SW partial
    // accumulators meet in LDS after the last chunk (fixed order).
    constexpr int NT = 256 * SW;
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave_all = tid >> 6;
    const int wave = wave_all & 3, part_id = wave_all >> 2;
    const int lr = lane & 15, lg = lane >> 4;
    const int o0 = (blockIdx.x * 4 + wave) * 16;                  // this wave's own positions
    const int own = o0 + lr;
    const bool own_ok = own < N;
    const size_t ob = (size_t)b * N + (own_ok ? own : 0);

    const bf16_t* sweep_v = OWN_I ? f : g;                        // d-vectors of the sweep side
    const bf16_t* own_v = OWN_I ? g : f;
    const bf16_t* sweep_t = (MODE == M_O || MODE == M_DF) ? h : xdy;       // C tensor of the sweep side

    // own-side operands (B operands: lane (column own, k-group lg))
    bf16x8 vo = zero8<bf16_t>();
    if (lg < DS / 8 && own_ok) vo = load8<bf16_t>(own_v + ob * DS + lg * 8);
    bf16x8 to[ACC_C ? 1 : KSC];
    if constexpr (!ACC_C) {
        const bf16_t* own_t = (MODE == M_DG) ? h : xdy;
#pragma unroll
        for (int ks = 0; ks < KSC; ++ks) {
            to[ks] = zero8<bf16_t>();
            if (own_ok) to[ks] = load8<bf16_t>(own_t + ob * CS + ks * 32 + lg * 8);
        }
    }
    float om = 0.f, oiz = 0.f, odl = 0.f;
    if (OWN_I && own_ok) {
        om = stats[ob * 2] * LOG2E;
        oiz = 1.f / stats[ob * 2 + 1];
        if (MODE == M_DG) odl = delta[ob];
    }

    constexpr int NACC = ACC_C ? CTC : 1;
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int q4 = lr >> 2, p4 = lr & 3;                          // tr-read address roles inside a 16-lane group
    const int zoff = CHUNK * RB + 32;                             // zero slot inside vs
    if (tid < 2) *reinterpret_cast<bf16x8*>(vs + CHUNK * RB + tid * 16 + 16) = zero8<bf16_t>();

    // ---- sweep-side staging with a one-chunk register prefetch: the global loads of chunk k+1 are in flight while
    // chunk k is consumed (the sweeps are latency-bound: 11 chunks, each one load round trip otherwise)
    constexpr int NV = (CHUNK * (DS / 8) + NT - 1) / NT, NTT = (CHUNK * (CS / 8) + NT - 1) / NT;
    bf16x8 pv[NV], ptn[NTT];
    float pm = 0.f, piz = 0.f, pdl = 0.f;
    auto issue = [&](int r0) {
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            const int idx = tid + it * NT, r = idx / (DS / 8), c = idx % (DS / 8);
            pv[it] = zero8<bf16_t>();
            if (idx < CHUNK * (DS / 8) && r0 + r < N) pv[it] = load8<bf16_t>(sweep_v + ((size_t)b * N + r0 + r) * DS + c * 8);
        }
#pragma unroll
        for (int it = 0; it < NTT; ++it) {
            const int idx = tid + it * NT, r = idx / (CS / 8), c = idx % (CS / 8);
            ptn[it] = zero8<bf16_t>();
            if (idx < CHUNK * (CS / 8) && r0 + r < N) ptn[it] = load8<bf16_t>(sweep_t + ((size_t)b * N + r0 + r) * CS + c * 8);
        }
        if (!OWN_I && tid < CHUNK) {
            pm = 0.f; piz = 0.f; pdl = 0.f;
            if (r0 + tid < N) {
                const size_t qq = (size_t)b * N + r0 + tid;
                pm = stats[qq * 2] * LOG2E; piz = 1.f / stats[qq * 2 + 1];
                if (MODE == M_DF) pdl = delta[qq];
            }
        }
    };
    issue(0);

    for (int r0 = 0; r0 < N; r0 += CHUNK) {
        __syncthreads();                                          // the previous chunk has been consumed
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            const int idx = tid + it * NT, r = idx / (DS / 8), c = idx % (DS / 8);
            if (idx < CHUNK * (DS / 8)) *reinterpret_cast<bf16x8*>(vs + r * RB + c * 16) = pv[it];
        }
#pragma unroll
        for (int it = 0; it < NTT; ++it) {
            const int idx = tid + it * NT, r = idx / (CS / 8), c = idx % (CS / 8);
            if (idx < CHUNK * (CS / 8)) *reinterpret_cast<bf16x8*>(ts + r * TS + c * 16) = ptn[it];
        }
        if (!OWN_I && tid < CHUNK) { st_m[tid] = pm; st_z[tid] = piz; st_d[tid] = pdl; }
        __syncthreads();
        if (r0 + CHUNK < N) issue(r0 + CHUNK);

        const int nstep = min(CHUNK, ((N - r0 + 31) / 32) * 32) / 32;
        for (int st = part_id; st < nstep; st += SW) {
            const int s0 = st * 32;
            // ---- two 16x16 score tiles: rows = sweep positions, columns = own positions
            f32x4 S[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int row = s0 + t * 16 + lr;
                bf16x8 a = *reinterpret_cast<const bf16x8*>(vs + (lg < DS / 8 ? row * RB + lg * 16 : zoff));
                S[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, vo, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            }
            // ---- P = exp(s - m) / Z  (lane: rows 4*lg + r of each tile, column lr)
            f32x4 P[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 mm, zz;
                if constexpr (OWN_I) { mm = f32x4{om, om, om, om}; zz = f32x4{oiz, oiz, oiz, oiz}; }
                else {
                    mm = *reinterpret_cast<const f32x4*>(st_m + s0 + t * 16 + lg * 4);
                    zz = *reinterpret_cast<const f32x4*>(st_z + s0 + t * 16 + lg * 4);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) P[t][r] = __builtin_amdgcn_exp2f(S[t][r] * LOG2E - mm[r]) * zz[r];
            }
            if constexpr (ACC_C) {
                const bf16x8 bop = pack8(P[0], P[1]);
                const unsigned char* base = ts + (s0 + lg * 4 + q4) * TS + p4 * 8;
#pragma unroll
                for (int ct = 0; ct < CTC; ++ct) {
                    bf16x8 a = tr_pair(base + ct * 32, base + 16 * TS + ct * 32);
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bop, acc[ct], 0, 0, 0);
                }
            } else {
                // d(beta)[sweep][own] = sum_c T_sweep[c] * T_own[c]
                f32x4 D[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    D[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                    const unsigned char* rowp = ts + (s0 + t * 16 + lr) * TS + lg * 16;
#pragma unroll
                    for (int ks = 0; ks < KSC; ++ks) {
                        bf16x8 a = *reinterpret_cast<const bf16x8*>(rowp + ks * 64);
                        D[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, to[ks], D[t], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x4 dl;
                    if constexpr (OWN_I) dl = f32x4{odl, odl, odl, odl};
                    else dl = *reinterpret_cast<const f32x4*>(st_d + s0 + t * 16 + lg * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) P[t][r] *= (D[t][r] - dl[r]);
                }
                const bf16x8 bop = pack8(P[0], P[1]);
                const unsigned char* base = vs + (s0 + lg * 4 + q4) * RB + p4 * 8;
                bf16x8 a = tr_pair(base, base + 16 * RB);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bop, acc[0], 0, 0, 0);
            }
        }
    }

    // ---- the SW partial sums of a group of own positions: parts 1.. through LDS, part 0 adds them in order and finishes
    if constexpr (SW > 1) {
        __syncthreads();                                          // the last chunk has been consumed
        static_assert((SW - 1) * 4 * NACC * 1024 <= TS_BYTES, "the partial sums are parked in the C-tensor tile");
        f32x4* red = reinterpret_cast<f32x4*>(ts);                // [part - 1][wave][tile][lane]
        if (part_id > 0) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) red[(((part_id - 1) * 4 + wave) * NACC + i) * 64 + lane] = acc[i];
        }
        __syncthreads();
        if (part_id > 0) return;
#pragma unroll
        for (int p = 1; p < SW; ++p)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] += red[(((p - 1) * 4 + wave) * NACC + i) * 64 + lane];
    }
    // ---- epilogue: lane (column own, q = lg) holds rows 4q + r of every accumulator tile
    if constexpr (ACC_C) {
        float part = 0.f;
#pragma unroll
        for (int ct = 0; ct < CTC; ++ct) {
            const int c = ct * 16 + lg * 4;
            f32x4 v = acc[ct];
            if (own_ok) {
                if (MODE == M_O) {
                    bf16x4 xv = load4<bf16_t>(xdy + ob * CS + c);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += (float)xv[r];
                } else {
                    bf16x4 hv = load4<bf16_t>(h + ob * CS + c);
#pragma unroll
                    for (int r = 0; r < 4; ++r) part += (float)hv[r] * v[r];
                }
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (bf16_t)v[r];
                store4<bf16_t>(out + ob * CS + c, o);
            }
        }
        if (MODE == M_DH) {
            part += __shfl_xor(part, 16, 64);
            part += __shfl_xor(part, 32, 64);
            if (lg == 0 && own_ok) delta[ob] = part;
        }
    } else {
        const int dd = lg * 4;
        if (own_ok && dd < DS) {
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16_t)acc[0][r];
            store4<bf16_t>(out + ob * DS + dd, o);
        }
    }
}

template <int DS, int CS, int MODE>
__global__ __launch_bounds__(256 * kSweepWays) void attn_sweep_mfma(const bf16_t* __restrict__ f, const bf16_t* __restrict__ g,
                                                       const bf16_t* __restrict__ h, const bf16_t* __restrict__ xdy,
                                                       const float* __restrict__ stats, float* __restrict__ delta,
                                                       bf16_t* __restrict__ out, int N) {
    attn_sweep_body<DS, CS, MODE, kSweepWays>(f, g, h, xdy, stats, delta, out, N);
}

// dg and df need the same inputs (delta from the DH sweep) and nothing from each other: one launch, blockIdx.z picks
// the mode, twice the workgroups in flight (the sweeps are latency-bound at 336 workgroups).
template <int DS, int CS>
__global__ __launch_bounds__(256 * kSweepWays) void attn_sweep_dgdf(const bf16_t* __restrict__ f, const bf16_t* __restrict__ g,
                                                       const bf16_t* __restrict__ h, const bf16_t* __restrict__ dy,
                                                       const float* __restrict__ stats, float* __restrict__ delta,
                                                       bf16_t* __restrict__ dg, bf16_t* __restrict__ df, int N) {
    if (blockIdx.z == 0) attn_sweep_body<DS, CS, M_DG, kSweepWays>(f, g, h, dy, stats, delta, dg, N);
    else attn_sweep_body<DS, CS, M_DF, kSweepWays>(f, g, h, dy, stats, delta, df, N);
}

template <int DS, int CS, int MODE>
int launch_sweep(hipStream_t s, const bf16_t* f, const bf16_t* g, const bf16_t* h, const bf16_t* xdy, const float* stats,
                 float* delta, bf16_t* out, int B, int N) {
    constexpr int lds = (CHUNK * DS * 2 + 64) + CHUNK * (CS * 2 + 16) + 3 * CHUNK * 4;
    hipLaunchKernelGGL((attn_sweep_mfma<DS, CS, MODE>), dim3(cdiv(N, 64), B), dim3(256 * kSweepWays), lds, s, f, g, h, xdy, stats, delta, out, N);
    MSAU_CHECK_LAUNCH("attn_sweep_mfma");
    return 0;
}

template <int DS, int CS>
int fwd_t(hipStream_t s, const void* f, const void* g, const void* h, const void* x, void* y, float* stats, int B, int N) {
    const bf16_t* fp = static_cast<const bf16_t*>(f); const bf16_t* gp = static_cast<const bf16_t*>(g);
    const int Npad = (N + 15) & ~15;
    const size_t lds = (size_t)Npad * DS * 2 + 64;
    if (lds > 150 * 1024) return msau_set_error(MSAU_ERR_LDS, "selfattn: N=%d too large for the statistics kernel", N);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_stats_mfma<DS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, MSAU_LDS_LIMIT);
        if (e != hipSuccess) return msau_set_error(MSAU_ERR_HIP, "attn: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL((attn_stats_mfma<DS>), dim3(cdiv(N, 64), B), dim3(256), lds, s, fp, gp, stats, N);
    MSAU_CHECK_LAUNCH("attn_stats_mfma");
    return launch_sweep<DS, CS, M_O>(s, fp, gp, static_cast<const bf16_t*>(h), static_cast<const bf16_t*>(x), stats, nullptr,
                                     static_cast<bf16_t*>(y), B, N);
}

template <int DS, int CS>
int bwd_t(hipStream_t s, const void* f, const void* g, const void* h, const void* dy, const float* stats, void* df, void* dg,
          void* dh, float* ws, int B, int N) {
    const bf16_t* fp = static_cast<const bf16_t*>(f); const bf16_t* gp = static_cast<const bf16_t*>(g);
    const bf16_t* hp = static_cast<const bf16_t*>(h); const bf16_t* dyp = static_cast<const bf16_t*>(dy);
    int rc = launch_sweep<DS, CS, M_DH>(s, fp, gp, hp, dyp, stats, ws, static_cast<bf16_t*>(dh), B, N);
    if (rc) return rc;
    constexpr int lds = (CHUNK * DS * 2 + 64) + CHUNK * (CS * 2 + 16) + 3 * CHUNK * 4;
    hipLaunchKernelGGL((attn_sweep_dgdf<DS, CS>), dim3(cdiv(N, 64), B, 2), dim3(256 * kSweepWays), lds, s, fp, gp, hp, dyp, stats, ws,
                       static_cast<bf16_t*>(dg), static_cast<bf16_t*>(df), N);
    MSAU_CHECK_LAUNCH("attn_sweep_dgdf");
    return 0;
}

}  // namespace

// returns 1 if the (Ds, Cs) pair has an MFMA instance (bf16 only), else 0
// (the statistics kernel keeps f of one whole sample in LDS: larger images take the VALU kernels of attention.hip)
int msau_attn_mfma_supported(int Ds, int Cs, int N) {
    if (!((Ds == 8 && (Cs == 32 || Cs == 64)) || (Ds == 16 && Cs == 128))) return 0;
    return (size_t)((N + 15) & ~15) * Ds * 2 + 64 <= 150 * 1024;
}

int msau_attn_mfma_fwd(hipStream_t s, const void* f, const void* g, const void* h, const void* x, void* y, float* stats,
                       int B, int N, int Ds, int Cs) {
    if (Ds == 8 && Cs == 32) return fwd_t<8, 32>(s, f, g, h, x, y, stats, B, N);
    if (Ds == 8 && Cs == 64) return fwd_t<8, 64>(s, f, g, h, x, y, stats, B, N);
    if (Ds == 16 && Cs == 128) return fwd_t<16, 128>(s, f, g, h, x, y, stats, B, N);
    return msau_set_error(MSAU_ERR_ARG, "selfattn mfma: unsupported (Ds,Cs)=(%d,%d)", Ds, Cs);
}

int msau_attn_mfma_bwd(hipStream_t s, const void* f, const void* g, const void* h, const void* dy, const float* stats,
                       void* df, void* dg, void* dh, float* ws, int B, int N, int Ds, int Cs) {
    if (Ds == 8 && Cs == 32) return bwd_t<8, 32>(s, f, g, h, dy, stats, df, dg, dh, ws, B, N);
    if (Ds == 8 && Cs == 64) return bwd_t<8, 64>(s, f, g, h, dy, stats, df, dg, dh, ws, B, N);
    if (Ds == 16 && Cs == 128) return bwd_t<16, 128>(s, f, g, h, dy, stats, df, dg, dh, ws, B, N);
    return msau_set_error(MSAU_ERR_ARG, "selfattn mfma: unsupported (Ds,Cs)=(%d,%d)", Ds, Cs);
}
