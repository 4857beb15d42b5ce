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
#include <type_traits>

#ifdef MSAU_STAMPS
__device__ unsigned long long* g_attn_stamps = nullptr;       // diagnostic build only
extern "C" int msau_debug_set_attn_stamps(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_attn_stamps), &p, sizeof(p)); }
#define ASTAMP(i) do { if (g_attn_stamps && threadIdx.x == 0) g_attn_stamps[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ASTAMP(i) do {} while (0)
#endif

namespace {

enum { M_O = 0, M_DH = 1, M_DG = 2, M_DF = 3 };
// Geometry of a sweep workgroup (round 4).  The round-3 form -- 4 groups of 16 own positions x 2 sweep halves = 512 threads, 336
// workgroups for 16 samples of 1344 positions -- left 176 CUs with one workgroup (two waves per SIMD: latency-bound, 12.5 us) and
// 80 CUs with two (18 us, the launch's duration); phase stamps, profiles/r04_attention.md.  Now OG groups of own positions per
// workgroup so that N = 1344 gives 14 workgroups per sample = 224 equal workgroups, ONE per CU, three waves per SIMD; the sweep
// side is staged CH rows at a time (one workgroup per CU: the LDS is there) -- half the barriers and half as many bytes staged.
template <int CS> struct SweepGeom {
    static constexpr int OG = 6;                          // groups of 16 own positions per workgroup
    static constexpr int SW = 2;                          // waves sharing a group: they split the swept rows (partial sums meet in LDS)
    static constexpr int CH = CS <= 64 ? 512 : 256;       // sweep rows staged in LDS per barrier phase
    static constexpr int NT = 64 * OG * SW;
    static constexpr int TS = CS * 2 + 32;                // padded row stride of the C tensor tile: 8 rows x 32 B of a tr-read hit 64 distinct banks
    template <int DS> static constexpr int lds() { return (CH * DS * 2 + 64) + CH * TS + 2 * CH * 4; }
};
constexpr float LOG2E = 1.4426950408889634f;

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
// statistics: m_i = max_j s[i][j], Z_i = sum_j exp(s[i][j] - m_i).  Wave = 16 rows i (as MFMA rows) x half of the columns.
//
// Round 4.  The round-3 form issued 4557 vector instructions per wave where ~1700 do the work: fmaxf's NaN canonicalisation
// (`v_max_f32 x, x` on both operands), a v_accvgpr_read of every score, and a v_cndmask per score for columns beyond N that
// only the last tile can have (PMC + ISA, profiles/r04_attention.md); and its 336 four-wave workgroups left a quarter of
// the CUs with twice the work of the rest.  Now: the geometry of the sweeps (6 row groups x 2 column halves = 768 threads,
// 224 equal workgroups at N = 1344), f of the sample staged in LDS once, full tiles without validity tests, v_med3 maxima,
// MFMA results in VGPRs (build flag); the two halves' (m, Z) pairs meet in LDS.
// (A version without LDS -- every lane loading its column's d-vector from global memory, one group of four tiles ahead --
// was load-latency bound at 14.5 us: a group is ~250 cycles of work, an L2 hit 500-800.)
// ---------------------------------------------------------------------------------------------
// max without fmaxf's NaN canonicalisation (`v_max_f32 x, x` on every operand): med3(a, b, huge) IS max(a, b) for numbers below huge.
// (Not inline asm: the compiler pads MFMA -> VALU read hazards only for instructions it can see; a hand-written v_max3_f32
// right behind the MFMA that produced its operand read the OLD register: NaNs at N = 1344.)
__device__ __forceinline__ float max2f(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, 3.0e38f); }      // (with +inf itself LLVM folds it back to maxnum + canonicalize)
__device__ __forceinline__ float max3f(float a, float b, float c) { return max2f(max2f(a, b), c); }

constexpr int kStatOG = 6, kStatNT = 64 * kStatOG * 2;

template <int DS>
__global__ __launch_bounds__(kStatNT) void attn_stats_mfma(const bf16_t* __restrict__ f, const bf16_t* __restrict__ g,
                                                           float* __restrict__ stats, int N) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int OG = kStatOG, NT = kStatNT, KG = DS / 8, RB = DS * 2;
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave_all % OG, half = wave_all / OG;
    const int lr = lane & 15, lg = lane >> 4;
    const int Npad = (N + 15) & ~15, T = Npad / 16;                  // column tiles
    // stage f[b] (zero rows beyond N)
    for (int idx = tid; idx < Npad * KG; idx += NT) {
        const int r = idx / KG, c = idx - r * KG;
        bf16x8 v = zero8<bf16_t>();
        if (r < N) v = load8<bf16_t>(f + ((size_t)b * N + r) * DS + c * 8);
        *reinterpret_cast<bf16x8*>(smem + r * RB + c * 16) = v;
    }
    float* comb = reinterpret_cast<float*>(smem + Npad * RB);        // [OG][16 rows][2]: the second half's (m log2e, Z)
    const int i0 = (blockIdx.x * OG + wave) * 16;
    bf16x8 afrag = zero8<bf16_t>();                                  // A: g rows, k = d (lane groups lg < KG)
    if (lg < KG && i0 + lr < N) afrag = load8<bf16_t>(g + ((size_t)b * N + i0 + lr) * DS + lg * 8);
    __syncthreads();
    // this wave's column tiles [t0, t1); a lane's fragment of tile t: row t*16 + lr, bytes (lg % KG)*16 (the lane groups beyond d
    // re-read real data: the A operand is zero there)
    const int t0 = half ? (T + 1) / 2 : 0, t1 = half ? T : (T + 1) / 2;
    const unsigned char* fcol = smem + lr * RB + (lg & (KG - 1)) * 16;
    auto score = [&](int t) {
        const bf16x8 bfrag = *reinterpret_cast<const bf16x8*>(fcol + t * 16 * RB);
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, bfrag, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    };
    const bool tail = (N & 15) != 0;                                 // the LAST tile has columns beyond N
    const int tf = (tail && t1 == T) ? t1 - 1 : t1;                  // full tiles end here
    float m[4], Z[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { m[r] = -1e30f; Z[r] = 0.f; }
    // ---- pass 1: maxima
    int t = t0;
    for (; t + 4 <= tf; t += 4) {
        f32x4 s4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) s4[u] = score(t + u);
#pragma unroll
        for (int r = 0; r < 4; ++r) m[r] = max3f(max3f(m[r], s4[0][r], s4[1][r]), s4[2][r], s4[3][r]);
    }
    for (; t < t1; ++t) {
        const f32x4 s = score(t);
        const bool valid = t * 16 + lr < N;
#pragma unroll
        for (int r = 0; r < 4; ++r) m[r] = valid ? max2f(m[r], s[r]) : m[r];
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) m[r] = max2f(m[r], __shfl_xor(m[r], o, 64));
    }
    float ml[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) ml[r] = m[r] * LOG2E;
    // ---- pass 2: Z = sum exp(s - m) over this half's columns, with this half's maximum
    t = t0;
    for (; t + 4 <= tf; t += 4) {
        f32x4 s4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) s4[u] = score(t + u);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) Z[r] += __builtin_amdgcn_exp2f(__builtin_fmaf(s4[u][r], LOG2E, -ml[r]));
    }
    for (; t < t1; ++t) {
        const f32x4 s = score(t);
        const bool valid = t * 16 + lr < N;
#pragma unroll
        for (int r = 0; r < 4; ++r) Z[r] += valid ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], LOG2E, -ml[r])) : 0.f;
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Z[r] += __shfl_xor(Z[r], o, 64);
    }
    // ---- the two halves: (m, Z) = (max, Z1 2^(ml1 - ml) + Z2 2^(ml2 - ml)); an empty half has m = -1e30, Z = 0
    if (half == 1 && lr == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { comb[(wave * 16 + lg * 4 + r) * 2] = m[r]; comb[(wave * 16 + lg * 4 + r) * 2 + 1] = Z[r]; }
    }
    __syncthreads();
    if (half == 0 && lr == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + lg * 4 + r;
            const float m2 = comb[(wave * 16 + lg * 4 + r) * 2], z2 = comb[(wave * 16 + lg * 4 + r) * 2 + 1];
            const float mm = max2f(m[r], m2);
            const float zz = Z[r] * __builtin_amdgcn_exp2f((m[r] - mm) * LOG2E) + z2 * __builtin_amdgcn_exp2f((m2 - mm) * LOG2E);
            if (i < N) { stats[((size_t)b * N + i) * 2] = mm; stats[((size_t)b * N + i) * 2 + 1] = zz; }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// the four sweep modes
//
// Round 4 (PMC: the waves of the round-3 form issued 30 % of their cycles and sat at s_waitcnt / barriers for the rest; the
// step loop was one serial chain -- fragment read, wait, MFMA, exp, read, wait, MFMA -- of ~75 vector instructions per step):
//   * softmax statistics enter as ONE number per row, lse2 = m log2(e) + log2(Z): P = exp2(s log2(e) - lse2) -- one fma and
//     one exp per score, no multiply by 1/Z;
//   * the lane groups beyond d read their row's first 16 bytes again instead of a zero slot (the own-side operand is zero
//     there): every LDS address of a step is then a per-lane base plus a compile-time offset;
//   * a full chunk's steps are unrolled phase by phase -- all score tiles, then all exponentials, then all second products --
//     so that the reads and MFMAs of one step hide behind the arithmetic of another.
// ---------------------------------------------------------------------------------------------
template <int DS, int CS, int MODE>
__device__ __forceinline__ void attn_sweep_body(const bf16_t* __restrict__ f, const bf16_t* __restrict__ g,
                                                const bf16_t* __restrict__ h, const bf16_t* __restrict__ xdy,
                                                const float* __restrict__ stats, float* __restrict__ delta,
                                                bf16_t* __restrict__ out, int N) {
    constexpr bool OWN_I = (MODE == M_DH || MODE == M_DG);        // own positions are rows of s
    constexpr bool ACC_C = (MODE == M_O || MODE == M_DH);         // second product over the C tensor
    constexpr int CTC = CS / 16, KSC = CS / 32, KG = DS / 8;
    constexpr int RB = DS * 2;                                    // bytes per d-vector row
    using G = SweepGeom<CS>;
    constexpr int SW = G::SW, OG = G::OG, CHUNK = G::CH, TS = G::TS;
    constexpr int VS_BYTES = CHUNK * RB + 64;                     // + slack for tr-reads past the last row
    constexpr int TS_BYTES = CHUNK * TS;
    constexpr int NSTEP = CHUNK / 32, KW = NSTEP / SW;            // steps per chunk; steps per wave of a full chunk
    constexpr int KU = KW >= 2 ? 2 : 1;                           // ... of which KU run side by side (4: 132-188 VGPRs, one workgroup per CU)
    static_assert(NSTEP % SW == 0 && (NSTEP / SW) % 2 == 0, "the sweep ways divide the steps of a chunk");
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* vs = smem;                                     // sweep-side d-vectors  [CHUNK][DS]
    unsigned char* ts = smem + VS_BYTES;                          // sweep-side C tensor   [CHUNK][CS] (padded)
    float* st_l = reinterpret_cast<float*>(smem + VS_BYTES + TS_BYTES);      // [CHUNK] lse2 of the sweep rows (own = j modes)
    float* st_d = st_l + CHUNK;                                              // [CHUNK] delta of the sweep rows (mode DF)

    // SW waves share a group of 16 own positions and split the swept rows between them (wave w takes the 32-row steps
    // w/4, w/4 + SW, ...).  The SW partial accumulators meet in LDS after the last chunk (fixed order).
    constexpr int NT = G::NT;
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave_all % OG, part_id = wave_all / OG;
    const int lr = lane & 15, lg = lane >> 4;
    const int o0 = (blockIdx.x * OG + wave) * 16;                 // this wave's own positions
    const int own = o0 + lr;
    const bool own_ok = own < N;
    const size_t ob = (size_t)b * N + (own_ok ? own : 0);

    const bf16_t* sweep_v = OWN_I ? f : g;                        // d-vectors of the sweep side
    const bf16_t* own_v = OWN_I ? g : f;
    const bf16_t* sweep_t = (MODE == M_O || MODE == M_DF) ? h : xdy;       // C tensor of the sweep side

    // own-side operands (B operands: lane (column own, k-group lg))
    bf16x8 vo = zero8<bf16_t>();
    if (lg < KG && own_ok) vo = load8<bf16_t>(own_v + ob * DS + lg * 8);
    bf16x8 to[ACC_C ? 1 : KSC];
    if constexpr (!ACC_C) {
        const bf16_t* own_t = (MODE == M_DG) ? h : xdy;
#pragma unroll
        for (int ks = 0; ks < KSC; ++ks) {
            to[ks] = zero8<bf16_t>();
            if (own_ok) to[ks] = load8<bf16_t>(own_t + ob * CS + ks * 32 + lg * 8);
        }
    }
    float olse = 0.f, odl = 0.f;
    if (OWN_I && own_ok) {
        olse = __builtin_fmaf(stats[ob * 2], LOG2E, __builtin_amdgcn_logf(stats[ob * 2 + 1]));
        if (MODE == M_DG) odl = delta[ob];
    }

    constexpr int NACC = ACC_C ? CTC : 1;
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int q4 = lr >> 2, p4 = lr & 3;                          // tr-read address roles inside a 16-lane group
    // per-lane LDS bases of a step's operands; a step at sweep row s0 adds s0 * (row stride), a compile-time number in the
    // unrolled path (this wave's first step of a chunk, part_id, is folded in here)
    const unsigned char* a_vs = vs + (part_id * 32 + lr) * RB + (lg & (KG - 1)) * 16;            // score A operand: row s0 + t*16 + lr
    const unsigned char* a_ts = ts + (part_id * 32 + lr) * TS + lg * 16;                          // D A operand (modes DG / DF)
    const unsigned char* a_trt = ts + (part_id * 32 + lg * 4 + q4) * TS + p4 * 8;                // tr-read base in the C tile
    const unsigned char* a_trv = vs + (part_id * 32 + lg * 4 + q4) * RB + p4 * 8;                // tr-read base in the d-vector tile
    const float* a_l = st_l + part_id * 32 + lg * 4;
    const float* a_d = st_d + part_id * 32 + lg * 4;

    // ---- sweep-side staging with a one-chunk register prefetch: the global loads of chunk k+1 are in flight while
    // chunk k is consumed
    constexpr int NV = (CHUNK * KG + NT - 1) / NT, NTT = (CHUNK * (CS / 8) + NT - 1) / NT;
    bf16x8 pv[NV], ptn[NTT];
    float pl = 0.f, pdl = 0.f;
    auto issue = [&](int r0) {
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            const int idx = tid + it * NT, r = idx / KG, c = idx % KG;
            pv[it] = zero8<bf16_t>();
            if (idx < CHUNK * KG && r0 + r < N) pv[it] = load8<bf16_t>(sweep_v + ((size_t)b * N + r0 + r) * DS + c * 8);
        }
#pragma unroll
        for (int it = 0; it < NTT; ++it) {
            const int idx = tid + it * NT, r = idx / (CS / 8), c = idx % (CS / 8);
            ptn[it] = zero8<bf16_t>();
            if (idx < CHUNK * (CS / 8) && r0 + r < N) ptn[it] = load8<bf16_t>(sweep_t + ((size_t)b * N + r0 + r) * CS + c * 8);
        }
        if (!OWN_I && tid < CHUNK) {
            pl = 3e38f; pdl = 0.f;                                // rows beyond N: P = exp2(-huge) = 0
            if (r0 + tid < N) {
                const size_t qq = (size_t)b * N + r0 + tid;
                pl = __builtin_fmaf(stats[qq * 2], LOG2E, __builtin_amdgcn_logf(stats[qq * 2 + 1]));
                if (MODE == M_DF) pdl = delta[qq];
            }
        }
    };
    ASTAMP(0);
    issue(0);

    // one 32-row step, phase by phase (K steps side by side): row offset of step k = k * SW * 32 (+ `extra` rows, run time)
    auto run_steps = [&](auto KC, int extra) {
        constexpr int K = decltype(KC)::value;
        const unsigned char* pvs = a_vs + extra * RB;
        const unsigned char* pts = a_ts + extra * TS;
        const unsigned char* ptt = a_trt + extra * TS;
        const unsigned char* ptv = a_trv + extra * RB;
        const float* plse = a_l + extra;
        const float* pdel = a_d + extra;
        f32x4 S[K][2];
        // ---- score tiles: rows = sweep positions, columns = own positions
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(pvs + (k * SW * 32 + t * 16) * RB);
                S[k][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, vo, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            }
        // ---- d(beta)[sweep][own] = sum_c T_sweep[c] * T_own[c]  (modes DG / DF)
        f32x4 D[ACC_C ? 1 : K][2];
        if constexpr (!ACC_C) {
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    D[k][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < KSC; ++ks) {
                        const bf16x8 a = *reinterpret_cast<const bf16x8*>(pts + (k * SW * 32 + t * 16) * TS + ks * 64);
                        D[k][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, to[ks], D[k][t], 0, 0, 0);
                    }
                }
        }
        // ---- P = exp(s - m) / Z = exp2(s log2e - lse2)   (lane: rows 4*lg + r of each tile, column lr)
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 ll;
                if constexpr (OWN_I) ll = f32x4{olse, olse, olse, olse};
                else ll = *reinterpret_cast<const f32x4*>(plse + k * SW * 32 + t * 16);
#pragma unroll
                for (int r = 0; r < 4; ++r) S[k][t][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(S[k][t][r], LOG2E, -ll[r]));
                if constexpr (!ACC_C) {
                    f32x4 dl;
                    if constexpr (OWN_I) dl = f32x4{odl, odl, odl, odl};
                    else dl = *reinterpret_cast<const f32x4*>(pdel + k * SW * 32 + t * 16);
#pragma unroll
                    for (int r = 0; r < 4; ++r) S[k][t][r] *= (D[k][t][r] - dl[r]);
                }
            }
        // ---- second product: the two tiles of a step are the 8 k-values of its B operand
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const bf16x8 bop = pack8(S[k][0], S[k][1]);
            if constexpr (ACC_C) {
#pragma unroll
                for (int ct = 0; ct < CTC; ++ct) {
                    const unsigned char* base = ptt + (k * SW * 32) * TS + ct * 32;
                    const bf16x8 a = tr_pair(base, base + 16 * TS);
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bop, acc[ct], 0, 0, 0);
                }
            } else {
                const unsigned char* base = ptv + (k * SW * 32) * RB;
                const bf16x8 a = tr_pair(base, base + 16 * RB);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bop, acc[0], 0, 0, 0);
            }
        }
    };

    for (int r0 = 0; r0 < N; r0 += CHUNK) {
        __syncthreads();                                          // the previous chunk has been consumed
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            const int idx = tid + it * NT, r = idx / KG, c = idx % KG;
            if (idx < CHUNK * KG) *reinterpret_cast<bf16x8*>(vs + r * RB + c * 16) = pv[it];
        }
#pragma unroll
        for (int it = 0; it < NTT; ++it) {
            const int idx = tid + it * NT, r = idx / (CS / 8), c = idx % (CS / 8);
            if (idx < CHUNK * (CS / 8)) *reinterpret_cast<bf16x8*>(ts + r * TS + c * 16) = ptn[it];
        }
        if (!OWN_I && tid < CHUNK) { st_l[tid] = pl; st_d[tid] = pdl; }
        __syncthreads();
        if (r0 < 6 * CHUNK) ASTAMP(1 + 2 * (r0 / CHUNK));
        if (r0 + CHUNK < N) issue(r0 + CHUNK);

        const int nstep = min(CHUNK, ((N - r0 + 31) / 32) * 32) / 32;
        if (nstep == NSTEP) {
#pragma unroll
            for (int u = 0; u < KW / KU; ++u) run_steps(std::integral_constant<int, KU>{}, u * KU * SW * 32);
        } else {
            for (int st = part_id; st < nstep; st += SW) run_steps(std::integral_constant<int, 1>{}, (st - part_id) * 32);
        }
        if (r0 < 6 * CHUNK) { asm volatile("" :: "v"(acc[0][0])); ASTAMP(2 + 2 * (r0 / CHUNK)); }
    }

    // ---- the SW partial sums of a group of own positions: parts 1.. through LDS, part 0 adds them in order and finishes
    if constexpr (SW > 1) {
        __syncthreads();                                          // the last chunk has been consumed
        static_assert((SW - 1) * OG * NACC * 1024 <= TS_BYTES, "the partial sums are parked in the C-tensor tile");
        f32x4* red = reinterpret_cast<f32x4*>(ts);                // [part - 1][wave][tile][lane]
        if (part_id > 0) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) red[(((part_id - 1) * OG + wave) * NACC + i) * 64 + lane] = acc[i];
        }
        __syncthreads();
        if (part_id > 0) return;
#pragma unroll
        for (int p = 1; p < SW; ++p)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] += red[(((p - 1) * OG + wave) * NACC + i) * 64 + lane];
    }
    ASTAMP(13);
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
__global__ __launch_bounds__(SweepGeom<CS>::NT, CS <= 64 ? SweepGeom<CS>::NT / 256 : 1) void attn_sweep_mfma(const bf16_t* __restrict__ f, const bf16_t* __restrict__ g,
                                                       const bf16_t* __restrict__ h, const bf16_t* __restrict__ xdy,
                                                       const float* __restrict__ stats, float* __restrict__ delta,
                                                       bf16_t* __restrict__ out, int N) {
    attn_sweep_body<DS, CS, MODE>(f, g, h, xdy, stats, delta, out, N);
}

// dg and df need the same inputs (delta from the DH sweep) and nothing from each other: one launch, blockIdx.z picks
// the mode, twice the workgroups in flight (the sweeps are latency-bound at 336 workgroups).
template <int DS, int CS>
__global__ __launch_bounds__(SweepGeom<CS>::NT, CS <= 64 ? SweepGeom<CS>::NT / 256 : 1) void attn_sweep_dgdf(const bf16_t* __restrict__ f, const bf16_t* __restrict__ g,
                                                       const bf16_t* __restrict__ h, const bf16_t* __restrict__ dy,
                                                       const float* __restrict__ stats, float* __restrict__ delta,
                                                       bf16_t* __restrict__ dg, bf16_t* __restrict__ df, int N) {
    if (blockIdx.z == 0) attn_sweep_body<DS, CS, M_DG>(f, g, h, dy, stats, delta, dg, N);
    else attn_sweep_body<DS, CS, M_DF>(f, g, h, dy, stats, delta, df, N);
}

template <int DS, int CS, int MODE>
int launch_sweep(hipStream_t s, const bf16_t* f, const bf16_t* g, const bf16_t* h, const bf16_t* xdy, const float* stats,
                 float* delta, bf16_t* out, int B, int N) {
    using G = SweepGeom<CS>;
    constexpr int lds = G::template lds<DS>();
    static bool attr_set = false;
    if (!attr_set && lds > 60 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_sweep_mfma<DS, CS, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, MSAU_LDS_LIMIT);
        if (e != hipSuccess) return msau_set_error(MSAU_ERR_HIP, "attn: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL((attn_sweep_mfma<DS, CS, MODE>), dim3(cdiv(N, G::OG * 16), B), dim3(G::NT), lds, s, f, g, h, xdy, stats, delta, out, N);
    MSAU_CHECK_LAUNCH("attn_sweep_mfma");
    return 0;
}

template <int DS, int CS>
int fwd_t(hipStream_t s, const void* f, const void* g, const void* h, const void* x, void* y, float* stats, int B, int N) {
    const bf16_t* fp = static_cast<const bf16_t*>(f); const bf16_t* gp = static_cast<const bf16_t*>(g);
    const int Npad = (N + 15) & ~15;
    const size_t lds = (size_t)Npad * DS * 2 + kStatOG * 16 * 2 * 4;
    if (lds > 150 * 1024) return msau_set_error(MSAU_ERR_LDS, "selfattn: N=%d too large for the statistics kernel", N);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_stats_mfma<DS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, MSAU_LDS_LIMIT);
        if (e != hipSuccess) return msau_set_error(MSAU_ERR_HIP, "attn: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL((attn_stats_mfma<DS>), dim3(cdiv(N, kStatOG * 16), B), dim3(kStatNT), lds, s, fp, gp, stats, N);
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
    using G = SweepGeom<CS>;
    constexpr int lds = G::template lds<DS>();
    static bool attr_set = false;
    if (!attr_set && lds > 60 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_sweep_dgdf<DS, CS>), hipFuncAttributeMaxDynamicSharedMemorySize, MSAU_LDS_LIMIT);
        if (e != hipSuccess) return msau_set_error(MSAU_ERR_HIP, "attn: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL((attn_sweep_dgdf<DS, CS>), dim3(cdiv(N, G::OG * 16), B, 2), dim3(G::NT), lds, s, fp, gp, hp, dyp, stats, ws,
                       static_cast<bf16_t*>(dg), static_cast<bf16_t*>(df), N);
    MSAU_CHECK_LAUNCH("attn_sweep_dgdf");
    return 0;
}

}  // namespace

// returns 1 if the (Ds, Cs) pair has an MFMA instance (bf16 only), else 0
// (the statistics kernel keeps f of one whole sample in LDS: larger images take the VALU kernels of attention.hip)
int msau_attn_mfma_supported(int Ds, int Cs, int N) {
    if (!((Ds == 8 && (Cs == 32 || Cs == 64)) || (Ds == 16 && Cs == 128))) return 0;
    return (size_t)((N + 15) & ~15) * Ds * 2 + kStatOG * 16 * 2 * 4 <= 150 * 1024;
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
