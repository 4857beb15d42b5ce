// Lean instances of the weight-gradient kernel (see conv_wgrad.hip for the algorithm): stride 1, no
// dilation, k in {1,3}, channel counts per chunk known at compile time, so every divisor, LDS stride
// and tap offset is a constant and the column-group offsets live in registers instead of an LDS table.
// Same slab layout as wgrad_kernel: slabs[workgroup][chunk][Cout][kext], k = [tap][channel] + ones column.
#include "msau_common.h"
#include <cstdlib>

namespace {

struct WLeanArgs {
    msau_wgrad_desc d;
    int kextc, nchunks;
    int tiles_x, tiles_y, ntiles;
    unsigned mag_tx, mag_ty;                     // ceil(2^32 / tiles_x), ceil(2^32 / tiles_y)
};

// up to kWGroup launches of one instance (same geometry, different tensors) share a grid: blockIdx.z picks the layer.  The
// level-2/3 weight gradients have 64-107 workgroups each; two or three of them side by side fill the device instead
// of queueing behind each other on the side stream.
constexpr int kWGroup = 4;
struct WLeanMulti { WLeanArgs a[kWGroup]; };

#define WL_ABS 0x40000000
#ifndef MSAU_WPS_MAX
#define MSAU_WPS_MAX 10
#endif

typedef __attribute__((address_space(3))) bf16x4* lds_v4;

template <typename T, int C8, int CO8, int KS, int DIL = 1, int STRIDE = 1>
struct WLeanCfg {
    static constexpr int ESZ = (int)sizeof(T);
    static constexpr int TI = 15 * STRIDE + (KS - 1) * DIL + 1;
    // Fragment reads (bf16): ds_read_b64_tr_b16, 8 bytes per lane, served in the two 32-lane halves; a half is conflict-free when
    // its 32 lanes cover the 256-byte bank row once.  A half is lane groups lg = 0, 1 (or 2, 3): 4 pixels (q) x 4 channel quads (p)
    // each, i.e. 8 pixels x 32 contiguous bytes.  NMAP: the 8 pixels are 8 CONSECUTIVE columns of one tile row (k index j of lane
    // group lg <-> column (lg & 1) * 4 + j for j < 4, 8 + (lg & 1) * 4 + j - 4 above) and the pixel stride is 32 (mod 64) bytes, so
    // the eight 32-byte pieces tile the bank row.  Rounds 1-3 took columns (lg & 1) * 8 + j at an odd number of 16-byte slots:
    // pieces at odd 16-byte offsets overlap, a 2-way conflict on every read (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.45 on the
    // 16- to 64-channel instances, profiles/r04_pmc_step.txt).  The 8-channel instances (16-byte pixels, two taps per k-tile) keep
    // the old mapping: they measured 0-0.14.  The sum over pixels is the same set in another order.
    static constexpr bool NMAP = ESZ == 2 && C8 >= 2;
    static constexpr int QS = NMAP ? 4 : 8;                   // column step between lane groups lg & 1
    static constexpr int HI = NMAP ? 8 : 4;                   // column step to the second half of a lane's k indices
    static constexpr int PSX = NMAP ? lds_pixel_stride(C8 * 8 * ESZ, ESZ, C8, STRIDE) : (((C8 * 8 * ESZ / 16) % 2 == 0) ? C8 * 8 * ESZ + 16 : C8 * 8 * ESZ);
    static constexpr int PSG = NMAP ? lds_pixel_stride(CO8 * 8 * ESZ, ESZ, CO8, 1) : (((CO8 * 8 * ESZ / 16) % 2 == 0) ? CO8 * 8 * ESZ + 16 : CO8 * 8 * ESZ);
    static constexpr int KREAL = KS * KS * C8 * 8;
    static constexpr int KEXT = ((KREAL + 8 + 15) / 16) * 16;
    static constexpr int NKT = KEXT / 16;
    static constexpr int NKW = (NKT + 3) / 4;
    static constexpr int CTN = (CO8 + 1) / 2;
    static constexpr int X_BYTES = ((TI * TI * PSX + 15) / 16) * 16;
    static constexpr int G_BYTES = 256 * PSG;
    // PS ("pixel split"): with few accumulator tiles every wave keeps ALL of them and takes a quarter of the tile's pixels,
    // instead of a quarter of the k-tiles and all the pixels.  The k-tile split leaves waves idle whenever NKT is not a
    // multiple of 4 (8 -> 8 channels 3x3: 5 k-tiles, wave 0 does 2 and sets the pace; 1x1: 1 k-tile, three waves idle)
    // and makes every wave read the same g fragments.  The four partial sums meet once, in LDS, after the last tile.
    static constexpr bool PS = ESZ == 2 && NKT * CTN <= MSAU_WPS_MAX;
    // W2 ("two wave sets", the bf16 instances too big for PS): 8 waves per workgroup -- waves 0-3 take the first four 32-pixel
    // blocks of the tile, waves 4-7 the other four, each set split over the k-tiles as before.  The level-2/3 launches have
    // 64-107 workgroups, i.e. one per CU with ONE wave per SIMD: every fragment read -> MFMA chain ran fully exposed
    // (PMC: 27.7 us for an 11 MB layer).  The two partial sums meet in LDS after the last tile (fixed order).
#ifndef MSAU_WGRAD_W2
#define MSAU_WGRAD_W2 1
#endif
    static constexpr bool W2 = MSAU_WGRAD_W2 && ESZ == 2 && !PS;
    static constexpr int NTH = W2 ? 512 : 256;
    static constexpr int RED_BYTES = PS ? 4 * NKT * CTN * 1024 : 0;
    static constexpr int TILE_BYTES = X_BYTES + G_BYTES + 64;
    static constexpr int LDS = TILE_BYTES > RED_BYTES ? TILE_BYTES : RED_BYTES;
};

template <typename T, int C8, int CO8, int KS, int DIL = 1, int STRIDE = 1>
__global__ __launch_bounds__((WLeanCfg<T, C8, CO8, KS, DIL, STRIDE>::NTH)) void wgrad_lean_kernel(const WLeanMulti m) {
    const WLeanArgs a = m.a[blockIdx.z];                               // by value: one scalar load of the layer's descriptor
    using Cfg = WLeanCfg<T, C8, CO8, KS, DIL, STRIDE>;
    typedef typename Vec8<T>::type V8;
    constexpr int ESZ = Cfg::ESZ, TI = Cfg::TI, PSX = Cfg::PSX, PSG = Cfg::PSG, NKT = Cfg::NKT, NKW = Cfg::NKW, CTN = Cfg::CTN;
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* lds_x = smem;
    unsigned char* lds_g = smem + Cfg::X_BYTES;
    constexpr int ONES = Cfg::X_BYTES + Cfg::G_BYTES;            // {1,0,0,0,0,0,0,0} of T

    const msau_wgrad_desc& d = a.d;
    constexpr int NTH = Cfg::NTH;
    const int tid = threadIdx.x, lane = tid & 63;
    // k-tile owner (and, W2, pixel-block half).  Waves that own no k-tile in round i (wave + 4 i >= NKT) still issue that
    // round's MFMAs, on an all-zero column group: NO branch surrounds an MFMA of the tile loop.  With wave-uniform
    // `if (wave + 4 * i < NKT)` around the MFMAs, hipcc (ROCm 7.2) moved the accumulators between AGPRs and VGPRs at
    // every branch, and in one layout of the fp32 16 -> 8 3x3 instance the taken edge of such a branch reached
    // `v_accvgpr_read_b32 a7` four issue slots after the `v_mfma_f32_16x16x4_f32 a[4:7]` that writes it -- the 8-pass
    // MFMA needs 11; the fall-through edge had another MFMA in between and was fine -- so waves 2 and 3 wrote back stale
    // accumulators and k-tiles 6 and 7 came out short.  Straight-line code gives the hazard
    // recogniser nothing to miss, keeps the accumulators in AGPRs, and costs nothing: the waves that own every round set the pace.
    const int wave = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);
    const int wset = Cfg::W2 ? __builtin_amdgcn_readfirstlane(tid >> 8) : 0;
    const int li = lane & 15, lg = lane >> 4;
    const int chunk = blockIdx.y;
    const bool relu_in = d.flags & MSAU_CONV_RELU_IN;

    // source of this chunk (a chunk never straddles the two sources)
    const int c0 = chunk * C8 * 8;                                // first stored channel of the chunk
    const bool second = d.C2 != 0 && c0 >= d.C1;
    const char* xsrc = static_cast<const char*>(second ? d.x2 : d.x1);
    const int Cx = second ? d.C2 : d.C1;
    const int cb = (second ? c0 - d.C1 : c0) * ESZ;               // byte offset of the chunk inside a source pixel
    const int in_px = Cx * ESZ, in_row = d.Win * in_px;
    const int g_px = d.Cout * ESZ, g_row = d.Wout * g_px;

    if (tid < 8) reinterpret_cast<T*>(smem + ONES)[tid] = (T)(tid == 0 ? 1.0f : 0.0f);

    // column-group byte offsets (relative to the pixel's slot in lds_x) for the k-tiles this wave owns
    constexpr bool PS = Cfg::PS;
    constexpr int NACC = PS ? NKT : NKW;                            // accumulator k-tiles per wave
    int coloff[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        const int nkt = PS ? i : wave + 4 * i;
        const int k = nkt * 16 + (sizeof(T) == 2 ? (li & 3) * 4 : (li >> 2) * 4);     // first k of the lane's 4-column group
        int off;
        if (!PS && nkt >= NKT) off = WL_ABS | (ONES + 4 * ESZ);        // not this wave's round: the four zeros after the one
        else if (k < Cfg::KREAL) {
            const int tap = k / (C8 * 8), c = k - tap * (C8 * 8);
            const int ky = tap / KS, kx = tap - ky * KS;
            off = (ky * DIL * TI + kx * DIL) * PSX + c * ESZ;
        } else if (k == Cfg::KREAL) off = WL_ABS | ONES;
        else off = WL_ABS | (ONES + 4 * ESZ);
        if (sizeof(T) == 4) off += (li & 3) * 4;
        coloff[i] = off;
    }

    f32x4 acc[NACC][CTN];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int ct = 0; ct < CTN; ++ct) acc[i][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

    // software pipeline: the global loads of tile t+1 are issued (into registers) before the MFMA phase of
    // tile t and written to LDS at the top of the next iteration -- a persistent workgroup has only
    // 2 waves per SIMD, so without this every tile pays the full HBM latency.
    constexpr int NITX = (TI * TI * C8 + NTH - 1) / NTH, NITG = (256 * CO8 + NTH - 1) / NTH;
    // beyond 8 staging registers usually cost more than they hide; the exception is the network's first conv
    // (64 -> 8 channels: 11 + 1), whose weight gradient is the tail of the backward sweep (105.8 -> 96.7 us)
    constexpr bool PIPE = NITX + NITG <= 8 || (C8 == 8 && CO8 == 1);
    V8 xr[NITX], gr[NITG];
    auto issue_loads = [&](int tile) {
        const int t1 = a.tiles_x > 1 ? __umulhi((unsigned)tile, a.mag_tx) : tile;
        const int txi = tile - t1 * a.tiles_x;
        const int b = a.tiles_y > 1 ? __umulhi((unsigned)t1, a.mag_ty) : t1;
        const int tyi = t1 - b * a.tiles_y;
        const int oy0 = tyi * 16, ox0 = txi * 16;
        const int vy0 = oy0 * STRIDE - d.pad_t, vx0 = ox0 * STRIDE - d.pad_l;
        {
            constexpr int NITEMS = TI * TI * C8;
            const char* base = xsrc + (long long)b * d.Hin * in_row + cb;
#pragma unroll
            for (int it = 0; it < NITX; ++it) {
                const int idx = tid + it * NTH;
                xr[it] = zero8<T>();
                if ((it + 1) * NTH <= NITEMS || idx < NITEMS) {
                    const int pix = idx / C8, cg = idx - pix * C8;
                    const int iy = pix / TI, ix = pix - iy * TI;
                    const int vy = vy0 + iy, vx = vx0 + ix;
                    if ((unsigned)vy < (unsigned)d.Hin && (unsigned)vx < (unsigned)d.Win)
                        xr[it] = *reinterpret_cast<const V8*>(base + (unsigned)(vy * in_row + vx * in_px + cg * 8 * ESZ));
                }
            }
        }
        {
            const char* base = static_cast<const char*>(d.g) + (long long)b * d.Hout * g_row;
#pragma unroll
            for (int it = 0; it < NITG; ++it) {
                const int idx = tid + it * NTH;
                const int m = idx / CO8, cg = idx - m * CO8;
                const int oy = oy0 + (m >> 4), ox = ox0 + (m & 15);
                gr[it] = zero8<T>();                                 // pixels outside the image contribute 0
                if (idx < 256 * CO8 && oy < d.Hout && ox < d.Wout) gr[it] = *reinterpret_cast<const V8*>(base + (unsigned)(oy * g_row + ox * g_px + cg * 8 * ESZ));
            }
        }
    };
    if (PIPE && (int)blockIdx.x < a.ntiles) issue_loads(blockIdx.x);

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        __syncthreads();
        if (!PIPE) issue_loads(tile);
        {
            constexpr int NITEMS = TI * TI * C8;
#pragma unroll
            for (int it = 0; it < NITX; ++it) {
                const int idx = tid + it * NTH;
                if ((it + 1) * NTH <= NITEMS || idx < NITEMS) {
                    const int pix = idx / C8, cg = idx - pix * C8;
                    V8 v = xr[it];
                    if (relu_in) v = relu8<T>(v);
                    *reinterpret_cast<V8*>(lds_x + pix * PSX + cg * 8 * ESZ) = v;
                }
            }
#pragma unroll
            for (int it = 0; it < NITG; ++it) {
                const int idx = tid + it * NTH;
                const int m = idx / CO8, cg = idx - m * CO8;
                if (idx < 256 * CO8) *reinterpret_cast<V8*>(lds_g + m * PSG + cg * 8 * ESZ) = gr[it];
            }
        }
        __syncthreads();
        if (PIPE && tile + (int)gridDim.x < a.ntiles) issue_loads(tile + gridDim.x);

        if constexpr (PS) {
            const int q = li >> 2, p = li & 3;
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const int blk = wave * 2 + bb;                         // this wave's 32-pixel blocks
                const int row = blk * 2 + (lg >> 1), col = (lg & 1) * Cfg::QS + q;
                const unsigned char* ga = lds_g + (row * 16 + col) * PSG + 4 * p * 2;
                bf16x8 afrag[CTN];
#pragma unroll
                for (int ct = 0; ct < CTN; ++ct) {
                    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(ga + ct * 32));
                    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(ga + Cfg::HI * PSG + ct * 32));
                    afrag[ct] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
                const int pb0 = (row * STRIDE * TI + col * STRIDE) * PSX;
#pragma unroll
                for (int i = 0; i < NKT; ++i) {
                    const int e = coloff[i];
                    // only the k-tile that holds the ones / padding columns can carry an absolute offset: after unrolling
                    // the test is a constant and the selects vanish from the other tiles
                    const bool abs_ok = (i + 1) * 16 > Cfg::KREAL;
                    const int o0 = (abs_ok && (e & WL_ABS)) ? (e & ~WL_ABS) : pb0 + e;
                    const int o1 = (abs_ok && (e & WL_ABS)) ? (e & ~WL_ABS) : pb0 + Cfg::HI * STRIDE * PSX + e;
                    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(smem + o0));
                    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(smem + o1));
                    bf16x8 bfrag = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                    for (int ct = 0; ct < CTN; ++ct)
                        acc[i][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[ct], bfrag, acc[i][ct], 0, 0, 0);
                }
            }
        } else if constexpr (sizeof(T) == 2) {
            const int q = li >> 2, p = li & 3;
            constexpr int NBLK = Cfg::W2 ? 4 : 8;
            // 64 -> 64 3x3 (160 accumulator registers of the 256 a wave of a 512-thread group has): unrolled over the pixel
            // blocks the scheduler hoists the next block's fragment reads and spills 98 registers (20 -> 52 us); rolled: 243, none
            constexpr int UNR = NKW * CTN >= 40 ? 1 : NBLK;
#pragma unroll UNR
            for (int bq = 0; bq < NBLK; ++bq) {
                const int blk = Cfg::W2 ? wset * 4 + bq : bq;
                // pixels of lane group lg: row = blk*2 + (lg>>1), columns (lg&1)*QS + q and + HI (see WLeanCfg::NMAP)
                const int row = blk * 2 + (lg >> 1), col = (lg & 1) * Cfg::QS + q;
                const unsigned char* ga = lds_g + (row * 16 + col) * PSG + 4 * p * 2;
                bf16x8 afrag[CTN];
#pragma unroll
                for (int ct = 0; ct < CTN; ++ct) {
                    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(ga + ct * 32));
                    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(ga + Cfg::HI * PSG + ct * 32));
                    afrag[ct] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
                const int pb0 = (row * STRIDE * TI + col * STRIDE) * PSX;
#pragma unroll
                for (int i = 0; i < NKW; ++i) {                        // rounds this wave does not own read zeros (see `wave`)
                    const int e = coloff[i];
                    const int o0 = (e & WL_ABS) ? (e & ~WL_ABS) : pb0 + e;
                    const int o1 = (e & WL_ABS) ? (e & ~WL_ABS) : pb0 + Cfg::HI * STRIDE * PSX + e;
                    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(smem + o0));
                    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(smem + o1));
                    bf16x8 bfrag = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                    for (int ct = 0; ct < CTN; ++ct)
                        acc[i][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[ct], bfrag, acc[i][ct], 0, 0, 0);
                }
            }
        } else {
            for (int step = 0; step < 64; ++step) {
                const int m = step * 4 + lg;
                float afrag[CTN];
#pragma unroll
                for (int ct = 0; ct < CTN; ++ct) afrag[ct] = *reinterpret_cast<const float*>(lds_g + m * PSG + (ct * 16 + li) * 4);
                const int pb = ((m >> 4) * STRIDE * TI + (m & 15) * STRIDE) * PSX;
#pragma unroll
                for (int i = 0; i < NKW; ++i) {
                    const int e = coloff[i];
                    const float bv = *reinterpret_cast<const float*>(smem + ((e & WL_ABS) ? (e & ~WL_ABS) : pb + e));
#pragma unroll
                    for (int ct = 0; ct < CTN; ++ct)
                        acc[i][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[ct], bv, acc[i][ct], 0, 0, 0);
                }
            }
        }
    }

    float* slab = d.slabs + ((size_t)blockIdx.x * a.nchunks + chunk) * d.Cout * a.kextc;
    if constexpr (PS) {
        // the four waves' partial sums -> LDS; wave w then adds up k-tiles w, w + 4, ... in the fixed order wave 0..3
        __syncthreads();                                              // the last tile's fragment reads are done
        f32x4* red = reinterpret_cast<f32x4*>(smem);                  // [wave][k-tile][ct][lane]
#pragma unroll
        for (int i = 0; i < NKT; ++i)
#pragma unroll
            for (int ct = 0; ct < CTN; ++ct) red[((wave * NKT + i) * CTN + ct) * 64 + lane] = acc[i][ct];
        __syncthreads();
#pragma unroll
        for (int ii = 0; ii < NKW; ++ii) {
            const int nkt = wave + 4 * ii;
            if (nkt >= NKT) continue;
#pragma unroll
            for (int ct = 0; ct < CTN; ++ct) {
                f32x4 v = red[((0 * NKT + nkt) * CTN + ct) * 64 + lane];
#pragma unroll
                for (int w = 1; w < 4; ++w) v += red[((w * NKT + nkt) * CTN + ct) * 64 + lane];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int co = ct * 16 + lg * 4 + j;
                    if (co < d.Cout) slab[(size_t)co * a.kextc + nkt * 16 + li] = v[j];
                }
            }
        }
        return;
    }
    if constexpr (Cfg::W2) {
        // the second wave set parks its partial sums in LDS (as many k-tile rows per round as the tile area holds)
        constexpr int PER = Cfg::TILE_BYTES / (4 * CTN * 1024) < 1 ? 1 : Cfg::TILE_BYTES / (4 * CTN * 1024);
        f32x4* red = reinterpret_cast<f32x4*>(smem);                  // [wave][i in round][ct][lane]
        for (int i0 = 0; i0 < NKW; i0 += PER) {
            __syncthreads();
            if (wset == 1) {
#pragma unroll
                for (int i = 0; i < NKW; ++i)
                    if (i >= i0 && i < i0 + PER)
#pragma unroll
                        for (int ct = 0; ct < CTN; ++ct) red[((wave * PER + (i - i0)) * CTN + ct) * 64 + lane] = acc[i][ct];
            }
            __syncthreads();
            if (wset == 0) {
#pragma unroll
                for (int i = 0; i < NKW; ++i)
                    if (i >= i0 && i < i0 + PER)
#pragma unroll
                        for (int ct = 0; ct < CTN; ++ct) acc[i][ct] += red[((wave * PER + (i - i0)) * CTN + ct) * 64 + lane];
            }
        }
        if (wset == 1) return;
    }
#pragma unroll
    for (int i = 0; i < NKW; ++i) {
        const int nkt = wave + 4 * i;
        if (nkt >= NKT) continue;
#pragma unroll
        for (int ct = 0; ct < CTN; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = ct * 16 + lg * 4 + j;
                if (co < d.Cout) slab[(size_t)co * a.kextc + nkt * 16 + li] = acc[i][ct][j];
            }
    }
}


// ---- 64-channel chunk -> 8 output channels, 3x3 (the net's first conv: 64 / 768 one-hot input channels -> featRoot): the
// tail of the backward sweep, alone on the device.  With 8 output channels on the MFMA rows half of every MFMA is padding
// and 37 k-tiles of input fragments are read per 32 pixels.  Here the roles are swapped:
//     dW[ci][tap][co] = sum over INPUT pixels q of x[q][ci] * dy[q + pad - tap][co]
// rows = 64 input channels (4 tiles), columns = (tap, co) pairs (72 -> 5 tiles of two taps), K = the pixels of a 16 x 16
// INPUT tile; the gradient tile carries the halo (18 x 18 x 8 channels, a tenth of the bytes the input halo cost).  Per 32
// pixels: 4 + 5 fragments, 20 MFMAs (before: 1 + 37 fragments, 37 MFMAs); waves split the pixels, the 20 accumulator tiles
// meet in LDS after the last tile.  The bias gradient (the ones column of the slab) is summed from the gradient tile's
// interior by the threads that stage it.  Slab layout unchanged: slab[co][tap * 64 + ci], ones column at 576.
struct WIn64 {
    // (strides and pixel mapping: see WLeanCfg::NMAP -- x pixels 160 bytes apart, 8 consecutive columns per 32-lane half; the
    //  gradient tile's 16-byte pixels sit 48 bytes apart so that the 9 consecutive pixels two neighbouring taps touch differ in bank)
    static constexpr int PSX = 64 * 2 + 32, PSG = 8 * 2 + 32, TG = 18;
    static constexpr int X_BYTES = 256 * PSX, G_BYTES = TG * TG * PSG;
    static constexpr int ZERO = X_BYTES + G_BYTES;                  // 16 zero bytes: the missing tenth tap
    static constexpr int TILE_BYTES = ZERO + 64;
    static constexpr int RED_BYTES = 4 * 10 * 1024;                 // half of the 20 accumulator tiles of 4 waves
    static constexpr int LDS = TILE_BYTES > RED_BYTES ? TILE_BYTES : RED_BYTES;
};

// IDS (MSAU_CONV_IDS): d.x1 is the int32 id mask; the one-hot input tile is synthesised in LDS (bit-identical to the dense launch)
template <bool IDS>
__global__ __launch_bounds__(256) void wgrad_in64_kernel(const WLeanArgs a) {
    typedef bf16_t T;
    typedef Vec8<T>::type V8;
    constexpr int PSX = WIn64::PSX, PSG = WIn64::PSG, TG = WIn64::TG;
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* lds_x = smem;
    unsigned char* lds_g = smem + WIn64::X_BYTES;
    const msau_wgrad_desc& d = a.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int chunk = blockIdx.y;
    const bool relu_in = d.flags & MSAU_CONV_RELU_IN;
    const int c0 = chunk * 64;
    const bool second = d.C2 != 0 && c0 >= d.C1;
    const char* xsrc = static_cast<const char*>(second ? d.x2 : d.x1);
    const int Cx = second ? d.C2 : d.C1;
    const int cb = (second ? c0 - d.C1 : c0) * 2;
    const int in_px = Cx * 2, in_row = d.Win * in_px;
    const int g_px = d.Cout * 2, g_row = d.Wout * g_px;
    if (tid < 4) reinterpret_cast<unsigned*>(smem + WIn64::ZERO)[tid] = 0u;

    // B fragments: column group (li & 3) of N-tile nt = taps 2nt + (group >> 1), channels 4 * (group & 1) ..+3, read at
    // the gradient pixel (r + 2 - ky, c + 2 - kx) of the haloed tile
    int goff[5];
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) {
        const int grp = li & 3, tap = 2 * nt + (grp >> 1);
        const int ky = tap / 3, kx = tap - ky * 3;
        goff[nt] = tap < 9 ? ((2 - ky) * TG + (2 - kx)) * PSG + (grp & 1) * 8 : WL_ABS | WIn64::ZERO;
    }
    f32x4 acc[4][5];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 5; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[2][8];
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum[it][j] = 0.f;

    V8 xr[IDS ? 1 : 8], gr[2];
    int idr = -1;                                                  // IDS: this thread's pixel of the 16 x 16 input tile
    auto issue_loads = [&](int tile) {
        const int t1 = a.tiles_x > 1 ? __umulhi((unsigned)tile, a.mag_tx) : tile;
        const int txi = tile - t1 * a.tiles_x;
        const int b = a.tiles_y > 1 ? __umulhi((unsigned)t1, a.mag_ty) : t1;
        const int tyi = t1 - b * a.tiles_y;
        const int iy0 = tyi * 16, ix0 = txi * 16;
        const char* xb = xsrc + (long long)b * d.Hin * in_row + cb;
        if constexpr (IDS) {
            const int iy = iy0 + (tid >> 4), ix = ix0 + (tid & 15);
            idr = -1;
            if (iy < d.Hin && ix < d.Win) idr = static_cast<const int*>(d.x1)[((long long)b * d.Hin + iy) * d.Win + ix];
        } else
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = tid + it * 256;
            const int pix = idx >> 3, cg = idx & 7;
            const int iy = iy0 + (pix >> 4), ix = ix0 + (pix & 15);
            xr[it] = zero8<T>();
            if (iy < d.Hin && ix < d.Win) xr[it] = *reinterpret_cast<const V8*>(xb + (unsigned)(iy * in_row + ix * in_px + cg * 16));
        }
        const char* gb = static_cast<const char*>(d.g) + (long long)b * d.Hout * g_row;
        const int gy0 = iy0 + d.pad_t - 2, gx0 = ix0 + d.pad_l - 2;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int m = tid + it * 256;
            const int i = m / TG, j = m - i * TG;
            const int oy = gy0 + i, ox = gx0 + j;
            gr[it] = zero8<T>();
            if (m < TG * TG && (unsigned)oy < (unsigned)d.Hout && (unsigned)ox < (unsigned)d.Wout)
                gr[it] = *reinterpret_cast<const V8*>(gb + (unsigned)(oy * g_row + ox * g_px));
        }
    };
    if ((int)blockIdx.x < a.ntiles) issue_loads(blockIdx.x);

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        __syncthreads();
        if constexpr (IDS) {
#pragma unroll
            for (int cg = 0; cg < 8; ++cg) {
                V8 v;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (T)(cg * 8 + j == idr ? 1.0f : 0.0f);
                *reinterpret_cast<V8*>(lds_x + tid * PSX + cg * 16) = v;
            }
        } else
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = tid + it * 256;
            V8 v = xr[it];
            if (relu_in) v = relu8<T>(v);
            *reinterpret_cast<V8*>(lds_x + (idx >> 3) * PSX + (idx & 7) * 16) = v;
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int m = tid + it * 256;
            if (m < TG * TG) {
                *reinterpret_cast<V8*>(lds_g + m * PSG) = gr[it];
                // bias: every output pixel belongs to exactly one tile's rows / columns [2 - pad, 17 - pad]
                const int i = m / TG, j = m - i * TG;
                if (chunk == 0 && i >= 2 - d.pad_t && i <= 17 - d.pad_t && j >= 2 - d.pad_l && j <= 17 - d.pad_l) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) bsum[it][e] += (float)gr[it][e];
                }
            }
        }
        __syncthreads();
        if (tile + (int)gridDim.x < a.ntiles) issue_loads(tile + gridDim.x);

        const int q = li >> 2, p = li & 3;
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
            const int blk = wave * 2 + bb;
            const int row = blk * 2 + (lg >> 1), col = (lg & 1) * 4 + q;
            const unsigned char* xa = lds_x + (row * 16 + col) * PSX + 8 * p;
            bf16x8 afrag[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(xa + mt * 32));
                bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(xa + 8 * PSX + mt * 32));
                afrag[mt] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            const int pb0 = WIn64::X_BYTES + (row * TG + col) * PSG;
#pragma unroll
            for (int nt = 0; nt < 5; ++nt) {
                const int e = goff[nt];
                const bool abs_ok = nt == 4;                           // only the last column tile holds the missing tenth tap
                const int o0 = (abs_ok && (e & WL_ABS)) ? (e & ~WL_ABS) : pb0 + e;
                const int o1 = (abs_ok && (e & WL_ABS)) ? (e & ~WL_ABS) : pb0 + 8 * PSG + e;
                bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(smem + o0));
                bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(smem + o1));
                bf16x8 bfrag = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[mt], bfrag, acc[mt][nt], 0, 0, 0);
            }
        }
    }

    float* slab = d.slabs + ((size_t)blockIdx.x * a.nchunks + chunk) * d.Cout * a.kextc;
    f32x4* red = reinterpret_cast<f32x4*>(smem);                      // [wave][10 tiles][lane]
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 10; ++t) {
            const int tt = half * 10 + t;
            red[(wave * 10 + t) * 64 + lane] = acc[tt / 5][tt % 5];
        }
        __syncthreads();
#pragma unroll
        for (int ii = 0; ii < 3; ++ii) {
            const int t = wave + 4 * ii;                              // wave-uniform
            if (t >= 10) continue;
            const int tt = half * 10 + t, mt = tt / 5, nt = tt % 5;
            f32x4 v = red[(0 * 10 + t) * 64 + lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) v += red[(w * 10 + t) * 64 + lane];
            const int tap = 2 * nt + (li >> 3), co = li & 7;
            if (tap < 9 && co < d.Cout) {
#pragma unroll
                for (int j = 0; j < 4; ++j) slab[(size_t)co * a.kextc + tap * 64 + mt * 16 + lg * 4 + j] = v[j];
            }
        }
    }
    // bias gradient and the rest of the padded ones tile: k = 576 .. kext-1
    __syncthreads();
    float* bred = reinterpret_cast<float*>(smem);                     // [2][256][8]
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int e = 0; e < 8; ++e) bred[(it * 256 + tid) * 8 + e] = bsum[it][e];
    __syncthreads();
    if (tid < 8 && tid < d.Cout) {
        float s = 0.f;
        for (int t = 0; t < 512; ++t) s += bred[t * 8 + tid];        // fixed order
        for (int k = 576; k < a.kextc; ++k) slab[(size_t)tid * a.kextc + k] = k == 576 && chunk == 0 ? s : 0.f;
    }
}

int launch_wgrad_in64(hipStream_t s, const WLeanArgs& a) {
    if (a.kextc != 592) return 0;
    if (a.d.flags & MSAU_CONV_IDS) {
        if (a.nchunks != 1 || a.d.C2) return msau_set_error(MSAU_ERR_ARG, "wgrad: MSAU_CONV_IDS needs one 64-channel one-hot source");
        hipLaunchKernelGGL(wgrad_in64_kernel<true>, dim3(a.d.nslabs, 1), dim3(256), WIn64::LDS, s, a);
    } else
    hipLaunchKernelGGL(wgrad_in64_kernel<false>, dim3(a.d.nslabs, a.nchunks), dim3(256), WIn64::LDS, s, a);
    MSAU_CHECK_LAUNCH("wgrad_in64_kernel");
    return 1;
}

template <typename T, int C8, int CO8, int KS, int DIL = 1, int STRIDE = 1>
int launch_wlean(hipStream_t s, const WLeanMulti& m, int n) {
    using Cfg = WLeanCfg<T, C8, CO8, KS, DIL, STRIDE>;
    const WLeanArgs& a = m.a[0];
    if (Cfg::KEXT != a.kextc) return 0;                              // geometry disagrees with the generic planner
    static bool attr_set = false;
    if (!attr_set && Cfg::LDS > 60 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_lean_kernel<T, C8, CO8, KS, DIL, STRIDE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, MSAU_LDS_LIMIT);
        if (e != hipSuccess) return msau_set_error(MSAU_ERR_HIP, "wgrad_lean: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL((wgrad_lean_kernel<T, C8, CO8, KS, DIL, STRIDE>), dim3(a.d.nslabs, a.nchunks, n), dim3(Cfg::NTH), Cfg::LDS, s, m);
    MSAU_CHECK_LAUNCH("wgrad_lean_kernel");
    return 1;
}

template <typename T, int KS>
int wlean_dispatch(hipStream_t s, const WLeanMulti& m, int n, int c8, int co8) {
#define WL_CASE(C, O) if (c8 == C && co8 == O) return launch_wlean<T, C, O, KS>(s, m, n);
    WL_CASE(1, 1) WL_CASE(1, 2) WL_CASE(2, 1) WL_CASE(2, 2) WL_CASE(2, 4) WL_CASE(4, 2) WL_CASE(4, 4) WL_CASE(8, 1)
    WL_CASE(4, 8) WL_CASE(8, 4) WL_CASE(8, 8)
#undef WL_CASE
    return 0;
}

// dilated 3x3 (encoder level-entry convs) and the stride-2 form (transposed-conv weight gradient, roles swapped)
template <typename T>
int wlean_special(hipStream_t s, const WLeanMulti& m, int n, int c8, int co8, int dil, int stride) {
    if (stride == 2 && dil == 1) {
        if (c8 == 1 && co8 == 2) return launch_wlean<T, 1, 2, 3, 1, 2>(s, m, n);
        if (c8 == 2 && co8 == 4) return launch_wlean<T, 2, 4, 3, 1, 2>(s, m, n);
        if (c8 == 4 && co8 == 8) return launch_wlean<T, 4, 8, 3, 1, 2>(s, m, n);
        return 0;
    }
    if (stride == 1) {
        if (dil == 2 && c8 == 1 && co8 == 2) return launch_wlean<T, 1, 2, 3, 2, 1>(s, m, n);
        if (dil == 4 && c8 == 2 && co8 == 4) return launch_wlean<T, 2, 4, 3, 4, 1>(s, m, n);
        if (dil == 8 && c8 == 4 && co8 == 8) return launch_wlean<T, 4, 8, 3, 8, 1>(s, m, n);
    }
    return 0;
}

static bool wlean_special_shape(int c8, int co8, int dil, int stride) {
    if (stride == 2 && dil == 1) return (c8 == 1 && co8 == 2) || (c8 == 2 && co8 == 4) || (c8 == 4 && co8 == 8);
    if (stride == 1) return (dil == 2 && c8 == 1 && co8 == 2) || (dil == 4 && c8 == 2 && co8 == 4) || (dil == 8 && c8 == 4 && co8 == 8);
    return false;
}

}  // namespace

static const int kLeanShapes[][2] = {{1, 1}, {1, 2}, {2, 1}, {2, 2}, {2, 4}, {4, 2}, {4, 4}, {8, 1}, {4, 8}, {8, 4}, {8, 8}};

int msau_wgrad_lean_applicable(int dtype, const msau_wgrad_desc* d, int cch) {
    if (d->KH != d->KW || (d->KH != 1 && d->KH != 3 && d->KH != 4)) return 0;
    const bool special = d->stride != 1 || d->dil != 1;
    if (special) {
        if (d->KH != 3 || d->C2 || cch != d->C1 || !wlean_special_shape(cch / 8, d->Cout / 8, d->dil, d->stride)) return 0;
        if (d->pad_t != d->dil || d->pad_l != d->dil) return 0;
    } else {
        if (d->Hin != d->Hout || d->Win != d->Wout || d->pad_t < 0 || d->pad_l < 0 || d->pad_t >= d->KH || d->pad_l >= d->KW) return 0;
    }
    if (d->KH == 4 && !(cch == 8 && d->Cout == 8)) return 0;
    const int esz = dtype == MSAU_F32 ? 4 : 2;
    if ((int64_t)d->Hin * d->Win * (d->C1 > d->C2 ? d->C1 : d->C2) * esz >= (1ll << 31)) return 0;
    if ((int64_t)d->Hout * d->Wout * d->Cout * esz >= (1ll << 31)) return 0;
    const int tx = cdiv(d->Wout, 16), ty = cdiv(d->Hout, 16);
    if ((int64_t)d->B * tx * ty >= (1 << 20) || tx >= 4096 || ty >= 4096) return 0;
    if (special) return 1;
    for (auto& sh : kLeanShapes)
        if (sh[0] == cch / 8 && sh[1] == d->Cout / 8 && cch % 8 == 0) return 1;
    return 0;
}

static void wlean_fill(WLeanArgs& a, const msau_wgrad_desc* d, int nchunks, int kextc) {
    a.d = *d;
    a.kextc = kextc; a.nchunks = nchunks;
    a.tiles_x = cdiv(d->Wout, 16); a.tiles_y = cdiv(d->Hout, 16);
    a.ntiles = d->B * a.tiles_x * a.tiles_y;
    a.mag_tx = (unsigned)((0x100000000ull + a.tiles_x - 1) / a.tiles_x);
    a.mag_ty = (unsigned)((0x100000000ull + a.tiles_y - 1) / a.tiles_y);
}

static bool wlean_in64(int dtype, const msau_wgrad_desc* d, int cch) {
    static const bool in64_off = std::getenv("MSAU_WGRAD_IN64") && std::getenv("MSAU_WGRAD_IN64")[0] == '0';
    return !in64_off && dtype == MSAU_BF16 && d->stride == 1 && d->dil == 1 && d->KH == 3 && cch == 64 && d->Cout == 8 &&
           d->pad_t <= 2 && d->pad_l <= 2;
}

// 1 = handled, 0 = not applicable (use the generic kernel), < 0 = error.  cch/nchunks/kextc are the generic geometry.
// ds[0..n): launches of ONE instance and geometry (msau_conv2d_wgrad_group checks that), n <= kWGroup.
int msau_wgrad_lean_group(hipStream_t s, int dtype, const msau_wgrad_desc* const* ds, int n, int cch, int nchunks, int kextc) {
    const msau_wgrad_desc* d = ds[0];
    if (n < 1 || n > kWGroup || !msau_wgrad_lean_applicable(dtype, d, cch)) return 0;
    WLeanMulti m;
    for (int i = 0; i < n; ++i) wlean_fill(m.a[i], ds[i], nchunks, kextc);
    for (int i = n; i < kWGroup; ++i) m.a[i] = m.a[0];
    if (d->stride != 1 || d->dil != 1)
        return dtype == MSAU_F32 ? wlean_special<float>(s, m, n, cch / 8, d->Cout / 8, d->dil, d->stride)
                                 : wlean_special<bf16_t>(s, m, n, cch / 8, d->Cout / 8, d->dil, d->stride);
    if (wlean_in64(dtype, d, cch)) {
        if (n != 1) return 0;
        return launch_wgrad_in64(s, m.a[0]);
    }
    if (d->KH == 4) return dtype == MSAU_F32 ? launch_wlean<float, 1, 1, 4>(s, m, n) : launch_wlean<bf16_t, 1, 1, 4>(s, m, n);
    if (dtype == MSAU_F32) return d->KH == 3 ? wlean_dispatch<float, 3>(s, m, n, cch / 8, d->Cout / 8) : wlean_dispatch<float, 1>(s, m, n, cch / 8, d->Cout / 8);
    return d->KH == 3 ? wlean_dispatch<bf16_t, 3>(s, m, n, cch / 8, d->Cout / 8) : wlean_dispatch<bf16_t, 1>(s, m, n, cch / 8, d->Cout / 8);
}

int msau_wgrad_lean_try(hipStream_t s, int dtype, const msau_wgrad_desc* d, int cch, int nchunks, int kextc) {
    return msau_wgrad_lean_group(s, dtype, &d, 1, cch, nchunks, kextc);
}

// launches that may share a grid: the same lean instance (not the 64 -> 8 one)
int msau_wgrad_lean_groupable(int dtype, const msau_wgrad_desc* d, int cch) {
    return msau_wgrad_lean_applicable(dtype, d, cch) && !wlean_in64(dtype, d, cch);
}
