// Two chained 3x3 SAME convolutions C -> C -> C in ONE launch, the intermediate tensor never re-read from HBM:
//   forward : MultiConvResidualBlock with res_depth 2 (model/model.py:37-50)
//                 mid = ReLU(W1 * ReLU(x) + b1) ;  y = ReLU(W2 * mid + b2 + x)
//   backward: its two data gradients
//                 mid = (W2^T * g) . [r1 > 0]   ;  y = (W1^T * mid) . [x0 > 0] + g
// `mid` is still written once (the weight gradients and the ReLU mask of the backward need it), but the second conv
// reads it from LDS: per pair of launches 5 tensor passes (forward) / 7 (backward) become 3 / 5, and one dependent
// launch disappears.
//
// Tiling: a workgroup owns a 14 x (16*TW - 2) output tile.  The intermediate is computed on the 16 x 16*TW lattice
// around it (one pixel of halo, recomputed by the neighbouring tiles with the same arithmetic -> bit-identical), from
// an 18 x (16*TW + 2) input tile; both phases use the MFMA mapping of conv_lean.hip (channels on rows, 16 pixels of a
// row on columns, a wave owns 4 rows), the same packed weight images and the same epilogue arithmetic.  Intermediate
// positions outside the image are forced to 0: that is the SAME zero padding the second conv must see.
//
// Every global LOAD of a tile is issued at ONE point (the register prefetch of the next tile): the input tile and, in
// the backward, the two ReLU-mask tiles, which are squeezed to one bit per element in LDS.  The residual / other-path
// operand (MSAU_CONV_ADD) is the input tensor itself and is read back from the LDS tile.  Why: vmcnt retires in order,
// so a wait for an operand loaded inside an epilogue also drains the prefetch issued before it.  The first version
// loaded masks and the residual in the epilogues and ran at 8 us per tile -- slower than the two launches it replaced.
#include "msau_common.h"
#include <cstdlib>

namespace {

struct PairArgs {
    msau_conv_pair_desc d;
    int kchunk;                                  // packed K elements per weight row (both convs)
    int px, row;                                 // bytes per pixel / per image row
    int tiles_x, tiles_y, ntiles;
    unsigned mag_tx, mag_ty;
    int per_xcd;
#ifdef MSAU_STAMPS
    unsigned long long* stamps;                  // diagnostic build only: 16 words per workgroup (s_memrealtime at phase ends)
#endif
};
#ifdef MSAU_STAMPS
#define PSTAMP(i) do { if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PSTAMP(i) do {} while (0)
#endif

// RPW: lattice rows per wave.  4 = four waves per column tile (one per SIMD).  2 = eight: two waves per SIMD on the same tile,
// each with half the accumulators, half the epilogue and half the staging -- the 32-channel launches (levels 2 / 3: one or two
// tiles per workgroup, one workgroup per CU by LDS) ran every LDS read -> MFMA chain and every epilogue of a lone wave exposed:
// phase stamps, 84 x 64 x 32 forward: MFMA phases 1.56 + 1.44 us for 0.48 + 0.48 us of MFMA time (profiles/HISTORY_r03_r04.md).
template <typename T, int C8, int TW, bool BWD, int RPW = 4>
struct PairCfg {
    static constexpr int ESZ = (int)sizeof(T);
    static constexpr int C = C8 * 8;
    static constexpr int CT = (C8 + 1) / 2;                   // 16-row output-channel tiles
    static constexpr int IW = 16 * TW;                        // intermediate lattice: 16 x IW
    static constexpr int OH = 14, OW = IW - 2;                // output tile
    static constexpr int XH = 18, XW = IW + 2;                // input tile; the intermediate tile is allocated alike
    static constexpr int NWR = 16 / RPW;                      // waves per column tile
    static constexpr int NT = 64 * NWR * TW;
    static constexpr int PSRAW = C * ESZ;
    static constexpr int PS = lds_pixel_stride(PSRAW, ESZ, C8, 1);
    static constexpr int NPIX = XH * XW;
    static constexpr int NG = 9 * C8;                         // real 8-channel k-groups
    static constexpr int NKS = (NG + 3) / 4;                  // MFMA k-steps of 32
    static constexpr int WS = lds_wrow_stride(NKS, ESZ);
    static constexpr int W_BYTES = CT * 16 * WS;              // one conv's weights in LDS
    static constexpr int X_BYTES = ((NPIX * PS + 15) / 16) * 16;
    static constexpr int M_BYTES = BWD ? 16 * IW * C8 : 0;    // one mask tile: a byte per (lattice pixel, 8-channel group)
    static constexpr int OFF_W = 2 * X_BYTES;
    static constexpr int OFF_M = OFF_W + 2 * W_BYTES;
    static constexpr int OFF_B = OFF_M + 2 * M_BYTES;
    static constexpr int LDS = OFF_B + 2 * C * 4;              // + LDS_PAD bytes of dummy slots for lanes that own nothing
    static constexpr int LDS_PAD = 1024;
    static constexpr int NITX = (NPIX * C8 + NT - 1) / NT;    // prefetch registers (16 B each): input tile
    static constexpr int NITM = BWD ? (16 * IW * C8) / NT : 0;                 //                 each mask tile
    // ALLFRAG: all fragment reads of a phase (NKS x (RPW + CT) ds_read_b128) are issued before its first MFMA and waited for with
    // counted lgkmcnt: the compiler's own schedule kept them one k-step ahead, and with eight waves reading, an LDS round trip
    // (~350 cycles under that load) per k-step was the phase: 1.4 us for 0.5 us of MFMA / 0.5 us of LDS-array time (phase stamps,
    // 84 x 64 x 32).  Needs NKS * (RPW + CT) * 4 registers: the bf16 instances.
#ifndef MSAU_PAIR_ALLFRAG
#define MSAU_PAIR_ALLFRAG 1
#endif
    static constexpr bool ALLFRAG = MSAU_PAIR_ALLFRAG && ESZ == 2 && NKS * (RPW + CT) * 4 <= 160;
};

template <typename T> __device__ __forceinline__ unsigned positive_bits(typename Vec8<T>::type v) {
    unsigned b = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) b |= ((float)v[e] > 0.f ? 1u : 0u) << e;
    return b;
}

// ---- raw buffer accesses: an out-of-range offset makes a load return 0 and drops a store IN HARDWARE, so the zero
// padding, the partial tiles and the "not my pixel" lanes need no branch.  That matters beyond instruction count: with
// every VMEM instruction in straight-line code the compiler knows how many stores follow the prefetch loads and waits
// for the loads with vmcnt(N) instead of vmcnt(0) -- a wait that would otherwise also sit out the write
// acknowledgements of the stores issued a moment ago (measured: 8 us -> 5.7 us -> ... per tile, see the header).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned kOOB = 0x80000000u;                     // >= any num_records here (images are < 2^31 bytes), +16 does not wrap

__device__ __forceinline__ __amdgpu_buffer_rsrc_t image_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);     // gfx9 raw buffer, 32-bit format
}
template <typename T> __device__ __forceinline__ typename Vec8<T>::type buf_load8(__amdgpu_buffer_rsrc_t r, unsigned off);
template <> __device__ __forceinline__ bf16x8 buf_load8<bf16_t>(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
template <> __device__ __forceinline__ f32x8 buf_load8<float>(__amdgpu_buffer_rsrc_t r, unsigned off) {
    const f32x4 lo = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
    const f32x4 hi = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off + 16, 0, 0));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t r, unsigned off, bf16x4 v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, off, 0, 0);
}
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t r, unsigned off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, 0);
}

// POOL: the forward launch also writes the max-pooled output (MSAU_CONV_POOL).  A template parameter, not a run-time
// flag: epilogue code that is merely present slows the plain launches (see conv_lean.hip, EPI).
// BITS: the ReLU masks travel as bit planes ([B][H][W][C/8] bytes: one bit per element) -- the forward launch writes the
// planes of its input (x > 0) and of its intermediate (mid > 0), the backward launch reads them instead of re-reading the
// two bf16 tensors for one bit per element (110 -> 69 MB per backward launch at level 0).
// CPL (MSAU_PAIR_COUPLE, 32 channels): the coupling conv z = ReLU(Wc concat(prev, y) + bc) of a coupled stage (model/model.py:143-148,
// 246-252) in the forward launch's second epilogue; 2 = its max-pooled output as well (the encoder levels, model.py:158-160).  With two
// channel tiles the epilogue's result layout IS the MFMA's B-fragment layout -- a lane holds channels 8 lg .. 8 lg + 7 of its pixel -- so
// the rounded y goes from the epilogue's registers straight into the 1x1 conv: two MFMAs per channel tile and row (prev, then y: the
// stand-alone launch's chunk order), `prev` one 16-byte load per lane and row issued ahead of the tile's phases, no LDS.
// DCP (MSAU_PAIR_DCOUPLE, 32 channels, backward): the coupling conv's two-output data gradient as a PROLOGUE.  The launch's input tile
// is d(z), the gradient of the coupling conv's output, with the tile's 2-pixel halo; phase 0 turns it IN PLACE (LDS) into the tile of
// g = d(y) = (Wc[:, y]^T d(z)) . [y > 0] the two phases below read -- pointwise, so the halo costs 324 / 196 of the 1x1 conv's MFMAs
// (42 per tile) and no extra bytes -- and writes g and d(prev) = Wc[:, prev]^T d(z) of the tile's own pixels (the weight gradients
// of the block's second conv and of the previous stage read them).  A wave owns column tiles of 16 consecutive tile pixels; the
// result layout is the B-fragment layout again (two channel tiles), so a lane writes back the 16 bytes it read.
template <typename T, int C8, int TW, bool BWD, bool POOL = false, bool BITS = false, int RPW = 4, int CPL = 0, bool DCP = false>
__global__ __launch_bounds__((PairCfg<T, C8, TW, BWD, RPW>::NT)) void conv_pair_kernel(const PairArgs a) {
    static_assert(!(BWD && POOL), "the pooled output belongs to the forward launch");
    static_assert(CPL == 0 || (!BWD && !POOL && C8 == 4 && sizeof(T) == 2), "the coupling rider: forward, 32 channels, bf16");
    static_assert(!DCP || (BWD && C8 == 4 && TW == 1 && sizeof(T) == 2), "the coupling data-gradient prologue: backward, 32 channels, bf16");
    static_assert(RPW == 4 || RPW == 2, "rows per wave");
    using Cfg = PairCfg<T, C8, TW, BWD, RPW>;
    typedef typename Vec8<T>::type V8;
    typedef typename Vec4<T>::type V4;
    constexpr int ESZ = Cfg::ESZ, XW = Cfg::XW, IW = Cfg::IW, PS = Cfg::PS, NKS = Cfg::NKS, NG = Cfg::NG, CT = Cfg::CT, NT = Cfg::NT;
    extern __shared__ __align__(16) unsigned char smem[];
    PSTAMP(0);
    unsigned char* xt = smem;
    unsigned char* rt = smem + Cfg::X_BYTES;
    unsigned char* mt = smem + Cfg::OFF_M;                          // [16][IW][C8] bits of mask_mid at the lattice
    unsigned char* at = smem + Cfg::OFF_M + Cfg::M_BYTES;           // [16][IW][C8] bits of mask_a at the output tile
    float* lbias = reinterpret_cast<float*>(smem + Cfg::OFF_B);
    const msau_conv_pair_desc& d = a.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave_all & (Cfg::NWR - 1), cwt = wave_all / Cfg::NWR;
    const int lr = lane & 15, lg = lane >> 4;
    const int H = d.H, W = d.W;
    const unsigned img_bytes = (unsigned)H * (unsigned)a.row;

    // ---- weights and biases of both convs: LDS, once per persistent workgroup.  All their loads are issued at one point, into
    // registers, ahead of the first tile's loads; written as a loop of load -> LDS store the compiler kept it rolled with a
    // vmcnt(0) per trip: nine serialised L2 round trips, 2.0 us of every launch (phase stamps).
    constexpr int WG8 = NKS * 4, NWI = 2 * CT * 16 * WG8, NITW = (NWI + NT - 1) / NT;
    static_assert(2 * Cfg::C <= NT, "one bias per thread");
    V8 wreg[NITW];
    float breg = 0.f;
    {
#pragma unroll
        for (int it = 0; it < NITW; ++it) {
            const int idx = min(tid + it * NT, NWI - 1);
            const int which = idx / (CT * 16 * WG8), rem = idx - which * (CT * 16 * WG8);
            const int r = rem / WG8, g8 = rem - r * WG8;
            const T* wp = static_cast<const T*>(which ? d.w2 : d.w1);
            wreg[it] = load8<T>(wp + (size_t)r * a.kchunk + g8 * 8);
        }
        if (tid < 2 * Cfg::C) {
            const float* src = tid < Cfg::C ? d.b1 : d.b2;
            if (src) breg = src[tid < Cfg::C ? tid : tid - Cfg::C];
        }
    }
    auto write_weights = [&]() {
#pragma unroll
        for (int it = 0; it < NITW; ++it) {
            const int idx = tid + it * NT;
            const int which = idx / (CT * 16 * WG8), rem = idx - which * (CT * 16 * WG8);
            const int r = rem / WG8, g8 = rem - r * WG8;
            if ((it + 1) * NT <= NWI || idx < NWI)
                *reinterpret_cast<V8*>(smem + Cfg::OFF_W + which * Cfg::W_BYTES + r * Cfg::WS + g8 * 8 * ESZ) = wreg[it];
        }
        if (tid < 2 * Cfg::C) lbias[tid] = breg;
    };

    int koff[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int G = ks * 4 + lg;
        const int tap = G / C8, cg = G - tap * C8;
        const int ky = tap / 3, kx = tap - ky * 3;
        koff[ks] = G < NG ? (ky * XW + kx) * PS + cg * 8 * ESZ : 0;
    }
    const int pix_off = ((wave * RPW) * XW + cwt * 16 + lr) * PS;      // lattice (wave*4 + pt, cwt*16 + lr), pt adds XW*PS
    const int ch0 = lg * (CT * 4);                                  // this lane's first channel (ct adds 4)
    const bool ch_ok = C8 > 1 || lg < 2;                            // 8-channel layers fill half of the 16 MFMA rows
    const int jcol = cwt * 16 + lr;                                 // this lane's lattice / tile column
    const int lane_c = jcol * a.px + ch0 * ESZ;                     // per-lane part of the global epilogue offsets
    // LDS slot of this lane in the intermediate tile (row i adds XW*PS); lanes without channels write a dummy slot
    const int rt_lane = ch_ok ? Cfg::X_BYTES + jcol * PS + ch0 * ESZ : Cfg::LDS + (lane & 15) * 16;

    V8 pre_x[Cfg::NITX];
    constexpr int NCT0 = (Cfg::NPIX + 15) / 16, NP0 = (NCT0 + NT / 64 - 1) / (NT / 64);   // DCP: column tiles of the input tile, per wave
    V8 pre_o[DCP ? NP0 : 1];                                                               // ... and the y values (ReLU mask) at their pixels
    V8 pre_m[BWD && !BITS ? Cfg::NITM : 1], pre_a[BWD && !BITS ? Cfg::NITM : 1];
    unsigned char bit_m[BWD && BITS ? Cfg::NITM : 1], bit_a[BWD && BITS ? Cfg::NITM : 1];
    constexpr int NITB = (16 * IW * C8) / NT;                      // bit-plane items (a byte each) per thread
    auto decode = [&](int tile, int& b, int& ty0, int& tx0) {
        const int t1 = a.tiles_x > 1 ? __umulhi((unsigned)tile, a.mag_tx) : tile;
        tx0 = (tile - t1 * a.tiles_x) * Cfg::OW;
        b = a.tiles_y > 1 ? __umulhi((unsigned)t1, a.mag_ty) : t1;
        ty0 = (t1 - b * a.tiles_y) * Cfg::OH;
    };
    // all loads of one tile; `live` false (no next tile) turns every offset out of range: nothing is fetched
    auto issue_loads = [&](int tile, bool live) {
        int b, ty0, tx0;
        decode(live ? tile : 0, b, ty0, tx0);
        const long long img = (long long)b * img_bytes;
        const __amdgpu_buffer_rsrc_t rx = image_rsrc(static_cast<const char*>(DCP ? d.dcp_dz : d.x) + img, live ? img_bytes : 0u);
        if constexpr (DCP) {
            const __amdgpu_buffer_rsrc_t ro = image_rsrc(static_cast<const char*>(d.dcp_mask) + img, live ? img_bytes : 0u);
#pragma unroll
            for (int i = 0; i < NP0; ++i) {
                const int p = (wave_all + i * (NT / 64)) * 16 + lr;                       // tile pixel of this lane in its i-th column tile
                const int iy = p / XW, ix = p - iy * XW;
                const int vy = ty0 - 2 + iy, vx = tx0 - 2 + ix;
                const bool ok = p < Cfg::NPIX && (unsigned)vy < (unsigned)H && (unsigned)vx < (unsigned)W;
                pre_o[i] = buf_load8<T>(ro, ok ? (unsigned)(vy * a.row + vx * a.px + lg * 8 * ESZ) : kOOB);
            }
        }
        constexpr int NITEMS = Cfg::NPIX * C8;
#pragma unroll
        for (int it = 0; it < Cfg::NITX; ++it) {
            const int idx = tid + it * NT;
            const int pix = idx / C8, cg = idx - pix * C8;
            const int iy = pix / XW, ix = pix - iy * XW;
            const int vy = ty0 - 2 + iy, vx = tx0 - 2 + ix;
            const bool ok = ((it + 1) * NT <= NITEMS || idx < NITEMS) && (unsigned)vy < (unsigned)H && (unsigned)vx < (unsigned)W;
            pre_x[it] = buf_load8<T>(rx, ok ? (unsigned)(vy * a.row + vx * a.px + cg * 8 * ESZ) : kOOB);
        }
        if constexpr (BWD && BITS) {
            const unsigned plane = (unsigned)H * (unsigned)W * C8;
            const __amdgpu_buffer_rsrc_t rm = image_rsrc(d.bits_mid + (long long)b * plane, live ? plane : 0u);
            const __amdgpu_buffer_rsrc_t ra = image_rsrc(d.bits_a + (long long)b * plane, live ? plane : 0u);
#pragma unroll
            for (int it = 0; it < NITB; ++it) {
                const int idx = tid + it * NT;
                const int pix = idx / C8, cg = idx - pix * C8;
                const int iy = pix / IW, ix = pix - iy * IW;
                const int my = ty0 - 1 + iy, mx = tx0 - 1 + ix;      // lattice position
                const bool mok = (unsigned)my < (unsigned)H && (unsigned)mx < (unsigned)W;
                bit_m[it] = __builtin_amdgcn_raw_buffer_load_b8(rm, mok ? (unsigned)((my * W + mx) * C8 + cg) : kOOB, 0, 0);
                const int ay = ty0 + iy, ax = tx0 + ix;              // output position
                bit_a[it] = __builtin_amdgcn_raw_buffer_load_b8(ra, (ay < H && ax < W) ? (unsigned)((ay * W + ax) * C8 + cg) : kOOB, 0, 0);
            }
        } else if constexpr (BWD) {
            const __amdgpu_buffer_rsrc_t rm = image_rsrc(static_cast<const char*>(d.mask_mid) + img, live ? img_bytes : 0u);
            const __amdgpu_buffer_rsrc_t ra = image_rsrc(static_cast<const char*>(d.mask_a) + img, live ? img_bytes : 0u);
#pragma unroll
            for (int it = 0; it < Cfg::NITM; ++it) {
                const int idx = tid + it * NT;
                const int pix = idx / C8, cg = idx - pix * C8;
                const int iy = pix / IW, ix = pix - iy * IW;
                const int my = ty0 - 1 + iy, mx = tx0 - 1 + ix;      // lattice position
                const bool mok = (unsigned)my < (unsigned)H && (unsigned)mx < (unsigned)W;
                pre_m[it] = buf_load8<T>(rm, mok ? (unsigned)(my * a.row + mx * a.px + cg * 8 * ESZ) : kOOB);
                const int ay = ty0 + iy, ax = tx0 + ix;              // output position
                pre_a[it] = buf_load8<T>(ra, (ay < H && ax < W) ? (unsigned)(ay * a.row + ax * a.px + cg * 8 * ESZ) : kOOB);
            }
        }
    };
    auto write_tiles = [&]() {
        constexpr int NITEMS = Cfg::NPIX * C8;
#pragma unroll
        for (int it = 0; it < Cfg::NITX; ++it) {
            const int idx = tid + it * NT;
            const int pix = idx / C8, cg = idx - pix * C8;
            const bool ok = (it + 1) * NT <= NITEMS || idx < NITEMS;
            *reinterpret_cast<V8*>(smem + (ok ? pix * PS + cg * 8 * ESZ : Cfg::LDS + 256 + (tid & 15) * 32)) = pre_x[it];   // raw
        }
        if constexpr (BWD && BITS) {
#pragma unroll
            for (int it = 0; it < NITB; ++it) {
                const int idx = tid + it * NT;
                mt[idx] = bit_m[it];
                at[idx] = bit_a[it];
            }
        } else if constexpr (BWD) {
#pragma unroll
            for (int it = 0; it < Cfg::NITM; ++it) {
                const int idx = tid + it * NT;
                mt[idx] = (unsigned char)positive_bits<T>(pre_m[it]);
                at[idx] = (unsigned char)positive_bits<T>(pre_a[it]);
            }
        }
    };
    // the coupling conv's weights: packed forward image [chunk: prev, y][32 rows][32 k] (msau_conv_pack_geometry(32, 32, 32, 1x1)), bias
    V8 cA[CPL ? 2 : 1][CPL ? CT : 1];
    f32x4 cbias[CPL ? CT : 1];
    if constexpr (CPL != 0) {
        const T* cw = static_cast<const T*>(d.cpl_w);
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) cA[ch][ct] = load8<T>(cw + ((ch * CT + ct) * 16 + lr) * 32 + lg * 8);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) cbias[ct] = *reinterpret_cast<const f32x4*>(d.cpl_b + ch0 + ct * 4);
    }
    // DCP: rows of the two-output data-gradient image ([2C rows][kchunk 32], msau_conv2d's row order for 4 channel tiles: slot
    // ct4*16 + 4 q4 + j <-> output channel 16 q4 + 4 ct4 + j, first C = prev, last C = y) picked per lane so that MFMA row lr of channel
    // tile ct is channel 8 (lr >> 2) + 4 ct + (lr & 3) of its half: the two-tile result layout
    V8 dA[DCP ? 2 : 1][DCP ? CT : 1];
    if constexpr (DCP) {
        const T* dw = static_cast<const T*>(d.dcp_w);
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int call = half * Cfg::C + (lr >> 2) * 8 + ct * 4 + (lr & 3);
                const int slot = ((call >> 2) & 3) * 16 + 4 * (call >> 4) + (call & 3);
                dA[half][ct] = load8<T>(dw + slot * 32 + lg * 8);
            }
    }
    int tile0 = blockIdx.x, tend = a.ntiles, tstep = gridDim.x;
    if (a.per_xcd) {
        const int xcd = blockIdx.x & 7;
        tile0 = xcd * a.per_xcd + (blockIdx.x >> 3);
        tend = min(a.ntiles, (xcd + 1) * a.per_xcd);
        tstep = gridDim.x >> 3;
    }
    if (tile0 >= tend) return;                                 // workgroup-uniform
    issue_loads(tile0, true);
    PSTAMP(1);
    write_weights();
    PSTAMP(2);
    write_tiles();
    __syncthreads();
    PSTAMP(3);
    int nt_done = 0;

    for (int tile = tile0; tile < tend; tile += tstep) {
        int b, ty0, tx0;
        decode(tile, b, ty0, tx0);
        const long long img = (long long)b * img_bytes;
        if constexpr (DCP) {
            // ================= phase 0: the input tile d(z) -> g = (Wy^T d(z)) . [y > 0] in place; g and d(prev) of the own pixels out ====
            const __amdgpu_buffer_rsrc_t rg = image_rsrc(const_cast<char*>(static_cast<const char*>(d.x)) + img, img_bytes);
            const __amdgpu_buffer_rsrc_t rdp = image_rsrc(static_cast<char*>(d.dcp_dprev) + img, img_bytes);
#pragma unroll
            for (int i = 0; i < NP0; ++i) {
                const int t = wave_all + i * (NT / 64);                                    // wave-uniform
                if (t < NCT0) {
                    const int p = t * 16 + lr;
                    const bool valid = p < Cfg::NPIX;
                    unsigned char* slot = xt + (valid ? p : 0) * PS + lg * 8 * ESZ;
                    const V8 bz = *reinterpret_cast<const V8*>(slot);
                    const int iy = p / XW, ix = p - iy * XW;
                    const int vy = ty0 - 2 + iy, vx = tx0 - 2 + ix;
                    const bool own = valid && iy >= 2 && iy < 2 + Cfg::OH && ix >= 2 && ix < 2 + Cfg::OW && vy < H && vx < W;
                    const unsigned goff = own ? (unsigned)(vy * a.row + vx * a.px + lg * 8 * ESZ) : kOOB;
                    V4 gq[CT], pq[CT];
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        const f32x4 gp = mma8(dA[0][ct], bz, f32x4{0.f, 0.f, 0.f, 0.f});
                        const f32x4 gy = mma8(dA[1][ct], bz, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            pq[ct][jj] = (T)gp[jj];
                            gq[ct][jj] = (T)(((float)pre_o[i][ct * 4 + jj] > 0.f) ? gy[jj] : 0.f);       // MSAU_CONV_MASK_B of the y half
                        }
                    }
                    const V8 gv = __builtin_shufflevector(gq[0], gq[CT - 1], 0, 1, 2, 3, 4, 5, 6, 7);
                    const V8 pv = __builtin_shufflevector(pq[0], pq[CT - 1], 0, 1, 2, 3, 4, 5, 6, 7);
                    if (valid) *reinterpret_cast<V8*>(slot) = gv;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, gv), rg, goff, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pv), rdp, goff, 0, 0);
                }
            }
            __syncthreads();
        }
        // the coupling conv's first source at this lane's output pixels, k-group lg (ahead of the next tile's prefetch: vmcnt retires in order)
        V8 cprev[CPL ? RPW : 1];
        if constexpr (CPL != 0) {
            const __amdgpu_buffer_rsrc_t rprev = image_rsrc(static_cast<const char*>(d.cpl_prev) + img, img_bytes);
#pragma unroll
            for (int pt = 0; pt < RPW; ++pt) {
                const int yy = ty0 + wave * RPW + pt, xx = tx0 + jcol;
                cprev[pt] = buf_load8<T>(rprev, (yy < H && xx < W) ? (unsigned)(yy * a.row + xx * a.px + lg * 8 * ESZ) : kOOB);
            }
        }
        // the next tile's loads fly during both phases; they are waited for at the bottom, BEHIND this tile's stores
        issue_loads(tile + tstep, tile + tstep < tend);

        const __amdgpu_buffer_rsrc_t rmid = image_rsrc(static_cast<char*>(d.mid) + img, img_bytes);
        const __amdgpu_buffer_rsrc_t ry = image_rsrc(static_cast<char*>(d.y) + img, img_bytes);
        // ================= phase 1: intermediate on the 16 x IW lattice, image position (ty0-1+i, tx0-1+j) ===========
        {
            f32x4 acc[CT][RPW];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int pt = 0; pt < RPW; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (Cfg::ALLFRAG) {
                // every fragment read of the phase is issued before its first MFMA (see PairCfg::ALLFRAG)
                V8 bfr[NKS][RPW], afr[NKS][CT];
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const unsigned char* p = xt + pix_off + koff[ks];
#pragma unroll
                    for (int pt = 0; pt < RPW; ++pt) bfr[ks][pt] = *reinterpret_cast<const V8*>(p + pt * XW * PS);
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        afr[ks][ct] = *reinterpret_cast<const V8*>(smem + Cfg::OFF_W + (ct * 16 + lr) * Cfg::WS + (ks * 32 + lg * 8) * ESZ);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
                    for (int pt = 0; pt < RPW; ++pt)
                        if constexpr (!BWD) bfr[ks][pt] = relu8<T>(bfr[ks][pt]);     // MSAU_PAIR_RELU_IN
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                        for (int pt = 0; pt < RPW; ++pt) acc[ct][pt] = mma8(afr[ks][ct], bfr[ks][pt], acc[ct][pt]);
                }
            } else {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const unsigned char* p = xt + pix_off + koff[ks];
                V8 bfrag[RPW];
#pragma unroll
                for (int pt = 0; pt < RPW; ++pt) {
                    bfrag[pt] = *reinterpret_cast<const V8*>(p + pt * XW * PS);
                    if constexpr (!BWD) bfrag[pt] = relu8<T>(bfrag[pt]);         // MSAU_PAIR_RELU_IN
                }
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const V8 af = *reinterpret_cast<const V8*>(smem + Cfg::OFF_W + (ct * 16 + lr) * Cfg::WS + (ks * 32 + lg * 8) * ESZ);
#pragma unroll
                    for (int pt = 0; pt < RPW; ++pt) acc[ct][pt] = mma8(af, bfrag[pt], acc[ct][pt]);
                }
            }
            }
            if (nt_done == 0) PSTAMP(4);
            const int xx = tx0 - 1 + jcol;
            const bool colin = (unsigned)xx < (unsigned)W;
            const bool colown = ch_ok && colin && jcol >= 1 && jcol <= Cfg::OW;
            f32x4 bias[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) bias[ct] = *reinterpret_cast<const f32x4*>(lbias + (ch_ok ? ch0 + ct * 4 : 0));
#pragma unroll
            for (int pt = 0; pt < RPW; ++pt) {
                const int i = wave * RPW + pt;                   // wave-uniform
                const int yy = ty0 - 1 + i;
                const bool rowin = (unsigned)yy < (unsigned)H;
                const bool inimg = colin && rowin;
                const bool own = colown && rowin && i >= 1 && i <= Cfg::OH;
                const unsigned goff = (unsigned)(yy * a.row + (tx0 - 1) * a.px + lane_c);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    f32x4 v = acc[ct][pt];
                    if constexpr (!BWD) {
                        v += bias[ct];
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) v[jj] = fmaxf(v[jj], 0.f);        // MSAU_PAIR_RELU_MID
                    } else {
                        const unsigned bits = mt[ch_ok ? (i * IW + jcol) * C8 + ((ch0 + ct * 4) >> 3) : 0] >> ((ch0 + ct * 4) & 7);
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) v[jj] = (bits >> jj) & 1u ? v[jj] : 0.f;   // MSAU_PAIR_MASK_MID
                    }
                    V4 o;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) o[jj] = (T)(inimg ? v[jj] : 0.f);
                    *reinterpret_cast<V4*>(smem + rt_lane + (ch_ok ? i * XW * PS + ct * 4 * ESZ : 0)) = o;
                    buf_store4(rmid, own ? goff + ct * 4 * ESZ : kOOB, o);
                }
            }
        }
        __syncthreads();
        if (nt_done == 0) PSTAMP(5);
        if constexpr (!BWD && BITS) {
            // bit planes of the input (x > 0, raw tile) and of the intermediate (mid > 0) at this tile's own pixels
            const unsigned plane = (unsigned)H * (unsigned)W * C8;
            const __amdgpu_buffer_rsrc_t rbm = image_rsrc(d.bits_mid + (long long)b * plane, plane);
            const __amdgpu_buffer_rsrc_t rba = image_rsrc(d.bits_a + (long long)b * plane, plane);
#pragma unroll
            for (int it = 0; it < NITB; ++it) {
                const int idx = tid + it * NT;
                const int pix = idx / C8, cg = idx - pix * C8;
                const int i = pix / IW, j = pix - i * IW;             // lattice position
                const int yy = ty0 - 1 + i, xx = tx0 - 1 + j;
                const bool ownpx = i >= 1 && i <= Cfg::OH && j >= 1 && j <= Cfg::OW && yy < H && xx < W;
                const V8 xv = *reinterpret_cast<const V8*>(xt + ((i + 1) * XW + (j + 1)) * PS + cg * 8 * ESZ);
                const V8 rv = *reinterpret_cast<const V8*>(rt + (i * XW + j) * PS + cg * 8 * ESZ);
                const unsigned off = ownpx ? (unsigned)((yy * W + xx) * C8 + cg) : kOOB;
                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)positive_bits<T>(xv), rba, off, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)positive_bits<T>(rv), rbm, off, 0, 0);
            }
        }
        // ================= phase 2: output tile, image position (ty0+oy, tx0+ox), reads the intermediate from LDS ====
        {
            f32x4 acc[CT][RPW];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int pt = 0; pt < RPW; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (Cfg::ALLFRAG) {
                V8 bfr[NKS][RPW], afr[NKS][CT];
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const unsigned char* p = rt + pix_off + koff[ks];
#pragma unroll
                    for (int pt = 0; pt < RPW; ++pt) bfr[ks][pt] = *reinterpret_cast<const V8*>(p + pt * XW * PS);
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        afr[ks][ct] = *reinterpret_cast<const V8*>(smem + Cfg::OFF_W + Cfg::W_BYTES + (ct * 16 + lr) * Cfg::WS + (ks * 32 + lg * 8) * ESZ);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                        for (int pt = 0; pt < RPW; ++pt) acc[ct][pt] = mma8(afr[ks][ct], bfr[ks][pt], acc[ct][pt]);
            } else {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const unsigned char* p = rt + pix_off + koff[ks];
                V8 bfrag[RPW];
#pragma unroll
                for (int pt = 0; pt < RPW; ++pt) bfrag[pt] = *reinterpret_cast<const V8*>(p + pt * XW * PS);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const V8 af = *reinterpret_cast<const V8*>(smem + Cfg::OFF_W + Cfg::W_BYTES + (ct * 16 + lr) * Cfg::WS + (ks * 32 + lg * 8) * ESZ);
#pragma unroll
                    for (int pt = 0; pt < RPW; ++pt) acc[ct][pt] = mma8(af, bfrag[pt], acc[ct][pt]);
                }
            }
            }
            if (nt_done == 0) PSTAMP(6);
            const bool colok = ch_ok && jcol < Cfg::OW && tx0 + jcol < W;
            f32x4 bias[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) bias[ct] = *reinterpret_cast<const f32x4*>(lbias + Cfg::C + (ch_ok ? ch0 + ct * 4 : 0));
            // the residual (forward) / other-path gradient (backward) operand is the input tensor itself: read it back
            // from the raw LDS input tile, position (oy + 2, ox + 2); lanes without channels read slot 0
            const int xt_lane = ch_ok ? (2 * XW + jcol + 2) * PS + ch0 * ESZ : 0;
            constexpr bool PL = POOL || CPL == 2;                            // a pooled output: of y, or of the coupling conv's z
            V4 keep[PL ? CT : 1][RPW];                                       // MSAU_CONV_POOL: the rounded results, 0 where nothing is stored
            const __amdgpu_buffer_rsrc_t rz = image_rsrc(CPL ? static_cast<char*>(d.cpl_y) + img : nullptr, CPL ? img_bytes : 0u);
#pragma unroll
            for (int pt = 0; pt < RPW; ++pt) {
                const int oy = wave * RPW + pt;                  // wave-uniform
                const int yy = ty0 + oy;
                const bool ok = colok && oy < Cfg::OH && yy < H;
                const unsigned goff = (unsigned)(yy * a.row + tx0 * a.px + lane_c);
                V4 yv[CT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const V4 r = *reinterpret_cast<const V4*>(xt + xt_lane + (ch_ok ? oy * XW * PS + ct * 4 * ESZ : 0));
                    f32x4 v = acc[ct][pt];
                    if constexpr (!BWD) {
                        v += bias[ct];
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) v[jj] = fmaxf(v[jj] + (float)r[jj], 0.f);      // ADD, RELU_OUT
                    } else {
                        const unsigned bits = at[ch_ok ? (oy * IW + jcol) * C8 + ((ch0 + ct * 4) >> 3) : 0] >> ((ch0 + ct * 4) & 7);
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) v[jj] = ((bits >> jj) & 1u ? v[jj] : 0.f) + (float)r[jj];   // MASK_A, ADD
                    }
                    V4 ov;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) ov[jj] = (T)v[jj];
                    buf_store4(ry, ok ? goff + ct * 4 * ESZ : kOOB, ov);
                    yv[ct] = ov;
                    if constexpr (POOL) {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) keep[ct][pt][jj] = ok ? ov[jj] : (T)0.f;
                    }
                }
                if constexpr (CPL != 0) {
                    // z = ReLU(Wc[:, prev] prev + Wc[:, y] y + bc): the rounded y of this lane IS the B fragment (channels 8 lg .. 8 lg + 7)
                    const V8 by = __builtin_shufflevector(yv[0], yv[CT - 1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        f32x4 zv = mma8(cA[0][ct], cprev[pt], f32x4{0.f, 0.f, 0.f, 0.f});
                        zv = mma8(cA[1][ct], by, zv);
                        zv += cbias[ct];
                        V4 zo;
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) zo[jj] = (T)fmaxf(zv[jj], 0.f);
                        buf_store4(rz, ok ? goff + ct * 4 * ESZ : kOOB, zo);
                        if constexpr (CPL == 2) {
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) keep[ct][pt][jj] = ok ? zo[jj] : (T)0.f;
                        }
                    }
                }
            }
            // MaxPool2d(2,2) of the zero-padded output (MSAU_CONV_POOL, model/model.py:158-160): tile origins are even, a
            // window is rows (2pp, 2pp + 1) of this lane and of its neighbour column lr ^ 1; even lanes write.  Same order
            // of comparisons (first maximum wins) and the same rounded values as msau_maxpool2x2_fwd on y.
            if constexpr (PL) {
                {
                    const int Ho = (H + 1) >> 1, Wo = (W + 1) >> 1;
                    constexpr int ND = (int)sizeof(V4) / 4;
                    typedef int dwords __attribute__((ext_vector_type(ND)));
                    const unsigned pimg = (unsigned)Ho * (unsigned)Wo * (unsigned)a.px;
                    void* const pool_y = CPL == 2 ? d.cpl_pool_y : d.pool_y;
                    uint8_t* const pool_idx = CPL == 2 ? d.cpl_pool_idx : d.pool_idx;
                    const __amdgpu_buffer_rsrc_t rp = image_rsrc(static_cast<char*>(pool_y) + (long long)b * pimg, pimg);
                    const __amdgpu_buffer_rsrc_t ri = image_rsrc(pool_idx ? pool_idx + (long long)b * (pimg / ESZ) : nullptr,
                                                                 pool_idx ? pimg / ESZ : 0u);
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
                        for (int pp = 0; pp < RPW / 2; ++pp) {
                            V4 nb[2];
#pragma unroll
                            for (int r = 0; r < 2; ++r) {
                                dwords src = __builtin_bit_cast(dwords, keep[ct][2 * pp + r]), dst;
#pragma unroll
                                for (int w = 0; w < ND; ++w) dst[w] = __builtin_amdgcn_mov_dpp(src[w], 0xB1, 0xf, 0xf, true);
                                nb[r] = __builtin_bit_cast(V4, dst);
                            }
                            const int oy = wave * RPW + 2 * pp;
                            const bool pok = colok && !(lr & 1) && oy < Cfg::OH && ty0 + oy < H;
                            V4 best;
                            unsigned idx = 0;
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) {
                                float bv_ = (float)keep[ct][2 * pp][jj];
                                unsigned bi = 0;
                                const float c1 = (float)nb[0][jj], c2 = (float)keep[ct][2 * pp + 1][jj], c3 = (float)nb[1][jj];
                                if (c1 > bv_) { bv_ = c1; bi = 1; }
                                if (c2 > bv_) { bv_ = c2; bi = 2; }
                                if (c3 > bv_) { bv_ = c3; bi = 3; }
                                best[jj] = (T)bv_;
                                idx |= bi << (8 * jj);
                            }
                            const unsigned e = (unsigned)((((ty0 + oy) >> 1) * Wo + ((tx0 + jcol) >> 1)) * Cfg::C + ch0 + ct * 4);   // elements
                            buf_store4(rp, pok ? e * ESZ : kOOB, best);
                            __builtin_amdgcn_raw_buffer_store_b32(idx, ri, pok ? e : kOOB, 0, 0);
                        }
                    }
                }
            }
        }
        __syncthreads();                                       // every read of the LDS tiles is done
        if (nt_done == 0) PSTAMP(7);
        write_tiles();                                         // needs the prefetch only: vmcnt(stores of this tile)
        __syncthreads();
        if (nt_done == 0) PSTAMP(8);
        ++nt_done;
    }
    PSTAMP(9);
#ifdef MSAU_STAMPS
    if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 16 + 10] = nt_done;
#endif
}

template <typename T, int C8, int TW, bool BWD, bool POOL = false, bool BITS = false, int RPW = 4, int CPL = 0, bool DCP = false>
int launch_pair(hipStream_t s, const PairArgs& a0) {
    using Cfg = PairCfg<T, C8, TW, BWD, RPW>;
    static_assert(Cfg::LDS + Cfg::LDS_PAD + 256 <= MSAU_LDS_LIMIT, "conv_pair instance does not fit the LDS");
    PairArgs a = a0;
    a.tiles_x = cdiv(a.d.W, Cfg::OW);
    a.tiles_y = cdiv(a.d.H, Cfg::OH);
    a.ntiles = a.d.B * a.tiles_x * a.tiles_y;
    a.mag_tx = (unsigned)((0x100000000ull + a.tiles_x - 1) / a.tiles_x);
    a.mag_ty = (unsigned)((0x100000000ull + a.tiles_y - 1) / a.tiles_y);
    static bool attr_set = false;
    if (!attr_set && Cfg::LDS + Cfg::LDS_PAD > 60 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pair_kernel<T, C8, TW, BWD, POOL, BITS, RPW, CPL, DCP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, MSAU_LDS_LIMIT);
        if (e != hipSuccess) return msau_set_error(MSAU_ERR_HIP, "conv_pair: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    // persistent grid = what is RESIDENT at once (registers and LDS both count): workgroups beyond that would start
    // only after the first ones have finished their whole share of tiles and run on a third-empty machine
    static int per_cu = 0;
    if (!per_cu) {
        // (hipOccupancyMaxActiveBlocksPerMultiprocessor budgets 64 KB of LDS per CU, not gfx950's 160 KB: compute it here)
        hipFuncAttributes fa;
        hipError_t e = hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&conv_pair_kernel<T, C8, TW, BWD, POOL, BITS, RPW, CPL, DCP>));
        if (e != hipSuccess) return msau_set_error(MSAU_ERR_HIP, "conv_pair: hipFuncGetAttributes: %s", hipGetErrorString(e));
        const int vgprs = ((fa.numRegs > 0 ? fa.numRegs : 64) + 7) & ~7;
        int waves_per_simd = 512 / vgprs;                      // 512 VGPRs per SIMD lane, 8 waves at most
        waves_per_simd = waves_per_simd > 8 ? 8 : waves_per_simd;
        const int by_regs = waves_per_simd / (Cfg::NT / 256);  // a workgroup puts NT / 256 waves on every SIMD
        const int by_lds = MSAU_LDS_LIMIT / (Cfg::LDS + Cfg::LDS_PAD + 256);
        int n = by_regs < by_lds ? by_regs : by_lds;
        static const int percu_max = std::getenv("MSAU_PAIR_PERCU") ? atoi(std::getenv("MSAU_PAIR_PERCU")) : 8;
        n = n > percu_max ? percu_max : n;
        per_cu = n < 1 ? 1 : n;
    }
    int grid = 256 * per_cu;
    if (grid > a.ntiles) grid = a.ntiles;
    static const bool xcd_off = std::getenv("MSAU_XCD") && std::getenv("MSAU_XCD")[0] == '0';
    a.per_xcd = 0;
    if (!xcd_off && grid >= 64) {
        grid &= ~7;
        a.per_xcd = cdiv(a.ntiles, 8);
    }
    hipLaunchKernelGGL((conv_pair_kernel<T, C8, TW, BWD, POOL, BITS, RPW, CPL, DCP>), dim3(grid), dim3(Cfg::NT), Cfg::LDS + Cfg::LDS_PAD, s, a);
    MSAU_CHECK_LAUNCH("conv_pair_kernel");
    return 0;
}

// tile width: 30-pixel tiles (512 threads) for wide images, 14-pixel tiles otherwise (less padding waste, more tiles)
int pair_tw(int dtype, const msau_conv_pair_desc* d) {
    const int c8 = d->C / 8;
    if (c8 == 4 || (dtype == MSAU_F32 && c8 == 2)) return 1;     // LDS: the wide tile does not fit
    static const int force = std::getenv("MSAU_PAIR_TW") ? atoi(std::getenv("MSAU_PAIR_TW")) : 0;
    if (force == 1 || force == 2) return force;
    return d->W >= 120 ? 2 : 1;
}

// the two flag combinations the kernel is compiled for: the residual block's forward, and its data gradient
#ifndef MSAU_PAIR_RPW32
#define MSAU_PAIR_RPW32 2                      // rows per wave of the 32-channel bf16 instances (PairCfg)
#endif
constexpr int kFwd1 = MSAU_PAIR_RELU_IN | MSAU_PAIR_RELU_MID, kFwd2 = MSAU_CONV_ADD | MSAU_CONV_RELU_OUT;
constexpr int kBwd1 = MSAU_PAIR_MASK_MID, kBwd2 = MSAU_CONV_MASK_A | MSAU_CONV_ADD;

}  // namespace

extern "C" int msau_conv_pair_applicable(int dtype, const msau_conv_pair_desc* d) {
    if (!d || (dtype != MSAU_F32 && dtype != MSAU_BF16)) return 0;
    static const int maxc = std::getenv("MSAU_PAIR_MAXC") ? atoi(std::getenv("MSAU_PAIR_MAXC")) : 32;   // measured: C = 32 (84x64 images) gains nothing, 4.03 vs 4.00 ms/step
    // ... at batch 16; a launch of a few hundred pixels is launch-bound whatever it computes: there the 32-channel block fuses too
    const bool tiny = (int64_t)d->B * d->H * d->W <= 16384;
    if ((d->C != 8 && d->C != 16 && d->C != 32) || (d->C > maxc && !(tiny && d->C == 32))) return 0;
    if (dtype == MSAU_F32 && d->C == 32) return 0;                 // two fp32 tiles + two weight sets exceed the LDS
    if (d->B <= 0 || d->H <= 0 || d->W <= 0) return 0;
    const int f1 = d->flags1 & ~(MSAU_PAIR_TILES | MSAU_PAIR_LRN_BWD | MSAU_PAIR_WGRAD1 | MSAU_PAIR_COUPLE | MSAU_PAIR_DCOUPLE);
    const bool fwd = f1 == kFwd1 && (d->flags2 & ~MSAU_CONV_POOL) == kFwd2, bwd = f1 == kBwd1 && d->flags2 == kBwd2;
    if (fwd && (d->flags2 & MSAU_CONV_POOL) && !d->pool_y) return 0;
    if (!fwd && !bwd) return 0;
    if (d->add != d->x) return 0;                                  // the ADD operand is read back from the input tile
    if ((d->flags1 & MSAU_PAIR_DCOUPLE) && !bwd) return 0;
    if ((d->flags1 & MSAU_PAIR_COUPLE) && !fwd) return 0;
    if (msau_rowpair_takes(dtype, d)) return 1;                    // 8 channels, bf16: the row-streaming kernel (conv_rows.hip)
    if (d->flags1 & (MSAU_PAIR_LRN_BWD | MSAU_PAIR_WGRAD1)) return 0;              // riders of the row-streaming instances only
    if (d->flags1 & MSAU_PAIR_DCOUPLE) {                                           // the coupling conv's data gradients as the backward's prologue
        static const bool dcp32 = !(std::getenv("MSAU_PAIR_DCOUPLE32") && std::getenv("MSAU_PAIR_DCOUPLE32")[0] == '0');
        if (!(dcp32 && dtype == MSAU_BF16 && d->C == 32 && d->dcp_dz && d->dcp_w && d->dcp_mask && d->dcp_dprev)) return 0;
        msau_conv_pack_geom g;                                                     // the image the prologue indexes: [64 rows][kchunk 32]
        if (msau_conv_pack_geometry(dtype, 32, 0, 64, 1, 1, 1, 1, 1, &g) != 0 || g.nchunks != 1 || g.kchunk != 32 || g.rows != 64) return 0;
    }
    if (d->flags1 & MSAU_PAIR_COUPLE) {                                            // ... the coupling conv: also the 32-channel bf16 forward tile pair
        static const bool cpl32 = !(std::getenv("MSAU_PAIR_COUPLE32") && std::getenv("MSAU_PAIR_COUPLE32")[0] == '0');
        if (!(cpl32 && fwd && dtype == MSAU_BF16 && d->C == 32 && !(d->flags2 & MSAU_CONV_POOL) && d->cpl_prev && d->cpl_w && d->cpl_b && d->cpl_y))
            return 0;
    }
    const int esz = dtype == MSAU_F32 ? 4 : 2;
    if ((int64_t)d->H * d->W * d->C * esz >= (1ll << 31)) return 0;             // 32-bit lane offsets inside an image
    const int tw = pair_tw(dtype, d);
    const int64_t tiles = (int64_t)d->B * cdiv(d->H, 14) * cdiv(d->W, 16 * tw - 2);
    static const int min_tiles = std::getenv("MSAU_PAIR_MIN_TILES") ? atoi(std::getenv("MSAU_PAIR_MIN_TILES")) : 1;
    // (round 2 sent launches below 64 tiles to the one-conv kernels; at those sizes the step is launch-bound -- the batch-1
    //  document loop of the reference, tools/funsd_loop.py -- and one launch instead of two is what counts)
    if (tiles < min_tiles || tiles >= (1 << 20)) return 0;
    if (cdiv(d->H, 14) >= 4096 || cdiv(d->W, 16 * tw - 2) >= 4096) return 0;   // the tile decode (__umulhi) is exact below 2^12 tiles per axis
    return 1;
}

// which instance msau_conv_pair launches for this descriptor: 0 none, 1 a tile kernel (conv_pair.hip), 2 a row-streaming kernel
// (conv_rows.hip) -- for profiling / roofline labels, like msau_conv2d_launch_info
extern "C" int msau_conv_pair_instance(int dtype, const msau_conv_pair_desc* d) {
    if (!msau_conv_pair_applicable(dtype, d)) return 0;
    return msau_rowpair_takes(dtype, d) ? 2 : 1;
}

extern "C" int msau_conv_pair_wgrad_slabs(int dtype, const msau_conv_pair_desc* d) {
    if (!d || !(d->flags1 & MSAU_PAIR_WGRAD1) || !msau_conv_pair_applicable(dtype, d) || !msau_rowpair_takes(dtype, d)) return 0;
    return msau_rowpair_workgroups(d);
}

// bytes of ONE ReLU-mask plane (bits_mid / bits_a) for this descriptor: the tile kernels keep a byte per (pixel, 8-channel
// group), the row-streaming kernel 32 bytes of lane ballots per (row, 30-column strip)
extern "C" int64_t msau_conv_pair_bits_bytes(int dtype, const msau_conv_pair_desc* d) {
    if (!d || d->C <= 0 || d->B <= 0 || d->H <= 0 || d->W <= 0) return 0;
    msau_conv_pair_desc probe = *d;                                // "would it take this shape once it has planes?"
    uint8_t one = 0;
    probe.bits_mid = probe.bits_a = &one;
    if (msau_rowpair_takes(dtype, &probe)) return msau_rowpair_plane_bytes(d);
    return (int64_t)d->B * d->H * d->W * (d->C / 8);
}

extern "C" int msau_conv_pair(void* stream, int dtype, const msau_conv_pair_desc* d) {
    MSAU_CHECK_ARG(d && d->x && d->w1 && d->w2 && d->mid && (d->y || (d->flags1 & MSAU_PAIR_LRN_BWD)), "conv_pair: null pointer");
    MSAU_CHECK_ARG(msau_conv_pair_applicable(dtype, d), "conv_pair: unsupported shape or flags (C %d, %dx%d, B %d, flags 0x%x / 0x%x; "
                   "MSAU_CONV_ADD must name the input tensor)", d->C, d->H, d->W, d->B, d->flags1, d->flags2);
    const bool bwd = (d->flags1 & ~(MSAU_PAIR_TILES | MSAU_PAIR_LRN_BWD | MSAU_PAIR_WGRAD1 | MSAU_PAIR_DCOUPLE)) == kBwd1;
    MSAU_CHECK_ARG(!bwd || (d->mask_mid && d->mask_a) || (d->bits_mid && d->bits_a), "conv_pair: backward without mask_mid / mask_a (tensors or bit planes)");
    MSAU_CHECK_ARG(!d->bits_mid == !d->bits_a, "conv_pair: bits_mid and bits_a come together");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (msau_rowpair_takes(dtype, d)) return msau_rowpair_launch(s, d);
    const int esz = dtype == MSAU_F32 ? 4 : 2;
    PairArgs a;
    a.d = *d;
#ifdef MSAU_STAMPS
    a.stamps = std::getenv("MSAU_STAMP_PTR") ? reinterpret_cast<unsigned long long*>(strtoull(std::getenv("MSAU_STAMP_PTR"), nullptr, 0)) : nullptr;
#endif
    a.kchunk = roundup(9 * d->C, 32);
    a.px = d->C * esz;
    a.row = d->W * a.px;
    const int c8 = d->C / 8, tw = pair_tw(dtype, d);
    const bool pool = !bwd && (d->flags2 & MSAU_CONV_POOL);
    const bool bits = d->bits_mid && d->bits_a;
#define PAIR_CASE(T, C8V, TWV, RPWV) if (c8 == C8V && tw == TWV) { \
        if (bwd) return bits ? launch_pair<T, C8V, TWV, true, false, true, RPWV>(s, a) : launch_pair<T, C8V, TWV, true, false, false, RPWV>(s, a); \
        if (pool) return bits ? launch_pair<T, C8V, TWV, false, true, true, RPWV>(s, a) : launch_pair<T, C8V, TWV, false, true, false, RPWV>(s, a); \
        return bits ? launch_pair<T, C8V, TWV, false, false, true, RPWV>(s, a) : launch_pair<T, C8V, TWV, false, false, false, RPWV>(s, a); }
    if (dtype == MSAU_BF16 && (d->flags1 & MSAU_PAIR_DCOUPLE))                     // (applicable: backward, 32 channels)
        return bits ? launch_pair<bf16_t, 4, 1, true, false, true, MSAU_PAIR_RPW32, 0, true>(s, a) : launch_pair<bf16_t, 4, 1, true, false, false, MSAU_PAIR_RPW32, 0, true>(s, a);
    if (dtype == MSAU_BF16 && (d->flags1 & MSAU_PAIR_COUPLE)) {                    // (applicable: forward, 32 channels, no pooled y)
        if (d->cpl_pool_y) return bits ? launch_pair<bf16_t, 4, 1, false, false, true, MSAU_PAIR_RPW32, 2>(s, a) : launch_pair<bf16_t, 4, 1, false, false, false, MSAU_PAIR_RPW32, 2>(s, a);
        return bits ? launch_pair<bf16_t, 4, 1, false, false, true, MSAU_PAIR_RPW32, 1>(s, a) : launch_pair<bf16_t, 4, 1, false, false, false, MSAU_PAIR_RPW32, 1>(s, a);
    }
    if (dtype == MSAU_BF16) {
        PAIR_CASE(bf16_t, 1, 1, 4) PAIR_CASE(bf16_t, 1, 2, 4) PAIR_CASE(bf16_t, 2, 1, 4) PAIR_CASE(bf16_t, 2, 2, 4) PAIR_CASE(bf16_t, 4, 1, MSAU_PAIR_RPW32)
    } else {
        PAIR_CASE(float, 1, 1, 4) PAIR_CASE(float, 1, 2, 4) PAIR_CASE(float, 2, 1, 4)
    }
#undef PAIR_CASE
    return msau_set_error(MSAU_ERR_ARG, "conv_pair: no instance for C %d tile width %d", d->C, tw);
}
