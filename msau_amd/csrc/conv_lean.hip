// Lean instances of the implicit-GEMM convolution for the layers that dominate the step: stride 1,
// no dilation, k in {1,3}, one K chunk, 16x16-pixel tiles, channel counts known at compile time.
//
// Why a second kernel: the generic conv_kernel spends ~300 VALU + 150 SALU instructions per 64-pixel
// wave-tile on run-time index arithmetic and is instruction-issue bound.  Here
// every divisor, LDS stride and tap offset is a compile-time constant, the grid is 3-D (no tile
// decode), global addresses are a scalar base plus a 32-bit lane offset, LDS fragment addresses are
// one VGPR plus immediates, and small weight sets live in registers.  Same arithmetic, same packed
// weight image, same epilogue contract as conv.hip (include/msau_hip.h).
#include "msau_common.h"
#include <cstdlib>

namespace {

struct LeanArgs {
    msau_conv_desc d;
    int kchunk;                                  // packed K elements per weight row
    int in_px1, in_px2, in_row1, in_row2;        // bytes
    int out_px, out_row;                         // bytes
    int tiles_x, tiles_y, ntiles;
    unsigned mag_tx, mag_ty;                     // ceil(2^32 / tiles_x), ceil(2^32 / tiles_y)
    int per_xcd;                                 // tiles per XCD chunk (0: plain grid-stride tile order)
    int ct_total;                                // SPLIT: 16-row output-channel tiles of the whole conv (grid.y of them)
#ifdef MSAU_STAMPS
    unsigned long long* stamps;                  // diagnostic build only: 8 words per workgroup (s_memrealtime at phase ends)
#endif
};
#ifdef MSAU_STAMPS
#define STAMP(i) do { if (a.stamps && threadIdx.x == 0) a.stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

template <typename T, int CIN8, int CT, int KS, bool DUAL, int DIL = 1, int WGW = 1, int STRIDE = 1>
struct LeanCfg {
    static constexpr int ESZ = (int)sizeof(T);
    static constexpr int TI = 15 * STRIDE + 1 + (KS - 1) * DIL;               // input tile rows
    static constexpr int TIW = (16 * WGW - 1) * STRIDE + 1 + (KS - 1) * DIL;  // input tile columns (output tile = 16 x 16*WGW)
    static constexpr int NT = 256 * WGW;                      // threads per workgroup
    static constexpr int PSRAW = CIN8 * 8 * ESZ;
    static constexpr int NPIX = TI * TIW;
    // the packed image is chunk-major and a chunk never straddles the two sources: dual = 2 chunks
    static constexpr int NCH = DUAL ? 2 : 1;
    static constexpr int C8H = CIN8 / NCH;                    // 8-channel groups per chunk
    static constexpr int PS = lds_pixel_stride(PSRAW, ESZ, C8H, STRIDE);      // pixel stride of the input tile in LDS
    static constexpr int NG = KS * KS * C8H;                  // real 8-channel k-groups per chunk
    static constexpr int NKSH = (NG + 3) / 4;                 // MFMA k-steps (32 k each) per chunk
    static constexpr int NKS = NCH * NKSH;
    static constexpr bool WREG = CT * NKS <= 8;               // weight fragments held in registers
    static constexpr int WS = lds_wrow_stride(NKSH, ESZ);     // LDS weight row stride (when !WREG), per chunk
    static constexpr int IN_BYTES = ((NPIX * PS + 15) / 16) * 16;
    static constexpr int LDS = IN_BYTES + (WREG ? 0 : NCH * CT * 16 * WS);
};

// SPLIT: the output-channel tiles of one conv are spread over blockIdx.y (CT = 1 per workgroup).  For the level-3
// layers (42x32 pixels: 96 pixel tiles for 256 CUs) this is what fills the device; every workgroup stages the same
// input tile (an L2 hit for all but the first) and a quarter of the weights.
// STRIDE = 2: the data gradient of a transposed conv (each output pixel gathers input pixels 2y + k - pad).
// UPS = 2: the transposed conv itself as a conv over the zero-stuffed input: the LDS tile is the virtual image, only
// its even/even positions are loaded (the MFMAs run over the zeros; the launch is bound by its 4x larger output).
// EOP: the epilogue-OPERAND bits of the flags (ADD, ACCUM, MASK_A, MASK_B) as a compile-time constant, or -1 = read them at
// run time.  The 8- and 16-channel instances are instruction-bound; with the unused operand paths compiled out a plain
// 8 -> 8 3x3 launch takes 11.1 instead of 12.2 us (tools/small_bench.py).  RELU_IN / RELU_OUT stay run-time flags.
constexpr int kOperandBits = MSAU_CONV_RELU_IN | MSAU_CONV_RELU_OUT | MSAU_CONV_ADD | MSAU_CONV_ACCUM | MSAU_CONV_MASK_A | MSAU_CONV_MASK_B;
// EPI: extra output compiled in -- 0 none, 1 LocalResponseNorm (MSAU_CONV_LRN), 2 max pool (MSAU_CONV_POOL).  Compile-time,
// not a run-time flag: with the epilogue code merely PRESENT every plain launch of the instance ran 1.3-4 us slower
// (8 -> 8 3x3: 12.8 -> 15.8 us, tools/small_bench.py), which ate most of what the fusion saved.
enum { EPI_NONE = 0, EPI_LRN = 1, EPI_POOL = 2, EPI_HEAD = 3 };
template <typename T, int CIN8, int CT, int KS, bool DUAL, int DIL = 1, int WGW = 1, bool DOUT = false, bool SPLIT = false,
          int STRIDE = 1, int UPS = 1, int EPI = EPI_NONE, int EOP = -1, bool IDS = false>
__global__ __launch_bounds__(256 * WGW) void conv_lean_kernel(const LeanArgs a) {
    // IDS (MSAU_CONV_IDS): x1 is an int32 id mask [B][Hin][Win]; input channel c of a pixel is (id == c).  The one-hot tile is
    // synthesised in LDS -- the dense one-hot tensor is never painted nor read (the net's first conv fed with
    // character ids), and the MFMA sequence is the dense launch's: bit-identical results.
    static_assert(!IDS || (!DUAL && STRIDE == 1 && UPS == 1 && !DOUT && !SPLIT), "id-mask input: plain single-source instances");
    static_assert(!SPLIT || (CT == 1 && !DUAL && !DOUT && WGW == 1), "SPLIT instances are single-source, one tile per workgroup");
    static_assert((STRIDE == 1 && UPS == 1) || (!DUAL && !DOUT && WGW == 1 && DIL == 1 && STRIDE * UPS == 2), "strided / upsampling instances");
    STAMP(0);
    const int cty = SPLIT ? (int)blockIdx.y : 0;                 // this workgroup's channel tile
    const int CTT = SPLIT ? a.ct_total : CT;                     // channel tiles of the conv
    using Cfg = LeanCfg<T, CIN8, CT, KS, DUAL, DIL, WGW, STRIDE>;
    constexpr int NT = Cfg::NT;
    typedef typename Vec8<T>::type V8;
    typedef typename Vec4<T>::type V4;
    constexpr int ESZ = Cfg::ESZ, TI = Cfg::TIW, PS = Cfg::PS, NKS = Cfg::NKS, NG = Cfg::NG, NKSH = Cfg::NKSH, C8H = Cfg::C8H;
    extern __shared__ __align__(16) unsigned char smem[];
    const msau_conv_desc& d = a.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave_all & 3, cwt = wave_all >> 2;       // 4 row groups x WGW column tiles
    const int lr = lane & 15, lg = lane >> 4;
    const bool relu_in = (EOP >= 0 ? EOP : d.flags) & MSAU_CONV_RELU_IN;

    // ---- weights: registers (small) or LDS, once per (persistent) workgroup
    const T* wp = static_cast<const T*>(d.wpack) + (SPLIT ? (size_t)cty * 16 * a.kchunk : 0);     // [row][k], one chunk
    V8 afr[Cfg::WREG ? CT : 1][Cfg::WREG ? NKS : 1];
    if constexpr (Cfg::WREG) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
                afr[ct][ks] = load8<T>(wp + ((size_t)(ks / NKSH) * (CT * 16) + ct * 16 + lr) * a.kchunk + (ks % NKSH) * 32 + lg * 8);
    }

    // ---- per-lane LDS offsets of the k-groups this lane feeds (lane group lg of every k-step)
    int koff[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int G = (ks % NKSH) * 4 + lg;                    // k-group inside chunk ks / NKSH
        const int tap = G / C8H, cg = G - tap * C8H;
        const int ky = tap / KS, kx = tap - ky * KS;
        koff[ks] = G < NG ? (ky * DIL * TI + kx * DIL) * PS + ((ks / NKSH) * C8H + cg) * 8 * ESZ : 0;
    }
    const unsigned char* pixp = smem + ((wave * 4) * STRIDE * TI + (cwt * 16 + lr) * STRIDE) * PS;

    const int flags = EOP >= 0 ? ((d.flags & ~kOperandBits) | EOP) : d.flags;
    const int Cout = d.Cout;
    f32x4 bv[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        bv[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (d.bias && lg * (CTT * 4) + (cty + ct) * 4 < Cout) bv[ct] = *reinterpret_cast<const f32x4*>(d.bias + lg * (CTT * 4) + (cty + ct) * 4);
    }
    const long long delta_add = static_cast<const char*>(d.add) - static_cast<const char*>(d.y);
    const long long delta_ma = static_cast<const char*>(d.mask_a) - static_cast<const char*>(d.y);
    const long long delta_mb = static_cast<const char*>(d.mask_b) - static_cast<const char*>(d.y);
    const int lane_out = (cwt * 16 + lr) * a.out_px + (lg * (CTT * 4) + cty * 4) * ESZ;

    // ---- staging: one source at a time so the base pointer stays scalar.  With few items per thread the
    // loads of tile t+1 are issued into registers before the MFMAs of tile t (software pipeline).
    constexpr int C8S = DUAL ? CIN8 / 2 : CIN8;                // 8-channel groups per source
    constexpr int NITEMS = Cfg::NPIX * C8S;
    constexpr int NIT = (NITEMS + NT - 1) / NT;
    constexpr int NSRC = DUAL ? 2 : 1;
    constexpr bool PIPE = IDS || NIT * NSRC <= 6;
    V8 pre[IDS ? 1 : NSRC][IDS ? 1 : NIT];
    constexpr int NIDS = (Cfg::NPIX + NT - 1) / NT;            // id-mask input: tile pixels per thread
    int pre_id[IDS ? NIDS : 1];
    auto decode = [&](int tile, int& b, int& oy0, int& ox0) {
        const int t1 = a.tiles_x > 1 ? __umulhi((unsigned)tile, a.mag_tx) : tile;
        ox0 = (tile - t1 * a.tiles_x) * (16 * WGW);
        b = a.tiles_y > 1 ? __umulhi((unsigned)t1, a.mag_ty) : t1;
        oy0 = (t1 - b * a.tiles_y) * 16;
    };
    auto issue_loads = [&](int tile) {
        int b, oy0, ox0;
        decode(tile, b, oy0, ox0);
        const int vy0 = oy0 * STRIDE - d.pad_t, vx0 = ox0 * STRIDE - d.pad_l;    // forward: SAME pad; data gradient: (k-1) - pad
        if constexpr (IDS) {
            const int* ids = static_cast<const int*>(d.x1) + (long long)b * d.Hin * d.Win;
#pragma unroll
            for (int it = 0; it < NIDS; ++it) {
                const int pix = tid + it * NT;
                const int iy = pix / TI, ix = pix - iy * TI;
                const int vy = vy0 + iy, vx = vx0 + ix;
                pre_id[it] = -1;
                if (pix < Cfg::NPIX && (unsigned)vy < (unsigned)d.Hin && (unsigned)vx < (unsigned)d.Win) pre_id[it] = ids[vy * d.Win + vx];
            }
        } else
#pragma unroll
        for (int sidx = 0; sidx < NSRC; ++sidx) {
            const char* base = static_cast<const char*>(sidx ? d.x2 : d.x1) + (long long)b * d.Hin * (sidx ? a.in_row2 : a.in_row1);
            const int in_row = sidx ? a.in_row2 : a.in_row1, in_px = sidx ? a.in_px2 : a.in_px1;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = tid + it * NT;
                pre[sidx][it] = zero8<T>();
                if ((it + 1) * NT <= NITEMS || idx < NITEMS) {
                    const int pix = idx / C8S, cg = idx - pix * C8S;
                    const int iy = pix / TI, ix = pix - iy * TI;
                    int vy = vy0 + iy, vx = vx0 + ix;
                    bool ok;
                    if constexpr (UPS == 2) {                      // zero-stuffed image: data at even / even virtual positions
                        ok = vy >= 0 && vx >= 0 && !((vy | vx) & 1) && (vy >> 1) < d.Hin && (vx >> 1) < d.Win;
                        vy >>= 1; vx >>= 1;
                    } else {
                        ok = (unsigned)vy < (unsigned)d.Hin && (unsigned)vx < (unsigned)d.Win;
                    }
                    if (ok)
                        pre[sidx][it] = *reinterpret_cast<const V8*>(base + (unsigned)(vy * in_row + vx * in_px + cg * 8 * ESZ));
                }
            }
        }
    };
    auto write_lds = [&]() {
        if constexpr (IDS) {
#pragma unroll
            for (int it = 0; it < NIDS; ++it) {
                const int pix = tid + it * NT;
                if (pix < Cfg::NPIX) {
                    const int id = pre_id[it];
#pragma unroll
                    for (int cg = 0; cg < C8S; ++cg) {
                        V8 v;
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = (T)(cg * 8 + j == id ? 1.0f : 0.0f);
                        *reinterpret_cast<V8*>(smem + pix * PS + cg * 8 * ESZ) = v;
                    }
                }
            }
        } else
#pragma unroll
        for (int sidx = 0; sidx < NSRC; ++sidx)
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = tid + it * NT;
                if ((it + 1) * NT <= NITEMS || idx < NITEMS) {
                    const int pix = idx / C8S, cg = idx - pix * C8S;
                    V8 v = pre[sidx][it];
                    if (relu_in) v = relu8<T>(v);
                    *reinterpret_cast<V8*>(smem + pix * PS + (sidx * C8S + cg) * 8 * ESZ) = v;
                }
            }
    };
    // Workgroup i is dispatched to XCD i % 8 and every XCD has its own L2: give each XCD one contiguous run of tiles
    // so that the halo rows and columns neighbouring tiles share are L2 hits instead of second HBM reads.
    int tile0 = blockIdx.x, tend = a.ntiles, tstep = gridDim.x;
    if (a.per_xcd) {
        const int xcd = blockIdx.x & 7;
        tile0 = xcd * a.per_xcd + (blockIdx.x >> 3);
        tend = min(a.ntiles, (xcd + 1) * a.per_xcd);
        tstep = gridDim.x >> 3;
    }
    // (Round 4 tried the first tile's loads ahead of the weight loads in the !PIPE instances too, and the epilogue operands requested
    //  before the MFMAs: phase stamps showed no gain -- the loads of a workgroup complete in order, the staging is bound by the
    //  ~60 KB it moves, not by round trips -- and the step LOST 20 us (3.182 vs 3.162 ms, interleaved A/B) to the extra live registers:
    //  removed again, as round 2 had found for the PIPE instances.  profiles/HISTORY_r03_r04.md)
    if (PIPE && tile0 < tend) issue_loads(tile0);
    if constexpr (!Cfg::WREG) {
        // weights -> LDS, behind the first tile's loads and up to 8 loads in flight per thread.  As a plain loop this was
        // load -> wait -> write per 256 x 16 bytes BEFORE the first tile was even requested: 3 (32 -> 16 rows, 3x3) to 18
        // (64 -> 64) serial memory round trips at the head of kernels that run 5-10 us in all.
        unsigned char* lds_w = smem + Cfg::IN_BYTES;
        constexpr int WG8 = NKSH * 4;                          // 8-element groups per (chunk, row)
        constexpr int WTOT = Cfg::NCH * CT * 16 * WG8, NWIT = (WTOT + NT - 1) / NT, WB = NWIT < 8 ? NWIT : 8;
#pragma unroll
        for (int it0 = 0; it0 < NWIT; it0 += WB) {
            V8 wr[WB];
#pragma unroll
            for (int j = 0; j < WB; ++j) {
                const int idx = tid + (it0 + j) * NT;
                if (it0 + j < NWIT && idx < WTOT) {
                    const int r = idx / WG8, g8 = idx - r * WG8;   // r = chunk * rows + row: the packed image order
                    wr[j] = load8<T>(wp + (size_t)r * a.kchunk + g8 * 8);
                }
            }
#pragma unroll
            for (int j = 0; j < WB; ++j) {
                const int idx = tid + (it0 + j) * NT;
                if (it0 + j < NWIT && idx < WTOT) {
                    const int r = idx / WG8, g8 = idx - r * WG8;
                    *reinterpret_cast<V8*>(lds_w + r * Cfg::WS + g8 * 8 * ESZ) = wr[j];
                }
            }
        }
    }

    STAMP(1);
    for (int tile = tile0; tile < tend; tile += tstep) {
        int b, oy0, ox0;
        decode(tile, b, oy0, ox0);
        __syncthreads();                                       // previous tile's fragment reads are done
        if (!PIPE) issue_loads(tile);
        write_lds();
        __syncthreads();
        if (tile == tile0) STAMP(2);
        if (PIPE && tile + tstep < tend) issue_loads(tile + tstep);
        const int oyw = oy0 + wave * 4;
        f32x4 acc[CT][4];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const unsigned char* p = pixp + koff[ks];
            V8 bfrag[4];
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) bfrag[pt] = *reinterpret_cast<const V8*>(p + pt * STRIDE * TI * PS);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                V8 af;
                if constexpr (Cfg::WREG) af = afr[ct][ks];
                else af = *reinterpret_cast<const V8*>(smem + Cfg::IN_BYTES + ((ks / NKSH) * (CT * 16) + ct * 16 + lr) * Cfg::WS +
                                                       ((ks % NKSH) * 32 + lg * 8) * ESZ);
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) acc[ct][pt] = mma8(af, bfrag[pt], acc[ct][pt]);
            }
        }

        // ---- epilogue: lane (pixel lr of row ty, q = lg) owns channels q*CT*4 + ct*4 + {0..3}
        if (tile == tile0) { asm volatile("" :: "v"(acc[0][0][0])); STAMP(3); }
        if constexpr (DOUT) {
            // two output tensors (MSAU_CONV_DOUT): Cout = 2 * CT*8, so lane groups q = 0,1 hold the channels of y and
            // q = 2,3 those of y2; flags, base and mask pointers are picked per lane
            const bool t2 = lg >= 2;
            const int lfl = t2 ? d.flags2 : flags;
            char* yb = static_cast<char*>(t2 ? d.y2 : d.y);
            const char* mb = static_cast<const char*>(t2 ? d.mask_b2 : d.mask_b);
            const char* ad = static_cast<const char*>(d.add);
            if (ox0 + lr < d.Wout) {
                const long long off0 = ((long long)(b * d.Hout + oyw)) * a.out_row + (long long)ox0 * a.out_px +
                                       lr * a.out_px + (lg & 1) * (CT * 4) * ESZ;
                // operand loads of all four rows first, then arithmetic and stores: with load -> wait -> store per row (the
                // stores may alias the next row's loads as far as the compiler knows) a launch paid up to 4 x 3 memory
                // round trips in a row at the end of every tile
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    V4 r_add[4], r_acc[4], r_mb[4];
#pragma unroll
                    for (int pt = 0; pt < 4; ++pt) {
                        if (oyw + pt < d.Hout) {                   // scalar
                            const long long o = off0 + ct * 4 * ESZ + (long long)pt * a.out_row;
                            if (lfl & MSAU_CONV_ADD) r_add[pt] = *reinterpret_cast<const V4*>(ad + o);
                            if (lfl & MSAU_CONV_ACCUM) r_acc[pt] = *reinterpret_cast<const V4*>(yb + o);
                            if (lfl & MSAU_CONV_MASK_B) r_mb[pt] = *reinterpret_cast<const V4*>(mb + o);
                        }
                    }
#pragma unroll
                    for (int pt = 0; pt < 4; ++pt) {
                        if (oyw + pt < d.Hout) {                   // scalar
                            const long long o = off0 + ct * 4 * ESZ + (long long)pt * a.out_row;
                            f32x4 v = acc[ct][pt] + bv[ct];
                            if (lfl & MSAU_CONV_ADD) {
#pragma unroll
                                for (int j = 0; j < 4; ++j) v[j] += (float)r_add[pt][j];
                            }
                            if (lfl & MSAU_CONV_ACCUM) {
#pragma unroll
                                for (int j = 0; j < 4; ++j) v[j] += (float)r_acc[pt][j];
                            }
                            if (lfl & MSAU_CONV_MASK_B) {
#pragma unroll
                                for (int j = 0; j < 4; ++j) v[j] = ((float)r_mb[pt][j] > 0.f) ? v[j] : 0.f;
                            }
                            V4 ov;
#pragma unroll
                            for (int j = 0; j < 4; ++j) ov[j] = (T)v[j];
                            *reinterpret_cast<V4*>(yb + o) = ov;
                        }
                    }
                }
            }
        } else {
        // MSAU_CONV_POOL keeps the rounded results (0 where nothing is stored: the zero padding of the pool)
        constexpr bool POOL_OK = EPI == EPI_POOL;
        static_assert(EPI == EPI_NONE || (CT <= 2 && !DOUT && STRIDE == 1 && UPS == 1), "fused epilogues: <= 2 channel tiles, one output");
        V4 keep[POOL_OK ? CT : 1][4];
        if constexpr (POOL_OK) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) keep[ct][pt][j] = (T)0.f;
        }
        if (ox0 + cwt * 16 + lr < d.Wout) {
            char* y = static_cast<char*>(d.y) + ((long long)(b * d.Hout + oyw)) * a.out_row + (long long)ox0 * a.out_px;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                if (lg * (CTT * 4) + (cty + ct) * 4 >= Cout) continue;
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) {
                    if (oyw + pt < d.Hout) {                       // scalar
                        char* yp = y + (unsigned)(lane_out + ct * 4 * ESZ + pt * a.out_row);
                        f32x4 v = acc[ct][pt] + bv[ct];
                        if (flags & (MSAU_CONV_MASK_A | MSAU_CONV_ADD | MSAU_CONV_ACCUM | MSAU_CONV_RELU_OUT | MSAU_CONV_MASK_B)) {
                            if (flags & MSAU_CONV_MASK_A) {
                                V4 m = *reinterpret_cast<const V4*>(yp + delta_ma);
#pragma unroll
                                for (int j = 0; j < 4; ++j) v[j] = ((float)m[j] > 0.f) ? v[j] : 0.f;
                            }
                            if (flags & MSAU_CONV_ADD) {
                                V4 r = *reinterpret_cast<const V4*>(yp + delta_add);
#pragma unroll
                                for (int j = 0; j < 4; ++j) v[j] += (float)r[j];
                            }
                            if (flags & MSAU_CONV_ACCUM) {
                                V4 r = *reinterpret_cast<const V4*>(yp);
#pragma unroll
                                for (int j = 0; j < 4; ++j) v[j] += (float)r[j];
                            }
                            if (flags & MSAU_CONV_RELU_OUT) {
#pragma unroll
                                for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                            }
                            if (flags & MSAU_CONV_MASK_B) {
                                V4 m = *reinterpret_cast<const V4*>(yp + delta_mb);
#pragma unroll
                                for (int j = 0; j < 4; ++j) v[j] = ((float)m[j] > 0.f) ? v[j] : 0.f;
                            }
                        }
                        V4 o;
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = (T)v[j];
                        *reinterpret_cast<V4*>(yp) = o;
                        if constexpr (POOL_OK) keep[ct][pt] = o;
                    }
                }
            }
        }
        // ---- MaxPool2d(2,2) of the zero-padded result as a further output (MSAU_CONV_POOL; model/model.py:158-160): the
        // tile origin is even, so a 2 x 2 window is two rows of this lane (pt 2pp, 2pp + 1) x this lane and its neighbour
        // lr ^ 1 (one DPP move per dword).  Even lanes compare in the order of msau_maxpool2x2_fwd (first maximum wins) on
        // the storage-rounded values and write the pooled pixel and, if asked for, the 1-byte positions.
        if constexpr (POOL_OK) {
            if (flags & MSAU_CONV_POOL) {
                const int Ho = (d.Hout + 1) >> 1, Wo = (d.Wout + 1) >> 1;
                const int col = ox0 + cwt * 16 + lr;
                constexpr int ND = (int)sizeof(V4) / 4;                    // dwords per 4 stored values
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp) {
                        V4 nb[2];
#pragma unroll
                        for (int r = 0; r < 2; ++r) {
                            typedef int dwords __attribute__((ext_vector_type(ND)));
                            dwords src = __builtin_bit_cast(dwords, keep[ct][2 * pp + r]), dst;
#pragma unroll
                            for (int w = 0; w < ND; ++w) dst[w] = __builtin_amdgcn_mov_dpp(src[w], 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
                            nb[r] = __builtin_bit_cast(V4, dst);
                        }
                        const int r0 = oyw + 2 * pp;
                        if (!(lr & 1) && r0 < d.Hout && col < d.Wout && lg * (CTT * 4) + (cty + ct) * 4 < Cout) {
                            V4 best;
                            unsigned idx = 0;
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                float bv_ = (float)keep[ct][2 * pp][j];
                                unsigned bi = 0;
                                const float c1 = (float)nb[0][j], c2 = (float)keep[ct][2 * pp + 1][j], c3 = (float)nb[1][j];
                                if (c1 > bv_) { bv_ = c1; bi = 1; }
                                if (c2 > bv_) { bv_ = c2; bi = 2; }
                                if (c3 > bv_) { bv_ = c3; bi = 3; }
                                best[j] = (T)bv_;
                                idx |= bi << (8 * j);
                            }
                            const long long e = (((long long)b * Ho + (r0 >> 1)) * Wo + (col >> 1)) * Cout + lg * (CTT * 4) + (cty + ct) * 4;
                            *reinterpret_cast<V4*>(static_cast<T*>(d.pool_y) + e) = best;
                            if (d.pool_idx) *reinterpret_cast<unsigned*>(d.pool_idx + e) = idx;
                        }
                    }
                }
            }
        }
        }
        // ---- LocalResponseNorm(size = Cout) of the result as a second output (MSAU_CONV_LRN; layers.py:145,161-162 after the
        // level-entry conv): y2 = y * (k + alpha/n * window sum of y^2)^-beta, from the storage-rounded y exactly as the
        // stand-alone msau_lrn_fwd reads it back.  The Cout channels of a pixel sit in RL lanes (lr, q = 0..RL-1), CT*4
        // consecutive channels each; window sums are differences of the exclusive prefix sum taken half the channels away,
        // i.e. RL/2 lanes away.  Outside the divergent store guards: every lane takes part in the shuffles.
        if constexpr (EPI == EPI_LRN) {
            static_assert(!SPLIT && !DUAL, "LRN needs all channels of a pixel in one workgroup");
            if (flags & MSAU_CONV_LRN) {
                const int RL = Cout == 8 ? 2 : 4;                          // lanes of a pixel that hold real channels
                const bool b075 = d.lrn_beta == 0.75f;
                char* y2 = static_cast<char*>(d.y2) + ((long long)(b * d.Hout + oyw)) * a.out_row + (long long)ox0 * a.out_px;
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) {
                    float x[CT * 4], P[CT * 4];
                    float run = 0.f;
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        const f32x4 v = acc[ct][pt] + bv[ct];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float r = (float)(T)v[j];
                            x[ct * 4 + j] = r;
                            run += r * r;
                            P[ct * 4 + j] = run;
                        }
                    }
                    float inc = run;
                    const float n1 = __shfl(inc, lane - 16);
                    if (lg >= 1) inc += n1;
                    const float n2 = __shfl(inc, lane - 32);
                    if (lg >= 2) inc += n2;
                    const float E = inc - run;
                    const float tot = __shfl(inc, lr + 16 * (RL - 1));
                    V4 o[CT];
#pragma unroll
                    for (int i = 0; i < CT * 4; ++i) {
                        const float X = E + (i ? P[i - 1] : 0.f);          // exclusive prefix at this lane's channel i
                        const float Xo = __shfl(X, lane ^ (8 * RL));
                        const float win = lg < RL / 2 ? Xo : tot - Xo;
                        const float dd = d.lrn_k + d.lrn_alpha_over_n * win;
                        float dnb;
                        if (b075) { const float r = __builtin_amdgcn_rsqf(dd); dnb = r * __builtin_amdgcn_sqrtf(r); }
                        else dnb = __expf(-d.lrn_beta * __logf(dd));
                        o[i >> 2][i & 3] = (T)(x[i] * dnb);
                    }
                    if (oyw + pt < d.Hout && ox0 + cwt * 16 + lr < d.Wout) {
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct)
                            if (lg * (CTT * 4) + (cty + ct) * 4 < Cout)
                                *reinterpret_cast<V4*>(y2 + (unsigned)(lane_out + ct * 4 * ESZ + pt * a.out_row)) = o[ct];
                    }
                }
            }
        }
        // ---- inference head on the end conv (MSAU_CONV_HEAD): the 16 channel rows of a pixel sit in the four lanes
        // (lr, q = 0..3); every lane gathers them, runs the shared softmax / first-max routine and writes its own
        // channels.  Outside the divergent store guards above so that all lanes take part in the shuffles.
        if constexpr (EPI == EPI_HEAD) {
            static_assert(CT == 1 && !DUAL && !DOUT, "the head gathers 16 channel rows of one tile");
            if (flags & MSAU_CONV_HEAD) {
                const int ncls = d.head_classes;
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) {
                    const f32x4 v = acc[0][pt] + bv[0];
                    float all[16];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float mine = (float)(T)v[j];                 // the storage-rounded logit, as written to y
#pragma unroll
                        for (int q = 0; q < 4; ++q) all[q * 4 + j] = __shfl(mine, q * 16 + lr);     // lanes of this wave
                    }
                    const int best = msau_head_softmax(all, ncls);
                    if (oyw + pt < d.Hout && ox0 + cwt * 16 + lr < d.Wout) {
                        const long long pix = ((long long)b * d.Hout + oyw + pt) * d.Wout + ox0 + cwt * 16 + lr;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float pr = all[j];
#pragma unroll
                            for (int q = 1; q < 4; ++q) pr = lg == q ? all[q * 4 + j] : pr;
                            if (lg * 4 + j < ncls) d.head_probs[pix * ncls + lg * 4 + j] = pr;
                        }
                        if (lg == 0) d.head_argmax[pix] = (unsigned char)best;
                    }
                }
            }
        }
    }
#ifdef MSAU_STAMPS
    if (a.stamps && threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(4);
        unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        a.stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 5] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
}


// ---- many input channels -> one 16-row tile, 3x3 (the first conv of a BERT-embedding chargrid: 768 -> 8, BASELINE configs[3]).
// The packed image is chunked ([chunk][row][k], 64 channels per chunk: what fits the generic kernel's LDS budget) and the
// generic kernel took the launch at 1.4 TB/s.  Here a persistent workgroup walks the chunks of its 16 x 16 tile with the
// accumulators in registers: per chunk one 18 x 18 x 64 input tile and one 16 x 576 weight block go through LDS, both
// prefetched into registers while the previous chunk's 72 MFMAs per wave run.
// Round 4: the same walk for the data gradient of the level-3 entry conv (64 -> 32, dilation 8: a 32 x 32 halo tile of 64 channels
// plus the weights is 184 KB -- no single-chunk instance fits, and the generic kernel took the launch at 0.2 TB/s): C8CH 8-channel
// groups per chunk, CT channel tiles, dilation DIL.
template <typename T, int C8CH = 8, int CT = 1, int DIL = 1>
__global__ __launch_bounds__(256) void conv_chunked_kernel(const LeanArgs a, const int nchunks) {
    using Cfg = LeanCfg<T, C8CH, CT, 3, false, DIL>;
    typedef typename Vec8<T>::type V8;
    typedef typename Vec4<T>::type V4;
    constexpr int ESZ = Cfg::ESZ, TI = Cfg::TIW, PS = Cfg::PS, NKS = Cfg::NKS, WS = lds_wrow_stride(NKS, ESZ);
    constexpr int NPIX = Cfg::NPIX, NITX = (NPIX * C8CH + 255) / 256, WG8 = NKS * 4, NITW = (CT * 16 * WG8 + 255) / 256;
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* lds_w = smem + Cfg::IN_BYTES;
    const msau_conv_desc& d = a.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    int koff[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int G = ks * 4 + lg, tap = G / C8CH, cg = G - tap * C8CH;
        const int ky = tap / 3, kx = tap - ky * 3;
        koff[ks] = G < 9 * C8CH ? (ky * DIL * TI + kx * DIL) * PS + cg * 8 * ESZ : 0;
    }
    const unsigned char* pixp = smem + ((wave * 4) * TI + lr) * PS;
    f32x4 bv[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        bv[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (d.bias && lg * (CT * 4) + ct * 4 < d.Cout) bv[ct] = *reinterpret_cast<const f32x4*>(d.bias + lg * (CT * 4) + ct * 4);
    }
    const T* wp = static_cast<const T*>(d.wpack);
    V8 xr[NITX], wr[NITW];
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int t1 = a.tiles_x > 1 ? __umulhi((unsigned)tile, a.mag_tx) : tile;
        const int ox0 = (tile - t1 * a.tiles_x) * 16;
        const int b = a.tiles_y > 1 ? __umulhi((unsigned)t1, a.mag_ty) : t1;
        const int oy0 = (t1 - b * a.tiles_y) * 16;
        const int vy0 = oy0 - d.pad_t, vx0 = ox0 - d.pad_l;
        const char* xb = static_cast<const char*>(d.x1) + (long long)b * d.Hin * a.in_row1;
        auto issue = [&](int ch) {
#pragma unroll
            for (int it = 0; it < NITX; ++it) {
                const int idx = tid + it * 256;
                const int pix = idx / C8CH, cg = idx - pix * C8CH;
                const int iy = pix / TI, ix = pix - iy * TI;
                const int vy = vy0 + iy, vx = vx0 + ix;
                xr[it] = zero8<T>();
                if (idx < NPIX * C8CH && (unsigned)vy < (unsigned)d.Hin && (unsigned)vx < (unsigned)d.Win)
                    xr[it] = *reinterpret_cast<const V8*>(xb + (unsigned)(vy * a.in_row1 + vx * a.in_px1 + (ch * C8CH * 8 + cg * 8) * ESZ));
            }
#pragma unroll
            for (int it = 0; it < NITW; ++it) {
                const int idx = tid + it * 256;
                const int r = idx / WG8, g8 = idx - r * WG8;
                wr[it] = zero8<T>();
                if (idx < CT * 16 * WG8) wr[it] = load8<T>(wp + ((size_t)ch * CT * 16 + r) * a.kchunk + g8 * 8);
            }
        };
        f32x4 acc[CT][4];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
        issue(0);
        for (int ch = 0; ch < nchunks; ++ch) {
            __syncthreads();                                      // the previous chunk's fragments have been read
#pragma unroll
            for (int it = 0; it < NITX; ++it) {
                const int idx = tid + it * 256;
                if (idx < NPIX * C8CH) *reinterpret_cast<V8*>(smem + (idx / C8CH) * PS + (idx % C8CH) * 8 * ESZ) = xr[it];
            }
#pragma unroll
            for (int it = 0; it < NITW; ++it) {
                const int idx = tid + it * 256;
                const int r = idx / WG8, g8 = idx - r * WG8;
                if (idx < CT * 16 * WG8) *reinterpret_cast<V8*>(lds_w + r * WS + g8 * 8 * ESZ) = wr[it];
            }
            __syncthreads();
            if (ch + 1 < nchunks) issue(ch + 1);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const unsigned char* p = pixp + koff[ks];
                V8 bfrag[4];
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) bfrag[pt] = *reinterpret_cast<const V8*>(p + pt * TI * PS);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const V8 af = *reinterpret_cast<const V8*>(lds_w + (ct * 16 + lr) * WS + (ks * 32 + lg * 8) * ESZ);
#pragma unroll
                    for (int pt = 0; pt < 4; ++pt) acc[ct][pt] = mma8(af, bfrag[pt], acc[ct][pt]);
                }
            }
        }
        const int oyw = oy0 + wave * 4;
        if (ox0 + lr < d.Wout) {
            char* y = static_cast<char*>(d.y) + ((long long)(b * d.Hout + oyw) * d.Wout + ox0 + lr) * a.out_px;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                if (lg * (CT * 4) + ct * 4 >= d.Cout) continue;
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) {
                    if (oyw + pt < d.Hout) {
                        f32x4 v = acc[ct][pt] + bv[ct];
                        if (d.flags & MSAU_CONV_RELU_OUT) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                        }
                        V4 o;
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = (T)v[j];
                        *reinterpret_cast<V4*>(y + (long long)pt * a.out_row + (lg * (CT * 4) + ct * 4) * ESZ) = o;
                    }
                }
            }
        }
    }
}

template <typename T, int CIN8, int CT, int KS, bool DUAL, int DIL = 1, int WGW = 1, bool DOUT = false, bool SPLIT = false,
          int STRIDE = 1, int UPS = 1, int EPI = EPI_NONE, int EOP = -1, bool IDS = false>
int launch_lean_e(hipStream_t s, const LeanArgs& a0) {
    using Cfg = LeanCfg<T, CIN8, CT, KS, DUAL, DIL, WGW, STRIDE>;
    if (Cfg::LDS + 256 > MSAU_LDS_LIMIT) return 0;          // does not fit: the generic kernel takes the launch
    LeanArgs a = a0;
    a.tiles_x = cdiv(a.d.Wout, 16 * WGW);
    a.ntiles = a.d.B * a.tiles_x * a.tiles_y;
    a.mag_tx = (unsigned)((0x100000000ull + a.tiles_x - 1) / a.tiles_x);
    static bool attr_set = false;
    if (!attr_set && Cfg::LDS > 60 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_lean_kernel<T, CIN8, CT, KS, DUAL, DIL, WGW, DOUT, SPLIT, STRIDE, UPS, EPI, EOP, IDS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, MSAU_LDS_LIMIT);
        if (e != hipSuccess) return msau_set_error(MSAU_ERR_HIP, "conv_lean: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    int per_cu = MSAU_LDS_LIMIT / (Cfg::LDS + 256);
    static const int percu_max = std::getenv("MSAU_LEAN_PERCU") ? atoi(std::getenv("MSAU_LEAN_PERCU")) : 4;      // persistent workgroups per CU (x 256 CUs): 8 -> 4 +0.6 %, 3 -1.3 %
    per_cu = per_cu < 1 ? 1 : (per_cu > percu_max / WGW ? (percu_max / WGW < 1 ? 1 : percu_max / WGW) : per_cu);
    int grid = 256 * per_cu;
    if (grid > a.ntiles) grid = a.ntiles;
    static const bool xcd_off = std::getenv("MSAU_XCD") && std::getenv("MSAU_XCD")[0] == '0';
    a.per_xcd = 0;
    if (!xcd_off && grid >= 64) {
        grid &= ~7;
        a.per_xcd = cdiv(a.ntiles, 8);
    }
    hipLaunchKernelGGL((conv_lean_kernel<T, CIN8, CT, KS, DUAL, DIL, WGW, DOUT, SPLIT, STRIDE, UPS, EPI, EOP, IDS>), dim3(grid, SPLIT ? a.ct_total : 1), dim3(256 * WGW), Cfg::LDS, s, a);
    MSAU_CHECK_LAUNCH("conv_lean_kernel");
    return 1;
}

// picks the operand-specialised instance for the 8- / 16-channel layers (the combinations the training step launches:
// none, MASK_B, ACCUM|MASK_B, MASK_A|ADD, ADD); everything else keeps run-time flags
template <typename T, int CIN8, int CT, int KS, bool DUAL, int DIL = 1, int WGW = 1, bool DOUT = false, bool SPLIT = false,
          int STRIDE = 1, int UPS = 1, int EPI = EPI_NONE>
int launch_lean(hipStream_t s, const LeanArgs& a) {
    if constexpr (CIN8 <= 2 && !DOUT && EPI != EPI_HEAD) {
        switch (a.d.flags & kOperandBits) {
#define EOP_CASE(V) case V: return launch_lean_e<T, CIN8, CT, KS, DUAL, DIL, WGW, DOUT, SPLIT, STRIDE, UPS, EPI, V>(s, a);
            EOP_CASE(0) EOP_CASE(2) EOP_CASE(3) EOP_CASE(6) EOP_CASE(8) EOP_CASE(12) EOP_CASE(20) EOP_CASE(32) EOP_CASE(40)
#undef EOP_CASE
            default: break;
        }
    }
    return launch_lean_e<T, CIN8, CT, KS, DUAL, DIL, WGW, DOUT, SPLIT, STRIDE, UPS, EPI, -1>(s, a);
}

template <typename T, int CIN8, int KS, bool DUAL>
int lean_ct(hipStream_t s, const LeanArgs& a, int CT) {
    if constexpr (sizeof(T) == 2 && (CIN8 == 1 || (DUAL && CIN8 == 2)) && KS == 3) {      // 4x4: measured, no gain
        // 16 x 32 output tile per 512-thread workgroup for the 8/16-channel layers: same work per wave, but a halo
        // row is 544 B (5-6 lines for 4.25 of payload) instead of 288 B (4 lines for 2.25)
        if (CT == 1 && a.d.Wout >= 64 && (int64_t)a.d.B * a.tiles_y * cdiv(a.d.Wout, 32) >= 512)
            return launch_lean<T, CIN8, 1, KS, DUAL, 1, 2>(s, a);
    }
    if (CT == 1) return launch_lean<T, CIN8, 1, KS, DUAL>(s, a);
    if (CT == 2) return launch_lean<T, CIN8, 2, KS, DUAL>(s, a);
    if constexpr (CIN8 >= 4 && KS != 4) {
        if (CT == 4) return launch_lean<T, CIN8, 4, KS, DUAL>(s, a);
    }
    return 0;
}

template <typename T, int KS>
int lean_cin(hipStream_t s, const LeanArgs& a, int cin8, bool dual, int CT) {
    if (!dual) {
        switch (cin8) {
            case 1: return lean_ct<T, 1, KS, false>(s, a, CT);
            case 2: return lean_ct<T, 2, KS, false>(s, a, CT);
            case 4: return lean_ct<T, 4, KS, false>(s, a, CT);
            case 8: return lean_ct<T, 8, KS, false>(s, a, CT);
            default: return 0;
        }
    }
    switch (cin8) {
        case 2: return lean_ct<T, 2, KS, true>(s, a, CT);
        case 4: return lean_ct<T, 4, KS, true>(s, a, CT);
        case 8: return lean_ct<T, 8, KS, true>(s, a, CT);
        case 16: if constexpr (KS == 1) return lean_ct<T, 16, KS, true>(s, a, CT); else return 0;
        default: return 0;
    }
}

// dilated 3x3 (the level-entry convs of the encoder and their data gradients): single source only
template <typename T, int DIL>
int lean_dil(hipStream_t s, const LeanArgs& a, int cin8, int CT) {
#define LD_CASE(C8, CTV) if (cin8 == C8 && CT == CTV) return launch_lean<T, C8, CTV, 3, false, DIL>(s, a);
    LD_CASE(1, 1) LD_CASE(2, 1) LD_CASE(2, 2) LD_CASE(4, 1) LD_CASE(4, 2) LD_CASE(4, 4) LD_CASE(8, 2)
#undef LD_CASE
    return 0;
}

}  // namespace

static bool lean_split_wanted(const msau_conv_desc* d, int CT);

// LeanCfg::LDS at run time (same formula), for the applicability queries
static int lean_lds_bytes(int esz, int cin8, int CT, int KS, int dil, int wgw, int stride, bool dual) {
    const int ti = 15 * stride + 1 + (KS - 1) * dil, tiw = (16 * wgw - 1) * stride + 1 + (KS - 1) * dil;
    const int psraw = cin8 * 8 * esz, ps = ((psraw / 16) % 2 == 0) ? psraw + 16 : psraw;
    const int nch = dual ? 2 : 1, c8h = cin8 / nch, ng = KS * KS * c8h, nksh = (ng + 3) / 4, nks = nch * nksh;
    const bool wreg = CT * nks <= 8;
    const int ws = nksh * 32 * esz + 16;
    const int in_bytes = ((ti * tiw * ps + 15) / 16) * 16;
    return in_bytes + (wreg ? 0 : nch * CT * 16 * ws);
}

// transposed conv (ups = 2) and its data gradient (stride = 2), 3x3, single source: (C1/8, CT) per level
static bool lean_strided_shape(const msau_conv_desc* d, int nchunks, int CT) {
    if (d->KH != 3 || d->KW != 3 || d->dil != 1 || d->C2 || nchunks != 1 || d->stride * d->ups != 2) return false;
    if (d->flags & (MSAU_CONV_DOUT | MSAU_CONV_HEAD | MSAU_CONV_RELU_IN)) return false;
    if (d->pad_t < 0 || d->pad_l < 0 || d->pad_t > 2 || d->pad_l > 2) return false;
    if ((int64_t)d->B * cdiv(d->Hout, 16) * cdiv(d->Wout, 16) < 64) return false;
    const int c8 = d->C1 / 8;
    if (d->ups == 2) return (c8 == 8 && CT == 2) || (c8 == 4 && CT == 1) || (c8 == 2 && CT == 1);
    return (c8 == 4 && CT == 4) || (c8 == 2 && CT == 2) || (c8 == 1 && CT == 1);
}
template <typename T>
int lean_strided(hipStream_t s, const LeanArgs& a, int c8, int CT, bool ups) {
#define ST_CASE(C8, CTV, ST, UP) if (c8 == C8 && CT == CTV) return launch_lean<T, C8, CTV, 3, false, 1, 1, false, false, ST, UP>(s, a);
    if (ups) { ST_CASE(8, 2, 1, 2) ST_CASE(4, 1, 1, 2) ST_CASE(2, 1, 1, 2) }
    else { ST_CASE(4, 4, 2, 1) ST_CASE(2, 2, 2, 1) ST_CASE(1, 1, 2, 1) }
#undef ST_CASE
    return 0;
}

// Returns 1 if a lean instance handled the launch, 0 if the caller must use the generic kernel,
// < 0 on error.  `kchunk` / `nchunks` / `CT` come from the generic geometry (same packed image).
int msau_conv_lean_applicable(int dtype, const msau_conv_desc* d, int nchunks, int CT) {
    if (d->flags & MSAU_CONV_ELU) return 0;                    // (ELU epilogues: the generic kernel only)
    if (d->stride * d->ups == 2) {
        const int esz = dtype == MSAU_F32 ? 4 : 2;
        if ((int64_t)d->Hin * d->Win * d->C1 * esz >= (1ll << 31) || (int64_t)d->Wout * d->Cout * esz * 20 >= (1ll << 31)) return 0;
        if (!lean_strided_shape(d, nchunks, CT)) return 0;
        return lean_lds_bytes(esz, d->C1 / 8, CT, 3, 1, 1, d->stride, false) + 256 <= MSAU_LDS_LIMIT;
    }
    if (d->stride != 1 || d->ups != 1 || d->KH != d->KW || (d->KH != 1 && d->KH != 3 && d->KH != 4)) return 0;
    if (d->dil != 1) {
        if (d->KH != 3 || d->C2 || (d->dil != 2 && d->dil != 4 && d->dil != 8)) return 0;
        const int c8 = d->C1 / 8;
        const bool ok = (c8 == 1 && CT == 1) || (c8 == 2 && CT <= 2) || (c8 == 4 && (CT == 1 || CT == 2 || CT == 4)) || (c8 == 8 && CT == 2);
        if (!ok) return 0;
    }
    if (nchunks != (d->C2 ? 2 : 1)) return 0;                // one chunk per source (same packed image as conv.hip)
    if (d->Hin != d->Hout || d->Win != d->Wout) return 0;
    if (d->pad_t < 0 || d->pad_l < 0 || d->pad_t > (d->KH - 1) * d->dil || d->pad_l > (d->KW - 1) * d->dil) return 0;
    if (CT > 4 || (CT == 4 && (d->C1 + d->C2) < 32)) return 0;
    if (d->KH == 4 && ((d->C1 + d->C2) != 8 || d->C2)) return 0;           // only the 8-channel end conv / its data gradient
    if ((int64_t)d->B * cdiv(d->Hout, 16) * cdiv(d->Wout, 16) < 64) return 0;
    const bool dual = d->C2 != 0;
    if (dual && d->C1 != d->C2) return 0;
    const int cin8 = (d->C1 + d->C2) / 8;
    const int esz = dtype == MSAU_F32 ? 4 : 2;
    if ((int64_t)d->Hin * d->Win * (d->C1 > d->C2 ? d->C1 : d->C2) * esz >= (1ll << 31)) return 0;   // 32-bit lane offsets
    if ((int64_t)d->Wout * d->Cout * esz * 20 >= (1ll << 31)) return 0;
    if (cin8 != 1 && cin8 != 2 && cin8 != 4 && cin8 != 8 && !(cin8 == 16 && dual && d->KH == 1)) return 0;
    if (dual && cin8 < 2) return 0;
    // the instance that would take the launch must fit the LDS (fp32 storage doubles every tile)
    const bool split = lean_split_wanted(d, CT);
    if (lean_lds_bytes(esz, cin8, split ? 1 : CT, d->KH, d->dil, 1, 1, dual) + 256 > MSAU_LDS_LIMIT) return 0;
    return 1;
}

// two-output data gradient (MSAU_CONV_DOUT): g [C] -> (dx1 [C], dx2 [C]) with C = CT*8 in {8, 16, 32}
int msau_conv_lean_dout_capable(int dtype, const msau_conv_desc* d, int nchunks, int CT) {
    if (!msau_conv_lean_applicable(dtype, d, nchunks, CT)) return 0;
    return d->C2 == 0 && d->dil == 1 && d->stride == 1 && d->ups == 1 && (d->KH == 1 || d->KH == 3) && (CT == 1 || CT == 2 || CT == 4) &&
           d->Cout == CT * 16 && d->C1 == CT * 8;
}

// ---- fused epilogues: the instances that exist (the featRoot-8 / featRoot-16 shapes of the reference's configurations);
// everything else runs the stand-alone LRN / pool launch.  One table for the capability queries and the dispatch.
//   LRN : level-entry convs  8 -> 8 (3x3), 8 -> 16 (dilation 2), 16 -> 16, 16 -> 32 (dilation 4 / 2), 64 -> 8 (the net's first conv)
//   POOL: coupling 1x1 convs over concat (8+8 -> 8, 16+16 -> 16, 32+32 -> 32) and the split 32 -> 32 3x3 (stage 0, level 2)
static bool lean_wide_tile(const msau_conv_desc* d, int tiles_y) {          // the 16 x 32 tile of lean_ct
    return d->Wout >= 64 && (int64_t)d->B * tiles_y * cdiv(d->Wout, 32) >= 512;
}
static int lean_epi_case(int dtype, const msau_conv_desc* d, int CT, int epi) {
    const int c8 = (d->C1 + d->C2) / 8, k = d->KH;
    const bool dual = d->C2 != 0, split = lean_split_wanted(d, CT);
    if (epi == EPI_LRN && !dual && !split && k == 3) {
        if (c8 == 1 && CT == 1 && d->dil == 1 && d->Cout == 8) return dtype == MSAU_BF16 && lean_wide_tile(d, cdiv(d->Hout, 16)) ? 1 : 2;
        if (c8 == 1 && CT == 1 && d->dil == 2 && d->Cout == 16) return 3;
        if (c8 == 2 && CT == 2 && d->dil == 4 && d->Cout == 32) return 4;
        if (c8 == 2 && CT == 1 && d->dil == 1 && d->Cout == 16) return 5;
        if (c8 == 2 && CT == 2 && d->dil == 2 && d->Cout == 32) return 6;
        // (the level-3 entry conv, 32 -> 64 channels at dilation 8 with four channel tiles, had an instance in round 4: 15.1 us against
        //  7.8 us for the conv + 1.7 us for the 64-channel LRN launch, 3.086 -> 3.077 ms without it -- removed, profiles/HISTORY_r03_r04.md)
        // (the net's first conv, 64 one-hot channels -> featRoot 8, had an instance too: 9 us SLOWER per step than the conv + the 8.4 us
        //  stand-alone LRN launch -- removed in round 4, profiles/HISTORY_r03_r04.md)
    }
    if (epi == EPI_POOL && d->dil == 1) {
        if (dual && k == 1 && !split) {
            if (c8 == 2 && CT == 1) return 10;
            if (c8 == 4 && CT == 1) return 11;
            if (c8 == 8 && CT == 2) return 12;
        }
        if (!dual && k == 3 && split && c8 == 4) return 13;
    }
    return 0;
}
template <typename T>
int lean_epi(hipStream_t s, const LeanArgs& a, int which) {
    switch (which) {
        case 1: if constexpr (sizeof(T) == 2) return launch_lean<T, 1, 1, 3, false, 1, 2, false, false, 1, 1, EPI_LRN>(s, a); else return 0;
        case 2: return launch_lean<T, 1, 1, 3, false, 1, 1, false, false, 1, 1, EPI_LRN>(s, a);
        case 3: return launch_lean<T, 1, 1, 3, false, 2, 1, false, false, 1, 1, EPI_LRN>(s, a);
        case 4: return launch_lean<T, 2, 2, 3, false, 4, 1, false, false, 1, 1, EPI_LRN>(s, a);
        case 5: return launch_lean<T, 2, 1, 3, false, 1, 1, false, false, 1, 1, EPI_LRN>(s, a);
        case 6: return launch_lean<T, 2, 2, 3, false, 2, 1, false, false, 1, 1, EPI_LRN>(s, a);
        case 10: return launch_lean<T, 2, 1, 1, true, 1, 1, false, false, 1, 1, EPI_POOL>(s, a);
        case 11: return launch_lean<T, 4, 1, 1, true, 1, 1, false, false, 1, 1, EPI_POOL>(s, a);
        case 12: return launch_lean<T, 8, 2, 1, true, 1, 1, false, false, 1, 1, EPI_POOL>(s, a);
        case 13: return launch_lean<T, 4, 1, 3, false, 1, 1, false, true, 1, 1, EPI_POOL>(s, a);
        default: return 0;
    }
}

// second output LRN(y) (MSAU_CONV_LRN): all Cout channels of a pixel in one workgroup, at most two 16-row tiles
int msau_conv_lean_lrn_capable(int dtype, const msau_conv_desc* d, int nchunks, int CT) {
    if (!msau_conv_lean_applicable(dtype, d, nchunks, CT)) return 0;
    if (d->stride != 1 || d->ups != 1 || (d->flags & (MSAU_CONV_DOUT | MSAU_CONV_HEAD | MSAU_CONV_POOL))) return 0;
    return lean_epi_case(dtype, d, CT, EPI_LRN) != 0;
}

// pooled output (MSAU_CONV_POOL): a 16 x 16 tile holds whole 2 x 2 windows
int msau_conv_lean_pool_capable(int dtype, const msau_conv_desc* d, int nchunks, int CT) {
    if (!msau_conv_lean_applicable(dtype, d, nchunks, CT)) return 0;
    if (d->stride != 1 || d->ups != 1 || (d->flags & (MSAU_CONV_DOUT | MSAU_CONV_HEAD | MSAU_CONV_LRN))) return 0;
    return lean_epi_case(dtype, d, CT, EPI_POOL) != 0;
}

template <typename T>
int lean_dout(hipStream_t s, const LeanArgs& a, int KS, int CT) {
#define DO_CASE(K, C) if (KS == K && CT == C) return launch_lean<T, C, C, K, false, 1, 1, true>(s, a);
    DO_CASE(1, 1) DO_CASE(1, 2) DO_CASE(1, 4) DO_CASE(3, 1) DO_CASE(3, 2) DO_CASE(3, 4)
#undef DO_CASE
    return 0;
}

// channel-split instances for small images: (CIN8, KS) in {4, 8} x {1, 3}
static int lean_split_tiles() {
    static const int v = std::getenv("MSAU_SPLIT_TILES") ? atoi(std::getenv("MSAU_SPLIT_TILES")) : 512;      // measured: 0 -> 5.04, 256 -> 4.90, 512 -> 4.83, 1024 -> 4.84 ms/step
    return v;
}
static bool lean_split_wanted(const msau_conv_desc* d, int CT) {
    const int cin8 = d->C1 / 8;
    return d->C2 == 0 && d->dil == 1 && (d->KH == 1 || d->KH == 3) && (CT == 2 || CT == 4) && (cin8 == 4 || cin8 == 8) &&
           !(d->flags & (MSAU_CONV_DOUT | MSAU_CONV_HEAD)) &&
           (int64_t)d->B * cdiv(d->Hout, 16) * cdiv(d->Wout, 16) < lean_split_tiles();
}
template <typename T>
int lean_split(hipStream_t s, const LeanArgs& a, int cin8, int KS) {
#define SP_CASE(C8, K) if (cin8 == C8 && KS == K) return launch_lean<T, C8, 1, K, false, 1, 1, false, true>(s, a);
    SP_CASE(4, 1) SP_CASE(4, 3) SP_CASE(8, 1) SP_CASE(8, 3)
#undef SP_CASE
    return 0;
}

// 1 if the lean instance that takes this launch implements MSAU_CONV_HEAD (the 4x4 end conv, one 16-row tile)
// id-mask input (MSAU_CONV_IDS): 64 one-hot channels -> one 16-row tile, 3x3, no other source / epilogue operand
int msau_conv_lean_ids_capable(int dtype, const msau_conv_desc* d, int nchunks, int CT) {
    msau_conv_desc e = *d;
    e.flags &= ~MSAU_CONV_IDS;
    return msau_conv_lean_applicable(dtype, &e, nchunks, CT) && d->C1 == 64 && d->C2 == 0 && CT == 1 && d->KH == 3 && d->dil == 1 &&
           d->stride == 1 && d->ups == 1 && !(d->flags & ~(MSAU_CONV_IDS | MSAU_CONV_RELU_OUT));
}

int msau_conv_lean_head_capable(int dtype, const msau_conv_desc* d, int nchunks, int CT) {
    return msau_conv_lean_applicable(dtype, d, nchunks, CT) && d->KH == 4 && CT == 1 && d->dil == 1;
}

// 1 = handled by conv_chunked_kernel, 0 = not this shape.  (cch / kchunk / nchunks: the generic geometry = the packed image.)
//   variant 1: many 64-channel chunks -> one 16-row tile, 3x3 (the 768 -> 8 first conv of cfg 4)
//   variant 2: two 32-channel chunks -> two 16-row tiles, 3x3 dilation 8, bf16 (the data gradient of the level-3 entry conv)
static int chunked_variant(int dtype, const msau_conv_desc* d, int cch, int nchunks, int CT) {
    static const bool off = std::getenv("MSAU_CONV_CHUNKED") && std::getenv("MSAU_CONV_CHUNKED")[0] == '0';
    if (off || nchunks < 2 || d->C2 || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->ups != 1) return 0;
    if (d->Hin != d->Hout || d->Win != d->Wout || d->pad_t < 0 || d->pad_l < 0 || d->pad_t > 2 * d->dil || d->pad_l > 2 * d->dil) return 0;
    if (d->flags & ~MSAU_CONV_RELU_OUT) return 0;
    const int esz = dtype == MSAU_F32 ? 4 : 2;
    if ((int64_t)d->Hin * d->Win * d->C1 * esz >= (1ll << 31)) return 0;              // 32-bit offsets inside an image
    const int64_t ntiles = (int64_t)d->B * cdiv(d->Wout, 16) * cdiv(d->Hout, 16);
    if (!(ntiles >= 64 && ntiles < (1 << 20) && cdiv(d->Wout, 16) < 4096 && cdiv(d->Hout, 16) < 4096)) return 0;
    if (cch == 64 && CT == 1 && d->dil == 1) return 1;
    if (cch == 32 && CT == 2 && d->dil == 8 && nchunks == 2 && dtype == MSAU_BF16) return 2;
    return 0;
}
int msau_conv_chunked_capable(int dtype, const msau_conv_desc* d, int cch, int nchunks, int CT) {
    if (d->flags & MSAU_CONV_ELU) return 0;
    return chunked_variant(dtype, d, cch, nchunks, CT) != 0;
}

template <typename T, int C8CH, int CT, int DIL>
static int launch_chunked(hipStream_t s, const LeanArgs& a, int nchunks) {
    using Cfg = LeanCfg<T, C8CH, CT, 3, false, DIL>;
    constexpr int lds = Cfg::IN_BYTES + CT * 16 * lds_wrow_stride(Cfg::NKS, Cfg::ESZ);
    static_assert(lds + 256 <= MSAU_LDS_LIMIT, "chunked conv: tile + weight chunk must fit the LDS");
    static bool attr_set = false;
    if (!attr_set && lds > 60 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_chunked_kernel<T, C8CH, CT, DIL>), hipFuncAttributeMaxDynamicSharedMemorySize, MSAU_LDS_LIMIT);
        if (e != hipSuccess) return msau_set_error(MSAU_ERR_HIP, "conv_chunked: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    int per_cu = MSAU_LDS_LIMIT / (lds + 256);
    per_cu = per_cu < 1 ? 1 : (per_cu > 3 ? 3 : per_cu);
    int grid = 256 * per_cu;
    if (grid > a.ntiles) grid = a.ntiles;
    hipLaunchKernelGGL((conv_chunked_kernel<T, C8CH, CT, DIL>), dim3(grid), dim3(256), lds, s, a, nchunks);
    MSAU_CHECK_LAUNCH("conv_chunked_kernel");
    return 1;
}

int msau_conv_chunked_try(hipStream_t s, int dtype, const msau_conv_desc* d, int cch, int kchunk, int nchunks, int CT) {
    const int variant = chunked_variant(dtype, d, cch, nchunks, CT);
    if (!variant) return 0;
    const int esz = dtype == MSAU_F32 ? 4 : 2;
    LeanArgs a;
    a.d = *d;
    a.kchunk = kchunk;
    a.in_px1 = d->C1 * esz; a.in_row1 = d->Win * a.in_px1; a.in_px2 = a.in_row2 = 0;
    a.out_px = d->Cout * esz; a.out_row = d->Wout * a.out_px;
    a.tiles_x = cdiv(d->Wout, 16); a.tiles_y = cdiv(d->Hout, 16);
    a.ntiles = d->B * a.tiles_x * a.tiles_y;
    a.mag_tx = (unsigned)((0x100000000ull + a.tiles_x - 1) / a.tiles_x);
    a.mag_ty = (unsigned)((0x100000000ull + a.tiles_y - 1) / a.tiles_y);
    a.per_xcd = 0; a.ct_total = CT;
#ifdef MSAU_STAMPS
    a.stamps = nullptr;
#endif
    if (variant == 2) return launch_chunked<bf16_t, 4, 2, 8>(s, a, nchunks);
    return dtype == MSAU_F32 ? launch_chunked<float, 8, 1, 1>(s, a, nchunks) : launch_chunked<bf16_t, 8, 1, 1>(s, a, nchunks);
}

int msau_conv_lean_try(hipStream_t s, int dtype, const msau_conv_desc* d, int kchunk, int nchunks, int CT) {
    if (!msau_conv_lean_applicable(dtype, d, nchunks, CT)) return 0;
    const bool dual = d->C2 != 0;
    const int cin8 = (d->C1 + d->C2) / 8;
    const int esz = dtype == MSAU_F32 ? 4 : 2;
    LeanArgs a;
    a.d = *d;
    a.kchunk = kchunk;
    a.in_px1 = d->C1 * esz; a.in_px2 = d->C2 * esz;
    a.in_row1 = d->Win * a.in_px1; a.in_row2 = d->Win * a.in_px2;
    const bool dout = d->flags & MSAU_CONV_DOUT;
    a.out_px = (dout ? d->Cout / 2 : d->Cout) * esz; a.out_row = d->Wout * a.out_px;
    a.tiles_x = cdiv(d->Wout, 16); a.tiles_y = cdiv(d->Hout, 16);
    a.ntiles = d->B * a.tiles_x * a.tiles_y;
    if (a.ntiles >= (1 << 20) || a.tiles_x >= 4096 || a.tiles_y >= 4096) return 0;
    a.mag_tx = (unsigned)((0x100000000ull + a.tiles_x - 1) / a.tiles_x);
    a.mag_ty = (unsigned)((0x100000000ull + a.tiles_y - 1) / a.tiles_y);
    a.ct_total = CT;
#ifdef MSAU_STAMPS
    a.stamps = std::getenv("MSAU_STAMP_PTR") ? reinterpret_cast<unsigned long long*>(strtoull(std::getenv("MSAU_STAMP_PTR"), nullptr, 0)) : nullptr;
#endif
    if (d->stride * d->ups == 2)
        return dtype == MSAU_F32 ? lean_strided<float>(s, a, cin8, CT, d->ups == 2) : lean_strided<bf16_t>(s, a, cin8, CT, d->ups == 2);
    if (d->flags & MSAU_CONV_IDS) {                              // id-mask input: the 64 -> 8/16 3x3 first conv (ids_capable)
        if (!msau_conv_lean_ids_capable(dtype, d, nchunks, CT)) return 0;
        return dtype == MSAU_F32 ? launch_lean_e<float, 8, 1, 3, false, 1, 1, false, false, 1, 1, EPI_NONE, -1, true>(s, a)
                                 : launch_lean_e<bf16_t, 8, 1, 3, false, 1, 1, false, false, 1, 1, EPI_NONE, -1, true>(s, a);
    }
    if (d->flags & MSAU_CONV_HEAD) {                             // forward-only: the 8-channel 4x4 end conv (head_capable)
        if (!msau_conv_lean_head_capable(dtype, d, nchunks, CT)) return 0;
        return dtype == MSAU_F32 ? launch_lean<float, 1, 1, 4, false, 1, 1, false, false, 1, 1, EPI_HEAD>(s, a)
                                 : launch_lean<bf16_t, 1, 1, 4, false, 1, 1, false, false, 1, 1, EPI_HEAD>(s, a);
    }
    if (d->flags & (MSAU_CONV_LRN | MSAU_CONV_POOL)) {
        const int which = lean_epi_case(dtype, d, CT, (d->flags & MSAU_CONV_LRN) ? EPI_LRN : EPI_POOL);
        if (!which) return msau_set_error(MSAU_ERR_ARG, "conv_lean: no instance with this fused epilogue (msau_conv2d_launch_info)");
        return dtype == MSAU_F32 ? lean_epi<float>(s, a, which) : lean_epi<bf16_t>(s, a, which);
    }
    if (lean_split_wanted(d, CT))
        return dtype == MSAU_F32 ? lean_split<float>(s, a, cin8, d->KH) : lean_split<bf16_t>(s, a, cin8, d->KH);
    if (dout) {
        if (!msau_conv_lean_dout_capable(dtype, d, nchunks, CT)) return 0;
        return dtype == MSAU_F32 ? lean_dout<float>(s, a, d->KH, CT) : lean_dout<bf16_t>(s, a, d->KH, CT);
    }
    if (d->dil == 2) return dtype == MSAU_F32 ? lean_dil<float, 2>(s, a, cin8, CT) : lean_dil<bf16_t, 2>(s, a, cin8, CT);
    if (d->dil == 4) return dtype == MSAU_F32 ? lean_dil<float, 4>(s, a, cin8, CT) : lean_dil<bf16_t, 4>(s, a, cin8, CT);
    if (d->dil == 8) return dtype == MSAU_F32 ? lean_dil<float, 8>(s, a, cin8, CT) : lean_dil<bf16_t, 8>(s, a, cin8, CT);
    if (d->KH == 4) return dtype == MSAU_F32 ? lean_ct<float, 1, 4, false>(s, a, CT) : lean_ct<bf16_t, 1, 4, false>(s, a, CT);
    if (dtype == MSAU_F32) return d->KH == 3 ? lean_cin<float, 3>(s, a, cin8, dual, CT) : lean_cin<float, 1>(s, a, cin8, dual, CT);
    return d->KH == 3 ? lean_cin<bf16_t, 3>(s, a, cin8, dual, CT) : lean_cin<bf16_t, 1>(s, a, cin8, dual, CT);
}
