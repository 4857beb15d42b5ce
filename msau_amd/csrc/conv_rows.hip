// Row-streaming form of the residual pair for the 8-channel layers (bf16): the two chained 3x3 SAME convs of
// MultiConvResidualBlock (model/model.py:37-50) and their two data gradients, the same arithmetic as conv_pair.hip
// (msau_conv_pair), restructured around what that kernel's counters showed (profiles/r02_pmc_conv_pair.txt: no wasted HBM
// traffic, LDS 42 % busy, waves parked 41 % of the time, half of every MFMA padding):
//
//  * pixel-pair packing.  With 8 output channels a 16-row MFMA tile is half empty.  Here the 16 rows are (pixel parity p,
//    channel co) and the 16 columns are pixel PAIRS (x = 2j + p): the pair reads a 3 x 4 x 8 = 96-element window, exactly
//    the three 32-deep k-steps the 72 -> 96 padded K cost before, so the same three MFMAs give 32 pixels x 8 channels
//    instead of 16 x 8.  A[(p, co)][(ky, kx4, ci)] = W[co][ci][ky][kx4 - p] (zero where kx4 - p is not a tap); it is built
//    in registers from the image msau_pack_params already writes (one 16-byte load per k-step and lane), no new pack kind.
//    Every lane of the result holds 4 channels of ONE pixel: no idle epilogue lanes, 8-byte stores, 512 contiguous bytes
//    per wave-instruction.
//  * a wave owns its tile; no workgroup barrier anywhere.  A wave walks down a strip of 30 output columns (32 lattice
//    columns = 16 pixel pairs, 34 input columns) one image row at a time.  The B fragment of k-step ky is "pixel
//    2*lr + lg of input row y + ky" -- a shape a global load can deliver directly, and the SAME registers serve as
//    ky = 2, 1, 0 of three consecutive rows: one 16-byte load per lane and row (three rows ahead), no LDS staging of the
//    input at all.  The intermediate row goes through LDS once (8-byte write in the result layout, 16-byte read in the
//    fragment layout; LDS operations of one wave execute in order, so no barrier) and is then carried in registers the
//    same way.  The residual operand (the input tensor itself) is fetched from the fragment registers of neighbour lanes
//    with two ds_bpermute.  Per row and wave: 1 load, 2 stores, 6 MFMAs, 1 LDS write, 1 LDS read, 2 permutes.
//    Waves drift apart and cover each other's latency.  The vertical halo is two rows per SEGMENT of rows.
//  * the ReLU masks travel as BALLOTS: the forward stores, per (row, strip), the four 64-bit lane masks (v[jj] > 0) of its
//    epilogue registers -- 32 bytes; the backward has the same lanes in the same places, loads the 32 bytes as scalars
//    and applies them with v_cndmask on an SGPR pair: one VALU instruction per element, no bit extraction.
//
// Tasks are (image, row segment, strip); four horizontally adjacent strips share a workgroup (their column halos hit in
// L1 / L2), consecutive workgroups of one XCD take consecutive tasks.
#include "msau_common.h"
#include <cstdlib>
#include <utility>

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;
constexpr unsigned kOOB = 0x80000000u;

constexpr int OW = 30, XW = 34;                   // output / input columns of a strip (lattice: 32)
constexpr int MP = XW * 16;                       // an intermediate row in LDS: 34 pixels x 8 bf16
constexpr int WAVE_LDS = 2 * MP;                  // two alternating slots per wave
constexpr int UNR = 6;                            // rows per loop body: 3 rows in use + 3 in flight, all statically named
typedef __attribute__((address_space(3))) bf16x4* lds_v4;
constexpr int WG_ROW = 36 * 16;                   // a row staged for a weight gradient: 34 pixels used, 16-byte slots

struct RowArgs {
    msau_conv_pair_desc d;
    int nstrips, nseg, SH, ntasks, tasks_per_xcd;
    int row_bytes;
    unsigned img_bytes, plane_img;                // bytes of one image / of one image's ballot plane (H * nstrips * 32)
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
// v = lane's bit of `mask` ? a : 0 -- the mask is a wave-uniform 64-bit value in an SGPR pair: one v_cndmask_b32 with the
// pair as its condition.  (A hand-written `asm("v_cndmask_b32 %0, 0, %1, %2")` computed wrong results here: the compiler
// re-used one SGPR pair for consecutive masks and an s_cselect overwrote it right behind the v_cndmask that still read it;
// the builtin leaves the scheduling and the hazards to the compiler.)
__device__ __forceinline__ float keep_if(float a, u64 mask) {
    return __builtin_amdgcn_inverse_ballot_w64(mask) ? a : 0.f;
}
__device__ __forceinline__ bf16x8 relu_bits(u32x4 v) { return relu8<bf16_t>(__builtin_bit_cast(bf16x8, v)); }
template <int I> struct IC { static constexpr int value = I; };

// MSAU_ROWS_SKEW=1 (compile time): run phase 2 one row behind phase 1 so that the two MFMA chains of an iteration are independent.
// The compiler does interleave them (ISA), the kernel got SLOWER: forward 15.1 -> 17.9 us, backward 15.5 -> 17.0 us (80 / 101
// registers instead of 76 / 88; three more iterations per task).  Off; kept as the measured alternative.
#ifndef MSAU_ROWS_SKEW
#define MSAU_ROWS_SKEW 0
#endif
constexpr int SKEW = MSAU_ROWS_SKEW;

// LRNB (MSAU_PAIR_LRN_BWD, backward flag set): the block's input is the output of LocalResponseNorm(size = 8) (layers.py:145,
// 161-162), whose backward used to be the next launch: read dy, read a, write da.  Here the finished row of dy is still in
// the result layout -- the 8 channels of a pixel in lanes l and l ^ 16 -- so the epilogue loads a (8 bytes per lane, two rows
// ahead), runs the same arithmetic as lrn_fast_kernel<G = 2> on the storage-rounded dy and stores da; y is not written.
//
// WG1 (MSAU_PAIR_WGRAD1, backward flag set): the weight gradient of the block's FIRST conv in the same launch.  Its g operand
// is the intermediate row this walk has just produced -- already in LDS as [pixel][8 channels], the layout rowwgrad8_kernel
// stages by hand -- so the stand-alone launch's read of g (22 MB at the bench size) and this launch's write of it (nothing
// else reads the gradient of the block's inner tensor) both disappear; what is added is one 16-byte load per lane and row
// of the block's input x0 (the weight gradient's other operand), its transposed fragment, and four MFMAs:
//     D[(a, ci)][(b, co)] += sum over 32 pixels  relu(x0)[ci](m + ky - 1, x + a) * g_mid[co](m, x + b)     (tap kx - 1 = a - b)
// exactly as in rowwgrad8_kernel.  A wave owns the lattice columns 1..30 of its own rows m; the other lattice columns
// (neighbour strips' pixels) are cleared in the fragment.  The four waves of a workgroup add up in LDS in a fixed order and
// the workgroup writes ONE slab (layout of msau_wgrad_reduce: [co][tap * 8 + ci], ones column 72).
//
// CPL (MSAU_PAIR_COUPLE, forward flag set): the coupling conv of a coupled stage (model/model.py:143-148,246-252) in the same launch:
// z = ReLU(Wc * concat(prev, y) + bc), a 1x1 conv, i.e. pixel-local on the output row this walk has just finished.  The row goes
// through LDS once more (8-byte write in the result layout, 16-byte read in the fragment layout) and ONE more MFMA per row yields the
// 32 pixels x 8 channels of z, with rowconv8_kernel<2, 1, 1>'s k order -- lane group lg = (source lg >> 1, pixel parity lg & 1): the
// lanes of groups 0, 1 carry `prev` (one 16-byte load per lane and row, three rows ahead), groups 2, 3 the row of y -- so the result is
// bit-identical to the stand-alone launch, whose read of y (22 MB at the bench size) and whose launch disappear.  CPL = 2 also writes
// the zero-padded 2x2 max pool of z (model/model.py:158-160): a window is this row and the previous one (kept in registers) of this
// lane and of lane ^ 32 (the odd column of the pair); same comparison order and rounded values as msau_maxpool2x2_fwd.
constexpr int CPL_ROW = 32 * 16;                  // a finished output row in LDS: 32 pixels x 8 bf16

template <bool BWD, bool BITS, bool LRNB = false, bool WG1 = false, int CPL = 0>
__global__ __launch_bounds__(256) void rowpair_c8_kernel(const RowArgs a) {
    static_assert(!LRNB || (BWD && !SKEW), "the LRN backward rides on the data-gradient launch");
    static_assert(!WG1 || (BWD && !SKEW), "the weight gradient rides on the data-gradient launch");
    static_assert(!CPL || (!BWD && !SKEW), "the coupling conv rides on the forward launch");
    __shared__ __align__(16) unsigned char smem[4 * WAVE_LDS + (WG1 ? 4 * WG_ROW + 8 * 80 * 4 : 0) + (CPL ? 4 * CPL_ROW : 0)];
    const msau_conv_pair_desc& d = a.d;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tloc = (blockIdx.x >> 3) * 4 + wave;
    const int task = (blockIdx.x & 7) * a.tasks_per_xcd + tloc;
    const bool live = tloc < a.tasks_per_xcd && task < a.ntasks;          // wave-uniform
    if constexpr (!WG1) { if (!live) return; }                            // (no barrier below; WG1: idle waves wait at the slab reduction)
    f32x4 wacc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, waccb = {0.f, 0.f, 0.f, 0.f};
    if (live) {
    const int t1 = task / a.nstrips, strip = task - t1 * a.nstrips;
    const int b = t1 / a.nseg, seg = t1 - b * a.nseg;
    const int H = d.H, W = d.W;
    const int x0 = strip * OW;
    const int y0 = seg * a.SH, y1 = min(H, y0 + a.SH);

    unsigned char* mr = smem + wave * WAVE_LDS;
    const int lr = lane & 15, lg = lane >> 4;
    const int c0 = (lg & 1) * 4;                                          // this lane's 4 channels ...
    const int j = 2 * lr + (lg >> 1);                                     // ... of lattice (phase 1) / output (phase 2) column j
    const int q = 2 * lr + lg;                                            // B fragment: pixel q of a 34-pixel row
    unsigned char* mwr = mr + j * 16 + c0 * 2;                            // result layout slot in an intermediate row
    const unsigned char* mrd = mr + q * 16;                               // fragment layout slot

    // ---- A fragments of both convs, in registers for the whole task
    bf16x8 A1[3], A2[3];
    {
        const int co = lr & 7, kx = lg - (lr >> 3);
        const bool tap = kx >= 0 && kx <= 2;
        const bf16_t* w1 = static_cast<const bf16_t*>(d.w1) + co * 96 + (tap ? kx : 0) * 8;
        const bf16_t* w2 = static_cast<const bf16_t*>(d.w2) + co * 96 + (tap ? kx : 0) * 8;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            A1[ky] = tap ? load8<bf16_t>(w1 + ky * 24) : zero8<bf16_t>();
            A2[ky] = tap ? load8<bf16_t>(w2 + ky * 24) : zero8<bf16_t>();
        }
    }
    f32x4 bias1 = {0.f, 0.f, 0.f, 0.f}, bias2 = bias1;
    if constexpr (!BWD) {
        if (d.b1) bias1 = *reinterpret_cast<const f32x4*>(d.b1 + c0);
        if (d.b2) bias2 = *reinterpret_cast<const f32x4*>(d.b2 + c0);
    }

    const long long img = (long long)b * a.img_bytes;
    const __amdgpu_buffer_rsrc_t rx = rsrc_of(static_cast<const char*>(d.x) + img, a.img_bytes);
    const __amdgpu_buffer_rsrc_t rmid = rsrc_of(static_cast<char*>(d.mid) + img, a.img_bytes);
    const __amdgpu_buffer_rsrc_t ry = rsrc_of(static_cast<char*>(d.y) + img, a.img_bytes);
    const __amdgpu_buffer_rsrc_t rbm = rsrc_of(BITS ? d.bits_mid + (long long)b * a.plane_img : nullptr, BITS ? a.plane_img : 0u);
    const __amdgpu_buffer_rsrc_t rba = rsrc_of(BITS ? d.bits_a + (long long)b * a.plane_img : nullptr, BITS ? a.plane_img : 0u);
    const __amdgpu_buffer_rsrc_t rla = rsrc_of(LRNB ? static_cast<const char*>(d.lrn_a) + img : nullptr, LRNB ? a.img_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rlda = rsrc_of(LRNB ? static_cast<char*>(d.lrn_da) + img : nullptr, LRNB ? a.img_bytes : 0u);
    const bool b075 = d.lrn_beta == 0.75f;
    // ---- WG1: x0 rows (columns x0 - 1 + lane, lanes 0..33) -> LDS -> transposed fragments; the g fragment comes from the M slots
    const __amdgpu_buffer_rsrc_t rwx = rsrc_of(WG1 ? static_cast<const char*>(d.wg1_x) + img : nullptr, WG1 ? a.img_bytes : 0u);
    unsigned char* wxbuf = smem + 4 * WAVE_LDS + wave * WG_ROW;
    const int wtr_off = (8 * (lane >> 4) + ((lane & 15) >> 2)) * 16 + (lane & 3) * 8;
    auto wfrag = [&](const unsigned char* buf) -> bf16x8 {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(buf + wtr_off));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(buf + wtr_off + 4 * 16));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    const int wcx = x0 - 1 + lane;
    const unsigned wxcol = lane < 34 && (unsigned)wcx < (unsigned)W ? (unsigned)(wcx * 16) : kOOB;
    auto load_wx = [&](int r) -> u32x4 {
        u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rwx, (unsigned)r < (unsigned)H && r <= y1 ? (unsigned)(r * a.row_bytes) + wxcol : kOOB, 0, 0);
        return __builtin_bit_cast(u32x4, relu_bits(v));                   // MSAU_PAIR_RELU_IN of the forward: the first conv saw max(x0, 0)
    };
    auto stage_wx = [&](u32x4 v) { if (lane < 34) *reinterpret_cast<u32x4*>(wxbuf + lane * 16) = v; };
    // element jj of the g fragment is lattice pixel 8 (lane >> 4) + jj + b, b = (lane & 15) >> 3: keep the own ones, 1..30
    u32x4 wgmask;
    {
        unsigned mk[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int p0 = 8 * lg + 2 * w + ((lane & 15) >> 3), p1 = p0 + 1;
            mk[w] = (p0 >= 1 && p0 <= OW ? 0x0000FFFFu : 0u) | (p1 >= 1 && p1 <= OW ? 0xFFFF0000u : 0u);
        }
        wgmask = u32x4{mk[0], mk[1], mk[2], mk[3]};
    }
    const u32x4 ones_bits = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    const bf16x8 WONES = __builtin_bit_cast(bf16x8, ones_bits);
    bf16x8 WAX[3] = {zero8<bf16_t>(), zero8<bf16_t>(), zero8<bf16_t>()};
    u32x4 WPX[3] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    if constexpr (WG1) {
        // the first intermediate row of the walk is m = y0 - 1 (not own: its g fragment is cleared), the first own one y0: it
        // needs x0 rows y0 - 1 (fragment now), y0 and y0 + 1 (staged by iterations 0 and 1); rows y0 .. y0 + 2 are requested here
        const u32x4 r1 = load_wx(y0 - 1);
#pragma unroll
        for (int k = 0; k < 3; ++k) WPX[k] = load_wx(y0 + k);
        stage_wx(r1);
        __builtin_amdgcn_wave_barrier();
        WAX[1] = wfrag(wxbuf);
        __builtin_amdgcn_wave_barrier();
    }

    // input row r in fragment layout: this lane's pixel is x0 - 2 + q; outside the image the offset is out of range -> 0
    const int lx = x0 - 2 + q;
    const unsigned lcol = (unsigned)lx < (unsigned)W ? (unsigned)(lx * 16) : kOOB;
    const int ylast = min(H - 1, y1 + 1);                                 // last input row this task reads
    auto load_row = [&](int r) -> u32x4 {
        const bool ok = r >= 0 && r <= ylast;
        return __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? (unsigned)(r * a.row_bytes) + lcol : kOOB, 0, 0);
    };
    // per-lane constants of the two epilogues
    const int xl = x0 - 1 + j;                                            // image column of lattice column j
    const float col_lim = (unsigned)xl < (unsigned)W ? INFINITY : 0.f;    // med3(v, 0, lim): ReLU inside the image, 0 outside
    const unsigned mid_col = (j >= 1 && j <= OW && xl < W) ? (unsigned)(xl * 16 + c0 * 2) : kOOB;
    const unsigned out_col = (j < OW && x0 + j < W) ? (unsigned)((x0 + j) * 16 + c0 * 2) : kOOB;
    const unsigned bal_off = lane < 8 ? (unsigned)(strip * 32 + lane * 4) : kOOB;
    const int plane_pitch = a.nstrips * 32;
    // the residual operand x(t, x0 + j), channels c0..c0+3, is pixel j + 2 of the input row: lanes with lg < 2 offer dwords
    // 0,1 (channels 0..3) of their pixel, lanes with lg >= 2 dwords 2,3; this lane fetches from the holder of its half
    const bool offer_hi = lg >= 2;
    const int res_src = ((lg & 1) ? lr + 16 * (2 + (lg >> 1)) : min(lr + 1, 15) + 16 * (lg >> 1)) * 4;

    // lanes 0..7 carry the eight dwords of a row's four ballots to memory: v_writelane puts a scalar into one lane of a
    // vector register (one VALU instruction per dword, straight-line; a `lane == k ? ... : ...` chain compiles to branches).
    // No hazard to pad: the lane select is an immediate, and the s_nop covers a ballot written by the VALU just before.
    auto lane_word = [&](const u64 (&bal)[4]) -> unsigned {
        unsigned w = 0;
        asm volatile("s_nop 1\n\t"
                     "v_writelane_b32 %0, %1, 0\n\tv_writelane_b32 %0, %2, 1\n\tv_writelane_b32 %0, %3, 2\n\tv_writelane_b32 %0, %4, 3\n\t"
                     "v_writelane_b32 %0, %5, 4\n\tv_writelane_b32 %0, %6, 5\n\tv_writelane_b32 %0, %7, 6\n\tv_writelane_b32 %0, %8, 7"
                     : "+v"(w)
                     : "s"((unsigned)bal[0]), "s"((unsigned)(bal[0] >> 32)), "s"((unsigned)bal[1]), "s"((unsigned)(bal[1] >> 32)),
                       "s"((unsigned)bal[2]), "s"((unsigned)(bal[2] >> 32)), "s"((unsigned)bal[3]), "s"((unsigned)(bal[3] >> 32)));
        return w;
    };

    // ballot words of the backward: scalar loads from the planes (constant address space -> s_load_dwordx8)
    typedef const __attribute__((address_space(4))) u64* cu64p;
    const unsigned long long pm0 = BWD ? (unsigned long long)(d.bits_mid + (long long)b * a.plane_img + strip * 32) : 0ull;
    const unsigned long long pa0 = BWD ? (unsigned long long)(d.bits_a + (long long)b * a.plane_img + strip * 32) : 0ull;
    auto plane_row = [&](unsigned long long base, int r) -> cu64p {
        const int rc = r < 0 ? 0 : (r >= H ? H - 1 : r);                  // rows outside the image: any valid row, the masks are cleared
        return (cu64p)(base + (unsigned long long)(rc * plane_pitch));
    };

    // ---- CPL: A fragment and bias of the coupling conv, the finished row's LDS slot, `prev` rows three ahead
    bf16x8 AC = zero8<bf16_t>();
    f32x4 cbias = {0.f, 0.f, 0.f, 0.f};
    unsigned char* ywr = smem + 4 * WAVE_LDS + wave * CPL_ROW + j * 16 + c0 * 2;                    // result layout
    const unsigned char* yrd = smem + 4 * WAVE_LDS + wave * CPL_ROW + (2 * lr + (lg & 1)) * 16;     // fragment layout: pixel 2 lr + parity
    const __amdgpu_buffer_rsrc_t rcp = rsrc_of(CPL ? static_cast<const char*>(d.cpl_prev) + img : nullptr, CPL ? a.img_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rcz = rsrc_of(CPL ? static_cast<char*>(d.cpl_y) + img : nullptr, CPL ? a.img_bytes : 0u);
    const int pcx = x0 + 2 * lr + (lg & 1);
    const unsigned pcol = lg < 2 && pcx < W ? (unsigned)(pcx * 16) : kOOB;                          // (groups 2, 3 carry y: no load)
    auto load_prev = [&](int r) -> u32x4 {
        return __builtin_amdgcn_raw_buffer_load_b128(rcp, r >= y0 && r < y1 ? (unsigned)(r * a.row_bytes) + pcol : kOOB, 0, 0);
    };
    u32x4 PV[3] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    u32x2 zprev = {0u, 0u};                                               // CPL = 2: the even row of the current pooling window
    const int Wo = (W + 1) >> 1, Ho = (H + 1) >> 1;
    const unsigned pimg = CPL == 2 ? (unsigned)Ho * (unsigned)Wo * 16u : 0u;
    const __amdgpu_buffer_rsrc_t rpz = rsrc_of(CPL == 2 ? static_cast<char*>(d.cpl_pool_y) + (long long)b * pimg : nullptr, pimg);
    const __amdgpu_buffer_rsrc_t rpi = rsrc_of(CPL == 2 && d.cpl_pool_idx ? d.cpl_pool_idx + (long long)b * (pimg / 2) : nullptr,
                                               CPL == 2 && d.cpl_pool_idx ? pimg / 2 : 0u);
    if constexpr (CPL != 0) {
        const int co = lr & 7, pp = lr >> 3;
        if ((lg & 1) == pp) AC = load8<bf16_t>(static_cast<const bf16_t*>(d.cpl_w) + (lg >> 1) * (16 * 32) + co * 32);
        if (d.cpl_b) cbias = *reinterpret_cast<const f32x4*>(d.cpl_b + c0);
#pragma unroll
        for (int k = 0; k < 3; ++k) PV[k] = load_prev(y0 - 2 + k);
    }

    // X[k]: input rows in fragment layout (raw until first used, ReLU'd after that in the forward); at iteration I row t + k
    // lives in X[(I + k) % 6].  RS[k % 3]: the half of the RAW pixel this lane offers as residual operand.  M[k % 3]:
    // intermediate rows in fragment layout.
    u32x4 X[UNR];
    u32x2 RS[3];
    bf16x8 M[3];
#pragma unroll
    for (int k = 0; k < 5; ++k) X[k] = load_row(y0 - 2 + k);
    auto first_use = [&](u32x4& x, u32x2& rs) {
        rs = offer_hi ? u32x2{x[2], x[3]} : u32x2{x[0], x[1]};
        if constexpr (!BWD) x = __builtin_bit_cast(u32x4, relu_bits(x));                            // MSAU_PAIR_RELU_IN
    };
    first_use(X[0], RS[0]);
    first_use(X[1], RS[1]);
    // stores the hardware drops: with them the loop is entered in the state its back edge leaves behind (rows t+2..t+4 in
    // flight, each followed by a row's stores), and the compiler's merged wait before the first rows is the steady-state
    // vmcnt(9), not vmcnt(1) -- which would wait for ALL rows in flight once per trip
    constexpr int kStoresPerRow = (WG1 ? 1 : 2) + (!BWD && BITS ? 2 : 0) + (CPL ? 1 : 0);
#pragma unroll
    for (int k = 0; k < 3 * kStoresPerRow; ++k) __builtin_amdgcn_raw_buffer_store_b64(u32x2{0u, 0u}, rmid, kOOB + 8u * k, 0, 0);   // (distinct: equal ones are merged)
    M[0] = M[1] = M[2] = zero8<bf16_t>();
    u32x2 RR[2] = {{0u, 0u}, {0u, 0u}};                                   // SKEW: the residual operand of row t, fetched one iteration before its use
    u64 mm[2][4], ma[2][4];                                               // masks of iterations i (slot i & 1), two rows ahead
    u32x2 AL[3] = {{0u, 0u}, {0u, 0u}, {0u, 0u}};                         // LRNB: a(t, x0 + j, c0..c0+3) of output row t in AL[t' % 3], two rows ahead
    if constexpr (BWD) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            cu64p pm = plane_row(pm0, y0 - 1 + k), pa = plane_row(pa0, y0 - 2 - SKEW + k);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) { mm[k][jj] = pm[jj]; ma[k][jj] = pa[jj]; }
        }
    }

    // one image row: t = tg + I; every register array index below is a compile-time constant
    auto step = [&](auto ic, const int tg) {
        constexpr int I = decltype(ic)::value, P = I & 1;
        const int t = tg + I, m = t + 1;                                  // output row, intermediate row of this iteration
        X[(I + 5) % 6] = load_row(t + 5);
        if constexpr (LRNB) {
            const int ta = t + 2;
            AL[(I + 2) % 3] = __builtin_amdgcn_raw_buffer_load_b64(rla, (ta >= y0 && ta < y1) ? (unsigned)(ta * a.row_bytes) + out_col : kOOB, 0, 0);
        }
        first_use(X[(I + 2) % 6], RS[(I + 2) % 3]);
        const bool rowin = (unsigned)m < (unsigned)H;
        // ================= phase 2: output row to = t - SKEW from intermediate rows to-1, to, to+1; columns x0 .. x0+29 ==========
        // SKEW = 1: the row finished ONE iteration ago -- nothing here depends on this iteration's phase 1, so the compiler can run
        // this epilogue in the shadow of phase 1's MFMAs and vice versa (un-skewed, 36-46 % of a wave's cycles were dependency stalls)
        auto phase2 = [&]() {
            const int to = t - SKEW;
            f32x4 acc = bias2;
            acc = mma8(A2[0], M[(I + 2 - SKEW + 3) % 3], acc);
            acc = mma8(A2[1], M[(I + 3 - SKEW) % 3], acc);
            acc = mma8(A2[2], M[(I + 1 - SKEW + 3) % 3], acc);
#ifdef MSAU_ROWS_KEEPALIVE
            asm volatile("" :: "v"(M[(I + 2) % 3]), "v"(M[I % 3]), "v"(M[(I + 1) % 3]));
#endif
            // residual (forward) / other-path gradient (backward): the raw input row t, from the neighbour lanes' registers
            u32x2 rr;
            if constexpr (SKEW) rr = RR[(I + 1) & 1];
            else {
                rr[0] = (unsigned)__builtin_amdgcn_ds_bpermute(res_src, (int)RS[I % 3][0]);
                rr[1] = (unsigned)__builtin_amdgcn_ds_bpermute(res_src, (int)RS[I % 3][1]);
            }
            const bf16x4 r = __builtin_bit_cast(bf16x4, rr);
            f32x4 v;
            if constexpr (!BWD) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) v[jj] = fmaxf(acc[jj] + (float)r[jj], 0.f);   // ADD, RELU_OUT
            } else {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) v[jj] = keep_if(acc[jj], ma[P][jj]) + (float)r[jj];   // MASK_A, ADD
            }
            bf16x4 o;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) o[jj] = (bf16_t)v[jj];
            const bool ownrow = to >= y0 && to < y1;
            if constexpr (LRNB) {
                // da = dy d^-beta - 2 beta alpha/n a * sum over the adjoint window [c - 3, c + 4] of dy a d^(-beta-1), d = k + alpha/n *
                // sum over [c - 4, c + 3] of a^2; window sums = differences of prefix sums, the other half from lane ^ 16
                const bf16x4 av = __builtin_bit_cast(bf16x4, AL[I % 3]);
                const int h = lg & 1;
                auto swz = [](float f) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, f), 0x401F)); };
                float xa[4], gg[4], Pf[4], run = 0.f;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) { xa[jj] = (float)av[jj]; gg[jj] = (float)o[jj]; run += xa[jj] * xa[jj]; Pf[jj] = run; }
                const float other = swz(run);
                const float E = h ? other : 0.f, tot = run + other;
                float dnb[4], Q[4], runq = 0.f;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float Xe = E + (jj ? Pf[jj - 1] : 0.f);        // exclusive prefix at channel 4h + jj
                    const float Xo = swz(Xe);
                    const float win = h ? tot - Xo : Xo;
                    const float dd = d.lrn_k + d.lrn_alpha_over_n * win;
                    float invd;
                    if (b075) { const float r = __builtin_amdgcn_rsqf(dd); dnb[jj] = r * __builtin_amdgcn_sqrtf(r); invd = r * r; }
                    else { dnb[jj] = __expf(-d.lrn_beta * __logf(dd)); invd = __builtin_amdgcn_rcpf(dd); }
                    runq += gg[jj] * xa[jj] * dnb[jj] * invd;
                    Q[jj] = runq;
                }
                const float otherq = swz(runq);
                const float Eq = h ? otherq : 0.f, totq = runq + otherq;
                const float c2 = 2.f * d.lrn_beta * d.lrn_alpha_over_n;
                bf16x4 da;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float Io = swz(Eq + Q[jj]);                     // the other half's inclusive prefix at the same jj
                    const float adj = h ? totq - Io : Io;
                    da[jj] = (bf16_t)(gg[jj] * dnb[jj] - c2 * xa[jj] * adj);
                }
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, da), rlda, ownrow ? (unsigned)(to * a.row_bytes) + out_col : kOOB, 0, 0);
            } else
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), ry, ownrow ? (unsigned)(to * a.row_bytes) + out_col : kOOB, 0, 0);
            if constexpr (!BWD && BITS) {
                u64 bal[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) bal[jj] = __builtin_amdgcn_ballot_w64((float)r[jj] > 0.f);      // (x > 0): MASK_A of the backward
                __builtin_amdgcn_raw_buffer_store_b32(lane_word(bal), rba, ownrow ? (unsigned)(to * plane_pitch) + bal_off : kOOB, 0, 0);
            }
            if constexpr (CPL != 0) {
                // ---- the coupling conv on the finished row: fragment = [prev px 2lr | prev px 2lr+1 | y px 2lr | y px 2lr+1]
                *reinterpret_cast<bf16x4*>(ywr) = o;
                __builtin_amdgcn_wave_barrier();
                const u32x4 yb = *reinterpret_cast<const u32x4*>(yrd);
                __builtin_amdgcn_wave_barrier();
                u32x4 bf;
#pragma unroll
                for (int w = 0; w < 4; ++w) bf[w] = lg < 2 ? PV[I % 3][w] : yb[w];
                PV[I % 3] = load_prev(to + 3);
                const f32x4 zacc = mma8(AC, __builtin_bit_cast(bf16x8, bf), cbias);
                bf16x4 zo;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) zo[jj] = (bf16_t)fmaxf(zacc[jj], 0.f);               // MSAU_CONV_RELU_OUT
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, zo), rcz, ownrow ? (unsigned)(to * a.row_bytes) + out_col : kOOB, 0, 0);
                if constexpr (CPL == 2) {
                    // rows to - 1 (even, in zprev) and to (odd); columns x0 + j (even: lanes of groups 0, 1) and x0 + j + 1 (lane ^ 32).
                    // The window of the image's last row / column, when H / W is odd, sees zeros there (zero padding).
                    const bool zok = ownrow && out_col != kOOB;
                    const u32x2 cur = zok ? __builtin_bit_cast(u32x2, zo) : u32x2{0u, 0u};
                    const bool odd = (to & 1) != 0;                       // wave-uniform (y0 is even)
                    const bool last_even = !odd && to == H - 1;           // a window with no second row
                    if (odd || last_even) {
                        const u32x2 top = odd ? zprev : cur, bot = odd ? cur : u32x2{0u, 0u};
                        const int nb = (lane ^ 32) * 4;
                        u32x2 ntop, nbot;
#pragma unroll
                        for (int w = 0; w < 2; ++w) {
                            ntop[w] = (unsigned)__builtin_amdgcn_ds_bpermute(nb, (int)top[w]);
                            nbot[w] = (unsigned)__builtin_amdgcn_ds_bpermute(nb, (int)bot[w]);
                        }
                        const bf16x4 vt = __builtin_bit_cast(bf16x4, top), vnt = __builtin_bit_cast(bf16x4, ntop),
                                     vb = __builtin_bit_cast(bf16x4, bot), vnb = __builtin_bit_cast(bf16x4, nbot);
                        bf16x4 best;
                        unsigned idx = 0;
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            float bv = (float)vt[jj];
                            unsigned bi = 0;
                            const float c1 = (float)vnt[jj], c2 = (float)vb[jj], c3 = (float)vnb[jj];
                            if (c1 > bv) { bv = c1; bi = 1; }
                            if (c2 > bv) { bv = c2; bi = 2; }
                            if (c3 > bv) { bv = c3; bi = 3; }
                            best[jj] = (bf16_t)bv;
                            idx |= bi << (8 * jj);
                        }
                        const int ty = odd ? to - 1 : to;
                        const bool pok = lg < 2 && j < OW && x0 + j < W && ty >= y0 && ty < y1;
                        const unsigned e = (unsigned)(((ty >> 1) * Wo + ((x0 + j) >> 1)) * 8 + c0);      // elements
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, best), rpz, pok ? e * 2 : kOOB, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(idx, rpi, pok ? e : kOOB, 0, 0);
                    }
                    zprev = cur;
                }
            }
        };
        if constexpr (SKEW) {
            phase2();
            // the residual operand of row t for the next iteration
            RR[I & 1][0] = (unsigned)__builtin_amdgcn_ds_bpermute(res_src, (int)RS[I % 3][0]);
            RR[I & 1][1] = (unsigned)__builtin_amdgcn_ds_bpermute(res_src, (int)RS[I % 3][1]);
        }
        // ================= phase 1: intermediate row m from input rows t, t+1, t+2; lattice columns 0..31 =================
        {
            f32x4 acc = bias1;
            acc = mma8(A1[0], __builtin_bit_cast(bf16x8, X[I % 6]), acc);
            acc = mma8(A1[1], __builtin_bit_cast(bf16x8, X[(I + 1) % 6]), acc);
            acc = mma8(A1[2], __builtin_bit_cast(bf16x8, X[(I + 2) % 6]), acc);
#ifdef MSAU_ROWS_KEEPALIVE
            asm volatile("" :: "v"(X[I % 6]), "v"(X[(I + 1) % 6]), "v"(X[(I + 2) % 6]));
#endif
            f32x4 v;
            if constexpr (!BWD) {
                const float lim = rowin ? col_lim : 0.f;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) v[jj] = __builtin_amdgcn_fmed3f(acc[jj], 0.f, lim);   // MSAU_PAIR_RELU_MID; zero outside = the second conv's padding
            } else {
                const u64 rowmask = rowin ? ~0ull : 0ull;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) v[jj] = keep_if(acc[jj], mm[P][jj] & rowmask);       // MSAU_PAIR_MASK_MID (the planes are zero outside the image)
            }
            bf16x4 o;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) o[jj] = (bf16_t)v[jj];
            *reinterpret_cast<bf16x4*>(mwr + P * MP) = o;
            const bool ownrow = m >= y0 && m < y1;
            if constexpr (!WG1)      // (WG1: the only other reader of this tensor was the weight-gradient launch)
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rmid, ownrow ? (unsigned)(m * a.row_bytes) + mid_col : kOOB, 0, 0);
            if constexpr (!BWD && BITS) {
                // (mid > 0) as the four lane masks of this row: what the backward's phase 1 applies to the same lanes
                u64 bal[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) bal[jj] = __builtin_amdgcn_ballot_w64((float)o[jj] > 0.f);
                __builtin_amdgcn_raw_buffer_store_b32(lane_word(bal), rbm, ownrow ? (unsigned)(m * plane_pitch) + bal_off : kOOB, 0, 0);
            }
        }
        // lanes exchange the row through LDS: to the compiler each lane is a thread whose own write and read may not even
        // overlap, so without this (it emits nothing) the read can be hoisted above the write -- it was, at 16 channels
        __builtin_amdgcn_wave_barrier();
        M[(I + 1) % 3] = *reinterpret_cast<const bf16x8*>(mrd + P * MP);    // the row just written, in fragment layout
        if constexpr (WG1) {
            // x0 row m + 1 = t + 2 -> fragment; rows m - 1, m are WAX[I % 3], WAX[(I + 1) % 3] from the iterations before
            stage_wx(WPX[I % 3]);
            WPX[I % 3] = load_wx(t + 5);
            __builtin_amdgcn_wave_barrier();
            WAX[(I + 2) % 3] = wfrag(wxbuf);
            u32x4 gb = __builtin_bit_cast(u32x4, wfrag(mr + P * MP));
            __builtin_amdgcn_wave_barrier();
            const bool ownm = m >= y0 && m < y1;
#pragma unroll
            for (int w = 0; w < 4; ++w) gb[w] = ownm ? gb[w] & wgmask[w] : 0u;
            const bf16x8 BG = __builtin_bit_cast(bf16x8, gb);
            wacc[0] = mma8(WAX[I % 3], BG, wacc[0]);                        // ky 0: x0 row m - 1
            wacc[1] = mma8(WAX[(I + 1) % 3], BG, wacc[1]);
            wacc[2] = mma8(WAX[(I + 2) % 3], BG, wacc[2]);
            waccb = mma8(WONES, BG, waccb);
        }
        if constexpr (!SKEW) phase2();
        if constexpr (BWD) {                                               // masks of iteration t + 2
            cu64p pm = plane_row(pm0, m + 2), pa = plane_row(pa0, t + 2 - SKEW);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) { mm[P][jj] = pm[jj]; ma[P][jj] = pa[jj]; }
        }
        // rows stay in program order: hoisting a later row's MFMAs above this point makes the compiler wait for that row's
        // load here, i.e. it would give the prefetch distance away
        __builtin_amdgcn_sched_barrier(0);
    };

    for (int tg = y0 - 2; tg < y1 + SKEW; tg += UNR) {
        step(IC<0>{}, tg);
        step(IC<1>{}, tg);
        step(IC<2>{}, tg);
        step(IC<3>{}, tg);
        step(IC<4>{}, tg);
        step(IC<5>{}, tg);
    }
    }   // live
    if constexpr (WG1) {
        // ---- the four waves' sums -> one slab, fixed order (as rowwgrad8_kernel)
        float* slab = reinterpret_cast<float*>(smem + 4 * WAVE_LDS + 4 * WG_ROW);
        for (int i = threadIdx.x; i < 8 * 80; i += 256) slab[i] = 0.f;
        __syncthreads();
        const int n = lane & 15, kg = lane >> 4, bb = n >> 3, co = n & 7;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (wave == w) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int row = 4 * kg + jj, aa = row >> 3, ci = row & 7;
                    if (!(aa == 1 && bb == 1)) {
                        const int kx = aa - bb + 1;
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky) slab[co * 80 + (ky * 3 + kx) * 8 + ci] += wacc[ky][jj];
                    }
                }
                if (kg == 0 && bb == 0) slab[co * 80 + 72] += waccb[0];
            }
            __syncthreads();
        }
        float* out = a.d.wg1_slabs + (size_t)blockIdx.x * (8 * 80);
        for (int i = threadIdx.x; i < 8 * 80; i += 256) out[i] = slab[i];
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// The same walk for the 16-channel layers (level 1).  16 output channels fill the 16 MFMA rows, so columns are 16 single
// pixels: a strip is 14 output columns (16 lattice, 18 input columns), a lane of the result holds channels 4*lg..4*lg+3 of
// pixel lr.  K per input row is 3 taps x 16 channels = 6 eight-channel groups, run as two k-steps of 4 groups with the last
// two groups' weights zero (the MFMA pipe is far from the bound; what matters is that a fragment is ROW-pure and can be
// carried in registers while its row serves as ky = 2, 1, 0): fragment a = groups (kx 0, 1) x (channels 0-7, 8-15), i.e.
// lane (lr, lg) holds pixel lr + (lg >> 1), channel group lg & 1; fragment b = kx 2, lane (lr, lg) holds pixel lr + 2,
// channel group lg & 1 (lanes lg >= 2 meet zero weights; their copy of the pixel is what the residual fetch reads).
// MSAU_CONV_POOL: rows arrive one at a time, so the 2x2 window is this row and the previous one (kept in registers) of
// this lane and its neighbour column lr ^ 1 -- strips and segments start at even coordinates; same comparison order and
// rounded values as msau_maxpool2x2_fwd.
constexpr int OW16 = 14, XW16 = 18;
constexpr int MP16 = XW16 * 32;
constexpr int WAVE_LDS16 = 2 * MP16;

// CPL (MSAU_PAIR_COUPLE): as in rowpair_c8_kernel; here K = 32 is concat(prev, y) of ONE pixel -- lane group lg = (source lg >> 1,
// 8-channel group lg & 1) -- and the pooled z takes the place of POOL's pooled y (same code, other operands).
constexpr int CPL_ROW16 = 16 * 32;                // a finished output row in LDS: 16 pixels x 16 bf16

// LRNB (MSAU_PAIR_LRN_BWD at 16 channels, round 5): as in rowpair_c8_kernel, with a pixel's 16 channels in the four lanes lr + 16 lg:
// window sums = differences of prefix sums whose other half sits in lane ^ 32, lane totals over lane ^ 16 and lane ^ 32.
template <bool BWD, bool BITS, bool POOL, int CPL = 0, bool LRNB = false>
__global__ __launch_bounds__(256) void rowpair_c16_kernel(const RowArgs a) {
    static_assert(!(BWD && POOL), "the pooled output belongs to the forward launch");
    static_assert(!LRNB || BWD, "the LRN backward rides on the data-gradient launch");
    static_assert(!CPL || (!BWD && !POOL), "the coupling conv rides on the forward launch; its pooled output replaces POOL's");
    __shared__ __align__(16) unsigned char smem[4 * WAVE_LDS16 + (CPL ? 4 * CPL_ROW16 : 0)];
    const msau_conv_pair_desc& d = a.d;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tloc = (blockIdx.x >> 3) * 4 + wave;
    const int task = (blockIdx.x & 7) * a.tasks_per_xcd + tloc;
    if (tloc >= a.tasks_per_xcd || task >= a.ntasks) return;              // wave-uniform; there is no barrier below
    const int t1 = task / a.nstrips, strip = task - t1 * a.nstrips;
    const int b = t1 / a.nseg, seg = t1 - b * a.nseg;
    const int H = d.H, W = d.W;
    const int x0 = strip * OW16;
    const int y0 = seg * a.SH, y1 = min(H, y0 + a.SH);

    unsigned char* mr = smem + wave * WAVE_LDS16;
    const int lr = lane & 15, lg = lane >> 4;
    const int c0 = lg * 4;                                                // this lane's 4 channels of lattice / output column lr
    unsigned char* mwr = mr + lr * 32 + c0 * 2;                           // result layout slot in an intermediate row
    const unsigned char* mrd_a = mr + (lr + (lg >> 1)) * 32 + (lg & 1) * 16;
    const unsigned char* mrd_b = mr + (lr + 2) * 32 + (lg & 1) * 16;
    // columns 16, 17 of the intermediate rows are read (by lanes whose outputs are not stored) and never written: not NaN
    for (int i = lane; i < WAVE_LDS16 / 16; i += 64) *reinterpret_cast<u32x4*>(mr + i * 16) = u32x4{0u, 0u, 0u, 0u};
    __builtin_amdgcn_wave_barrier();

    // ---- A fragments: [conv][ky][a | b]
    bf16x8 A1[3][2], A2[3][2];
    {
        const int kxa = lg >> 1, cg = lg & 1;
        const bf16_t* w1 = static_cast<const bf16_t*>(d.w1) + lr * 160 + cg * 8;
        const bf16_t* w2 = static_cast<const bf16_t*>(d.w2) + lr * 160 + cg * 8;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            A1[ky][0] = load8<bf16_t>(w1 + (ky * 3 + kxa) * 16);
            A2[ky][0] = load8<bf16_t>(w2 + (ky * 3 + kxa) * 16);
            A1[ky][1] = lg < 2 ? load8<bf16_t>(w1 + (ky * 3 + 2) * 16) : zero8<bf16_t>();
            A2[ky][1] = lg < 2 ? load8<bf16_t>(w2 + (ky * 3 + 2) * 16) : zero8<bf16_t>();
        }
    }
    f32x4 bias1 = {0.f, 0.f, 0.f, 0.f}, bias2 = bias1;
    if constexpr (!BWD) {
        if (d.b1) bias1 = *reinterpret_cast<const f32x4*>(d.b1 + c0);
        if (d.b2) bias2 = *reinterpret_cast<const f32x4*>(d.b2 + c0);
    }

    const long long img = (long long)b * a.img_bytes;
    const __amdgpu_buffer_rsrc_t rx = rsrc_of(static_cast<const char*>(d.x) + img, a.img_bytes);
    const __amdgpu_buffer_rsrc_t rmid = rsrc_of(static_cast<char*>(d.mid) + img, a.img_bytes);
    const __amdgpu_buffer_rsrc_t ry = rsrc_of(static_cast<char*>(d.y) + img, a.img_bytes);
    const __amdgpu_buffer_rsrc_t rbm = rsrc_of(BITS ? d.bits_mid + (long long)b * a.plane_img : nullptr, BITS ? a.plane_img : 0u);
    const __amdgpu_buffer_rsrc_t rba = rsrc_of(BITS ? d.bits_a + (long long)b * a.plane_img : nullptr, BITS ? a.plane_img : 0u);

    const __amdgpu_buffer_rsrc_t rla = rsrc_of(LRNB ? static_cast<const char*>(d.lrn_a) + img : nullptr, LRNB ? a.img_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rlda = rsrc_of(LRNB ? static_cast<char*>(d.lrn_da) + img : nullptr, LRNB ? a.img_bytes : 0u);
    const bool b075 = d.lrn_beta == 0.75f;
    u32x2 AL[3] = {{0u, 0u}, {0u, 0u}, {0u, 0u}};                         // LRNB: a(t, x0 + lr, c0..c0+3) of output row t, two rows ahead
    // input row r in fragment layout (two 16-byte loads per lane); outside the image the offset is out of range -> 0
    const int lxa = x0 - 2 + lr + (lg >> 1), lxb = x0 - 2 + lr + 2;
    const unsigned lcol_a = (unsigned)lxa < (unsigned)W ? (unsigned)(lxa * 32 + (lg & 1) * 16) : kOOB;
    const unsigned lcol_b = (unsigned)lxb < (unsigned)W ? (unsigned)(lxb * 32 + (lg & 1) * 16) : kOOB;
    const int ylast = min(H - 1, y1 + 1);
    // per-lane constants of the two epilogues
    const int xl = x0 - 1 + lr;                                           // image column of lattice column lr
    const float col_lim = (unsigned)xl < (unsigned)W ? INFINITY : 0.f;
    const unsigned mid_col = (lr >= 1 && lr <= OW16 && xl < W) ? (unsigned)(xl * 32 + c0 * 2) : kOOB;
    const bool out_ok = lr < OW16 && x0 + lr < W;
    const unsigned out_col = out_ok ? (unsigned)((x0 + lr) * 32 + c0 * 2) : kOOB;
    const unsigned bal_off = lane < 8 ? (unsigned)(strip * 32 + lane * 4) : kOOB;
    const int plane_pitch = a.nstrips * 32;
    // the residual operand x(t, x0 + lr), channels c0..c0+3 = half (lg & 1) of channel group (lg >> 1) of input pixel lr + 2:
    // fragment b of the row; lanes lg' < 2 offer dwords 0,1 of their group lg' & 1, lanes lg' >= 2 dwords 2,3
    const bool offer_hi = lg >= 2;
    const int res_src = (lr + 16 * ((lg >> 1) + 2 * (lg & 1))) * 4;

    auto lane_word = [&](const u64 (&bal)[4]) -> unsigned {
        unsigned w = 0;
        asm volatile("s_nop 1\n\t"
                     "v_writelane_b32 %0, %1, 0\n\tv_writelane_b32 %0, %2, 1\n\tv_writelane_b32 %0, %3, 2\n\tv_writelane_b32 %0, %4, 3\n\t"
                     "v_writelane_b32 %0, %5, 4\n\tv_writelane_b32 %0, %6, 5\n\tv_writelane_b32 %0, %7, 6\n\tv_writelane_b32 %0, %8, 7"
                     : "+v"(w)
                     : "s"((unsigned)bal[0]), "s"((unsigned)(bal[0] >> 32)), "s"((unsigned)bal[1]), "s"((unsigned)(bal[1] >> 32)),
                       "s"((unsigned)bal[2]), "s"((unsigned)(bal[2] >> 32)), "s"((unsigned)bal[3]), "s"((unsigned)(bal[3] >> 32)));
        return w;
    };
    typedef const __attribute__((address_space(4))) u64* cu64p;
    const unsigned long long pm0 = BWD ? (unsigned long long)(d.bits_mid + (long long)b * a.plane_img + strip * 32) : 0ull;
    const unsigned long long pa0 = BWD ? (unsigned long long)(d.bits_a + (long long)b * a.plane_img + strip * 32) : 0ull;
    auto plane_row = [&](unsigned long long base, int r) -> cu64p {
        const int rc = r < 0 ? 0 : (r >= H ? H - 1 : r);
        return (cu64p)(base + (unsigned long long)(rc * plane_pitch));
    };

    // XA / XB[k]: input rows in fragment layout; at iteration I row t + k lives in slot (I + k) % 6.  Rows are loaded two ahead.
    constexpr int NXR = 6;
    u32x4 XA[NXR], XB[NXR];
    u32x2 RS[3];
    bf16x8 MA[3], MB[3];
    auto load_row = [&](int r, u32x4& xa, u32x4& xb) {
        const bool ok = r >= 0 && r <= ylast;
        const unsigned ro = (unsigned)(r * a.row_bytes);
        xa = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? ro + lcol_a : kOOB, 0, 0);
        xb = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? ro + lcol_b : kOOB, 0, 0);
    };
#pragma unroll
    for (int k = 0; k < 5; ++k) load_row(y0 - 2 + k, XA[k], XB[k]);
    auto first_use = [&](u32x4& xa, u32x4& xb, u32x2& rs) {
        rs = offer_hi ? u32x2{xb[2], xb[3]} : u32x2{xb[0], xb[1]};
        if constexpr (!BWD) {                                              // MSAU_PAIR_RELU_IN
            xa = __builtin_bit_cast(u32x4, relu_bits(xa));
            xb = __builtin_bit_cast(u32x4, relu_bits(xb));
        }
    };
    first_use(XA[0], XB[0], RS[0]);
    first_use(XA[1], XB[1], RS[1]);
    constexpr int kStoresPerRow = 2 + (!BWD && BITS ? 2 : 0) + (CPL ? 1 : 0);            // see rowpair_c8_kernel
#pragma unroll
    for (int k = 0; k < 3 * kStoresPerRow; ++k)
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{0u, 0u}, rmid, kOOB + 8u * k, 0, 0);
    MA[0] = MA[2] = MB[0] = MB[2] = zero8<bf16_t>();
    u64 mm[2][4], ma[2][4];
    if constexpr (BWD) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            cu64p pm = plane_row(pm0, y0 - 1 + k), pa = plane_row(pa0, y0 - 2 + k);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) { mm[k][jj] = pm[jj]; ma[k][jj] = pa[jj]; }
        }
    }
    bf16x4 prev_out = {};                                                 // POOL / CPL = 2: the even row of the current window
    constexpr bool PL = POOL || CPL == 2;
    const int Wo = (W + 1) >> 1, Ho = (H + 1) >> 1;
    const unsigned pimg = PL ? (unsigned)Ho * (unsigned)Wo * 32u : 0u;
    void* const pool_y = CPL == 2 ? d.cpl_pool_y : d.pool_y;
    uint8_t* const pool_idx = CPL == 2 ? d.cpl_pool_idx : d.pool_idx;
    const __amdgpu_buffer_rsrc_t rp = rsrc_of(PL ? static_cast<char*>(pool_y) + (long long)b * pimg : nullptr, pimg);
    const __amdgpu_buffer_rsrc_t ri = rsrc_of(PL && pool_idx ? pool_idx + (long long)b * (pimg / 2) : nullptr, PL && pool_idx ? pimg / 2 : 0u);
    // ---- CPL: A fragment and bias of the coupling conv, the finished row's LDS slot, `prev` rows three ahead
    bf16x8 AC = zero8<bf16_t>();
    f32x4 cbias = {0.f, 0.f, 0.f, 0.f};
    unsigned char* ywr = smem + 4 * WAVE_LDS16 + wave * CPL_ROW16 + lr * 32 + c0 * 2;               // result layout
    const unsigned char* yrd = smem + 4 * WAVE_LDS16 + wave * CPL_ROW16 + lr * 32 + (lg & 1) * 16;  // fragment layout: pixel lr, group lg & 1
    const __amdgpu_buffer_rsrc_t rcp = rsrc_of(CPL ? static_cast<const char*>(d.cpl_prev) + img : nullptr, CPL ? a.img_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rcz = rsrc_of(CPL ? static_cast<char*>(d.cpl_y) + img : nullptr, CPL ? a.img_bytes : 0u);
    const unsigned pcol = lg < 2 && x0 + lr < W ? (unsigned)((x0 + lr) * 32 + (lg & 1) * 16) : kOOB;     // (groups 2, 3 carry y: no load)
    auto load_prev = [&](int r) -> u32x4 {
        return __builtin_amdgcn_raw_buffer_load_b128(rcp, r >= y0 && r < y1 ? (unsigned)(r * a.row_bytes) + pcol : kOOB, 0, 0);
    };
    u32x4 PV[3] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    if constexpr (CPL != 0) {
        AC = load8<bf16_t>(static_cast<const bf16_t*>(d.cpl_w) + (lg >> 1) * (16 * 32) + lr * 32 + (lg & 1) * 8);
        if (d.cpl_b) cbias = *reinterpret_cast<const f32x4*>(d.cpl_b + c0);
#pragma unroll
        for (int k = 0; k < 3; ++k) PV[k] = load_prev(y0 - 2 + k);
    }

    auto step = [&](auto ic, const int tg) {
        constexpr int I = decltype(ic)::value, P = I & 1;
        const int t = tg + I, m = t + 1;
        load_row(t + 5, XA[(I + 5) % 6], XB[(I + 5) % 6]);
        if constexpr (LRNB) {
            const int ta = t + 2;
            AL[(I + 2) % 3] = __builtin_amdgcn_raw_buffer_load_b64(rla, (ta >= y0 && ta < y1) ? (unsigned)(ta * a.row_bytes) + out_col : kOOB, 0, 0);
        }
        first_use(XA[(I + 2) % 6], XB[(I + 2) % 6], RS[(I + 2) % 3]);
        const bool rowin = (unsigned)m < (unsigned)H;
        // ================= phase 1: intermediate row m =================
        {
            f32x4 acc = bias1;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                acc = mma8(A1[ky][0], __builtin_bit_cast(bf16x8, XA[(I + ky) % 6]), acc);
                acc = mma8(A1[ky][1], __builtin_bit_cast(bf16x8, XB[(I + ky) % 6]), acc);
            }
            f32x4 v;
            if constexpr (!BWD) {
                const float lim = rowin ? col_lim : 0.f;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) v[jj] = __builtin_amdgcn_fmed3f(acc[jj], 0.f, lim);
            } else {
                const u64 rowmask = rowin ? ~0ull : 0ull;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) v[jj] = keep_if(acc[jj], mm[P][jj] & rowmask);
            }
            bf16x4 o;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) o[jj] = (bf16_t)v[jj];
            *reinterpret_cast<bf16x4*>(mwr + P * MP16) = o;
            const bool ownrow = m >= y0 && m < y1;
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rmid, ownrow ? (unsigned)(m * a.row_bytes) + mid_col : kOOB, 0, 0);
            if constexpr (!BWD && BITS) {
                u64 bal[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) bal[jj] = __builtin_amdgcn_ballot_w64((float)o[jj] > 0.f);
                __builtin_amdgcn_raw_buffer_store_b32(lane_word(bal), rbm, ownrow ? (unsigned)(m * plane_pitch) + bal_off : kOOB, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();                                  // see rowpair_c8_kernel: the write above, then the reads
        MA[(I + 1) % 3] = *reinterpret_cast<const bf16x8*>(mrd_a + P * MP16);
        MB[(I + 1) % 3] = *reinterpret_cast<const bf16x8*>(mrd_b + P * MP16);
        // ================= phase 2: output row t =================
        {
            f32x4 acc = bias2;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                acc = mma8(A2[ky][0], MA[(I + 2 + ky) % 3], acc);
                acc = mma8(A2[ky][1], MB[(I + 2 + ky) % 3], acc);
            }
            u32x2 rr;
            rr[0] = (unsigned)__builtin_amdgcn_ds_bpermute(res_src, (int)RS[I % 3][0]);
            rr[1] = (unsigned)__builtin_amdgcn_ds_bpermute(res_src, (int)RS[I % 3][1]);
            const bf16x4 r = __builtin_bit_cast(bf16x4, rr);
            f32x4 v;
            if constexpr (!BWD) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) v[jj] = fmaxf(acc[jj] + (float)r[jj], 0.f);
            } else {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) v[jj] = keep_if(acc[jj], ma[P][jj]) + (float)r[jj];
            }
            bf16x4 o;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) o[jj] = (bf16_t)v[jj];
            const bool ownrow = t >= y0 && t < y1;
            if constexpr (LRNB) {
                // da = dy d^-beta - 2 beta alpha/n a * sum over the adjoint window [c - 7, c + 8] of dy a d^(-beta-1), d = k + alpha/n * sum over
                // [c - 8, c + 7] of a^2 (layers.py:145,161-162; size 16): this lane holds channels 4 lg .. 4 lg + 3 of its pixel
                const bf16x4 av = __builtin_bit_cast(bf16x4, AL[I % 3]);
                const int h = lg >> 1;
                auto x16 = [](float f) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, f), 0x401F)); };            // lane ^ 16
                const int nb32 = (lane ^ 32) * 4;
                auto x32 = [&](float f) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(nb32, __builtin_bit_cast(int, f))); };   // lane ^ 32
                float xa[4], gg[4], Pf[4], run = 0.f;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) { xa[jj] = (float)av[jj]; gg[jj] = (float)o[jj]; run += xa[jj] * xa[jj]; Pf[jj] = run; }
                const float t1 = x16(run), pr = run + t1, t2 = x32(pr);
                const float tot = pr + t2;
                const float E = ((lg & 1) ? t1 : 0.f) + (h ? t2 : 0.f);          // channels below this lane's
                float dnb[4], Q[4], runq = 0.f;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float Xe = E + (jj ? Pf[jj - 1] : 0.f);                 // exclusive prefix at channel 4 lg + jj
                    const float Xo = x32(Xe);                                     // ... at channel (4 lg + jj) +- 8
                    const float win = h ? tot - Xo : Xo;
                    const float dd = d.lrn_k + d.lrn_alpha_over_n * win;
                    float invd;
                    if (b075) { const float rq = __builtin_amdgcn_rsqf(dd); dnb[jj] = rq * __builtin_amdgcn_sqrtf(rq); invd = rq * rq; }
                    else { dnb[jj] = __expf(-d.lrn_beta * __logf(dd)); invd = __builtin_amdgcn_rcpf(dd); }
                    runq += gg[jj] * xa[jj] * dnb[jj] * invd;
                    Q[jj] = runq;
                }
                const float q1 = x16(runq), qp = runq + q1, q2 = x32(qp);
                const float totq = qp + q2;
                const float Eq = ((lg & 1) ? q1 : 0.f) + (h ? q2 : 0.f);
                const float c2 = 2.f * d.lrn_beta * d.lrn_alpha_over_n;
                bf16x4 da;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float Io = x32(Eq + Q[jj]);                             // the inclusive prefix 8 channels away
                    const float adj = h ? totq - Io : Io;
                    da[jj] = (bf16_t)(gg[jj] * dnb[jj] - c2 * xa[jj] * adj);
                }
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, da), rlda, ownrow ? (unsigned)(t * a.row_bytes) + out_col : kOOB, 0, 0);
            } else
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), ry, ownrow ? (unsigned)(t * a.row_bytes) + out_col : kOOB, 0, 0);
            if constexpr (!BWD && BITS) {
                u64 bal[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) bal[jj] = __builtin_amdgcn_ballot_w64((float)r[jj] > 0.f);
                __builtin_amdgcn_raw_buffer_store_b32(lane_word(bal), rba, ownrow ? (unsigned)(t * plane_pitch) + bal_off : kOOB, 0, 0);
            }
            bf16x4 po = o;                                                // what is pooled: y, or (CPL = 2) z
            if constexpr (CPL != 0) {
                // ---- the coupling conv on the finished row: fragment = [prev ch 0-7 | prev ch 8-15 | y ch 0-7 | y ch 8-15] of pixel lr
                *reinterpret_cast<bf16x4*>(ywr) = o;
                __builtin_amdgcn_wave_barrier();
                const u32x4 yb = *reinterpret_cast<const u32x4*>(yrd);
                __builtin_amdgcn_wave_barrier();
                u32x4 bf;
#pragma unroll
                for (int w = 0; w < 4; ++w) bf[w] = lg < 2 ? PV[I % 3][w] : yb[w];
                PV[I % 3] = load_prev(t + 3);
                const f32x4 zacc = mma8(AC, __builtin_bit_cast(bf16x8, bf), cbias);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) po[jj] = (bf16_t)fmaxf(zacc[jj], 0.f);               // MSAU_CONV_RELU_OUT
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, po), rcz, ownrow ? (unsigned)(t * a.row_bytes) + out_col : kOOB, 0, 0);
            }
            if constexpr (PL) {
                // rows t - 1 (even, in prev_out) and t (odd), columns lr and lr ^ 1; the window of the image's last row, when H
                // is odd, sees zeros below (model/model.py:158-160: zero padding).  Values 0 where nothing is stored.
                bf16x4 cur;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) cur[jj] = ownrow && out_ok ? po[jj] : (bf16_t)0.f;
                const bool odd = (t & 1) != 0;                            // wave-uniform (y0 is even)
                const bool last_even = !odd && t == H - 1;                // a window with no second row
                if (odd || last_even) {
                    const bf16x4 top = odd ? prev_out : cur, bot = odd ? cur : bf16x4{};
                    typedef int dw2 __attribute__((ext_vector_type(2)));
                    dw2 st = __builtin_bit_cast(dw2, top), sb = __builtin_bit_cast(dw2, bot), nt, nb;
#pragma unroll
                    for (int w = 0; w < 2; ++w) {
                        nt[w] = __builtin_amdgcn_mov_dpp(st[w], 0xB1, 0xf, 0xf, true);     // quad_perm [1,0,3,2]: column lr ^ 1
                        nb[w] = __builtin_amdgcn_mov_dpp(sb[w], 0xB1, 0xf, 0xf, true);
                    }
                    const bf16x4 ntop = __builtin_bit_cast(bf16x4, nt), nbot = __builtin_bit_cast(bf16x4, nb);
                    bf16x4 best;
                    unsigned idx = 0;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        float bv = (float)top[jj];
                        unsigned bi = 0;
                        const float c1 = (float)ntop[jj], c2 = (float)bot[jj], c3 = (float)nbot[jj];
                        if (c1 > bv) { bv = c1; bi = 1; }
                        if (c2 > bv) { bv = c2; bi = 2; }
                        if (c3 > bv) { bv = c3; bi = 3; }
                        best[jj] = (bf16_t)bv;
                        idx |= bi << (8 * jj);
                    }
                    const int ty = odd ? t - 1 : t;
                    const bool pok = out_ok && !(lr & 1) && ty >= y0 && ty < y1;
                    const unsigned e = (unsigned)(((ty >> 1) * Wo + ((x0 + lr) >> 1)) * 16 + c0);       // elements
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, best), rp, pok ? e * 2 : kOOB, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(idx, ri, pok ? e : kOOB, 0, 0);
                }
                prev_out = cur;
            }
        }
        if constexpr (BWD) {
            cu64p pm = plane_row(pm0, m + 2), pa = plane_row(pa0, t + 2);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) { mm[P][jj] = pm[jj]; ma[P][jj] = pa[jj]; }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    for (int tg = y0 - 2; tg < y1; tg += UNR) {
        step(IC<0>{}, tg);
        step(IC<1>{}, tg);
        step(IC<2>{}, tg);
        step(IC<3>{}, tg);
        step(IC<4>{}, tg);
        step(IC<5>{}, tg);
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Single convolutions of the 8-channel level in the same form (msau_conv2d descriptors; dispatched from conv.hip before the
// tile kernels): stride 1, 8 output channels pixel-pair packed, one or two 8-channel sources, 3x3 / 1x1 / 4x4.  A strip is
// 32 output columns (16 pixel pairs); a pair's window is KW + 1 columns wide:
//   3x3: one k-step per (source, tap row): lane group lg = window column; the B fragment is pixel x0 - pad + 2*lr + lg of
//        the input row -- one 16-byte load per lane, row and source, carried in registers while the row serves as
//        ky = KH-1 .. 0;
//   1x1 over concat(x1, x2) (the coupling conv, model/model.py:143-148): ONE k-step per row: lane group lg = (source lg >> 1,
//        pixel parity lg & 1);
//   4x4 (the end conv, model/model.py:375-376,390, and its data gradient): window columns 0..3 as above plus a second
//        k-step per tap row whose only live group is window column 4 (all lanes load pixel 2*lr + 4; the weights of lane
//        groups 1..3 are zero) -- 8 MFMAs per row instead of the 5 a dense packing needs, but every fragment stays
//        row-pure and the MFMA pipe is nowhere near the bound.
// Epilogue operands (MASK_A, ADD, ACCUM, MASK_B: 8 bytes per lane in the result layout) are loaded two rows ahead.
// MSAU_CONV_LRN: the 8 channels of a pixel sit in lanes l and l ^ 16; same arithmetic as the stand-alone pass.
struct RowConvArgs {
    msau_conv_desc d;
    int nstrips, nseg, SH, ntasks, tasks_per_xcd;
    int row_bytes, wrow, wchunk;                  // bytes of an image row; elements per packed weight row / per source chunk
    unsigned img_bytes;
};

// WG (MSAU_CONV_WGRAD, the 1x1 two-output instance = the data gradient of the coupling conv z = ReLU(Wc concat(prev, y) + bc),
// model/model.py:143-148,246-252): the conv's weight gradient in the same launch.  g is this launch's input row (lanes of groups 0, 1
// hold a pixel each); y is already here as the ReLU-mask operand of the second output (the forward tensor itself, 8 bytes per lane in
// the result layout); prev costs one more 16-byte load per lane and row.  The three rows go to LDS as [pixel][8 channels] and come
// back transposed (ds_read_b64_tr_b16: pixels become the MFMA's K): per 32 pixels ONE MFMA gives
//     D[(s, ci)][co] += sum_px  x_s[ci](px) * g[co](px)          (s = 0: prev, 1: y)
// -- the 32 bytes a lane group reads per pixel are chunks {ch 0-3, ch 4-7} of source 0 and of source 1 instead of two neighbouring
// pixels of one source (rowwgrad8_kernel's shifted windows) -- and one more against ones the bias gradient.  The four waves of a
// workgroup add up in LDS in a fixed order and write ONE slab [2 chunks][8][16] (kext 16, ones column 8 of chunk 0): the layout
// msau_wgrad_reduce expects for wgrad_lean<C8, CO8, K1>, whose launch (66 MB at the bench size) disappears.
constexpr int WGC_ROW = 32 * 16;                  // a staged row: 32 pixels x 8 bf16

template <int NS, int KH, int KW, int EPI, int EPI2 = 0, bool WG = false>
__global__ __launch_bounds__(256) void rowconv8_kernel(const RowConvArgs a) {
    constexpr bool DOUT = (EPI & MSAU_CONV_DOUT) != 0;                     // two outputs (stored weight rows 0..7 -> y, 8..15 -> y2)
    static_assert(!WG || (DOUT && KW == 1 && NS == 1 && EPI == MSAU_CONV_DOUT && EPI2 == MSAU_CONV_MASK_B), "the rider belongs to the coupling conv's data gradient");
    __shared__ __align__(16) unsigned char wsm[WG ? 4 * 3 * WGC_ROW + 64 + 2 * 8 * 16 * 4 : 16];
    constexpr bool K1D = KW == 1 && NS == 2;                               // 1x1 over concat(x1, x2)
    static_assert((KW == 3 && (NS == 1 || NS == 2)) || (K1D && KH == 1) || (KW == 1 && NS == 1 && KH == 1 && DOUT) || (KW == 4 && NS == 1),
                  "instances: 3x3, 1x1 dual, 1x1 two-output, 4x4");
    static_assert(!DOUT || (!(EPI & ~(MSAU_CONV_DOUT | MSAU_CONV_ADD | MSAU_CONV_ACCUM | MSAU_CONV_MASK_B)) && !(EPI2 & ~(MSAU_CONV_ACCUM | MSAU_CONV_MASK_B)) &&
                            NS == 1 && KW != 4), "two outputs: EPI = flags of y, EPI2 = flags2 of y2");
    static_assert(DOUT || EPI2 == 0, "EPI2 belongs to the second output");
    constexpr int XL = KW == 4 ? 2 : 1;                                    // loads per lane, row and source
    constexpr int NSL = K1D ? 1 : NS;                                      // source "slots" per row (1x1 dual: the lane picks its source)
#ifndef MSAU_ROWCONV_PF
#define MSAU_ROWCONV_PF 3
#endif
    constexpr int NXR = KH == 4 ? 8 : KH + MSAU_ROWCONV_PF;                // rows in registers: KH in use, the rest in flight (even: operand slots alternate)
    constexpr bool HAS_MA = (EPI & MSAU_CONV_MASK_A) != 0, HAS_ADD = (EPI & MSAU_CONV_ADD) != 0, HAS_ACC = (EPI & MSAU_CONV_ACCUM) != 0,
                   HAS_MB = (EPI & MSAU_CONV_MASK_B) != 0, RELU_OUT = (EPI & MSAU_CONV_RELU_OUT) != 0, LRN = (EPI & MSAU_CONV_LRN) != 0;
    const msau_conv_desc& d = a.d;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tloc = (blockIdx.x >> 3) * 4 + wave;
    const int task = (blockIdx.x & 7) * a.tasks_per_xcd + tloc;
    const bool live = tloc < a.tasks_per_xcd && task < a.ntasks;          // wave-uniform
    if constexpr (!WG) { if (!live) return; }                             // (WG: idle waves wait at the slab reduction)
    f32x4 wacc = {0.f, 0.f, 0.f, 0.f}, waccb = {0.f, 0.f, 0.f, 0.f};
    if (live) {
    const int t1 = task / a.nstrips, strip = task - t1 * a.nstrips;
    const int b = t1 / a.nseg, seg = t1 - b * a.nseg;
    const int H = d.Hout, W = d.Wout;
    const int x0 = strip * 32;
    const int y0 = seg * a.SH, y1 = min(H, y0 + a.SH);
    const int lr = lane & 15, lg = lane >> 4;
    const int c0 = (lg & 1) * 4;
    const int j = 2 * lr + (lg >> 1);                                     // output column of this lane's result

    // ---- A fragments
    bf16x8 A[NSL][KH][XL];
    bf16x8 A2[DOUT ? KH : 1];
    {
        const int co = lr & 7, pp = lr >> 3;
        const bf16_t* w = static_cast<const bf16_t*>(d.wpack) + co * a.wrow;
#pragma unroll
        for (int s = 0; s < NSL; ++s)
#pragma unroll
            for (int ky = 0; ky < KH; ++ky) {
                if constexpr (K1D) {
                    const bool ok = (lg & 1) == pp;                        // this lane group's pixel is the row's pixel
                    A[s][ky][0] = ok ? load8<bf16_t>(w + (lg >> 1) * a.wchunk) : zero8<bf16_t>();
                } else {
                    const int kx = lg - pp;
                    const bool ok = kx >= 0 && kx < KW;
                    A[s][ky][0] = ok ? load8<bf16_t>(w + s * a.wchunk + (ky * KW + (ok ? kx : 0)) * 8) : zero8<bf16_t>();
                    if constexpr (KW == 4) {
                        const bool ok4 = lg == 0 && pp == 1;               // window column 4 = tap column 3 of the odd pixel
                        A[s][ky][1] = ok4 ? load8<bf16_t>(w + s * a.wchunk + (ky * KW + 3) * 8) : zero8<bf16_t>();
                    }
                    if constexpr (DOUT) A2[ky] = ok ? load8<bf16_t>(w + 8 * a.wrow + (ky * KW + (ok ? kx : 0)) * 8) : zero8<bf16_t>();
                }
            }
    }
    f32x4 bias = {0.f, 0.f, 0.f, 0.f}, bias2 = {0.f, 0.f, 0.f, 0.f};
    if (d.bias) bias = *reinterpret_cast<const f32x4*>(d.bias + c0);
    if (DOUT && d.bias) bias2 = *reinterpret_cast<const f32x4*>(d.bias + 8 + c0);
    constexpr bool ACC2 = (EPI2 & MSAU_CONV_ACCUM) != 0, MB2 = (EPI2 & MSAU_CONV_MASK_B) != 0;       // the second output's operands

    const long long img = (long long)b * a.img_bytes;
    const __amdgpu_buffer_rsrc_t rx1 = rsrc_of(static_cast<const char*>(d.x1) + img, a.img_bytes);
    const __amdgpu_buffer_rsrc_t rx2 = rsrc_of(NS == 2 && !K1D ? static_cast<const char*>(d.x2) + img : nullptr, NS == 2 && !K1D ? a.img_bytes : 0u);
    const __amdgpu_buffer_rsrc_t ry = rsrc_of(static_cast<char*>(d.y) + img, a.img_bytes);
    const __amdgpu_buffer_rsrc_t ry2 = rsrc_of(LRN || DOUT ? static_cast<char*>(d.y2) + img : nullptr, LRN || DOUT ? a.img_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rma = rsrc_of(HAS_MA ? static_cast<const char*>(d.mask_a) + img : nullptr, HAS_MA ? a.img_bytes : 0u);
    const __amdgpu_buffer_rsrc_t radd = rsrc_of(HAS_ADD ? static_cast<const char*>(d.add) + img : nullptr, HAS_ADD ? a.img_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rmb = rsrc_of(HAS_MB ? static_cast<const char*>(d.mask_b) + img : nullptr, HAS_MB ? a.img_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rmb2 = rsrc_of(MB2 ? static_cast<const char*>(d.mask_b2) + img : nullptr, MB2 ? a.img_bytes : 0u);

    // fragment loads of input row r
    const int lxm = K1D ? x0 + 2 * lr + (lg & 1) : x0 - d.pad_l + 2 * lr + lg;
    const int lxe = x0 - d.pad_l + 2 * lr + 4;
    const unsigned lcol_m = (unsigned)lxm < (unsigned)W && !(KW == 1 && !K1D && lg >= 2) ? (unsigned)(lxm * 16) : kOOB;   // (1x1 single source: groups 2,3 carry nothing)
    const unsigned lcol_e = (unsigned)lxe < (unsigned)W ? (unsigned)(lxe * 16) : kOOB;
    const int rlast = min(H - 1, y1 - 1 - d.pad_t + KH - 1);
    const bool b075 = d.lrn_beta == 0.75f;
    u32x4 X[NSL][NXR][XL];
    // 1x1 over two sources: lanes of groups 0,1 read x1, of groups 2,3 x2 -- ONE load with a per-lane base address.  No halo,
    // so the only invalid lanes are pixels beyond the image width: they read (and never store) the pixel at the image edge.
    const char* src1x1 = nullptr;
    if constexpr (K1D) src1x1 = static_cast<const char*>(lg < 2 ? d.x1 : d.x2) + img + (lxm < W ? lxm : W - 1) * 16;
    auto load_row = [&](int r, auto slot) {
        constexpr int S = decltype(slot)::value;
        const bool ok = r >= 0 && r <= rlast;
        const unsigned ro = (unsigned)(r * a.row_bytes);
        if constexpr (K1D) {
            X[0][S][0] = *reinterpret_cast<const u32x4*>(src1x1 + (long long)(ok ? r : y0) * a.row_bytes);
        } else {
            X[0][S][0] = __builtin_amdgcn_raw_buffer_load_b128(rx1, ok ? ro + lcol_m : kOOB, 0, 0);
            if constexpr (KW == 4) X[0][S][1] = __builtin_amdgcn_raw_buffer_load_b128(rx1, ok ? ro + lcol_e : kOOB, 0, 0);
            if constexpr (NS == 2) X[1][S][0] = __builtin_amdgcn_raw_buffer_load_b128(rx2, ok ? ro + lcol_m : kOOB, 0, 0);
        }
    };
    // epilogue operands of output row t (result layout), two rows ahead
    const bool out_ok = x0 + j < W;
    const unsigned out_col = out_ok ? (unsigned)((x0 + j) * 16 + c0 * 2) : kOOB;
    u32x2 OPA[2], OPD[2], OPB[2], OPY[2], OPB2[2], OPY2[2];
    auto load_ops = [&](int t, auto slot) {
        constexpr int P = decltype(slot)::value;
        const unsigned o = (t >= y0 && t < y1) ? (unsigned)(t * a.row_bytes) + out_col : kOOB;
        if constexpr (ACC2) OPY2[P] = __builtin_amdgcn_raw_buffer_load_b64(ry2, o, 0, 0);
        if constexpr (MB2) OPB2[P] = __builtin_amdgcn_raw_buffer_load_b64(rmb2, o, 0, 0);
        if constexpr (HAS_MA) OPA[P] = __builtin_amdgcn_raw_buffer_load_b64(rma, o, 0, 0);
        if constexpr (HAS_ADD) OPD[P] = __builtin_amdgcn_raw_buffer_load_b64(radd, o, 0, 0);
        if constexpr (HAS_ACC) OPY[P] = __builtin_amdgcn_raw_buffer_load_b64(ry, o, 0, 0);
        if constexpr (HAS_MB) OPB[P] = __builtin_amdgcn_raw_buffer_load_b64(rmb, o, 0, 0);
    };

    // ---- WG: prev rows in the ring beside the g rows; the three staging rows of this wave; the transposed-read addresses
    u32x4 PVX[WG ? NXR : 1];
    const __amdgpu_buffer_rsrc_t rwp = rsrc_of(WG ? static_cast<const char*>(d.wg_x1) + img : nullptr, WG ? a.img_bytes : 0u);
    auto load_prev = [&](int r, auto slot) {
        constexpr int S = decltype(slot)::value;
        if constexpr (WG) PVX[S] = __builtin_amdgcn_raw_buffer_load_b128(rwp, r >= y0 && r < y1 ? (unsigned)(r * a.row_bytes) + lcol_m : kOOB, 0, 0);
    };
    unsigned char* const wbase = wsm + wave * 3 * WGC_ROW;                 // [g | prev | y]
    unsigned char* const wzero = wsm + 4 * 3 * WGC_ROW;                    // 16 zero bytes (+ pad): MFMA rows 8..15 of the g operand
    const int wpx = (2 * lr + lg) * 16;                                    // groups 0, 1: this lane's pixel of the row
    const int wry = j * 16 + c0 * 2;                                       // result layout slot (the mask operand's)
    const int trp = 8 * (lane >> 4) + ((lane & 15) >> 2), trc = lane & 3;  // transposed read: pixel trp (+ 4), chunk trc of the 32-byte window
    const unsigned char* const tr_x = wbase + (trc < 2 ? WGC_ROW : 2 * WGC_ROW) + trp * 16 + (trc & 1) * 8;
    const unsigned char* const tr_g = trc < 2 ? wbase + trp * 16 + (trc & 1) * 8 : wzero + (trc & 1) * 8;
    const int tr_g_hi = trc < 2 ? 4 * 16 : 0;
    auto trfrag = [&](const unsigned char* p, int hi) -> bf16x8 {
        const bf16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)p);
        const bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(p + hi));
        return __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    const u32x4 wones_bits = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    if constexpr (WG) {
        if (lane < 4) reinterpret_cast<unsigned*>(wzero)[lane] = 0u;       // (every live wave writes the same zeros)
        __builtin_amdgcn_wave_barrier();
    }

    // prologue: rows r0 .. r0 + NXR - 2 of the first output row (r0 = y0 - pad_t); operands of rows y0, y0 + 1
    {
        const int r0 = y0 - d.pad_t;
        [&]<int... K>(std::integer_sequence<int, K...>) { (load_row(r0 + K, IC<K>{}), ...); }(std::make_integer_sequence<int, NXR - 1>{});
        [&]<int... K>(std::integer_sequence<int, K...>) { (load_prev(r0 + K, IC<K>{}), ...); }(std::make_integer_sequence<int, NXR - 1>{});
    }
    load_ops(y0, IC<0>{});
    load_ops(y0 + 1, IC<1>{});

    auto step = [&](auto ic, const int tg) {
        constexpr int I = decltype(ic)::value, P = I & 1;
        const int t = tg + I;
        load_row(t - d.pad_t + NXR - 1, IC<(I + NXR - 1) % NXR>{});
        load_prev(t - d.pad_t + NXR - 1, IC<(I + NXR - 1) % NXR>{});
        f32x4 acc = bias;
#pragma unroll
        for (int ky = 0; ky < KH; ++ky)
#pragma unroll
            for (int s = 0; s < NSL; ++s)
#pragma unroll
                for (int e = 0; e < XL; ++e) acc = mma8(A[s][ky][e], __builtin_bit_cast(bf16x8, X[s][(I + ky) % NXR][e]), acc);
        const unsigned oo = (t >= y0 && t < y1) ? (unsigned)(t * a.row_bytes) + out_col : kOOB;
        if constexpr (DOUT) {
            f32x4 acc_b = bias2;
#pragma unroll
            for (int ky = 0; ky < KH; ++ky) acc_b = mma8(A2[ky], __builtin_bit_cast(bf16x8, X[0][(I + ky) % NXR][0]), acc_b);
            bf16x4 o1, o2;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                float v1 = acc[jj];                                        // same order as the tile kernel: ADD, ACCUM, MASK_B
                if constexpr (HAS_ADD) v1 += (float)__builtin_bit_cast(bf16x4, OPD[P])[jj];
                if constexpr (HAS_ACC) v1 += (float)__builtin_bit_cast(bf16x4, OPY[P])[jj];
                if constexpr (HAS_MB) v1 = (float)__builtin_bit_cast(bf16x4, OPB[P])[jj] > 0.f ? v1 : 0.f;
                float v2 = acc_b[jj];
                if constexpr (ACC2) v2 += (float)__builtin_bit_cast(bf16x4, OPY2[P])[jj];
                if constexpr (MB2) v2 = (float)__builtin_bit_cast(bf16x4, OPB2[P])[jj] > 0.f ? v2 : 0.f;
                o1[jj] = (bf16_t)v1;
                o2[jj] = (bf16_t)v2;
            }
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o1), ry, oo, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o2), ry2, oo, 0, 0);
            if constexpr (WG) {
                // rows of g, prev, y of output row t -> LDS as [pixel][8 channels] (rows outside [y0, y1) were loaded as zeros)
                if (lg < 2) {
                    *reinterpret_cast<u32x4*>(wbase + wpx) = X[0][I % NXR][0];
                    *reinterpret_cast<u32x4*>(wbase + WGC_ROW + wpx) = PVX[I % NXR];
                }
                *reinterpret_cast<u32x2*>(wbase + 2 * WGC_ROW + wry) = OPB2[P];
                __builtin_amdgcn_wave_barrier();
                const bf16x8 xf = trfrag(tr_x, 4 * 16);                    // rows (source, ci), k = 8 pixels of this lane group
                const bf16x8 gf = trfrag(tr_g, tr_g_hi);                   // rows co (8..15: zeros)
                __builtin_amdgcn_wave_barrier();
                wacc = mma8(xf, gf, wacc);
                waccb = mma8(__builtin_bit_cast(bf16x8, wones_bits), gf, waccb);
            }
            load_ops(t + 2, IC<P>{});
            __builtin_amdgcn_sched_barrier(0);
            return;
        }
        f32x4 v = acc;
        if constexpr (HAS_MA) {
            const bf16x4 mk = __builtin_bit_cast(bf16x4, OPA[P]);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) v[jj] = (float)mk[jj] > 0.f ? v[jj] : 0.f;
        }
        if constexpr (HAS_ADD) {
            const bf16x4 ad = __builtin_bit_cast(bf16x4, OPD[P]);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) v[jj] += (float)ad[jj];
        }
        if constexpr (HAS_ACC) {
            const bf16x4 yo = __builtin_bit_cast(bf16x4, OPY[P]);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) v[jj] += (float)yo[jj];
        }
        if constexpr (RELU_OUT) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) v[jj] = fmaxf(v[jj], 0.f);
        }
        if constexpr (HAS_MB) {
            const bf16x4 mk = __builtin_bit_cast(bf16x4, OPB[P]);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) v[jj] = (float)mk[jj] > 0.f ? v[jj] : 0.f;
        }
        bf16x4 o;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) o[jj] = (bf16_t)v[jj];
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), ry, oo, 0, 0);
        if constexpr (LRN) {
            // y2 = y * (k + alpha/n * sum over [c - 4, c + 3] of y^2)^-beta from the rounded y (layers.py:145,161-162); lane l holds
            // channels 4h..4h+3 (h = lg & 1), lane l ^ 16 the other half; window sums = differences of exclusive prefix sums
            float x[4], Pf[4], run = 0.f;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) { x[jj] = (float)o[jj]; run += x[jj] * x[jj]; Pf[jj] = run; }
            const int h = lg & 1;
            const float other = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, run), 0x401F));   // lane ^ 16
            const float inc = h ? run + other : run;
            const float E = inc - run;
            const float tot = h ? inc : other + run;
            bf16x4 o2;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float Xe = E + (jj ? Pf[jj - 1] : 0.f);
                const float Xo = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, Xe), 0x401F));
                const float win = h ? tot - Xo : Xo;
                const float dd = d.lrn_k + d.lrn_alpha_over_n * win;
                float dnb;
                if (b075) { const float r = __builtin_amdgcn_rsqf(dd); dnb = r * __builtin_amdgcn_sqrtf(r); }
                else dnb = __expf(-d.lrn_beta * __logf(dd));
                o2[jj] = (bf16_t)(x[jj] * dnb);
            }
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o2), ry2, oo, 0, 0);
        }
        load_ops(t + 2, IC<P>{});
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int tg = y0; tg < y1; tg += NXR)
        [&]<int... K>(std::integer_sequence<int, K...>) { (step(IC<K>{}, tg), ...); }(std::make_integer_sequence<int, NXR>{});
    }   // live
    if constexpr (WG) {
        // ---- the four waves' sums -> one slab [chunk = source][co][16], fixed order
        float* slab = reinterpret_cast<float*>(wsm + 4 * 3 * WGC_ROW + 64);
        for (int i = threadIdx.x; i < 2 * 8 * 16; i += 256) slab[i] = 0.f;
        __syncthreads();
        const int n = lane & 15, kg = lane >> 4;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (wave == w && n < 8) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int row = 4 * kg + jj;                           // (source, ci)
                    slab[(row >> 3) * 128 + n * 16 + (row & 7)] += wacc[jj];
                }
                if (kg == 0) slab[n * 16 + 8] += waccb[0];                 // the ones column: the bias gradient
            }
            __syncthreads();
        }
        float* out = a.d.wg_slabs + (size_t)blockIdx.x * (2 * 8 * 16);
        for (int i = threadIdx.x; i < 2 * 8 * 16; i += 256) out[i] = slab[i];
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// The level-1 -> level-0 transposed conv (Deconv2DBnLrnDrop, model/layers/layers.py:249-250: ConvTranspose2d(k 3, stride 2, padding 1,
// output_padding from output_size), 16 -> 8 channels) in the same form.  The tile kernel runs it as a 3x3 conv over the zero-stuffed
// input (msau_conv2d, ups = 2): three of four products multiply a stuffed zero and it staged 18 x 18 stuffed tiles per 16 x 16
// outputs -- 14.8 us for 33 MB back to back, half the rate of the level's other launches.  Here only the live taps exist.  With the
// (flipped) image msau_pack_params writes for the ups = 2 conv, out[Y][X] = sum W'[ky][kx] s[Y + ky - 1][X + kx - 1], s[2y][2x] = in[y][x]:
//     Y = 2y    : ky = 1 on input row y;          Y = 2y + 1 : ky = 0 on row y and ky = 2 on row y + 1        (the same in x)
// so an output pixel PAIR (2x, 2x + 1) reads input pixels x and x + 1: K = (dx, 16 channels) = 32, one k-step per live tap row.
// MFMA rows = (output parity px, co), columns = 16 input columns of a strip; A[(px, co)][(dx, ci)] = W'[ky][kx(px, dx)][co][ci] with
// kx(0, 0) = 1, kx(1, 0) = 0, kx(1, 1) = 2 and (0, 1) empty; the B fragment "pixel x + dx, channel group cg" (lane group lg = 2 dx + cg)
// is one 16-byte global load per lane and INPUT row -- no LDS, no staging -- and serves the two output rows it feeds from registers:
// one MFMA for an even output row, two for an odd one.  A result lane holds 4 channels of one output pixel: 8-byte stores, 512
// contiguous bytes per wave-instruction.  Tasks are (image, segment of output rows, strip of 16 input = 32 output columns).
struct RowDeconvArgs {
    msau_conv_desc d;
    int nstrips, nseg, SH, ntasks, tasks_per_xcd;
    int in_row_bytes, out_row_bytes;
    unsigned in_img_bytes, out_img_bytes;
};

__global__ __launch_bounds__(256) void rowdeconv8_kernel(const RowDeconvArgs a) {
    const msau_conv_desc& d = a.d;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tloc = (blockIdx.x >> 3) * 4 + wave;
    const int task = (blockIdx.x & 7) * a.tasks_per_xcd + tloc;
    if (tloc >= a.tasks_per_xcd || task >= a.ntasks) return;
    const int t1 = task / a.nstrips, strip = task - t1 * a.nstrips;
    const int b = t1 / a.nseg, seg = t1 - b * a.nseg;
    const int Hin = d.Hin, Win = d.Win, Hout = d.Hout, Wout = d.Wout;
    const int xh0 = strip * 16;                                           // first input column of the strip
    const int Y0 = seg * a.SH, Y1 = min(Hout, Y0 + a.SH);                 // output rows (SH is a multiple of 8: Y0 is even)
    const int lr = lane & 15, lg = lane >> 4;
    const int c0 = (lg & 1) * 4;

    // ---- A fragments of the three tap rows
    bf16x8 A[3];
    {
        const int co = lr & 7, px = lr >> 3, dx = lg >> 1, cg = lg & 1;
        const int kx = px == 0 ? (dx == 0 ? 1 : -1) : (dx == 0 ? 0 : 2);
        const bf16_t* w = static_cast<const bf16_t*>(d.wpack) + co * 160 + cg * 8;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) A[ky] = kx >= 0 ? load8<bf16_t>(w + (ky * 3 + kx) * 16) : zero8<bf16_t>();
    }
    f32x4 bias = {0.f, 0.f, 0.f, 0.f};
    if (d.bias) bias = *reinterpret_cast<const f32x4*>(d.bias + c0);

    const __amdgpu_buffer_rsrc_t rx = rsrc_of(static_cast<const char*>(d.x1) + (long long)b * a.in_img_bytes, a.in_img_bytes);
    const __amdgpu_buffer_rsrc_t ry = rsrc_of(static_cast<char*>(d.y) + (long long)b * a.out_img_bytes, a.out_img_bytes);
    const int xi = xh0 + lr + (lg >> 1);
    const unsigned lcol = xi < Win ? (unsigned)(xi * 32 + (lg & 1) * 16) : kOOB;             // (beyond the image: the stuffed zeros)
    const int ylast = min(Hin - 1, Y1 >> 1);                              // last input row this task reads
    auto load_row = [&](int r) -> u32x4 {
        return __builtin_amdgcn_raw_buffer_load_b128(rx, r <= ylast ? (unsigned)(r * a.in_row_bytes) + lcol : kOOB, 0, 0);
    };
    const int Xo = 2 * (xh0 + lr) + (lg >> 1);
    const unsigned out_col = Xo < Wout ? (unsigned)(Xo * 16 + c0 * 2) : kOOB;
    auto store_row = [&](int Y, const f32x4& acc) {
        bf16x4 o;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) o[jj] = (bf16_t)acc[jj];
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), ry, Y < Y1 ? (unsigned)(Y * a.out_row_bytes) + out_col : kOOB, 0, 0);
    };

    // X[k]: input rows in fragment layout; at iteration I (input row y) rows y, y + 1 are X[I % 4], X[(I + 1) % 4], row y + 3 is requested
    u32x4 X[4];
    const int yh0 = Y0 >> 1;
#pragma unroll
    for (int k = 0; k < 3; ++k) X[k] = load_row(yh0 + k);
    auto step = [&](auto ic, const int yg) {
        constexpr int I = decltype(ic)::value;
        const int y = yg + I;
        X[(I + 3) % 4] = load_row(y + 3);
        f32x4 e = mma8(A[1], __builtin_bit_cast(bf16x8, X[I % 4]), bias);                    // Y = 2y
        f32x4 o = mma8(A[0], __builtin_bit_cast(bf16x8, X[I % 4]), bias);                    // Y = 2y + 1
        o = mma8(A[2], __builtin_bit_cast(bf16x8, X[(I + 1) % 4]), o);
        store_row(2 * y, e);
        store_row(2 * y + 1, o);
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int yg = yh0; 2 * yg < Y1; yg += 4) {
        step(IC<0>{}, yg);
        step(IC<1>{}, yg);
        step(IC<2>{}, yg);
        step(IC<3>{}, yg);
    }
}

// The level-2 -> level-1 transposed conv, 32 -> 16 channels, the same way: rows = 16 output channels, columns = 16 input columns,
// K = 32 input channels = one k-step per live (tap row, tap column).  Even / odd output columns need different weights, so they are two
// accumulators (X = 2x: kx 1 on in[x]; X = 2x + 1: kx 0 on in[x] and kx 2 on in[x + 1]): 3 MFMAs for an even output row, 6 for an odd one,
// on two 16-byte loads per lane and INPUT row (pixels x and x + 1, channel group lg).  A result lane holds 4 channels of output
// pixels 2x and 2x + 1 of a row: two 8-byte stores.
__global__ __launch_bounds__(256) void rowdeconv16_kernel(const RowDeconvArgs a) {
    const msau_conv_desc& d = a.d;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tloc = (blockIdx.x >> 3) * 4 + wave;
    const int task = (blockIdx.x & 7) * a.tasks_per_xcd + tloc;
    if (tloc >= a.tasks_per_xcd || task >= a.ntasks) return;
    const int t1 = task / a.nstrips, strip = task - t1 * a.nstrips;
    const int b = t1 / a.nseg, seg = t1 - b * a.nseg;
    const int Hin = d.Hin, Win = d.Win, Hout = d.Hout, Wout = d.Wout;
    const int xh0 = strip * 16;
    const int Y0 = seg * a.SH, Y1 = min(Hout, Y0 + a.SH);
    const int lr = lane & 15, lg = lane >> 4;

    bf16x8 A[3][3];
    {
        const bf16_t* w = static_cast<const bf16_t*>(d.wpack) + lr * 288 + lg * 8;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) A[ky][kx] = load8<bf16_t>(w + (ky * 3 + kx) * 32);
    }
    f32x4 bias = {0.f, 0.f, 0.f, 0.f};
    if (d.bias) bias = *reinterpret_cast<const f32x4*>(d.bias + lg * 4);

    const __amdgpu_buffer_rsrc_t rx = rsrc_of(static_cast<const char*>(d.x1) + (long long)b * a.in_img_bytes, a.in_img_bytes);
    const __amdgpu_buffer_rsrc_t ry = rsrc_of(static_cast<char*>(d.y) + (long long)b * a.out_img_bytes, a.out_img_bytes);
    const int xi = xh0 + lr;
    const unsigned lcol0 = xi < Win ? (unsigned)(xi * 64 + lg * 16) : kOOB;
    const unsigned lcol1 = xi + 1 < Win ? (unsigned)((xi + 1) * 64 + lg * 16) : kOOB;             // (beyond the image: the stuffed zeros)
    const int ylast = min(Hin - 1, Y1 >> 1);
    auto load_row = [&](int r, u32x4& x0, u32x4& x1) {
        const unsigned ro = r <= ylast ? (unsigned)(r * a.in_row_bytes) : kOOB;
        x0 = __builtin_amdgcn_raw_buffer_load_b128(rx, r <= ylast ? ro + lcol0 : kOOB, 0, 0);
        x1 = __builtin_amdgcn_raw_buffer_load_b128(rx, r <= ylast ? ro + lcol1 : kOOB, 0, 0);
    };
    const unsigned oc0 = 2 * xi < Wout ? (unsigned)(2 * xi * 32 + lg * 8) : kOOB;
    const unsigned oc1 = 2 * xi + 1 < Wout ? (unsigned)((2 * xi + 1) * 32 + lg * 8) : kOOB;
    auto store_px = [&](int Y, const f32x4& acc, unsigned oc) {
        bf16x4 o;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) o[jj] = (bf16_t)acc[jj];
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), ry, Y < Y1 ? (unsigned)(Y * a.out_row_bytes) + oc : kOOB, 0, 0);
    };

    u32x4 X0[4], X1[4];
    const int yh0 = Y0 >> 1;
#pragma unroll
    for (int k = 0; k < 3; ++k) load_row(yh0 + k, X0[k], X1[k]);
    auto step = [&](auto ic, const int yg) {
        constexpr int I = decltype(ic)::value;
        const int y = yg + I;
        load_row(y + 3, X0[(I + 3) % 4], X1[(I + 3) % 4]);
        const bf16x8 c0 = __builtin_bit_cast(bf16x8, X0[I % 4]), c1 = __builtin_bit_cast(bf16x8, X1[I % 4]);
        const bf16x8 n0 = __builtin_bit_cast(bf16x8, X0[(I + 1) % 4]), n1 = __builtin_bit_cast(bf16x8, X1[(I + 1) % 4]);
        f32x4 e0 = mma8(A[1][1], c0, bias);                                                    // (2y, 2x)
        f32x4 e1 = mma8(A[1][0], c0, bias);                                                    // (2y, 2x + 1)
        e1 = mma8(A[1][2], c1, e1);
        f32x4 o0 = mma8(A[0][1], c0, bias);                                                    // (2y + 1, 2x)
        o0 = mma8(A[2][1], n0, o0);
        f32x4 o1 = mma8(A[0][0], c0, bias);                                                    // (2y + 1, 2x + 1)
        o1 = mma8(A[0][2], c1, o1);
        o1 = mma8(A[2][0], n0, o1);
        o1 = mma8(A[2][2], n1, o1);
        store_px(2 * y, e0, oc0);
        store_px(2 * y, e1, oc1);
        store_px(2 * y + 1, o0, oc0);
        store_px(2 * y + 1, o1, oc1);
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int yg = yh0; 2 * yg < Y1; yg += 4) {
        step(IC<0>{}, yg);
        step(IC<1>{}, yg);
        step(IC<2>{}, yg);
        step(IC<3>{}, yg);
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Weight gradient of the 8 -> 8 3x3 convs in the same form (msau_conv2d_wgrad descriptors; dispatched from conv_wgrad.hip).
// The tile kernel re-reads tile halos (PMC, profiles/r03_traffic.json: x1.49 the algorithmic bytes); here a wave walks a
// 30-column strip and every row of x and g is read exactly once.  Pixels are the MFMA's K: per 32 pixels of a row and per
// kernel row ky ONE MFMA produces all three horizontal taps at once --
//     D[(a, ci)][(b, co)] = sum_x  x[ci](y + ky - 1, x + a) * g[co](y, x + b)        a, b in {0, 1}
// is the tap kx - 1 = a - b: (0,1) -> kx 0, (0,0) -> kx 1, (1,0) -> kx 2 ((1,1) repeats kx 1 and is dropped).  With the row
// in LDS as [pixel][8 channels], the operand "8 consecutive pixels of channel c, shifted by a" is a 4 x 16 block whose 16
// columns are the 32 contiguous bytes from pixel p: ds_read_b64_tr_b16 delivers it transposed, the shift is free.  g is zero
// outside the strip's own 30 columns, so both shifted windows count every own pixel exactly once.  The x fragment of a row
// serves ky = 2, 1, 0 of three consecutive output rows from registers.  The bias gradient is a fourth MFMA against ones.
// Persistent workgroups (one per slab, the plan's nslabs): accumulators live across tasks, the four waves add up in LDS in
// a fixed order and write the slab in the layout msau_wgrad_reduce expects ([co][tap * 8 + ci], ones column 72).
struct RowWgradArgs {
    msau_wgrad_desc d;
    int nstrips, nseg, SH, ntasks;
    int row_bytes, kext;
    unsigned img_bytes;
};


constexpr int WG_WAVES = 8;                                               // waves per workgroup = tasks per slab and round

// KS = 4 (the end conv, model/model.py:375-376,390; SAME padding 1 before / 2 after): the horizontal taps kx - 1 = -1 .. 2 need a
// second x fragment two pixels further on (a' = 2: with b = 0 it is kx = 3; its other three quadrants repeat taps and are dropped);
// an x row serves four output rows.  Slab: [co][(ky * 4 + kx) * 8 + ci], ones column 128, 144 columns.
template <bool RELU_IN, int KS = 3>
__global__ __launch_bounds__(64 * WG_WAVES) void rowwgrad8_kernel(const RowWgradArgs a) {
    static_assert(KS == 3 || KS == 4, "3x3 and 4x4");
    constexpr int NF = KS == 4 ? 2 : 1;                                   // x fragments per row
    constexpr int KEXT = KS == 4 ? 144 : 80, ONESC = KS * KS * 8;
    constexpr int XL = KS == 4 ? 36 : 34;                                 // lanes that stage x
    __shared__ __align__(16) unsigned char smem[WG_WAVES * 2 * WG_ROW + 8 * KEXT * 4];
    const msau_wgrad_desc& d = a.d;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char* xbuf = smem + wave * 2 * WG_ROW;
    unsigned char* gbuf = xbuf + WG_ROW;
    float* slab = reinterpret_cast<float*>(smem + WG_WAVES * 2 * WG_ROW);
    const int H = d.Hout, W = d.Wout;
    // zero the staging rows once: pixels of g outside the own columns stay zero for the whole kernel
    for (int i = lane; i < 2 * WG_ROW / 16; i += 64) *reinterpret_cast<u32x4*>(xbuf + i * 16) = u32x4{0u, 0u, 0u, 0u};
    __builtin_amdgcn_wave_barrier();
    // transposed-read address of this lane: 16-lane group kg = lane >> 4 takes pixels 8 kg + 4 h + q, q = (lane & 15) >> 2,
    // chunk p = lane & 3 of the 32 bytes that start at that pixel
    const int tr_off = (8 * (lane >> 4) + ((lane & 15) >> 2)) * 16 + (lane & 3) * 8;
    auto frag_of = [&](const unsigned char* buf) -> bf16x8 {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(buf + tr_off));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(buf + tr_off + 4 * 16));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    const u32x4 ones_bits = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    const bf16x8 ONES = __builtin_bit_cast(bf16x8, ones_bits);
    f32x4 acc[KS][NF], accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < KS; ++k)
#pragma unroll
        for (int f = 0; f < NF; ++f) acc[k][f] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int gw = blockIdx.x * WG_WAVES + wave, nw = gridDim.x * WG_WAVES;
    for (int task = gw; task < a.ntasks; task += nw) {                    // wave-uniform
        const int t1 = task / a.nstrips, strip = task - t1 * a.nstrips;
        const int b = t1 / a.nseg, seg = t1 - b * a.nseg;
        const int x0 = strip * 30;
        const int y0 = seg * a.SH, y1 = min(H, y0 + a.SH);
        const long long img = (long long)b * a.img_bytes;
        const __amdgpu_buffer_rsrc_t rx = rsrc_of(static_cast<const char*>(d.x1) + img, a.img_bytes);
        const __amdgpu_buffer_rsrc_t rg = rsrc_of(static_cast<const char*>(d.g) + img, a.img_bytes);
        // x: lanes 0..XL-1 hold image columns x0 - 1 + lane; g: lanes 0..29 hold the own columns x0 + lane (staged at pixel 1 + lane)
        const int cx = x0 - 1 + lane, cg = x0 + lane;
        const unsigned xcol = lane < XL && (unsigned)cx < (unsigned)W ? (unsigned)(cx * 16) : kOOB;
        const unsigned gcol = lane < 30 && cg < W ? (unsigned)(cg * 16) : kOOB;
        auto load_x = [&](int r) -> u32x4 {
            u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rx, (unsigned)r < (unsigned)H && r <= y1 + KS - 3 ? (unsigned)(r * a.row_bytes) + xcol : kOOB, 0, 0);
            if constexpr (RELU_IN) v = __builtin_bit_cast(u32x4, relu_bits(v));
            return v;
        };
        auto load_g = [&](int r) -> u32x4 {
            return __builtin_amdgcn_raw_buffer_load_b128(rg, r >= y0 && r < y1 ? (unsigned)(r * a.row_bytes) + gcol : kOOB, 0, 0);
        };
        auto stage_x = [&](u32x4 v) { if (lane < XL) *reinterpret_cast<u32x4*>(xbuf + lane * 16) = v; };
        auto stage_g = [&](u32x4 v) { if (lane < 30) *reinterpret_cast<u32x4*>(gbuf + (1 + lane) * 16) = v; };
        auto frags = [&](bf16x8 (&ax)[NF]) {
            ax[0] = frag_of(xbuf);
            if constexpr (NF == 2) ax[1] = frag_of(xbuf + 2 * 16);
        };
        // AX[k]: fragments of x rows; at step I (output row y) row y - 1 + k lives in AX[(I + k) % KS]
        bf16x8 AX[KS][NF];
        u32x4 PX[KS], PG[KS];
        {   // fragments of x rows y0 - 1 .. y0 + KS - 3; the first KS rows of the loop are requested behind them, before the staging
            u32x4 r0[KS - 1];
#pragma unroll
            for (int k = 0; k < KS - 1; ++k) r0[k] = load_x(y0 - 1 + k);
#pragma unroll
            for (int k = 0; k < KS; ++k) { PX[k] = load_x(y0 + KS - 2 + k); PG[k] = load_g(y0 + k); }
#pragma unroll
            for (int k = 0; k < KS - 1; ++k) {
                stage_x(r0[k]);
                __builtin_amdgcn_wave_barrier();
                frags(AX[k]);
                __builtin_amdgcn_wave_barrier();
            }
        }
        auto step = [&](auto ic, const int tg) {
            constexpr int I = decltype(ic)::value;
            const int y = tg + I;
            stage_x(PX[I]);                                               // x row y + KS - 2
            stage_g(PG[I]);                                               // g row y
            PX[I] = load_x(y + 2 * KS - 2);
            PG[I] = load_g(y + KS);
            __builtin_amdgcn_wave_barrier();
            frags(AX[(I + KS - 1) % KS]);
            const bf16x8 BG = frag_of(gbuf);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ky = 0; ky < KS; ++ky)                               // ky: x row y - 1 + ky
#pragma unroll
                for (int f = 0; f < NF; ++f) acc[ky][f] = mma8(AX[(I + ky) % KS][f], BG, acc[ky][f]);
            accb = mma8(ONES, BG, accb);
            __builtin_amdgcn_sched_barrier(0);
        };
        for (int tg = y0; tg < y1; tg += KS)
            [&]<int... K>(std::integer_sequence<int, K...>) { (step(IC<K>{}, tg), ...); }(std::make_integer_sequence<int, KS>{});
    }
    // ---- the waves' sums -> one slab, fixed order
    for (int i = threadIdx.x; i < 8 * KEXT; i += 64 * WG_WAVES) slab[i] = 0.f;
    __syncthreads();
    const int n = lane & 15, kg = lane >> 4, bb = n >> 3, co = n & 7;
#pragma unroll
    for (int w = 0; w < WG_WAVES; ++w) {
        if (wave == w) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int row = 4 * kg + jj, aa = row >> 3, ci = row & 7;
                if (!(aa == 1 && bb == 1)) {
                    const int kx = aa - bb + 1;
#pragma unroll
                    for (int ky = 0; ky < KS; ++ky) slab[co * KEXT + (ky * KS + kx) * 8 + ci] += acc[ky][0][jj];
                }
                if constexpr (NF == 2) {
                    if (aa == 0 && bb == 0) {                              // a' = 2, b = 0: kx = 3
#pragma unroll
                        for (int ky = 0; ky < KS; ++ky) slab[co * KEXT + (ky * KS + 3) * 8 + ci] += acc[ky][1][jj];
                    }
                }
            }
            if (kg == 0 && bb == 0) slab[co * KEXT + ONESC] += accb[0];
        }
        __syncthreads();
    }
    float* out = d.slabs + (size_t)blockIdx.x * (8 * KEXT);
    for (int i = threadIdx.x; i < 8 * KEXT; i += 64 * WG_WAVES) out[i] = slab[i];
}

constexpr int kFwd1 = MSAU_PAIR_RELU_IN | MSAU_PAIR_RELU_MID, kFwd2 = MSAU_CONV_ADD | MSAU_CONV_RELU_OUT;
constexpr int kBwd1 = MSAU_PAIR_MASK_MID, kBwd2 = MSAU_CONV_MASK_A | MSAU_CONV_ADD;

// environment switches of this file, read once; msau_reload_env() makes the next call read them again (tests, A/B tools)
struct RowsEnv { int on, sh, sh16, waves, min_tasks, maxc, conv, wgrad, wgrad4, dout, pairwg, couple, deconv, cplwg, lrnb16; };
RowsEnv g_env;
bool g_env_ok = false;
const RowsEnv& rows_env() {
    if (!g_env_ok) {
        auto geti = [](const char* n, int dflt) { const char* v = std::getenv(n); return v && *v ? atoi(v) : dflt; };
        g_env.on = geti("MSAU_PAIR_ROWS", 1);
        g_env.sh = geti("MSAU_ROWS_SH", 0);                      // rows per segment (0: from MSAU_ROWS_WAVES)
        g_env.sh16 = geti("MSAU_ROWS_SH16", 16);                 // the same for the 16-channel pair alone (0: from MSAU_ROWS_WAVES; 16: fwd 15.2 -> 14.6, bwd 13.4 -> 12.9 us)
        g_env.waves = geti("MSAU_ROWS_WAVES", 3072);             // tasks (= waves) per launch to aim for
        g_env.min_tasks = geti("MSAU_ROWS_MIN_TASKS", 16);       // below: the tile kernels (a wave per task needs a few tasks per XCD at least)
        g_env.wgrad = geti("MSAU_WGRAD_ROWS", 1);                // the 8 -> 8 3x3 weight gradients on the row kernel
        g_env.wgrad4 = geti("MSAU_WGRAD_ROWS4", 0);              // ... and the 4x4 end conv's: correct (tests), but 13 us per step SLOWER than the tile kernel beside the main stream: off
        g_env.conv = geti("MSAU_CONV_ROWS", 1);                  // single convolutions of the 8-channel level on the row kernels
        g_env.pairwg = geti("MSAU_PAIR_WGRAD", 1);               // the first conv's weight gradient inside the pair's data-gradient launch
        g_env.lrnb16 = geti("MSAU_LRN_BWD16", 1);                // the LRN backward also on the 16-channel pair's data-gradient launch
        g_env.cplwg = geti("MSAU_COUPLE_WGRAD", 1);              // the coupling conv's weight gradient inside its data-gradient launch
        g_env.deconv = geti("MSAU_DECONV_ROWS", 2);              // transposed convs from their live taps on the row kernels: 1 = 16 -> 8, 2 = also 32 -> 16
        g_env.couple = geti("MSAU_PAIR_COUPLE", 1);              // the coupling 1x1 conv inside the pair's forward launch
        g_env.dout = geti("MSAU_DOUT_ROWS", 1);                  // the two-output data gradients (8 -> 8 + 8, 3x3 and 1x1) on the row kernels
        g_env.maxc = geti("MSAU_ROWS_MAXC", 16);                 // widest layer the row kernels take (8: the 16-channel pairs stay on the tile kernels)
        g_env_ok = true;
    }
    return g_env;
}

// segment height: a wave takes one task; aim for MSAU_ROWS_WAVES tasks in one round, at least 8 rows each.
int segment_rows(int B, int H, int nstrips, int warm, bool c16 = false) {
    const RowsEnv& e = rows_env();
    if (e.sh > 0) return e.sh < H ? e.sh : H;
    if (c16 && e.sh16 > 0) return e.sh16 < H ? e.sh16 : H;
    int nseg = e.waves / (B * nstrips);
    if (nseg < 1) nseg = 1;
    int sh = cdiv(H, nseg);
    if (sh < 10) sh = 10;
    sh = cdiv(sh + warm, UNR) * UNR - warm;              // `warm` iterations besides the rows: the trips of 6 come out even
    return sh < H ? sh : H;
}

}  // namespace

// which msau_conv_pair descriptors the row-streaming kernels take: 8 or 16 channels, bf16, the residual block's forward flag
// set (with or without ballot planes; at 16 channels also with MSAU_CONV_POOL) or its data-gradient flag set WITH ballot planes
int msau_rowpair_takes(int dtype, const msau_conv_pair_desc* d) {
    const RowsEnv& e = rows_env();
    if (!e.on || dtype != MSAU_BF16 || (d->C != 8 && d->C != 16) || d->C > e.maxc || (d->flags1 & MSAU_PAIR_TILES)) return 0;
    const int pool_ok = d->C == 16 ? MSAU_CONV_POOL : 0;
    const bool lrnb = d->flags1 & MSAU_PAIR_LRN_BWD, wg1 = d->flags1 & MSAU_PAIR_WGRAD1, cpl = d->flags1 & MSAU_PAIR_COUPLE;
    const bool fwd = (d->flags1 & ~MSAU_PAIR_COUPLE) == kFwd1 && (d->flags2 & ~pool_ok) == kFwd2,
               bwd = (d->flags1 & ~(MSAU_PAIR_LRN_BWD | MSAU_PAIR_WGRAD1)) == kBwd1 && d->flags2 == kBwd2;
    if (!fwd && !bwd) return 0;
    if (lrnb && !(bwd && (d->C == 8 || (d->C == 16 && e.lrnb16)) && d->lrn_a && d->lrn_da && d->lrn_k > 0.f)) return 0;
    if (wg1 && !(bwd && d->C == 8 && d->wg1_x && d->wg1_slabs && e.pairwg)) return 0;
    if (cpl) {
        // the coupling conv rides on the forward launch: its packed image must be the one the kernels index (1x1 over
        // concat(C, C) -> C: one chunk per source, 16 rows of 32)
        if (!fwd || !e.couple || (d->flags2 & MSAU_CONV_POOL) || !d->cpl_prev || !d->cpl_w || !d->cpl_y) return 0;
        msau_conv_pack_geom g;
        if (msau_conv_pack_geometry(dtype, d->C, d->C, d->C, 1, 1, 1, 1, 1, &g) != 0 || g.cch != d->C || g.nchunks != 2 || g.kchunk != 32 || g.rows != 16) return 0;
        if (d->cpl_pool_y && ((d->H | d->W) < 2)) return 0;
    }
    if (fwd && (d->flags2 & MSAU_CONV_POOL) && !d->pool_y) return 0;
    if (bwd && !(d->bits_mid && d->bits_a)) return 0;
    if (d->add != d->x) return 0;
    if ((int64_t)d->H * d->W * d->C * 2 >= (1ll << 31)) return 0;
    const int nstrips = cdiv(d->W, d->C == 8 ? OW : OW16);
    if ((int64_t)d->H * nstrips * 32 >= (1ll << 31)) return 0;
    if ((int64_t)d->B * nstrips * cdiv(d->H, 8) < e.min_tasks) return 0;    // too little work to fill the device: the tile kernels
    return 1;
}

extern "C" void msau_reload_env(void) { g_env_ok = false; }

int64_t msau_rowpair_plane_bytes(const msau_conv_pair_desc* d) {
    return (int64_t)d->B * d->H * cdiv(d->W, d->C == 8 ? OW : OW16) * 32;
}

// workgroups (= slabs of an MSAU_PAIR_WGRAD1 launch) of the descriptor's launch
int msau_rowpair_workgroups(const msau_conv_pair_desc* d) {
    const bool c8 = d->C == 8;
    const int nstrips = cdiv(d->W, c8 ? OW : OW16);
    const int SH = segment_rows(d->B, d->H, nstrips, c8 ? 2 + SKEW : 2, !c8);
    const int ntasks = d->B * nstrips * cdiv(d->H, SH);
    return 8 * (roundup(cdiv(ntasks, 8), 4) / 4);
}

int msau_rowpair_launch(hipStream_t s, const msau_conv_pair_desc* d) {
    RowArgs a;
    a.d = *d;
    const bool c8 = d->C == 8;
    const bool bwd = (d->flags1 & ~(MSAU_PAIR_LRN_BWD | MSAU_PAIR_WGRAD1)) == kBwd1, bits = d->bits_mid && d->bits_a, pool = !bwd && (d->flags2 & MSAU_CONV_POOL);
    const int cpl = !bwd && (d->flags1 & MSAU_PAIR_COUPLE) ? (d->cpl_pool_y ? 2 : 1) : 0;
    a.nstrips = cdiv(d->W, c8 ? OW : OW16);
    a.SH = segment_rows(d->B, d->H, a.nstrips, c8 ? 2 + SKEW : 2, !c8);
    if ((pool || cpl == 2) && (a.SH & 1) && a.SH < d->H) ++a.SH;            // 2x2 windows do not straddle segments
    a.nseg = cdiv(d->H, a.SH);
    a.ntasks = d->B * a.nstrips * a.nseg;
    a.tasks_per_xcd = roundup(cdiv(a.ntasks, 8), 4);
    a.row_bytes = d->W * d->C * 2;
    a.img_bytes = (unsigned)d->H * (unsigned)a.row_bytes;
    a.plane_img = (unsigned)d->H * (unsigned)a.nstrips * 32u;
    const dim3 grid(8 * (a.tasks_per_xcd / 4)), block(256);
    if (d->flags1 & MSAU_PAIR_WGRAD1)        // (the task split follows MSAU_ROWS_* switches: a plan built under other settings must not be launched)
        MSAU_CHECK_ARG((int)grid.x == d->wg1_nslabs, "conv_pair: MSAU_PAIR_WGRAD1 launch has %d workgroups, the caller allocated %d slabs "
                       "(msau_conv_pair_wgrad_slabs under other MSAU_ROWS_* settings?)", (int)grid.x, d->wg1_nslabs);
    if (c8) {
        const bool lrnb = d->flags1 & MSAU_PAIR_LRN_BWD, wg1 = d->flags1 & MSAU_PAIR_WGRAD1;
        if (bwd && lrnb && wg1) hipLaunchKernelGGL((rowpair_c8_kernel<true, true, true, true>), grid, block, 0, s, a);
        else if (bwd && wg1) hipLaunchKernelGGL((rowpair_c8_kernel<true, true, false, true>), grid, block, 0, s, a);
        else if (bwd && lrnb) hipLaunchKernelGGL((rowpair_c8_kernel<true, true, true>), grid, block, 0, s, a);
        else if (bwd) hipLaunchKernelGGL((rowpair_c8_kernel<true, true>), grid, block, 0, s, a);
        else if (cpl == 2 && bits) hipLaunchKernelGGL((rowpair_c8_kernel<false, true, false, false, 2>), grid, block, 0, s, a);
        else if (cpl == 2) hipLaunchKernelGGL((rowpair_c8_kernel<false, false, false, false, 2>), grid, block, 0, s, a);
        else if (cpl && bits) hipLaunchKernelGGL((rowpair_c8_kernel<false, true, false, false, 1>), grid, block, 0, s, a);
        else if (cpl) hipLaunchKernelGGL((rowpair_c8_kernel<false, false, false, false, 1>), grid, block, 0, s, a);
        else if (bits) hipLaunchKernelGGL((rowpair_c8_kernel<false, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((rowpair_c8_kernel<false, false>), grid, block, 0, s, a);
    } else {
        if (bwd && (d->flags1 & MSAU_PAIR_LRN_BWD)) hipLaunchKernelGGL((rowpair_c16_kernel<true, true, false, 0, true>), grid, block, 0, s, a);
        else if (bwd) hipLaunchKernelGGL((rowpair_c16_kernel<true, true, false>), grid, block, 0, s, a);
        else if (cpl == 2 && bits) hipLaunchKernelGGL((rowpair_c16_kernel<false, true, false, 2>), grid, block, 0, s, a);
        else if (cpl == 2) hipLaunchKernelGGL((rowpair_c16_kernel<false, false, false, 2>), grid, block, 0, s, a);
        else if (cpl && bits) hipLaunchKernelGGL((rowpair_c16_kernel<false, true, false, 1>), grid, block, 0, s, a);
        else if (cpl) hipLaunchKernelGGL((rowpair_c16_kernel<false, false, false, 1>), grid, block, 0, s, a);
        else if (bits && pool) hipLaunchKernelGGL((rowpair_c16_kernel<false, true, true>), grid, block, 0, s, a);
        else if (bits) hipLaunchKernelGGL((rowpair_c16_kernel<false, true, false>), grid, block, 0, s, a);
        else if (pool) hipLaunchKernelGGL((rowpair_c16_kernel<false, false, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((rowpair_c16_kernel<false, false, false>), grid, block, 0, s, a);
    }
    MSAU_CHECK_LAUNCH("rowpair_kernel");
    return 0;
}

// ---- single convolutions (msau_conv2d descriptors)
namespace {
template <int NS, int KH, int KW, int EPI, int EPI2 = 0>
int launch_rowconv8(hipStream_t s, const RowConvArgs& a) {
    hipLaunchKernelGGL((rowconv8_kernel<NS, KH, KW, EPI, EPI2>), dim3(8 * (a.tasks_per_xcd / 4)), dim3(256), 0, s, a);
    MSAU_CHECK_LAUNCH("rowconv8_kernel");
    return 0;
}
// instance index of a descriptor, 0 = none
int rowconv_case(int dtype, const msau_conv_desc* d) {
    const RowsEnv& e = rows_env();
    if (!e.on || !e.conv || dtype != MSAU_BF16 || (d->flags & MSAU_CONV_ELU)) return 0;
    if (d->ups == 2) {                                                     // the 16 -> 8 / 32 -> 16 transposed convs (rowdeconv8 / rowdeconv16)
        const bool c16 = d->C1 == 16 && d->Cout == 8, c32 = d->C1 == 32 && d->Cout == 16 && e.deconv >= 2;
        if (!e.deconv || !(c16 || c32) || d->C2 != 0 || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->dil != 1 || d->flags) return 0;
        if (d->pad_t != 1 || d->pad_l != 1) return 0;
        if ((d->Hout != 2 * d->Hin && d->Hout != 2 * d->Hin - 1) || (d->Wout != 2 * d->Win && d->Wout != 2 * d->Win - 1)) return 0;
        if ((int64_t)d->Hout * d->Wout * d->Cout * 2 >= (1ll << 31)) return 0;
        if ((int64_t)d->B * cdiv(d->Win, 16) * cdiv(d->Hout, 8) < e.min_tasks) return 0;
        return c16 ? 12 : 16;
    }
    const bool dout = d->flags & MSAU_CONV_DOUT;
    if (d->Cout != (dout ? 16 : 8) || d->C1 != 8 || (d->C2 != 0 && d->C2 != 8) || d->stride != 1 || d->ups != 1 || d->dil != 1) return 0;
    if (d->Hin != d->Hout || d->Win != d->Wout || d->KH != d->KW) return 0;
    if ((int64_t)d->Hout * d->Wout * 16 >= (1ll << 31)) return 0;
    if ((int64_t)d->B * cdiv(d->Wout, 32) * cdiv(d->Hout, 8) < e.min_tasks) return 0;
    const int f = d->flags, k = d->KH, dual = d->C2 != 0;
    if (dout) {                                                            // the data gradient of a conv over concat(x1, x2): g -> (dx1, dx2)
        if (!e.dout || dual || !d->y2) return 0;
        // the flag sets the reference's nets produce (anything else: the tile kernel's run-time epilogue)
        const int f1 = f & ~MSAU_CONV_DOUT, f2 = d->flags2;
        if (k == 3 && d->pad_t == 1 && d->pad_l == 1 && f2 == 0) return f1 == 0 ? 9 : f1 == MSAU_CONV_ACCUM ? 10 : 0;
        if (k == 1 && d->pad_t == 0 && d->pad_l == 0 && f1 == 0 && f2 == MSAU_CONV_MASK_B && d->mask_b2) return 11;
        if (k == 1 && d->pad_t == 0 && d->pad_l == 0 && f1 == MSAU_CONV_WGRAD && f2 == MSAU_CONV_MASK_B && d->mask_b2 && d->wg_x1 && d->wg_slabs && e.cplwg) return 13;
        return 0;
    }
    if (k == 3 && !dual && d->pad_t == 1 && d->pad_l == 1) {
        if (f == MSAU_CONV_LRN) return d->y2 && d->lrn_k > 0.f ? 1 : 0;
        if (f == 0) return 2;
        if (f == MSAU_CONV_ACCUM) return 3;
    }
    if (k == 3 && dual && d->pad_t == 1 && d->pad_l == 1 && f == 0) return 4;
    if (k == 1 && dual && d->pad_t == 0 && d->pad_l == 0 && f == MSAU_CONV_RELU_OUT) return 5;
    if (k == 4 && !dual && d->pad_t == d->pad_l && (d->pad_t == 1 || d->pad_t == 2)) {    // forward: 1; data gradient (flipped image): 2
        if (f == 0) return 6;
        if (f == MSAU_CONV_MASK_B) return 7;
        if (f == (MSAU_CONV_ACCUM | MSAU_CONV_MASK_B)) return 8;
    }
    return 0;
}
}  // namespace

int msau_rowconv_takes(int dtype, const msau_conv_desc* d) { return rowconv_case(dtype, d) != 0; }

namespace {
// task split of a rowconv8 launch (3x3 / 1x1 / 4x4 instances)
void rowconv_split(const msau_conv_desc* d, RowConvArgs& a) {
    a.nstrips = cdiv(d->Wout, 32);
    const int unr = d->KH == 4 ? 8 : d->KH + MSAU_ROWCONV_PF;
    const RowsEnv& e = rows_env();
    int nseg = e.waves / (d->B * a.nstrips);
    if (nseg < 1) nseg = 1;
    int sh = e.sh > 0 ? e.sh : cdiv(d->Hout, nseg);
    if (sh < 8) sh = 8;
    sh = roundup(sh, unr);
    a.SH = sh < d->Hout ? sh : d->Hout;
    a.nseg = cdiv(d->Hout, a.SH);
    a.ntasks = d->B * a.nstrips * a.nseg;
    a.tasks_per_xcd = roundup(cdiv(a.ntasks, 8), 4);
}
}  // namespace

// slabs (= workgroups) of an MSAU_CONV_WGRAD launch of this descriptor; 0 if no instance takes the flag
extern "C" int msau_conv2d_rider_slabs(int dtype, const msau_conv_desc* d) {
    if (!d || !(d->flags & MSAU_CONV_WGRAD) || rowconv_case(dtype, d) != 13) return 0;
    RowConvArgs a;
    rowconv_split(d, a);
    return 8 * (a.tasks_per_xcd / 4);
}

int msau_rowconv_launch(hipStream_t s, int dtype, const msau_conv_desc* d, int kchunk, int rows) {
    const int which = rowconv_case(dtype, d);
    if (which == 12 || which == 16) {
        MSAU_CHECK_ARG(kchunk == (which == 12 ? 160 : 288) && rows == 16, "rowdeconv: packed image of another geometry (kchunk %d, rows %d)", kchunk, rows);
        RowDeconvArgs a;
        a.d = *d;
        a.nstrips = cdiv(d->Win, 16);
        const RowsEnv& e = rows_env();
        int nseg = e.waves / (d->B * a.nstrips);
        if (nseg < 1) nseg = 1;
        int sh = e.sh > 0 ? e.sh : cdiv(d->Hout, nseg);
        if (sh < 8) sh = 8;
        a.SH = roundup(sh, 8);                                             // (4 input rows per loop trip)
        a.nseg = cdiv(d->Hout, a.SH);
        a.ntasks = d->B * a.nstrips * a.nseg;
        a.tasks_per_xcd = roundup(cdiv(a.ntasks, 8), 4);
        a.in_row_bytes = d->Win * d->C1 * 2;
        a.out_row_bytes = d->Wout * d->Cout * 2;
        a.in_img_bytes = (unsigned)d->Hin * (unsigned)a.in_row_bytes;
        a.out_img_bytes = (unsigned)d->Hout * (unsigned)a.out_row_bytes;
        if (which == 12) hipLaunchKernelGGL(rowdeconv8_kernel, dim3(8 * (a.tasks_per_xcd / 4)), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(rowdeconv16_kernel, dim3(8 * (a.tasks_per_xcd / 4)), dim3(256), 0, s, a);
        MSAU_CHECK_LAUNCH("rowdeconv_kernel");
        return 0;
    }
    RowConvArgs a;
    a.d = *d;
    rowconv_split(d, a);
    a.row_bytes = d->Wout * 16;
    a.img_bytes = (unsigned)d->Hout * (unsigned)a.row_bytes;
    a.wrow = kchunk;
    a.wchunk = rows * kchunk;
    switch (which) {
        case 1: return launch_rowconv8<1, 3, 3, MSAU_CONV_LRN>(s, a);
        case 2: return launch_rowconv8<1, 3, 3, 0>(s, a);
        case 3: return launch_rowconv8<1, 3, 3, MSAU_CONV_ACCUM>(s, a);
        case 4: return launch_rowconv8<2, 3, 3, 0>(s, a);
        case 5: return launch_rowconv8<2, 1, 1, MSAU_CONV_RELU_OUT>(s, a);
        case 6: return launch_rowconv8<1, 4, 4, 0>(s, a);
        case 7: return launch_rowconv8<1, 4, 4, MSAU_CONV_MASK_B>(s, a);
        case 8: return launch_rowconv8<1, 4, 4, MSAU_CONV_ACCUM | MSAU_CONV_MASK_B>(s, a);
        case 9: return launch_rowconv8<1, 3, 3, MSAU_CONV_DOUT>(s, a);
        case 10: return launch_rowconv8<1, 3, 3, MSAU_CONV_DOUT | MSAU_CONV_ACCUM>(s, a);
        case 11: return launch_rowconv8<1, 1, 1, MSAU_CONV_DOUT, MSAU_CONV_MASK_B>(s, a);
        case 13:
            MSAU_CHECK_ARG(8 * (a.tasks_per_xcd / 4) == d->wg_nslabs, "conv2d: MSAU_CONV_WGRAD launch has %d workgroups, the caller allocated %d slabs "
                           "(msau_conv2d_rider_slabs under other MSAU_ROWS_* settings?)", 8 * (a.tasks_per_xcd / 4), d->wg_nslabs);
            hipLaunchKernelGGL((rowconv8_kernel<1, 1, 1, MSAU_CONV_DOUT, MSAU_CONV_MASK_B, true>), dim3(8 * (a.tasks_per_xcd / 4)), dim3(256), 0, s, a);
            MSAU_CHECK_LAUNCH("rowconv8_kernel");
            return 0;
    }
    return msau_set_error(MSAU_ERR_ARG, "rowconv: no instance");
}

// ---- weight gradients (msau_conv2d_wgrad descriptors)
int msau_rowwgrad_takes(int dtype, const msau_wgrad_desc* d, int cch, int nchunks, int kextc) {
    const RowsEnv& e = rows_env();
    if (!e.on || !e.wgrad || dtype != MSAU_BF16) return 0;
    if (d->C1 != 8 || d->C2 != 0 || d->Cout != 8 || (d->KH != 3 && d->KH != 4) || d->KW != d->KH || d->dil != 1 || d->stride != 1) return 0;
    if (d->pad_t != 1 || d->pad_l != 1 || d->Hin != d->Hout || d->Win != d->Wout) return 0;        // (4x4: the SAME padding, 1 before / 2 after)
    if (d->flags & ~MSAU_CONV_RELU_IN) return 0;
    if (cch != 8 || nchunks != 1 || kextc != (d->KH == 4 ? 144 : 80) || d->nslabs < 1) return 0;
    if (d->KH == 4 && !e.wgrad4) return 0;
    if ((int64_t)d->Hout * d->Wout * 16 >= (1ll << 31)) return 0;
    if ((int64_t)d->B * cdiv(d->Wout, 30) * cdiv(d->Hout, 8) < e.min_tasks) return 0;
    return 1;
}

int msau_rowwgrad_launch(hipStream_t s, const msau_wgrad_desc* d) {
    RowWgradArgs a;
    a.d = *d;
    a.nstrips = cdiv(d->Wout, 30);
    const RowsEnv& e = rows_env();
    // one round: at most one task per wave (nslabs workgroups x 8 waves), as many as that allows up to MSAU_ROWS_WAVES
    int target = WG_WAVES * d->nslabs;
    if (target > e.waves) target = e.waves;
    int nseg = target / (d->B * a.nstrips);
    if (nseg < 1) nseg = 1;
    int sh = e.sh > 0 ? e.sh : cdiv(d->Hout, nseg);
    if (sh < 9) sh = 9;
    sh = roundup(sh, d->KH);
    a.SH = sh < d->Hout ? sh : d->Hout;
    a.nseg = cdiv(d->Hout, a.SH);
    a.ntasks = d->B * a.nstrips * a.nseg;
    a.row_bytes = d->Wout * 16;
    a.img_bytes = (unsigned)d->Hout * (unsigned)a.row_bytes;
    a.kext = d->KH == 4 ? 144 : 80;
    const bool relu = d->flags & MSAU_CONV_RELU_IN;
    if (d->KH == 4) {
        if (relu) hipLaunchKernelGGL((rowwgrad8_kernel<true, 4>), dim3(d->nslabs), dim3(64 * WG_WAVES), 0, s, a);
        else hipLaunchKernelGGL((rowwgrad8_kernel<false, 4>), dim3(d->nslabs), dim3(64 * WG_WAVES), 0, s, a);
    } else if (relu) hipLaunchKernelGGL((rowwgrad8_kernel<true>), dim3(d->nslabs), dim3(64 * WG_WAVES), 0, s, a);
    else hipLaunchKernelGGL((rowwgrad8_kernel<false>), dim3(d->nslabs), dim3(64 * WG_WAVES), 0, s, a);
    MSAU_CHECK_LAUNCH("rowwgrad8_kernel");
    return 0;
}
