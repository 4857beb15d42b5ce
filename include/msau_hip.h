/*
 * msau_hip.h -- C ABI of libmsau_hip.so: the MI355X (gfx950) kernels behind the MSAU train path.
 *
 * The reference (datvo06/MSAU) has no FFI layer: its hot path reaches the backend through
 * torch.nn modules.  Each entry point below replaces the backend op(s) one reference call site
 * reaches; the call site is cited as <file>:<line> relative to the reference root.  The Python
 * host side (msau_amd/) binds these with ctypes and passes raw device pointers; INTEGRATION.md
 * shows the binding a reference maintainer would add.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer unless the name ends in _host.
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises.
 *  - activations are NHWC ("pixel-major, channel-minor"): [B][H][W][Cs] with Cs = stored channels,
 *    a multiple of 8 (real channels first, zero padding after).  dtype selects the STORAGE type of
 *    activations and packed weights (MSAU_F32 or MSAU_BF16); accumulation is always fp32.
 *  - every function returns 0 on success, a negative msau_status otherwise; msau_last_error()
 *    returns a thread-local message.  Nothing throws across the ABI.
 *  - functions are re-entrant per stream; handles/buffers are not thread-safe.
 */
#ifndef MSAU_HIP_H
#define MSAU_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { MSAU_F32 = 0, MSAU_BF16 = 1 } msau_dtype;

typedef enum {
    MSAU_OK = 0,
    MSAU_ERR_ARG = -1,      /* bad shape / flag / alignment               */
    MSAU_ERR_LDS = -2,      /* tile does not fit the 160 KiB LDS          */
    MSAU_ERR_HIP = -3       /* a HIP runtime call failed                  */
} msau_status;

const char* msau_last_error(void);
int msau_version(void);
/* 16 hex digits: sha256 over the kernel sources + this header the library was built from (msau_amd/build.py::source_hash);
 * a binding that ships beside the sources compares the two at load time (msau_amd/_lib.py::load). */
const char* msau_source_hash(void);
/* sizeof() of the structs below as the library was compiled, for bindings that mirror them (a mirror that is too short
 * makes the library read past it): which = 0 msau_conv_desc, 1 msau_wgrad_desc, 2 msau_pack_entry, 3 msau_unpack_entry,
 * 4 msau_op, 5 msau_lrn_args, 6 msau_pool_args, 7 msau_attn_args, 8 msau_csum_args, 9 msau_reduce_args,
 * 10 msau_conv_pack_geom, 11 msau_wgrad_geom, 12 msau_conv_pair_desc, 13 msau_box_args, 14 msau_allreduce_args, 15 msau_owner_ctx,
 * 16 msau_attn_proj_bwd_args, 17 msau_dgrad2_args;
 * -1 for anything else. */
int msau_sizeof(int which);
/* LDS strides (bytes) the tile kernels give a pixel of `raw_bytes` channels / a weight row of `nks` 32-deep k-steps (csrc/msau_common.h:
 * lds_pixel_stride, lds_wrow_stride) -- exported so that a CPU test can check them against the bank model of MI355X_MICROARCH.md. */
int msau_lds_pixel_stride(int raw_bytes, int esz, int c8_per_chunk, int read_stride);
int msau_lds_wrow_stride(int nks, int esz);

/* ------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution, used for: SAME conv 3x3 / dilated 3x3 / 1x1 / 4x4 forward
 * (model/layers/layers.py:82-102,152-164 -> torch.nn.Conv2d + utils.pad_2d), its data gradient,
 * the transposed-conv forward (layers.py:249-250 -> ConvTranspose2d, `ups`=2) and the
 * transposed-conv data gradient (`stride`=2).  The concat of model/model.py:147,242,251 is folded
 * in as a second source pointer.
 *
 *   v   = sum_{tap,c} in(b, oy*stride + ky*dil - pad_t, ox*stride + kx*dil - pad_l, c) * W[co][tap][c]
 *         (+ bias[co])
 *   in  = concat(x1, x2) along channels, optionally ReLU'd on load (MSAU_CONV_RELU_IN); with ups=2 the
 *         input is the zero-stuffed image (value at even virtual coords only)
 *   v  *= (mask_a > 0)            if MSAU_CONV_MASK_A        (backward through a ReLU on the conv input)
 *   v  += add                     if MSAU_CONV_ADD           (forward: residual; backward: other grad path)
 *   v  += y_old                   if MSAU_CONV_ACCUM
 *   v   = max(v, 0)               if MSAU_CONV_RELU_OUT
 *   v  *= (mask_b > 0)            if MSAU_CONV_MASK_B        (backward through the ReLU that produced y's tensor)
 *   y   = v                       (mask_a, add, mask_b, y share y's shape [B][Hout][Wout][Cout])
 * ------------------------------------------------------------------------------------------ */
enum {
    MSAU_CONV_RELU_IN  = 1,
    MSAU_CONV_RELU_OUT = 2,
    MSAU_CONV_ADD      = 4,
    MSAU_CONV_ACCUM    = 8,
    MSAU_CONV_MASK_A   = 16,
    MSAU_CONV_MASK_B   = 32,
    MSAU_CONV_DOUT     = 128,   /* two output tensors (the data gradient of a conv over concat(x1, x2), one launch instead
                                   of two): the Cout stored rows are the channels of y (first half) and y2 (second half),
                                   each [B][Hout][Wout][Cout/2]; `flags` / add / mask_b apply to y, `flags2` (ACCUM and
                                   MASK_B only) / mask_b2 to y2.  Single-source 1x1 / 3x3 convs with C1 == Cout/2 in
                                   {8, 16, 32}; msau_conv2d_launch_info: info[7] & 2 when the launch can take it. */
    MSAU_CONV_LRN      = 256,   /* second output y2 = LocalResponseNorm(size = Cout)(y) (layers.py:145,161-162: the LRN
                                   behind the level-entry conv), computed from the storage-rounded y with lrn_alpha_over_n,
                                   lrn_beta, lrn_k: y2 = y * (k + alpha/n * sum over [c - n/2, c + (n-1)/2] of y^2)^-beta.
                                   Only with RELU_IN as the other flag; msau_conv2d_launch_info: info[7] & 4 when the
                                   launch can take it (otherwise run msau_lrn_fwd on y). */
    MSAU_CONV_POOL     = 512,   /* further output pool_y = MaxPool2d(2,2) of the zero-padded y (model/model.py:158-160),
                                   [B][ceil(Hout/2)][ceil(Wout/2)][Cout], and, when pool_idx is not NULL, the position
                                   0..3 (= 2*dy + dx) of the first maximum per element (what msau_maxpool2x2_bwd reads);
                                   info[7] & 8 when the launch can take it (otherwise run msau_maxpool2x2_fwd on y). */
    MSAU_CONV_IDS      = 1024,  /* x1 is an int32 id mask [B][Hin][Win] instead of a tensor: input channel c of a pixel is
                                   (id == c), ids outside [0, C1) are empty pixels -- the one-hot chargrid is synthesised in
                                   LDS, never painted nor read, and the result equals the dense launch bit for bit.  C1 = 64,
                                   3x3, <= 16 output channels; info[7] & 16.  msau_conv2d_wgrad takes the same flag. */
    MSAU_CONV_OWNER    = 2048,  /* the net's first conv fed with BOX LISTS instead of a painted tensor (csrc/ownerconv.hip): x1 is a
                                   HOST pointer to an msau_owner_ctx -- per pixel the index of the owning box (msau_raster_owner), the
                                   box list, the feature table.  The piecewise-constant embedding chargrid (data_generator_funsd_bert.py:
                                   64-93,240; 1536 bytes per pixel at 768 channels) is never painted: the forward gathers per-tap partial
                                   products from a [feature row][tap][co] table, the weight gradient sums the output gradient per box
                                   and tap.  3x3, stride 1, C1 -> 8; other flags: RELU_OUT only; also on msau_wgrad_desc.flags (x1 =
                                   the same context; the first msau_owner_slabs(d) slabs are written -- the slab reduction must be told).
                                   msau_conv2d_launch_info: info[7] & 32 when the launch can take it. */
    MSAU_CONV_NCHW     = 4096,  /* the net's first conv fed with the API's input tensor itself (csrc/conv_first.hip): x1 is fp32 NCHW
                                   [B][C][Hin][Win] with C = head_classes real channels (<= C1 = 64 stored; the missing ones are zero), and
                                   y2, if not NULL, receives its NHWC copy in the storage dtype -- what msau_nchw_to_nhwc would have
                                   written, for the weight gradient of the backward.  bf16, 3x3 SAME stride 1, 64 -> 8, Win % 4 == 0,
                                   16 <= Win <= 320 as far as the three-row LDS ring fits (msau_firstconv_takes decides;
                                   msau_conv2d_launch_info reports it); other flags: RELU_OUT only; results equal msau_nchw_to_nhwc + the dense launch bit for bit.
                                   msau_conv2d_launch_info: info[7] & 64 when the launch can take it. */
    MSAU_CONV_WGRAD    = 8192,  /* with MSAU_CONV_DOUT on the two-output data gradient of the 1x1 coupling conv over concat(8, 8) -> 8
                                   (model/model.py:143-148,246-252): the conv's WEIGHT gradient rides on this launch.  Its g operand is
                                   this launch's input x1 and the second source of the forward conv is already here as mask_b2 (the
                                   ReLU mask of y2 is the forward tensor itself), so only the first source is read in addition: wg_x1.
                                   wg_slabs = msau_conv2d_rider_slabs() slabs of [2 chunks][8][16] fp32 in the layout msau_wgrad_reduce
                                   expects for that conv (cch 8, kext 16, ones column 8 of chunk 0); the stand-alone weight-gradient
                                   launch, its read of g and of the second source disappear.  Row-streaming 8-channel bf16 instance only. */
    MSAU_CONV_ELU      = 32768, /* activation_name="elu" (model/model.py:412-416): MSAU_CONV_RELU_OUT means v = ELU(v) (alpha 1: v > 0 ? v : exp(v) - 1)
                                   and MSAU_CONV_MASK_B (the tensor's own stored output y) means v *= (y > 0 ? 1 : y + 1), the derivative of ELU at
                                   the pre-activation.  MSAU_CONV_RELU_IN / MASK_A stay ReLU: the reference's residual block starts with a
                                   hard-wired torch.nn.ReLU (model/model.py:35,39).  Generic kernel only (no fused / row / lean instance). */
    MSAU_CONV_HEAD     = 64     /* inference head (kv_model.py:305-313): besides y, write softmax over the Cout real
                                   channels of the (storage-rounded) result to head_probs (fp32 [B][Hout][Wout][Cout],
                                   dense) and the index of its first maximum to head_argmax (uint8 [B][Hout][Wout]).
                                   Forward convs with Cout <= 16 and no other epilogue flag; see
                                   msau_conv2d_launch_info: info[7] != 0 when the launch can take the flag, otherwise
                                   run msau_softmax_argmax_nhwc on y. */
};

typedef struct {
    int32_t B, Hin, Win, Hout, Wout;
    int32_t C1, C2;             /* stored channels of x1 / x2 (C2 = 0: single source)               */
    int32_t Cout;               /* stored output channels                                           */
    int32_t KH, KW, dil;
    int32_t pad_t, pad_l;
    int32_t stride;             /* 1 | 2                                                            */
    int32_t ups;                /* 1 | 2                                                            */
    int32_t flags;
    const void* x1;
    const void* x2;
    const void* wpack;          /* geometry.bytes bytes, written by msau_pack_params           */
    const float* bias;          /* [Cout] fp32 or NULL                                              */
    const void* add;
    const void* mask_a;
    const void* mask_b;
    void* y;
    float* head_probs;          /* MSAU_CONV_HEAD only: fp32 [B][Hout][Wout][head_classes]          */
    uint8_t* head_argmax;       /* MSAU_CONV_HEAD only: uint8 [B][Hout][Wout]                       */
    int32_t head_classes;       /* MSAU_CONV_HEAD only: real classes (<= Cout, <= 16)               */
    int32_t flags2;             /* MSAU_CONV_DOUT only: epilogue flags of y2                        */
    void* y2;                   /* MSAU_CONV_DOUT only                                              */
    const void* mask_b2;        /* MSAU_CONV_DOUT only                                              */
    float lrn_alpha_over_n, lrn_beta, lrn_k;   /* MSAU_CONV_LRN only (y2 is the LRN output)          */
    int32_t reserved0;
    void* pool_y;               /* MSAU_CONV_POOL only                                              */
    uint8_t* pool_idx;          /* MSAU_CONV_POOL only, may be NULL                                 */
    const void* wg_x1;          /* MSAU_CONV_WGRAD only: the forward conv's first source [B][H][W][8]  */
    float* wg_slabs;            /* MSAU_CONV_WGRAD only: [msau_conv2d_rider_slabs()][2][8][16] partial sums, one slab per workgroup */
    int32_t wg_nslabs;          /* MSAU_CONV_WGRAD only: the slab count the caller allocated: the launch refuses to write another number */
    int32_t reserved1;
} msau_conv_desc;
/* number of slabs an MSAU_CONV_WGRAD launch of this descriptor writes (= its workgroups); 0 if no instance takes the flag */
int msau_conv2d_rider_slabs(int dtype, const msau_conv_desc* d);

/* Geometry of the packed weight image the conv kernel expects for a given layer.
 * rows = roundup16(Cout) padded up to a power-of-two number of 16-row tiles; K is laid out as
 * [chunk][tap][channel-in-chunk] with `cch` channels per chunk, padded to a multiple of 32.        */
typedef struct {
    int32_t cch;        /* channels staged per K-chunk (multiple of 8)      */
    int32_t nchunks;
    int32_t kchunk;     /* padded K elements per chunk (multiple of 32)     */
    int32_t rows;       /* padded rows                                      */
    int64_t bytes;      /* total bytes of the packed image                  */
} msau_conv_pack_geom;

int msau_conv_pack_geometry(int dtype, int C1_stored, int C2_stored, int Cout_stored, int KH, int KW, int dil,
                            int stride, int ups, msau_conv_pack_geom* out);
int msau_conv2d(void* stream, int dtype, const msau_conv_desc* d);
/* which template instance msau_conv2d launches for this descriptor (for profiling / roofline):
 * info[0] = CT (16-row output-channel tiles), info[1] = PT (pixel tiles per wave: tile = 4*PT x 16),
 * info[2] = dynamic LDS bytes, info[3] = workgroups, info[4] = channel chunk, info[5] = chunks,
 * info[6] = 1 if a compile-time-specialised "lean" instance (conv_lean.hip) takes the launch, 2 if the chunked-K instance does,
 *           3 if a row-streaming instance (conv_rows.hip) does,
 * info[7] = bit 0: that instance implements MSAU_CONV_HEAD for this descriptor, bit 1: MSAU_CONV_DOUT,
 *           bit 2: MSAU_CONV_LRN, bit 3: MSAU_CONV_POOL, bit 4: MSAU_CONV_IDS, bit 5: MSAU_CONV_OWNER, bit 6: MSAU_CONV_NCHW */
int msau_conv2d_launch_info(int dtype, const msau_conv_desc* d, int32_t* info8);

/* ------------------------------------------------------------------------------------------
 * Two chained 3x3 SAME convolutions C -> C -> C (C stored channels in {8, 16, 32}) in one launch: the residual block
 * of model/model.py:37-50 with res_depth 2 (forward) and its two data gradients (backward).  The intermediate tensor
 * `mid` is written once and consumed from LDS, never re-read from HBM.
 *   t   = W1 * in(x) + b1            in(x) = max(x, 0) if MSAU_PAIR_RELU_IN else x ;  b1 may be NULL
 *   t   = max(t, 0)                  if MSAU_PAIR_RELU_MID      (forward: the first conv's ReLU)
 *   t  *= (mask_mid > 0)             if MSAU_PAIR_MASK_MID      (backward: through that ReLU)
 *   mid = t                          ([B][H][W][C], zero outside the image: the second conv's SAME padding)
 *   y   = epilogue(W2 * mid + b2)    flags2 = MSAU_CONV_{MASK_A, ADD, ACCUM, RELU_OUT, MASK_B, POOL} exactly as msau_conv2d
 * w1 / w2 are the packed images msau_pack_params writes for a single-source 3x3 conv C -> C (forward or flipped
 * data-gradient image).  msau_conv_pair_applicable: 1 if an instance exists for the shape (C, enough tiles, LDS).
 * ------------------------------------------------------------------------------------------ */
enum { MSAU_PAIR_RELU_IN = 1, MSAU_PAIR_RELU_MID = 2, MSAU_PAIR_MASK_MID = 4,
       MSAU_PAIR_LRN_BWD = 16,  /* backward flag set only: x0 (whose gradient this launch produces) is the output of
                                   LocalResponseNorm(size = C) applied to lrn_a (layers.py:145,161-162), and nothing else
                                   contributes to its gradient: the launch runs the LRN backward on the storage-rounded result in
                                   its epilogue and writes lrn_da (gradient w.r.t. lrn_a) INSTEAD of y -- the msau_lrn_bwd launch,
                                   its read of dy and the write of dy disappear.  Row-streaming 8-channel instance only
                                   (msau_conv_pair_applicable says so); lrn_alpha_over_n / lrn_beta / lrn_k as in msau_conv_desc */
       MSAU_PAIR_WGRAD1 = 32,   /* backward flag set only: the weight gradient of the block's FIRST conv in the same launch -- its g
                                   operand is the intermediate gradient this launch produces, so that tensor is neither written
                                   (`mid` is left untouched) nor re-read; wg1_x = the block's forward input x0 (ReLU applied on load),
                                   wg1_slabs = msau_conv_pair_wgrad_slabs() slabs of [8][80] fp32 in the layout msau_wgrad_reduce expects
                                   (kext 80, ones column 72).  Row-streaming 8-channel instance only. */
       MSAU_PAIR_COUPLE = 64,   /* forward flag set only: the coupling conv that follows the block in a coupled stage
                                   (model/model.py:143-148,246-252) rides on this launch --
                                       z = max(Wc * concat(cpl_prev, y) + bc, 0)           ([B][H][W][C], written to cpl_y)
                                   computed from the storage-rounded y exactly as the stand-alone 1x1 launch reads it; y itself is still
                                   written (the backward reads it).  cpl_w / cpl_b = the packed image / bias msau_pack_params writes
                                   for a 1x1 conv over concat(C, C) -> C.  With cpl_pool_y != NULL also the zero-padded MaxPool2d(2,2)
                                   of z (cpl_pool_idx: the 1-byte positions, may be NULL), as MSAU_CONV_POOL does for y.  The row-streaming
                                   8- and 16-channel instances and the 32-channel bf16 tile pair have it (msau_conv_pair_applicable
                                   says so); not together with MSAU_CONV_POOL in flags2. */
       MSAU_PAIR_DCOUPLE = 128, /* backward flag set only: the block's output y feeds nothing but a coupling conv z = ReLU(Wc concat(prev, y) + bc)
                                   whose two-output data gradient (MSAU_CONV_DOUT) becomes the PROLOGUE of this launch: the input tile is read
                                   from dcp_dz = d(z) (masked by z > 0 already) instead of x, and
                                       g = d(y) = (Wc[:, y]^T d(z)) . [dcp_mask > 0]   (dcp_mask = y)   feeds the two data gradients below AND is
                                                                                      written to x -- the weight gradient of the block's second
                                                                                      conv reads it there AFTER this launch
                                       d(prev)  =  Wc[:, prev]^T d(z)                 written to dcp_dprev (plain write: nothing accumulated, no mask)
                                   dcp_w = the packed two-output data-gradient image of the coupling conv (msau_conv_pack_geometry(C, 0, 2C, 1x1)).
                                   The 32-channel bf16 tile pair has it (msau_conv_pair_applicable says so). */
       MSAU_PAIR_TILES = 8 };   /* take the tile kernels (conv_pair.hip) even where the row-streaming kernel has an instance: the
                                   forward and the backward launch of one block must agree on the layout of the mask planes, so a
                                   caller whose forward carries a flag only the tile kernels implement (MSAU_CONV_POOL) sets this on
                                   both descriptors */
typedef struct {
    int32_t B, H, W, C;
    int32_t flags1, flags2;
    const void* x;
    const void* w1;
    const float* b1;
    const void* mask_mid;
    void* mid;
    const void* w2;
    const float* b2;
    const void* add;
    const void* mask_a;
    const void* mask_b;
    void* y;
    void* pool_y;               /* flags2 & MSAU_CONV_POOL (forward flag set only): MaxPool2d(2,2) of the zero-padded y, */
    uint8_t* pool_idx;          /* [B][ceil(H/2)][ceil(W/2)][C] and the 1-byte positions (may be NULL), as msau_conv2d   */
    uint8_t* bits_mid;          /* both NULL, or ReLU masks as bit planes [B][H][W][C/8] (bit c%8 of byte c/8 = element > 0): */
    uint8_t* bits_a;            /* the forward flag set WRITES (mid > 0) and (x > 0); the backward flag set READS them     */
                                /* instead of the tensors mask_mid / mask_a (which may then be NULL)                       */
    const void* lrn_a;          /* MSAU_PAIR_LRN_BWD: the LRN's input [B][H][W][C] ...                                     */
    void* lrn_da;               /* ... and the gradient w.r.t. it (written; y is not)                                      */
    float lrn_alpha_over_n, lrn_beta, lrn_k;
    int32_t reserved0;
    const void* wg1_x;          /* MSAU_PAIR_WGRAD1: x0, [B][H][W][C]                                                     */
    float* wg1_slabs;           /* [msau_conv_pair_wgrad_slabs()][C][80] partial sums, one slab per workgroup             */
    int32_t wg1_nslabs;         /* the slab count the caller allocated: the launch refuses to write another number        */
    int32_t reserved1;
    const void* cpl_prev;       /* MSAU_PAIR_COUPLE: the coupling conv's first source (the previous stage's tensor)      */
    const void* cpl_w;          /* ... its packed weight image (1x1, concat(C, C) -> C) and bias                          */
    const float* cpl_b;
    void* cpl_y;                /* ... its output z                                                                       */
    void* cpl_pool_y;           /* ... NULL, or the pooled z [B][ceil(H/2)][ceil(W/2)][C]                                 */
    uint8_t* cpl_pool_idx;      /* ... and its 1-byte positions (may be NULL)                                             */
    const void* dcp_dz;         /* MSAU_PAIR_DCOUPLE: gradient of the coupling conv's output, [B][H][W][C]                */
    const void* dcp_w;          /* ... its packed two-output data-gradient image (2C rows)                                */
    const void* dcp_mask;       /* ... the block's forward output y (ReLU mask of d(y))                                   */
    void* dcp_dprev;            /* ... gradient w.r.t. the coupling conv's first source (written)                         */
} msau_conv_pair_desc;
int msau_conv_pair_applicable(int dtype, const msau_conv_pair_desc* d);
int msau_conv_pair(void* stream, int dtype, const msau_conv_pair_desc* d);
/* bytes of ONE mask plane (bits_mid or bits_a) for this descriptor's shape.  The planes are private between the forward and
 * the backward launch of one residual block and their layout follows the instance that takes the shape: a byte per
 * (pixel, 8-channel group) for the tile kernels (conv_pair.hip), 32 bytes of lane ballots per (row, 30-column strip) for the
 * row-streaming 8-channel bf16 kernel (conv_rows.hip).  Allocate with this, never from the layout comment above. */
int64_t msau_conv_pair_bits_bytes(int dtype, const msau_conv_pair_desc* d);
/* number of slabs an MSAU_PAIR_WGRAD1 launch of this descriptor writes (= its workgroups); 0 if no instance takes the flag */
int msau_conv_pair_wgrad_slabs(int dtype, const msau_conv_pair_desc* d);
/* which instance takes the descriptor: 0 none, 1 a tile kernel (conv_pair.hip), 2 a row-streaming kernel (conv_rows.hip) */
int msau_conv_pair_instance(int dtype, const msau_conv_pair_desc* d);
/* The library reads its MSAU_* environment switches once.  msau_reload_env() makes the row-streaming kernel's switches
 * (MSAU_PAIR_ROWS, MSAU_CONV_ROWS, MSAU_WGRAD_ROWS, MSAU_ROWS_SH, MSAU_ROWS_WAVES, MSAU_ROWS_MIN_TASKS, MSAU_ROWS_MAXC) be read again on the next call: for tests and A/B
 * tools that change them inside one process. */
void msau_reload_env(void);

/* ------------------------------------------------------------------------------------------
 * Weight / bias gradient of the same convolution (autograd of torch.nn.Conv2d reached from
 * train_chargrid_funsd_msau.py:57 `loss.backward()`), and with stride=2 of the transposed conv
 * (roles of input and output-gradient swapped, see msau_amd/plan.py).
 *   slab[s][co][k]  partial sums over the pixel tiles workgroup s visited (deterministic, no atomics)
 *   k = [chunk][tap][channel] as in the forward pack, plus one extra column holding sum(g) = dbias.
 * msau_wgrad_reduce() sums the slabs and scatters into the flat fp32 gradient buffer.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t B, Hin, Win, Hout, Wout;
    int32_t C1, C2, Cout;
    int32_t KH, KW, dil, pad_t, pad_l, stride;
    int32_t flags;              /* MSAU_CONV_RELU_IN; MSAU_CONV_IDS (x1 = int32 id mask, C1 = 64, Cout = 8, 3x3) */
    const void* x1;
    const void* x2;
    const void* g;              /* [B][Hout][Wout][Cout] gradient w.r.t. the conv's pre-activation  */
    float* slabs;               /* [nslabs][Cout][kext] fp32                                        */
    int32_t nslabs;             /* number of workgroups per chunk to launch (<= tiles)              */
} msau_wgrad_desc;

typedef struct {
    int32_t cch, nchunks;
    int32_t kext;               /* columns per row of a slab: nchunks*taps*cch + 8, rounded to 16   */
    int32_t max_slabs;          /* number of pixel tiles (upper bound for nslabs)                   */
    int64_t slab_bytes;         /* bytes of ONE slab                                                */
    int32_t lean;               /* 1: a compile-time-specialised instance (wgrad_lean.hip) takes it; 2: the row-streaming
                                   instance (conv_rows.hip; needs d->nslabs set) */
    int32_t reserved;
} msau_wgrad_geom;

int msau_wgrad_geometry(int dtype, const msau_wgrad_desc* d, msau_wgrad_geom* out);
int msau_conv2d_wgrad(void* stream, int dtype, const msau_wgrad_desc* d);
/* Up to 4 weight gradients of one shape (everything but the tensors equal: msau_conv2d_wgrad_groupable(a, b) != 0) in ONE
 * grid -- the level-2/3 launches have 64-107 workgroups each and otherwise queue behind each other.  Same slabs, same bits
 * as n separate launches. */
int msau_conv2d_wgrad_groupable(int dtype, const msau_wgrad_desc* a, const msau_wgrad_desc* b);
int msau_conv2d_wgrad_group(void* stream, int dtype, const msau_wgrad_desc* const* ds, int n);

/* ------------------------------------------------------------------------------------------
 * Parameter packing (fp32 master parameters in the reference's OIHW / IOHW layouts -> packed
 * images for msau_conv2d) and gradient un-packing (slabs -> flat fp32 gradient in the reference's
 * layouts).  One launch handles a whole table.  The table lives in DEVICE memory.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int64_t src_off;        /* element offset of the parameter in the flat fp32 parameter buffer         */
    int64_t dst_off;        /* BYTE offset of the packed image in the pack arena                        */
    int32_t kind;           /* 0 = weight image, 1 = bias vector (fp32, padded to rows_store)           */
    int32_t dim0, dim1;     /* parameter dims 0 and 1 (OIHW: O,I ; ConvTranspose IOHW: I,O)             */
    int32_t KH, KW;
    int32_t row_is_dim0;    /* 1: image rows index dim0 (conv fwd, deconv dgrad); 0: rows index dim1   */
    int32_t flip;           /* 1: taps reversed (data-gradient / transposed-conv forward)              */
    int32_t row_off;        /* first parameter index (along the row dim) covered by this image          */
    int32_t rows_real;      /* valid rows                                                               */
    int32_t rows_pad;       /* geometry.rows                                                            */
    int32_t k1_real, k1_store, k2_real, k2_store;   /* channel split of the K dim (source 1 | source 2)  */
    int32_t cch, nchunks, kchunk;
    int32_t dtype;
} msau_pack_entry;

int msau_pack_params(void* stream, const float* flat_params, void* pack_arena,
                     const msau_pack_entry* table_dev, int n_entries, int max_elems_per_entry);

typedef struct {
    int64_t slab_off;       /* element offset of slab 0 in the fp32 slab arena                          */
    int64_t w_off;          /* element offset of the weight gradient in the flat gradient buffer        */
    int64_t b_off;          /* element offset of the bias gradient, or -1                               */
    int64_t b_src_off;      /* bias partial sums: element (slab s, channel r) of the slab arena is      */
    int64_t b_slab_stride;  /*   b_src_off + s*b_slab_stride + r*b_elem_stride, s < b_nslabs            */
    int32_t b_elem_stride;  /*   (the "ones" column of a wgrad slab, or msau_channel_sum partials)      */
    int32_t b_nslabs;
    int32_t b_count;        /* number of bias elements (conv: rows_real; transposed conv: its out channels) */
    int32_t nslabs;
    int32_t slab_elems;     /* nchunks * Cout_store * kext                                              */
    int32_t kext;
    int32_t dim0, dim1, KH, KW;
    int32_t row_is_dim0;    /* 1: slab rows index dim0, slab channels index dim1 ; 0: the other way     */
    int32_t rows_real;
    int32_t k1_real, k1_store, k2_real, k2_store;
    int32_t cch, nchunks;
    int32_t accumulate;     /* 1: add to the existing gradient (second wgrad of the same parameter)     */
} msau_unpack_entry;

int msau_wgrad_reduce(void* stream, const float* slab_arena, float* flat_grads,
                      const msau_unpack_entry* table_dev, int n_entries, int max_elems_per_entry);

/* partial per-channel sums of a gradient tensor (bias gradient of the transposed conv, whose wgrad
 * runs with swapped roles): partials[blk][c] = sum over the pixels block blk visited; nblk rows. */
int msau_channel_sum(void* stream, int dtype, const void* g, int64_t npix, int Cs, float* partials, int nblk);

/* ------------------------------------------------------------------------------------------
 * Layout conversion at the boundary: the reference API speaks NCHW fp32
 * (train_chargrid_funsd_msau.py:50-53, model/model.py:435-437).
 * ------------------------------------------------------------------------------------------ */
int msau_nchw_to_nhwc(void* stream, int dtype, const float* src, void* dst, int B, int C, int Cs, int H, int W);
int msau_nhwc_to_nchw(void* stream, int dtype, const void* src, float* dst, int B, int C, int Cs, int H, int W);
/* gradient of nhwc_to_nchw: dst[b][h][w][c] (+)= src[b][c][h][w] ; padded channels written as 0 */
int msau_nchw_grad_to_nhwc(void* stream, int dtype, const float* src, void* dst, int B, int C, int Cs, int H, int W,
                           int accumulate);

/* ------------------------------------------------------------------------------------------
 * LocalResponseNorm(size = n) across channels, alpha=1e-4, beta=0.75, k=1
 * (model/layers/layers.py:145,161-162 -> torch.nn.LocalResponseNorm).  C = real channels.
 *   y_c = a_c * (k + alpha/n * sum_{c' in [c-n/2, c+(n-1)/2]} a_c'^2)^-beta
 * ------------------------------------------------------------------------------------------ */
int msau_lrn_fwd(void* stream, int dtype, const void* a, void* y, int64_t npix, int C, int Cs, int n,
                 float alpha, float beta, float k);
int msau_lrn_bwd(void* stream, int dtype, const void* a, const void* dy, void* da, int64_t npix, int C, int Cs,
                 int n, float alpha, float beta, float k);

/* ------------------------------------------------------------------------------------------
 * 2x2 stride-2 max pool after zero SAME padding (model/model.py:158-160).  idx = argmax position
 * (0..3, first maximum in row-major window order, as torch CPU) kept as one byte per output element.
 * bwd: dx = ((accumulate & 1) ? dx : 0) + scatter(dy); dx *= (mask > 0) if mask != NULL -- or, with bit 1 of `accumulate` set (the
 * masked tensor is the output of an ELU, activation_name="elu"), dx *= (mask > 0 ? 1 : mask + 1).
 * ------------------------------------------------------------------------------------------ */
int msau_maxpool2x2_fwd(void* stream, int dtype, const void* x, void* y, uint8_t* idx, int B, int H, int W, int Cs);
int msau_maxpool2x2_bwd(void* stream, int dtype, const void* dy, const uint8_t* idx, void* dx, const void* mask,
                        int B, int H, int W, int Cs, int accumulate);

/* ------------------------------------------------------------------------------------------
 * Bottleneck self-attention core (model/layers/attention.py:156-162), given the 1x1 projections
 * f, g ([B][N][Ds]) and h ([B][N][Cs]):   y = x + h . softmax_rows(g^T f)
 * Two passes, no N x N matrix: stats (row max m_i and row sum Z_i of exp), then the output.
 * bwd: df, dg, dh from dy (the residual path dx += dy is the caller's).  ws: B*N*(Cs+4) floats.
 * ------------------------------------------------------------------------------------------ */
int msau_selfattn_fwd(void* stream, int dtype, const void* f, const void* g, const void* h, const void* x, void* y,
                      float* stats /* [B][N][2] */, int B, int N, int Ds, int Cs);
int msau_selfattn_bwd(void* stream, int dtype, const void* f, const void* g, const void* h, const void* dy,
                      const float* stats, void* df, void* dg, void* dh, float* ws, int B, int N, int Ds, int Cs);

/* The data gradients of the attention block's three 1x1 projections (model/layers/attention.py:152-154: f, g: C -> C/8, h: C -> C, all
 * reading the same tensor x) as ONE launch instead of three accumulating msau_conv2d launches:
 *   v  = Wf^T df + Wg^T dg + Wh^T dh        (fp32 accumulation over K = C/8 + C/8 + C in one MFMA chain)
 *   v += add                  if add        (the attention residual y = o + x: dy)
 *   v += dx_old               if accumulate (an earlier contribution to dx, e.g. the decoder's transposed conv)
 *   v  = mask_b > 0 ? v : 0   if mask_b     (ReLU of x's producer)
 *   dx = round(v)
 * wf_pack / wg_pack / wh_pack: the packed DATA-GRADIENT images of the three convs exactly as msau_conv2d takes them
 * (msau_conv_pack_geometry(dtype, C/8 or C, 0, C, 1, 1, 1, 1, 1): one chunk, 64 rows, kchunk 32 / 32 / 64) -- the launch reads its
 * weight fragments straight from them.  bf16, C = 64 only (MSAU_ERR_ARG otherwise: the caller keeps the three launches). */
typedef struct {
    const void* df; const void* dg; const void* dh;         /* [npix][C/8], [npix][C/8], [npix][C] */
    const void* wf_pack; const void* wg_pack; const void* wh_pack;
    const void* add;                                        /* [npix][C] or NULL */
    const void* mask_b;                                     /* [npix][C] or NULL */
    void* dx;                                               /* [npix][C] */
    int64_t npix;
    int32_t C, accumulate;
} msau_attn_proj_bwd_args;
int msau_attn_proj_bwd(void* stream, int dtype, const msau_attn_proj_bwd_args* a);

/* The two data gradients of a 1x1 conv over concat(x1, x2) with 64 + 64 input channels (the coupling conv of a coupled stage at the
 * bottleneck level, model/model.py:143-148) as ONE launch -- the 64-channel counterpart of MSAU_CONV_DOUT, which the tile / row kernels
 * implement up to 32 + 32:
 *   dx1 = [mask1 > 0 ?] (W1^T g [+ dx1_old]),   dx2 = [mask2 > 0 ?] (W2^T g [+ dx2_old])
 * w1_pack / w2_pack: the packed data-gradient images of the two sources exactly as msau_conv2d takes them
 * (msau_conv_pack_geometry(dtype, 64, 0, 64, 1, 1, 1, 1, 1): one chunk, 64 rows, kchunk 64).  bf16, C = 64 only. */
typedef struct {
    const void* g;                                          /* [npix][C]: gradient of the conv's output (already masked by its ReLU) */
    const void* w1_pack; const void* w2_pack;
    void* dx1; void* dx2;                                   /* [npix][C] each */
    const void* mask1; const void* mask2;                   /* [npix][C] or NULL */
    int64_t npix;
    int32_t C, accumulate1, accumulate2, reserved;
} msau_dgrad2_args;
int msau_dgrad2_1x1(void* stream, int dtype, const msau_dgrad2_args* a);

/* ------------------------------------------------------------------------------------------
 * Masked cross entropy (model/model.py:446-459) with the batch rule of SURVEY 8(e):
 *   loss = scale * sum_b 1/max(cnt_b,1) * sum_{p: label!=0} -log softmax(logits_p)[label_p]
 *   dlogits = scale/cnt_b * (softmax - onehot) at labelled pixels, 0 elsewhere (and in padded channels)
 * counts: [B] int32 (written by msau_label_counts).  loss_accum: one float, += (deterministic:
 * per-block partials then one ordered pass).  ws: >= msau_ce_ws_floats(npix) floats.
 * ------------------------------------------------------------------------------------------ */
int msau_label_counts(void* stream, const int64_t* labels, int32_t* counts, int B, int64_t hw);
/* the same count with every sample spread over K workgroups: partial[b * K + k] (K <= 64) is written; msau_masked_ce_multi adds the
 * K integers of a sample up itself (counts_k = K) -- no atomics, no follow-up launch */
int msau_label_counts_split(void* stream, const int64_t* labels, int32_t* partial, int B, int64_t hw, int K);
int64_t msau_ce_ws_floats(int64_t npix_total);
int msau_masked_ce(void* stream, int dtype, const void* logits, const int64_t* labels, const int32_t* counts,
                   void* dlogits, float* loss_accum, float* ws, int B, int64_t hw, int C, int Cs, float scale);
/* final + auxiliary logits in one launch (model/model.py:455-458): loss[0] = CE(logits) + CE(aux) (aux may be NULL),
 * written, not accumulated; ws: msau_ce_multi_ws_floats(B*hw) floats of scratch.  Same arithmetic per pixel as
 * msau_masked_ce; the block sums are added in index order by a one-wave follow-up kernel.  counts_k = 1: counts is [B] as
 * msau_label_counts writes it; counts_k = K > 1: [B][K] partial counts of msau_label_counts_split (B <= 1024). */
int64_t msau_ce_multi_ws_floats(int64_t npix_total);
int msau_masked_ce_multi(void* stream, int dtype, const void* logits, const void* aux, const int64_t* labels,
                         const int32_t* counts, void* dlogits, void* daux, float* loss, float* ws,
                         int B, int64_t hw, int C, int Cs, float scale, int counts_k);

/* plain cross entropy over EVERY pixel (label 0 is a class), as model/training/cost.py:35-65 `UNetLoss`:
 *   loss += scale * sum_p -log softmax(logits_p)[label_p] ; dlogits = scale * (softmax - onehot)           */
int msau_softmax_ce(void* stream, int dtype, const void* logits, const int64_t* labels, void* dlogits,
                    float* loss_accum, float* ws, int B, int64_t hw, int C, int Cs, float scale);
/* the same with per-class weights, torch.nn.CrossEntropyLoss(weight) as `UNetLoss(class_weights=...)` builds it
 * (model/training/cost.py:24-31): class_w = float[C] on the device;
 *   sums[0] += sum_p w_p nll_p, sums[1] += sum_p w_p (w_p = class_w[label_p]);  dlogits = w_p (softmax - onehot)   -- UN-normalised:
 * the loss is sums[0] / sums[1] and the caller scales dlogits by 1 / sums[1] (known only after the pass). ws: msau_ce_ws_floats. */
int msau_softmax_ce_weighted(void* stream, int dtype, const void* logits, const int64_t* labels, const float* class_w,
                             void* dlogits, float* sums, float* ws, int B, int64_t hw, int C, int Cs);

/* ------------------------------------------------------------------------------------------
 * Global-norm clip + Adam on flat fp32 buffers (train_chargrid_funsd_msau.py:24-26,58-59).
 * state (device, 8 floats): [0]=step count, [1]=grad norm (after grad_scale), [2]=clip coef,
 *                           [3]=1-b1^t, [4]=1-b2^t.   step is incremented by the call.
 * grad_scale is applied to the gradient first (1/world after an all-reduce sum).
 * ------------------------------------------------------------------------------------------ */
int64_t msau_adam_ws_floats(int64_t n);
int msau_clip_adam_step(void* stream, float* params, const float* grads, float* m, float* v, float* state,
                        float* ws, int64_t n, float lr, float beta1, float beta2, float eps, float max_norm,
                        float grad_scale);

/* ------------------------------------------------------------------------------------------
 * Chargrid rasteriser (next row N1; data_generator_funsd_bert.py:149-186 get_box_mask_box_label_word).
 * boxes: int32 [n][6] = {sample, y0, y1, x0, x1, value}, half-open, painted in order (later overwrites
 * earlier), clipped to the grid.  owner: int32 [B][H][W] scratch (index of the last covering box, -1).
 *   msau_raster_onehot : grid[b][y][x][c] = (c == value of the owning box); value < 0 paints zeros
 *   msau_raster_labels : labels[b][y][x]  = value of the owning box, 0 where none
 * ------------------------------------------------------------------------------------------ */
int msau_raster_owner(void* stream, const int32_t* boxes, int n, int32_t* owner, int B, int H, int W);
int msau_raster_onehot(void* stream, int dtype, const int32_t* boxes, const int32_t* owner, void* grid_nhwc,
                       int B, int H, int W, int C, int Cs);
int msau_raster_labels(void* stream, const int32_t* boxes, const int32_t* owner, int64_t* labels, int B, int H, int W);
/* BERT-embedding chargrid (data_generator_funsd_bert.py:64-93 get_box_mask_box_label, :240): the owning box's feature
 * vector feats[value][0..C) (fp32 [n_vectors][C], device) at every covered pixel, zeros elsewhere. */
int msau_raster_dense(void* stream, int dtype, const int32_t* boxes, const int32_t* owner, const float* feats,
                      void* grid_nhwc, int B, int H, int W, int C, int Cs);

/* ------------------------------------------------------------------------------------------
 * Box convolution for the model/model_box.py variant (MultiBoxConvBlock, model_box.py:9-59: BoxConv2d(c, 3, 28, 28)
 * from the third-party package `box_convolution`, then a 1x1 conv).  PARITY UNPINNED: that package is absent and the
 * reference holds no fixtures for it; the arithmetic follows the published definition (Burkov & Lempitsky, NeurIPS
 * 2018) and is checked against oracle/box_oracle.py only.  Pixels are unit squares, zero outside the image:
 *   out[b,y,x,c*F+f] = 1/A * integral over rows [y+hmin, y+hmax+1) x columns [x+wmin, x+wmax+1) of in[b,.,.,c],
 *   A = (hmax-hmin+1)*(wmax-wmin+1), (hmin,hmax,wmin,wmax)[c][f] = stored parameter * max box size (real-valued).
 *   msau_box_integral  : NHWC `dtype` input -> fp32 integral image, channel-planar [B][C][H+1][W+1]
 *   msau_box_params    : stored parameters (four [C][F] tensors inside the flat fp32 parameter buffer, element offsets)
 *                        -> boxes in pixels fp32 [4][C][F], kept valid (|edge| <= max size, max >= min), and the
 *                        reflected boxes the input gradient uses
 *   msau_box_filter    : the box filter read from an integral image; sum_filters = 1: out[c] = sum over the F filters
 *                        (input gradient = msau_box_filter on the integral image of the output gradient with the
 *                        reflected boxes); accumulate = 1: added to `out`
 *   msau_box_param_grad: d loss / d stored parameters, written (not accumulated) into the flat fp32 gradient buffer;
 *                        ws: msau_box_pgrad_ws_floats() floats of scratch (per-workgroup partials, ordered final sum)
 * ------------------------------------------------------------------------------------------ */
int64_t msau_box_integral_ws_floats(int B, int H, int W, int C);      /* scratch of msau_box_integral (row-segment column sums) */
int msau_box_integral(void* stream, int dtype, const void* in_nhwc, float* ii, float* ws, int B, int H, int W, int C, int Cs, int relu_in);
int msau_box_params(void* stream, const float* flat_params, int64_t off_hmin, int64_t off_hmax, int64_t off_wmin, int64_t off_wmax,
                    int C, int F, float max_h, float max_w, float* params_fwd, float* params_refl);
/* sum_filters = 0: ii has C planes, out[c*F+f].  sum_filters = 1 (input gradient): ii has C*F planes (plane c*F+f is
 * filtered with box c*F+f), out[c] = sum over f.  Epilogue as msau_conv2d: v *= (mask_a > 0); v += add; v += out
 * (accumulate); v *= (mask_b > 0); the three operands share out's shape and may be NULL. */
int msau_box_filter(void* stream, int dtype, const float* ii, const float* params, void* out_nhwc, int B, int H, int W, int C, int F,
                    int Cs_out, int sum_filters, int accumulate, const void* mask_a, const void* add, const void* mask_b);
int64_t msau_box_pgrad_ws_floats(int B, int H, int W, int C, int F);
int msau_box_param_grad(void* stream, int dtype, const float* ii, const float* params, const void* gout_nhwc, float* ws, float* flat_grads,
                        int64_t off_hmin, int64_t off_hmax, int64_t off_wmin, int64_t off_wmax, int B, int H, int W, int C, int F,
                        int Cs_out, float max_h, float max_w);

/* one box conv inside a launch sequence (MSAU_OP_BOX_FWD / MSAU_OP_BOX_BWD):
 *   forward : ii = integral(ReLU?(in)); out = box_filter(ii, params_fwd)
 *   backward: box-parameter gradient (ii, gout) -> flat_grads; ii_g = integral(gout); gin = epilogue(box_filter(ii_g, params_refl, sum)) */
typedef struct {
    const void* in; float* ii; const float* params_fwd; const float* params_refl; void* out;
    const void* gout; void* gin; float* ii_g; float* ws; float* flat_grads; float* ws_ii;
    const void* mask_a; const void* add; const void* mask_b;
    int64_t off_hmin, off_hmax, off_wmin, off_wmax;
    int32_t B, H, W, C, F, Cs_in, Cs_out, relu_in, accumulate;
    float max_h, max_w;
} msau_box_args;
int msau_box_fwd(void* stream, int dtype, const msau_box_args* a);
int msau_box_bwd(void* stream, int dtype, const msau_box_args* a);

/* ------------------------------------------------------------------------------------------
 * Launch-sequence executor: one call enqueues a whole pre-built list of the launches above (the static
 * plan of a forward or backward sweep), so the host cost per launch is a switch, not a Python/ctypes
 * round trip.  `args` points to the msau_*_args / descriptor struct of the op's kind; all pointers
 * inside must stay valid until the call returns (they are read at enqueue time only).
 * ------------------------------------------------------------------------------------------ */
enum {
    MSAU_OP_CONV2D = 1,      /* args: msau_conv_desc          */
    MSAU_OP_WGRAD = 2,       /* args: msau_wgrad_desc         */
    MSAU_OP_LRN_FWD = 3,     /* args: msau_lrn_args           */
    MSAU_OP_LRN_BWD = 4,     /* args: msau_lrn_args           */
    MSAU_OP_POOL_FWD = 5,    /* args: msau_pool_args          */
    MSAU_OP_POOL_BWD = 6,    /* args: msau_pool_args          */
    MSAU_OP_ATTN_FWD = 7,    /* args: msau_attn_args          */
    MSAU_OP_ATTN_BWD = 8,    /* args: msau_attn_args          */
    MSAU_OP_CHANNEL_SUM = 9, /* args: msau_csum_args          */
    MSAU_OP_WGRAD_REDUCE = 10, /* args: msau_reduce_args      */
    MSAU_OP_CONV_PAIR = 11,  /* args: msau_conv_pair_desc     */
    MSAU_OP_BOX_FWD = 12,    /* args: msau_box_args           */
    MSAU_OP_BOX_BWD = 13,    /* args: msau_box_args           */
    MSAU_OP_ALLREDUCE = 14,  /* args: msau_allreduce_args     */
    MSAU_OP_ATTN_PROJ_BWD = 15, /* args: msau_attn_proj_bwd_args */
    MSAU_OP_DGRAD2_1X1 = 16  /* args: msau_dgrad2_args        */
};
typedef struct { int32_t kind; int32_t dtype; const void* args; } msau_op;
typedef struct { const void* a; const void* dy; void* out; int64_t npix; int32_t C, Cs, n; float alpha, beta, k; } msau_lrn_args;
typedef struct { const void* x_or_dy; void* y_or_dx; uint8_t* idx; const void* mask; int32_t B, H, W, Cs, accumulate; } msau_pool_args;
typedef struct { const void* f; const void* g; const void* h; const void* x_or_dy; void* y; float* stats;
                 void* df; void* dg; void* dh; float* ws; int32_t B, N, Ds, Cs; } msau_attn_args;
typedef struct { const void* g; int64_t npix; int32_t Cs; float* partials; int32_t nblk; } msau_csum_args;
typedef struct { const float* slab_arena; float* flat_grads; const msau_unpack_entry* table_dev; int32_t n_entries, max_elems; } msau_reduce_args;
typedef struct { void* comm; float* buf; int64_t count; } msau_allreduce_args;
int msau_run_ops(void* stream, const msau_op* ops, int n);
/* As msau_run_ops, but ops whose kind carries MSAU_OP_SIDE are enqueued on `side_stream` after everything
 * enqueued so far on `stream` (event fork); `stream` waits for `side_stream` at the end (join).  Used for
 * the weight gradients: they depend on a finished output gradient and feed nothing but the final slab
 * reduction, so they run beside the data-gradient chain. */
#define MSAU_OP_SIDE 0x100
int msau_run_ops_overlap(void* stream, void* side_stream, const msau_op* ops, int n, int join);   /* join=0: leave the side stream running */
/* Ops whose kind carries MSAU_OP_PROBE are bracketed by a pair of timing HIP events on the stream they are launched
 * on, inside the normal sequence (concurrent side-stream work included): the in-situ duration bench.py's roofline
 * object quotes.  msau_probe_read synchronises on the recorded events, writes up to `cap` durations in microseconds
 * in launch order, stores the count in *n and clears the list. */
/* An op carrying MSAU_OP_JOIN first makes `stream` wait for everything enqueued so far on `side_stream` (used by the
 * plan's deterministic mode: the level-0 LRN backward then never shares the device with a weight-gradient kernel). */
#define MSAU_OP_JOIN 0x800
#define MSAU_OP_PROBE 0x200
/* msau_run_ops_dp: as msau_run_ops_overlap, with a third stream for the gradient exchange.  An op carrying MSAU_OP_COMM
 * (MSAU_OP_ALLREDUCE records: one bucket of the flat gradient, placed behind the slab reduction that completes it) is
 * enqueued on `comm_stream` after everything enqueued so far on `side_stream` and on `stream`; the sweep goes on meanwhile.
 * With join != 0 `stream` finally waits for both other streams: the optimiser step may follow.  msau_run_ops runs such ops
 * in place on its one stream. */
#define MSAU_OP_COMM 0x400
int msau_run_ops_dp(void* stream, void* side_stream, void* comm_stream, const msau_op* ops, int n, int join);
int msau_probe_read(float* us, int cap, int* n);
/* cost of an event pair with nothing between its two records, averaged over `reps` pairs on `stream` (microseconds):
 * the part of a probed duration that is the probe itself */
int msau_probe_overhead(void* stream, int reps, float* us);

/* ------------------------------------------------------------------------------------------
 * Data-parallel gradient exchange (SURVEY 8b "msau_allreduce_bucket", 8e): one process per GPU, RCCL over xGMI.  The
 * reference has no distributed code; the exchange sits where a DDP hook would, between loss.backward() and
 * optimizer.step() (train_chargrid_funsd_msau.py:57-59).
 *   msau_comm_available   : 1 if librccl.so could be loaded (it is resolved at run time, not a link dependency)
 *   msau_comm_unique_id   : rank 0 draws the 128-byte id of a new communicator; ship it to the other ranks
 *   msau_comm_init        : collective -- every rank, with its device current, the same id; *comm_out is the handle
 *   msau_allreduce_bucket : in-place fp32 SUM over ranks of buf[0..count) on `stream`; asynchronous like a kernel launch
 * ------------------------------------------------------------------------------------------ */
int msau_comm_available(void);
int msau_comm_unique_id(void* id128_out, int bytes);
int msau_comm_init(void** comm_out, int world, int rank, const void* id128, int bytes);
int msau_comm_destroy(void* comm);
int msau_allreduce_bucket(void* stream, void* comm, float* buf, int64_t count);

/* context of an MSAU_CONV_OWNER launch (host struct; every pointer inside is a DEVICE pointer) */
typedef struct {
    const int32_t* owner;       /* [B][H][W]: index of the box that owns the pixel, -1 = none (msau_raster_owner)              */
    const int32_t* boxes;       /* [n_boxes][6] = sample, y0, y1, x0, x1, value (= row of feats)                               */
    const float* feats;         /* [n_vec][C] fp32 feature table                                                              */
    const float* w;             /* the conv's fp32 master weight, OIHW [8][C][3][3] (rounded to the storage type on the fly)   */
    float* wt;                  /* workspace, wt_floats >= 72 * roundup(C, 32) floats (the weight re-ordered and rounded for the table
                                   product: fp32 storage uses 72 * C floats, bf16 storage 80 * roundup(C, 32) bf16 values)          */
    float* table;               /* workspace, n_vec * 72 floats  (forward)                                                    */
    float* sums;                /* workspace, n_boxes * 72 floats (weight gradient)                                           */
    float* csum;                /* workspace, csum_blocks * 8 floats (bias gradient partials, msau_channel_sum)               */
    int32_t n_boxes, n_vec, C, csum_blocks;
    int32_t wt_floats;          /* floats the caller allocated behind `wt`: the launch refuses a workspace that is too small      */
    int32_t reserved0;
} msau_owner_ctx;

int msau_owner_slabs(const msau_wgrad_desc* d);   /* slabs an MSAU_CONV_OWNER weight gradient of this descriptor writes (<= d->nslabs) */

/* misc */
/* occupy the stream for ~microseconds (<= 200000) with a single sleeping wave: measurement aid only */
int msau_spin(void* stream, int microseconds);
int msau_fill_zero(void* stream, void* p, int64_t bytes);
/* stress check of what msau_run_ops_overlap's FORK events rest on (tests): `iters` rounds of [a kernel on `stream` rewrites `words`
 * 32-bit words, an event created with (system_fence = 1) or without (0: hipEventDisableSystemFence) the system-scope fence forks, a
 * kernel on `side_stream` counts the words that do not hold the new pattern, a default event joins]; *mismatches = that count, summed */
int msau_fork_visibility_check(void* stream, void* side_stream, int iters, int64_t words, int system_fence, int64_t* mismatches);
/* a non-blocking stream with queue priority -1 (device's highest), 0 (default) or +1 (device's lowest); the caller owns it */
int msau_stream_create(int priority, void** stream_out);
int msau_stream_destroy(void* stream);
int msau_softmax_channels_nchw(void* stream, const float* logits, float* pred, int B, int C, int64_t hw);

/* ------------------------------------------------------------------------------------------
 * Inference head and input painter (next row N2; inference/kv_model.py:274-276,305-313).
 *   msau_softmax_argmax_nhwc : probs[p][c] = softmax_c(logits[p][0..C)), argmax[p] = first maximum of probs[p];
 *                              logits [npix][Cs] in `dtype` storage, probs fp32 [npix][C] dense, argmax uint8 [npix]
 *                              (the stand-alone form of MSAU_CONV_HEAD, bit-identical to it; any C <= 255)
 *   msau_onehot_ids          : grid[p][c] = (c == ids[p]) for c < C, 0 for C <= c < Cs -- to_categorical() of the
 *                              character-id mask (generic_util.py:97-98) written straight into the NHWC input
 * ------------------------------------------------------------------------------------------ */
int msau_softmax_argmax_nhwc(void* stream, int dtype, const void* logits, float* probs, uint8_t* argmax,
                             int64_t npix, int C, int Cs);
int msau_onehot_ids(void* stream, int dtype, const int32_t* ids, void* grid_nhwc, int64_t npix, int C, int Cs);

#ifdef __cplusplus
}
#endif
#endif /* MSAU_HIP_H */
