/*
 * popcorn_hip.h -- C ABI of libpopcorn_hip.so: the MI355X (gfx950) kernels behind POPCORN's dense
 * per-pixel CNN path.
 *
 * The reference (prs-eth/Popcorn) has no FFI: its boundary is the Python nn.Module API
 * (model/get_model.py:19-61, model/popcorn.py:20-22,100-101) and all arithmetic is stock PyTorch ops.
 * This header is the FFI a maintainer would bind *inside* that nn.Module: every
 * entry point below cites the reference op it replaces.  Conventions:
 *   - plain C, device pointers + sizes only, no torch types;
 *   - every function enqueues on the hipStream_t passed as `void* stream` (0 = the null stream),
 *     never synchronises, and returns 0 or a hipError_t / negative PC_E* code -- with ONE exception: pc_train_step, the eager
 *     whole-step executor, measures on its FIRST call per caller stream which of its side streams runs beside that stream (one host
 *     synchronisation, ~3 ms, cached per stream) and forks / joins through events of its own; it cannot be called while `stream` is
 *     being captured (PC_ENOTSUP) -- captured steps are built from the per-launch entry points;
 *   - tensors are NCHW, described by pc_src / pc_dst (strides in ELEMENTS; element type = `dtype`: fp32, or bf16 for the
 *     activation / activation-gradient tensors of PC_PREC_BF16); parameters, weight gradients and scalars are always fp32;
 *   - pointers are borrowed for the duration of the enqueued work only.
 */
#ifndef POPCORN_HIP_H
#define POPCORN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PC_ABI_VERSION 9

/* error codes (negative; positive values are hipError_t) */
#define PC_EINVAL (-1)     /* bad argument / unsupported channel combination */
#define PC_ENOGPU (-2)     /* no HIP device */
#define PC_ENOMEM (-3)     /* pc_train_step: the arena is too small (pc_step_io.arena_needed says how much it takes) */
#define PC_ENOTSUP (-4)    /* pc_train_step: a variant the native executor does not cover (the caller keeps the per-launch path) */

/* ---- tensor descriptors ------------------------------------------------------------------------- */

/* how a conv loader maps conv-domain coordinates onto a source tensor */
enum pc_src_mode {
    PC_SRC_DIRECT  = 0,  /* src(y-oy, x-ox); zero outside the source extent (Up's zero F.pad,
                            model/DDA_model/utils/networks.py:309-312) */
    PC_SRC_POOL2   = 1,  /* max over the 2x2 window of a source at twice the resolution: nn.MaxPool2d(2) fused into
                            the consumer (networks.py:289) */
    PC_SRC_REFLECT = 2   /* reflect padding (oy rows on top, ox cols on the left) + channel gather through `chmap`:
                            add_padding + channel reorder fused into the first conv (model/popcorn.py:231-258,130-134) */
};

enum pc_dtype { PC_F32 = 0, PC_BF16 = 1 };

typedef struct pc_src {
    const float* ptr;    /* element (b=0, c=0, y=0, x=0); points at 2-byte elements when dtype == PC_BF16 */
    int32_t C;           /* channels taken from this source (0 = unused) */
    int32_t H, W;        /* spatial extent of the source tensor */
    int64_t bstride;     /* elements between consecutive batch items */
    int64_t cstride;     /* elements between consecutive channels */
    int32_t rstride;     /* elements between consecutive rows */
    int32_t mode;        /* enum pc_src_mode */
    int32_t oy, ox;      /* DIRECT: placement of the source origin in the conv domain; REFLECT: top/left pad */
    int32_t chmap[4];    /* REFLECT only: conv channel c reads source channel chmap[c] */
    int32_t dtype;       /* enum pc_dtype */
    int32_t xstride;     /* elements between x-neighbours: 0 or 1 = planar rows (NCHW); a channels-last tensor
                            (bf16 mode, see below) has cstride = 1 and xstride = its channel count */
} pc_src;

typedef struct pc_dst {
    float* ptr;          /* 2-byte elements when dtype == PC_BF16 */
    int64_t bstride, cstride;
    int32_t rstride;
    int32_t dtype;       /* enum pc_dtype */
    int32_t xstride;     /* as pc_src.xstride */
    int32_t _pad;
} pc_dst;

/* folded BatchNorm2d(eval) + conv bias of one layer: y = relu(conv * scale + shift) with
   scale = gamma / sqrt(var + eps), shift = (bias - mean) * scale + beta   (networks.py:259-266) */
typedef struct pc_bn {
    const float* conv_bias;   /* [C] (may be NULL = 0) */
    const float* gamma;       /* [C] (NULL = no BN: scale 1, shift = conv_bias) */
    const float* beta;
    const float* mean;
    const float* var;
    float eps;
    int32_t _pad;
} pc_bn;

/* ---- library ------------------------------------------------------------------------------------- */
int pc_abi_version(void);
/* number of HIP devices visible (0 on a CPU-only host; never initialises a context) */
int pc_device_count(void);
const char* pc_error_string(int code);
/* sizeof of the ABI structs as compiled: 0 = pc_src, 1 = pc_dst, 2 = pc_bn, 3 = pc_conv_fwd_desc, 4 = pc_adam_groups,
   5 = pc_level2_fwd_desc, 6 = pc_step_plan, 7 = pc_step_io (lets a binding verify its struct layout) */
int pc_sizeof(int which);

/* ---- arithmetic mode -----------------------------------------------------------------------------------
 * PC_PREC_FP32 (default): fp32 operands on the fp32 MFMA (v_mfma_f32_16x16x4_f32), the reference's arithmetic.
 * PC_PREC_BF16: bf16 mixed precision (BASELINE config 4; the reference has no such mode -- its only hook is the unused
 *   `half` flag of to_cuda_inplace, utils/utils.py:22-27).  Rounding points, restated by oracle/popcorn_oracle.py (bf16_mode):
 *     - every operand of a matrix product (conv / transposed conv / 1x1 head layer: activations, gradients, weights) is a
 *       bf16 value (round-to-nearest-even); products accumulate in fp32;
 *     - every activation / activation-gradient tensor a kernel writes (conv + BN + ReLU outputs, pooled copies, transposed-
 *       conv outputs, the feature map, all data gradients) is rounded to bf16 by the producing epilogue, after the fp32
 *       bias / BN / ReLU-mask / accumulate arithmetic; the head's hidden layers are rounded after their ReLU;
 *     - everything else stays fp32: BN folding, partial logits and the building score, masks, occupancy product, census
 *       sums, loss, all WEIGHT gradients, the gradient all-reduce, clip, Adam and the master weights.
 *   Activation / activation-gradient tensors (everything a conv, transposed-conv or head kernel hands to another kernel:
 *   pc_src / pc_dst operands) are CHANNELS-LAST bf16 tensors in this mode: dtype = PC_BF16, cstride = 1, xstride = number
 *   of channels of the tensor (8 or 16; a slice of 8 channels of a 16-channel tensor is fine), ptr / rstride / bstride
 *   multiples of 8 elements -- one aligned 16-byte slot per pixel and 8-channel group (torch.channels_last of a bf16 tensor).
 *   The model input (PC_SRC_REFLECT sources), the partial logits / building score, masks, popdensemap, scale map, all
 *   parameters and all weight gradients stay planar fp32.  In PC_PREC_FP32 every tensor is planar fp32 (xstride 0 / 1).
 *   A call whose descriptors do not match the mode returns PC_EINVAL.  The mode is read when a call is enqueued.
 * Process-global; returns the previous mode (pc_set_precision) / the current one. */
enum pc_precision { PC_PREC_FP32 = 0, PC_PREC_BF16 = 1 };
int pc_set_precision(int mode);
int pc_get_precision(void);

/* fp32 mode only: how the sparse head (popcorn.py:80-85,161-190; pc_head_fwd / pc_head_bwd) multiplies.
 *   1 (default; POPCORN_HEAD_SPLIT=0 in the environment starts with 0): every fp32 operand is split EXACTLY into three bf16 numbers
 *     (8 + 8 + 8 mantissa bits) and a product is the fp32-accumulated sum of six bf16 x bf16 partial products on
 *     v_mfma_f32_16x16x32_bf16 / 16x16x16_bf16 -- the three dropped ones are below 2^-23 of the product, i.e. the per-product error is
 *     of the size of one fp32 rounding; results pass every fp32 parity test of the repository unchanged;
 *   0: v_mfma_f32_16x16x4_f32 (one fma chain per accumulator: the summation structure closest to the reference's fp32 GEMMs; the
 *     strict flat-bar parity tests and A/B runs use it).
 * Process-global, read when a call is enqueued (a forward call packs the weight images of BOTH head kernels: switch between steps, not
 * between a forward and its backward).  Returns the previous / the current setting. */
int pc_set_head_split(int on);
int pc_get_head_split(void);

/* fp32 mode only (round 6, ABI 9): how the 3x3 convolution kernels that have a split-operand form multiply: the fused data + weight
 * gradient pc_conv3x3_bwd_group (networks.py:259-266 backward) and the forward convs pc_conv3x3_bn_relu_fwd[_group] /
 * pc_conv3x3_up_fwd_group on aligned planar tensors with 8 or 16 channels (csrc/conv3x3_fwd_s3.h; results differ from the
 * v_mfma_f32_16x16x4_f32 form by rounding only, both are measured against float64 in tests/test_gpu_conv_fwd_split.py).  Same arithmetic
 * as pc_set_head_split:
 *   1 (default; POPCORN_CONV_SPLIT=0 in the environment starts with 0): operands are split exactly into three bf16 numbers when a strip is
 *     staged into LDS (once per strip, not per use), products are six bf16 x bf16 partial products accumulated in fp32 on
 *     v_mfma_f32_16x16x32_bf16; also opens the forms that exist only this way (16 gradient channels, the Down blocks' pool_act);
 *   0: v_mfma_f32_16x16x4_f32 kernels only (8 -> 8 channels; the strict flat-bar tests and A/B runs).
 * Process-global, read when a call is enqueued.  Returns the previous / the current setting. */
int pc_set_conv_split(int on);
int pc_get_conv_split(void);
/* does pc_conv3x3_bwd_group take these fp32 tensors (planar, aligned; Cg = 8 or 16 gradient channels, 8 x channels)?  1 / 0 */
int pc_conv3x3_bwd_ok(const pc_src* g, const pc_src* x, const pc_dst* out, const pc_src* pool_act, int B, int H, int W);

/* ---- conv3x3 (+BN +ReLU) forward: nn.Conv2d(3,pad 1) -> BatchNorm2d(eval) -> ReLU, networks.py:259-266.
 * Input channels = a.C + b.C (torch.cat([skip, up]) fused, networks.py:318); conv domain H x W, batch B.
 * w: [Cout][Cin][3][3].  Supported (Cin,Cout): (2,8) (4,8) (8,8) (16,8) (32,8) (8,16) (16,16). */
int pc_conv3x3_bn_relu_fwd(const pc_src* a, const pc_src* b, const float* w, const pc_bn* bn, int relu,
                           const pc_dst* out, int B, int H, int W, int Cin, int Cout, void* stream);

/* Grouped form: up to PC_MAX_GROUP independent problems of identical geometry (B,H,W,Cin,Cout, loader modes) in ONE
 * launch -- e.g. the SAR and optical streams of a layer, or the frozen building extractor next to the trainable U-Net
 * (the four share every layer shape).  Cuts launch count and fills the chip on the 32x32 layers. */
#define PC_MAX_GROUP 4
typedef struct pc_conv_fwd_desc {
    const pc_src* a; const pc_src* b; const float* w; const pc_bn* bn; const pc_dst* out;
    /* optional second output: MaxPool2d(2) of `out` (B x Cout x H/2 x W/2), written by the same epilogue, so that the
     * `Down` block that follows (networks.py:284-295) reads a quarter of the bytes instead of pooling the full-resolution
     * map on the fly.  Only when pc_conv3x3_pool_out_ok(out, H, W) (W % 32 == 0, H % 4 == 0, 16-byte aligned `out`);
     * otherwise the call returns PC_EINVAL. */
    const pc_dst* pool_out;
    /* optional, Cout == 8 only: instead of `out` (may then be NULL) write the single-channel map
     * dot_out[b][0][y][x] = sum_co dot_w[co] * relu(bn(conv))[co] -- the contribution of this layer's 8 feature channels to a
     * following 1x1 convolution (the frozen building extractor's fusion_out_conv, popcorn.py:301: its feature map has no
     * other consumer, so it is never written).  PC_PREC_FP32: any geometry, planar fp32 dot_out; PC_PREC_BF16: the geometry
     * condition of pool_out. */
    const float* dot_w; const pc_dst* dot_out;
    /* optional (PC_PREC_BF16, Cin == 8, no second source): w is [Cout][w_cin][3][3] and applies to input channels
     * [w_ci0, w_ci0 + w_cin) of `a`; the other channels of the 8-channel slot get zero weights.  w_cin == 0: the ordinary full
     * weight.  Lets the first convolutions of both streams (2 SAR / 4 optical channels, networks.py:130-133) read ONE shared
     * 8-channel channels-last input (pc_ingest_cl8) with the standard 8 -> 8 kernel, all (network, stream) pairs in one launch. */
    int32_t w_ci0, w_cin;
    /* optional (PC_PREC_BF16, 8 -> 8 layers, ReLU, no second source / pooled / dot output): the ConvTranspose2d(8, 8, 2, stride 2) of the
     * Up block that consumes this layer's output (networks.py:302-306: weight upt_w [8][8][2][2], bias upt_b) runs in this launch's
     * epilogue and writes upt_out (B x 8 x 2H x 2W, channels-last bf16): the rounded output of a lane IS the matrix operand of the
     * transposed conv, so the separate launch and its read of `out` disappear (`out` is still written for the backward pass).  All
     * problems of a launch or none. */
    const float* upt_w; const float* upt_b; const pc_dst* upt_out;
} pc_conv_fwd_desc;
int pc_conv3x3_pool_out_ok(const pc_dst* out, int H, int W);
int pc_conv3x3_bn_relu_fwd_group(int n, const pc_conv_fwd_desc* d, int relu, int B, int H, int W, int Cin, int Cout,
                                 void* stream);

/* ---- the first conv of an Up block WITHOUT the up-sampled map (PC_PREC_FP32): for an exactly 2x geometry,
 *     conv3x3(torch.cat([skip, ConvTranspose2d(C, C, 2, 2)(z)]))        (networks.py:302-318)
 * is computed from `skip` (Cs channels, H x W) and the LOW-resolution map z (C channels, H/2 x W/2) directly: conv3x3 o convT is a
 * linear map with a 2 x 2 low-resolution neighbourhood per output-pixel parity, whose weights (and the transposed conv's bias seen
 * through the taps that stay inside the image) are composed from w [8][Cs + C][3][3], wt [C][C][2][2], bt [C] into `ws`
 * (pc_conv3x3_up_ws_bytes(C) bytes) by a small first launch.  The up-sampled tensor is neither written nor read, and its half of
 * the convolution costs 2/3 of the MFMAs on a quarter of the input bytes.  (Cs, C) = (8, 8) or (16, 16), Cout = 8, H % 4 == 0,
 * W even, planar fp32 with 16-byte aligned rows (pc_conv3x3_up_fwd_ok); otherwise PC_EINVAL and the callers run
 * pc_convt2x2_fwd_group + pc_conv3x3_bn_relu_fwd_group.  (The backward below exists for W = 64 and 128.) */
typedef struct pc_conv_up_fwd_desc {
    const pc_src* skip; const pc_src* z; const float* w; const float* wt; const float* bt; const pc_bn* bn; const pc_dst* out; void* ws;
} pc_conv_up_fwd_desc;
int64_t pc_conv3x3_up_ws_bytes(int C);
int pc_conv3x3_up_fwd_ok(const pc_src* skip, const pc_src* z, const pc_dst* out, int H, int W, int Cs, int C);
int pc_conv3x3_up_fwd_group(int n, const pc_conv_up_fwd_desc* d, int relu, int B, int H, int W, int Cs, int C, void* stream);
/* The composition alone, for up to 2 * PC_MAX_GROUP convolutions of any mix of the two (Cs, C) shapes in ONE launch (only w, wt, bt, ws of
 * the descriptors are read); pc_conv3x3_up_fwd_group called with relu | PC_UP_PRECOMPOSED then takes ws as it is. */
#define PC_UP_PRECOMPOSED 2
int pc_conv3x3_up_compose_group(int n, const pc_conv_up_fwd_desc* d, const int* Cs, const int* C, void* stream);
/* Backward of that up-sampled half, again without the up-sampled tensor: from g = dL/d(conv output) (8 channels, H x W, already times
 * relu' * bn scale of the conv's layer) and z, ONE pass over g produces
 *   gz (optional) = relu'(z) * z_bn scale * dL/dz                    (what pc_convt2x2_bwd_group wrote for the transposed conv's input)
 *   dw[:, Cs:Cs + C] (=|+=), dwt, dbt                                 (gradients of w's up-sampled half, wt and bt; chain rule through the
 *                                                                     composed weights, incl. the bias seen through the in-image taps)
 * fwd_ws = the workspace the forward call filled (composed operand images), ws = pc_conv3x3_up_bwd_ws_bytes(B, H, C) bytes of scratch.
 * Three launches (the pass, the reduction over its workgroups, the chain rule).  Replaces the up-sampled column block's share of
 * pc_conv3x3_bwd_group / pc_conv3x3_dgrad_group + pc_conv3x3_wgrad_partial_group, pc_convt2x2_bwd_group and pc_convt2x2_fwd_group.
 * The skip half of w and the bias gradient of the conv still come from the skip column block's launch. */
typedef struct pc_conv_up_bwd_desc {
    const pc_src* g; const pc_src* z; const pc_bn* z_bn; const pc_dst* gz;
    const float* w; const float* wt; const float* bt; const void* fwd_ws; void* ws;
    float* dw; float* dwt; float* dbt;
} pc_conv_up_bwd_desc;
int64_t pc_conv3x3_up_bwd_ws_bytes(int B, int H, int C);
int pc_conv3x3_up_bwd_ok(const pc_src* g, const pc_src* z, const pc_dst* gz, int H, int W, int Cs, int C);
int pc_conv3x3_up_bwd_group(int n, const pc_conv_up_bwd_desc* d, int accumulate, int B, int H, int W, int Cs, int C, void* stream);
/* The same in pieces, for callers that batch the reduction with their other weight-gradient reductions: the pass alone (partials
 * [*nwg_out][*part_out] floats at d[i].ws), then pc_wgrad_reduce_batch with a kind-2 entry {partial = ws, dw = ws + nwg * part,
 * Cin = part} per problem, then the chain rule. */
int pc_conv3x3_up_bwd_partial_group(int n, const pc_conv_up_bwd_desc* d, int B, int H, int W, int Cs, int C, int* nwg_out,
                                    int* part_out, void* stream);
int pc_conv3x3_up_chain_group(int n, const pc_conv_up_bwd_desc* d, int accumulate, int nwg, int Cs, int C, void* stream);
/* the chain rules of an (8, 8) level and a (16, 16) level in one launch */
int pc_conv3x3_up_chain_both(int n8, const pc_conv_up_bwd_desc* d8, int nwg8, int n16, const pc_conv_up_bwd_desc* d16, int nwg16,
                             int accumulate, void* stream);

/* ---- conv3x3 data gradient (autograd of the op above w.r.t. its input).
 * g: gradient w.r.t. the conv output (already multiplied by relu-mask * bn-scale), Cg = forward Cout channels.
 * Produces the gradient for forward input channels [c0, c0+Cn) of a forward weight w: [Cg][Cin_total][3][3].
 * Epilogue (what the consumer of that gradient needs):
 *   act == NULL                : out (=|+=) dgrad
 *   act != NULL, pool == 0     : out (=|+=) dgrad * (act > 0) * act_bn.scale      (ReLU + frozen-BN backward of the
 *                                producing layer, whose post-ReLU output is `act`)
 *   act != NULL, pool == 1     : MaxPool2d(2) backward fused: dgrad lives at the pooled resolution; it is routed to the
 *                                first arg-max of each 2x2 window of `act` (full resolution), times (act>0)*scale, and
 *                                ACCUMULATED into out (full resolution).
 */
int pc_conv3x3_dgrad(const pc_src* g, const float* w, int Cin_total, int c0, int Cn,
                     const pc_src* act, const pc_bn* act_bn, int pool, int accumulate,
                     const pc_dst* out, int B, int H, int W, int Cg, void* stream);

typedef struct pc_conv_dgrad_desc {
    const pc_src* g; const float* w; const pc_src* act; const pc_bn* act_bn; const pc_dst* out;
} pc_conv_dgrad_desc;
int pc_conv3x3_dgrad_group(int n, const pc_conv_dgrad_desc* d, int Cin_total, int c0, int Cn, int pool,
                           int accumulate, int B, int H, int W, int Cg, void* stream);
/* ---- backward of one conv3x3 layer in ONE launch (PC_PREC_BF16 only): data gradient AND weight / bias gradient
 * partials from one pass over the layer's output gradient g and input x (both channels-last bf16, 8 or 16 channels).
 *   out (=|+=) relu'(x) * bn_scale(x_bn) * conv^T(g, w[:, c0:c0+XC])     (x_bn NULL: no ReLU / BN factor)
 *   ws <- per-workgroup partials of dW[:, c0:c0+XC] and db, to be finished by pc_wgrad_reduce_batch (Cin = XC, Cout = GC,
 *         dw = &dW[0][c0][0][0], dw_co_stride = Cin_total * 9; *nwg_out partials per problem)
 * (GC, XC) = channels of g and x: (8,8), (8,16), (16,16); with pool_act (16,8), (16,16).  x may be one column block of a
 * concatenated input: x->oy / ox = its placement in the conv domain (zero outside).  Replaces a pc_conv3x3_dgrad +
 * pc_conv3x3_wgrad_partial pair (5 tensor reads + 1 write) by 2 reads + 1 write. */
typedef struct pc_conv_bwd_desc {
    const pc_src* g; const pc_src* x; const float* w; const pc_bn* x_bn; const pc_dst* out; void* ws;
    /* optional (Down blocks): x is the 2x2-max-pooled copy of this full-resolution activation; the data gradient is then
     * scattered (+=) to the first arg-max of every window of `out` (twice the resolution), times relu'(pool_act) * bn_scale(x_bn) */
    const pc_src* pool_act;
    /* added to the call's c0 for THIS problem: the column blocks of a concat layer (same g, different x) can share one launch */
    int32_t c0_add; int32_t _pad;
} pc_conv_bwd_desc;
int pc_conv3x3_bwd_group(int n, const pc_conv_bwd_desc* d, int Cin_total, int c0, int accumulate, int B, int H, int W,
                         int* nwg_out, void* stream);
/* ablation switches of the split-operand fused backward kernel (tools/time_conv_bwd.py --ablate; 0 = normal operation): 1 no split / LDS
 * writes, 2 no data-gradient matrix phase, 4 no weight-gradient matrix phase, 8 no prefetch loads, 16 no epilogue.  Honoured by builds with
 * -DPOPCORN_CONV_ABLATE only (tools/build_variant.sh): as run-time flags they cost the shipped kernel its wait placement */
void pc_debug_conv_bwd(int dbg);
/* ablation switches for tools/ablate_conv.py (0,0 = normal operation); the split-form forward kernel (tools/time_conv_fwd.py --ablate:
 * 1 no loads / LDS writes, 2 no matrix phase, 4 no epilogue, 32 no split + LDS writes, 64 no loads) honours them in -DPOPCORN_CONV_ABLATE
 * builds only */
void pc_debug_conv(int dbg, int max_grid);
/* debug: buffer of 8 x int64 per workgroup receiving wall-clock stamps of the conv kernels' phases (NULL = off; tools/conv_timeline.py) */
void pc_debug_conv_ts(void* buf);

/* ---- conv3x3 weight/bias gradient.  x = forward input (a,b sources as in fwd), g as in dgrad.
 * dw: [Cout][Cin][3][3], db: [Cout]; (=|+=).  ws: workspace of pc_conv3x3_wgrad_ws_bytes(). */
int64_t pc_conv3x3_wgrad_ws_bytes(int Cin, int Cout);
int pc_conv3x3_wgrad(const pc_src* a, const pc_src* b, const pc_src* g, float* dw, float* db, int accumulate,
                     void* ws, int B, int H, int W, int Cin, int Cout, void* stream);

/* Deferred form: stage 1 only (one partial per workgroup into ws, *nwg_out = number of partials); the second stage of
 * many layers is then done by ONE pc_wgrad_reduce_batch launch.  Each deferred call needs its own ws slice. */
int pc_conv3x3_wgrad_partial(const pc_src* a, const pc_src* b, const pc_src* g, void* ws, int B, int H, int W,
                             int Cin, int Cout, int* nwg_out, void* stream);
/* grouped first stage (e.g. the SAR and optical streams of a layer): one launch, each problem with its own ws slice;
 * *nwg_out = partials per problem (identical for all) */
typedef struct pc_conv_wgrad_desc { const pc_src* a; const pc_src* b; const pc_src* g; void* ws; } pc_conv_wgrad_desc;
int pc_conv3x3_wgrad_partial_group(int n, const pc_conv_wgrad_desc* d, int B, int H, int W, int Cin, int Cout,
                                   int* nwg_out, void* stream);
int pc_convt2x2_wgrad_partial(const pc_src* x, const pc_src* g, void* ws, int B, int H, int W, int C, int* nwg_out,
                              void* stream);
/* grouped form: n <= PC_MAX_GROUP problems of identical geometry (the two streams of an Up block) in one launch; every
 * problem gets *nwg_out partials in its own ws */
typedef struct pc_convt_wgrad_desc { const pc_src* x; const pc_src* g; void* ws; } pc_convt_wgrad_desc;
int pc_convt2x2_wgrad_partial_group(int n, const pc_convt_wgrad_desc* d, int B, int H, int W, int C, int* nwg_out,
                                    void* stream);
/* Whole backward of a transposed conv in ONE launch: the weight / bias gradient partials of pc_convt2x2_wgrad_partial_group AND the
 * data gradient of pc_convt2x2_dgrad_group (out = relu'(x) * bn_scale(x_bn) * W^T g; x_bn NULL: no factor) -- both read the 2x
 * resolution gradient g, which passes through the kernel once.  fp32: aligned tensors with W % 16 == 0 only (PC_EINVAL otherwise:
 * the callers keep the two launches); PC_PREC_BF16: any geometry. */
typedef struct pc_convt_bwd_desc {
    const pc_src* x; const pc_src* g; const float* w; const pc_bn* x_bn; const pc_dst* out; void* ws;
} pc_convt_bwd_desc;
int pc_convt2x2_bwd_group(int n, const pc_convt_bwd_desc* d, int B, int H, int W, int C, int* nwg_out, void* stream);

/* ---- the whole 32 x 32 level of a U-Net stream in ONE launch (PC_PREC_FP32): Down.mpconv[1] = DoubleConv(16, 16)
 * (networks.py:253-271,284-295) on an already pooled 16-channel 32 x 32 map, followed by Up.up = ConvTranspose2d(16, 16, 2, 2)
 * (networks.py:302,306):   c1 = relu(bn1(conv3x3(x, w1)));  c2 = relu(bn2(conv3x3(c1, w2)));  u2 = convT(c2, wt) + bt.
 * One workgroup per (tile, problem) keeps x, c1 and c2 in LDS (a whole map is 64 KB: no halo exists at whole-tile residency);
 * only u2 (B x 16 x 64 x 64) and, when c1 / c2 are given (networks whose backward pass needs them), those two maps go to HBM.
 * Replaces two pc_conv3x3_bn_relu_fwd_group launches and one pc_convt2x2_fwd_group launch.  Geometry: x 16 channels, 32 x 32,
 * planar fp32, 16-byte aligned rows (pc_level2_fwd_ok); anything else returns PC_EINVAL and the callers keep the three launches.
 * PC_PREC_BF16 (level2_cl.hip): the same call on channels-last bf16 tensors (x, c1, c2: 16-channel pixels; u2 B x 16 x 64 x 64) --
 * the whole tile as the LDS strip image of the channels-last conv kernel, bit-identical to the three launches it replaces. */
typedef struct pc_level2_fwd_desc {
    const pc_src* x; const float* w1; const pc_bn* bn1; const float* w2; const pc_bn* bn2; const float* wt; const float* bt;
    const pc_dst* c1; const pc_dst* c2;   /* optional (NULL: not written) */
    const pc_dst* u2;                     /* optional when c2 is given: NULL skips the transposed conv (a consumer that takes c2
                                             through pc_conv3x3_up_fwd_group needs no up-sampled map) */
} pc_level2_fwd_desc;
int pc_level2_fwd_ok(const pc_src* x, const pc_dst* u2);
int pc_level2_fwd_group(int n, const pc_level2_fwd_desc* d, int B, void* stream);
/* Backward of the two convolutions of that level in ONE launch (both arithmetic modes: level2.hip; channels-last bf16 tensors
 * in PC_PREC_BF16: level2_cl.hip, one partial per tile, *nwg_out = B, replacing two pc_conv3x3_bwd_group launches), given g2 = dL/d(conv2 output) (already times
 * relu'(c2) * bn2 scale: pc_convt2x2_bwd_group writes it): weight / bias gradient partials of both layers (ws2: dW2, db2 from
 * g2 x c1; ws1: dW1, db1 from g1 x x; fp32: 2 partials per tile in the layout pc_wgrad_reduce_batch finishes as kind 0, Cin = Cout = 16:
 * *nwg_out = 2 B, each ws pc_level2_bwd_ws_bytes(B) bytes), the data gradient g1 = relu'(c1) * bn1 scale * conv^T(g2, w2) kept in
 * LDS, and the MaxPool2d(2) backward of conv^T(g1, w1) accumulated into `out` (B x 16 x 64 x 64: the first arg-max of every 2 x 2
 * window of `act`, times relu'(act) * act_bn scale).  Replaces two pc_conv3x3_wgrad_partial_group and two
 * pc_conv3x3_dgrad_group launches; same geometry rule as the forward (pc_level2_bwd_ok). */
typedef struct pc_level2_bwd_desc {
    const pc_src* g2; const pc_src* c1; const pc_src* x; const float* w1; const float* w2; const pc_bn* bn1;
    const pc_src* act; const pc_bn* act_bn; const pc_dst* out; void* ws1; void* ws2;
} pc_level2_bwd_desc;
int64_t pc_level2_bwd_ws_bytes(int B);
int pc_level2_bwd_ok(const pc_src* g2, const pc_src* c1, const pc_src* x, const pc_src* act, const pc_dst* out);
int pc_level2_bwd_group(int n, const pc_level2_bwd_desc* d, int B, int* nwg_out, void* stream);
/* debug: 16 x int64 buffer receiving wall-clock (100 MHz) phase stamps of workgroup (0, 0) of pc_level2_bwd_group (NULL = off) */
void pc_debug_level2_ts(void* buf);
/* test hook: the cross-lane helpers the kernels use instead of __shfl_xor (= ds_bpermute_b32), applied to one 64-lane wave of floats.
 * out[384]: [0,64) sum over the 8 lanes sharing lane >> 3 in the association order of the xor-1, -2, -4 butterfly; [64,128) max(x, x of
 * lane ^ 8); [128,192) x + x of lane ^ 16; [192,256) x + x of lane ^ 32; [256,320) x of lane ^ 1; [320,384) x of lane ^ 2 */
int pc_debug_lane_ops(const float* in, float* out, void* stream);
typedef struct pc_wgrad_reduce_desc {
    const float* partial;   /* ws of the deferred call */
    float* dw; float* db;   /* outputs ([Cout][Cin][3][3] / [Cin][Cout][2][2]; db may be NULL) */
    int32_t nwg, Cin, Cout; /* convT: Cin = Cout = C */
    int32_t kind;           /* 0: conv3x3, 1: convT 2x2, 2: raw sum of Cin floats per partial into dw (pc_conv3x3_up_bwd_partial_group),
                               3: the sparse head's partials (pc_head_bwd with PC_HEAD_BWD_DEFER_REDUCE; partial / nwg from
                               pc_head_bwd_partials): dw = HOST array of the head's 8 gradient tensors (device pointers in the order of
                               pc_head_bwd's dhw, NULL = skip), db / Cin / Cout unused; at most one such entry per call */
    int32_t accumulate;
    int32_t dw_co_stride;   /* conv3x3: elements between output channels of dw (0 = Cin * 9); > Cin * 9 when the entry is one
                               8-channel column block of a wider weight gradient (dw then points at its first column) */
    int32_t src_cin, src_ci0; /* conv3x3: the partials were computed over src_cin input channels (0 = Cin) of which this entry
                               writes the Cin channels starting at src_ci0 into dw -- the weight gradient of a first layer whose
                               input is a channel window of the shared 8-channel input (pc_conv_fwd_desc.w_ci0 / w_cin) */
} pc_wgrad_reduce_desc;
int pc_wgrad_reduce_batch(int n, const pc_wgrad_reduce_desc* d, void* stream);

/* ---- ConvTranspose2d(C, C, 2, stride 2), networks.py:302,306.  w: [Cin][Cout][2][2].  x: B x C x H x W -> out B x C x 2H x 2W */
int pc_convt2x2_fwd(const pc_src* x, const float* w, const float* bias, const pc_dst* out, int B, int H, int W, int C,
                    void* stream);
/* data gradient, fused with the ReLU+BN backward of the layer that produced x (act = x's post-ReLU values) */
int pc_convt2x2_dgrad(const pc_src* g, const float* w, const pc_src* act, const pc_bn* act_bn,
                      const pc_dst* out, int B, int H, int W, int C, void* stream);
/* grouped forms (problems of identical geometry in one launch, blockIdx.y = problem) */
typedef struct pc_convt_fwd_desc { const pc_src* x; const float* w; const float* bias; const pc_dst* out; } pc_convt_fwd_desc;
typedef struct pc_convt_dgrad_desc {
    const pc_src* g; const float* w; const pc_src* act; const pc_bn* act_bn; const pc_dst* out;
} pc_convt_dgrad_desc;
int pc_convt2x2_fwd_group(int n, const pc_convt_fwd_desc* d, int B, int H, int W, int C, void* stream);
int pc_convt2x2_dgrad_group(int n, const pc_convt_dgrad_desc* d, int B, int H, int W, int C, void* stream);
int64_t pc_convt2x2_wgrad_ws_bytes(int C);
int pc_convt2x2_wgrad(const pc_src* x, const pc_src* g, float* dw, float* db, int accumulate, void* ws,
                      int B, int H, int W, int C, void* stream);

/* ---- fusion_out_conv (1x1, 16->1) + sigmoid + crop: create_building_score's tail, popcorn.py:301,317-320;
 * networks.py:232,323-330.  feat: B x C x Hp x Wp with C = 16 (fusion_out_conv) or 8 (sar_out_conv / optical_out_conv of the
 * single-modality variants, networks.py:217-228), out: B x 1 x H x W taken at offset (py,px). */
int pc_outconv_sigmoid_crop(const pc_src* feat, const float* w, const float* bias, const pc_dst* out,
                            int B, int H, int W, int py, int px, void* stream);

/* ---- sparsity mask, popcorn.py:361-377 (non-sparse_unet branch).
 * mask[b,y,x] = region & ((building>0) | (rowsel[y] & colsel[x])), region = (admin_mask == census_idx[b]);
 * rowsel/colsel: uint8 [H]/[W] (the sorted multinomial draws, made on the host CPU generator like the reference).
 * If the whole batch selects nothing the mask falls back to the region (popcorn.py:374-375).
 * counts: int32[2] device scratch {nsel, nregion}. */
int pc_sparsity_mask(const float* building, const float* admin_mask, const int64_t* census_idx,
                     const uint8_t* rowsel, const uint8_t* colsel, int occupancymodel,
                     uint8_t* mask, int32_t* counts, int B, int H, int W, void* stream);

/* The sparse_unet branch of get_sparsity_mask (popcorn.py:336-359; no caller in the reference uses it):
 * mask = ((building > threshold) | (rowsel[y] & colsel[x])) & region, and per sample the undersampling ratio
 * ratio[b] = #(region & building <= threshold) / (#(grid & region & building <= threshold) + 1e-5).  threshold = 0.001 and a
 * 250 x 250 grid in the reference. */
int pc_sparsity_mask_unet(const float* building, const float* admin_mask, const int64_t* census_idx,
                          const uint8_t* rowsel, const uint8_t* colsel, float threshold, uint8_t* mask, float* ratio,
                          int B, int H, int W, void* stream);

/* pc_outconv_sigmoid_crop + pc_sparsity_mask in ONE launch (the building score feeds the mask directly; the empty-selection
 * fallback and the count fix-up are done by the last block to finish).  Same arguments and results as the two calls.
 * Must not run concurrently with itself on two streams of one process (library-owned accumulator + ticket). */
int pc_building_score_mask(const pc_src* feat, const float* w, const float* bias, const pc_dst* building_out,
                           const float* admin_mask, const int64_t* census_idx, const uint8_t* rowsel,
                           const uint8_t* colsel, int occupancymodel, uint8_t* mask, int32_t* counts,
                           int B, int H, int W, int py, int px, void* stream);

/* ---- the sparse head, popcorn.py:80-85,161-190,195-228:
 * per selected pixel 16 -> 64 -> 64 -> 64 -> (channel 0 of 2) 1x1-conv MLP with ReLUs, scale = relu(out),
 * popdensemap = scale * building, popcount[b] = sum over region pixels.  mask == NULL: dense head.
 * admin_mask == NULL: popcount = plain sum (popcorn.py:189-190).
 * feat: B x 16 x Hp x Wp read at crop offset (py,px) (revert_padding fused, popcorn.py:155).
 * hw: the 8 head tensors {w0,b0,w2,b2,w4,b4,w6,b6}.  ws: pc_head_ws_bytes().
 * stats (optional, device double[2]) receives {Nsel, sum(scale)}: Nsel = nsel_counts[0] (the int32 count written by
 * pc_sparsity_mask) or B*H*W for the dense head -- the inputs of the scale regulariser (utils/losses.py:74). */
int64_t pc_head_ws_bytes(int B, int H, int W);
int pc_head_fwd(const pc_src* feat, int py, int px, const float* const* hw, const uint8_t* mask,
                const float* building, const float* admin_mask, const int64_t* census_idx,
                float* scale_map, float* popdensemap, float* popcount, double* stats,
                const int32_t* nsel_counts, void* ws, int B, int H, int W, int flags, void* stream);
/* flags of pc_head_fwd / pc_head_bwd.  Both kernels read the head's weights from an LDS image that a small launch assembles in `ws`
 * per call.  A training step runs pc_head_fwd and pc_head_bwd on the SAME weights and workspace: PC_HEAD_FWD_PACK_BOTH makes the
 * forward call's pack launch write the backward's image as well, PC_HEAD_BWD_PACKED tells the backward call that it is there (the
 * caller guarantees: same hw contents, same ws, same B / H / W, same arithmetic mode in between) -- one launch less per step. */
#define PC_HEAD_FWD_PACK_BOTH 1
#define PC_HEAD_BWD_PACKED 1
/* PC_HEAD_FWD_DEFER_REDUCE: pc_head_fwd leaves popcount / stats unreduced (its last, single-block launch is skipped);
 * pc_head_popcount_loss(ws, ...) -- same ws, B, H, W -- then finishes popcount[B] and stats[2] AND computes the loss forward + backward
 * of pc_loss_fwd_bwd in one single-block launch (a single-process training step: one launch less; a data-parallel step all-reduces
 * the stats between the two and keeps the separate calls). */
#define PC_HEAD_FWD_DEFER_REDUCE 2
/* PC_HEAD_BWD_DEFER_REDUCE: pc_head_bwd leaves its per-workgroup weight-gradient partials in ws unreduced (its last launch is
 * skipped; g_feat is complete); the caller finishes them with an entry of kind 3 in the pc_wgrad_reduce_batch call of the same
 * backward pass -- partial / nwg from pc_head_bwd_partials(ws, B, H, W) (same ws, B, H, W, same arithmetic mode) -- one launch less. */
#define PC_HEAD_BWD_DEFER_REDUCE 2
int pc_head_bwd_partials(void* ws, int B, int H, int W, const float** partial, int* nwg);
int pc_head_popcount_loss(void* ws, int B, int H, int W, const int32_t* nsel_counts, const float* y, const float* lam4,
                          float scale_regularization, float lam_weak, float inv_B, float* popcount, double* stats,
                          float* loss_out, float* g_popcount, float* g_scale_const, void* stream);

/* backward of pc_head_fwd (the forward chain is recomputed in registers).  Upstream gradients, all optional (NULL):
 *   g_popcount[B]; g_popdense[B][H][W]; g_scale_map[B][H][W] (gradient w.r.t. scale = relu(out), e.g. the scattered
 *   gradient of the compacted scale vector); g_scale_const: DEVICE scalar added on every selected pixel (the
 *   scale-regularisation term scale_regularization * lam / Nsel, utils/losses.py:74-76).
 * Produces the 8 head gradients dhw[i] (=|+=; entries may be NULL; head.6 row/bias 1 get exact zeros, as autograd
 * gives for the unused second output channel) and the gradient w.r.t. the padded 16-channel feature map g_feat
 * (contiguous B x 16 x Hp x Wp; zero outside the crop and on unselected pixels).  If feat_bn_sar/feat_bn_opt are given
 * (BN of the layers that produced feature channels 0-7 / 8-15, networks.py:263-266) g_feat is additionally multiplied by
 * (feat > 0) * bn_scale, i.e. it is the gradient w.r.t. those layers' conv outputs. */
int pc_head_bwd(const pc_src* feat, int py, int px, const float* const* hw, const uint8_t* mask,
                const float* building, const float* admin_mask, const int64_t* census_idx,
                const float* g_popcount, const float* g_popdense, const float* g_scale_map,
                const float* g_scale_const, float* const* dhw, int accumulate,
                const pc_dst* g_feat, const pc_bn* feat_bn_sar, const pc_bn* feat_bn_opt,
                int Hp, int Wp, void* ws, int B, int H, int W, int flags, void* stream);

/* ---- compaction of scale[mask] in row-major (b,y,x) order (the boolean-index gather of popcorn.py:173).
 * out must hold B*H*W floats; *n_out (device int32) receives Nsel. */
int pc_compact_masked(const float* src, const uint8_t* mask, float* out, int32_t* n_out, void* ws, int64_t n,
                      void* stream);
int64_t pc_compact_ws_bytes(int64_t n);
/* The inverse (autograd of that gather): out[i] = mask[i] ? src[rank of i among the selected] : 0, i < n.  src holds at least
 * Nsel floats; ws as for pc_compact_masked. */
int pc_scatter_masked(const float* src, const uint8_t* mask, float* out, void* ws, int64_t n, void* stream);

/* F.pad(x, (left, right, top, bottom), mode="reflect") on `planes` contiguous H x W planes -> (H+top+bottom) x (W+left+right):
 * add_padding as a standalone op (popcorn.py:231-258).  The model path never calls it (the padding is fused into the first
 * convolution's loader); it backs the POPCORN.add_padding API method. */
int pc_reflect_pad(const float* in, float* out, int64_t planes, int H, int W, int top, int bottom, int left, int right,
                   void* stream);
/* The same with the channel gather of popcorn.py:130-134 (S1 / S2 band selection and order): out[b][j] = pad(in[b][sel[j]]) for
 * in (B, Cin, H, W) -> out (B, nsel, Hp, Wp), nsel <= 8, sel = HOST array.  The fp32 path materialises the padded input once per
 * forward pass with it: the first convolutions of all (network, stream) pairs and their weight gradients then take the aligned
 * DIRECT loader instead of running the reflect loader once per consumer. */
int pc_reflect_pad_select(const float* in, float* out, int B, int Cin, int nsel, const int* sel, int H, int W, int top, int bottom,
                          int left, int right, void* stream);
/* The ingest of a raw tile in ONE pass: band selection (data/PopulationDataset.py:566-568) + apply_normalize (utils/utils.py:105-127)
 * + the reflect padding / channel order above:  out[b][j] = pad((raw[b][band[j]] - mean[j]) / std[j]),  raw (B, Craw, H, W),
 * band / mean / std = HOST arrays of nsel <= 8 entries.  Replaces pc_select_normalize followed by pc_reflect_pad_select. */
int pc_select_normalize_pad(const float* raw, float* out, int B, int Craw, int nsel, const int* band, const float* mean,
                            const float* stdv, int H, int W, int top, int bottom, int left, int right, void* stream);
/* The same ingest for PC_PREC_BF16: out is ONE channels-last bf16 tensor (B, 8, Hp, Wp) -- an aligned 16-byte slot per pixel
 * holding the nsel <= 8 selected, normalised (mean / stdv may be NULL: input already normalised), reflect-padded channels rounded
 * to bf16, the remaining channels zero.  Both streams' first convolutions (pc_conv_fwd_desc.w_ci0 / w_cin) and their weight
 * gradients (pc_wgrad_reduce_desc.src_cin / src_ci0) read this one tensor through the standard 8-channel kernels instead of
 * running a planar-fp32 reflect loader per consumer. */
int pc_ingest_cl8(const float* raw, void* out, int B, int Craw, int nsel, const int* band, const float* mean, const float* stdv,
                  int H, int W, int top, int bottom, int left, int right, void* stream);
/* Either ingest from the TWO tensors a loader ships per sample (data/PopulationDataset.py:566-600: the Sentinel-2 GeoTIFF bands and the
 * Sentinel-1 bands, separately): s2 (B, C2, H, W) UINT16 digital numbers (reflectance x 10,000: 0 .. ~10,000 -- the file's own type, two
 * thirds of the host-to-device bytes of an fp32 tile) and s1 (B, C1, H, W) fp32 backscatter.  band[j] indexes the concatenation
 * [s2 | s1]; the uint16 -> fp32 conversion is exact, so the result equals the fp32-fed ingest bit for bit.  cl8 = 0: planar fp32
 * out (B, nsel, Hp, Wp) as pc_select_normalize_pad writes it; cl8 = 1: the channels-last bf16 slot tensor of pc_ingest_cl8. */
int pc_ingest_split(const uint16_t* s2, int C2, const float* s1, int C1, void* out, int cl8, int B, int nsel, const int* band,
                    const float* mean, const float* stdv, int H, int W, int top, int bottom, int left, int right, void* stream);

/* ---- training-step scalars ------------------------------------------------------------------------- */

/* Population loss + scale regulariser and their gradients, utils/losses.py:49-76 + autograd.
 * lam4 = weights of {l1_loss, log_l1_loss, mse_loss, log_mse_loss} (host array of 4); the batch mean is taken with
 * inv_B = 1 / (world_size * B) so that the SUM of all ranks' gradients equals the single-process gradient.
 * stats: device double[2] {Nsel, sum(scale)} from pc_head_fwd (all-reduced across ranks in data-parallel runs).
 * loss_out[2] = {loss, regulariser};  g_popcount[B] = d(lam_weak*loss)/d popcount;  *g_scale_const = lam_weak *
 * scale_regularization / Nsel (the constant gradient of the regulariser on every selected pixel). */
int pc_loss_fwd_bwd(const float* popcount, const float* y, const double* stats, const float* lam4,
                    float scale_regularization, float lam_weak, float inv_B, int B,
                    float* loss_out, float* g_popcount, float* g_scale_const, void* stream);

/* L2 norm of a flat gradient buffer (torch.nn.utils.clip_grad_norm_'s total_norm, run_train.py:233-234); deterministic. */
int pc_grad_norm(const float* g, int n, float* norm_out, void* stream);

/* Parameter groups of the flat buffer for the Adam step.  The reference gives the encoder (limit1) or the whole U-Net
 * (limit2) no gradient on large samples (run_train.py:191-198; networks.py:124-132), and torch.optim.Adam skips a
 * parameter whose .grad is None entirely: no weight decay, no moment update, no per-parameter step increment.  The flat
 * buffer is cut into nseg consecutive segments [seg_end[i-1], seg_end[i]) (seg_end[nseg-1] == n), each belonging to one
 * of PC_ADAM_GROUPS groups; groups whose bit is clear in active_mask are left untouched (p, m, v and the group's step
 * counter), and step_dev is then int32[PC_ADAM_GROUPS] -- one counter per group, torch's per-parameter `step`. */
#define PC_ADAM_MAX_SEG 8
#define PC_ADAM_GROUPS 4
typedef struct pc_adam_groups {
    int32_t nseg;
    int32_t seg_end[PC_ADAM_MAX_SEG];
    int32_t seg_group[PC_ADAM_MAX_SEG];
    int32_t active_mask;
} pc_adam_groups;

/* clip_grad_norm_(max_norm) + torch.optim.Adam step over flat buffers (run_train.py:82-90,233-238).  Weight decay
 * (L2, added to the gradient) applies to elements [0, n_decay) only.  hyper_dev: device float[1] {lr};
 * step_dev: device int32 step counter(s), incremented by the call (groups == NULL: one counter, everything updated).
 * max_norm <= 0 or norm_dev == NULL: no clipping. */
int pc_adam_clip_step(float* p, const float* g, float* m, float* v, int n, int n_decay, const float* hyper_dev,
                      float weight_decay, float beta1, float beta2, float eps, float max_norm,
                      const float* norm_dev, int32_t* step_dev, const pc_adam_groups* groups, void* stream);

/* The same in ONE launch: every workgroup computes the total norm of g itself (written to norm_out_dev if non-NULL), then
 * clips and updates; max_norm <= 0: no clipping.  g must be 16-byte aligned.  Must not run concurrently with itself on
 * two streams of one process (a device-side ticket decides which workgroup advances the step counter). */
int pc_adam_clip_step_fused(float* p, const float* g, float* m, float* v, int n, int n_decay, const float* hyper_dev,
                            float weight_decay, float beta1, float beta2, float eps, float max_norm,
                            float* norm_out_dev, int32_t* step_dev, const pc_adam_groups* groups, void* stream);

/* Band selection + per-band (x - mean) / std: data/PopulationDataset.py:566-568 + utils/utils.py:105-127.
 * raw: B x Craw x H x W; out: B x 6 x H x W; band6/mean6/std6: host arrays of 6. */
int pc_select_normalize(const float* raw, int Craw, const int* band6, const float* mean6, const float* std6,
                        float* out, int B, int H, int W, void* stream);

/* The trainer's augmentations applied in ONE pass while the raw 6-channel tile is assembled (run_train.py:386-402, utils/transform.py:
 * RandomBrightness / RandomGamma on the 4 Sentinel-2 bands, then vertical flip, horizontal flip and a rotation by rot quarter turns
 * counter-clockwise, jointly on input and admin_mask; one coin per batch, drawn by the caller with the reference's generators):
 *   raw[b] = rot90^rot(hflip(vflip(cat[aug(s2[b]), s1[b]]))),   admin_out[b] = the same geometric map of admin[b]
 * s2 (B, 4, H, W) digital numbers, s1 (B, 2, H, W), admin (B, H, W), all fp32; raw (B, 6, Ho, Wo), admin_out (B, Ho, Wo) with
 * (Ho, Wo) = (W, H) for odd rot.  bright / gam: apply x -> clamp(x / 1e4 * beta, 0, 1) * 1e4, then x -> clamp((max(x, 0) / 1e4)^gamma, 0, 1) * 1e4.
 * Replaces apply_transformations_and_normalize's augmentation half (utils/utils.py:130-214); the normalisation is the step's ingest. */
int pc_augment_raw(const float* s2, const float* s1, const float* admin, float* raw, float* admin_out, int B, int H, int W,
                   int vflip, int hflip, int rot, int bright, float beta, int gam, float gamma, void* stream);

/* ---- native executor of ONE training step: the body of the reference's inner loop (run_train.py:186-238) as one call -----------
 *   forward(train, padding=False, sparse=True) (popcorn.py:100-193) -> get_loss (utils/losses.py:49-76) -> x lam_weak -> backward
 *   -> clip_grad_norm_ -> Adam step -> zero_grad
 * for ANY batch geometry (B, H, W) and truncation regime (encoder_no_grad / unet_no_grad: run_train.py:191-198): the reference
 * trains on weak_batch_size = 2 census regions of varying size (arguments/train.py:16,34-36), which cannot replay a captured HIP
 * graph -- and launching the step's ~45 kernels one ctypes call at a time costs more host time than a small region's kernels take
 * (tools/host_time_eager.py).  pc_train_step computes the geometry of both networks (the trainable U-Net on the add_padding domain,
 * popcorn.py:231-258; the frozen building extractor on its own 14-pixel reflect-padded domain, popcorn.py:279-322), carves every
 * activation / gradient / workspace out of ONE caller-provided arena (bump allocation, rows padded to 16 bytes so that every conv
 * launch takes the aligned staged loaders whatever the region's width), fills the descriptors and issues all launches from C++.
 * The plan (parameter pointers, BN descriptors, optimiser constants) is built once per trainer (pc_step_create); a step passes only
 * its batch.  PC_PREC_FP32, dual-stream (S1 + S2) models whose building score comes from the frozen extractor; anything else
 * returns PC_ENOTSUP and the caller keeps the per-launch path (same kernels, launched through the entry points above).
 * Small regions fork the frozen extractor's chain and the weight-gradient launches onto an internal side stream; WHICH stream is
 * measured at the first call on a caller's stream (HIP multiplexes streams onto a few hardware queues, and a side stream on the
 * caller's queue serialises with it): that first call launches a few microsecond-long probe kernels and synchronises once, so
 * pc_train_step is an EAGER entry point -- not to be called on a stream that is being captured into a graph.
 * Conv layer order of the arrays below: inc.conv.0, inc.conv.3, down1.conv.0, down1.conv.3, down2.conv.0, down2.conv.3,
 * up2.conv.0, up2.conv.3, up1.conv.0, up1.conv.3 (networks.py:121-151,253-320); transposed convs: up2.up, up1.up. */
#define PC_STEP_CONVS 10
typedef struct pc_step_stream {
    const float* w[PC_STEP_CONVS];      /* conv weights [Cout][Cin][3][3] */
    pc_bn bn[PC_STEP_CONVS];            /* conv bias + BatchNorm2d(eval) of the layer */
    const float* wt[2]; const float* bt[2];      /* ConvTranspose2d weights [C][C][2][2] and biases */
    float* dw[PC_STEP_CONVS]; float* db[PC_STEP_CONVS]; float* dwt[2]; float* dbt[2];   /* gradient targets (trainable network only) */
    int32_t chan[4];                    /* model-input channel of conv channel c ([R,G,B,NIR,VV,VH]: SAR 4,5; optical 2,1,0,3) */
    int32_t cin, feat_c0;               /* 2 / 4 input channels; first channel of this stream in the 16-channel feature map */
} pc_step_stream;
typedef struct pc_step_net { pc_step_stream s[2]; const float* fusion_w; const float* fusion_b; } pc_step_net;
typedef struct pc_step_plan {
    pc_step_net unet, extractor;        /* model.unetmodel (trainable), model.building_extractor (frozen) */
    const float* head_w[8]; float* head_dw[8];     /* {w0,b0,w2,b2,w4,b4,w6,b6} and their gradients */
    float* flat_p; float* flat_g; float* adam_m; float* adam_v;     /* flat parameter / gradient / moment buffers (n floats) */
    int32_t n, n_decay, n_head, occupancymodel;
    const float* hyper_dev; int32_t* step_dev; float* norm_dev; double* stats_dev; float* loss_dev; float* g_scale_const_dev;
    pc_adam_groups groups;              /* segments of the flat buffer; active_mask is set per step from the regime */
    float weight_decay, beta1, beta2, eps, max_norm, scale_regularization, lam_weak;
    float lam4[4];
    int32_t extractor_pad;              /* 14 (popcorn.py:46) */
    int32_t band[8]; float mean[8]; float stdv[8];      /* raw-tile ingest: model channel c = (raw band band[c] - mean[c]) / stdv[c] */
} pc_step_plan;
enum pc_step_data { PC_DATA_INPUT = 0,   /* data = normalised model input (B, 6, H, W) fp32 */
                    PC_DATA_RAW = 1,     /* data = raw tile (B, craw, H, W) fp32 (data/PopulationDataset.py:566-568 + utils/utils.py:105-127 fused) */
                    PC_DATA_SPLIT = 2 }; /* data = uint16 S2 digital numbers (B, 4, H, W), data2 = fp32 S1 (B, 2, H, W) */
#define PC_STEP_SEL_MAX 16384
#define PC_STEP_FWD 1      /* ingest .. head forward (+ popcount / stats unless the loss launch finishes them) */
#define PC_STEP_BWD 2      /* loss, head backward, U-Net backward into flat_g */
#define PC_STEP_UPD 4      /* clip + Adam */
typedef struct pc_step_io {
    int32_t B, H, W, data_kind;
    const void* data; const void* data2; int32_t craw, dp;          /* dp != 0: data parallel -- the caller all-reduces stats_dev between the
                                                                       FWD and BWD phases and flat_g between BWD and UPD */
    const float* admin_mask; const int64_t* census_idx; const float* y;
    const uint8_t* sel;          /* DEVICE: H row flags then W column flags (the two multinomial draws of get_sparsity_mask, popcorn.py:366-369) */
    int32_t encoder_no_grad, unet_no_grad; float inv_B; int32_t _pad;
    const uint8_t* sel_host;     /* or HOST: the same H + W flags in host memory (sel is then ignored): they travel bit-packed in the kernel
                                    arguments of a one-block launch -- no H2D copy command (and none of its dispatch latency) per step;
                                    H + W <= PC_STEP_SEL_MAX */
    void* arena; int64_t arena_bytes;       /* device scratch, 256-byte aligned; contents are undefined between steps */
    /* results */
    int64_t arena_needed;                   /* bytes this geometry takes (always set; PC_ENOMEM when arena_bytes is smaller) */
    int64_t off_popcount, off_popdense, off_scale, off_mask, off_building;   /* byte offsets of the step's outputs inside the arena:
                                               popcount f32[B], popdensemap f32[B][H][W], scale map f32[B][H][W], mask u8[B][H][W],
                                               building score f32[B][1][H][W] -- valid until the next call */
    int32_t launches, _pad2;                /* kernels enqueued by the call */
} pc_step_io;
void* pc_step_create(const pc_step_plan* plan);
void pc_step_destroy(void* handle);
int pc_train_step(void* handle, pc_step_io* io, int phases, void* stream);
/* debug: host nanoseconds the LAST pc_train_step call spent in its sizing pass (out[0]) and in its launching pass (out[1])
 * (tools/host_time_eager.py) */
void pc_debug_step_host_ns(double* out);

/* pc_reflect_pad_select / pc_select_normalize_pad / pc_ingest_split (planar form) for rows of any width and an output whose rows are
 * out_rstride >= Wp floats apart (planes of Hp * out_rstride): kind as enum pc_step_data; mean == NULL: no normalisation. */
int pc_ingest_pad_strided(int kind, const void* data, const void* data2, int Cin, float* out, int out_rstride, int B, int nsel,
                          const int* sel, const float* mean, const float* stdv, int H, int W, int top, int bottom, int left, int right,
                          void* stream);
/* p[0 .. n) = 0 by a kernel (zero fills inside a captured step must not be memset nodes: DESIGN.md); p 16-byte aligned */
int pc_zero_fill(float* p, int64_t n, void* stream);

/* ---- census aggregation + sliding-window stitching ----------------------- */

/* Region totals: sums[id] = sum of pred over pixels with boundary == id, id in [0, num_ids) (other ids, e.g. the -1
 * fill, are ignored); counts[id] (optional) = pixel count.  One pass instead of the per-census-row bbox-crop loop of
 * convert_popmap_to_census (data/PopulationDataset.py:705-712).  fp64 accumulation; pred/boundary: any 4-byte aligned
 * address (a row band of a larger map): scalar head up to the first 16-byte boundary, 16-byte reads from there. */
int pc_census_sum(const float* pred, const int32_t* boundary, int64_t n, int num_ids, double* sums, int32_t* counts,
                  void* stream);
/* Dasymetric adjustment, adjust_map_to_census (PopulationDataset.py:823-852): pred[i] *= pop[id] / float(sums[id]) for
 * regions with a census entry (has_entry[id] != 0, NULL = all) and a non-zero total. */
int pc_census_adjust(float* pred, const int32_t* boundary, int64_t n, int num_ids, const double* sums, const float* pop,
                     const uint8_t* has_entry, void* stream);
/* One sliding window of an M-member ensemble into the (H,W) device accumulators (run_eval.py:108-135): interior pixels
 * only (border of `overlap` excluded, PopulationDataset.py:656-672), window origin (yl,xl) = img_coords.
 * popdense/scale: [M][ps_y][ps_x] (scale may be NULL). */
int pc_stitch_accumulate(const float* popdense, const float* scale, int M, int ps_y, int ps_x, int overlap, int yl, int xl,
                         float* out_sum, float* out_sq, float* scale_sum, float* scale_sq, int16_t* count, int H, int W,
                         void* stream);
/* The visit counts of nwin windows that OTHER ranks computed (sharded sliding-window inference: every rank needs the complete count map,
 * run_eval.py:140-154), in one pass: count[r][c] += M for every window whose interior rows [x0, x1) x columns [y0, y1) cover (r, c).
 * win: DEVICE array of nwin x {x0, x1, y0, y1} (already clipped to the raster); overlapping interiors (the catch-up windows at the bottom /
 * right edge) add up.  No scratch plane, no atomics on the map. */
int pc_stitch_count_windows(const int32_t* win, int nwin, int M, int16_t* count, int H, int W, void* stream);
/* run_eval.py:140-154: where count > 1: sum -> mean, sq -> unbiased std; pixels visited once keep their raw values. */
int pc_stitch_finalize(float* out_sum, float* out_sq, float* scale_sum, float* scale_sq, const int16_t* count, int64_t n,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif /* POPCORN_HIP_H */
