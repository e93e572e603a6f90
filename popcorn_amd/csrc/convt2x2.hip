// convt2x2.hip -- ConvTranspose2d(C, C, kernel 2, stride 2) forward / data-gradient / weight-gradient on fp32 MFMA.
//
// Replaces nn.ConvTranspose2d in the decoder's Up block (reference model/DDA_model/utils/networks.py:302,306):
//     out[co][2i+a][2j+b] = bias[co] + sum_ci x[ci][i][j] * w[ci][co][a][b]
// Stride == kernel, so it is a pure per-pixel GEMM: no halo, no LDS staging; fragments are loaded straight from
// global memory (each input element is used by exactly one wave).
//   fwd   : M = 16 consecutive input x,   K = ci,          N = (co,a,b)
//   dgrad : M = 16 consecutive input x,   K = (co,a,b),    N = ci     (+ ReLU/BN backward of the producer fused)
//   wgrad : M = ci,                       K = 4 input x,   N = (co,a,b)   (deterministic two-stage reduction)
#include "common.h"

namespace {

struct CtArgs;
struct CtGroup;            // up to PC_MAX_GROUP CtArgs of identical geometry: blockIdx.y selects the problem

struct CtArgs {
    pc_src x;              // fwd: input;  dgrad: g (2H x 2W);  wgrad: input
    pc_src g;              // wgrad: g (2H x 2W)
    const float* w;        // [C][C][2][2]
    const float* bias;     // fwd
    const float* act;      // dgrad: post-ReLU activations of the layer that produced x (mask), same geometry as out
    int64_t act_bstride, act_cstride;
    int act_rstride, act_xstride, act_dtype;
    pc_bn bn;              // dgrad: BN of that layer
    pc_dst out;
    float* partial;        // wgrad
    int B, H, W;           // input resolution
    int groups_x, ngroups;
    pc_fastdiv div_gx, div_gimg;   // by groups_x, by groups per image
    int bf;                // PC_PREC_BF16: the channels-last bf16 kernels
};

struct CtGroup {
    CtArgs pr[PC_MAX_GROUP];
};

// the problem's descriptor, pinned in scalar registers (common.h: pc_pin; round 6: the group loops re-loaded its fields from the kernel
// arguments 10 - 34 times per iteration)
__device__ __forceinline__ void ct_pin(CtArgs& p) {
    pc_pin(p.x); pc_pin(p.g); pc_pin(p.out);
    p.w = pc_pin_ptr(p.w); p.bias = pc_pin_ptr(p.bias); p.act = pc_pin_ptr(p.act); p.partial = pc_pin_ptr(p.partial);
    pc_pin(p.act_bstride); pc_pin(p.act_cstride); pc_pin(p.act_rstride); pc_pin(p.act_xstride);
    pc_pin(p.B); pc_pin(p.H); pc_pin(p.W); pc_pin(p.groups_x); pc_pin(p.ngroups); pc_pin(p.div_gx); pc_pin(p.div_gimg);
}

// fp32 kernels (planar tensors); the channels-last bf16 kernels of PC_PREC_BF16 follow below
template <int C>
__global__ __launch_bounds__(256) void convt2x2_fwd_kernel(const CtGroup grp_) {
    using act_t = float;
    CtArgs p = grp_.pr[blockIdx.y];
    ct_pin(p);
    constexpr int KS = C / 4, NBK = C / 4;
    const int lane = threadIdx.x & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int gwave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;

    float bw[KS][NBK];
    float binit[NBK];
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) {
        const int ng = nb * 16 + li;
        binit[nb] = p.bias ? p.bias[ng >> 2] : 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float wv = p.w[(4 * ks + lk) * 4 * C + ng];
            bw[ks][nb] = wv;
        }
    }

    for (int grp = gwave; grp < p.ngroups; grp += nwaves) {
        const int b = (int)pc_div((uint32_t)grp, p.div_gimg);
        const int rem = grp - b * p.groups_x * p.H;
        const int i = (int)pc_div((uint32_t)rem, p.div_gx);
        const int gx = rem - i * p.groups_x;
        const int j0 = gx * 16;
        float av[KS];
        const act_t* xp = reinterpret_cast<const act_t*>(p.x.ptr) + b * p.x.bstride + (int64_t)i * p.x.rstride + j0 + li;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) av[ks] = (j0 + li < p.W) ? pc_ld1(xp + (4 * ks + lk) * p.x.cstride) : 0.f;
        f32x4 acc[NBK];
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb) acc[nb] = f32x4{binit[nb], binit[nb], binit[nb], binit[nb]};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int nb = 0; nb < NBK; ++nb)
                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], bw[ks][nb], acc[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb) {
            const int ng = nb * 16 + li;
            const int co = ng >> 2, a = (ng >> 1) & 1, bb = ng & 1;
            // lane pair (bb = 0, 1) holds the even / odd output x of the same 8 consecutive floats: swap halves so that
            // each lane owns 4 consecutive x and issues ONE 16-byte store instead of four strided 4-byte stores
            const f32x4 mine = acc[nb];
            f32x4 other;
#pragma unroll
            for (int r = 0; r < 4; ++r) other[r] = pc_lane_xor1(mine[r]);
            // bb = 0 writes x = 2*jb + {0,1,2,3} (pixels r = 0,1); bb = 1 writes x = 2*jb + {4,5,6,7} (pixels r = 2,3)
            const f32x4 v = bb == 0 ? f32x4{mine[0], other[0], mine[1], other[1]} : f32x4{other[2], mine[2], other[3], mine[3]};
            const int jb = j0 + 4 * lk;
            act_t* op = reinterpret_cast<act_t*>(p.out.ptr) + b * p.out.bstride + co * p.out.cstride + (int64_t)(2 * i + a) * p.out.rstride + 2 * jb + 4 * bb;
            if (jb + 3 < p.W && ((p.out.rstride & 3) == 0) && ((reinterpret_cast<uintptr_t>(op) & 15) == 0)) {
                pc_st4(op, v);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int xo = 2 * jb + 4 * bb + e;            // output x
                    if ((xo >> 1) < p.W) pc_st1(op + e, v[e]);
                }
            }
        }
    }
}

template <int C>
__global__ __launch_bounds__(256) void convt2x2_dgrad_kernel(const CtGroup grp_) {
    using act_t = float;
    CtArgs p = grp_.pr[blockIdx.y];
    ct_pin(p);
    const int lane = threadIdx.x & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int gwave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;

    float bw[C];
#pragma unroll
    for (int co = 0; co < C; ++co) {
        const float wv = li < C ? p.w[(li * C + co) * 4 + lk] : 0.f;
        bw[co] = wv;
    }
    float e_scale = 1.f, e_shift = 0.f;
    if (p.act && li < C) pc_bn_fold(p.bn, li, e_scale, e_shift);

    for (int grp = gwave; grp < p.ngroups; grp += nwaves) {
        const int b = (int)pc_div((uint32_t)grp, p.div_gimg);
        const int rem = grp - b * p.groups_x * p.H;
        const int i = (int)pc_div((uint32_t)rem, p.div_gx);
        const int gx = rem - i * p.groups_x;
        const int j0 = gx * 16;
        // A[pixel li][k = (a,b) = lk] of k-step co
        const act_t* gp = reinterpret_cast<const act_t*>(p.x.ptr) + b * p.x.bstride + (int64_t)(2 * i + (lk >> 1)) * p.x.rstride + 2 * (j0 + li) + (lk & 1);
        const bool ok = j0 + li < p.W;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int co = 0; co < C; ++co) {
            const float av = ok ? pc_ld1(gp + co * p.x.cstride) : 0.f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bw[co], acc, 0, 0, 0);
        }
        if (li < C) {
            const int j = j0 + 4 * lk;
            act_t* op = reinterpret_cast<act_t*>(p.out.ptr) + b * p.out.bstride + li * p.out.cstride + (int64_t)i * p.out.rstride + j;
            const act_t* ap = p.act ? reinterpret_cast<const act_t*>(p.act) + b * p.act_bstride + li * p.act_cstride + (int64_t)i * p.act_rstride + j : nullptr;
            const bool vec = j + 3 < p.W && ((reinterpret_cast<uintptr_t>(op) & 15) == 0) &&
                             (!ap || (reinterpret_cast<uintptr_t>(ap) & 15) == 0);
            if (vec) {
                // the lane's 4 consecutive x of one channel row: one 16-byte load of the mask, one 16-byte store
                f32x4 o = acc;
                if (ap) {
                    const f32x4 a4 = pc_ld4(ap);
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = a4[r] > 0.f ? o[r] * e_scale : 0.f;
                }
                pc_st4(op, o);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (j + r < p.W) {
                        float o = acc[r];
                        if (ap) o = pc_ld1(ap + r) > 0.f ? o * e_scale : 0.f;
                        pc_st1(op + r, o);
                    }
                }
            }
        }
    }
}

template <int C>
struct CtWgradCfg {
    static constexpr int NBK = C / 4;
    static constexpr int E = NBK * 256 + NBK * 64;
};

// =====================================================================================================================
// Channels-last bf16 kernels (PC_PREC_BF16): x / g / out / act are channels-last bf16 tensors (one 16-byte slot per pixel and
// 8 channels, conv3x3.hip).  bf16 MFMA with the pixel on N, so that D hands a lane 4 consecutive channels of one pixel:
//   fwd   : A = weights (M = 16 columns of (a, b, co)), B = 16 input pixels x K = ci: the lane's k-slots are 4 channels of its
//           pixel = one 8-byte load; one v_mfma_f32_16x16x16_bf16 per M tile, 8-byte stores
//   dgrad : K = (a, b, co): k-group lk = the output pixel (a, b) of the 2x2 block, slots = 8 channels = ONE 16-byte load of g;
//           A = w[ci][co][a][b] (M = ci); v_mfma_f32_16x16x32_bf16 per 8 output channels
//   wgrad : contraction over pixels -> both operands are transposed through the wave's LDS region with ds_read_b64_tr_b16
//           (see conv3x3_wgrad.hip): group = 32 input pixels of a row and their 2 x 64 gradient pixels
typedef short ct_s4 __attribute__((ext_vector_type(4)));
typedef short ct_s8 __attribute__((ext_vector_type(8)));
typedef __bf16 ct_bf8 __attribute__((ext_vector_type(8)));
typedef unsigned ct_u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ short ct_bf(float x) { return (short)pc_f2bf(x); }
template <int C>
__global__ __launch_bounds__(256) void convt2x2_fwd_cl_kernel(const CtGroup grp_) {
    CtArgs p = grp_.pr[blockIdx.y];
    ct_pin(p);
    constexpr int NT = C / 4;                 // M tiles of the (a, b, co) space: C = 8: tile = a, m = b*8 + co;  C = 16: tile = (a, b), m = co
    const int lane = threadIdx.x & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int gwave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    // A fragments: lane (m = li, k-group lk): ci = 4*lk + e
    ct_s4 aw[NT];
    float binit[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int a = C == 8 ? t : t >> 1, b = C == 8 ? li >> 3 : t & 1, co = C == 8 ? li & 7 : li;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ci = 4 * lk + e;
            aw[t][e] = ci < C ? ct_bf(p.w[((ci * C + co) * 2 + a) * 2 + b]) : (short)0;
        }
        // D rows of this lane: m = 4*lk + r
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = 4 * lk + r;
            binit[t][r] = p.bias ? p.bias[C == 8 ? (m & 7) : m] : 0.f;
        }
    }
    const pc_bf16_t* const xb = reinterpret_cast<const pc_bf16_t*>(p.x.ptr);
    pc_bf16_t* const ob = reinterpret_cast<pc_bf16_t*>(p.out.ptr);
    for (int grp = gwave; grp < p.ngroups; grp += nwaves) {
        const int b = (int)pc_div((uint32_t)grp, p.div_gimg);
        const int rem = grp - b * p.groups_x * p.H;
        const int i = (int)pc_div((uint32_t)rem, p.div_gx);
        const int j = (rem - i * p.groups_x) * 16 + li;
        const bool ok = j < p.W;
        const bool kok = ok && 4 * lk < C;
        const uint2 xv = *reinterpret_cast<const uint2*>(xb + b * p.x.bstride + (int64_t)i * p.x.rstride + (kok ? (int64_t)j * p.x.xstride + 4 * lk : 0));
        ct_s4 bv;
        bv[0] = kok ? (short)(xv.x & 0xffffu) : (short)0; bv[1] = kok ? (short)(xv.x >> 16) : (short)0;
        bv[2] = kok ? (short)(xv.y & 0xffffu) : (short)0; bv[3] = kok ? (short)(xv.y >> 16) : (short)0;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x4 acc = f32x4{binit[t][0], binit[t][1], binit[t][2], binit[t][3]};
            acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(aw[t], bv, acc, 0, 0, 0);
            const int a = C == 8 ? t : t >> 1, bb = C == 8 ? lk >> 1 : t & 1, co4 = C == 8 ? 4 * (lk & 1) : 4 * lk;
            if (ok) pc_st4(ob + b * p.out.bstride + (int64_t)(2 * i + a) * p.out.rstride + (int64_t)(2 * j + bb) * p.out.xstride + co4, acc);
        }
    }
}

template <int C>
__global__ __launch_bounds__(256) void convt2x2_dgrad_cl_kernel(const CtGroup grp_) {
    CtArgs p = grp_.pr[blockIdx.y];
    ct_pin(p);
    constexpr int NH = C / 8;                 // 8-channel halves of co = MFMAs per group
    const int lane = threadIdx.x & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int gwave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    // A fragments: lane (m = ci = li, k-group lk = (a, b)): slots e = co 8*h + e
    ct_bf8 aw[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        ct_s8 t;
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = li < C ? ct_bf(p.w[(li * C + 8 * h + e) * 4 + lk]) : (short)0;
        aw[h] = __builtin_bit_cast(ct_bf8, t);
    }
    // D rows of this lane: ci = 4*lk + r
    float e_scale[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float sh;
        e_scale[r] = 1.f;
        if (p.act && 4 * lk + r < C) pc_bn_fold(p.bn, 4 * lk + r, e_scale[r], sh);
    }
    const pc_bf16_t* const gb = reinterpret_cast<const pc_bf16_t*>(p.x.ptr);
    const pc_bf16_t* const ab = reinterpret_cast<const pc_bf16_t*>(p.act);
    pc_bf16_t* const ob = reinterpret_cast<pc_bf16_t*>(p.out.ptr);
    for (int grp = gwave; grp < p.ngroups; grp += nwaves) {
        const int b = (int)pc_div((uint32_t)grp, p.div_gimg);
        const int rem = grp - b * p.groups_x * p.H;
        const int i = (int)pc_div((uint32_t)rem, p.div_gx);
        const int j = (rem - i * p.groups_x) * 16 + li;
        const bool ok = j < p.W;
        const pc_bf16_t* gp = gb + b * p.x.bstride + (ok ? (int64_t)(2 * i + (lk >> 1)) * p.x.rstride + (int64_t)(2 * j + (lk & 1)) * p.x.xstride : 0);
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            ct_u4 gv = *reinterpret_cast<const ct_u4*>(gp + 8 * h);
            if (!ok) gv = ct_u4{0u, 0u, 0u, 0u};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[h], __builtin_bit_cast(ct_bf8, gv), acc, 0, 0, 0);
        }
        if (ok && 4 * lk < C) {
            if (ab) {
                const f32x4 a4 = pc_ld4(ab + b * p.act_bstride + (int64_t)i * p.act_rstride + (int64_t)j * p.act_xstride + 4 * lk);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = a4[r] > 0.f ? acc[r] * e_scale[r] : 0.f;
            }
            pc_st4(ob + b * p.out.bstride + (int64_t)i * p.out.rstride + (int64_t)j * p.out.xstride + 4 * lk, acc);
        }
    }
}

__device__ __forceinline__ ct_s4 ct_tr(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ct_s4*)(p));
}

// group = 32 consecutive input pixels of one row.  LDS per wave: x [32 px][C] and g [2 rows][64 px][C] (bf16), plain copies.
// DG: the data gradient of the same 32 pixels from the same two images (pc_convt2x2_bwd_group): convt2x2_dgrad_cl_kernel's MFMA with
// the pixel operand = one 16-byte slot of the g image, the ReLU mask of x's producer from the x image.
template <int C, bool DG>
__global__ __launch_bounds__(256) void convt2x2_wgrad_cl_kernel(const CtGroup grp_) {
    CtArgs p = grp_.pr[blockIdx.y];
    ct_pin(p);
    constexpr int NBK = C / 4, NT = C / 4;    // N tiles: C = 8: tile = a, n = b*8 + co;  C = 16: tile = (a, b), n = co
    constexpr int PB = 2 * C;                 // bytes per pixel
    constexpr int XB = 32 * PB, GB = 2 * 64 * PB, WB = XB + GB;
    using Cfg = CtWgradCfg<C>;
    __shared__ __attribute__((aligned(16))) unsigned char ldsb[4 * WB > 4 * NBK * 256 * 4 ? 4 * WB : 4 * NBK * 256 * 4];
    float* const lds = reinterpret_cast<float*>(ldsb);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int gwave = (blockIdx.x * blockDim.x + tid) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    unsigned char* const wx = ldsb + wave * WB;
    unsigned char* const wg = wx + XB;
    const int groups32 = (p.W + 31) / 32, ngroups = p.B * p.H * groups32;

    f32x4 acc[NT];
    float bsum[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; bsum[t] = 0.f; }
    constexpr int NH = C / 8;
    ct_bf8 aw[DG ? NH : 1];
    float e_scale[4] = {1.f, 1.f, 1.f, 1.f};
    if constexpr (DG) {
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            ct_s8 t;
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = li < C ? ct_bf(p.w[(li * C + 8 * h + e) * 4 + lk]) : (short)0;
            aw[h] = __builtin_bit_cast(ct_bf8, t);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float sh;
            if (p.act && 4 * lk + r < C) pc_bn_fold(p.bn, 4 * lk + r, e_scale[r], sh);
        }
    }
    constexpr int NXP = XB / 16 / 64 > 0 ? XB / 16 / 64 : 1;       // 16-byte pieces per lane: x (32 or 64 pieces), g (128 or 256)
    constexpr int NGP = GB / 16 / 64;
    ct_u4 RX[NXP], RG[NGP];
    const pc_bf16_t* const xb = reinterpret_cast<const pc_bf16_t*>(p.x.ptr);
    const pc_bf16_t* const gb = reinterpret_cast<const pc_bf16_t*>(p.g.ptr);
    auto issue = [&](int grp) {
        const int b = grp / (p.H * groups32);
        const int rem = grp - b * p.H * groups32;
        const int i = rem / groups32, j0 = (rem - i * groups32) * 32;
#pragma unroll
        for (int k = 0; k < NXP; ++k) {
            const int id = lane + 64 * k;                    // piece = (pixel, 8-channel half)
            const int px = id / (C / 8), hf = id % (C / 8);
            const bool ok = id < 32 * (C / 8) && j0 + px < p.W;
            RX[k] = *reinterpret_cast<const ct_u4*>(xb + b * p.x.bstride + (int64_t)i * p.x.rstride + (ok ? (int64_t)(j0 + px) * p.x.xstride + 8 * hf : 0));
            if (!ok) RX[k] = ct_u4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int k = 0; k < NGP; ++k) {
            const int id = lane + 64 * k;
            const int a = id / (64 * (C / 8)), r2 = id % (64 * (C / 8));
            const int px = r2 / (C / 8), hf = r2 % (C / 8);
            const bool ok = 2 * j0 + px < 2 * p.W;
            RG[k] = *reinterpret_cast<const ct_u4*>(gb + b * p.g.bstride + (int64_t)(2 * i + a) * p.g.rstride + (ok ? (int64_t)(2 * j0 + px) * p.g.xstride + 8 * hf : 0));
            if (!ok) RG[k] = ct_u4{0u, 0u, 0u, 0u};
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int k = 0; k < NXP; ++k)
            if (lane + 64 * k < 32 * (C / 8)) *reinterpret_cast<ct_u4*>(wx + (lane + 64 * k) * 16) = RX[k];
#pragma unroll
        for (int k = 0; k < NGP; ++k) *reinterpret_cast<ct_u4*>(wg + (lane + 64 * k) * 16) = RG[k];
    };
    // transposing reads: lane supplies row jj = li >> 2 (pixel 8*lk + 4*e + jj) and column quad q = li & 3
    const int t_j = li >> 2, t_q = li & 3;
    int grp = gwave;
    if (grp < ngroups) issue(grp);
    for (; grp < ngroups; grp += nwaves) {
        commit();
        if (grp + nwaves < ngroups) issue(grp + nwaves);
        if constexpr (DG) {
            const int b = grp / (p.H * groups32);
            const int rem = grp - b * p.H * groups32;
            const int i = rem / groups32, j0 = (rem - i * groups32) * 32;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const int px = 16 * t2 + li;
                f32x4 dacc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    const ct_u4 gv = *reinterpret_cast<const ct_u4*>(wg + ((lk >> 1) * 64 + 2 * px + (lk & 1)) * PB + 16 * h);
                    dacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[h], __builtin_bit_cast(ct_bf8, gv), dacc, 0, 0, 0);
                }
                if (j0 + px < p.W && 4 * lk < C) {
                    if (p.act) {
                        const f32x4 a4 = pc_ld4(reinterpret_cast<const pc_bf16_t*>(wx + px * PB) + 4 * lk);
#pragma unroll
                        for (int r = 0; r < 4; ++r) dacc[r] = a4[r] > 0.f ? dacc[r] * e_scale[r] : 0.f;
                    }
                    pc_st4(reinterpret_cast<pc_bf16_t*>(p.out.ptr) + b * p.out.bstride + (int64_t)i * p.out.rstride +
                               (int64_t)(j0 + px) * p.out.xstride + 4 * lk, dacc);
                }
            }
        }
        // A (M = ci): columns 4q .. 4q+3 of pixel j; for C = 8 the quads 2, 3 repeat 0, 1 (rows 8..15 of D are not used)
        const unsigned char* xa = wx + (8 * lk + t_j) * PB + 8 * (C == 8 ? (t_q & 1) : t_q);
        const ct_s4 alo = ct_tr(xa), ahi = ct_tr(xa + 4 * PB);
        const ct_bf8 av = __builtin_bit_cast(ct_bf8, __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            // B (N tile t): C = 8: quad -> (b = q >> 1, channels 4*(q&1)..) of gradient row 2i + t; C = 16: (a, b) = t, channels 4q..
            const int a = C == 8 ? t : t >> 1;
            const int bq = C == 8 ? t_q >> 1 : t & 1, c4 = C == 8 ? 4 * (t_q & 1) : 4 * t_q;
            const unsigned char* gp = wg + (a * 64 + 2 * (8 * lk + t_j) + bq) * PB + 2 * c4;
            const ct_s4 blo = ct_tr(gp), bhi = ct_tr(gp + 8 * PB);
            float sb = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) sb += __uint_as_float((unsigned)(unsigned short)blo[e] << 16) + __uint_as_float((unsigned)(unsigned short)bhi[e] << 16);
            bsum[t] += sb;
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(ct_bf8, __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7)), acc[t], 0, 0, 0);
        }
    }

    // one partial per workgroup in the layout of the planar kernels: D[m = ci][n = ng & 15] of tile ng >> 4, ng = co*4 + a*2 + b
    float* part = p.partial + (int64_t)blockIdx.x * Cfg::E;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT; ++t) *reinterpret_cast<f32x4*>(&lds[((wave * NT + t) * 64 + lane) * 4]) = acc[t];
    __syncthreads();
    for (int e = tid; e < NBK * 256; e += 256) {
        const int r = e & 3, ln = (e >> 2) & 63, nb = e >> 8;
        const int ci = (ln >> 4) * 4 + r, ng = nb * 16 + (ln & 15);
        const int co = ng >> 2, a = (ng >> 1) & 1, bb = ng & 1;
        const int t = C == 8 ? a : a * 2 + bb, n = C == 8 ? bb * 8 + co : co;
        const int e2 = ((t * 64) + (ci >> 2) * 16 + n) * 4 + (ci & 3);
        part[e] = ((lds[e2] + lds[NT * 256 + e2]) + lds[2 * NT * 256 + e2]) + lds[3 * NT * 256 + e2];
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT; ++t) lds[(wave * NT + t) * 64 + lane] = bsum[t];
    __syncthreads();
    for (int e = tid; e < NBK * 64; e += 256) {
        const int nb = e >> 6, lk2 = (e >> 4) & 3, ng = nb * 16 + (e & 15);
        const int co = ng >> 2, a = (ng >> 1) & 1, bb = ng & 1;
        const int t = C == 8 ? a : a * 2 + bb, n = C == 8 ? bb * 8 + co : co;
        const int e2 = t * 64 + lk2 * 16 + n;
        part[NBK * 256 + e] = ((lds[e2] + lds[NT * 64 + e2]) + lds[2 * NT * 64 + e2]) + lds[3 * NT * 64 + e2];
    }
}

bool ct_cl_ok(const CtArgs& p, bool with_g) {
    if (!pc_cl_ok(p.x)) return false;
    if (with_g) return pc_cl_ok(p.g);
    if (!pc_cl_ok(p.out)) return false;
    if (p.act && !pc_cl_ok(p.act, p.act_dtype, p.act_bstride, p.act_cstride, p.act_rstride, p.act_xstride)) return false;
    return true;
}

// VEC: aligned tensors with W % 16 == 0.  The K dimension of the GEMM is "pixels", and any pixel order works as long as
// both operands use it: lane (., lk) takes pixels 4*lk .. 4*lk+3 of the 16-pixel group, so k-step ks holds pixel 4*lk + ks
// and a lane's four A values are ONE 16-byte load of x, its four B values the even or odd floats of TWO 16-byte loads of
// the g row (4 + 4*C/4 scalar gathers before: 12 / 20 four-byte loads per group and lane).
// DG (with VEC): the data gradient of the same 16 pixels in the same pass (pc_convt2x2_bwd_group).  It contracts the gradient rows
// the weight gradient has just loaded over (co, a, b): they go through a wave-private LDS image [co][a][32 floats] (row pitch 36)
// to become the A operand lane (pixel li, k = (a, b)); the ReLU mask of x's producer is the lane's own x values (D hands lane
// (ci = li, k-group lk) the pixels 4*lk .. +3 -- exactly its A operand of the weight gradient).
constexpr int CT_GLRS = 36;
template <int C, bool VEC, bool DG>
__global__ __launch_bounds__(256) void convt2x2_wgrad_kernel(const CtGroup grp_) {
    using act_t = float;
    CtArgs p = grp_.pr[blockIdx.y];
    ct_pin(p);
    constexpr int NBK = C / 4;
    using Cfg = CtWgradCfg<C>;
    constexpr int GLW = C * 2 * CT_GLRS;                   // floats of a wave's gradient image (DG)
    __shared__ __attribute__((aligned(16))) float lds[(DG && 4 * GLW > 4 * NBK * 256) ? 4 * GLW : 4 * NBK * 256];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int gwave = (blockIdx.x * blockDim.x + tid) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;

    f32x4 acc[NBK];
    float bsum[NBK];
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) { acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f}; bsum[nb] = 0.f; }
    // DG: B fragments of the data gradient, lane (k = (a, b) = lk, n = ci = li), one per co; BN scale of x's producer
    float bwd[DG ? C : 1];
    float e_scale = 1.f;
    float* const gl = lds + wave * GLW;
    if constexpr (DG) {
#pragma unroll
        for (int co = 0; co < C; ++co) bwd[co] = li < C ? p.w[(li * C + co) * 4 + lk] : 0.f;
        float e_shift;
        if (p.act && li < C) pc_bn_fold(p.bn, li, e_scale, e_shift);
    }

    // group = 16 consecutive input x of one row; 4 k-steps of 4 pixels each
    for (int grp = gwave; grp < p.ngroups; grp += nwaves) {
        const int b = (int)pc_div((uint32_t)grp, p.div_gimg);
        const int rem = grp - b * p.groups_x * p.H;
        const int i = (int)pc_div((uint32_t)rem, p.div_gx);
        const int gx = rem - i * p.groups_x;
        const int j0 = gx * 16;
        const act_t* xp = reinterpret_cast<const act_t*>(p.x.ptr) + b * p.x.bstride + li * p.x.cstride + (int64_t)i * p.x.rstride;
        const act_t* gbase = reinterpret_cast<const act_t*>(p.g.ptr);
        float av[4], bv[4][NBK];
        if (VEC) {
            const f32x4 xv = li < C ? pc_ld4(xp + j0 + 4 * lk) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) av[ks] = xv[ks];
#pragma unroll
            for (int nb = 0; nb < NBK; ++nb) {
                const int ng = nb * 16 + li;
                const int co = ng >> 2, a = (ng >> 1) & 1, bb = ng & 1;
                const act_t* gp = gbase + b * p.g.bstride + co * p.g.cstride + (int64_t)(2 * i + a) * p.g.rstride + 2 * j0 + 8 * lk;
                const f32x4 g0 = pc_ld4(gp), g1 = pc_ld4(gp + 4);
                if constexpr (DG) {
                    if (bb == 0) {                          // the b = 1 lane of the pair holds the same eight floats
                        float* d = gl + (co * 2 + a) * CT_GLRS + 8 * lk;
                        *reinterpret_cast<f32x4*>(d) = g0;
                        *reinterpret_cast<f32x4*>(d + 4) = g1;
                    }
                }
                bv[0][nb] = bb ? g0[1] : g0[0];
                bv[1][nb] = bb ? g0[3] : g0[2];
                bv[2][nb] = bb ? g1[1] : g1[0];
                bv[3][nb] = bb ? g1[3] : g1[2];
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int j = j0 + 4 * ks + lk;
                av[ks] = (li < C && j < p.W) ? pc_ld1(xp + j) : 0.f;
#pragma unroll
                for (int nb = 0; nb < NBK; ++nb) {
                    const int ng = nb * 16 + li;
                    const int co = ng >> 2, a = (ng >> 1) & 1, bb = ng & 1;
                    bv[ks][nb] = j < p.W ? pc_ld1(gbase + b * p.g.bstride + co * p.g.cstride + (int64_t)(2 * i + a) * p.g.rstride + 2 * j + bb)
                                         : 0.f;
                }
            }
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int nb = 0; nb < NBK; ++nb) {
                bsum[nb] += bv[ks][nb];
                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], bv[ks][nb], acc[nb], 0, 0, 0);
            }
        if constexpr (DG && VEC) {
            // gx[ci][i][j] = sum_{co,a,b} w[ci][co][a][b] * g[co][2i+a][2j+b]   (DS operations of a wave execute in order: no barrier)
            f32x4 dacc = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* ga = gl + (lk >> 1) * CT_GLRS + 2 * li + (lk & 1);
#pragma unroll
            for (int co = 0; co < C; ++co) dacc = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[co * 2 * CT_GLRS], bwd[co], dacc, 0, 0, 0);
            if (li < C) {
                if (p.act) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) dacc[r] = av[r] > 0.f ? dacc[r] * e_scale : 0.f;
                }
                pc_st4(reinterpret_cast<float*>(p.out.ptr) + b * p.out.bstride + li * p.out.cstride + (int64_t)i * p.out.rstride + j0 + 4 * lk, dacc);
            }
        }
    }

    float* part = p.partial + (int64_t)blockIdx.x * Cfg::E;
    if constexpr (DG) __syncthreads();         // the reduction slices overlap the other waves' gradient images
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) *reinterpret_cast<f32x4*>(&lds[((wave * NBK + nb) * 64 + lane) * 4]) = acc[nb];
    __syncthreads();
    for (int e = tid; e < NBK * 256; e += 256)
        part[e] = ((lds[e] + lds[NBK * 256 + e]) + lds[2 * NBK * 256 + e]) + lds[3 * NBK * 256 + e];
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) lds[(wave * NBK + nb) * 64 + lane] = bsum[nb];
    __syncthreads();
    for (int e = tid; e < NBK * 64; e += 256)
        part[NBK * 256 + e] = ((lds[e] + lds[NBK * 64 + e]) + lds[2 * NBK * 64 + e]) + lds[3 * NBK * 64 + e];
}

struct CtReduceArgs {
    const float* partial;
    int nwg;
    float* dw;
    float* db;
    int accumulate;
};

template <int C>
__global__ __launch_bounds__(256) void convt2x2_wgrad_reduce_kernel(const CtReduceArgs p) {
    constexpr int NBK = C / 4;
    using Cfg = CtWgradCfg<C>;
    __shared__ float red[256];
    const int tid = threadIdx.x;
    const int slice = tid >> 4, o = blockIdx.x * 16 + (tid & 15);     // 16 outputs x 16 slices
    constexpr int n_w = C * C * 4, n_out = n_w + C;
    float s0 = 0.f, s1 = 0.f;
    if (o < n_w) {
        const int ci = o / (4 * C), ng = o % (4 * C);     // dw[ci][co][a][b], ng = co*4 + a*2 + b
        const int e = (((ng >> 4) * 64) + (ci >> 2) * 16 + (ng & 15)) * 4 + (ci & 3);
        int w = slice;
        for (; w + 16 < p.nwg; w += 32) {
            s0 += p.partial[(int64_t)w * Cfg::E + e];
            s1 += p.partial[(int64_t)(w + 16) * Cfg::E + e];
        }
        for (; w < p.nwg; w += 16) s0 += p.partial[(int64_t)w * Cfg::E + e];
    } else if (o < n_out) {
        const int co = o - n_w;
        for (int w = slice; w < p.nwg; w += 16) {
            const float* q = p.partial + (int64_t)w * Cfg::E + NBK * 256;
            float t = 0.f;
#pragma unroll
            for (int ab = 0; ab < 4; ++ab) {
                const int ng = co * 4 + ab;
#pragma unroll
                for (int lk = 0; lk < 4; ++lk) t += q[(ng >> 4) * 64 + lk * 16 + (ng & 15)];
            }
            s0 += t;
        }
    }
    red[tid] = s0 + s1;
    __syncthreads();
    if (tid < 16 && o < n_out) {
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) tot += red[k * 16 + tid];
        float* dstp = o < n_w ? p.dw + o : p.db + (o - n_w);
        if (o >= n_w && p.db == nullptr) return;
        *dstp = p.accumulate ? *dstp + tot : tot;
    }
}

constexpr int CT_MAX_WG = 512;

int fill_groups(CtArgs& p) {
    p.groups_x = (p.W + 15) / 16;
    p.ngroups = p.B * p.H * p.groups_x;
    p.div_gx = pc_make_fastdiv(p.groups_x);
    p.div_gimg = pc_make_fastdiv(p.groups_x * p.H);
    int nwg = (p.ngroups + 3) / 4;
    if (nwg > 2048) nwg = 2048;
    return nwg < 1 ? 1 : nwg;
}

}  // namespace

extern "C" int pc_convt2x2_fwd_group(int n, const pc_convt_fwd_desc* d, int B, int H, int W, int C, void* stream) {
    if (n < 1 || n > PC_MAX_GROUP || !d) return PC_EINVAL;
    CtGroup g{};
    int nwg = 1;
    for (int i = 0; i < n; ++i) {
        if (!d[i].x || !d[i].w || !d[i].out) return PC_EINVAL;
        CtArgs& p = g.pr[i];
        p.x = *d[i].x; p.w = d[i].w; p.bias = d[i].bias; p.out = *d[i].out; p.B = B; p.H = H; p.W = W; p.bf = g_pc_precision == PC_PREC_BF16;
        nwg = fill_groups(p);
    }
    nwg = (nwg + n - 1) / n < 64 ? nwg : (nwg + n - 1) / n;     // keep the total grid size roughly constant
    const bool bf = g_pc_precision == PC_PREC_BF16;
    for (int i = 0; i < n; ++i) {
        if (g.pr[i].x.dtype != (bf ? PC_BF16 : PC_F32) || g.pr[i].out.dtype != (bf ? PC_BF16 : PC_F32)) return PC_EINVAL;
        if (bf ? !ct_cl_ok(g.pr[i], false) : !(pc_planar(g.pr[i].x) && pc_planar(g.pr[i].out))) return PC_EINVAL;
    }
    if (C == 16 && bf) hipLaunchKernelGGL((convt2x2_fwd_cl_kernel<16>), dim3(nwg, n), dim3(256), 0, (hipStream_t)stream, g);
    else if (C == 16) hipLaunchKernelGGL((convt2x2_fwd_kernel<16>), dim3(nwg, n), dim3(256), 0, (hipStream_t)stream, g);
    else if (C == 8 && bf) hipLaunchKernelGGL((convt2x2_fwd_cl_kernel<8>), dim3(nwg, n), dim3(256), 0, (hipStream_t)stream, g);
    else if (C == 8) hipLaunchKernelGGL((convt2x2_fwd_kernel<8>), dim3(nwg, n), dim3(256), 0, (hipStream_t)stream, g);
    else return PC_EINVAL;
    PC_CHECK_LAUNCH();
    return 0;
}

extern "C" int pc_convt2x2_fwd(const pc_src* x, const float* w, const float* bias, const pc_dst* out, int B, int H, int W,
                               int C, void* stream) {
    pc_convt_fwd_desc d{x, w, bias, out};
    return pc_convt2x2_fwd_group(1, &d, B, H, W, C, stream);
}

extern "C" int pc_convt2x2_dgrad_group(int n, const pc_convt_dgrad_desc* d, int B, int H, int W, int C, void* stream) {
    if (n < 1 || n > PC_MAX_GROUP || !d) return PC_EINVAL;
    CtGroup g{};
    int nwg = 1;
    for (int i = 0; i < n; ++i) {
        if (!d[i].g || !d[i].w || !d[i].out) return PC_EINVAL;
        CtArgs& p = g.pr[i];
        p.x = *d[i].g; p.w = d[i].w; p.out = *d[i].out; p.B = B; p.H = H; p.W = W; p.bf = g_pc_precision == PC_PREC_BF16;
        if (d[i].act) {
            if (!d[i].act_bn) return PC_EINVAL;
            p.act = d[i].act->ptr; p.act_bstride = d[i].act->bstride; p.act_cstride = d[i].act->cstride;
            p.act_rstride = d[i].act->rstride; p.act_xstride = d[i].act->xstride; p.act_dtype = d[i].act->dtype;
            if (d[i].act->dtype != (g_pc_precision == PC_PREC_BF16 ? PC_BF16 : PC_F32)) return PC_EINVAL;
            p.bn = *d[i].act_bn;
        }
        nwg = fill_groups(p);
    }
    nwg = (nwg + n - 1) / n < 64 ? nwg : (nwg + n - 1) / n;
    const bool bf = g_pc_precision == PC_PREC_BF16;
    for (int i = 0; i < n; ++i) {
        if (g.pr[i].x.dtype != (bf ? PC_BF16 : PC_F32) || g.pr[i].out.dtype != (bf ? PC_BF16 : PC_F32)) return PC_EINVAL;
        if (bf ? !ct_cl_ok(g.pr[i], false) : !(pc_planar(g.pr[i].x) && pc_planar(g.pr[i].out) && g.pr[i].act_xstride <= 1)) return PC_EINVAL;
    }
    if (C == 16 && bf) hipLaunchKernelGGL((convt2x2_dgrad_cl_kernel<16>), dim3(nwg, n), dim3(256), 0, (hipStream_t)stream, g);
    else if (C == 16) hipLaunchKernelGGL((convt2x2_dgrad_kernel<16>), dim3(nwg, n), dim3(256), 0, (hipStream_t)stream, g);
    else if (C == 8 && bf) hipLaunchKernelGGL((convt2x2_dgrad_cl_kernel<8>), dim3(nwg, n), dim3(256), 0, (hipStream_t)stream, g);
    else if (C == 8) hipLaunchKernelGGL((convt2x2_dgrad_kernel<8>), dim3(nwg, n), dim3(256), 0, (hipStream_t)stream, g);
    else return PC_EINVAL;
    PC_CHECK_LAUNCH();
    return 0;
}

extern "C" int pc_convt2x2_dgrad(const pc_src* g, const float* w, const pc_src* act, const pc_bn* act_bn,
                                 const pc_dst* out, int B, int H, int W, int C, void* stream) {
    pc_convt_dgrad_desc d{g, w, act, act_bn, out};
    return pc_convt2x2_dgrad_group(1, &d, B, H, W, C, stream);
}

extern "C" int64_t pc_convt2x2_wgrad_ws_bytes(int C) {
    (void)C;
    return (int64_t)CT_MAX_WG * (4 * 256 + 4 * 64) * sizeof(float);
}

namespace {
bool ct_wgrad_vec_ok(const CtArgs& p) {
    auto al = [](const pc_src& s) {
        return ((reinterpret_cast<uintptr_t>(s.ptr) & 15) == 0) && s.rstride % 4 == 0 && s.cstride % 4 == 0 &&
               s.bstride % 4 == 0;
    };
    return p.W % 16 == 0 && al(p.x) && al(p.g);
}
int launch_ct_wgrad_group(const CtGroup& g, int n, int C, int nwg, hipStream_t st, bool dg = false) {
    bool vec = true;
    const bool bf = g_pc_precision == PC_PREC_BF16;
    for (int i = 0; i < n; ++i) {
        vec = vec && ct_wgrad_vec_ok(g.pr[i]);
        if (g.pr[i].x.dtype != (bf ? PC_BF16 : PC_F32) || g.pr[i].g.dtype != (bf ? PC_BF16 : PC_F32)) return PC_EINVAL;
        if (bf ? !ct_cl_ok(g.pr[i], true) : !(pc_planar(g.pr[i].x) && pc_planar(g.pr[i].g))) return PC_EINVAL;
    }
    if (bf) {
        if (C == 16 && dg) hipLaunchKernelGGL((convt2x2_wgrad_cl_kernel<16, true>), dim3(nwg, n), dim3(256), 0, st, g);
        else if (C == 8 && dg) hipLaunchKernelGGL((convt2x2_wgrad_cl_kernel<8, true>), dim3(nwg, n), dim3(256), 0, st, g);
        else if (C == 16) hipLaunchKernelGGL((convt2x2_wgrad_cl_kernel<16, false>), dim3(nwg, n), dim3(256), 0, st, g);
        else if (C == 8) hipLaunchKernelGGL((convt2x2_wgrad_cl_kernel<8, false>), dim3(nwg, n), dim3(256), 0, st, g);
        else return PC_EINVAL;
        PC_CHECK_LAUNCH();
        return 0;
    }
    if (dg && !vec) return PC_EINVAL;          // the fused form exists for aligned tensors with W % 16 == 0 only
#define PC_CTW(CC, VV, DD) hipLaunchKernelGGL((convt2x2_wgrad_kernel<CC, VV, DD>), dim3(nwg, n), dim3(256), 0, st, g)
    if (C == 16) {
        if (dg) PC_CTW(16, true, true); else if (vec) PC_CTW(16, true, false); else PC_CTW(16, false, false);
    } else if (C == 8) {
        if (dg) PC_CTW(8, true, true); else if (vec) PC_CTW(8, true, false); else PC_CTW(8, false, false);
    } else return PC_EINVAL;
#undef PC_CTW
    PC_CHECK_LAUNCH();
    return 0;
}
int launch_ct_wgrad(const CtArgs& p, int C, int nwg, hipStream_t st) {
    CtGroup g{};
    g.pr[0] = p;
    return launch_ct_wgrad_group(g, 1, C, nwg, st);
}
}  // namespace

extern "C" int pc_convt2x2_wgrad_partial(const pc_src* x, const pc_src* g, void* ws, int B, int H, int W, int C, int* nwg_out,
                                         void* stream) {
    if (!x || !g || !ws || !nwg_out) return PC_EINVAL;
    CtArgs p{};
    p.x = *x; p.g = *g; p.B = B; p.H = H; p.W = W; p.bf = g_pc_precision == PC_PREC_BF16;
    p.partial = reinterpret_cast<float*>(ws);
    int nwg = fill_groups(p);
    if (nwg > CT_MAX_WG) nwg = CT_MAX_WG;
    const int rc = launch_ct_wgrad(p, C, nwg, (hipStream_t)stream);
    if (rc) return rc;
    *nwg_out = nwg;
    return 0;
}

extern "C" int pc_convt2x2_wgrad_partial_group(int n, const pc_convt_wgrad_desc* d, int B, int H, int W, int C, int* nwg_out,
                                               void* stream) {
    if (n < 1 || n > PC_MAX_GROUP || !d || !nwg_out) return PC_EINVAL;
    CtGroup g{};
    int nwg = 1;
    for (int i = 0; i < n; ++i) {
        if (!d[i].x || !d[i].g || !d[i].ws) return PC_EINVAL;
        CtArgs& p = g.pr[i];
        p.x = *d[i].x; p.g = *d[i].g; p.B = B; p.H = H; p.W = W; p.bf = g_pc_precision == PC_PREC_BF16;
        p.partial = reinterpret_cast<float*>(d[i].ws);
        nwg = fill_groups(p);
    }
    if (nwg > CT_MAX_WG / n) nwg = CT_MAX_WG / n;      // the same total number of workgroups as a single-problem launch
    if (nwg < 1) nwg = 1;
    const int rc = launch_ct_wgrad_group(g, n, C, nwg, (hipStream_t)stream);
    if (rc) return rc;
    *nwg_out = nwg;
    return 0;
}

extern "C" int pc_convt2x2_bwd_group(int n, const pc_convt_bwd_desc* d, int B, int H, int W, int C, int* nwg_out, void* stream) {
    if (n < 1 || n > PC_MAX_GROUP || !d || !nwg_out) return PC_EINVAL;
    const bool bf = g_pc_precision == PC_PREC_BF16;
    CtGroup g{};
    int nwg = 1;
    for (int i = 0; i < n; ++i) {
        if (!d[i].x || !d[i].g || !d[i].w || !d[i].out || !d[i].ws) return PC_EINVAL;
        CtArgs& p = g.pr[i];
        p.x = *d[i].x; p.g = *d[i].g; p.w = d[i].w; p.out = *d[i].out; p.B = B; p.H = H; p.W = W; p.bf = bf;
        p.partial = reinterpret_cast<float*>(d[i].ws);
        if (d[i].x_bn) {                       // the mask is x itself (post-ReLU output of the layer x_bn belongs to)
            p.act = p.x.ptr;
            p.bn = *d[i].x_bn;
        }
        if (p.out.dtype != (bf ? PC_BF16 : PC_F32)) return PC_EINVAL;
        if (bf) {
            if (!pc_cl_ok(p.out)) return PC_EINVAL;
        } else if (!pc_planar(p.out) || (reinterpret_cast<uintptr_t>(p.out.ptr) & 15) || p.out.rstride % 4 || p.out.cstride % 4 ||
                   p.out.bstride % 4) {
            return PC_EINVAL;
        }
        nwg = fill_groups(p);
    }
    if (nwg > CT_MAX_WG / n) nwg = CT_MAX_WG / n;
    if (nwg < 1) nwg = 1;
    const int rc = launch_ct_wgrad_group(g, n, C, nwg, (hipStream_t)stream, true);
    if (rc) return rc;
    *nwg_out = nwg;
    return 0;
}

extern "C" int pc_convt2x2_wgrad(const pc_src* x, const pc_src* g, float* dw, float* db, int accumulate, void* ws,
                                 int B, int H, int W, int C, void* stream) {
    if (!x || !g || !dw || !ws) return PC_EINVAL;
    CtArgs p{};
    p.x = *x; p.g = *g; p.B = B; p.H = H; p.W = W; p.bf = g_pc_precision == PC_PREC_BF16;
    p.partial = reinterpret_cast<float*>(ws);
    int nwg = fill_groups(p);
    if (nwg > CT_MAX_WG) nwg = CT_MAX_WG;
    CtReduceArgs r{};
    r.partial = p.partial; r.nwg = nwg; r.dw = dw; r.db = db; r.accumulate = accumulate;
    hipStream_t st = (hipStream_t)stream;
    const int rc = launch_ct_wgrad(p, C, nwg, st);
    if (rc) return rc;
    if (C == 16) hipLaunchKernelGGL(convt2x2_wgrad_reduce_kernel<16>, dim3((16 * 16 * 4 + 16 + 15) / 16), dim3(256), 0, st, r);
    else hipLaunchKernelGGL(convt2x2_wgrad_reduce_kernel<8>, dim3((8 * 8 * 4 + 8 + 15) / 16), dim3(256), 0, st, r);
    PC_CHECK_LAUNCH();
    return 0;
}
