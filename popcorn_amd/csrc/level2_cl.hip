// level2_cl.hip -- the whole 32 x 32 level of a U-Net stream in ONE launch, PC_PREC_BF16 form (channels-last bf16 activations).
//
// Counterpart of level2.hip for the bf16 mode: for a pooled 16-channel 32 x 32 map
//     Down.mpconv[1] = DoubleConv(16, 16)           (reference model/DDA_model/utils/networks.py:253-271,284-295)
//     Up.up          = ConvTranspose2d(16, 16, 2, 2) (networks.py:302,306)
// = conv3x3+BN+ReLU -> conv3x3+BN+ReLU -> convT 2x2: three launches of the layer-by-layer path, of which the two convolutions
// run 6 us of fixed cost for 3 us of traffic each at this size.  A per-tile map is 16 ch x 32 x 32 x 2 B = 32 KB: one 512-thread
// workgroup owns one (tile, network-stream) problem; the input and both intermediates live in LDS as the strip image of
// conv3x3_cl_kernel extended to the whole tile -- [8-channel chunk][34 rows][48 slots] of 16 bytes, data at rows 1..32 / slots
// 4..35, zero halo -- so every wave runs that kernel's 32 x 4 strip arithmetic (v_mfma_f32_16x16x32_bf16, K = 4 input rows x 8
// channels, A = weights, B = pixels) straight from the shared image: rows 4 w .. 4 w + 3 for wave w.  The epilogue rounds to bf16
// (the mode's rounding point) into the other image for the next stage and, for networks whose backward needs them, to c1 / c2 in
// HBM; the transposed conv reads conv2's image (v_mfma_f32_16x16x16_bf16, K = ci: 8 bytes of the pixel's slot per lane) and
// writes u2.  Same instruction order per accumulator as the three kernels it replaces: results are BIT-identical to them.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int LC_SL = 48;                    // slots per image row (== 0 mod 16, as the strip rows of conv3x3_cl_kernel)
constexpr int LC_ROWS = 34;
constexpr int LC_COL0 = 4;                   // slot of x = 0 (left halo at slot 3, right halo at slot 36)
constexpr int LC_CH = LC_ROWS * LC_SL;       // slots of one 8-channel chunk image
constexpr int LC_IMG = 2 * LC_CH;            // slots of a 16-channel map image (52,224 bytes)
constexpr int LC_WCO = 48;                   // weight image: halfwords per output channel (2 chunks x 3 dx x 8 ci)
constexpr int LC_WDY = 16 * LC_WCO;          // halfwords per dy plane (4 planes, plane 3 all zero)
constexpr int LC_WIMG = 4 * LC_WDY;          // halfwords per conv
constexpr size_t LC_LDS = (size_t)2 * LC_IMG * 16 + (size_t)2 * LC_WIMG * 2;      // 116,736 bytes: one workgroup per CU

struct LcProb {
    const pc_bf16_t* x; int64_t x_bs; int x_rs, x_xs;            // pooled input (B,16,32,32) channels-last bf16
    const float* w1; const float* w2; const float* wt; const float* bt;
    pc_bn bn1, bn2;
    pc_bf16_t* c1; int64_t c1_bs; int c1_rs, c1_xs;              // NULL = not saved
    pc_bf16_t* c2; int64_t c2_bs; int c2_rs, c2_xs;
    pc_bf16_t* u2; int64_t u2_bs; int u2_rs, u2_xs;              // (B,16,64,64), NULL = no transposed conv
};
struct LcArgs { LcProb pr[PC_MAX_GROUP]; };

__global__ __launch_bounds__(512) void level2_fwd_cl_kernel(const LcArgs args) {
    extern __shared__ __attribute__((aligned(16))) u32x4 lds4[];
    const LcProb& q = args.pr[blockIdx.y];
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    u32x4* const imgA = lds4;
    u32x4* const imgB = lds4 + LC_IMG;
    unsigned short* const w1h = reinterpret_cast<unsigned short*>(lds4 + 2 * LC_IMG);
    unsigned short* const w2h = w1h + LC_WIMG;

    // ---- input tile: 1024 pixels x 2 chunks = 2048 16-byte pieces, 4 per thread (32 contiguous bytes per pixel)
    u32x4 xr[4];
    {
        const pc_bf16_t* xp = q.x + b * q.x_bs;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 512 * i, ch = idx & 1, px = idx >> 1;
            xr[i] = *reinterpret_cast<const u32x4*>(xp + (int64_t)(px >> 5) * q.x_rs + (int64_t)(px & 31) * q.x_xs + 8 * ch);
        }
    }
    // ---- both convolutions' weights ([16][16][3][3] fp32): 2 x 2304 values, 9 per thread
    float wr[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int e = tid + 512 * k;
        wr[k] = e < 2304 ? q.w1[e] : q.w2[e - 2304];
    }
    // per-lane epilogue constants (channels nb * 8 + 4 * (lk & 1) + r), both layers
    const int c4 = 4 * (lk & 1);
    float sc1[2][4], sh1[2][4], sc2[2][4], sh2[2][4];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pc_bn_fold(q.bn1, nb * 8 + c4 + r, sc1[nb][r], sh1[nb][r]);
            pc_bn_fold(q.bn2, nb * 8 + c4 + r, sc2[nb][r], sh2[nb][r]);
        }
    // transposed-conv A fragments (M tile t = (a, b), m = co = li; k-group lk: ci = 4 lk + e) and bias of the D rows co = 4 lk + r
    s16x4 aw[4];
    float bint[4];
    if (q.u2) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) aw[t][e] = (short)pc_f2bf(q.wt[(((4 * lk + e) * 16 + li) * 2 + (t >> 1)) * 2 + (t & 1)]);
#pragma unroll
        for (int r = 0; r < 4; ++r) bint[r] = q.bt ? q.bt[4 * lk + r] : 0.f;
    }
    // ---- zero both images (halo + everything else) and both weight images (dy plane 3 stays zero)
    for (int e = tid; e < 2 * LC_IMG + (2 * LC_WIMG * 2) / 16; e += 512) lds4[e] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 512 * i, ch = idx & 1, px = idx >> 1;
        imgA[ch * LC_CH + ((px >> 5) + 1) * LC_SL + LC_COL0 + (px & 31)] = xr[i];
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int e0 = tid + 512 * k;
        unsigned short* wh = e0 < 2304 ? w1h : w2h;
        const int e = e0 < 2304 ? e0 : e0 - 2304;
        const int tap = e % 9, ci = (e / 9) % 16, co = e / 144;
        wh[(tap / 3) * LC_WDY + co * LC_WCO + (ci >> 3) * 24 + (tap % 3) * 8 + (ci & 7)] = pc_f2bf(wr[k]);
    }
    __syncthreads();

    // ---- one conv3x3 + BN + ReLU over the whole tile: wave w owns output rows 4 w .. 4 w + 3 (the 32 x 4 strip of conv3x3_cl_kernel)
    const int y0 = 4 * wave;
    const int a_s = li >> 3, a_co = li & 7;
    const int wplane = ((unsigned)(lk - a_s) <= 2u) ? lk - a_s : 3;
    auto conv = [&](const u32x4* src, u32x4* dst, const unsigned short* wh, const float (&sc)[2][4], const float (&sh)[2][4],
                    pc_bf16_t* outp, int64_t o_bs, int o_rs, int o_xs) {
        f32x4 acc[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[u][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned short* wl = wh + wplane * LC_WDY + a_co * LC_WCO;
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            bf16x8 bw[3][2];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
                    bw[dx][nb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wl + nb * 8 * LC_WCO + ch * 24 + dx * 8));
            const u32x4* lrow = src + ch * LC_CH + (y0 + lk) * LC_SL + (LC_COL0 - 1) + li;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                bf16x8 av[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) av[u] = __builtin_bit_cast(bf16x8, lrow[(u >> 1) * 2 * LC_SL + (u & 1) * 16 + dx]);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
                        acc[u][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[dx][nb], av[u], acc[u][nb], 0, 0, 0);
            }
        }
        // D: lane (x = li, lk), register r: row s = lk >> 1 of the pair, channel 4 * (lk & 1) + r
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int y = y0 + 2 * (u >> 1) + (lk >> 1), x = (u & 1) * 16 + li;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                f32x4 v = acc[u][nb];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r] * sc[nb][r] + sh[nb][r], 0.f);
                const uint2 pk = make_uint2(pc_pack_bf16(v[0], v[1]), pc_pack_bf16(v[2], v[3]));
                reinterpret_cast<uint2*>(dst + nb * LC_CH + (y + 1) * LC_SL + LC_COL0 + x)[lk & 1] = pk;
                if (outp) *reinterpret_cast<uint2*>(outp + b * o_bs + (int64_t)y * o_rs + (int64_t)x * o_xs + nb * 8 + c4) = pk;
            }
        }
    };
    conv(imgA, imgB, w1h, sc1, sh1, q.c1, q.c1_bs, q.c1_rs, q.c1_xs);
    __syncthreads();
    conv(imgB, imgA, w2h, sc2, sh2, q.c2, q.c2_bs, q.c2_rs, q.c2_xs);
    if (!q.u2) return;
    __syncthreads();

    // ---- ConvTranspose2d(16, 16, 2, 2) of conv2's image: 64 groups of 16 pixels (row i, half row), 8 per wave
#pragma unroll 2
    for (int gi = 0; gi < 8; ++gi) {
        const int g = wave * 8 + gi, i = g >> 1, j = (g & 1) * 16 + li;
        const uint2 xv = reinterpret_cast<const uint2*>(imgA + (lk >> 1) * LC_CH + (i + 1) * LC_SL + LC_COL0 + j)[lk & 1];
        s16x4 bv;
        bv[0] = (short)(xv.x & 0xffffu); bv[1] = (short)(xv.x >> 16); bv[2] = (short)(xv.y & 0xffffu); bv[3] = (short)(xv.y >> 16);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 acc = f32x4{bint[0], bint[1], bint[2], bint[3]};
            acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(aw[t], bv, acc, 0, 0, 0);
            pc_st4(q.u2 + b * q.u2_bs + (int64_t)(2 * i + (t >> 1)) * q.u2_rs + (int64_t)(2 * j + (t & 1)) * q.u2_xs + 4 * lk, acc);
        }
    }
}

bool cl16(const void* ptr, int dtype, int64_t bs, int64_t cs, int rs, int xs) { return pc_cl_ok(ptr, dtype, bs, cs, rs, xs) && xs >= 16; }

}  // namespace

// called by level2.hip's extern "C" entry points when the arithmetic mode is PC_PREC_BF16
bool pc_level2_fwd_cl_ok(const pc_src* x, const pc_dst* u2) {
    if (!x || x->C != 16 || x->H != 32 || x->W != 32 || x->mode != PC_SRC_DIRECT || x->oy || x->ox) return false;
    if (!cl16(x->ptr, x->dtype, x->bstride, x->cstride, x->rstride, x->xstride)) return false;
    return !u2 || cl16(u2->ptr, u2->dtype, u2->bstride, u2->cstride, u2->rstride, u2->xstride);
}

int pc_level2_fwd_cl_launch(int n, const pc_level2_fwd_desc* d, int B, hipStream_t stream) {
    LcArgs a;
    for (int i = 0; i < n; ++i) {
        const pc_level2_fwd_desc& s = d[i];
        if (!s.x || (!s.u2 && !s.c2) || !s.w1 || !s.w2 || !s.wt || !s.bn1 || !s.bn2 || !pc_level2_fwd_cl_ok(s.x, s.u2)) return PC_EINVAL;
        LcProb& p = a.pr[i];
        p = LcProb{};
        p.x = reinterpret_cast<const pc_bf16_t*>(s.x->ptr); p.x_bs = s.x->bstride; p.x_rs = s.x->rstride; p.x_xs = s.x->xstride;
        p.w1 = s.w1; p.w2 = s.w2; p.wt = s.wt; p.bt = s.bt; p.bn1 = *s.bn1; p.bn2 = *s.bn2;
        if (s.c1) {
            if (!cl16(s.c1->ptr, s.c1->dtype, s.c1->bstride, s.c1->cstride, s.c1->rstride, s.c1->xstride)) return PC_EINVAL;
            p.c1 = reinterpret_cast<pc_bf16_t*>(s.c1->ptr); p.c1_bs = s.c1->bstride; p.c1_rs = s.c1->rstride; p.c1_xs = s.c1->xstride;
        }
        if (s.c2) {
            if (!cl16(s.c2->ptr, s.c2->dtype, s.c2->bstride, s.c2->cstride, s.c2->rstride, s.c2->xstride)) return PC_EINVAL;
            p.c2 = reinterpret_cast<pc_bf16_t*>(s.c2->ptr); p.c2_bs = s.c2->bstride; p.c2_rs = s.c2->rstride; p.c2_xs = s.c2->xstride;
        }
        if (s.u2) { p.u2 = reinterpret_cast<pc_bf16_t*>(s.u2->ptr); p.u2_bs = s.u2->bstride; p.u2_rs = s.u2->rstride; p.u2_xs = s.u2->xstride; }
    }
    static pc_once_per_device once;
    if (once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&level2_fwd_cl_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LC_LDS);
        if (e != hipSuccess) return (int)e;
        once.mark();
    }
    hipLaunchKernelGGL(level2_fwd_cl_kernel, dim3(B, n), dim3(512), LC_LDS, stream, a);
    PC_CHECK_LAUNCH();
    return 0;
}
