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

// ---- backward of down2's DoubleConv on the 32 x 32 level in ONE launch (PC_PREC_BF16) -----------------------------------------------
// Replaces conv3x3_bwd_cl_kernel<16, 16, false> (layer d2b) and <16, 16, true> (layer d2a + MaxPool2d(2) backward) at this level:
//     g1        = relu'(c1) * bn1 scale * conv^T(g2, w2)        (rounded to bf16: the mode's rounding point, as the stored tensor was)
//     out      += scatter of conv^T(g1, w1) to the first arg-max of every 2 x 2 window of `act`, times relu'(act) * act_bn scale
//     dW2, db2  = g2 (x) c1,   dW1, db1 = g1 (x) x               (one partial per workgroup and layer, pc_wgrad_reduce_batch finishes them)
// One 512-thread workgroup per (tile, network-stream) problem; wave w owns rows 4 w .. 4 w + 3 and runs exactly the per-strip
// arithmetic of conv3x3_bwd_cl_kernel on the whole-tile images (same MFMA order per accumulator: g1 and the scattered gradient are
// BIT-identical to the two launches; the weight-gradient partials are summed over other pixel groups, i.e. differ in the last bits).
// LDS: two tile images (g2 | c1, then x | g1: g1 stays in registers across the first reduction and takes c1's place) + both layers'
// data-gradient weight images; the cross-wave reduction of the weight-gradient blocks overlays the images between the two phases.
constexpr int LB_EC = 16 * 16 * 9 + 16;      // floats of one partial: dW[co][ci][tap] + db[co] (BwdCfg<16, 16>::EC of conv3x3_bwd.hip)
constexpr int LB_NBLK = 12;                  // weight-gradient accumulator blocks per 8 output channels: [dx][4 rows x 4 channels of x]
constexpr size_t LB_LDS = LC_LDS + (size_t)(8 * 2 * 16 + 16) * sizeof(float);      // + bias-sum scratch + the scatter's scale table
constexpr int LB_GP = 20;                    // floats per pixel of the scatter's fp32 image (16 channels + pad: 80-byte rows)
static_assert((size_t)(8 * LB_NBLK * 256 + LB_NBLK * 256) * sizeof(float) <= (size_t)2 * LC_IMG * 16 + (size_t)LC_WIMG * 2,
              "reduction scratch overlays the two images and layer 2's weight image");
static_assert((size_t)1024 * LB_GP * sizeof(float) <= (size_t)2 * LC_IMG * 16, "scatter image overlays the two images");

struct LbProb {
    const pc_bf16_t* g2; int64_t g2_bs; int g2_rs, g2_xs;        // dL/d(conv2 output), masked (B,16,32,32) channels-last bf16
    const pc_bf16_t* c1; int64_t c1_bs; int c1_rs, c1_xs;
    const pc_bf16_t* x; int64_t x_bs; int x_rs, x_xs;            // pooled input of the level
    const pc_bf16_t* act; int64_t a_bs; int a_rs, a_xs;          // full-resolution activation the pooled map was taken from (B,16,64,64)
    pc_bf16_t* out; int64_t o_bs; int o_rs, o_xs;                // its gradient map (+=)
    const float* w1; const float* w2;
    pc_bn bn1, act_bn;
    float* ws1; float* ws2;
};
struct LbArgs { LbProb pr[PC_MAX_GROUP]; long long* ts; };

__device__ __forceinline__ s16x4 lb_tr(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}
__device__ __forceinline__ bf16x8 lb_pair(s16x4 a, s16x4 b) { return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7)); }

__global__ __launch_bounds__(512) void level2_bwd_cl_kernel(const LbArgs args) {
    extern __shared__ __attribute__((aligned(16))) u32x4 lds4[];
    const LbProb& q = args.pr[blockIdx.y];
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    u32x4* const imgA = lds4;
    u32x4* const imgB = lds4 + LC_IMG;
    unsigned short* const wh2 = reinterpret_cast<unsigned short*>(lds4 + 2 * LC_IMG);
    unsigned short* const wh1 = wh2 + LC_WIMG;
    float* const bred = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(lds4) + LC_LDS);      // [8 waves][2][16] bias sums
    float* const scl = bred + 8 * 2 * 16;                                                               // act_bn scale per channel
    // debug (tools/level2_phases.py --bf16): wall-clock stamps of workgroup (0, 0), every phase closed by a barrier
    auto stamp = [&](int i) {
        if (args.ts) {
            __syncthreads();
            if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) args.ts[i] = wall_clock64();
        }
    };
    stamp(0);

    // ---- the three tiles: 1024 pixels x 2 chunks = 2048 16-byte pieces each, 4 per thread
    u32x4 gr[4], cr[4], xr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 512 * i, ch = idx & 1, px = idx >> 1, y = px >> 5, x = px & 31;
        gr[i] = *reinterpret_cast<const u32x4*>(q.g2 + b * q.g2_bs + (int64_t)y * q.g2_rs + (int64_t)x * q.g2_xs + 8 * ch);
        cr[i] = *reinterpret_cast<const u32x4*>(q.c1 + b * q.c1_bs + (int64_t)y * q.c1_rs + (int64_t)x * q.c1_xs + 8 * ch);
        xr[i] = *reinterpret_cast<const u32x4*>(q.x + b * q.x_bs + (int64_t)y * q.x_rs + (int64_t)x * q.x_xs + 8 * ch);
    }
    // ---- data-gradient weights of both layers: "output" channel = forward input channel ci, "input" channel = g channel co, taps
    //      flipped: weight(ci, co, tap) = w[co][ci][8 - tap]; element e = (ci, co, tap) of layer 2 for e < 2304, of layer 1 behind
    float wr[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int e0 = tid + 512 * k, e = e0 < 2304 ? e0 : e0 - 2304;
        const int tap = e % 9, co = (e / 9) % 16, ci = e / 144;
        wr[k] = (e0 < 2304 ? q.w2 : q.w1)[co * 144 + ci * 9 + (8 - tap)];
    }
    const int c4 = 4 * (lk & 1), e_s = lk >> 1;
    float sc1[2][4];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = nb * 8 + c4 + r;
            sc1[nb][r] = q.bn1.gamma ? q.bn1.gamma[c] * (1.0f / sqrtf(q.bn1.var[c] + q.bn1.eps)) : 1.f;
        }
    if (tid < 16) scl[tid] = q.act_bn.gamma ? q.act_bn.gamma[tid] * (1.0f / sqrtf(q.act_bn.var[tid] + q.act_bn.eps)) : 1.f;
    const int n_zero = 2 * LC_IMG + (2 * LC_WIMG * 2) / 16;
    for (int e = tid; e < n_zero; e += 512) lds4[e] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 512 * i, ch = idx & 1, px = idx >> 1;
        const int s = ch * LC_CH + ((px >> 5) + 1) * LC_SL + LC_COL0 + (px & 31);
        imgA[s] = gr[i];
        imgB[s] = cr[i];
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int e0 = tid + 512 * k, e = e0 < 2304 ? e0 : e0 - 2304;
        const int tap = e % 9, co = (e / 9) % 16, ci = e / 144;
        (e0 < 2304 ? wh2 : wh1)[(tap / 3) * LC_WDY + ci * LC_WCO + (co >> 3) * 24 + (tap % 3) * 8 + (co & 7)] = pc_f2bf(wr[k]);
    }
    __syncthreads();
    stamp(1);

    const int y0 = 4 * wave;
    const int a_s = li >> 3, a_co = li & 7;
    const int wplane = ((unsigned)(lk - a_s) <= 2u) ? lk - a_s : 3;
    // ---- data gradient of one layer over the wave's 32 x 4 strip: a 3x3 conv over the gradient image (K = 4 rows x 8 channels per MFMA)
    auto dgrad = [&](const u32x4* src, const unsigned short* wh, f32x4 (&acc)[4][2]) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[u][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned short* wl = wh + wplane * LC_WDY + a_co * LC_WCO;
#pragma unroll
        for (int gc = 0; gc < 2; ++gc) {
            const u32x4* lrow = src + gc * LC_CH + (y0 + lk) * LC_SL + (LC_COL0 - 1) + li;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                bf16x8 wq[2];
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    wq[nb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wl + nb * 8 * LC_WCO + gc * 24 + dx * 8));
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bf16x8 av = __builtin_bit_cast(bf16x8, lrow[(u >> 1) * 2 * LC_SL + (u & 1) * 16 + dx]);
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) acc[u][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[nb], av, acc[u][nb], 0, 0, 0);
                }
            }
        }
    };
    // ---- weight gradient of the strip: D_dx[(s, co)][(v, ci)] += sum_x g[co][y0 + 2 rpi + s][x] * x[ci][y0 + 2 rpi + v - 1][x + dx - 1]
    //      (operands through transposing reads, conv3x3_bwd_cl_kernel's addressing with the strip's first row = image row y0)
    const int t_j = li >> 2, t_q = li & 3;
    const int a_off = ((y0 + 1 + (t_q >> 1)) * LC_SL + (LC_COL0 - 1) + 1 + 8 * lk + t_j) * 16 + 8 * (t_q & 1);
    const int b_off = ((y0 + t_q) * LC_SL + (LC_COL0 - 1) + 8 * lk + t_j) * 16;
    const bf16x8 ones8 = __builtin_bit_cast(bf16x8, u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});
    auto wgrad = [&](const u32x4* gimg, const u32x4* ximg, f32x4 (&wacc)[2][LB_NBLK], f32x4 (&bacc)[2]) {
        const unsigned char* const gb = reinterpret_cast<const unsigned char*>(gimg);
        const unsigned char* const xb0 = reinterpret_cast<const unsigned char*>(ximg);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            bacc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < LB_NBLK; ++i) wacc[mb][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int rpi = 0; rpi < 2; ++rpi) {
            bf16x8 av[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const unsigned char* ga = gb + mb * LC_CH * 16 + 2 * rpi * LC_SL * 16 + a_off;
                av[mb] = lb_pair(lb_tr(ga), lb_tr(ga + 4 * 16));
                bacc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[mb], ones8, bacc[mb], 0, 0, 0);
            }
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const unsigned char* xb = xb0 + (nb >> 1) * LC_CH * 16 + 2 * rpi * LC_SL * 16 + b_off + dx * 16 + 8 * (nb & 1);
                    const bf16x8 bv = lb_pair(lb_tr(xb), lb_tr(xb + 4 * 16));
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
                        wacc[mb][dx * 4 + nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[mb], bv, wacc[mb][dx * 4 + nb], 0, 0, 0);
                }
        }
    };
    // ---- cross-wave reduction (fixed order) of a layer's blocks into the workgroup's partial; the scratch overlays the two images
    //      (and, behind them, layer 2's weight image, which phase 1's data gradient was the last to read).  Two stages per 8-channel
    //      half: the eight waves' blocks summed with whole-line reads in the D layout (768 items of 16 bytes), then the 1152 outputs of
    //      the half as the sum of their two D entries (rows s = 0 / 1 of the pair mapping)
    auto reduce = [&](float* part, const f32x4 (&wacc)[2][LB_NBLK], const f32x4 (&bacc)[2], auto&& after_first_write) {
        f32x4* const red4 = reinterpret_cast<f32x4*>(lds4);
        f32x4* const sum4 = red4 + 8 * LB_NBLK * 64;
        const float* const sumf = reinterpret_cast<const float*>(sum4);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            __syncthreads();
#pragma unroll
            for (int nb = 0; nb < LB_NBLK; ++nb) red4[(wave * LB_NBLK + nb) * 64 + lane] = wacc[mb][nb];
            if (mb == 0 && li == 0) {
#pragma unroll
                for (int m = 0; m < 2; ++m) *reinterpret_cast<f32x4*>(&bred[(wave * 2 + m) * 16 + 4 * lk]) = bacc[m];
            }
            if (mb == 0) after_first_write();
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int i = tid + 512 * j;
                if (i < LB_NBLK * 64) {
                    f32x4 sv = red4[i];
#pragma unroll
                    for (int w = 1; w < 8; ++w) sv += red4[w * LB_NBLK * 64 + i];
                    sum4[i] = sv;
                }
            }
            if (mb == 0 && tid < 16) {
                const int m = tid >> 3, c8 = tid & 7;
                const int ea = m * 16 + c8, eb = ea + 8;
                float sa = bred[ea], sb = bred[eb];
#pragma unroll
                for (int w = 1; w < 8; ++w) { sa += bred[w * 32 + ea]; sb += bred[w * 32 + eb]; }
                part[16 * 16 * 9 + tid] = sa + sb;
            }
            __syncthreads();
            for (int idx = tid; idx < 8 * 16 * 9; idx += 512) {
                const int c8 = idx / 144, rem = idx - c8 * 144;
                const int cil = rem / 9, tap = rem - cil * 9, dy = tap / 3, dx = tap - dy * 3;
                const int blk = dx * 4 + (cil >> 2);
                const int n0 = 4 * dy + (cil & 3), n1 = n0 + 4;
                const int m0 = c8, m1 = 8 + c8;
                const int e0 = (blk * 64 + (m0 >> 2) * 16 + n0) * 4 + (m0 & 3);
                const int e1 = (blk * 64 + (m1 >> 2) * 16 + n1) * 4 + (m1 & 3);
                part[(mb * 8 + c8) * 144 + rem] = sumf[e0] + sumf[e1];
            }
        }
    };

    f32x4 acc[4][2];
    f32x4 wacc[2][LB_NBLK], bacc[2];
    // ---- phase 1: layer d2b.  g1 of the wave's rows stays in registers (bf16 pairs) until c1's image is free
    uint2 g1pk[4][2];
    dgrad(imgA, wh2, acc);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int y = y0 + 2 * (u >> 1) + e_s, x = (u & 1) * 16 + li;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            f32x4 v = acc[u][nb];
            const f32x4 a4 = pc_ld4(reinterpret_cast<const pc_bf16_t*>(imgB + nb * LC_CH + (y + 1) * LC_SL + LC_COL0 + x) + c4);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = a4[r] > 0.f ? v[r] * sc1[nb][r] : 0.f;
            g1pk[u][nb] = make_uint2(pc_pack_bf16(v[0], v[1]), pc_pack_bf16(v[2], v[3]));
        }
    }
    stamp(2);
    wgrad(imgA, imgB, wacc, bacc);
    stamp(3);
    reduce(q.ws2 + (int64_t)b * LB_EC, wacc, bacc, [] {});
    stamp(4);
    __syncthreads();
    for (int e = tid; e < 2 * LC_IMG; e += 512) lds4[e] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 512 * i, ch = idx & 1, px = idx >> 1;
        imgA[ch * LC_CH + ((px >> 5) + 1) * LC_SL + LC_COL0 + (px & 31)] = xr[i];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int y = y0 + 2 * (u >> 1) + e_s, x = (u & 1) * 16 + li;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) reinterpret_cast<uint2*>(imgB + nb * LC_CH + (y + 1) * LC_SL + LC_COL0 + x)[lk & 1] = g1pk[u][nb];
    }
    __syncthreads();

    stamp(5);
    // ---- phase 2: layer d2a.  The data gradient of the pooled map waits in registers while the weight gradient and its reduction use
    //      the images; then it goes through LDS as an fp32 image [pixel][16 channels] so that the MaxPool2d(2) backward runs with one
    //      thread per (pooled pixel, 8-channel chunk): whole 16-byte slots of `act` and `out`, all loads of a thread in flight at once
    //      (conv3x3_bwd_cl_kernel<16, 16, true>'s epilogue holds 4 channels of a pixel per lane: 8-byte accesses, 64 bytes apart)
    dgrad(imgB, wh1, acc);
    stamp(6);
    wgrad(imgB, imgA, wacc, bacc);
    stamp(7);
    // the scatter's operands (16 loads of 16 bytes per thread from `act`, 16 from `out`: 48 MB over the launch) are requested as soon as
    // half of the weight-gradient blocks have left the registers: they arrive during the reduction
    const int a_rs = q.a_rs, a_xs = q.a_xs, o_rs = q.o_rs, o_xs = q.o_xs;
    u32x4 A[4][4], O[4][4];
    reduce(q.ws1 + (int64_t)b * LB_EC, wacc, bacc, [&] {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 512 * i, ch = idx & 1, px = idx >> 1, y = px >> 5, x = px & 31;
            const pc_bf16_t* a0 = q.act + b * q.a_bs + (int64_t)(2 * y) * a_rs + (int64_t)(2 * x) * a_xs + 8 * ch;
            const pc_bf16_t* o0 = q.out + b * q.o_bs + (int64_t)(2 * y) * o_rs + (int64_t)(2 * x) * o_xs + 8 * ch;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                A[i][w] = *reinterpret_cast<const u32x4*>(a0 + (w >> 1) * a_rs + (w & 1) * a_xs);
                O[i][w] = *reinterpret_cast<const u32x4*>(o0 + (w >> 1) * o_rs + (w & 1) * o_xs);
            }
        }
    });
    stamp(8);
    __syncthreads();
    float* const gp = reinterpret_cast<float*>(lds4);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int y = y0 + 2 * (u >> 1) + e_s, x = (u & 1) * 16 + li;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) *reinterpret_cast<f32x4*>(&gp[(y * 32 + x) * LB_GP + nb * 8 + c4]) = acc[u][nb];
    }
    __syncthreads();
    {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 512 * i, ch = idx & 1, px = idx >> 1, y = px >> 5, x = px & 31;
            pc_bf16_t* o0 = q.out + b * q.o_bs + (int64_t)(2 * y) * o_rs + (int64_t)(2 * x) * o_xs + 8 * ch;
            float v[8], sc[8];
            *reinterpret_cast<f32x4*>(&v[0]) = *reinterpret_cast<const f32x4*>(&gp[px * LB_GP + ch * 8]);
            *reinterpret_cast<f32x4*>(&v[4]) = *reinterpret_cast<const f32x4*>(&gp[px * LB_GP + ch * 8 + 4]);
            *reinterpret_cast<f32x4*>(&sc[0]) = *reinterpret_cast<const f32x4*>(&scl[ch * 8]);
            *reinterpret_cast<f32x4*>(&sc[4]) = *reinterpret_cast<const f32x4*>(&scl[ch * 8 + 4]);
            auto bf = [](const u32x4& t, int j) { return (j & 1) ? __uint_as_float(t[j >> 1] & 0xffff0000u) : __uint_as_float(t[j >> 1] << 16); };
            float R[4][8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int am = 0;
                float m = bf(A[i][0], j);
                const float a1 = bf(A[i][1], j), a2 = bf(A[i][2], j), a3 = bf(A[i][3], j);
                if (a1 > m) { m = a1; am = 1; }
                if (a2 > m) { m = a2; am = 2; }
                if (a3 > m) { m = a3; am = 3; }
                const float gv = m > 0.f ? v[j] * sc[j] : 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) R[w][j] = bf(O[i][w], j) + (am == w ? gv : 0.f);
            }
#pragma unroll
            for (int w = 0; w < 4; ++w)
                *reinterpret_cast<u32x4*>(o0 + (w >> 1) * o_rs + (w & 1) * o_xs) =
                    u32x4{pc_pack_bf16(R[w][0], R[w][1]), pc_pack_bf16(R[w][2], R[w][3]), pc_pack_bf16(R[w][4], R[w][5]), pc_pack_bf16(R[w][6], R[w][7])};
        }
    }
    stamp(9);
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

bool pc_level2_bwd_cl_ok(const pc_src* g2, const pc_src* c1, const pc_src* x, const pc_src* act, const pc_dst* out) {
    auto src16 = [](const pc_src* s, int Hh, int Ww) {
        return s && s->C == 16 && s->H == Hh && s->W == Ww && s->mode == PC_SRC_DIRECT && !s->oy && !s->ox &&
               cl16(s->ptr, s->dtype, s->bstride, s->cstride, s->rstride, s->xstride);
    };
    return out && src16(g2, 32, 32) && src16(c1, 32, 32) && src16(x, 32, 32) && src16(act, 64, 64) &&
           cl16(out->ptr, out->dtype, out->bstride, out->cstride, out->rstride, out->xstride);
}

int pc_level2_bwd_cl_launch(int n, const pc_level2_bwd_desc* d, int B, int* nwg_out, long long* ts, hipStream_t stream) {
    LbArgs a;
    a.ts = ts;
    for (int i = 0; i < n; ++i) {
        const pc_level2_bwd_desc& s = d[i];
        if (!s.w1 || !s.w2 || !s.bn1 || !s.act_bn || !s.ws1 || !s.ws2 || !pc_level2_bwd_cl_ok(s.g2, s.c1, s.x, s.act, s.out)) return PC_EINVAL;
        LbProb& p = a.pr[i];
        p = LbProb{};
        p.g2 = reinterpret_cast<const pc_bf16_t*>(s.g2->ptr); p.g2_bs = s.g2->bstride; p.g2_rs = s.g2->rstride; p.g2_xs = s.g2->xstride;
        p.c1 = reinterpret_cast<const pc_bf16_t*>(s.c1->ptr); p.c1_bs = s.c1->bstride; p.c1_rs = s.c1->rstride; p.c1_xs = s.c1->xstride;
        p.x = reinterpret_cast<const pc_bf16_t*>(s.x->ptr); p.x_bs = s.x->bstride; p.x_rs = s.x->rstride; p.x_xs = s.x->xstride;
        p.act = reinterpret_cast<const pc_bf16_t*>(s.act->ptr); p.a_bs = s.act->bstride; p.a_rs = s.act->rstride; p.a_xs = s.act->xstride;
        p.out = reinterpret_cast<pc_bf16_t*>(s.out->ptr); p.o_bs = s.out->bstride; p.o_rs = s.out->rstride; p.o_xs = s.out->xstride;
        p.w1 = s.w1; p.w2 = s.w2; p.bn1 = *s.bn1; p.act_bn = *s.act_bn;
        p.ws1 = reinterpret_cast<float*>(s.ws1); p.ws2 = reinterpret_cast<float*>(s.ws2);
    }
    static pc_once_per_device once;
    if (once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&level2_bwd_cl_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LB_LDS);
        if (e != hipSuccess) return (int)e;
        once.mark();
    }
    hipLaunchKernelGGL(level2_bwd_cl_kernel, dim3(B, n), dim3(512), LB_LDS, stream, a);
    PC_CHECK_LAUNCH();
    *nwg_out = B;                        // one partial per tile and layer (each ws holds pc_level2_bwd_ws_bytes(B) >= B * LB_EC floats)
    return 0;
}
