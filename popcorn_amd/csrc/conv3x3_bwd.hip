// conv3x3_bwd.hip -- backward of one 3x3 convolution in ONE launch (PC_PREC_BF16, channels-last bf16 tensors, 8 / 16 channels):
// the data gradient (conv3x3_cl_kernel<dgrad>) and the weight / bias gradient (conv3x3_wgrad_cl_kernel) of a layer read the SAME
// two tensors -- the layer's output gradient g (conv input of the one, pixel-contraction operand of the other) and the layer's
// input x (ReLU mask of the one, second operand of the other).  As two launches they are 5 tensor reads + 1 write; here each
// strip of g and x goes through LDS once: 2 reads + 1 write, and one launch less per layer.
//     gx[ci][y][x]     = relu'(x[ci]) * bn_scale[ci] * sum_{co,tap} w[co][c0+ci][tap] * g[co][y - dy + 1][x - dx + 1]     (+= optional)
//     dW[co][c0+ci][tap] = sum_px g[co][px] * x[ci][px + tap],   db[co] = sum_px g[co][px]
// x may be one 8-channel half of a concatenated input (torch.cat([skip, up]), networks.py:318): c0 selects the weight columns,
// the placement offset of the up-sampled half is the source's (oy, ox), and the ReLU mask applies to the skip half only.
// Both images are [6 rows][48 slots] of 16 bytes (one slot = 8 channels of a pixel), exactly conv3x3_cl_kernel's strip image;
// the weight-gradient operands come out of them through ds_read_b64_tr_b16 (see conv3x3_wgrad.hip).  Weight-gradient partials
// use the compacted per-workgroup layout of the other kernels (pc_wgrad_reduce_batch finishes them).
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int TW = 32, TH = 16;
constexpr int SROWS = 6, BSLOTS = 48, COL0 = 4, PX = 34, PIECES = SROWS * PX;
constexpr int IMG_B = SROWS * BSLOTS * 16;              // bytes of one strip image
constexpr int MAXG = PC_MAX_GROUP;

struct BwdProb {
    pc_src g, x;            // output gradient (conv domain) and the layer's input (placement offset in oy / ox)
    const float* w;         // forward weights [Cout = GC][Cin_total][3][3], already offset to input channel c0
    int w_ci_stride;        // Cin_total * 9
    int mask;               // 1: gx *= (x > 0) * bn_scale   (x is the post-ReLU output of a conv + BN layer)
    pc_bn bn;               // BN of the layer that produced x (mask = 1)
    pc_dst out;             // gx (POOL: the gradient map at twice the resolution, accumulated into)
    pc_src pool_act;        // POOL: the full-resolution activation the pooled x was taken from (arg-max + ReLU mask)
    float* partial;         // [nwg][GC * XC * 9 + GC]
};

struct BwdArgs {
    BwdProb pr[MAXG];
    int accumulate;         // gx += instead of =
    int dbg;                // ablation switches of conv3x3_bwd_s3_kernel (pc_debug_conv_bwd): 1 no split / LDS writes, 2 no data-gradient matrix
                            // phase, 4 no weight-gradient matrix phase, 8 no prefetch loads, 16 no epilogue
    int B, H, W;
    int tiles_x, tiles_y, ntiles;
    pc_fastdiv div_tx, div_tpi;
};

__device__ __forceinline__ s16x4 bw_tr(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}
__device__ __forceinline__ bf16x8 bw_pair(s16x4 a, s16x4 b) { return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7)); }

// GC = channels of g (the layer's output channels), XC = channels of x (this block of the layer's input channels): 8 or 16
template <int GC, int XC>
struct BwdCfg {
    static constexpr int NG = GC / 8, NX = XC / 8;           // 8-channel images of the two strips
    static constexpr int NBP = XC / 4;                        // N blocks (4 rows x 4 channels of x) per tap
    static constexpr int NBLK = 3 * NBP;                      // weight-gradient accumulator blocks per 8 output channels: [dx][nbp]
    static constexpr int EC = GC * XC * 9 + GC;               // floats of one workgroup partial: dW[co][ci][tap] + db[co]
    static constexpr size_t WAVE_B = (size_t)(NG + NX) * IMG_B;
    static constexpr size_t W_B = (size_t)4 * XC * NG * 24 * 2;          // weight image [dy plane][ci][g chunk][dx][8 co] bf16
    static constexpr size_t RED_B = (size_t)4 * NBLK * 256 * sizeof(float);
    static constexpr size_t LDS_B = 4 * WAVE_B + W_B > RED_B ? 4 * WAVE_B + W_B : RED_B;
};

// POOL: x is the 2x2-max-pooled copy of `pool_act` (the Down block's input, networks.py:289); the data gradient is scattered to the
// first arg-max of every window of the full-resolution gradient map (+=), with the ReLU / BN factor of pool_act's producer
template <int GC, int XC, bool POOL>
__global__ __launch_bounds__(256) void conv3x3_bwd_cl_kernel(const BwdArgs p) {
    using Cfg = BwdCfg<GC, XC>;
    constexpr int NG = Cfg::NG, NX = Cfg::NX, NBP = Cfg::NBP, NBLK = Cfg::NBLK, EC = Cfg::EC;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    // the problem's descriptor and the launch geometry, pinned in scalar registers (common.h: pc_pin -- read from the kernel arguments
    // inside the strip loop they were ~30 scalar-memory round trips per strip, round 6)
    BwdProb q = p.pr[blockIdx.y];
    pc_pin(q.g); pc_pin(q.x); pc_pin(q.out); pc_pin(q.mask);
    if constexpr (POOL) pc_pin(q.pool_act);
    int pH = p.H, pW = p.W, pacc = p.accumulate, ntl = p.ntiles, tlx = p.tiles_x, tly = p.tiles_y, gdim = (int)gridDim.x;
    pc_pin(pH); pc_pin(pW); pc_pin(pacc); pc_pin(ntl); pc_pin(tlx); pc_pin(tly); pc_pin(gdim);
    pc_fastdiv dtx = p.div_tx, dtpi = p.div_tpi;
    pc_pin(dtx); pc_pin(dtpi);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    unsigned char* const wimg = ldsb + wave * Cfg::WAVE_B;
    u32x4* const ig = reinterpret_cast<u32x4*>(wimg);                       // g strip: NG images
    u32x4* const ix = reinterpret_cast<u32x4*>(wimg + NG * IMG_B);          // x strip: NX images
    unsigned short* const w2h = reinterpret_cast<unsigned short*>(ldsb + 4 * Cfg::WAVE_B);

    // ---- loaders: piece id = lane + 64 * i -> (strip row, pixel of the 34-pixel row), for both tensors
    int l_slot[4], l_r[4], l_px[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = lane + 64 * i;
        l_r[i] = id / PX;
        l_px[i] = id - l_r[i] * PX;
        l_slot[i] = id < PIECES ? l_r[i] * BSLOTS + (COL0 - 1) + l_px[i] : -1;
    }
    u32x4 RG[NG][4], RX[NX][4];
    unsigned gvalid = 0, xvalid = 0;
    const pc_bf16_t* const gptr = reinterpret_cast<const pc_bf16_t*>(q.g.ptr);
    const pc_bf16_t* const xptr = reinterpret_cast<const pc_bf16_t*>(q.x.ptr);
    auto issue = [&](int b, int y0, int x0) {
        unsigned gm = 0, xm = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int y = y0 - 1 + l_r[i], x = x0 - 1 + l_px[i];
            const bool in = l_slot[i] >= 0 && (unsigned)y < (unsigned)pH && (unsigned)x < (unsigned)pW;
            const pc_bf16_t* gp = gptr + b * q.g.bstride + (in ? (int64_t)y * q.g.rstride + (int64_t)x * q.g.xstride : 0);
#pragma unroll
            for (int c = 0; c < NG; ++c) RG[c][i] = *reinterpret_cast<const u32x4*>(gp + 8 * c);
            const int ys = y - q.x.oy, xs = x - q.x.ox;
            const bool inx = in && (unsigned)ys < (unsigned)q.x.H && (unsigned)xs < (unsigned)q.x.W;
            const pc_bf16_t* xp = xptr + b * q.x.bstride + (inx ? (int64_t)ys * q.x.rstride + (int64_t)xs * q.x.xstride : 0);
#pragma unroll
            for (int c = 0; c < NX; ++c) RX[c][i] = *reinterpret_cast<const u32x4*>(xp + 8 * c);
            gm |= (in ? 1u : 0u) << i;
            xm |= (inx ? 1u : 0u) << i;
        }
        gvalid = gm; xvalid = xm;
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (l_slot[i] >= 0) {
#pragma unroll
                for (int c = 0; c < NG; ++c) ig[c * SROWS * BSLOTS + l_slot[i]] = ((gvalid >> i) & 1u) ? RG[c][i] : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
                for (int c = 0; c < NX; ++c) ix[c * SROWS * BSLOTS + l_slot[i]] = ((xvalid >> i) & 1u) ? RX[c][i] : u32x4{0u, 0u, 0u, 0u};
            }
    };
    const int my_tiles = ntl > (int)blockIdx.x ? (ntl - 1 - (int)blockIdx.x) / gdim + 1 : 0;
    auto strip_coords = [&](int k, int& b, int& y0, int& x0) {
        const int tile = pc_xcd_remap(blockIdx.x + k * gdim, ntl);
        b = (int)pc_div((uint32_t)tile, dtpi);
        const int rem = tile - b * tlx * tly;
        const int ty = (int)pc_div((uint32_t)rem, dtx);
        x0 = (rem - ty * tlx) * TW;
        y0 = ty * TH + 4 * wave;
    };
    int b = 0, y0 = 0, x0 = 0;
    if (my_tiles > 0) {
        strip_coords(0, b, y0, x0);
        issue(b, y0, x0);
    }

    // ---- data-gradient weights: "output" channel = forward input channel ci, "input" channel = g channel co, taps flipped:
    //      weight(ci, co, tap) = w[co][ci][8 - tap].  Image [dy plane 0..3][ci][g chunk][dx][8 co] bf16, plane 3 all zero.
    constexpr int BW_CO = NG * 24, BW_DYS = XC * BW_CO;
    constexpr int NWR = (GC * XC * 9 + 255) / 256;
    for (int e = tid; e < 4 * BW_DYS / 2; e += 256) reinterpret_cast<unsigned*>(w2h)[e] = 0u;
    float wreg[NWR];
#pragma unroll
    for (int k = 0; k < NWR; ++k) {
        const int e = tid + k * 256;
        const int ec = e < GC * XC * 9 ? e : 0;
        const int tap = ec % 9, co = (ec / 9) % GC, ci = ec / (9 * GC);
        wreg[k] = q.w[ci * 9 + co * q.w_ci_stride + (8 - tap)];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NWR; ++k) {
        const int e = tid + k * 256;
        if (e < GC * XC * 9) {
            const int tap = e % 9, co = (e / 9) % GC, ci = e / (9 * GC);
            w2h[(tap / 3) * BW_DYS + ci * BW_CO + (co / 8) * 24 + (tap % 3) * 8 + (co % 8)] = pc_f2bf(wreg[k]);
        }
    }
    const int c4 = 4 * (lk & 1), e_s = lk >> 1;
    float e_scale[NX][4];
#pragma unroll
    for (int nb = 0; nb < NX; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = nb * 8 + c4 + r;
            e_scale[nb][r] = (q.mask && q.bn.gamma) ? q.bn.gamma[c] * (1.0f / sqrtf(q.bn.var[c] + q.bn.eps)) : 1.f;
        }
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < NX; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("" : : "v"(e_scale[nb][r]));
    const int a_s = li >> 3, a_co = li & 7;
    const unsigned short* const wlane = w2h + (((unsigned)(lk - a_s) <= 2u) ? lk - a_s : 3) * BW_DYS + a_co * BW_CO;

    // ---- weight-gradient reads: lane supplies pixel row j = li >> 2 and column quad tq = li & 3 of a [4 pixels][16 columns] block
    const int t_j = li >> 2, t_q = li & 3;
    // A (M = (s, co8)): quad = (s = tq >> 1, channel half tq & 1) of g rows y0 + 2*rpi + s = image row 1 + 2*rpi + s, pixel x0 + 8*lk + 4*e + j
    const int a_off = ((1 + (t_q >> 1)) * BSLOTS + (COL0 - 1) + 1 + 8 * lk + t_j) * 16 + 8 * (t_q & 1);
    // B (N = (v, ci4)): quad = input row v = tq (image row 2*rpi + v), pixel index 8*lk + 4*e + j + dx, channel half nb
    const int b_off = (t_q * BSLOTS + (COL0 - 1) + 8 * lk + t_j) * 16;
    f32x4 wacc[NG][NBLK];
    // bias gradient = row sums of the A operand over the pixels: one more MFMA against an all-ones operand (every column of the
    // block then holds the row sums) instead of ~14 VALU per transposed read pair
    f32x4 bacc[NG];
    const bf16x8 ones8 = __builtin_bit_cast(bf16x8, u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});
#pragma unroll
    for (int mb = 0; mb < NG; ++mb) {
        bacc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NBLK; ++i) wacc[mb][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    pc_bf16_t* const outp = reinterpret_cast<pc_bf16_t*>(q.out.ptr);
    const unsigned char* const igb = reinterpret_cast<const unsigned char*>(ig);
    const unsigned char* const ixb = reinterpret_cast<const unsigned char*>(ix);

    for (int k = 0; k < my_tiles; ++k) {
        commit();
        int nb_ = b, ny0 = y0, nx0 = x0;
        if (k + 1 < my_tiles) {
            strip_coords(k + 1, nb_, ny0, nx0);
            issue(nb_, ny0, nx0);
        }
        // ---- data gradient: a 3x3 conv over g (K = 4 rows x 8 channels per MFMA)
        f32x4 acc[4][NX];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int nb = 0; nb < NX; ++nb) acc[u][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int gc = 0; gc < NG; ++gc) {
            const u32x4* lrow = ig + gc * SROWS * BSLOTS + lk * BSLOTS + (COL0 - 1) + li;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                bf16x8 wq[NX];
#pragma unroll
                for (int nb = 0; nb < NX; ++nb)
                    wq[nb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wlane + nb * 8 * BW_CO + gc * 24 + dx * 8));
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bf16x8 av = __builtin_bit_cast(bf16x8, lrow[(u >> 1) * 2 * BSLOTS + (u & 1) * 16 + dx]);
#pragma unroll
                    for (int nb = 0; nb < NX; ++nb) acc[u][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[nb], av, acc[u][nb], 0, 0, 0);
                }
            }
        }
        // lane holds pixel (y0 + 2*(u>>1) + e_s, x0 + (u&1)*16 + li), channels c4 + r of gx
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int y = y0 + 2 * (u >> 1) + e_s, x = x0 + (u & 1) * 16 + li;
            if (y < pH && x < pW) {
                if constexpr (POOL) {
                    const pc_bf16_t* const actp = reinterpret_cast<const pc_bf16_t*>(q.pool_act.ptr);
                    const int a_rs = q.pool_act.rstride, a_xs = q.pool_act.xstride, o_rs = q.out.rstride, o_xs = q.out.xstride;
#pragma unroll
                    for (int nb = 0; nb < NX; ++nb) {
                        const f32x4 v = acc[u][nb];
                        const pc_bf16_t* a0 = actp + b * q.pool_act.bstride + (int64_t)(2 * y) * a_rs + (int64_t)(2 * x) * a_xs + nb * 8 + c4;
                        pc_bf16_t* o0 = outp + b * q.out.bstride + (int64_t)(2 * y) * o_rs + (int64_t)(2 * x) * o_xs + nb * 8 + c4;
                        const f32x4 A00 = pc_ld4(a0), A01 = pc_ld4(a0 + a_xs), A10 = pc_ld4(a0 + a_rs), A11 = pc_ld4(a0 + a_rs + a_xs);
                        f32x4 O00 = pc_ld4(o0), O01 = pc_ld4(o0 + o_xs), O10 = pc_ld4(o0 + o_rs), O11 = pc_ld4(o0 + o_rs + o_xs);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            int am = 0;
                            float m = A00[r];
                            if (A01[r] > m) { m = A01[r]; am = 1; }
                            if (A10[r] > m) { m = A10[r]; am = 2; }
                            if (A11[r] > m) { m = A11[r]; am = 3; }
                            const float gv = m > 0.f ? v[r] * e_scale[nb][r] : 0.f;
                            O00[r] += am == 0 ? gv : 0.f;
                            O01[r] += am == 1 ? gv : 0.f;
                            O10[r] += am == 2 ? gv : 0.f;
                            O11[r] += am == 3 ? gv : 0.f;
                        }
                        pc_st4(o0, O00); pc_st4(o0 + o_xs, O01); pc_st4(o0 + o_rs, O10); pc_st4(o0 + o_rs + o_xs, O11);
                    }
                } else
#pragma unroll
                for (int nb = 0; nb < NX; ++nb) {
                    f32x4 v = acc[u][nb];
                    if (q.mask) {
                        // the layer's input at this pixel sits in the x image (row 1 + ..., pixel index 1 + ...)
                        const f32x4 a4 = pc_ld4(reinterpret_cast<const pc_bf16_t*>(ixb + nb * IMG_B + ((1 + 2 * (u >> 1) + e_s) * BSLOTS + (COL0 - 1) + 1 + (u & 1) * 16 + li) * 16) + c4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = a4[r] > 0.f ? v[r] * e_scale[nb][r] : 0.f;
                    }
                    pc_bf16_t* op = outp + b * q.out.bstride + (int64_t)y * q.out.rstride + (int64_t)x * q.out.xstride + nb * 8 + c4;
                    if (pacc) {
                        const f32x4 o4 = pc_ld4(op);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += o4[r];
                    }
                    pc_st4(op, v);
                }
            }
        }
        // ---- weight gradient of the strip: D_dx[(s,co)][(v,ci)] += sum_x g[co][y0+2rpi+s][x] * x[ci][y0+2rpi+v-1][x+dx-1]
#pragma unroll
        for (int rpi = 0; rpi < 2; ++rpi) {
            bf16x8 av[NG];
#pragma unroll
            for (int mb = 0; mb < NG; ++mb) {
                const unsigned char* ga = igb + mb * IMG_B + 2 * rpi * BSLOTS * 16 + a_off;
                const s16x4 lo = bw_tr(ga), hi = bw_tr(ga + 4 * 16);
                av[mb] = bw_pair(lo, hi);
                bacc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[mb], ones8, bacc[mb], 0, 0, 0);
            }
#pragma unroll
            for (int nb = 0; nb < NBP; ++nb)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const unsigned char* xb = ixb + (nb >> 1) * IMG_B + 2 * rpi * BSLOTS * 16 + b_off + dx * 16 + 8 * (nb & 1);
                    const bf16x8 bv = bw_pair(bw_tr(xb), bw_tr(xb + 4 * 16));
#pragma unroll
                    for (int mb = 0; mb < NG; ++mb)
                        wacc[mb][dx * NBP + nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[mb], bv, wacc[mb][dx * NBP + nb], 0, 0, 0);
                }
        }
        b = nb_; y0 = ny0; x0 = nx0;
    }

    // ---- cross-wave reduction through LDS (fixed order), one compacted partial per workgroup (layout of conv3x3_wgrad_cl_kernel)
    float* lds = reinterpret_cast<float*>(ldsb);
    float* part = q.partial + (int64_t)blockIdx.x * EC;
    auto wsum = [&](int e) { return ((lds[e] + lds[NBLK * 256 + e]) + lds[2 * NBLK * 256 + e]) + lds[3 * NBLK * 256 + e]; };
#pragma unroll
    for (int mb = 0; mb < NG; ++mb) {
        __syncthreads();
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb) *reinterpret_cast<f32x4*>(&lds[((wave * NBLK + nb) * 64 + lane) * 4]) = wacc[mb][nb];
        __syncthreads();
        for (int idx = tid; idx < 8 * XC * 9; idx += 256) {
            const int c8 = idx / (XC * 9), rem = idx - c8 * (XC * 9);
            const int cil = rem / 9, tap = rem - cil * 9, dy = tap / 3, dx = tap - dy * 3;
            // D_dx[m = s*8 + c8][n = 4*v + (cil & 3)] in block dx*NBP + (cil >> 2); D register layout: lane = (m>>2)*16 + n, reg = m&3
            const int blk = dx * NBP + (cil >> 2);
            const int n0 = 4 * dy + (cil & 3), n1 = n0 + 4;
            const int m0 = c8, m1 = 8 + c8;
            const int e0 = (blk * 64 + (m0 >> 2) * 16 + n0) * 4 + (m0 & 3);
            const int e1 = (blk * 64 + (m1 >> 2) * 16 + n1) * 4 + (m1 & 3);
            part[(mb * 8 + c8) * (XC * 9) + rem] = wsum(e0) + wsum(e1);
        }
    }
    __syncthreads();
    // column 0 of the ones-block = lanes li == 0: row m = 4 * lk + register, m = s * 8 + c8
    if (li == 0) {
#pragma unroll
        for (int mb = 0; mb < NG; ++mb) *reinterpret_cast<f32x4*>(&lds[(wave * NG + mb) * 16 + 4 * lk]) = bacc[mb];
    }
    __syncthreads();
    if (tid < GC) {
        const int mb = tid >> 3, c8 = tid & 7;
        const int ea = mb * 16 + c8, eb = ea + 8;
        const float sa = ((lds[ea] + lds[NG * 16 + ea]) + lds[2 * NG * 16 + ea]) + lds[3 * NG * 16 + ea];
        const float sb = ((lds[eb] + lds[NG * 16 + eb]) + lds[2 * NG * 16 + eb]) + lds[3 * NG * 16 + eb];
        part[GC * XC * 9 + tid] = sa + sb;
    }
}

template <int GC, int XC, bool POOL>
int launch_bwd(BwdArgs& p, int n, int* nwg_out, hipStream_t stream) {
    using Cfg = BwdCfg<GC, XC>;
    static int resident = 0;
    static pc_once_per_device once;
    if (once.need()) {
        const void* fn = reinterpret_cast<const void*>(&conv3x3_bwd_cl_kernel<GC, XC, POOL>);
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_B);
        if (e != hipSuccess) return (int)e;
        hipFuncAttributes fa;
        e = hipFuncGetAttributes(&fa, fn);
        if (e != hipSuccess) return (int)e;
        resident = pc_resident_workgroups(fa.numRegs, Cfg::LDS_B);
        once.mark();
        if (getenv("POPCORN_CONV_DBG"))
            fprintf(stderr, "conv3x3_bwd<%d,%d,%d>: %d regs, %zu B LDS -> %d resident workgroups\n", GC, XC, (int)POOL, fa.numRegs, (size_t)Cfg::LDS_B, resident);
    }
    int nwg = resident / n;
    if (nwg > 512) nwg = 512;                // partials per problem (the workspace slice holds more)
    if (nwg > p.ntiles) nwg = p.ntiles;
    if (nwg < 1) nwg = 1;
    const int rounds = (p.ntiles + nwg - 1) / nwg;
    nwg = (p.ntiles + rounds - 1) / rounds;
    hipLaunchKernelGGL((conv3x3_bwd_cl_kernel<GC, XC, POOL>), dim3(nwg, n), dim3(256), Cfg::LDS_B, stream, p);
    PC_CHECK_LAUNCH();
    *nwg_out = nwg;
    return 0;
}


// ---- fp32 (planar tensors, PC_PREC_FP32): the same fusion on v_mfma_f32_16x16x4_f32, 8 -> 8 channels -------------------------
// The data gradient is conv3x3_mfma_kernel<8, 8, dgrad>'s mapping (K = 4 strip rows, N = (row of the output pair, 8 channels)), the
// weight gradient conv3x3_wgrad_wave_kernel<8, 8>'s (K = 4 pixels, M = (row, g channel), N = (x channel, strip row, dx)); each wave
// stages its 6 x 40 strip of g (row stride 48, channel stride 292: the pixel-contraction operand is read across channels) and of x
// (row stride 44, channel stride 264: the conflict-free layout of the weight-gradient kernel) once per strip.  The ReLU mask comes from
// the x image.  Aligned sources only (W % 4 == 0, no placement offset): the callers keep the separate launches for anything else.
constexpr int F_RS = 48, F_CSW = 6 * F_RS + 4;
constexpr int F_XRS = 44, F_XCSW = 6 * F_XRS;
constexpr int F_WAVE = 8 * (F_CSW + F_XCSW);               // floats of a wave's LDS region
constexpr int F_WRL = 8 * 3 + 4, F_WDYS = 8 * F_WRL + 16;  // weight image [dy plane][x channel][g channel * 3 + dx] (conv3x3.hip)
constexpr int F_NBLK = 6, F_EC = 8 * 8 * 9 + 8;
constexpr size_t F_LDS_B = (size_t)(4 * F_WAVE + 4 * F_WDYS) * sizeof(float);

__global__ __launch_bounds__(256) void conv3x3_bwd_f32_kernel(const BwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) float ldsf[];
    const BwdProb& q = p.pr[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int s_row = li >> 3, col = li & 7;
    float* const wg = ldsf + wave * F_WAVE;                // g image
    float* const wx = wg + 8 * F_CSW;                      // x image
    float* const w2 = ldsf + 4 * F_WAVE;

    // ---- loader: lane = (row of the 6-row strip, 16-byte segment of the 40-float row), both tensors
    const int l_r = lane / 10, l_seg = lane - l_r * 10;
    const bool l_act = lane < 60;
    f32x4 RG[8], RX[8];
    bool rvalid = false;
    auto issue = [&](int b, int y0, int x0) {
        const int xg = x0 - 4 + 4 * l_seg, y = y0 - 1 + l_r;
        const bool ok = l_act && xg >= 0 && xg < p.W && (unsigned)y < (unsigned)p.H;
        rvalid = ok;
        const float* gp = q.g.ptr + (ok ? b * q.g.bstride + (int64_t)y * q.g.rstride + xg : 0);
        const float* xp = q.x.ptr + (ok ? b * q.x.bstride + (int64_t)y * q.x.rstride + xg : 0);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            RG[it] = *reinterpret_cast<const f32x4*>(gp + it * q.g.cstride);
            RX[it] = *reinterpret_cast<const f32x4*>(xp + it * q.x.cstride);
        }
    };
    auto commit = [&]() {
        if (l_act) {
            float* dg = wg + l_r * F_RS + 4 * l_seg;
            float* dx_ = wx + l_r * F_XRS + 4 * l_seg;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                *reinterpret_cast<f32x4*>(dg + it * F_CSW) = rvalid ? RG[it] : f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(dx_ + it * F_XCSW) = rvalid ? RX[it] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    const int my_tiles = p.ntiles > (int)blockIdx.x ? (p.ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    auto strip_coords = [&](int k, int& b, int& y0, int& x0) {
        const int tile = pc_xcd_remap(blockIdx.x + k * gridDim.x, p.ntiles);
        b = (int)pc_div((uint32_t)tile, p.div_tpi);
        const int rem = tile - b * p.tiles_x * p.tiles_y;
        const int ty = (int)pc_div((uint32_t)rem, p.div_tx);
        x0 = (rem - ty * p.tiles_x) * TW;
        y0 = ty * TH + 4 * wave;
    };
    int b = 0, y0 = 0, x0 = 0;
    if (my_tiles > 0) {
        strip_coords(0, b, y0, x0);
        issue(b, y0, x0);
    }

    // ---- data-gradient weights: output channel = x channel, K channel = g channel, taps flipped (one round trip, see conv3x3.hip)
    float wreg[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int e = tid + k * 256;
        const int ec = e < 576 ? e : 0;
        const int tap = ec % 9, gch = (ec / 9) % 8, xch = ec / 72;
        wreg[k] = q.w[xch * 9 + gch * q.w_ci_stride + (8 - tap)];
    }
    float bn_g = 1.f, bn_v = 1.f;
    if (q.mask && q.bn.gamma) { bn_g = q.bn.gamma[col]; bn_v = q.bn.var[col]; }
    for (int e = tid; e < F_WDYS; e += 256) w2[3 * F_WDYS + e] = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int e = tid + k * 256;
        if (e < 576) {
            const int tap = e % 9, gch = (e / 9) % 8, xch = e / 72;
            w2[(tap / 3) * F_WDYS + xch * F_WRL + gch * 3 + (tap % 3)] = wreg[k];
        }
    }
    __syncthreads();
    const float* const wlane = w2 + (((unsigned)(lk - s_row) <= 2u) ? lk - s_row : 3) * F_WDYS + col * F_WRL;
    const float e_scale = (q.mask && q.bn.gamma) ? bn_g * (1.0f / sqrtf(bn_v + q.bn.eps)) : 1.f;
    asm volatile("" : : "v"(e_scale));

    // ---- weight-gradient operand addresses
    const int koff = ((lk & 1) << 4) | ((lk >> 1) << 3);          // {0,16,8,24}: a lane's 8 k-steps are 8 consecutive floats
    int boff[F_NBLK];
#pragma unroll
    for (int nb = 0; nb < F_NBLK; ++nb) {
        const int ng = nb * 16 + li;
        const int ci = ng / 12, rem = ng % 12, v = rem / 3, dx = rem % 3;
        boff[nb] = ci * F_XCSW + v * F_XRS + (COL0 - 1) + dx + koff;
    }
    const float* const ga = wg + col * F_CSW + (1 + s_row) * F_RS + COL0 + koff;      // A: lane (i = (s, g channel), k)
    f32x4 wacc[F_NBLK];
    float bsum = 0.f;
#pragma unroll
    for (int nb = 0; nb < F_NBLK; ++nb) wacc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int k = 0; k < my_tiles; ++k) {
        commit();
        int nb_ = b, ny0 = y0, nx0 = x0;
        if (k + 1 < my_tiles) {
            strip_coords(k + 1, nb_, ny0, nx0);
            issue(nb_, ny0, nx0);
        }
        // ---- data gradient
        {
            float bw[8][3];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(wlane + 4 * j);
#pragma unroll
                for (int e = 0; e < 4; ++e) bw[(4 * j + e) / 3][(4 * j + e) % 3] = t[e];
            }
            f32x4 acc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* lrow = wg + lk * F_RS + (COL0 - 1) + li;
#pragma unroll
            for (int ci = 0; ci < 8; ++ci)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    float av[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) av[u] = lrow[ci * F_CSW + (u >> 1) * 2 * F_RS + (u & 1) * 16 + dx];
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bw[ci][dx], acc[u], 0, 0, 0);
                }
            // lane holds (x channel col, y = y0 + 2*(u>>1) + s_row, x = x0 + (u&1)*16 + 4*lk + r)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int y = y0 + 2 * (u >> 1) + s_row, x = x0 + (u & 1) * 16 + 4 * lk;
                if (y < p.H && x < p.W) {
                    f32x4 v = acc[u];
                    if (q.mask) {
                        const f32x4 a4 = *reinterpret_cast<const f32x4*>(wx + col * F_XCSW + (1 + 2 * (u >> 1) + s_row) * F_XRS + COL0 + (u & 1) * 16 + 4 * lk);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = a4[r] > 0.f ? v[r] * e_scale : 0.f;
                    }
                    float* op = reinterpret_cast<float*>(q.out.ptr) + b * q.out.bstride + col * q.out.cstride + (int64_t)y * q.out.rstride + x;
                    if (p.accumulate) {
                        const f32x4 o4 = pc_ld4(op);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += o4[r];
                    }
                    pc_st4(op, v);
                }
            }
        }
        // ---- weight gradient: D[(s,co)][(ci,v,dx)] += sum_x g[co][y0+2rpi+s][x] * x[ci][y0+2rpi+v-1][x+dx-1]
#pragma unroll
        for (int rpi = 0; rpi < 2; ++rpi) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(ga + 2 * rpi * F_RS), a1 = *reinterpret_cast<const f32x4*>(ga + 2 * rpi * F_RS + 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float a = j < 4 ? a0[j & 3] : a1[j & 3];
                float bv[F_NBLK];
#pragma unroll
                for (int nb = 0; nb < F_NBLK; ++nb) bv[nb] = wx[boff[nb] + 2 * rpi * F_XRS + j];
                bsum += a;
#pragma unroll
                for (int nb = 0; nb < F_NBLK; ++nb) wacc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[nb], wacc[nb], 0, 0, 0);
            }
        }
        b = nb_; y0 = ny0; x0 = nx0;
    }

    // ---- cross-wave reduction through LDS (fixed order), one compacted partial per workgroup (conv3x3_wgrad_wave_kernel's)
    float* lds = ldsf;
    float* part = q.partial + (int64_t)blockIdx.x * F_EC;
    auto wsum = [&](int e) { return ((lds[e] + lds[F_NBLK * 256 + e]) + lds[2 * F_NBLK * 256 + e]) + lds[3 * F_NBLK * 256 + e]; };
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < F_NBLK; ++nb) *reinterpret_cast<f32x4*>(&lds[((wave * F_NBLK + nb) * 64 + lane) * 4]) = wacc[nb];
    __syncthreads();
    for (int idx = tid; idx < 8 * 8 * 9; idx += 256) {
        const int c8 = idx / 72, rem = idx - c8 * 72;
        const int cil = rem / 9, tap = rem - cil * 9, dy = tap / 3, dx = tap - dy * 3;
        const int ng0 = cil * 12 + dy * 3 + dx, ng1 = ng0 + 3;
        const int m0 = c8, m1 = 8 + c8;
        const int e0 = ((ng0 >> 4) * 64 + (m0 >> 2) * 16 + (ng0 & 15)) * 4 + (m0 & 3);
        const int e1 = ((ng1 >> 4) * 64 + (m1 >> 2) * 16 + (ng1 & 15)) * 4 + (m1 & 3);
        part[c8 * 72 + rem] = wsum(e0) + wsum(e1);
    }
    __syncthreads();
    lds[wave * 64 + lane] = bsum;
    __syncthreads();
    if (tid < 8) {
        float t = 0.f;
#pragma unroll
        for (int lk2 = 0; lk2 < 4; ++lk2) {
            const int ea = lk2 * 16 + tid, eb = ea + 8;
            const float sa = ((lds[ea] + lds[64 + ea]) + lds[2 * 64 + ea]) + lds[3 * 64 + ea];
            const float sb = ((lds[eb] + lds[64 + eb]) + lds[2 * 64 + eb]) + lds[3 * 64 + eb];
            t += sa + sb;
        }
        part[8 * 8 * 9 + tid] = t;
    }
}

int launch_bwd_f32(BwdArgs& p, int n, int* nwg_out, hipStream_t stream) {
    static int resident = 0;
    static pc_once_per_device once;
    if (once.need()) {
        const void* fn = reinterpret_cast<const void*>(&conv3x3_bwd_f32_kernel);
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)F_LDS_B);
        if (e != hipSuccess) return (int)e;
        hipFuncAttributes fa;
        e = hipFuncGetAttributes(&fa, fn);
        if (e != hipSuccess) return (int)e;
        resident = pc_resident_workgroups(fa.numRegs, F_LDS_B);
        once.mark();
        if (getenv("POPCORN_CONV_DBG"))
            fprintf(stderr, "conv3x3_bwd_f32: %d regs, %zu B LDS -> %d resident workgroups\n", fa.numRegs, (size_t)F_LDS_B, resident);
    }
    int nwg = resident / n;
    if (nwg > 512) nwg = 512;
    if (nwg > p.ntiles) nwg = p.ntiles;
    if (nwg < 1) nwg = 1;
    const int rounds = (p.ntiles + nwg - 1) / nwg;
    nwg = (p.ntiles + rounds - 1) / rounds;
    hipLaunchKernelGGL(conv3x3_bwd_f32_kernel, dim3(nwg, n), dim3(256), F_LDS_B, stream, p);
    PC_CHECK_LAUNCH();
    *nwg_out = nwg;
    return 0;
}

// ---- fp32 (planar tensors) on the bf16 matrix pipe (round 6): conv3x3_bwd_s3_kernel<GC, POOL> ----------------------------------------
// The channels-last kernel above with fp32 tensors at both ends.  The planar fp32 strips of g (GC = 8 or 16 channels) and x (8 channels)
// are loaded exactly as conv3x3_bwd_f32_kernel loads them (one aligned 16-byte piece per lane and channel, prefetched one strip ahead)
// and SPLIT ONCE, while they are written to LDS: every fp32 value is exactly the sum of three bf16 numbers (common.h: pc_split_pair), so
// a lane turns its 4 pixels x 8 channels into 3 planes x 4 channels-last 16-byte slots.  From there on the operands are
// conv3x3_bwd_cl_kernel's: one ds_read_b128 is the K = 32 = (4 rows x 8 channels) pixel operand of the data gradient, two
// ds_read_b64_tr_b16 the K = 32 pixels operand of the weight gradient; every product is six bf16 x bf16 partial products (the three
// smallest of the nine, together below 2^-23 of the product, are dropped), accumulated in fp32 smallest first -- 6 x 16 matrix cycles per
// 32 K-slots against 8 x 32 on v_mfma_f32_16x16x4_f32, and ONE LDS read per 24 (data gradient) / 6-12 (weight gradient) instructions
// against one per instruction.  Data-gradient operands are swapped against the channels-last kernel (A = pixels, B = weights) so that a
// lane holds FOUR CONSECUTIVE PIXELS of one channel: the planar fp32 epilogue of conv3x3_bwd_f32_kernel (ReLU mask from the x image,
// BN scale, +=, 16-byte stores), or the MaxPool2d(2) backward scatter of conv3x3_mfma_kernel's vector epilogue (POOL: x is the saved
// pooled copy of pool_act).  Three planes of (GC / 8 + 1) strip images per wave = 27 / 34.6 KB: one workgroup per CU, one wave per SIMD,
// up to 512 registers -- the weights' B fragments (all three planes) stay in registers.  GC = 16 with a 16-channel x runs as two
// problems (column blocks of x) over the same g.
// Phase profile of conv3x3_bwd_s3_kernel (debug builds only: tools/build_variant.sh prof -DPOPCORN_BWD_PROF; POPCORN_BWD_PROF=1 prints
// cycles per strip and wave: waiting for the prefetched loads, split + LDS writes, issue of the next prefetch, data-gradient matrix
// phase, epilogue, weight-gradient matrix phase)
#ifdef POPCORN_BWD_PROF
__device__ long long g_bwd_prof[4096 * 8];
#define BQ_DECL long long bq0 = 0, bq1 = 0, bq2 = 0, bq3 = 0, bq4 = 0, bq5 = 0, bq_t = 0, bq_n = 0
#define BQ_NOW(v) do { __builtin_amdgcn_sched_barrier(0); v = (long long)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define BQ_START BQ_NOW(bq_t)
#define BQ_CLOSE(acc) do { BQ_NOW(bq_n); acc += bq_n - bq_t; bq_t = bq_n; } while (0)
#define BQ_WAITVM asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define BQ_WAITLGKM asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define BQ_DUMP do { if (lane == 0) { long long* o_ = g_bwd_prof + ((blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8; \
                     o_[0] = bq0; o_[1] = bq1; o_[2] = bq2; o_[3] = bq3; o_[4] = bq4; o_[5] = bq5; o_[6] = my_tiles; } } while (0)
#else
#define BQ_DECL
#define BQ_START
#define BQ_CLOSE(acc)
#define BQ_WAITVM
#define BQ_WAITLGKM
#define BQ_DUMP
#endif

template <int GC>
struct S3Cfg {
    static constexpr int NG = GC / 8;
    static constexpr int BSL = 34;                             // slots per strip row: logical slot s = pixel x0 - 1 + s, stored at s3_sl(s)
    static constexpr int IMG = SROWS * BSL * 16;               // bytes of one 8-channel plane image
    static constexpr int PLANE = (NG + 1) * IMG;               // one split plane of a wave: NG images of g, one of x
    static constexpr int WAVE_B = 3 * PLANE;
    static constexpr int BW_CO = NG * 24, BW_DYS = 8 * BW_CO;  // weight image of one plane: [dy plane 0..3][x channel][g chunk][dx][8 g ch] bf16
    static constexpr int WPL = 4 * BW_DYS * 2;                 // ... in bytes (prologue only: it overlays the strip images)
    static constexpr int NBLK = 6;                             // weight-gradient accumulator blocks per 8 g channels: [dx][2 halves of x's channels]
    static constexpr int EC = GC * 72 + GC;
    static constexpr size_t RED_B = (size_t)4 * NBLK * 256 * sizeof(float);
    static constexpr size_t LDS_B = (size_t)4 * WAVE_B > RED_B ? (size_t)4 * WAVE_B : RED_B;
    static_assert(3 * WPL <= 4 * WAVE_B, "the prologue's weight image overlays the strip images");
    static constexpr int WAVES_PER_SIMD = (size_t)2 * LDS_B <= 160 * 1024 ? 2 : 1;
};

// Physical slot of logical slot s inside a strip row: an XOR swizzle of the low two bits with the 8-slot block index.  With the plain
// layout the 8 lanes of a ds_write_b128 group (segments 4 slots apart) hit 2 of 8 bank groups (4-way conflict) and the two strip rows of a
// 16-lane ds_read_b128 group overlap (2-way): SQ_LDS_BANK_CONFLICT was 11.1 M of 18.0 M LDS-active cycles per launch, the LDS 70 % busy
// (profiles/r6_pmc_conv_bwd_after.json).  A bank simulator over row strides, per-row rotations and XOR swizzles (tools/lds_bank_sim.py) puts
// this one at 996 LDS cycles per strip against 1,368 (conflict-free: 504; that needs a 48-slot stride = one workgroup per CU).
__device__ __forceinline__ int s3_sl(int s) { return s ^ ((s >> 3) & 3); }

// a wave-uniform value the compiler must keep in a scalar register: without this hipcc re-loads kernel-argument fields (descriptor
// pointers, strides, flags) with s_load + s_waitcnt at every use inside the strip loop -- four dependent scalar-memory round trips per
// epilogue, 1,800 cycles per strip in the first version of this kernel (profiles/r6_conv_bwd_s3_phases.json)
template <typename T>
__device__ __forceinline__ T s3_pin(T v) {
    asm volatile("" : "+s"(v));
    return v;
}
// ... and a pinned base pointer that keeps the GLOBAL address space (a pointer that went through the asm as a generic one comes back as
// flat_load / flat_store, which count on vmcnt AND lgkmcnt: every LDS wait then drains the prefetch)
typedef __attribute__((address_space(1))) char* s3_gptr;
typedef __attribute__((address_space(1))) const f32x4* s3_gld4;
typedef __attribute__((address_space(1))) f32x4* s3_gst4;
__device__ __forceinline__ s3_gptr s3_pin_global(const void* ptr) {
    uint64_t v = reinterpret_cast<uint64_t>(ptr);
    asm volatile("" : "+s"(v));
    return (s3_gptr)v;
}

template <int GC, bool POOL>
__global__ __launch_bounds__(256, S3Cfg<GC>::WAVES_PER_SIMD) void conv3x3_bwd_s3_kernel(const BwdArgs p) {
    using Cfg = S3Cfg<GC>;
    constexpr int NG = Cfg::NG, BSL = Cfg::BSL, IMG = Cfg::IMG, PLANE = Cfg::PLANE, NBLK = Cfg::NBLK, EC = Cfg::EC;
    constexpr int BW_CO = Cfg::BW_CO, BW_DYS = Cfg::BW_DYS;
    constexpr int SL0 = 1;                                                   // slot of pixel x0
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    const BwdProb& q = p.pr[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int s_row = li >> 3, col = li & 7;
    unsigned char* const wimg = ldsb + wave * Cfg::WAVE_B;                   // plane pl at wimg + pl * PLANE: g images, then the x image

    // ---- wave-uniform descriptor fields, pinned in scalar registers for the whole kernel
    const s3_gptr g_base = s3_pin_global(q.g.ptr), x_base = s3_pin_global(q.x.ptr), o_base = s3_pin_global(q.out.ptr);
    const unsigned g_bs = s3_pin((unsigned)q.g.bstride), g_cs = s3_pin((unsigned)q.g.cstride * 4u), g_rs = s3_pin((unsigned)q.g.rstride);
    const unsigned x_bs = s3_pin((unsigned)q.x.bstride), x_cs = s3_pin((unsigned)q.x.cstride * 4u), x_rs = s3_pin((unsigned)q.x.rstride);
    const unsigned o_bs = s3_pin((unsigned)q.out.bstride), o_cs = s3_pin((unsigned)q.out.cstride), o_rs = s3_pin((unsigned)q.out.rstride);
    const int H = s3_pin(p.H), W = s3_pin(p.W);
    const int maskf = s3_pin(q.mask), accf = s3_pin(p.accumulate);
    // ablation switches (pc_debug_conv_bwd; tools/time_conv_bwd.py --ablate) exist in -DPOPCORN_CONV_ABLATE builds only (tools/build_variant.sh):
    // as run-time flags they made commit() conditional, and a path that may skip it leaves its loads pending -- the compiler then waits
    // vmcnt(0) before it re-uses their registers for the next prefetch
#ifdef POPCORN_CONV_ABLATE
    const int dbg = s3_pin(p.dbg);
#else
    constexpr int dbg = 0;
#endif
    const s3_gptr a_base = s3_pin_global(POOL ? q.pool_act.ptr : q.x.ptr);
    const unsigned a_bs = s3_pin((unsigned)(POOL ? q.pool_act.bstride : 0)), a_cs = s3_pin((unsigned)(POOL ? q.pool_act.cstride : 0)),
                   a_rs = s3_pin((unsigned)(POOL ? q.pool_act.rstride : 0));

    // ---- loader: lane = (row of the 6-row strip, 16-byte segment of the 40-float row x0 - 4 ..), both tensors.  Byte offsets inside a
    //      tensor are 32-bit (the launcher checks the extents): one scalar base per channel + one vector offset per lane
    const int l_r = lane / 10, l_seg = lane - l_r * 10;
    const bool l_act = lane < 60;
    f32x4 RG[GC], RX[8];
    bool rvalid = false;
    int c_sl[4];                                                             // physical slots of the lane's four pixels (commit)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int sl = 4 * l_seg + e - 3;
        c_sl[e] = s3_sl(sl < 0 ? 0 : (sl > 33 ? 33 : sl));
    }
    auto issue = [&](int b, int y0, int x0) {
        const int xg = x0 - 4 + 4 * l_seg, y = y0 - 1 + l_r;
        const bool ok = l_act && xg >= 0 && xg < W && (unsigned)y < (unsigned)H;
        rvalid = ok;
        const unsigned go = ok ? ((unsigned)b * g_bs + (unsigned)y * g_rs + (unsigned)xg) * 4u : 0u;
        const unsigned xo = ok ? ((unsigned)b * x_bs + (unsigned)y * x_rs + (unsigned)xg) * 4u : 0u;
#pragma unroll
        for (int it = 0; it < GC; ++it) RG[it] = *(s3_gld4)(g_base + (size_t)(it * g_cs) + go);
#pragma unroll
        for (int it = 0; it < 8; ++it) RX[it] = *(s3_gld4)(x_base + (size_t)(it * x_cs) + xo);
    };
    // split + transpose while staging: pixel e of the lane's four = slot 4 * l_seg + e - 3 of row l_r (the outer three pixels of the first and
    // of the last segment are not part of the strip), its 8 channels = one 16-byte slot per plane
    auto commit = [&]() {
        // strips that touch the image border: zero what lies outside (wave-uniform test; interior strips skip the selects)
        if (__builtin_amdgcn_ballot_w64(l_act && !rvalid) != 0) {
#pragma unroll
            for (int it = 0; it < GC; ++it)
#pragma unroll
                for (int e = 0; e < 4; ++e) RG[it][e] = rvalid ? RG[it][e] : 0.f;
#pragma unroll
            for (int it = 0; it < 8; ++it)
#pragma unroll
                for (int e = 0; e < 4; ++e) RX[it][e] = rvalid ? RX[it][e] : 0.f;
        }
        u32x4* const d0 = reinterpret_cast<u32x4*>(wimg) + l_r * BSL;
#pragma unroll
        for (int c = 0; c < NG + 1; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                u32x4 q1, q2, q3;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const float v0 = c < NG ? RG[(c < NG ? c : 0) * 8 + 2 * d][e] : RX[2 * d][e];
                    const float v1 = c < NG ? RG[(c < NG ? c : 0) * 8 + 2 * d + 1][e] : RX[2 * d + 1][e];
                    unsigned a1, a2, a3;
                    pc_split_pair(v0, v1, a1, a2, a3);
                    q1[d] = a1; q2[d] = a2; q3[d] = a3;
                }
                // e = 3 exists in every segment but the last, e = 0 in every segment but the first, e = 1, 2 in the inner ones
                const bool w_ok = l_act && (e == 3 ? l_seg < 9 : (e == 0 ? l_seg > 0 : (l_seg > 0 && l_seg < 9)));
                if (w_ok) {
                    u32x4* d = d0 + c * (IMG / 16) + c_sl[e];
                    d[0] = q1;
                    d[PLANE / 16] = q2;
                    d[2 * (PLANE / 16)] = q3;
                }
            }
    };
    const int gdim = s3_pin((int)gridDim.x), ntl = s3_pin(p.ntiles);
    const int my_tiles = ntl > (int)blockIdx.x ? (ntl - 1 - (int)blockIdx.x) / gdim + 1 : 0;
    auto strip_coords = [&](int k, int& b, int& y0, int& x0) {
        const int tile = pc_xcd_remap(blockIdx.x + k * gdim, ntl);
        b = (int)pc_div((uint32_t)tile, p.div_tpi);
        const int rem = tile - b * p.tiles_x * p.tiles_y;
        const int ty = (int)pc_div((uint32_t)rem, p.div_tx);
        x0 = (rem - ty * p.tiles_x) * TW;
        y0 = ty * TH + 4 * wave;
    };
    int b = 0, y0 = 0, x0 = 0;
    if (my_tiles > 0) {
        strip_coords(0, b, y0, x0);
        issue(b, y0, x0);
    }

    // ---- data-gradient weights, split: "output" channel = x channel, K channel = g channel, taps flipped; three planes of
    //      [dy plane 0..3][x ch][g chunk][dx][8 g ch] bf16, plane dy = 3 all zero (the (row, output row) pairs that are not a tap).  The image
    //      lives where the strip images will (the first strip is still in registers) and only until the B fragments are read from it.
    unsigned short* const w2h = reinterpret_cast<unsigned short*>(ldsb);
    constexpr int NWR = (GC * 8 * 9 + 255) / 256;
    float wreg[NWR];
#pragma unroll
    for (int k = 0; k < NWR; ++k) {
        const int e = tid + k * 256;
        const int ec = e < GC * 72 ? e : 0;
        const int tap = ec % 9, gch = (ec / 9) % GC, xch = ec / (9 * GC);
        wreg[k] = q.w[xch * 9 + gch * q.w_ci_stride + (8 - tap)];
    }
    float bn_g = 1.f, bn_v = 1.f;
    if (maskf && q.bn.gamma) { bn_g = q.bn.gamma[col]; bn_v = q.bn.var[col]; }
    for (int e = tid; e < 3 * 4 * BW_DYS / 2; e += 256) reinterpret_cast<unsigned*>(w2h)[e] = 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NWR; ++k) {
        const int e = tid + k * 256;
        if (e < GC * 72) {
            const int tap = e % 9, gch = (e / 9) % GC, xch = e / (9 * GC);
            const float a1 = pc_bf16r(wreg[k]), r1 = wreg[k] - a1, a2 = pc_bf16r(r1), a3 = r1 - a2;
            const int o = (tap / 3) * BW_DYS + xch * BW_CO + (gch / 8) * 24 + (tap % 3) * 8 + (gch % 8);
            w2h[o] = (unsigned short)(__float_as_uint(a1) >> 16);
            w2h[4 * BW_DYS + o] = (unsigned short)(__float_as_uint(a2) >> 16);
            w2h[8 * BW_DYS + o] = pc_f2bf(a3);
        }
    }
    __syncthreads();
    const float e_scale = (maskf && q.bn.gamma) ? bn_g * (1.0f / sqrtf(bn_v + q.bn.eps)) : 1.f;
    asm volatile("" : : "v"(e_scale));
    // B fragments of the data gradient, all planes, for the whole kernel: lane (n = (s_row, x channel col), K group lk = strip row)
    bf16x8 wq[3][NG][3];
    {
        const unsigned short* const wlane = w2h + (((unsigned)(lk - s_row) <= 2u) ? lk - s_row : 3) * BW_DYS + col * BW_CO;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int gc = 0; gc < NG; ++gc)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
                    wq[pl][gc][dx] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wlane + pl * 4 * BW_DYS + gc * 24 + dx * 8));
    }
    __syncthreads();                 // the weight image is dead from here on: the strip images take its place

    // ---- weight-gradient reads (conv3x3_bwd_cl_kernel's): lane supplies pixel row j = li >> 2 and column quad tq = li & 3
    const int t_j = li >> 2, t_q = li & 3;
    int a_off[2], b_off[3][2];                                               // [.., hi]: the second transposing read sits 4 pixels on
#pragma unroll
    for (int hi = 0; hi < 2; ++hi) {
        a_off[hi] = ((1 + (t_q >> 1)) * BSL + s3_sl(SL0 + 8 * lk + t_j + 4 * hi)) * 16 + 8 * (t_q & 1);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) b_off[dx][hi] = (t_q * BSL + s3_sl((SL0 - 1) + 8 * lk + t_j + dx + 4 * hi)) * 16;
    }
    int dg_sl[3][2];                                                         // data gradient: the lane's pixel slot per (dx, half of the strip row)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int h = 0; h < 2; ++h) dg_sl[dx][h] = s3_sl((SL0 - 1) + li + dx + 16 * h);
    f32x4 wacc[NG][NBLK], bacc[NG];
    const bf16x8 ones8 = __builtin_bit_cast(bf16x8, u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});
#pragma unroll
    for (int mb = 0; mb < NG; ++mb) {
        bacc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NBLK; ++i) wacc[mb][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const unsigned char* const ixb = wimg + NG * IMG;                       // plane 0 of the x image
    // sign of the layer's input at the lane's output pixels: first split plane of the x image (rn_bf16 keeps sign and zero)
    const unsigned short* const xs0 = reinterpret_cast<const unsigned short*>(ixb + ((1 + s_row) * BSL) * 16) + col;
    int m_sl[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) m_sl[h][r] = s3_sl(SL0 + 16 * h + 4 * lk + r);

    BQ_DECL;
    BQ_START;
    for (int k = 0; k < my_tiles; ++k) {
        BQ_WAITVM;
        BQ_CLOSE(bq0);
        if (!(dbg & 1)) commit();
        BQ_WAITLGKM;
        BQ_CLOSE(bq1);
        int nb_ = b, ny0 = y0, nx0 = x0;
        if (k + 1 < my_tiles) {
            strip_coords(k + 1, nb_, ny0, nx0);
            if (!(dbg & 8)) issue(nb_, ny0, nx0);
        }
        BQ_CLOSE(bq2);
        // the mask operands of the epilogue, read ahead of the matrix phase
        unsigned short xm[POOL ? 1 : 16];
        if constexpr (!POOL) {
            if (maskf) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) xm[4 * u + r] = xs0[((u >> 1) * 2 * BSL + m_sl[u & 1][r]) * 8];
            }
        }
        // ---- data gradient: a 3x3 conv over g, K = 4 rows x 8 channels per instruction, six partial products per K-step
        f32x4 acc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!(dbg & 2))
#pragma unroll
        for (int gc = 0; gc < NG; ++gc)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                bf16x8 av[3][4];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const u32x4* lrow = reinterpret_cast<const u32x4*>(wimg + pl * PLANE + gc * IMG) + lk * BSL;
#pragma unroll
                    for (int u = 0; u < 4; ++u) av[pl][u] = __builtin_bit_cast(bf16x8, lrow[(u >> 1) * 2 * BSL + dg_sl[dx][u & 1]]);
                }
#pragma unroll
                for (int pw = 2; pw >= 0; --pw)               // weight split index; pixel split indices 2 - pw .. 0: smallest products first
#pragma unroll
                    for (int pa = 2 - pw; pa >= 0; --pa)
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[pa][u], wq[pw][gc][dx], acc[u], 0, 0, 0);
            }
#ifdef POPCORN_BWD_PROF
        asm volatile("" : : "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]));
#endif
        BQ_CLOSE(bq3);
        // lane holds (x channel col, y = y0 + 2*(u>>1) + s_row, x = x0 + (u&1)*16 + 4*lk + r)
        if (dbg & 16) {
        } else if constexpr (POOL) {
            // MaxPool2d(2) backward: the lane's four pooled pixels cover 8 x 2 full-resolution pixels = two 16-byte pieces per row of
            // pool_act and of the accumulated output; the gradient goes to the first arg-max of every window
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int y = y0 + 2 * (u >> 1) + s_row, x = x0 + (u & 1) * 16 + 4 * lk;
                if (y < H && x < W) {
                    const f32x4 v = acc[u];
                    const s3_gptr a0 = a_base + (size_t)(((unsigned)b * a_bs + (unsigned)col * a_cs + (unsigned)(2 * y) * a_rs + (unsigned)(2 * x)) * 4u);
                    const s3_gptr o0 = o_base + (size_t)(((unsigned)b * o_bs + (unsigned)col * o_cs + (unsigned)(2 * y) * o_rs + (unsigned)(2 * x)) * 4u);
                    f32x4 A[2][2], O[2][2];
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            A[rr][h] = *(s3_gld4)(a0 + (size_t)((rr * a_rs + 4 * h) * 4u));
                            O[rr][h] = *(s3_gld4)(o0 + (size_t)((rr * o_rs + 4 * h) * 4u));
                        }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int h = r >> 1, e = (r & 1) * 2;
                        const float w00 = A[0][h][e], w01 = A[0][h][e + 1], w10 = A[1][h][e], w11 = A[1][h][e + 1];
                        int am = 0;
                        float m = w00;
                        if (w01 > m) { m = w01; am = 1; }
                        if (w10 > m) { m = w10; am = 2; }
                        if (w11 > m) { m = w11; am = 3; }
                        const float gv = m > 0.f ? v[r] * e_scale : 0.f;
                        O[0][h][e] += am == 0 ? gv : 0.f;
                        O[0][h][e + 1] += am == 1 ? gv : 0.f;
                        O[1][h][e] += am == 2 ? gv : 0.f;
                        O[1][h][e + 1] += am == 3 ? gv : 0.f;
                    }
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                        for (int h = 0; h < 2; ++h) *(s3_gst4)(o0 + (size_t)((rr * o_rs + 4 * h) * 4u)) = O[rr][h];
                }
            }
        } else {
            if (maskf) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const unsigned short h = xm[4 * u + r];
                        acc[u][r] = (h != 0 && !(h & 0x8000u)) ? acc[u][r] * e_scale : 0.f;
                    }
            }
            const unsigned ob = ((unsigned)b * o_bs + (unsigned)col * o_cs + (unsigned)(y0 + s_row) * o_rs + (unsigned)(x0 + 4 * lk)) * 4u;
            const bool full = __builtin_amdgcn_readfirstlane((y0 + 4 <= H) && (x0 + TW <= W));
            if (full) {
                // interior strip: no bounds checks, four 16-byte stores from one base address
                s3_gptr op[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) op[u] = o_base + (size_t)(ob + ((u >> 1) * 2 * o_rs + (u & 1) * 16) * 4u);
                if (accf) {
                    f32x4 o4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) o4[u] = *(s3_gld4)op[u];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[u][r] += o4[u][r];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) *(s3_gst4)op[u] = acc[u];
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int y = y0 + 2 * (u >> 1) + s_row, x = x0 + (u & 1) * 16 + 4 * lk;
                    if (y < H && x < W) {
                        const s3_gptr op = o_base + (size_t)(ob + ((u >> 1) * 2 * o_rs + (u & 1) * 16) * 4u);
                        f32x4 v = acc[u];
                        if (accf) {
                            const f32x4 o4 = *(s3_gld4)op;
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] += o4[r];
                        }
                        *(s3_gst4)op = v;
                    }
                }
            }
        }
        BQ_CLOSE(bq4);
        // ---- weight gradient of the strip: D_dx[(s,co)][(v,ci)] += sum_x g[co][y0+2rpi+s][x] * x[ci][y0+2rpi+v-1][x+dx-1]
        //      three accumulator blocks (the dx of one half of x's channels) take turns: no instruction waits for its predecessor
        if (!(dbg & 4))
#pragma unroll
        for (int rpi = 0; rpi < 2; ++rpi) {
            bf16x8 av[NG][3];
#pragma unroll
            for (int mb = 0; mb < NG; ++mb)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const unsigned char* ga = wimg + pl * PLANE + mb * IMG + 2 * rpi * BSL * 16;
                    av[mb][pl] = bw_pair(bw_tr(ga + a_off[0]), bw_tr(ga + a_off[1]));
                }
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                bf16x8 bv[3][3];
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        const unsigned char* xb = ixb + pl * PLANE + 2 * rpi * BSL * 16 + 8 * nb;
                        bv[dx][pl] = bw_pair(bw_tr(xb + b_off[dx][0]), bw_tr(xb + b_off[dx][1]));
                    }
                if (nb == 0) {
#pragma unroll
                    for (int pl = 2; pl >= 0; --pl)
#pragma unroll
                        for (int mb = 0; mb < NG; ++mb) bacc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[mb][pl], ones8, bacc[mb], 0, 0, 0);
                }
#pragma unroll
                for (int pg = 2; pg >= 0; --pg)
#pragma unroll
                    for (int px = 2 - pg; px >= 0; --px)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                            for (int mb = 0; mb < NG; ++mb)
                                wacc[mb][dx * 2 + nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[mb][pg], bv[dx][px], wacc[mb][dx * 2 + nb], 0, 0, 0);
            }
        }
#ifdef POPCORN_BWD_PROF
#pragma unroll
        for (int mb = 0; mb < NG; ++mb)
#pragma unroll
            for (int i = 0; i < NBLK; ++i) asm volatile("" : : "v"(wacc[mb][i]));
#endif
        BQ_CLOSE(bq5);
        b = nb_; y0 = ny0; x0 = nx0;
    }
    BQ_DUMP;

    // ---- cross-wave reduction through LDS (fixed order), one compacted partial per workgroup (conv3x3_bwd_cl_kernel's, XC = 8)
    float* lds = reinterpret_cast<float*>(ldsb);
    float* part = q.partial + (int64_t)blockIdx.x * EC;
    auto wsum = [&](int e) { return ((lds[e] + lds[NBLK * 256 + e]) + lds[2 * NBLK * 256 + e]) + lds[3 * NBLK * 256 + e]; };
#pragma unroll
    for (int mb = 0; mb < NG; ++mb) {
        __syncthreads();
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb) *reinterpret_cast<f32x4*>(&lds[((wave * NBLK + nb) * 64 + lane) * 4]) = wacc[mb][nb];
        __syncthreads();
        for (int idx = tid; idx < 8 * 72; idx += 256) {
            const int c8 = idx / 72, rem = idx - c8 * 72;
            const int cil = rem / 9, tap = rem - cil * 9, dy = tap / 3, dx = tap - dy * 3;
            const int blk = dx * 2 + (cil >> 2);
            const int n0 = 4 * dy + (cil & 3), n1 = n0 + 4;
            const int m0 = c8, m1 = 8 + c8;
            const int e0 = (blk * 64 + (m0 >> 2) * 16 + n0) * 4 + (m0 & 3);
            const int e1 = (blk * 64 + (m1 >> 2) * 16 + n1) * 4 + (m1 & 3);
            part[(mb * 8 + c8) * 72 + rem] = wsum(e0) + wsum(e1);
        }
    }
    __syncthreads();
    if (li == 0) {
#pragma unroll
        for (int mb = 0; mb < NG; ++mb) *reinterpret_cast<f32x4*>(&lds[(wave * NG + mb) * 16 + 4 * lk]) = bacc[mb];
    }
    __syncthreads();
    if (tid < GC) {
        const int mb = tid >> 3, c8 = tid & 7;
        const int ea = mb * 16 + c8, eb = ea + 8;
        const float sa = ((lds[ea] + lds[NG * 16 + ea]) + lds[2 * NG * 16 + ea]) + lds[3 * NG * 16 + ea];
        const float sb = ((lds[eb] + lds[NG * 16 + eb]) + lds[2 * NG * 16 + eb]) + lds[3 * NG * 16 + eb];
        part[GC * 72 + tid] = sa + sb;
    }
}

template <int GC, bool POOL>
int launch_bwd_s3(BwdArgs& p, int n, int* nwg_out, hipStream_t stream) {
    using Cfg = S3Cfg<GC>;
    static int resident = 0;
    static pc_once_per_device once;
    if (once.need()) {
        const void* fn = reinterpret_cast<const void*>(&conv3x3_bwd_s3_kernel<GC, POOL>);
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_B);
        if (e != hipSuccess) return (int)e;
        hipFuncAttributes fa;
        e = hipFuncGetAttributes(&fa, fn);
        if (e != hipSuccess) return (int)e;
        resident = pc_resident_workgroups(fa.numRegs, Cfg::LDS_B);
        once.mark();
        if (getenv("POPCORN_CONV_DBG"))
            fprintf(stderr, "conv3x3_bwd_s3<%d,%d>: %d regs, %zu B LDS -> %d resident workgroups\n", GC, (int)POOL, fa.numRegs, (size_t)Cfg::LDS_B, resident);
    }
    int nwg = resident / n;
    if (nwg > 512) nwg = 512;
    if (nwg > p.ntiles) nwg = p.ntiles;
    if (nwg < 1) nwg = 1;
    const int rounds = (p.ntiles + nwg - 1) / nwg;
    nwg = (p.ntiles + rounds - 1) / rounds;
    hipLaunchKernelGGL((conv3x3_bwd_s3_kernel<GC, POOL>), dim3(nwg, n), dim3(256), Cfg::LDS_B, stream, p);
    PC_CHECK_LAUNCH();
    *nwg_out = nwg;
#ifdef POPCORN_BWD_PROF
    if (getenv("POPCORN_BWD_PROF")) {
        static long long hp[4096 * 8];
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_bwd_prof), sizeof(hp));
        double tot[6] = {0}, strips = 0;
        const int nw = nwg * n * 4 < 4096 ? nwg * n * 4 : 4096;
        for (int w = 0; w < nw; ++w) {
            for (int k = 0; k < 6; ++k) tot[k] += (double)hp[w * 8 + k];
            strips += (double)hp[w * 8 + 6];
        }
        fprintf(stderr, "conv3x3_bwd_s3<%d,%d> phases, cycles per strip and wave (load wait, split + LDS writes, prefetch issue, dgrad, epilogue, wgrad; %d waves, %.1f strips each):",
                GC, (int)POOL, nw, strips / nw);
        for (int k = 0; k < 6; ++k) fprintf(stderr, " %.0f", tot[k] / strips);
        fprintf(stderr, "\n");
    }
#endif
    return 0;
}
}  // namespace

static int g_bwd_dbg = 0;
extern "C" void pc_debug_conv_bwd(int dbg) { g_bwd_dbg = dbg; }

extern "C" int pc_conv3x3_bwd_group(int n, const pc_conv_bwd_desc* d, int Cin_total, int c0, int accumulate, int B, int H, int W,
                                    int* nwg_out, void* stream) {
    if (n < 1 || n > MAXG || !d || !nwg_out) return PC_EINVAL;
    const bool f32 = g_pc_precision != PC_PREC_BF16;
    BwdArgs p{};
    int GC = 0, XC = 0;
    bool use_s3 = true;
    for (int i = 0; i < n; ++i) {
        if (!d[i].g || !d[i].x || !d[i].w || !d[i].out || !d[i].ws) return PC_EINVAL;
        BwdProb& q = p.pr[i];
        q.g = *d[i].g; q.x = *d[i].x;
        if (i == 0) { GC = q.g.C; XC = q.x.C; }
        const int c0i = c0 + d[i].c0_add;
        if (q.g.C != GC || q.x.C != XC || q.x.mode != PC_SRC_DIRECT || q.g.mode != PC_SRC_DIRECT || c0i < 0 || c0i + XC > Cin_total)
            return PC_EINVAL;
        if (f32) {
            // planar fp32, 16-byte aligned rows, x (8 channels) placed at (0, 0) with the extent of g; 8 gradient channels, or -- split-operand
            // form only -- 16, or the Down blocks' pool_act scatter
            auto al = [](const void* ptr, int64_t bs, int64_t cs, int rs) {
                return (reinterpret_cast<uintptr_t>(ptr) & 15) == 0 && bs % 4 == 0 && cs % 4 == 0 && rs % 4 == 0;
            };
            const pc_dst& o = *d[i].out;
            // (the split-operand kernel addresses with 32-bit byte offsets inside a tensor: extents beyond 4 GB keep the fp32-MFMA kernel)
            auto fits32 = [&](int64_t bs) { return bs > 0 && (int64_t)B * bs < ((int64_t)1 << 30); };
            bool s3 = pc_conv_split_on() != 0 && fits32(q.g.bstride) && fits32(q.x.bstride) && fits32(o.bstride) &&
                      (!d[i].pool_act || fits32(d[i].pool_act->bstride));
            use_s3 = use_s3 && s3;
            if (!((GC == 8 && !d[i].pool_act) || (s3 && (GC == 8 || GC == 16))) || XC != 8 || W % 4 != 0 || q.g.dtype != PC_F32 ||
                q.x.dtype != PC_F32 || o.dtype != PC_F32 ||
                !pc_planar(q.g) || !pc_planar(q.x) || !pc_planar(o) || q.x.oy != 0 || q.x.ox != 0 || q.x.H != H || q.x.W != W ||
                !al(q.g.ptr, q.g.bstride, q.g.cstride, q.g.rstride) || !al(q.x.ptr, q.x.bstride, q.x.cstride, q.x.rstride) ||
                !al(o.ptr, o.bstride, o.cstride, o.rstride))
                return PC_EINVAL;
            if (d[i].pool_act) {
                const pc_src& a = *d[i].pool_act;
                // (the scatter always accumulates, as in bf16 mode)
                if (a.dtype != PC_F32 || !pc_planar(a) || a.mode != PC_SRC_DIRECT || a.oy || a.ox || !al(a.ptr, a.bstride, a.cstride, a.rstride))
                    return PC_EINVAL;
            }
        } else if (!pc_cl_ok(q.g) || !pc_cl_ok(q.x) || !pc_cl_ok(*d[i].out)) {
            return PC_EINVAL;
        }
        q.w = d[i].w + (int64_t)c0i * 9;
        q.w_ci_stride = Cin_total * 9;
        q.mask = d[i].x_bn != nullptr;
        if (d[i].x_bn) q.bn = *d[i].x_bn;
        q.out = *d[i].out;
        if ((d[i].pool_act != nullptr) != (d[0].pool_act != nullptr)) return PC_EINVAL;
        if (d[i].pool_act) {
            // the scatter always accumulates and always applies the producer's ReLU / BN factor
            q.pool_act = *d[i].pool_act;
            if ((!f32 && !pc_cl_ok(q.pool_act)) || !d[i].x_bn || q.pool_act.C != XC || q.pool_act.H < 2 * H || q.pool_act.W < 2 * W) return PC_EINVAL;
        }
        q.partial = reinterpret_cast<float*>(d[i].ws);
    }
    p.accumulate = accumulate;
    p.dbg = g_bwd_dbg;
    p.B = B; p.H = H; p.W = W;
    p.tiles_x = (W + TW - 1) / TW;
    p.tiles_y = (H + TH - 1) / TH;
    p.ntiles = B * p.tiles_x * p.tiles_y;
    if (p.ntiles <= 0) { *nwg_out = 0; return 0; }
    p.div_tx = pc_make_fastdiv(p.tiles_x);
    p.div_tpi = pc_make_fastdiv(p.tiles_x * p.tiles_y);
    hipStream_t st = (hipStream_t)stream;
    if (f32) {
        if (!use_s3) return launch_bwd_f32(p, n, nwg_out, st);
        if (d[0].pool_act) return GC == 8 ? launch_bwd_s3<8, true>(p, n, nwg_out, st) : launch_bwd_s3<16, true>(p, n, nwg_out, st);
        return GC == 8 ? launch_bwd_s3<8, false>(p, n, nwg_out, st) : launch_bwd_s3<16, false>(p, n, nwg_out, st);
    }
    if (d[0].pool_act) {
        if (GC == 16 && XC == 8) return launch_bwd<16, 8, true>(p, n, nwg_out, st);
        if (GC == 16 && XC == 16) return launch_bwd<16, 16, true>(p, n, nwg_out, st);
        return PC_EINVAL;
    }
    if (GC == 8 && XC == 8) return launch_bwd<8, 8, false>(p, n, nwg_out, st);
    if (GC == 8 && XC == 16) return launch_bwd<8, 16, false>(p, n, nwg_out, st);
    if (GC == 16 && XC == 16) return launch_bwd<16, 16, false>(p, n, nwg_out, st);
    return PC_EINVAL;
}

// fp32 mode: would pc_conv3x3_bwd_group take this problem?  (the executors ask before they choose between the fused launch and the
// data-gradient + weight-gradient pair)
extern "C" int pc_conv3x3_bwd_ok(const pc_src* g, const pc_src* x, const pc_dst* out, const pc_src* pool_act, int B, int H, int W) {
    if (!g || !x || !out || g_pc_precision == PC_PREC_BF16) return 0;
    auto al = [](const void* ptr, int64_t bs, int64_t cs, int rs) {
        return (reinterpret_cast<uintptr_t>(ptr) & 15) == 0 && bs % 4 == 0 && cs % 4 == 0 && rs % 4 == 0;
    };
    auto fits32 = [](int64_t bs, int B) { return bs > 0 && (int64_t)B * bs < ((int64_t)1 << 30); };
    const bool s3 = pc_conv_split_on() != 0 && fits32(g->bstride, B) && fits32(x->bstride, B) && fits32(out->bstride, B) &&
                    (!pool_act || fits32(pool_act->bstride, B));
    if (!((g->C == 8 && !pool_act) || (s3 && (g->C == 8 || g->C == 16))) || x->C != 8 || W % 4 != 0) return 0;
    if (g->dtype != PC_F32 || x->dtype != PC_F32 || out->dtype != PC_F32 || !pc_planar(*g) || !pc_planar(*x) || !pc_planar(*out)) return 0;
    if (g->mode != PC_SRC_DIRECT || x->mode != PC_SRC_DIRECT || x->oy || x->ox || x->H != H || x->W != W) return 0;
    if (!al(g->ptr, g->bstride, g->cstride, g->rstride) || !al(x->ptr, x->bstride, x->cstride, x->rstride) ||
        !al(out->ptr, out->bstride, out->cstride, out->rstride))
        return 0;
    if (pool_act) {
        const pc_src& a = *pool_act;
        if (a.dtype != PC_F32 || !pc_planar(a) || a.mode != PC_SRC_DIRECT || a.oy || a.ox || a.C != 8 || a.H < 2 * H || a.W < 2 * W ||
            !al(a.ptr, a.bstride, a.cstride, a.rstride))
            return 0;
    }
    return 1;
}
