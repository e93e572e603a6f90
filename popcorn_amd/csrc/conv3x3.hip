// conv3x3.hip -- 3x3 / pad-1 convolution on fp32 MFMA (v_mfma_f32_16x16x4_f32) for gfx950.
//
// Replaces nn.Conv2d(3, padding=1) + BatchNorm2d(eval) + ReLU (reference model/DDA_model/utils/networks.py:259-266)
// and, with transposed weight indexing, its autograd data-gradient.
//
// GEMM mapping ("vertical pair" im2col): one MFMA produces a 16(x) x 2(rows) patch for 8 output channels:
//     M (16)  = 16 consecutive output x
//     N (16)  = (s, co):  s in {0,1} = which of the two output rows, co = 8 output channels
//     K (4)   = v: the 4 input rows (y0-1 .. y0+2) that the two output rows touch
//     one MFMA per (input channel ci, horizontal tap dx);  B[v][(s,co)] = w[co][ci][v-s][dx] (0 when v-s not in 0..2)
// so 9 useful taps ride on 12 K-slots: 75 % MFMA efficiency at Cout = 8 (a plain im2col with N = Cout = 8 wastes half
// of every MFMA) and no padding waste in N for any Cout that is a multiple of 8.
//
// Data movement: a workgroup (4 waves) owns a 32 x 16 output tile; the (CHUNK x 18 x 34) input halo tile is staged
// in LDS with row stride 48 (= 16 mod 32 banks), which makes the A-operand read -- lane (i = x, k = row) ->
// lds[ci][2*rp + k][x + dx] -- bank-conflict free.  Weights live in registers as ready-made B fragments for the
// whole kernel; workgroups are persistent over tiles (XCD-aware tile order) so they are fetched once.
#include "common.h"

namespace {

constexpr int TW = 32, TH = 16;          // output tile
constexpr int LROWS = TH + 2, LCOLS = TW + 2;
constexpr int RS = 48;                   // LDS row stride  (== 16 mod 32)
constexpr int CS = LROWS * RS;           // LDS channel stride

struct ConvArgs {
    pc_src a, b;          // input sources (channels a.C then b.C)
    const float* w;       // weights
    int w_co_stride;      // element stride between output channels in w
    int w_ci_stride;      // element stride between input channels in w
    int w_flip;           // 1: tap index 8 - t (dgrad)
    pc_bn bn;             // FWD: this layer's BN; DGRAD: BN of the layer that produced `act`
    int relu;             // FWD
    const float* act;     // DGRAD: post-ReLU activations of the producer (NULL = plain)
    int64_t act_bstride, act_cstride;
    int act_rstride;
    int pool;             // DGRAD: max-pool backward scatter into a 2x resolution output
    int accumulate;       // DGRAD: out += instead of out =
    int outH, outW;       // DGRAD+pool: extent of the full-resolution output
    pc_dst out;
    int B, H, W;
    int tiles_x, tiles_y, ntiles;
};

enum { MODE_FWD = 0, MODE_DGRAD = 1 };

template <int CIN, int COUT, int MODE>
__global__ __launch_bounds__(256) void conv3x3_mfma_kernel(const ConvArgs p) {
    constexpr int CHUNK = CIN < 16 ? CIN : 16;
    constexpr int NCHUNK = CIN / CHUNK;
    constexpr int NB = COUT / 8;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;      // A: (i, k);  B: (k, n = li);  D: (n = li, rows 4*lk + r)
    const int s_row = li >> 3, col = li & 7;

    // ---- B fragments: bw[ci][dx][nb] = w[co = nb*8+col][ci][dy = lk - s_row][dx]
    float bw[CIN][3][NB];
    {
        const int dy = lk - s_row;
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    float v = 0.f;
                    if (dy >= 0 && dy <= 2) {
                        int tap = dy * 3 + dx;
                        if (p.w_flip) tap = 8 - tap;
                        v = p.w[(nb * 8 + col) * p.w_co_stride + ci * p.w_ci_stride + tap];
                    }
                    bw[ci][dx][nb] = v;
                }
    }
    // ---- per-lane epilogue constants for co = nb*8 + col
    float e_scale[NB], e_shift[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        if (MODE == MODE_FWD || p.act != nullptr) pc_bn_fold(p.bn, nb * 8 + col, e_scale[nb], e_shift[nb]);
        else { e_scale[nb] = 1.f; e_shift[nb] = 0.f; }
    }

    const int CA = p.a.C;
    for (int t = blockIdx.x; t < p.ntiles; t += gridDim.x) {
        const int tile = pc_xcd_remap(t, p.ntiles);
        const int tx = tile % p.tiles_x;
        const int ty = (tile / p.tiles_x) % p.tiles_y;
        const int b = tile / (p.tiles_x * p.tiles_y);
        const int x0 = tx * TW, y0 = ty * TH;

        f32x4 acc[4][NB];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[u][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
        for (int ch = 0; ch < NCHUNK; ++ch) {
            __syncthreads();   // previous readers of the LDS tile are done
            for (int idx = tid; idx < CHUNK * LROWS * LCOLS; idx += 256) {
                const int c = idx % LCOLS;
                const int r = (idx / LCOLS) % LROWS;
                const int ci = idx / (LCOLS * LROWS);
                const int cg = ch * CHUNK + ci;
                const float v = cg < CA ? pc_fetch(p.a, b, cg, y0 - 1 + r, x0 - 1 + c, p.H, p.W)
                                        : pc_fetch(p.b, b, cg - CA, y0 - 1 + r, x0 - 1 + c, p.H, p.W);
                lds[ci * CS + r * RS + c] = v;
            }
            __syncthreads();
#pragma unroll
            for (int ci = 0; ci < CHUNK; ++ci) {
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    float av[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int rp = 2 * wave + (u >> 1), xb = u & 1;
                        av[u] = lds[ci * CS + (2 * rp + lk) * RS + xb * 16 + li + dx];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[u][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bw[ch * CHUNK + ci][dx][nb],
                                                                              acc[u][nb], 0, 0, 0);
                }
            }
        }

        // ---- epilogue: lane holds (co = nb*8+col, y = y0 + 2*rp + s_row, x = x0 + xb*16 + 4*lk + r), r = 0..3
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rp = 2 * wave + (u >> 1), xb = u & 1;
            const int y = y0 + 2 * rp + s_row;
            const int x = x0 + xb * 16 + 4 * lk;
            if (y >= p.H || x >= p.W) continue;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int co = nb * 8 + col;
                f32x4 v = acc[u][nb];
                if (MODE == MODE_FWD) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float o = v[r] * e_scale[nb] + e_shift[nb];
                        v[r] = p.relu ? fmaxf(o, 0.f) : o;
                    }
                    float* op = p.out.ptr + b * p.out.bstride + co * p.out.cstride + (int64_t)y * p.out.rstride + x;
                    if (x + 3 < p.W && ((p.out.rstride & 3) == 0) && ((reinterpret_cast<uintptr_t>(op) & 15) == 0)) {
                        *reinterpret_cast<f32x4*>(op) = v;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (x + r < p.W) op[r] = v[r];
                    }
                } else if (!p.pool) {
                    float* op = p.out.ptr + b * p.out.bstride + co * p.out.cstride + (int64_t)y * p.out.rstride + x;
                    const float* ap = p.act ? p.act + b * p.act_bstride + co * p.act_cstride + (int64_t)y * p.act_rstride + x
                                            : nullptr;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (x + r < p.W) {
                            float o = v[r];
                            if (ap) o = ap[r] > 0.f ? o * e_scale[nb] : 0.f;
                            if (p.accumulate) o += op[r];
                            op[r] = o;
                        }
                    }
                } else {
                    // MaxPool2d(2) backward: (y,x) is a pooled coordinate; route to the first arg-max of the window.
                    const float* a0 = p.act + b * p.act_bstride + co * p.act_cstride + (int64_t)(2 * y) * p.act_rstride;
                    const float* a1 = a0 + p.act_rstride;
                    float* o0 = p.out.ptr + b * p.out.bstride + co * p.out.cstride + (int64_t)(2 * y) * p.out.rstride;
                    float* o1 = o0 + p.out.rstride;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int xx = x + r;
                        if (xx < p.W) {
                            const float w00 = a0[2 * xx], w01 = a0[2 * xx + 1], w10 = a1[2 * xx], w11 = a1[2 * xx + 1];
                            int am = 0;
                            float m = w00;
                            if (w01 > m) { m = w01; am = 1; }
                            if (w10 > m) { m = w10; am = 2; }
                            if (w11 > m) { m = w11; am = 3; }
                            const float g = m > 0.f ? v[r] * e_scale[nb] : 0.f;
                            o0[2 * xx] += am == 0 ? g : 0.f;
                            o0[2 * xx + 1] += am == 1 ? g : 0.f;
                            o1[2 * xx] += am == 2 ? g : 0.f;
                            o1[2 * xx + 1] += am == 3 ? g : 0.f;
                        }
                    }
                }
            }
        }
    }
}

template <int CIN, int COUT, int MODE>
int launch_conv(ConvArgs& p, hipStream_t stream) {
    constexpr int CHUNK = CIN < 16 ? CIN : 16;
    p.tiles_x = (p.W + TW - 1) / TW;
    p.tiles_y = (p.H + TH - 1) / TH;
    p.ntiles = p.B * p.tiles_x * p.tiles_y;
    if (p.ntiles <= 0) return 0;
    const size_t lds = (size_t)CHUNK * CS * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_mfma_kernel<CIN, COUT, MODE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int max_grid = 256 * 4;
    const int grid = p.ntiles < max_grid ? p.ntiles : max_grid;
    hipLaunchKernelGGL((conv3x3_mfma_kernel<CIN, COUT, MODE>), dim3(grid), dim3(256), lds, stream, p);
    PC_CHECK_LAUNCH();
    return 0;
}

template <int MODE>
int dispatch_conv(ConvArgs& p, int Cin, int Cout, hipStream_t stream) {
#define PC_CASE(ci, co) \
    if (Cin == ci && Cout == co) return launch_conv<ci, co, MODE>(p, stream);
    PC_CASE(2, 8) PC_CASE(4, 8) PC_CASE(8, 8) PC_CASE(16, 8) PC_CASE(32, 8) PC_CASE(8, 16) PC_CASE(16, 16)
#undef PC_CASE
    return PC_EINVAL;
}

pc_src empty_src() {
    pc_src s{};
    return s;
}

}  // namespace

extern "C" int pc_conv3x3_bn_relu_fwd(const pc_src* a, const pc_src* b, const float* w, const pc_bn* bn, int relu,
                                      const pc_dst* out, int B, int H, int W, int Cin, int Cout, void* stream) {
    if (!a || !w || !bn || !out) return PC_EINVAL;
    ConvArgs p{};
    p.a = *a;
    p.b = b ? *b : empty_src();
    if (p.a.C + p.b.C != Cin) return PC_EINVAL;
    p.w = w;
    p.w_co_stride = Cin * 9;
    p.w_ci_stride = 9;
    p.w_flip = 0;
    p.bn = *bn;
    p.relu = relu;
    p.out = *out;
    p.B = B; p.H = H; p.W = W;
    return dispatch_conv<MODE_FWD>(p, Cin, Cout, (hipStream_t)stream);
}

extern "C" int pc_conv3x3_dgrad(const pc_src* g, const float* w, int Cin_total, int c0, int Cn,
                                const pc_src* act, const pc_bn* act_bn, int pool, int accumulate,
                                const pc_dst* out, int B, int H, int W, int Cg, void* stream) {
    if (!g || !w || !out || g->C != Cg) return PC_EINVAL;
    if (pool && !act) return PC_EINVAL;
    ConvArgs p{};
    p.a = *g;
    p.b = empty_src();
    // forward weight w[cg][Cin_total][3][3]; as a conv over g producing input channel (c0 + co):
    //   weight(out = co, in = cg, tap) = w[cg][c0 + co][8 - tap]
    p.w = w + (int64_t)c0 * 9;
    p.w_co_stride = 9;
    p.w_ci_stride = Cin_total * 9;
    p.w_flip = 1;
    if (act) {
        if (!act_bn) return PC_EINVAL;
        p.bn = *act_bn;
        p.act = act->ptr;
        p.act_bstride = act->bstride;
        p.act_cstride = act->cstride;
        p.act_rstride = act->rstride;
        p.outH = act->H;
        p.outW = act->W;
    }
    p.pool = pool;
    p.accumulate = accumulate;
    p.out = *out;
    p.B = B; p.H = H; p.W = W;
    return dispatch_conv<MODE_DGRAD>(p, Cg, Cn, (hipStream_t)stream);
}
