// level2.hip -- the whole 32 x 32 level of a U-Net stream in ONE launch (fp32, PC_PREC_FP32).
//
// Replaces, for a pooled 16-channel 32 x 32 map (the conv domain 128 x 128 of 100 x 100 tiles after two MaxPool2d(2)):
//     Down.mpconv[1] = DoubleConv(16, 16)   (reference model/DDA_model/utils/networks.py:253-271,284-295)
//     Up.up          = ConvTranspose2d(16, 16, 2, stride 2)   (networks.py:302,306)
// i.e. conv3x3+BN+ReLU -> conv3x3+BN+ReLU -> convT 2x2, three launches of the layer-by-layer path.  At this resolution a WHOLE
// per-tile map is 16 ch x 32 x 32 x 4 B = 64 KB: one workgroup owns one (tile, network-stream) problem, keeps the input and
// both intermediates in LDS (no halo recompute exists at whole-tile residency, nothing between the three layers touches
// HBM) and writes only what the backward pass needs (c1, c2: trainable networks only) plus the up-sampled 64 x 64 output.
// 64 tiles x 4 problems = 256 workgroups = one per CU; 8 waves per workgroup (two per SIMD).
//
// LDS image of a map: [16 ch][34 rows][36 cols] fp32 with the data at rows 1..32, cols 4..35 (16-byte aligned rows for the
// staging / epilogue stores), zero halo rows 0 and 33, zero cols 0..3 (col 3 = left halo; col 36 = col 0 of the next row =
// right halo), channel stride 1232 == 16 (mod 32) so that the two taps a 32-lane half reads together never collide.
// GEMM mapping of a conv (as the N16 mapping of conv3x3.hip): M = 16 x of one row, N = 16 output channels, K = 4 taps of an
// 8-channel chunk's 72; 18 MFMAs (v_mfma_f32_16x16x4_f32) per chunk and 16-px unit, every K-slot a tap.  The K-slot -> tap
// table pairs, inside each 32-lane half of the A-operand read, either the dx = 0 / 1 taps of one (channel, row) (addresses
// one apart: the overlapping lanes read the same dwords) or the dx = 2 taps of two neighbouring channels (a channel stride
// apart: disjoint banks).
#include "common.h"

namespace {

constexpr int L2_RS = 36;                       // row stride
constexpr int L2_CS = 34 * L2_RS + 8;           // channel stride: 1232 == 16 (mod 32)
constexpr int L2_WS = 73;                       // weight image row stride (one 8-channel chunk: 72 K-slots), odd
constexpr int L2_BUF = 16 * L2_CS;              // floats per map image
constexpr size_t L2_LDS = (size_t)(2 * L2_BUF + 16 * L2_WS) * sizeof(float);

struct L2Prob {
    const float* x; int64_t x_bs, x_cs; int x_rs;          // pooled input (B,16,32,32)
    const float* w1; const float* w2; const float* wt; const float* bt;
    pc_bn bn1, bn2;
    float* c1; int64_t c1_bs, c1_cs; int c1_rs;            // NULL = not saved
    float* c2; int64_t c2_bs, c2_cs; int c2_rs;
    float* u2; int64_t u2_bs, u2_cs; int u2_rs;            // (B,16,64,64)
};
struct L2Args { L2Prob pr[PC_MAX_GROUP]; };

// K-slot k = 4 m + lk of a chunk -> tap (channel ci of the chunk, dy, dx)
__device__ __forceinline__ void l2_tap(int k, int& ci, int& dy, int& dx) {
    const int pi = k >> 1, e = k & 1;           // pair index (two per step), element of the pair
    if (pi < 24) { ci = pi / 3; dy = pi - 3 * ci; dx = e; }
    else { const int q = (pi - 24) / 3; dy = (pi - 24) - 3 * q; ci = 2 * q + e; dx = 2; }
}

__global__ __launch_bounds__(512) void level2_fwd_kernel(const L2Args args) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const L2Prob& q = args.pr[blockIdx.y];
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    float* const bufA = lds;
    float* const bufB = lds + L2_BUF;
    float* const wimg = lds + 2 * L2_BUF;

    // ---- weights of stage (layer, chunk) into registers: 16 x 72 = 1152 values, 3 per thread (K-slot order)
    auto load_w = [&](const float* w, int ch, float (&wr)[3]) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int e = tid + 512 * i;
            float v = 0.f;
            if (e < 16 * 72) {
                const int co = e / 72, k = e - 72 * co;
                int ci, dy, dx;
                l2_tap(k, ci, dy, dx);
                v = w[(co * 16 + ch * 8 + ci) * 9 + dy * 3 + dx];
            }
            wr[i] = v;
        }
    };
    auto store_w = [&](const float (&wr)[3]) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int e = tid + 512 * i;
            if (e < 16 * 72) {
                const int co = e / 72, k = e - 72 * co;
                wimg[co * L2_WS + k] = wr[i];
            }
        }
    };
    float wr[3];
    load_w(q.w1, 0, wr);

    // ---- input tile -> bufA (8 x 16-byte pieces per thread), halo / pad zeros of both images
    {
        f32x4 t[8];
        const float* xp = q.x + b * q.x_bs;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + 512 * i, ch = idx >> 8, row = (idx >> 3) & 31, seg = idx & 7;
            t[i] = *reinterpret_cast<const f32x4*>(xp + ch * q.x_cs + (int64_t)row * q.x_rs + 4 * seg);
        }
        const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
        // per channel and image: row 0 (9 pieces), row 33 + pad (11 pieces), cols 0..3 of rows 1..32 (32 pieces) = 52 pieces
        for (int e = tid; e < 2 * 16 * 52; e += 512) {
            const int img = e / (16 * 52), r = e - img * 16 * 52, ch = r / 52, pc = r - ch * 52;
            float* base = lds + img * L2_BUF + ch * L2_CS;
            int off;
            if (pc < 9) off = 4 * pc;
            else if (pc < 20) off = 33 * L2_RS + 4 * (pc - 9);
            else off = (pc - 20 + 1) * L2_RS;
            *reinterpret_cast<f32x4*>(base + off) = z;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + 512 * i, ch = idx >> 8, row = (idx >> 3) & 31, seg = idx & 7;
            *reinterpret_cast<f32x4*>(bufA + ch * L2_CS + (row + 1) * L2_RS + 4 + 4 * seg) = t[i];
        }
    }
    // transposed-conv B fragments (k = ci = 4 s + lk, n = nb * 16 + li = (co, a, b)) and bias: registers for the last phase
    float bwt[4][4], bint[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        bint[nb] = q.bt ? q.bt[(nb * 16 + li) >> 2] : 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) bwt[s][nb] = q.wt[(4 * s + lk) * 64 + nb * 16 + li];
    }

    // per-lane A offsets of the 18 steps (floats, relative to the image): tap + lane x + left-halo column + the wave's rows
    int aoff[18];
#pragma unroll
    for (int m = 0; m < 18; ++m) {
        int ci, dy, dx;
        l2_tap(4 * m + lk, ci, dy, dx);
        aoff[m] = ci * L2_CS + (4 * wave + dy) * L2_RS + dx + 3 + li;
    }
    const float* const wlane = wimg + li * L2_WS + lk;

    // ---- conv + BN + ReLU: src image -> dst image (+ optional global copy)
    auto conv = [&](const float* src, float* dstimg, const float* w, const float* wnext, int wnext_ch, const pc_bn& bn,
                    float* gout, int64_t g_bs, int64_t g_cs, int g_rs) {
        float e_scale, e_shift;
        pc_bn_fold(bn, li, e_scale, e_shift);
        f32x4 acc[4][2];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int h = 0; h < 2; ++h) acc[r][h] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int ch = 0; ch < 2; ++ch) {
            __syncthreads();                       // readers of the previous weight chunk / writers of the source image are done
            store_w(wr);
            if (ch == 0) load_w(w, 1, wr);         // next stage's weights: in flight during this chunk's MFMAs
            else if (wnext) load_w(wnext, wnext_ch, wr);
            __syncthreads();
            const float* s0 = src + ch * 8 * L2_CS;
#pragma unroll
            for (int m = 0; m < 18; ++m) {
                const float bwv = wlane[4 * m];
                float av[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) av[u] = s0[aoff[m] + (u >> 1) * L2_RS + (u & 1) * 16];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    acc[u >> 1][u & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bwv, acc[u >> 1][u & 1], 0, 0, 0);
            }
        }
        // epilogue: lane = channel li, 4 consecutive x = 16 h + 4 lk .. + 3 of row 4 wave + r
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[r][h][e] * e_scale + e_shift, 0.f);
                const int y = 4 * wave + r, x = 16 * h + 4 * lk;
                *reinterpret_cast<f32x4*>(dstimg + li * L2_CS + (y + 1) * L2_RS + 4 + x) = v;
                if (gout) *reinterpret_cast<f32x4*>(gout + b * g_bs + li * g_cs + (int64_t)y * g_rs + x) = v;
            }
    };
    conv(bufA, bufB, q.w1, q.w2, 0, q.bn1, q.c1, q.c1_bs, q.c1_cs, q.c1_rs);
    conv(bufB, bufA, q.w2, nullptr, 0, q.bn2, q.c2, q.c2_bs, q.c2_cs, q.c2_rs);
    __syncthreads();                               // c2 complete in bufA

    // ---- ConvTranspose2d(16, 16, 2, 2): out[co][2i+a][2j+b] = bias[co] + sum_ci c2[ci][i][j] * wt[ci][co][a][b]
    // M = 16 x of row i, K = ci (4 steps), N = (co, a, b) (4 blocks); lane pairs (b = 0, 1) swap halves so that every lane
    // stores 4 consecutive output x (one 16-byte store)
#pragma unroll 1
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * wave + r;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float av[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) av[s] = bufA[(4 * s + lk) * L2_CS + (i + 1) * L2_RS + 4 + 16 * h + li];
            f32x4 acc[4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[nb] = f32x4{bint[nb], bint[nb], bint[nb], bint[nb]};
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bwt[s][nb], acc[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const int ng = nb * 16 + li;
                const int co = ng >> 2, a = (ng >> 1) & 1, bb = ng & 1;
                const f32x4 mine = acc[nb];
                f32x4 other;
#pragma unroll
                for (int e = 0; e < 4; ++e) other[e] = __shfl_xor(mine[e], 1);
                const f32x4 v = bb == 0 ? f32x4{mine[0], other[0], mine[1], other[1]} : f32x4{other[2], mine[2], other[3], mine[3]};
                const int jb = 16 * h + 4 * lk;
                *reinterpret_cast<f32x4*>(q.u2 + b * q.u2_bs + co * q.u2_cs + (int64_t)(2 * i + a) * q.u2_rs + 2 * jb + 4 * bb) = v;
            }
        }
    }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
bool plane_ok(const void* p, int64_t bs, int64_t cs, int rs, int xs, int dtype) {
    return p && dtype == PC_F32 && xs <= 1 && aligned16(p) && bs % 4 == 0 && cs % 4 == 0 && rs % 4 == 0;
}

}  // namespace

extern "C" int pc_level2_fwd_ok(const pc_src* x, const pc_dst* u2) {
    if (g_pc_precision != PC_PREC_FP32 || !x || !u2) return 0;
    if (x->C != 16 || x->H != 32 || x->W != 32 || x->mode != PC_SRC_DIRECT || x->oy || x->ox) return 0;
    return plane_ok(x->ptr, x->bstride, x->cstride, x->rstride, x->xstride, x->dtype) &&
           plane_ok(u2->ptr, u2->bstride, u2->cstride, u2->rstride, u2->xstride, u2->dtype);
}

extern "C" int pc_level2_fwd_group(int n, const pc_level2_fwd_desc* d, int B, void* stream) {
    if (n < 1 || n > PC_MAX_GROUP || B < 1 || !d) return PC_EINVAL;
    L2Args a;
    for (int i = 0; i < n; ++i) {
        const pc_level2_fwd_desc& s = d[i];
        if (!s.x || !s.u2 || !s.w1 || !s.w2 || !s.wt || !s.bn1 || !s.bn2 || !pc_level2_fwd_ok(s.x, s.u2)) return PC_EINVAL;
        L2Prob& p = a.pr[i];
        p.x = s.x->ptr; p.x_bs = s.x->bstride; p.x_cs = s.x->cstride; p.x_rs = s.x->rstride;
        p.w1 = s.w1; p.w2 = s.w2; p.wt = s.wt; p.bt = s.bt; p.bn1 = *s.bn1; p.bn2 = *s.bn2;
        p.c1 = p.c2 = nullptr; p.c1_bs = p.c1_cs = p.c2_bs = p.c2_cs = 0; p.c1_rs = p.c2_rs = 0;
        if (s.c1) {
            if (!plane_ok(s.c1->ptr, s.c1->bstride, s.c1->cstride, s.c1->rstride, s.c1->xstride, s.c1->dtype)) return PC_EINVAL;
            p.c1 = s.c1->ptr; p.c1_bs = s.c1->bstride; p.c1_cs = s.c1->cstride; p.c1_rs = s.c1->rstride;
        }
        if (s.c2) {
            if (!plane_ok(s.c2->ptr, s.c2->bstride, s.c2->cstride, s.c2->rstride, s.c2->xstride, s.c2->dtype)) return PC_EINVAL;
            p.c2 = s.c2->ptr; p.c2_bs = s.c2->bstride; p.c2_cs = s.c2->cstride; p.c2_rs = s.c2->rstride;
        }
        p.u2 = s.u2->ptr; p.u2_bs = s.u2->bstride; p.u2_cs = s.u2->cstride; p.u2_rs = s.u2->rstride;
    }
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&level2_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)L2_LDS);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    hipLaunchKernelGGL(level2_fwd_kernel, dim3(B, n), dim3(512), L2_LDS, (hipStream_t)stream, a);
    PC_CHECK_LAUNCH();
    return 0;
}
