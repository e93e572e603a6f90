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
#include <type_traits>

static long long* g_l2_ts = nullptr;       // debug: phase timestamps of workgroup (0, 0) (tools/level2_phases.py)
extern "C" void pc_debug_level2_ts(void* buf) { g_l2_ts = (long long*)buf; }

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
    if (!q.u2) return;                             // nobody reads the up-sampled map of this problem (wave-uniform: whole workgroup)
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
                for (int e = 0; e < 4; ++e) other[e] = pc_lane_xor1(mine[e]);
                const f32x4 v = bb == 0 ? f32x4{mine[0], other[0], mine[1], other[1]} : f32x4{other[2], mine[2], other[3], mine[3]};
                const int jb = 16 * h + 4 * lk;
                *reinterpret_cast<f32x4*>(q.u2 + b * q.u2_bs + co * q.u2_cs + (int64_t)(2 * i + a) * q.u2_rs + 2 * jb + 4 * bb) = v;
            }
        }
    }
}


// =====================================================================================================================
// Backward of the two convolutions of the level (down2's DoubleConv) in ONE launch, same whole-tile residency:
//     given  G2 = dL/d(conv2 output)  (already multiplied by relu'(c2) * bn2-scale: the transposed-conv backward writes it),
//     dW2[co][ci][t] = sum_px G2[co][px] c1[ci][px + t]        db2[co] = sum_px G2[co][px]
//     G1 = relu'(c1) * bn1-scale * conv3x3(G2, w2 transposed / flipped)                  (never leaves LDS)
//     dW1[co][ci][t] = sum_px G1[co][px] x[ci][px + t]         db1[co] = sum_px G1[co][px]       (x = the pooled input)
//     Gp = conv3x3(G1, w1 transposed / flipped)                                           (registers only)
//     MaxPool2d(2) backward: G_b2[c][first arg-max of the 2x2 window of b2] += relu'(b2 max) * bn-scale(d1b) * Gp
// Replaces two weight-gradient and two data-gradient launches of the layer-by-layer path (the second one with the pooling
// scatter); G1 and Gp are never written to HBM.  Two LDS images: {G2, later x} and {c1, overwritten in place by G1}.
// Image channel stride 1252 == 4 (mod 32): the data-gradient reads (lanes = 16 consecutive x, tap pairs = dx 0 / 1 of one
// (channel, row) or dx 2 of channels ci / ci + 4: 4 * 1252 == 16 mod 32) are conflict-free, the weight-gradient reads
// (lanes = 16 channels) are 2-way conflicted -- 40 LDS cycles per 9 MFMAs, far below the matrix time.
// Weight-gradient mapping: M = co, N = ci, K = 4 consecutive pixels of a row; per 4-pixel group ONE A read (G) feeds the 9
// tap MFMAs, whose B operands are the 9 shifted reads of x.  Every wave writes its own partial (8 per tile) in the layout of
// pc_wgrad_reduce_batch (kind 0, Cin = Cout = 16): no cross-wave reduction.
// Work split: TWO workgroups per tile (rows 0..15 and 16..31: 128 tile-problems would fill only half of the 256 CUs).  Each
// recomputes the one-row halo of G1 it needs (18 rows instead of 16: the four extra 16-px units go to waves 0..3) and stages
// 20 rows of G2 / 18 rows of c1 and x; its weight-gradient sums cover its own 16 rows.
constexpr int B2_ROWS = 22;                      // image rows: global rows r0 - 3 .. r0 + 18 (r0 = first row of the half)
constexpr int B2_CS = B2_ROWS * L2_RS + 12;      // 804 == 4 (mod 32)
constexpr int B2_BUF = 16 * B2_CS;
constexpr int B2_SCR = 4 * (9 * 64 * 4 + 16);   // reduction scratch: 4 wave slots x (9 taps x 64 lanes x 4 + 16 bias sums) floats
constexpr size_t B2_LDS = (size_t)(2 * B2_BUF + B2_SCR) * sizeof(float);
constexpr int B2_EC = 16 * 16 * 9 + 16;          // floats of one partial (WgradCfg::EC of conv3x3_wgrad.hip)
constexpr int B2_PARTIALS = 2;                   // partials per tile and layer: one per half-tile workgroup

struct B2Prob {
    const float* g2; int64_t g2_bs, g2_cs; int g2_rs;
    const float* c1; int64_t c1_bs, c1_cs; int c1_rs;
    const float* x;  int64_t x_bs, x_cs; int x_rs;
    const float* w1; const float* w2;
    pc_bn bn1;                                   // BN of conv1 (scale of relu'(c1))
    const float* act; int64_t a_bs, a_cs; int a_rs;     // b2: full-resolution activations that were pooled (B,16,64,64)
    pc_bn act_bn;                                // BN of the layer that produced b2
    float* out; int64_t o_bs, o_cs; int o_rs;    // G_b2 (B,16,64,64), accumulated
    float* ws2; float* ws1;                      // partials: [B * B2_PARTIALS][B2_EC]
};
struct B2Args { B2Prob pr[PC_MAX_GROUP]; long long* ts; };

// K-slot k = 4 m + lk (m = 0..35) -> tap (input channel ci 0..15, dy, dx) of the data-gradient convolutions
__device__ __forceinline__ void b2_tap(int k, int& ci, int& dy, int& dx) {
    const int pi = k >> 1, e = k & 1;
    if (pi < 48) { ci = pi / 3; dy = pi - 3 * ci; dx = e; }
    else { const int q = pi - 48, cq = q / 3; dy = q - 3 * cq; ci = (cq >> 2) * 8 + (cq & 3) + 4 * e; dx = 2; }
}

__global__ __launch_bounds__(512) void level2_bwd_kernel(const B2Args args) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const B2Prob& q = args.pr[blockIdx.y];
    const int b = blockIdx.x >> 1, hf = blockIdx.x & 1, r0 = 16 * hf;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    float* const img0 = lds;                    // G2, later x
    float* const img1 = lds + B2_BUF;           // c1, later G1
    auto stamp = [&](int i) {
        if (args.ts && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) args.ts[i] = wall_clock64();
    };
    stamp(0);

    // rows [lo, lo + NR) of a (B,16,32,32) tensor -> registers (zeros outside the tile); image row of global row gr = gr - r0 + 3
    auto fetch = [&](const float* src, int64_t bs, int64_t cs, int rs, int lo, int nr, f32x4 (&t)[5]) {
        const float* sp = src + b * bs;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int idx = tid + 512 * i, ch = idx / (nr * 8), rem = idx - ch * nr * 8, row = rem >> 3, seg = rem & 7;
            const int gr = lo + row;
            const bool ok = ch < 16 && (unsigned)gr < 32u;
            t[i] = ok ? *reinterpret_cast<const f32x4*>(sp + ch * cs + (int64_t)gr * rs + 4 * seg) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto commit = [&](float* img, int lo, int nr, const f32x4 (&t)[5]) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int idx = tid + 512 * i, ch = idx / (nr * 8), rem = idx - ch * nr * 8, row = rem >> 3, seg = rem & 7;
            if (ch < 16) *reinterpret_cast<f32x4*>(img + ch * B2_CS + (lo + row - r0 + 3) * L2_RS + 4 + 4 * seg) = t[i];
        }
    };
    // data-gradient weights of layer `w` ([co][ci][3][3]): B[k = tap (input channel co_in, dy, dx)][n = li = output channel ci_out]
    //   = w[co_in][ci_out][8 - (3 dy + dx)]
    auto load_wT = [&](const float* w, float (&wv)[36]) {
#pragma unroll
        for (int m = 0; m < 36; ++m) {
            int ci, dy, dx;
            b2_tap(4 * m + lk, ci, dy, dx);
            wv[m] = w[(ci * 16 + li) * 9 + 8 - (3 * dy + dx)];
        }
    };
    f32x4 tg[5], tc[5];
    fetch(q.g2, q.g2_bs, q.g2_cs, q.g2_rs, r0 - 2, 20, tg);
    fetch(q.c1, q.c1_bs, q.c1_cs, q.c1_rs, r0 - 1, 18, tc);
    float wv[36];
    load_wT(q.w2, wv);
    {
        const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};          // everything that is not staged below must read as zero
        for (int e = tid; e < 2 * B2_BUF / 4; e += 512) *reinterpret_cast<f32x4*>(lds + 4 * e) = z;
    }
    __syncthreads();
    commit(img0, r0 - 2, 20, tg);
    commit(img1, r0 - 1, 18, tc);
    // tap offsets (floats) of the 36 steps relative to an output pixel's image position: channel, row dy - 1, column dx - 1
    int aoff[36];
#pragma unroll
    for (int m = 0; m < 36; ++m) {
        int ci, dy, dx;
        b2_tap(4 * m + lk, ci, dy, dx);
        aoff[m] = ci * B2_CS + (dy - 1) * L2_RS + dx + 3 + li;
    }
    __syncthreads();
    stamp(1);

    // ---- weight gradient of one layer over this wave's 2 rows (image rows 2 wave + 3, + 4) -> its own partial in ws
    auto wgrad = [&](const float* gimg, const float* ximg, float* ws) {
        f32x4 acc[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        float dbs = 0.f;
        const float* gl = gimg + li * B2_CS + (2 * wave + 3) * L2_RS + 4 + lk;        // A: (co = li, px + lk)
        const float* xl = ximg + li * B2_CS + (2 * wave + 2) * L2_RS + 3 + lk;        // B: (ci = li, row + dy - 1, px + lk + dx - 1)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
#pragma unroll
            for (int g4 = 0; g4 < 8; ++g4) {
                const float a = gl[r * L2_RS + 4 * g4];
                dbs += a;
                float bx[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) bx[t] = xl[(r + t / 3) * L2_RS + 4 * g4 + (t % 3)];
#pragma unroll
                for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bx[t], acc[t], 0, 0, 0);
            }
        }
        // the 8 waves' sums -> ONE partial per workgroup, in a fixed order (deterministic): waves 4..7 park theirs in the scratch
        // slots, waves 0..3 add them and park the result, then all threads add the four slots and write the partial
        // (D[co = 4 lk + e][ci = li] of tap t sits at slot[(t * 64 + lane) * 4 + e]; bias sums of lanes lk == 0 at slot[2304 + co])
        dbs = pc_xor16_sum(dbs);
        dbs = pc_xor32_sum(dbs);
        float* const scr = lds + 2 * B2_BUF;
        constexpr int SLOT = 9 * 64 * 4 + 16;
        float* const mine = scr + (wave & 3) * SLOT;
        if (wave >= 4) {
#pragma unroll
            for (int t = 0; t < 9; ++t) *reinterpret_cast<f32x4*>(mine + (t * 64 + lane) * 4) = acc[t];
            if (lk == 0) mine[2304 + li] = dbs;
        }
        __syncthreads();
        if (wave < 4) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const f32x4 o = *reinterpret_cast<const f32x4*>(mine + (t * 64 + lane) * 4);
                *reinterpret_cast<f32x4*>(mine + (t * 64 + lane) * 4) = acc[t] + o;
            }
            if (lk == 0) mine[2304 + li] += dbs;
        }
        __syncthreads();
        float* pw = ws + (int64_t)(b * 2 + hf) * B2_EC;
        for (int o = tid; o < B2_EC; o += 512) {
            float v;
            if (o < 2304) {
                const int co = o / 144, rem = o - co * 144, ci = rem / 9, t = rem - ci * 9;
                const int idx = (t * 64 + (co >> 2) * 16 + ci) * 4 + (co & 3);
                v = (scr[idx] + scr[SLOT + idx]) + (scr[2 * SLOT + idx] + scr[3 * SLOT + idx]);
            } else {
                const int idx = 2304 + (o - 2304);
                v = (scr[idx] + scr[SLOT + idx]) + (scr[2 * SLOT + idx] + scr[3 * SLOT + idx]);
            }
            pw[o] = v;
        }
        // (the next use of the scratch is a whole phase and at least one barrier away)
    };
    // ---- data-gradient convolution with the register weights for NU units; unit u: output pixels (image row urow[u], x = ux[u] + M index)
    // acc[u] = (channel li, x = ux + 4 lk + e)
    auto dgrad = [&](const float* img, auto& acc, const int (&uoff)[5], auto NU) {
        constexpr int nu = decltype(NU)::value;
#pragma unroll
        for (int u = 0; u < nu; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 36; ++m) {
            float av[nu];
#pragma unroll
            for (int u = 0; u < nu; ++u) av[u] = img[aoff[m] + uoff[u]];
#pragma unroll
            for (int u = 0; u < nu; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], wv[m], acc[u], 0, 0, 0);
        }
    };
    // units of this wave: its two rows x two halves (image rows 2 wave + 3, + 4); the fifth: one half of a G1 halo row (image row 2
    // or 19, waves 0..3; the other waves repeat their first unit and drop it)
    int uoff[5], urow[5], ucol[5];
#pragma unroll
    for (int u = 0; u < 4; ++u) { urow[u] = 2 * wave + 3 + (u >> 1); ucol[u] = 16 * (u & 1); }
    urow[4] = wave < 2 ? 2 : (wave < 4 ? 19 : urow[0]);
    ucol[4] = wave < 4 ? 16 * (wave & 1) : ucol[0];
#pragma unroll
    for (int u = 0; u < 5; ++u) uoff[u] = urow[u] * L2_RS + ucol[u];

    wgrad(img0, img1, q.ws2);                      // dW2, db2: G2 x c1
    stamp(2);
    f32x4 tx[5];
    fetch(q.x, q.x_bs, q.x_cs, q.x_rs, r0 - 1, 18, tx);      // the pooled input: in flight during the data gradient below
    __syncthreads();                               // every wave has finished reading c1's neighbour rows
    stamp(3);
    {
        f32x4 acc[5];
        dgrad(img0, acc, uoff, std::integral_constant<int, 5>());      // conv(G2, w2^T) on 18 rows
        float sc, sh;
        pc_bn_fold(q.bn1, li, sc, sh);
        (void)sh;
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            if (u == 4 && wave >= 4) continue;
            float* p1 = img1 + li * B2_CS + urow[u] * L2_RS + 4 + ucol[u] + 4 * lk;
            const f32x4 c = *reinterpret_cast<const f32x4*>(p1);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = c[e] > 0.f ? acc[u][e] * sc : 0.f;
            *reinterpret_cast<f32x4*>(p1) = v;      // G1 over c1, same positions (rows outside the tile: c1 = 0 -> G1 = 0)
        }
    }
    stamp(4);
    load_wT(q.w1, wv);                             // in flight during the weight gradient below
    __syncthreads();                               // G1 complete, G2 dead
    {
        const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};          // G2 rows r0 - 2 and r0 + 17 are not overwritten by x: clear them
        for (int e = tid; e < 16 * 2 * 8; e += 512) {
            const int ch = e >> 4, rr = (e >> 3) & 1, seg = e & 7;
            *reinterpret_cast<f32x4*>(img0 + ch * B2_CS + (rr ? 20 : 1) * L2_RS + 4 + 4 * seg) = z;
        }
    }
    commit(img0, r0 - 1, 18, tx);
    __syncthreads();
    stamp(5);
    wgrad(img1, img0, q.ws1);                      // dW1, db1: G1 x pooled input
    stamp(6);
    {
        // MaxPool2d(2) backward: the lane's four pooled pixels cover 2 rows x 8 full-resolution pixels.  Its loads (the pooled
        // activations and the gradient accumulated so far: 100 MB over the launch) are issued BEFORE the last data gradient
        float sc, sh;
        pc_bn_fold(q.act_bn, li, sc, sh);
        (void)sh;
        f32x4 A[4][2][2], O[4][2][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int y = r0 + 2 * wave + (u >> 1), x = ucol[u] + 4 * lk;
            const float* a0 = q.act + b * q.a_bs + li * q.a_cs + (int64_t)(2 * y) * q.a_rs + 2 * x;
            const float* o0 = q.out + b * q.o_bs + li * q.o_cs + (int64_t)(2 * y) * q.o_rs + 2 * x;
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    A[u][rr][hh] = *reinterpret_cast<const f32x4*>(a0 + rr * q.a_rs + 4 * hh);
                    O[u][rr][hh] = *reinterpret_cast<const f32x4*>(o0 + rr * q.o_rs + 4 * hh);
                }
        }
        f32x4 acc[5];
        dgrad(img1, acc, uoff, std::integral_constant<int, 4>());      // Gp = conv(G1, w1^T): gradient w.r.t. the pooled map
        stamp(7);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int y = r0 + 2 * wave + (u >> 1), x = ucol[u] + 4 * lk;
            float* o0 = q.out + b * q.o_bs + li * q.o_cs + (int64_t)(2 * y) * q.o_rs + 2 * x;
            const f32x4 v = acc[u];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int hh = e >> 1, c = (e & 1) * 2;
                const float w00 = A[u][0][hh][c], w01 = A[u][0][hh][c + 1], w10 = A[u][1][hh][c], w11 = A[u][1][hh][c + 1];
                int am = 0;
                float mx = w00;
                if (w01 > mx) { mx = w01; am = 1; }
                if (w10 > mx) { mx = w10; am = 2; }
                if (w11 > mx) { mx = w11; am = 3; }
                const float g = mx > 0.f ? v[e] * sc : 0.f;
                O[u][0][hh][c] += am == 0 ? g : 0.f;
                O[u][0][hh][c + 1] += am == 1 ? g : 0.f;
                O[u][1][hh][c] += am == 2 ? g : 0.f;
                O[u][1][hh][c + 1] += am == 3 ? g : 0.f;
            }
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) *reinterpret_cast<f32x4*>(o0 + rr * q.o_rs + 4 * hh) = O[u][rr][hh];
        }
    }
    stamp(8);
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
bool plane_ok(const void* p, int64_t bs, int64_t cs, int rs, int xs, int dtype) {
    return p && dtype == PC_F32 && (xs == 1 || xs == 0) && aligned16(p) && bs % 4 == 0 && cs % 4 == 0 && rs % 4 == 0;
}

}  // namespace

// level2_cl.hip: the channels-last bf16 form of the forward kernel (PC_PREC_BF16)
bool pc_level2_fwd_cl_ok(const pc_src* x, const pc_dst* u2);
int pc_level2_fwd_cl_launch(int n, const pc_level2_fwd_desc* d, int B, hipStream_t stream);
bool pc_level2_bwd_cl_ok(const pc_src* g2, const pc_src* c1, const pc_src* x, const pc_src* act, const pc_dst* out);
int pc_level2_bwd_cl_launch(int n, const pc_level2_bwd_desc* d, int B, int* nwg_out, long long* ts, hipStream_t stream);

extern "C" int pc_level2_fwd_ok(const pc_src* x, const pc_dst* u2) {
    if (g_pc_precision == PC_PREC_BF16) return pc_level2_fwd_cl_ok(x, u2) ? 1 : 0;
    if (g_pc_precision != PC_PREC_FP32 || !x) return 0;
    if (x->C != 16 || x->H != 32 || x->W != 32 || x->mode != PC_SRC_DIRECT || x->oy || x->ox) return 0;
    return plane_ok(x->ptr, x->bstride, x->cstride, x->rstride, x->xstride, x->dtype) &&
           (!u2 || plane_ok(u2->ptr, u2->bstride, u2->cstride, u2->rstride, u2->xstride, u2->dtype));
}

extern "C" int pc_level2_fwd_group(int n, const pc_level2_fwd_desc* d, int B, void* stream) {
    if (n < 1 || n > PC_MAX_GROUP || B < 1 || !d) return PC_EINVAL;
    if (g_pc_precision == PC_PREC_BF16) return pc_level2_fwd_cl_launch(n, d, B, (hipStream_t)stream);
    L2Args a;
    for (int i = 0; i < n; ++i) {
        const pc_level2_fwd_desc& s = d[i];
        if (!s.x || (!s.u2 && !s.c2) || !s.w1 || !s.w2 || !s.wt || !s.bn1 || !s.bn2 || !pc_level2_fwd_ok(s.x, s.u2)) return PC_EINVAL;
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
        p.u2 = nullptr; p.u2_bs = p.u2_cs = 0; p.u2_rs = 0;
        if (s.u2) { p.u2 = s.u2->ptr; p.u2_bs = s.u2->bstride; p.u2_cs = s.u2->cstride; p.u2_rs = s.u2->rstride; }
    }
    static pc_once_per_device once;
    if (once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&level2_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)L2_LDS);
        if (e != hipSuccess) return (int)e;
        once.mark();
    }
    hipLaunchKernelGGL(level2_fwd_kernel, dim3(B, n), dim3(512), L2_LDS, (hipStream_t)stream, a);
    PC_CHECK_LAUNCH();
    return 0;
}

extern "C" int64_t pc_level2_bwd_ws_bytes(int B) { return (int64_t)B * B2_PARTIALS * B2_EC * (int64_t)sizeof(float); }

static bool src_ok(const pc_src* s, int Cc, int Hh, int Ww) {
    return s && s->C == Cc && s->H == Hh && s->W == Ww && s->mode == PC_SRC_DIRECT && !s->oy && !s->ox &&
           plane_ok(s->ptr, s->bstride, s->cstride, s->rstride, s->xstride, s->dtype);
}

extern "C" int pc_level2_bwd_ok(const pc_src* g2, const pc_src* c1, const pc_src* x, const pc_src* act, const pc_dst* out) {
    if (g_pc_precision == PC_PREC_BF16) return pc_level2_bwd_cl_ok(g2, c1, x, act, out) ? 1 : 0;
    if (g_pc_precision != PC_PREC_FP32 || !out) return 0;
    return src_ok(g2, 16, 32, 32) && src_ok(c1, 16, 32, 32) && src_ok(x, 16, 32, 32) && src_ok(act, 16, 64, 64) &&
           plane_ok(out->ptr, out->bstride, out->cstride, out->rstride, out->xstride, out->dtype);
}

extern "C" int pc_level2_bwd_group(int n, const pc_level2_bwd_desc* d, int B, int* nwg_out, void* stream) {
    if (n < 1 || n > PC_MAX_GROUP || B < 1 || !d || !nwg_out) return PC_EINVAL;
    if (g_pc_precision == PC_PREC_BF16) return pc_level2_bwd_cl_launch(n, d, B, nwg_out, g_l2_ts, (hipStream_t)stream);
    B2Args a;
    a.ts = g_l2_ts;
    for (int i = 0; i < n; ++i) {
        const pc_level2_bwd_desc& s = d[i];
        if (!s.w1 || !s.w2 || !s.bn1 || !s.act_bn || !s.ws1 || !s.ws2 || !pc_level2_bwd_ok(s.g2, s.c1, s.x, s.act, s.out)) return PC_EINVAL;
        B2Prob& p = a.pr[i];
        p.g2 = s.g2->ptr; p.g2_bs = s.g2->bstride; p.g2_cs = s.g2->cstride; p.g2_rs = s.g2->rstride;
        p.c1 = s.c1->ptr; p.c1_bs = s.c1->bstride; p.c1_cs = s.c1->cstride; p.c1_rs = s.c1->rstride;
        p.x = s.x->ptr; p.x_bs = s.x->bstride; p.x_cs = s.x->cstride; p.x_rs = s.x->rstride;
        p.w1 = s.w1; p.w2 = s.w2; p.bn1 = *s.bn1; p.act_bn = *s.act_bn;
        p.act = s.act->ptr; p.a_bs = s.act->bstride; p.a_cs = s.act->cstride; p.a_rs = s.act->rstride;
        p.out = s.out->ptr; p.o_bs = s.out->bstride; p.o_cs = s.out->cstride; p.o_rs = s.out->rstride;
        p.ws1 = (float*)s.ws1; p.ws2 = (float*)s.ws2;
    }
    static pc_once_per_device once;
    if (once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&level2_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)B2_LDS);
        if (e != hipSuccess) return (int)e;
        once.mark();
    }
    hipLaunchKernelGGL(level2_bwd_kernel, dim3(2 * B, n), dim3(512), B2_LDS, (hipStream_t)stream, a);
    PC_CHECK_LAUNCH();
    *nwg_out = B2_PARTIALS * B;
    return 0;
}
