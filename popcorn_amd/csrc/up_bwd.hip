// up_bwd.hip -- backward of the up-sampled half of an Up block WITHOUT the up-sampled tensor (fp32, PC_PREC_FP32).
//
// Forward (conv3x3.hip, pc_conv3x3_up_fwd_group): the first conv of an Up block (reference model/DDA_model/utils/networks.py:302-318)
//     y = conv3x3(cat[skip, ConvTranspose2d(C, C, 2, 2)(z)])
// reads the LOW-resolution map z through composed weights: for an output pixel (Y, X) = (2I + pY, 2J + pX)
//     y[co][Y][X] += sum_{ci, v, c3} z[ci][I + v - 1][J + c3 - 1] * Weff[pY][pX][co][ci][v][c3]        (+ the bias through the taps)
// with (v, c3) in a 2 x 2 neighbourhood that depends on the parity.  This file is its backward, from G = dL/dy (8 channels, H x W,
// already times relu' * bn-scale of y's layer) and z, in ONE pass over G:
//     G_z[ci][i][j]  = relu'(z) * bn-scale(z's layer) * sum_{co, r, c} G[co][2i - 1 + r][2j - 1 + c] * Kd[co][ci][r][c]     (4 x 4, stride 2)
//     dWeff[pY][pX][co][ci][v][c3] = sum_{I, J} G[co][2I + pY][2J + pX] * z[ci][I + v - 1][J + c3 - 1]
//     parity / border sums of G (what a constant-one input channel = the transposed conv's bias would contribute)
// followed by a reduction over the workgroups and a small chain-rule kernel that turns dWeff into the gradients of the module's own
// parameters: the conv weight's up-sampled half W[:, Cs:], the transposed conv's weight Wt and bias bt.
// Replaces: the data + weight gradient of the conv's up-sampled column block at full resolution (half of a 4-problem
// conv3x3_bwd launch: 4.8 GFLOP), the transposed conv's backward launch (reads the full-resolution gradient again) and the
// transposed conv's forward (whose output only these two consumed).  Per level the pass moves G once (67 MB at 128 x 128) and
// issues 2.1 GFLOP.
//
// Workgroup = 4 low-res rows x the full width of one tile; LDS: G rows 2 i0 - 1 .. 2 i0 + 8 (10 rows, one-pixel column halo,
// channel stride == 4 mod 32), z rows i0 - 1 .. i0 + 4 (channel stride == 2 mod 32): both MFMA operand reads are conflict-free.
//   data gradient:  M = 16 low-res j, N = ci, K = window column c (4); one MFMA per (co, window row r): 32 per unit
//   weight gradient: M = (pX, co), N = ci (C = 16) or (row tap tv, ci) (C = 8), K = 4 low-res J; one MFMA per (pY, [tv,] column
//                    offset c in {-1, 0, +1}); the (pX, c) combinations that are not taps are dropped by the chain kernel.
#include "common.h"

namespace {

constexpr int UB_RB = 4;                         // low-res rows per workgroup
constexpr int UB_GROWS = 2 * UB_RB + 2;

template <int C> struct UbCfg {
    static constexpr int NACC = C == 16 ? 12 : 6;                 // weight-gradient accumulators (f32x4) per lane
    static constexpr int PART = NACC * 256 + 128;                 // floats of one partial: accumulators + 8 co x 16 border sums
};

struct UbProb {
    const float* g; int64_t g_bs, g_cs; int g_rs;
    const float* z; int64_t z_bs, z_cs; int z_rs;
    pc_bn z_bn;
    float* gz; int64_t gz_bs, gz_cs; int gz_rs;
    const float* wd;          // [32 steps m = co * 4 + r][4 lk = c][16 li = ci]
    float* part;              // [nwg][PART]
};
struct UbArgs { UbProb pr[PC_MAX_GROUP]; int H, W, ntx, nblocks, blocks_per_img; };

// Round 5: the image is cut into column tiles of W (template) full-resolution columns (a.W = image width, a.ntx tiles per row block; the
// last one may be ragged in multiples of 8 columns): the one-pixel column halo of an interior tile comes from the neighbouring columns
// instead of the image border's zeros, so any width >= 16 that is a multiple of 8 is taken (census regions: run_train.py:186-202) -- at
// W == a.W (the 64- and 128-wide levels of the 100 x 100 tiles) nothing changes, bit for bit.
// Persistent workgroups: each walks blocks (tile, 4 low-res rows) with stride gridDim.x, keeps the weight-gradient accumulators and
// the border sums in registers over all of them and writes ONE partial at the end; the G / z rows of block n + 1 are loaded into
// registers while block n computes (the first version staged every block synchronously behind runtime integer divides and wrote a
// partial per block: 118 us at 128 x 128 for 24 us of MFMA issue and 26 us of HBM time).
template <int C, int W>
__global__ __launch_bounds__(256) void up_bwd_kernel(const UbArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // the problem's descriptor and the launch geometry, pinned in scalar registers (common.h: pc_pin; round 6: 14 - 24 re-loads per block)
    UbProb q = a.pr[blockIdx.y];
    q.g = pc_pin_ptr(q.g); pc_pin(q.g_bs); pc_pin(q.g_cs); pc_pin(q.g_rs);
    q.z = pc_pin_ptr(q.z); pc_pin(q.z_bs); pc_pin(q.z_cs); pc_pin(q.z_rs);
    q.gz = pc_pin_ptr(q.gz); pc_pin(q.gz_bs); pc_pin(q.gz_cs); pc_pin(q.gz_rs);
    q.part = pc_pin_ptr(q.part);
    int aH = a.H, aW = a.W, a_ntx = a.ntx, a_nblocks = a.nblocks, a_bpi = a.blocks_per_img, gdim = (int)gridDim.x;
    pc_pin(aH); pc_pin(aW); pc_pin(a_ntx); pc_pin(a_nblocks); pc_pin(a_bpi); pc_pin(gdim);
    constexpr int w = W / 2, W4 = W / 4, w4 = w / 4;
    const int H = aH, h = H >> 1;
    const int WI = aW, wI = WI >> 1;                        // image width (full / low resolution)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    constexpr int GRS = W + 8;                               // G row: data at cols 4 .. W + 3, halo cols 3 and W + 4
    constexpr int GCS = UB_GROWS * GRS + 20;                 // == 4 (mod 32)
    constexpr int ZRS = C == 8 ? (w + 8 + 31) / 32 * 32 + 16 : w + 8;        // data at cols 4 .. w + 3; C = 8: == 16 (mod 32)
    constexpr int ZCS = (UB_RB + 2) * ZRS + (C == 8 ? 2 : ((34 - ((UB_RB + 2) * ZRS) % 32) % 32));        // == 2 (mod 32)
    static_assert(GCS % 32 == 4 && ZCS % 32 == 2, "bank layout");
    float* const gimg = lds;
    float* const zimg = lds + 8 * GCS;
    constexpr int NACC = UbCfg<C>::NACC;
    constexpr int NG = 8 * UB_GROWS * W4 / 256;              // float4 pieces of G per thread and block (10 at W = 128)
    constexpr int NZ = (C * (UB_RB + 2) * w4 + 255) / 256;   // float4 pieces of z per thread and block (3)
    static_assert(8 * UB_GROWS * W4 % 256 == 0, "G pieces");

    // halo columns of both images are never written by the staging: zero once
    for (int e = tid; e < 8 * UB_GROWS * 2; e += 256) {
        const int co = e / (UB_GROWS * 2), rem = e - co * UB_GROWS * 2;
        gimg[co * GCS + (rem >> 1) * GRS + ((rem & 1) ? W + 4 : 3)] = 0.f;
    }
    for (int e = tid; e < C * (UB_RB + 2) * 2; e += 256) {
        const int ci = e / ((UB_RB + 2) * 2), rem = e - ci * (UB_RB + 2) * 2;
        zimg[ci * ZCS + (rem >> 1) * ZRS + ((rem & 1) ? w + 4 : 3)] = 0.f;
    }
    // data-gradient operands: B[k = lk = window column c][n = li = ci] of step m = (co, window row r)
    float wdv[32];
#pragma unroll
    for (int m = 0; m < 32; ++m) wdv[m] = q.wd[(m * 4 + lk) * 16 + li];
    float zsc = 0.f, zsh;
    if (li < C) pc_bn_fold(q.z_bn, li, zsc, zsh);
    (void)zsh;

    f32x4 RG[NG], RZ[NZ];
    float RGh = 0.f, RZh = 0.f;                              // this thread's element of the column halos (interior tile edges)
    // thread -> halo element: G: (co, row, side) for tid < 8 * UB_GROWS * 2 = 160; z: (ci, row, side) for tid < C * (UB_RB + 2) * 2
    const int hg_side = tid & 1, hg_row = (tid >> 1) % UB_GROWS, hg_co = (tid >> 1) / UB_GROWS;
    const int hz_side = tid & 1, hz_row = (tid >> 1) % (UB_RB + 2), hz_ci = (tid >> 1) / (UB_RB + 2);
    auto decode = [&](int blk, int& b, int& i0, int& X0) {
        b = blk / a_bpi;
        const int r = blk - b * a_bpi, rb = r / a_ntx;
        i0 = rb * UB_RB;
        X0 = (r - rb * a_ntx) * W;
    };
    auto fetch = [&](int blk) {
        int b, i0, X0;
        decode(blk, b, i0, X0);
        const int x0 = X0 >> 1;
        const float* gp = q.g + b * q.g_bs;
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const int e = tid + 256 * k, seg = e % W4, pr = e / W4, row = pr % UB_GROWS, co = pr / UB_GROWS;     // compile-time divisors
            const int Y = 2 * i0 - 1 + row;
            const int Yc = Y < 0 ? 0 : (Y >= H ? H - 1 : Y);
            const int X = X0 + 4 * seg;
            RG[k] = *reinterpret_cast<const f32x4*>(gp + co * q.g_cs + (int64_t)Yc * q.g_rs + (X < WI ? X : 0));
        }
        const float* zp = q.z + b * q.z_bs;
#pragma unroll
        for (int k = 0; k < NZ; ++k) {
            const int e = tid + 256 * k, seg = e % w4, pr = e / w4, row = pr % (UB_RB + 2), ci = pr / (UB_RB + 2);
            const int I = i0 - 1 + row;
            const int Ic = I < 0 ? 0 : (I >= h ? h - 1 : I);
            const int x = x0 + 4 * seg;
            RZ[k] = ci < C ? *reinterpret_cast<const f32x4*>(zp + ci * q.z_cs + (int64_t)Ic * q.z_rs + (x < wI ? x : 0)) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (a_ntx > 1) {
            if (tid < 8 * UB_GROWS * 2) {
                const int Y = 2 * i0 - 1 + hg_row, Yc = Y < 0 ? 0 : (Y >= H ? H - 1 : Y);
                const int X = hg_side ? X0 + W : X0 - 1;
                RGh = gp[hg_co * q.g_cs + (int64_t)Yc * q.g_rs + ((unsigned)X < (unsigned)WI ? X : 0)];
            }
            if (tid < C * (UB_RB + 2) * 2) {
                const int I = i0 - 1 + hz_row, Ic = I < 0 ? 0 : (I >= h ? h - 1 : I);
                const int x = hz_side ? x0 + w : x0 - 1;
                RZh = zp[hz_ci * q.z_cs + (int64_t)Ic * q.z_rs + ((unsigned)x < (unsigned)wI ? x : 0)];
            }
        }
    };
    auto commit = [&](int blk) {
        int b, i0, X0;
        decode(blk, b, i0, X0);
        const int x0 = X0 >> 1;
        const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const int e = tid + 256 * k, seg = e % W4, pr = e / W4, row = pr % UB_GROWS, co = pr / UB_GROWS;
            const int Y = 2 * i0 - 1 + row;
            *reinterpret_cast<f32x4*>(gimg + co * GCS + row * GRS + 4 + 4 * seg) = ((unsigned)Y < (unsigned)H && X0 + 4 * seg < WI) ? RG[k] : zero;
        }
#pragma unroll
        for (int k = 0; k < NZ; ++k) {
            const int e = tid + 256 * k, seg = e % w4, pr = e / w4, row = pr % (UB_RB + 2), ci = pr / (UB_RB + 2);
            const int I = i0 - 1 + row;
            if (ci < C) *reinterpret_cast<f32x4*>(zimg + ci * ZCS + row * ZRS + 4 + 4 * seg) = ((unsigned)I < (unsigned)h && x0 + 4 * seg < wI) ? RZ[k] : zero;
        }
        if (a_ntx > 1) {
            if (tid < 8 * UB_GROWS * 2) {
                const int Y = 2 * i0 - 1 + hg_row, X = hg_side ? X0 + W : X0 - 1;
                gimg[hg_co * GCS + hg_row * GRS + (hg_side ? W + 4 : 3)] = ((unsigned)Y < (unsigned)H && (unsigned)X < (unsigned)WI) ? RGh : 0.f;
            }
            if (tid < C * (UB_RB + 2) * 2) {
                const int I = i0 - 1 + hz_row, x = hz_side ? x0 + w : x0 - 1;
                zimg[hz_ci * ZCS + hz_row * ZRS + (hz_side ? w + 4 : 3)] = ((unsigned)I < (unsigned)h && (unsigned)x < (unsigned)wI) ? RZh : 0.f;
            }
        }
    };

    f32x4 wacc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t) wacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // border sums of this thread's (co, image row) pair, accumulated over the blocks: [row parity ^ ... see below]
    float sg[4][4];                                  // [k: parity 0 rows, parity 1 rows, row Y == 0, row Y == H - 1][se, so, gf, gl] (quarter-row partials)
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int c = 0; c < 4; ++c) sg[k][c] = 0.f;

    int blk = blockIdx.x;
    if (blk < a_nblocks) fetch(blk);
    for (; blk < a_nblocks; blk += gdim) {
        int b, i0, X0;
        decode(blk, b, i0, X0);
        const int x0 = X0 >> 1;
        __syncthreads();                             // the previous block's MFMA reads of the images are done
        commit(blk);
        if (blk + gdim < a_nblocks) fetch(blk + gdim);
        __syncthreads();

        // ---- data gradient: units (low-res row il, 16-column block jb), round-robin over the waves
        if (q.gz) {
            constexpr int NBLK = w / 16;
#pragma unroll
            for (int uu = 0; uu < UB_RB * NBLK / 4; ++uu) {
                const int u = wave + 4 * uu, il = u / NBLK, jb = u % NBLK;
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                const float* ga = gimg + (2 * il) * GRS + 2 * (li + 16 * jb) + lk + 3;
#pragma unroll
                for (int m = 0; m < 32; ++m) {
                    const float av = ga[(m >> 2) * GCS + (m & 3) * GRS];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wdv[m], acc, 0, 0, 0);
                }
                // D[j = 4 lk + e][ci = li]
                const int i = i0 + il, j = 16 * jb + 4 * lk;
                if (li < C && i < h && x0 + j < wI) {
                    const float* zr = zimg + li * ZCS + (il + 1) * ZRS + j + 4;
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = zr[e] > 0.f ? acc[e] * zsc : 0.f;
                    *reinterpret_cast<f32x4*>(q.gz + b * q.gz_bs + li * q.gz_cs + (int64_t)i * q.gz_rs + x0 + j) = v;
                }
            }
        }
        // ---- weight gradient over this block's (I, 4-J group) pairs
        {
            constexpr int NGRP = w / 4;
            const int pX = li >> 3, co = li & 7;
#pragma unroll 2
            for (int uu = 0; uu < UB_RB * NGRP / 4; ++uu) {
                const int u = wave + 4 * uu, il = u / NGRP, g4 = u % NGRP;
                if (i0 + il >= h) continue;
#pragma unroll
                for (int pY = 0; pY < 2; ++pY) {
                    const float av = gimg[co * GCS + (2 * il + 1 + pY) * GRS + 2 * (4 * g4 + lk) + pX + 4];
                    if (C == 16) {
#pragma unroll
                        for (int tv = 0; tv < 2; ++tv)
#pragma unroll
                            for (int c = 0; c < 3; ++c) {
                                const float bv = zimg[li * ZCS + (il + tv + pY) * ZRS + 4 * g4 + lk + c + 3];
                                wacc[(pY * 2 + tv) * 3 + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, wacc[(pY * 2 + tv) * 3 + c], 0, 0, 0);
                            }
                    } else {
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const float bv = zimg[(li & 7) * ZCS + (il + (li >> 3) + pY) * ZRS + 4 * g4 + lk + c + 3];
                            wacc[pY * 3 + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, wacc[pY * 3 + c], 0, 0, 0);
                        }
                    }
                }
            }
        }
        // ---- parity / border sums of G over the block's own rows (image rows 1 .. 2 RB): thread = (co, row, quarter of the row)
        {
            const int pair = tid >> 2, qd = tid & 3, co = pair >> 3, row = 1 + (pair & 7);
            const int Y = 2 * i0 - 1 + row;
            if (Y < H) {
                const float* gr = gimg + co * GCS + row * GRS + 4 + qd * W4;
                float se = 0.f, so = 0.f;
#pragma unroll
                for (int x = 0; x < W4; x += 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(gr + x);
                    se += v[0] + v[2];
                    so += v[1] + v[3];
                }
                // first / last column of the IMAGE row (the tile that holds it, the quarter that holds it)
                const int lc = WI - 1 - X0 - qd * W4;                 // image column WI - 1 relative to this quarter
                const float gf = (qd == 0 && X0 == 0) ? gr[0] : 0.f, gl = (unsigned)lc < (unsigned)W4 ? gr[lc] : 0.f;
                const int par = row & 1 ? 0 : 1;     // image row 1 = Y = 2 i0 (even)
                sg[par][0] += se; sg[par][1] += so; sg[par][2] += gf; sg[par][3] += gl;
                if (Y == 0) { sg[2][0] += se; sg[2][1] += so; sg[2][2] += gf; sg[2][3] += gl; }
                if (Y == H - 1) { sg[3][0] += se; sg[3][1] += so; sg[3][2] += gf; sg[3][3] += gl; }
            }
        }
    }
    // ---- one partial per workgroup: cross-wave reduction of the accumulators through LDS (fixed order), border sums per co
    __syncthreads();
    float* const scr = lds;                          // [4 waves][NACC * 256] -- the launcher sizes the LDS for it
    float* const part = q.part + (int64_t)blockIdx.x * UbCfg<C>::PART;
#pragma unroll
    for (int t = 0; t < NACC; ++t) *reinterpret_cast<f32x4*>(scr + wave * NACC * 256 + (t * 64 + lane) * 4) = wacc[t];
    __syncthreads();
    for (int o = tid; o < NACC * 256; o += 256)
        part[o] = (scr[o] + scr[NACC * 256 + o]) + (scr[2 * NACC * 256 + o] + scr[3 * NACC * 256 + o]);
    __syncthreads();
    // border sums: 32 threads per co (8 rows x 4 quarters) -> sum in a fixed order through LDS
    {
        float* const sgl = lds;                      // [256 threads][16]
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < 4; ++c) sgl[tid * 16 + k * 4 + c] = sg[k][c];
        __syncthreads();
        if (tid < 128) {
            const int co = tid >> 4, kc = tid & 15;
            float s = 0.f;
            for (int t = 0; t < 32; ++t) s += sgl[(co * 32 + t) * 16 + kc];
            part[NACC * 256 + tid] = s;
        }
    }
}

// ---- reduction over the workgroups' partials: total[o] = sum_w part[w][o]  (16 slices per output, as wgrad_reduce_batch) ----------
struct UrArgs { const float* part[PC_MAX_GROUP]; float* total[PC_MAX_GROUP]; int nwg, PART; };
__global__ __launch_bounds__(256) void up_reduce_kernel(const UrArgs a) {
    __shared__ float red[256];
    const float* part = a.part[blockIdx.y];
    const int tid = threadIdx.x, slice = tid >> 4, o = blockIdx.x * 16 + (tid & 15);
    float s0 = 0.f, s1 = 0.f;
    if (o < a.PART) {
        int wg = slice;
        for (; wg + 16 < a.nwg; wg += 32) {
            s0 += part[(int64_t)wg * a.PART + o];
            s1 += part[(int64_t)(wg + 16) * a.PART + o];
        }
        for (; wg < a.nwg; wg += 16) s0 += part[(int64_t)wg * a.PART + o];
    }
    red[tid] = s0 + s1;
    __syncthreads();
    if (tid < 16 && o < a.PART) {
        float t = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) t += red[s * 16 + tid];
        a.total[blockIdx.y][o] = t;
    }
}

// ---- chain rule: dWeff (+ the border sums) -> gradients of W[:, Cs:Cs + C], Wt, bt ------------------------------------------------
__device__ __forceinline__ void ub_rowmap(int p, int d, int& v, int& s) {
    const int t = p + d - 1;
    const int i = t < 0 ? -1 : (t >> 1);
    v = i + 1;
    s = t - 2 * i;
}
struct UcProb { const float* total; const float* w; const float* wt; const float* bt; float* dw; float* dwt; float* dbt; };
struct UcArgs { UcProb pr[PC_MAX_GROUP]; int Cs, C, accumulate; };

constexpr int UC_SLICES = 8;          // blocks per problem: each rebuilds the (small) canonical table and takes every 8th output

template <int C>
__device__ __forceinline__ void up_chain_body(const UcArgs& a, int prob, int slice) {
    const UcProb& q = a.pr[prob];
    constexpr int NACC = UbCfg<C>::NACC;
    // canonical dWeff[pY][pX][co][ci (C real + 1 "ones")][v][c3], (v, c3) in 0..2 (only the parity's 2 x 2 entries are non-zero)
    __shared__ float E[2 * 2 * 8 * (C + 1) * 9];
    __shared__ float sW[8 * C * 9], sT[(C + 1) * C * 4];          // W[:, Cs:] as [co][c'][tap];  Wt extended by the ones row = bt
    const int tid = threadIdx.x, Ct = a.Cs + C;
    // all global loads of the prologue are issued back to back (fully unrolled, compile-time trip counts): as run-time loops they
    // were ~20 dependent L2 round trips = 17 / 37 us of a kernel that computes for 2
    constexpr int NT = (NACC * 256 + 255) / 256, NW = (8 * C * 9 + 255) / 256, NTT = ((C + 1) * C * 4 + 255) / 256;
    float rt[NT], rw[NW], rtt[NTT];
#pragma unroll
    for (int k = 0; k < NT; ++k) rt[k] = q.total[tid + 256 * k];
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const int e = tid + 256 * k, ec = e < 8 * C * 9 ? e : 0, co = ec / (C * 9), r = ec - co * C * 9;
        rw[k] = q.w[(co * Ct + a.Cs) * 9 + r];
    }
#pragma unroll
    for (int k = 0; k < NTT; ++k) {
        const int e = tid + 256 * k;
        rtt[k] = e < C * C * 4 ? q.wt[e] : ((e < (C + 1) * C * 4 && q.bt) ? q.bt[(e - C * C * 4) >> 2] : 0.f);
    }
    for (int e = tid; e < 2 * 2 * 8 * (C + 1) * 9; e += 256) E[e] = 0.f;
#pragma unroll
    for (int k = 0; k < NW; ++k) if (tid + 256 * k < 8 * C * 9) sW[tid + 256 * k] = rw[k];
#pragma unroll
    for (int k = 0; k < NTT; ++k) if (tid + 256 * k < (C + 1) * C * 4) sT[tid + 256 * k] = rtt[k];
    __syncthreads();
    // accumulators -> canonical entries
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        const int e = tid + 256 * k;
        const int acc = e >> 8, lane = (e >> 2) & 63, el = e & 3, lk = lane >> 4, li = lane & 15;
        const int i = 4 * lk + el, pX = i >> 3, co = i & 7;
        int pY, tv, c, ci;
        if (C == 16) { pY = acc / 6; tv = (acc / 3) & 1; c = acc % 3; ci = li; }
        else { pY = acc / 3; c = acc % 3; tv = li >> 3; ci = li & 7; }
        if (c == pX || c == pX + 1) E[((((pY * 2 + pX) * 8 + co) * (C + 1) + ci) * 3 + (tv + pY)) * 3 + c] = rt[k];
    }
    // the ones channel from the border sums: total[NACC*256 + co*16 + k*4 + comp], k = {pY 0, pY 1, row 0, row H-1}, comp = {se, so, gf, gl}
    if (tid < 8 * 2 * 2 * 2 * 2) {
        const int co = tid >> 4, pY = (tid >> 3) & 1, pX = (tid >> 2) & 1, tv = (tid >> 1) & 1, tc = tid & 1;
        const float* sg = q.total + NACC * 256 + co * 16;
        const int v = tv + pY, c3 = tc + pX;
        float s = sg[pY * 4 + pX];                                           // parity sum over all rows
        float edge = (pX == 0 && c3 == 0) ? sg[pY * 4 + 2] : ((pX == 1 && c3 == 2) ? sg[pY * 4 + 3] : 0.f);
        if (pY == 0 && v == 0) { s -= sg[2 * 4 + pX]; edge -= (pX == 0 && c3 == 0) ? sg[2 * 4 + 2] : ((pX == 1 && c3 == 2) ? sg[2 * 4 + 3] : 0.f); }
        if (pY == 1 && v == 2) { s -= sg[3 * 4 + pX]; edge -= (pX == 0 && c3 == 0) ? sg[3 * 4 + 2] : ((pX == 1 && c3 == 2) ? sg[3 * 4 + 3] : 0.f); }
        E[((((pY * 2 + pX) * 8 + co) * (C + 1) + C) * 3 + v) * 3 + c3] = s - edge;
    }
    __syncthreads();
    // Both output loops run with the tap / sub-pixel indices as COMPILE-TIME constants (a thread owns a (co, c') or (ci, c') pair and
    // walks its taps in straight-line code): with run-time taps every table look-up and filter was a dependent memory access.
    constexpr int V[2][3] = {{0, 1, 1}, {1, 1, 2}};          // parity p, tap d -> low-res offset index v
    constexpr int S[2][3] = {{1, 0, 1}, {0, 1, 0}};          //                 -> sub-pixel (a or b)
    // dW[co][Cs + c'][dy][dx] = sum_{pY, pX, ci <= C} E[pY][pX][co][ci][v][c3] * Wt_ext[ci][c'][sa][sb]
    for (int pr = slice * 256 + tid; pr < 8 * C; pr += UC_SLICES * 256) {
        const int co = pr / C, cp = pr - co * C;
        float* d = q.dw + (co * Ct + a.Cs + cp) * 9;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                float sacc = 0.f;
#pragma unroll
                for (int pY = 0; pY < 2; ++pY)
#pragma unroll
                    for (int pX = 0; pX < 2; ++pX) {
                        const float* e = E + (((pY * 2 + pX) * 8 + co) * (C + 1)) * 9 + V[pY][dy] * 3 + V[pX][dx];
                        const float* t = sT + (cp * 2 + S[pY][dy]) * 2 + S[pX][dx];
#pragma unroll
                        for (int ci = 0; ci <= C; ++ci) sacc += e[ci * 9] * t[ci * C * 4];
                    }
                d[dy * 3 + dx] = a.accumulate ? d[dy * 3 + dx] + sacc : sacc;
            }
    }
    // dWt_ext[ci][c'][sa][sb] = sum_{co, (pY, dy) -> sa, (pX, dx) -> sb} E[..] * W[co][Cs + c'][dy][dx];  row ci = C: dbt (summed over sa, sb)
    for (int pr = tid; pr < (C + 1) * C; pr += 256) {
        if ((pr % UC_SLICES) != slice) continue;
        const int ci = pr / C, cp = pr - ci * C;
        float r4[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
        for (int pY = 0; pY < 2; ++pY)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int pX = 0; pX < 2; ++pX)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const float* e = E + ((((pY * 2 + pX) * 8) * (C + 1) + ci) * 3 + V[pY][dy]) * 3 + V[pX][dx];
                        const float* wv = sW + cp * 9 + dy * 3 + dx;
                        float sacc = 0.f;
#pragma unroll
                        for (int co = 0; co < 8; ++co) sacc += e[co * (C + 1) * 9] * wv[co * C * 9];
                        r4[S[pY][dy]][S[pX][dx]] += sacc;
                    }
        if (ci < C) {
            float* d = q.dwt + (ci * C + cp) * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = a.accumulate ? d[k] + r4[k >> 1][k & 1] : r4[k >> 1][k & 1];
        } else if (q.dbt) {
            const float sb_ = (r4[0][0] + r4[0][1]) + (r4[1][0] + r4[1][1]);
            q.dbt[cp] = a.accumulate ? q.dbt[cp] + sb_ : sb_;
        }
    }
}

template <int C>
__global__ __launch_bounds__(256) void up_chain_kernel(const UcArgs a) { up_chain_body<C>(a, blockIdx.x, blockIdx.y); }

// both Up levels (8- and 16-channel) of a backward pass in one launch: blocks [0, n8) -> the 8-channel problems
__global__ __launch_bounds__(256) void up_chain_both_kernel(const UcArgs a8, const UcArgs a16, int n8) {
    if ((int)blockIdx.x < n8) up_chain_body<8>(a8, blockIdx.x, blockIdx.y);
    else up_chain_body<16>(a16, blockIdx.x - n8, blockIdx.y);
}

bool ub_plane_ok(const void* p, int64_t bs, int64_t cs, int rs, int xs, int dtype) {
    return p && dtype == PC_F32 && xs <= 1 && (reinterpret_cast<uintptr_t>(p) & 15) == 0 && bs % 4 == 0 && cs % 4 == 0 && rs % 4 == 0;
}

template <int C, int W>
int launch_up_bwd(const UbArgs& a, int n, int nwg, hipStream_t st) {
    constexpr int w = W / 2;
    constexpr int GRS = W + 8, GCS = UB_GROWS * GRS + 20;
    constexpr int ZRS = C == 8 ? (w + 8 + 31) / 32 * 32 + 16 : w + 8;
    constexpr int ZCS = (UB_RB + 2) * ZRS + (C == 8 ? 2 : ((34 - ((UB_RB + 2) * ZRS) % 32) % 32));
    size_t fl = (size_t)8 * GCS + (size_t)C * ZCS;
    const size_t scratch = (size_t)4 * UbCfg<C>::NACC * 256;
    if (scratch > fl) fl = scratch;
    if ((size_t)256 * 16 > fl) fl = 256 * 16;
    const size_t lds = fl * sizeof(float);
    static pc_once_per_device once;
    if (once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&up_bwd_kernel<C, W>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        once.mark();
    }
    hipLaunchKernelGGL((up_bwd_kernel<C, W>), dim3(nwg, n), dim3(256), lds, st, a);
    PC_CHECK_LAUNCH();
    return 0;
}

constexpr int UB_MAX_WG = 256;       // persistent workgroups per problem (2 per CU)

}  // namespace

// floats of the per-problem scratch of pc_conv3x3_up_bwd_group: partials [nwg][PART] followed by the reduced total [PART]
extern "C" int64_t pc_conv3x3_up_bwd_ws_bytes(int B, int H, int C) {
    (void)B; (void)H;
    const int64_t part = C == 16 ? UbCfg<16>::PART : UbCfg<8>::PART;
    return (UB_MAX_WG + 1) * part * (int64_t)sizeof(float);
}

extern "C" int pc_conv3x3_up_bwd_ok(const pc_src* g, const pc_src* z, const pc_dst* gz, int H, int W, int Cs, int C) {
    if (g_pc_precision != PC_PREC_FP32 || !g || !z) return 0;
    if (!((Cs == 8 && C == 8) || (Cs == 16 && C == 16)) || (H & 3) || (W & 7) || W < 16) return 0;
    if (g->C != 8 || g->H != H || g->W != W || g->mode != PC_SRC_DIRECT || g->oy || g->ox || z->C != C || z->H * 2 != H || z->W * 2 != W ||
        z->mode != PC_SRC_DIRECT || z->oy || z->ox)
        return 0;
    if (!ub_plane_ok(g->ptr, g->bstride, g->cstride, g->rstride, g->xstride, g->dtype) || z->dtype != PC_F32 || !pc_planar(*z)) return 0;
    return !gz || ub_plane_ok(gz->ptr, gz->bstride, gz->cstride, gz->rstride, gz->xstride, gz->dtype);
}

// the pass alone: per-workgroup partials into ws (d[i].ws: [*nwg_out][*part_out] floats, then room for the total); the caller sums them
// (pc_wgrad_reduce_batch kind 2 -> ws + nwg * part) and runs pc_conv3x3_up_chain_group
extern "C" int pc_conv3x3_up_bwd_partial_group(int n, const pc_conv_up_bwd_desc* d, int B, int H, int W, int Cs, int C, int* nwg_out,
                                               int* part_out, void* stream) {
    if (n < 1 || n > PC_MAX_GROUP || !d || B < 1 || !nwg_out || !part_out) return PC_EINVAL;
    UbArgs a{};
    const int TWc = W <= 64 ? 64 : 128;                    // column tile (template parameter of the kernel)
    a.H = H; a.W = W; a.ntx = (W + TWc - 1) / TWc;
    a.blocks_per_img = (((H >> 1) + UB_RB - 1) / UB_RB) * a.ntx;
    a.nblocks = B * a.blocks_per_img;
    const int nwg = a.nblocks < UB_MAX_WG ? a.nblocks : UB_MAX_WG;
    for (int i = 0; i < n; ++i) {
        const pc_conv_up_bwd_desc& s = d[i];
        if (!s.g || !s.z || !s.z_bn || !s.fwd_ws || !s.ws || !pc_conv3x3_up_bwd_ok(s.g, s.z, s.gz, H, W, Cs, C)) return PC_EINVAL;
        UbProb& p = a.pr[i];
        p.g = s.g->ptr; p.g_bs = s.g->bstride; p.g_cs = s.g->cstride; p.g_rs = s.g->rstride;
        p.z = s.z->ptr; p.z_bs = s.z->bstride; p.z_cs = s.z->cstride; p.z_rs = s.z->rstride;
        p.z_bn = *s.z_bn;
        p.gz = nullptr; p.gz_bs = p.gz_cs = 0; p.gz_rs = 0;
        if (s.gz) { p.gz = s.gz->ptr; p.gz_bs = s.gz->bstride; p.gz_cs = s.gz->cstride; p.gz_rs = s.gz->rstride; }
        p.wd = (const float*)s.fwd_ws + (C / 8) * 1536 + 72;
        p.part = (float*)s.ws;
    }
    hipStream_t st = (hipStream_t)stream;
    const int rc = C == 16 ? (TWc == 64 ? launch_up_bwd<16, 64>(a, n, nwg, st) : launch_up_bwd<16, 128>(a, n, nwg, st))
                           : (TWc == 64 ? launch_up_bwd<8, 64>(a, n, nwg, st) : launch_up_bwd<8, 128>(a, n, nwg, st));
    *nwg_out = nwg;
    *part_out = C == 16 ? UbCfg<16>::PART : UbCfg<8>::PART;
    return rc;
}

// the chain rule of an 8-channel level (n8 problems) and a 16-channel level (n16 problems) in ONE launch
extern "C" int pc_conv3x3_up_chain_both(int n8, const pc_conv_up_bwd_desc* d8, int nwg8, int n16, const pc_conv_up_bwd_desc* d16, int nwg16,
                                        int accumulate, void* stream) {
    if (n8 < 1 || n8 > PC_MAX_GROUP || n16 < 1 || n16 > PC_MAX_GROUP || !d8 || !d16) return PC_EINVAL;
    UcArgs a8{}, a16{};
    for (int i = 0; i < n8; ++i) {
        const pc_conv_up_bwd_desc& s = d8[i];
        if (!s.w || !s.wt || !s.ws || !s.dw || !s.dwt) return PC_EINVAL;
        a8.pr[i] = UcProb{(const float*)s.ws + (int64_t)nwg8 * UbCfg<8>::PART, s.w, s.wt, s.bt, s.dw, s.dwt, s.dbt};
    }
    for (int i = 0; i < n16; ++i) {
        const pc_conv_up_bwd_desc& s = d16[i];
        if (!s.w || !s.wt || !s.ws || !s.dw || !s.dwt) return PC_EINVAL;
        a16.pr[i] = UcProb{(const float*)s.ws + (int64_t)nwg16 * UbCfg<16>::PART, s.w, s.wt, s.bt, s.dw, s.dwt, s.dbt};
    }
    a8.Cs = 8; a8.C = 8; a8.accumulate = accumulate;
    a16.Cs = 16; a16.C = 16; a16.accumulate = accumulate;
    hipLaunchKernelGGL(up_chain_both_kernel, dim3(n8 + n16, UC_SLICES), dim3(256), 0, (hipStream_t)stream, a8, a16, n8);
    PC_CHECK_LAUNCH();
    return 0;
}

// chain rule from the reduced totals (d[i].ws + nwg * part) to dw[:, Cs:], dwt, dbt
extern "C" int pc_conv3x3_up_chain_group(int n, const pc_conv_up_bwd_desc* d, int accumulate, int nwg, int Cs, int C, void* stream) {
    if (n < 1 || n > PC_MAX_GROUP || !d) return PC_EINVAL;
    UcArgs uc{};
    const int PART = C == 16 ? UbCfg<16>::PART : UbCfg<8>::PART;
    for (int i = 0; i < n; ++i) {
        const pc_conv_up_bwd_desc& s = d[i];
        if (!s.w || !s.wt || !s.ws || !s.dw || !s.dwt) return PC_EINVAL;
        uc.pr[i] = UcProb{(const float*)s.ws + (int64_t)nwg * PART, s.w, s.wt, s.bt, s.dw, s.dwt, s.dbt};
    }
    uc.Cs = Cs; uc.C = C; uc.accumulate = accumulate;
    hipStream_t st = (hipStream_t)stream;
    if (C == 16) hipLaunchKernelGGL(up_chain_kernel<16>, dim3(n, UC_SLICES), dim3(256), 0, st, uc);
    else hipLaunchKernelGGL(up_chain_kernel<8>, dim3(n, UC_SLICES), dim3(256), 0, st, uc);
    PC_CHECK_LAUNCH();
    return 0;
}

extern "C" int pc_conv3x3_up_bwd_group(int n, const pc_conv_up_bwd_desc* d, int accumulate, int B, int H, int W, int Cs, int C, void* stream) {
    if (n < 1 || n > PC_MAX_GROUP || !d || B < 1) return PC_EINVAL;
    UbArgs a{};
    UrArgs ur{};
    UcArgs uc{};
    const int TWc = W <= 64 ? 64 : 128;
    a.H = H; a.W = W; a.ntx = (W + TWc - 1) / TWc;
    a.blocks_per_img = (((H >> 1) + UB_RB - 1) / UB_RB) * a.ntx;
    a.nblocks = B * a.blocks_per_img;
    const int nwg = a.nblocks < UB_MAX_WG ? a.nblocks : UB_MAX_WG;
    const int PART = C == 16 ? UbCfg<16>::PART : UbCfg<8>::PART;
    for (int i = 0; i < n; ++i) {
        const pc_conv_up_bwd_desc& s = d[i];
        if (!s.g || !s.z || !s.z_bn || !s.w || !s.wt || !s.fwd_ws || !s.ws || !s.dw || !s.dwt || !pc_conv3x3_up_bwd_ok(s.g, s.z, s.gz, H, W, Cs, C))
            return PC_EINVAL;
        UbProb& p = a.pr[i];
        p.g = s.g->ptr; p.g_bs = s.g->bstride; p.g_cs = s.g->cstride; p.g_rs = s.g->rstride;
        p.z = s.z->ptr; p.z_bs = s.z->bstride; p.z_cs = s.z->cstride; p.z_rs = s.z->rstride;
        p.z_bn = *s.z_bn;
        p.gz = nullptr; p.gz_bs = p.gz_cs = 0; p.gz_rs = 0;
        if (s.gz) { p.gz = s.gz->ptr; p.gz_bs = s.gz->bstride; p.gz_cs = s.gz->cstride; p.gz_rs = s.gz->rstride; }
        p.wd = (const float*)s.fwd_ws + (C / 8) * 1536 + 72;        // the data-gradient operand image written by compose_up_kernel
        p.part = (float*)s.ws;
        ur.part[i] = p.part;
        ur.total[i] = p.part + (int64_t)nwg * PART;
        uc.pr[i] = UcProb{ur.total[i], s.w, s.wt, s.bt, s.dw, s.dwt, s.dbt};
    }
    ur.nwg = nwg; ur.PART = PART;
    uc.Cs = Cs; uc.C = C; uc.accumulate = accumulate;
    hipStream_t st = (hipStream_t)stream;
    int rc = C == 16 ? (TWc == 64 ? launch_up_bwd<16, 64>(a, n, nwg, st) : launch_up_bwd<16, 128>(a, n, nwg, st))
                     : (TWc == 64 ? launch_up_bwd<8, 64>(a, n, nwg, st) : launch_up_bwd<8, 128>(a, n, nwg, st));
    if (rc) return rc;
    hipLaunchKernelGGL(up_reduce_kernel, dim3((PART + 15) / 16, n), dim3(256), 0, st, ur);
    PC_CHECK_LAUNCH();
    if (C == 16) hipLaunchKernelGGL(up_chain_kernel<16>, dim3(n, UC_SLICES), dim3(256), 0, st, uc);
    else hipLaunchKernelGGL(up_chain_kernel<8>, dim3(n, UC_SLICES), dim3(256), 0, st, uc);
    PC_CHECK_LAUNCH();
    return 0;
}
