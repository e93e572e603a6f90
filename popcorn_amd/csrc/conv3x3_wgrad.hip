// conv3x3_wgrad.hip -- weight / bias gradient of the 3x3 pad-1 convolution on fp32 MFMA (gfx950).
//
// Replaces the autograd weight-gradient of nn.Conv2d(3, padding=1) (reference model/DDA_model/utils/networks.py:259,263;
// 44 % of the reference's CPU train step).
//
// GEMM mapping (reduction over pixels rides on K):  D[(s,co)][(ci,v,dx)] += sum_x g[co][yp+s][x] * in[ci][yp+v-1][x+dx-1]
//     M (16) = (s, co8): s = row of an output-row pair, 8 output channels      -> A[(s,co)][k] = g[co][yp+s][x0+k]
//     N (16) = 16 consecutive columns of the (ci, v, dx) space, v = 0..3       -> B[k][(ci,v,dx)] = in[ci][yp+v-1][x0+k+dx-1]
//     K (4)  = 4 consecutive x of the row pair
// dW[co][ci][dy][dx] = D[(0,co)][(ci,dy,dx)] + D[(1,co)][(ci,dy+1,dx)]  (folded by the reduce kernel): 9 useful taps on
// 12 columns = 75 % MFMA efficiency with M fully used even at Cout = 8.
//
// A workgroup stages a (CINC x 18 x 34) input halo tile and the (COUT x 16 x 32) gradient tile in LDS with strides
// chosen so both fragment reads are bank-conflict free, accumulates over a persistent loop of tiles in registers,
// and writes ONE partial per workgroup; a second kernel reduces the partials in a fixed order (deterministic --
// no atomics).
#include "common.h"
#include "tile_loader.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4w __attribute__((ext_vector_type(4)));

constexpr int TW = 32, TH = 16;
constexpr int IN_RS = 40;                    // == 8 mod 32: the 4 input rows v of a k-step land 8 banks apart
constexpr int IN_CS = 18 * IN_RS + 20;       // 740 == 4 mod 32: the (at most) two channels of an N-block interleave
constexpr int IN_COL0 = 4;                   // LDS column of tile x0 (16-byte aligned interior for ds_write_b128)
constexpr int G_RS = 34;                     // == 2 mod 32
constexpr int G_CS = 16 * G_RS + 4;          // 548 == 4 mod 32
constexpr int MAX_WG = 256;
constexpr int PC_WGRAD_SINGLE_WG = 512;      // workgroups of a single-problem launch (256 / 512 / 1024 / 2048 measured: 2.098 / 2.083 / 2.085 / 2.099 ms per step)

struct WgradArgs {
    pc_src a, b, g;
    float* partial;       // [nwg][E]
    int ci0;              // first input channel of this launch's chunk (grid.y selects further chunks)
    int B, H, W;
    int tiles_x, tiles_y, ntiles;
    pc_fastdiv div_tx, div_tpi;
    int fast_a, fast_b, fast_g;
    int bf;               // PC_PREC_BF16: channels-last bf16 operands, conv3x3_wgrad_cl_kernel (dW / db stay fp32)
};

template <int CINC, int COUT>
struct WgradCfg {
    static constexpr int NCOL = CINC * 12;
    static constexpr int NBLK = (NCOL + 15) / 16;
    static constexpr int MB = COUT / 8;
    static constexpr int E = MB * NBLK * 256 + MB * 64;   // accumulators + bias partial sums (MFMA fragment layout, in LDS)
    static constexpr int EC = COUT * CINC * 9 + COUT;     // one partial in global memory: dW[co][cil][tap] + db[co], compacted
    static constexpr int LDS_FLOATS = CINC * IN_CS + COUT * G_CS;
};

template <int CINC, int COUT>
__global__ __launch_bounds__(256) void conv3x3_wgrad_kernel(const WgradArgs p) {
    using Cfg = WgradCfg<CINC, COUT>;
    constexpr int NBLK = Cfg::NBLK, MB = Cfg::MB;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* lin = lds;
    float* lg = lds + CINC * IN_CS;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int chunk = blockIdx.y;
    const int cbase = p.ci0 + chunk * CINC;

    // per-lane fragment offsets
    int boff[NBLK];
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb) {
        int ng = nb * 16 + li;
        if (ng >= Cfg::NCOL) ng = Cfg::NCOL - 1;      // dead columns: any in-range address, never read back
        const int ci = ng / 12, rem = ng % 12, v = rem / 3, dx = rem % 3;
        boff[nb] = ci * IN_CS + v * IN_RS + (IN_COL0 - 1) + dx + lk;
    }
    int aoff[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) aoff[mb] = (mb * 8 + (li & 7)) * G_CS + (li >> 3) * G_RS + lk;

    f32x4 acc[MB][NBLK];
    float bsum[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        bsum[mb] = 0.f;
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    for (int t = blockIdx.x; t < p.ntiles; t += gridDim.x) {
        const int tile = pc_xcd_remap(t, p.ntiles);
        const int b = (int)pc_div((uint32_t)tile, p.div_tpi);
        const int rem = tile - b * p.tiles_x * p.tiles_y;
        const int ty = (int)pc_div((uint32_t)rem, p.div_tx);
        const int x0 = (rem - ty * p.tiles_x) * TW, y0 = ty * TH;
        __syncthreads();
        pc_load_halo_tile<CINC, IN_RS, IN_CS, IN_COL0, true>(lin, p.a, p.b, p.fast_a, p.fast_b, cbase, b, y0, x0, p.H, p.W, tid);
        // gradient tile: COUT x 16 rows x 32 cols, no halo; 8 lanes x float4 per row
        for (int job = tid >> 3; job < COUT * 16; job += 32) {
            const int co = job >> 4, r = job & 15, l8 = tid & 7;
            const int y = y0 + r, x = x0 + 4 * l8;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.fast_g) {
                if (y < p.H && x < p.W)
                    v = *reinterpret_cast<const f32x4*>(p.g.ptr + b * p.g.bstride + co * p.g.cstride + (int64_t)y * p.g.rstride + x);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = pc_fetch(p.g, b, co, y, x + e, p.H, p.W);
            }
            float* d = lg + co * G_CS + r * G_RS + 4 * l8;
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = v[e];
        }
        __syncthreads();
#pragma unroll
        for (int rpi = 0; rpi < 2; ++rpi) {
            const int rp = 2 * wave + rpi;
#pragma unroll 2
            for (int j = 0; j < 8; ++j) {
                float av[MB], bv[NBLK];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    av[mb] = lg[aoff[mb] + 2 * rp * G_RS + 4 * j];
                    bsum[mb] += av[mb];
                }
#pragma unroll
                for (int nb = 0; nb < NBLK; ++nb) bv[nb] = lin[boff[nb] + 2 * rp * IN_RS + 4 * j];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NBLK; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mb], bv[nb], acc[mb][nb], 0, 0, 0);
            }
        }
    }

    // ---- cross-wave reduction through LDS (fixed order), one partial per workgroup
    // The partial leaves the workgroup COMPACTED to dW[co][cil][tap] + db[co]: the fragment layout carries every weight
    // twice (output-row slot s = 0 / 1 of the pair mapping) plus dead (row, tap) slots, 2.7x the bytes, and the batched
    // second stage is bound by reading the partials.
    float* part = p.partial + ((int64_t)(blockIdx.y * gridDim.x + blockIdx.x)) * Cfg::EC;
    auto wsum = [&](int e) { return ((lds[e] + lds[NBLK * 256 + e]) + lds[2 * NBLK * 256 + e]) + lds[3 * NBLK * 256 + e]; };
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        __syncthreads();
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb)
            *reinterpret_cast<f32x4*>(&lds[((wave * NBLK + nb) * 64 + lane) * 4]) = acc[mb][nb];
        __syncthreads();
        for (int idx = tid; idx < 8 * CINC * 9; idx += 256) {
            const int c8 = idx / (CINC * 9), rem = idx - c8 * (CINC * 9);
            const int cil = rem / 9, tap = rem - cil * 9, dy = tap / 3, dx = tap - dy * 3;
            // D[m = s*8 + c8][ng]; lane = (m>>2)*16 + (ng&15), reg = m&3
            const int ng0 = cil * 12 + dy * 3 + dx, ng1 = ng0 + 3;
            const int m0 = c8, m1 = 8 + c8;
            const int e0 = (((ng0 >> 4)) * 64 + (m0 >> 2) * 16 + (ng0 & 15)) * 4 + (m0 & 3);
            const int e1 = (((ng1 >> 4)) * 64 + (m1 >> 2) * 16 + (ng1 & 15)) * 4 + (m1 & 3);
            part[(mb * 8 + c8) * (CINC * 9) + rem] = wsum(e0) + wsum(e1);
        }
    }
    __syncthreads();
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) lds[(wave * MB + mb) * 64 + lane] = bsum[mb];
    __syncthreads();
    if (tid < COUT) {
        // bias: the 8 lanes (s in 0..1, lk in 0..3) that carry channel co
        const int mb = tid >> 3, c8 = tid & 7;
        float t = 0.f;
#pragma unroll
        for (int lk = 0; lk < 4; ++lk) {
            const int ea = mb * 64 + lk * 16 + c8, eb = ea + 8;
            const float sa = ((lds[ea] + lds[MB * 64 + ea]) + lds[2 * MB * 64 + ea]) + lds[3 * MB * 64 + ea];
            const float sb = ((lds[eb] + lds[MB * 64 + eb]) + lds[2 * MB * 64 + eb]) + lds[3 * MB * 64 + eb];
            t += sa + sb;
        }
        part[COUT * CINC * 9 + tid] = t;
    }
}

// ---- wave-private variant (aligned DIRECT / POOL2 sources) -------------------------------------------------------------
// Same GEMM mapping, but every wave owns a 32 x 4 strip and its own pipeline (no workgroup barrier in the loop, see
// conv3x3.hip): the (CINC x 6 x 40) input strip is register-staged into a private LDS region one strip ahead, and the
// gradient strip is not staged at all -- with the K ordering x(j,k) = x0 + {0,16,8,24}[k] + j a lane's A operands for
// the 8 k-steps of a row pair are 8 consecutive floats, i.e. two 16-byte global loads straight into registers.
constexpr int WIN_RS = 44;                   // == 12 mod 32: rows v = 0..3 land on banks {0,12,24,4} (+dx, +16 for k)
constexpr int WIN_CSW = 6 * WIN_RS;          // 264

struct WgradGroup {
    WgradArgs pr[PC_MAX_GROUP];      // problems of identical geometry and loader kind: blockIdx.z selects
};

// the problem's descriptor, pinned in scalar registers (common.h: pc_pin; round 6: the strip loops re-loaded its fields from the kernel
// arguments 26 - 78 times per iteration).  The REFLECT loader indexes the source's channel map per lane: that instantiation keeps the
// reference into the kernel arguments.
template <int LD>
__device__ __forceinline__ const WgradArgs& wgrad_pinned(const WgradArgs& in, WgradArgs& local) {
    if constexpr (LD == 3) {
        return in;
    } else {
        local = in;
        pc_pin(local.a); pc_pin(local.b); pc_pin(local.g);
        local.partial = pc_pin_ptr(local.partial);
        pc_pin(local.ci0); pc_pin(local.B); pc_pin(local.H); pc_pin(local.W); pc_pin(local.tiles_x); pc_pin(local.tiles_y); pc_pin(local.ntiles);
        pc_pin(local.div_tx); pc_pin(local.div_tpi);
        return local;
    }
}

template <int CINC, int COUT, int LD>
__global__ __launch_bounds__(256) void conv3x3_wgrad_wave_kernel(const WgradGroup grp_) {
    WgradArgs p_local;
    const WgradArgs& p = wgrad_pinned<LD>(grp_.pr[blockIdx.z], p_local);
    using Cfg = WgradCfg<CINC, COUT>;
    constexpr int NBLK = Cfg::NBLK, MB = Cfg::MB;
    constexpr int NIT = CINC;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int chunk = blockIdx.y;
    const int cbase = p.ci0 + chunk * CINC;
    float* const wl = lds + wave * (CINC * WIN_CSW);

    const int koff = ((lk & 1) << 4) | ((lk >> 1) << 3);          // {0,16,8,24}
    int boff[NBLK];
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb) {
        int ng = nb * 16 + li;
        if (ng >= Cfg::NCOL) ng = Cfg::NCOL - 1;      // dead columns: any in-range address, never read back
        const int ci = ng / 12, rem = ng % 12, v = rem / 3, dx = rem % 3;
        boff[nb] = ci * WIN_CSW + v * WIN_RS + (IN_COL0 - 1) + dx + koff;
    }

    // ---- input-strip loader: lane = (row r of 6, 16-byte segment of the 40-float row)
    const int l_r = lane / 10, l_seg = lane - l_r * 10;
    const bool l_act = lane < 60;
    const int CA = p.a.C;
    const int64_t in_bs = p.a.bstride;
    const int in_rs = p.a.rstride;
    f32x4 R[NIT];
    bool rvalid = false;
    // ---- gradient strip: lane (i = (s, co8), k): rows y0 + 2*rpi + s, 8 floats at x0 + koff
    const int g_s = li >> 3, g_c = li & 7;
    f32x4 G[2][MB][2];
    unsigned gvalid = 0;      // bit (rpi*2 + half)

    auto issue = [&](int b, int y0, int x0) {
        const int xg = x0 - 4 + 4 * l_seg, y = y0 - 1 + l_r;
        const bool ok = l_act && xg >= 0 && xg < p.W && (unsigned)y < (unsigned)p.H;
        rvalid = ok;
        if (LD == 3) {
            rvalid = l_act;
#pragma unroll
            for (int it = 0; it < NIT; ++it)
                R[it] = l_act ? pc_fetch_reflect_seg(p.a, b, cbase + it, y, xg, p.H, p.W) : f32x4{0.f, 0.f, 0.f, 0.f};
        } else if (LD == 1) {
            const int64_t off = ok ? b * in_bs + (int64_t)y * in_rs + xg : 0;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int cg = cbase + it;
                const float* cp = cg < CA ? p.a.ptr + cg * p.a.cstride : p.b.ptr + (cg - CA) * p.b.cstride;
                R[it] = *reinterpret_cast<const f32x4*>(cp + off);
            }
        } else {
            const int64_t off = ok ? b * in_bs + (int64_t)(2 * y) * in_rs + 2 * xg : 0;
            const int rs1 = ok ? in_rs : 0;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const float* s0 = p.a.ptr + (cbase + it) * p.a.cstride + off;
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(s0), a1 = *reinterpret_cast<const f32x4*>(s0 + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(s0 + rs1), b1 = *reinterpret_cast<const f32x4*>(s0 + rs1 + 4);
                f32x4 v;
                v[0] = fmaxf(fmaxf(a0[0], a0[1]), fmaxf(b0[0], b0[1]));
                v[1] = fmaxf(fmaxf(a0[2], a0[3]), fmaxf(b0[2], b0[3]));
                v[2] = fmaxf(fmaxf(a1[0], a1[1]), fmaxf(b1[0], b1[1]));
                v[3] = fmaxf(fmaxf(a1[2], a1[3]), fmaxf(b1[2], b1[3]));
                R[it] = v;
            }
        }
        unsigned gm = 0;
#pragma unroll
        for (int rpi = 0; rpi < 2; ++rpi) {
            const int yy = y0 + 2 * rpi + g_s;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int xx = x0 + koff + 4 * h;
                const bool gok = yy < p.H && xx < p.W;
                gm |= gok ? 1u << (rpi * 2 + h) : 0u;
                const int64_t goff = gok ? b * p.g.bstride + (int64_t)yy * p.g.rstride + xx : 0;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
                    G[rpi][mb][h] = *reinterpret_cast<const f32x4*>(p.g.ptr + (mb * 8 + g_c) * p.g.cstride + goff);
            }
        }
        gvalid = gm;
    };
    auto commit = [&]() {
        if (l_act) {
            float* d = wl + l_r * WIN_RS + 4 * l_seg;
#pragma unroll
            for (int it = 0; it < NIT; ++it)
                *reinterpret_cast<f32x4*>(d + it * WIN_CSW) = rvalid ? R[it] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };

    f32x4 acc[MB][NBLK];
    float bsum[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        bsum[mb] = 0.f;
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

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
    for (int k = 0; k < my_tiles; ++k) {
        commit();
        // A operands of this strip (masked), then prefetch the next strip
        float av[2][MB][8];
#pragma unroll
        for (int rpi = 0; rpi < 2; ++rpi)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const bool gok = (gvalid >> (rpi * 2 + h)) & 1u;
#pragma unroll
                    for (int e = 0; e < 4; ++e) av[rpi][mb][h * 4 + e] = gok ? G[rpi][mb][h][e] : 0.f;
                }
        if (k + 1 < my_tiles) {
            strip_coords(k + 1, b, y0, x0);
            issue(b, y0, x0);
        }
#pragma unroll
        for (int rpi = 0; rpi < 2; ++rpi) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float bv[NBLK];
#pragma unroll
                for (int nb = 0; nb < NBLK; ++nb) bv[nb] = wl[boff[nb] + 2 * rpi * WIN_RS + j];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    bsum[mb] += av[rpi][mb][j];
#pragma unroll
                    for (int nb = 0; nb < NBLK; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rpi][mb][j], bv[nb], acc[mb][nb], 0, 0, 0);
                }
            }
        }
    }

    // ---- cross-wave reduction through LDS (fixed order), one partial per workgroup
    // The partial leaves the workgroup COMPACTED to dW[co][cil][tap] + db[co]: the fragment layout carries every weight
    // twice (output-row slot s = 0 / 1 of the pair mapping) plus dead (row, tap) slots, 2.7x the bytes, and the batched
    // second stage is bound by reading the partials.
    float* part = p.partial + ((int64_t)(blockIdx.y * gridDim.x + blockIdx.x)) * Cfg::EC;
    auto wsum = [&](int e) { return ((lds[e] + lds[NBLK * 256 + e]) + lds[2 * NBLK * 256 + e]) + lds[3 * NBLK * 256 + e]; };
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        __syncthreads();
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb)
            *reinterpret_cast<f32x4*>(&lds[((wave * NBLK + nb) * 64 + lane) * 4]) = acc[mb][nb];
        __syncthreads();
        for (int idx = tid; idx < 8 * CINC * 9; idx += 256) {
            const int c8 = idx / (CINC * 9), rem = idx - c8 * (CINC * 9);
            const int cil = rem / 9, tap = rem - cil * 9, dy = tap / 3, dx = tap - dy * 3;
            // D[m = s*8 + c8][ng]; lane = (m>>2)*16 + (ng&15), reg = m&3
            const int ng0 = cil * 12 + dy * 3 + dx, ng1 = ng0 + 3;
            const int m0 = c8, m1 = 8 + c8;
            const int e0 = (((ng0 >> 4)) * 64 + (m0 >> 2) * 16 + (ng0 & 15)) * 4 + (m0 & 3);
            const int e1 = (((ng1 >> 4)) * 64 + (m1 >> 2) * 16 + (ng1 & 15)) * 4 + (m1 & 3);
            part[(mb * 8 + c8) * (CINC * 9) + rem] = wsum(e0) + wsum(e1);
        }
    }
    __syncthreads();
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) lds[(wave * MB + mb) * 64 + lane] = bsum[mb];
    __syncthreads();
    if (tid < COUT) {
        // bias: the 8 lanes (s in 0..1, lk in 0..3) that carry channel co
        const int mb = tid >> 3, c8 = tid & 7;
        float t = 0.f;
#pragma unroll
        for (int lk = 0; lk < 4; ++lk) {
            const int ea = mb * 64 + lk * 16 + c8, eb = ea + 8;
            const float sa = ((lds[ea] + lds[MB * 64 + ea]) + lds[2 * MB * 64 + ea]) + lds[3 * MB * 64 + ea];
            const float sb = ((lds[eb] + lds[MB * 64 + eb]) + lds[2 * MB * 64 + eb]) + lds[3 * MB * 64 + eb];
            t += sa + sb;
        }
        part[COUT * CINC * 9 + tid] = t;
    }
}

// ---- channels-last bf16 kernel (PC_PREC_BF16) -------------------------------------------------------------------------------
// Activations and gradients are channels-last bf16 (one 16-byte slot per pixel and 8-channel group, conv3x3.hip), but the
// weight gradient contracts over PIXELS: both MFMA operands need "8 pixels of one channel" per lane, the transpose of what
// memory holds.  The strips (input: 6 rows x 34 pixels, gradient: 4 rows x 32 pixels) are therefore copied slot by slot into
// the wave's LDS region and read back with ds_read_b64_tr_b16, the gfx950 transposing LDS read: the 16 lanes of a group
// hand in 16 addresses of 4 contiguous bf16 (lane L: row L / 4, column quad L % 4 of a 4 x 16 block) and lane c receives
// column c of the block.  With rows = 4 consecutive pixels:
//     A (M = (s, co8)):  quad q = (s = q >> 1, channels 4 * (q & 1) ..) of gradient row yp + s           -> lane (s * 8 + co)
//     B (N = (v, ci4)):  quad q = input row yp + v - 1, channels 4 * (nb & 1) .. of chunk nb >> 1, pixel + dx -> lane (4 * v + ci)
// two reads (pixels 8 * lk + 0..3 and + 4..7) make the 8 k-slots of a lane for v_mfma_f32_16x16x32_bf16 (K = 32 = a strip row):
//     D_dx[(s,co)][(v,ci)] += sum_x g[co][yp+s][x] * in[ci][yp+v-1][x+dx-1];   dW[co][ci][dy][dx] = D_dx[(0,co)][(dy,ci)] + D_dx[(1,co)][(dy+1,ci)]
// Bank layout: rows sit 16 dwords apart (row strides 52 / 36 slots) and the two 8-byte halves of a slot are swapped for pixels
// with bit 3 set, so the two lane groups of a pass (pixels 8 apart) hit different banks.
// Same compacted per-workgroup partial as the other kernels; the first layers (reflect loader, planar fp32 model input, 2 / 4
// channels) use channel slots 0..3 of one block.
constexpr int CLW_IN_RS = 52, CLW_G_RS = 36;        // slots per strip row

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ s16x4 clw_tr(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}
__device__ __forceinline__ bf16x8 clw_pair(s16x4 a, s16x4 b) {
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ u32x4w clw_swz(u32x4w v, bool sw) { return sw ? u32x4w{v[2], v[3], v[0], v[1]} : v; }
__device__ __forceinline__ u32x4w clw_max8(u32x4w a, u32x4w b) {
    u32x4w o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float lo = fmaxf(__uint_as_float(a[e] << 16), __uint_as_float(b[e] << 16));
        const float hi = fmaxf(__uint_as_float(a[e] & 0xffff0000u), __uint_as_float(b[e] & 0xffff0000u));
        o[e] = (__float_as_uint(hi) & 0xffff0000u) | (__float_as_uint(lo) >> 16);
    }
    return o;
}

template <int CINC, int COUT>
struct WgradClCfg {
    static constexpr int MB = COUT / 8;
    static constexpr int NCH = CINC <= 8 ? 1 : CINC / 8;      // 8-channel images of the input strip
    static constexpr int NBP = CINC <= 4 ? 1 : CINC / 4;      // N blocks (4 rows x 4 channels) per tap
    static constexpr int NBLK = 3 * NBP;
    static constexpr int IN_B = 6 * CLW_IN_RS * 16, G_B = 4 * CLW_G_RS * 16;      // bytes of one image
    static constexpr size_t WAVE_B = (size_t)NCH * IN_B + (size_t)MB * G_B;
    static constexpr size_t RED_B = (size_t)4 * NBLK * 256 * sizeof(float);
    static constexpr size_t LDS_B = 4 * WAVE_B > RED_B ? 4 * WAVE_B : RED_B;
};

template <int CINC, int COUT, int LD>
__global__ __launch_bounds__(256) void conv3x3_wgrad_cl_kernel(const WgradGroup grp_) {
    WgradArgs p_local;
    const WgradArgs& p = wgrad_pinned<LD>(grp_.pr[blockIdx.z], p_local);
    using Cfg = WgradCfg<CINC, COUT>;
    using Cl = WgradClCfg<CINC, COUT>;
    constexpr int MB = Cl::MB, NCH = Cl::NCH, NBP = Cl::NBP, NBLK = Cl::NBLK;
    constexpr int NIT = CINC < 8 ? CINC : 1;      // REFLECT loader: one 16-byte segment of a planar fp32 row per channel
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int chunk = blockIdx.y;
    const int cbase = p.ci0 + chunk * CINC;
    unsigned char* const win = reinterpret_cast<unsigned char*>(lds) + wave * Cl::WAVE_B;
    unsigned char* const wg = win + NCH * Cl::IN_B;

    // ---- loaders: input pieces id = lane + 64 * i -> (row of 6, pixel of 34); gradient pieces -> (mb, row of 4, pixel of 32)
    int i_r[4], i_px[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = lane + 64 * i;
        i_r[i] = id / 34;
        i_px[i] = id - i_r[i] * 34;
    }
    const int CA = p.a.C;
    u32x4w R[NCH][LD == 2 ? 16 : 4];
    f32x4 RF[LD == 3 ? NIT : 1];
    u32x4w G[2 * MB];
    unsigned rvalid = 0, gvalid = 0;
    const int r_r = lane / 10, r_seg = lane - r_r * 10;
    const pc_bf16_t* const g_ptr = reinterpret_cast<const pc_bf16_t*>(p.g.ptr);
    auto issue = [&](int b, int y0, int x0) {
        if constexpr (LD == 3) {
#pragma unroll
            for (int it = 0; it < NIT; ++it)
                RF[it] = lane < 60 ? pc_fetch_reflect_seg(p.a, b, cbase + it, y0 - 1 + r_r, x0 - 4 + 4 * r_seg, p.H, p.W) : f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
            unsigned vm = 0;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int cg = cbase + 8 * c;
                const bool useb = LD == 1 && cg >= CA;
                const pc_src& s = useb ? p.b : p.a;
                const pc_bf16_t* base = reinterpret_cast<const pc_bf16_t*>(s.ptr) + b * s.bstride + (useb ? cg - CA : cg);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int y = y0 - 1 + i_r[i], x = x0 - 1 + i_px[i];
                    bool ok = lane + 64 * i < 204 && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
                    if constexpr (LD == 1) {
                        const int ys = y - s.oy, xq = x - s.ox;
                        ok = ok && (unsigned)ys < (unsigned)s.H && (unsigned)xq < (unsigned)s.W;
                        const int64_t off = ok ? (int64_t)ys * s.rstride + (int64_t)xq * s.xstride : 0;
                        R[c][i] = *reinterpret_cast<const u32x4w*>(base + off);
                    } else {
                        const int64_t off = ok ? (int64_t)(2 * y) * s.rstride + (int64_t)(2 * x) * s.xstride : 0;
                        const int rs1 = ok ? s.rstride : 0, xs1 = ok ? s.xstride : 0;
                        R[c][4 * i + 0] = *reinterpret_cast<const u32x4w*>(base + off);
                        R[c][4 * i + 1] = *reinterpret_cast<const u32x4w*>(base + off + xs1);
                        R[c][4 * i + 2] = *reinterpret_cast<const u32x4w*>(base + off + rs1);
                        R[c][4 * i + 3] = *reinterpret_cast<const u32x4w*>(base + off + rs1 + xs1);
                    }
                    vm |= (ok ? 1u : 0u) << (4 * c + i);
                }
            }
            rvalid = vm;
        }
        unsigned gm = 0;
#pragma unroll
        for (int i = 0; i < 2 * MB; ++i) {
            const int id = lane + 64 * i;
            const int mb = id >> 7, r = (id >> 5) & 3, px = id & 31;
            const int y = y0 + r, x = x0 + px;
            const bool ok = y < p.H && x < p.W;
            const int64_t off = ok ? b * p.g.bstride + (int64_t)y * p.g.rstride + (int64_t)x * p.g.xstride + 8 * mb : 0;
            G[i] = *reinterpret_cast<const u32x4w*>(g_ptr + off);
            gm |= (ok ? 1u : 0u) << i;
        }
        gvalid = gm;
    };
    auto commit = [&]() {
        if constexpr (LD == 3) {
            if (lane < 60) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int pi = 4 * r_seg - 3 + e;          // pixel x0 - 4 + 4 * seg + e  ->  strip pixel index (x0 - 1 = 0)
                    if (pi >= 0 && pi < 34) {
                        u32x4w t = u32x4w{0u, 0u, 0u, 0u};
#pragma unroll
                        for (int h = 0; h < (NIT + 1) / 2; ++h)
                            t[h] = pc_pack_bf16(RF[2 * h][e], 2 * h + 1 < NIT ? RF[(2 * h + 1) % NIT][e] : 0.f);
                        *reinterpret_cast<u32x4w*>(win + (r_r * CLW_IN_RS + pi) * 16) = clw_swz(t, (pi >> 3) & 1);
                    }
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (lane + 64 * i < 204) {
                        u32x4w v;
                        if constexpr (LD == 2) v = clw_max8(clw_max8(R[c][4 * i], R[c][4 * i + 1]), clw_max8(R[c][4 * i + 2], R[c][4 * i + 3]));
                        else v = R[c][i];
                        if (!((rvalid >> (4 * c + i)) & 1u)) v = u32x4w{0u, 0u, 0u, 0u};
                        *reinterpret_cast<u32x4w*>(win + c * Cl::IN_B + (i_r[i] * CLW_IN_RS + i_px[i]) * 16) = clw_swz(v, (i_px[i] >> 3) & 1);
                    }
                }
        }
#pragma unroll
        for (int i = 0; i < 2 * MB; ++i) {
            const int id = lane + 64 * i;
            const int mb = id >> 7, r = (id >> 5) & 3, px = id & 31;
            const u32x4w v = ((gvalid >> i) & 1u) ? G[i] : u32x4w{0u, 0u, 0u, 0u};
            *reinterpret_cast<u32x4w*>(wg + mb * Cl::G_B + (r * CLW_G_RS + px) * 16) = clw_swz(v, (px >> 3) & 1);
        }
    };

    // ---- transposing-read addresses of this lane: pixel row j = li >> 2 of the 4 x 16 block, column quad q = li & 3
    const int t_j = li >> 2, t_q = li & 3;
    // A: quad = (s = q >> 1, channel half q & 1), pixels 8 * lk + 4 * e + j  (bit 3 of the pixel = lk & 1)
    const int a_off = ((t_q >> 1) * CLW_G_RS + 8 * lk + t_j) * 16 + 8 * ((t_q & 1) ^ (lk & 1));
    // B: quad = input row v = q; pixel index 8 * lk + 4 * e + j + dx, channel half (nb & 1) of image nb >> 1
    const int b_row = t_q * CLW_IN_RS;

    f32x4 acc[MB][NBLK];
    float bsum[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        bsum[mb] = 0.f;
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

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
    for (int k = 0; k < my_tiles; ++k) {
        commit();
        if (k + 1 < my_tiles) {
            strip_coords(k + 1, b, y0, x0);
            issue(b, y0, x0);
        }
#pragma unroll
        for (int rpi = 0; rpi < 2; ++rpi) {
            bf16x8 av[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const unsigned char* ga = wg + mb * Cl::G_B + 2 * rpi * CLW_G_RS * 16 + a_off;
                const s16x4 lo = clw_tr(ga), hi = clw_tr(ga + 4 * 16);
                av[mb] = clw_pair(lo, hi);
                float t = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) t += __uint_as_float((unsigned)(unsigned short)lo[e] << 16) + __uint_as_float((unsigned)(unsigned short)hi[e] << 16);
                bsum[mb] += t;
            }
#pragma unroll
            for (int nb = 0; nb < NBP; ++nb)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    // pixel index of read e: 8 * lk + 4 * e + j + dx  (the swizzle bit is bit 3 of THAT index)
                    const int p0 = 8 * lk + t_j + dx, p1 = p0 + 4;
                    const unsigned char* ib = win + (nb >> 1) * Cl::IN_B + (2 * rpi * CLW_IN_RS + b_row) * 16;
                    const s16x4 lo = clw_tr(ib + p0 * 16 + 8 * ((nb & 1) ^ ((p0 >> 3) & 1)));
                    const s16x4 hi = clw_tr(ib + p1 * 16 + 8 * ((nb & 1) ^ ((p1 >> 3) & 1)));
                    const bf16x8 bv = clw_pair(lo, hi);
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
                        acc[mb][dx * NBP + nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[mb], bv, acc[mb][dx * NBP + nb], 0, 0, 0);
                }
        }
    }

    // ---- cross-wave reduction through LDS (fixed order), one compacted partial per workgroup (layout of the fp32 kernels)
    float* part = p.partial + ((int64_t)(blockIdx.y * gridDim.x + blockIdx.x)) * Cfg::EC;
    auto wsum = [&](int e) { return ((lds[e] + lds[NBLK * 256 + e]) + lds[2 * NBLK * 256 + e]) + lds[3 * NBLK * 256 + e]; };
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        __syncthreads();
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb)
            *reinterpret_cast<f32x4*>(&lds[((wave * NBLK + nb) * 64 + lane) * 4]) = acc[mb][nb];
        __syncthreads();
        for (int idx = tid; idx < 8 * CINC * 9; idx += 256) {
            const int c8 = idx / (CINC * 9), rem = idx - c8 * (CINC * 9);
            const int cil = rem / 9, tap = rem - cil * 9, dy = tap / 3, dx = tap - dy * 3;
            // D_dx[m = s*8 + c8][n = 4*v + (cil & 3)] in block dx*NBP + (cil >> 2); D register layout: lane = (m>>2)*16 + n, reg = m&3
            const int blk = dx * NBP + (cil >> 2);
            const int n0 = 4 * dy + (cil & 3), n1 = n0 + 4;
            const int m0 = c8, m1 = 8 + c8;
            const int e0 = (blk * 64 + (m0 >> 2) * 16 + n0) * 4 + (m0 & 3);
            const int e1 = (blk * 64 + (m1 >> 2) * 16 + n1) * 4 + (m1 & 3);
            part[(mb * 8 + c8) * (CINC * 9) + rem] = wsum(e0) + wsum(e1);
        }
    }
    __syncthreads();
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) lds[(wave * MB + mb) * 64 + lane] = bsum[mb];
    __syncthreads();
    if (tid < COUT) {
        const int mb = tid >> 3, c8 = tid & 7;
        float t = 0.f;
#pragma unroll
        for (int lk2 = 0; lk2 < 4; ++lk2) {
            const int ea = mb * 64 + lk2 * 16 + c8, eb = ea + 8;
            const float sa = ((lds[ea] + lds[MB * 64 + ea]) + lds[2 * MB * 64 + ea]) + lds[3 * MB * 64 + ea];
            const float sb = ((lds[eb] + lds[MB * 64 + eb]) + lds[2 * MB * 64 + eb]) + lds[3 * MB * 64 + eb];
            t += sa + sb;
        }
        part[COUT * CINC * 9 + tid] = t;
    }
}

struct WreduceArgs {
    const float* partial;
    int nwg;              // partials per chunk
    int nchunk;
    float* dw;            // [COUT][CIN][3][3]
    float* db;            // [COUT]
    int Cin;
    int accumulate;
};

// 16 outputs x 16 slices per block: every thread sums nwg/16 workgroup partials with 4 independent loads in flight,
// then the 16 slices combine in a fixed order (deterministic).  (The first version used 4 slices and a serial
// dependent loop: 40 us per call for a 2 KB result -- profiles/r1_v0.)
template <int CINC, int COUT>
__global__ __launch_bounds__(256) void conv3x3_wgrad_reduce_kernel(const WreduceArgs p) {
    using Cfg = WgradCfg<CINC, COUT>;
    __shared__ float red[256];
    const int tid = threadIdx.x;
    const int slice = tid >> 4, o = blockIdx.x * 16 + (tid & 15);
    const int n_w = COUT * p.Cin * 9;
    const int n_out = n_w + COUT;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (o < n_w) {
        const int tap = o % 9, ci = (o / 9) % p.Cin, co = o / (9 * p.Cin);
        const int dy = tap / 3, dx = tap % 3;
        const int chunk = ci / CINC, cil = ci % CINC;
        (void)dy; (void)dx;
        const int oc = co * (CINC * 9) + cil * 9 + tap;       // compacted partial: dW[co][cil][tap]
        const float* base = p.partial + (int64_t)chunk * p.nwg * Cfg::EC;
        int w = slice;
        for (; w + 48 < p.nwg; w += 64) {
            const float* q0 = base + (int64_t)w * Cfg::EC;
            s0 += q0[oc];
            s1 += q0[(int64_t)16 * Cfg::EC + oc];
            s2 += q0[(int64_t)32 * Cfg::EC + oc];
            s3 += q0[(int64_t)48 * Cfg::EC + oc];
        }
        for (; w < p.nwg; w += 16) s0 += base[(int64_t)w * Cfg::EC + oc];
    } else if (o < n_out) {
        // bias: chunk 0 only
        const int co = o - n_w;
        for (int w = slice; w < p.nwg; w += 16) s0 += p.partial[(int64_t)w * Cfg::EC + COUT * CINC * 9 + co];
    }
    red[tid] = (s0 + s1) + (s2 + s3);
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

// geometry + loader classification of one problem; returns the loader kind (0 generic, 1 direct, 2 pool, 3 reflect)
template <int CINC, int COUT>
int prepare_wgrad(WgradArgs& p, int Cin, void* ws, int& nwg, int& nchunk) {
    p.tiles_x = (p.W + TW - 1) / TW;
    p.tiles_y = (p.H + TH - 1) / TH;
    p.ntiles = p.B * p.tiles_x * p.tiles_y;
    p.div_tx = pc_make_fastdiv(p.tiles_x);
    p.div_tpi = pc_make_fastdiv(p.tiles_x * p.tiles_y);
    nchunk = Cin / CINC;
    nwg = p.ntiles < MAX_WG / nchunk ? p.ntiles : MAX_WG / nchunk;
    if (nwg < 1) nwg = 1;
    p.partial = reinterpret_cast<float*>(ws);
    p.ci0 = 0;
    p.bf = g_pc_precision == PC_PREC_BF16;
    // container types follow the mode: bf16 mode = bf16 activations and gradients (the reflect-padded model input stays fp32)
    const int want = p.bf ? PC_BF16 : PC_F32;
    if (p.a.dtype != (p.a.mode == PC_SRC_REFLECT ? PC_F32 : want) || (p.b.C && p.b.dtype != want) || p.g.dtype != want) return -1;
    if (p.bf) {
        // bf16 mode: channels-last operands (the model input: planar fp32 through the reflect loader); every geometry and
        // placement offset runs on the channels-last kernel -- kind 1 direct (+ concat), 2 pooled, 3 reflect
        if (!pc_cl_ok(p.g) || p.g.C != COUT) return -1;
        if (p.a.mode == PC_SRC_REFLECT) return (CINC <= 4 && p.b.C == 0 && pc_planar(p.a)) ? 3 : -1;
        if (CINC < 8 || !pc_cl_ok(p.a) || p.a.C % 8 != 0 || (p.b.C && (!pc_cl_ok(p.b) || p.b.C % 8 != 0 || p.b.mode != PC_SRC_DIRECT))) return -1;
        if (p.a.mode == PC_SRC_POOL2) return (p.b.C == 0 && p.a.W >= 2 * p.W && p.a.H >= 2 * p.H) ? 2 : -1;
        return 1;
    }
    if (!pc_planar(p.a) || !pc_planar(p.b) || !pc_planar(p.g)) return -1;
    auto mode_of = [&](const pc_src& s) -> int {      // 1 = aligned DIRECT, 2 = aligned POOL2
        if (s.C == 0) return 0;
        if ((reinterpret_cast<uintptr_t>(s.ptr) & 15) || s.rstride % 4 || s.cstride % 4 || s.bstride % 4) return 0;
        if (s.mode == PC_SRC_DIRECT && s.oy == 0 && s.ox == 0 && s.H == p.H && s.W == p.W && (p.W % 4) == 0) return 1;
        if (s.mode == PC_SRC_POOL2 && s.W == 2 * p.W && s.H >= 2 * p.H && (p.W % 4) == 0) return 2;
        return 0;
    };
    const int ma = mode_of(p.a), mbb = mode_of(p.b), mg = mode_of(p.g);
    p.fast_a = pc_src_fast_mode(p.a, p.H, p.W);      // (the generic kernel's vector paths)
    p.fast_b = pc_src_fast_mode(p.b, p.H, p.W);
    p.fast_g = pc_src_fast_mode(p.g, p.H, p.W) == 1;
    const bool gok = mg == 1;
    const bool lay = p.b.C == 0 || (p.a.bstride == p.b.bstride && p.a.rstride == p.b.rstride);
    if (p.a.mode == PC_SRC_REFLECT && p.b.C == 0 && gok && CINC <= 4) return 3;
    if (ma == 1 && (p.b.C == 0 || mbb == 1) && lay && gok) return 1;
    if (ma == 2 && p.b.C == 0 && gok) return 2;
    return 0;
}

template <int CINC, int COUT>
int launch_wgrad_wave(const WgradGroup& g, int n, int kind, int& nwg, int nchunk, hipStream_t stream) {
    using Cfg = WgradCfg<CINC, COUT>;
    const size_t lred = (size_t)4 * Cfg::NBLK * 256 * sizeof(float);     // staging of the cross-wave reduction (same buffer)
    size_t lw_f32 = (size_t)4 * CINC * WIN_CSW * sizeof(float);          // fp32 strips
    if (lw_f32 < lred) lw_f32 = lred;
    const size_t lw_bf = WgradClCfg<CINC, COUT>::LDS_B;                   // channels-last bf16 strips / reduction staging
    static int resident[5] = {0, 0, 0, 0, 0};
    static pc_once_per_device once[5];    // workgroups of the instantiation that fit on the chip at once
    auto go = [&](auto kern, int slot) -> int {
        const size_t lw = slot >= 3 ? lw_bf : lw_f32;
        if (once[slot].need()) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lw);
            if (e != hipSuccess) return (int)e;
            hipFuncAttributes fa;
            e = hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern));
            if (e != hipSuccess) return (int)e;
            resident[slot] = pc_resident_workgroups(fa.numRegs, lw);
            once[slot].mark();
            if (getenv("POPCORN_CONV_DBG"))
                fprintf(stderr, "wgrad<%d,%d,%d>: %d regs, %zu B LDS -> %d resident workgroups\n", CINC, COUT, slot, fa.numRegs, lw,
                        resident[slot]);
        }
        // all workgroups of the launch resident at once (the 16-channel instantiations hold ONE workgroup per CU: a
        // fixed 2-per-CU grid ran them in two rounds), never more partial sums than the workspace holds
        int cap = resident[slot] / (nchunk * n);
        if (cap < 1) cap = 1;
        if (nwg > cap) nwg = cap;
        hipLaunchKernelGGL(kern, dim3(nwg, nchunk, n), dim3(256), lw, stream, g);
        return 0;
    };
    int rc = 0;
    bool bf = true;
    for (int i = 0; i < n; ++i) bf = bf && g.pr[i].bf != 0;
    if (bf) {
        if constexpr (CINC >= 8) {
            if (kind == 1) rc = go(&conv3x3_wgrad_cl_kernel<CINC, COUT, 1>, 3);
            else if (kind == 2) rc = go(&conv3x3_wgrad_cl_kernel<CINC, COUT, 2>, 4);
            else return PC_EINVAL;
        } else {
            if (kind == 3) rc = go(&conv3x3_wgrad_cl_kernel<CINC, COUT, 3>, 3);
            else return PC_EINVAL;
        }
        if (rc) return rc;
        PC_CHECK_LAUNCH();
        return 0;
    }
    if (kind == 3) rc = go(&conv3x3_wgrad_wave_kernel<CINC, COUT, 3>, 2);
    else if (kind == 1) rc = go(&conv3x3_wgrad_wave_kernel<CINC, COUT, 1>, 0);
    else rc = go(&conv3x3_wgrad_wave_kernel<CINC, COUT, 2>, 1);
    if (rc) return rc;
    PC_CHECK_LAUNCH();
    return 0;
}

template <int CINC, int COUT>
int launch_wgrad(WgradArgs& p, int Cin, float* dw, float* db, int accumulate, void* ws, hipStream_t stream,
                 int* nwg_out = nullptr) {
    using Cfg = WgradCfg<CINC, COUT>;
    int nwg, nchunk;
    const int kind = prepare_wgrad<CINC, COUT>(p, Cin, ws, nwg, nchunk);
    if (kind < 0) return PC_EINVAL;
    if (kind != 0) {
        // A single-problem launch (the first layers: their Cin differs per stream, so they cannot be grouped) gets the
        // workgroups a grouped launch would spread over its problems -- as many partials as the workspace slice holds
        // (it is sized for the largest layer), at most PC_WGRAD_SINGLE_WG; launch_wgrad_wave caps it to one resident round.
        const int64_t room = (int64_t)MAX_WG * (2 * 12 * 256 + 128) / ((int64_t)Cfg::EC * nchunk);
        int want = PC_WGRAD_SINGLE_WG;
        if (want > room) want = (int)room;
        if (want > p.ntiles) want = p.ntiles;
        if (want > nwg) nwg = want;
        WgradGroup g{};
        g.pr[0] = p;
        const int rc = launch_wgrad_wave<CINC, COUT>(g, 1, kind, nwg, nchunk, stream);
        if (rc) return rc;
    } else {
        const size_t ldsb = (size_t)Cfg::LDS_FLOATS * sizeof(float);
        static pc_once_per_device once;
        if (once.need()) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wgrad_kernel<CINC, COUT>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
            if (e != hipSuccess) return (int)e;
            once.mark();
        }
        hipLaunchKernelGGL((conv3x3_wgrad_kernel<CINC, COUT>), dim3(nwg, nchunk), dim3(256), ldsb, stream, p);
        PC_CHECK_LAUNCH();
    }
    if (nwg_out) {            // deferred: the caller batches the reductions of many layers into one launch
        *nwg_out = nwg;
        return 0;
    }
    WreduceArgs r{};
    r.partial = p.partial; r.nwg = nwg; r.nchunk = nchunk; r.dw = dw; r.db = db; r.Cin = Cin; r.accumulate = accumulate;
    const int n_out = COUT * Cin * 9 + COUT;
    hipLaunchKernelGGL((conv3x3_wgrad_reduce_kernel<CINC, COUT>), dim3((n_out + 15) / 16), dim3(256), 0, stream, r);
    PC_CHECK_LAUNCH();
    return 0;
}

// grouped first stage: one launch for all problems when they share the loader kind, else one launch each
template <int CINC, int COUT>
int launch_wgrad_group(WgradArgs* ps, void* const* wss, int n, int Cin, hipStream_t stream, int* nwg_out) {
    WgradGroup g{};
    int nwg = 0, nchunk = 0, kind0 = -1;
    bool same = true;
    for (int i = 0; i < n; ++i) {
        int w, c;
        const int k = prepare_wgrad<CINC, COUT>(ps[i], Cin, wss[i], w, c);
        if (k < 0) return PC_EINVAL;
        if (i == 0) { kind0 = k; nwg = w; nchunk = c; }
        same = same && k == kind0 && w == nwg;
        g.pr[i] = ps[i];
    }
    if (same && kind0 != 0) {
        const int rc = launch_wgrad_wave<CINC, COUT>(g, n, kind0, nwg, nchunk, stream);
        *nwg_out = nwg;
        return rc;
    }
    for (int i = 0; i < n; ++i) {
        const int rc = launch_wgrad<CINC, COUT>(ps[i], Cin, nullptr, nullptr, 0, wss[i], stream, nwg_out);
        if (rc) return rc;
    }
    return 0;
}

constexpr int cinc_of(int cin) { return cin < 16 ? cin : 16; }

// ---- batched second stage: the reductions of ALL layers of a backward pass in one launch ---------------------------------
// (24 conv + 4 transposed-conv reductions per step were 8-12 us each: launch + latency bound; profiles/r1_v3.)
constexpr int RB_MAX = 32;
struct RbEntry {
    const float* partial; float* dw; float* db;
    int nwg, Cin, Cout, kind, accumulate;     // kind 0: conv3x3, 1: convT 2x2 (C = Cin), 2: raw sum of Cin floats
    int dw_co_stride;                         // conv3x3: elements between output channels of dw (0 = Cin * 9)
    int src_cin, src_ci0;                     // conv3x3: channel window of the partials (pc_wgrad_reduce_desc), src_cin = Cin: all
};
// blk0: first workgroup of every entry (prefix sums of ceil(outputs / RB_OUT)): a flat grid -- as (max outputs / 16) x entries, two
// thirds of the workgroups of a step's list had nothing to do.  head: the 8 gradient tensors of the sparse head (kind 3, at most one
// such entry per launch)
struct RbArgs { RbEntry e[RB_MAX]; int blk0[RB_MAX + 1]; int n; float* head[8]; };

// sum of at(w) over this thread's slice of the partial list (w = slice, slice + RB_SL, ...): sixteen independent chains, so that a thread
// has sixteen loads in flight per round (the kernel's time is rounds x loaded memory latency: 512 partials = 2 rounds; 8 with four chains)
constexpr int RB_OUT = 32, RB_SL = 256 / RB_OUT;      // outputs per workgroup (one 128-byte line of every partial) x slices of the list
template <class F>
__device__ __forceinline__ float rb_sum(int slice, int nwg, F at) {
    float s[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) s[k] = 0.f;
    int w = slice;
    for (; w + 15 * RB_SL < nwg; w += 16 * RB_SL) {
#pragma unroll
        for (int k = 0; k < 16; ++k) s[k] += at(w + RB_SL * k);
    }
    for (; w + 3 * RB_SL < nwg; w += 4 * RB_SL) {
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] += at(w + RB_SL * k);
    }
    for (; w < nwg; w += RB_SL) s[0] += at(w);
#pragma unroll
    for (int h = 8; h >= 1; h >>= 1)
#pragma unroll
        for (int k = 0; k < h; ++k) s[k] += s[k + h];
    return s[0];
}

__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const RbArgs a) {
    // workgroups bx, bx + 1 read the two 64-byte halves of the same 128-byte lines of every partial: consecutive ones on the same XCD
    // (its L2 then fetches a line once; spread round-robin over the eight XCDs every line came out of HBM twice)
    const int gb = pc_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    int ei = 0;
    while (ei + 1 < a.n && gb >= a.blk0[ei + 1]) ++ei;
    const RbEntry& q = a.e[ei];
    const int bx = gb - a.blk0[ei];
    __shared__ float red[256];
    const int tid = threadIdx.x;
    const int slice = tid / RB_OUT, o = bx * RB_OUT + (tid % RB_OUT);
    float s0 = 0.f;
    int n_w, n_out;
    if (q.kind == 3) {
        // the head backward's workgroup partials (head.hip, common.h: pc_head_partial_target): 16 consecutive elements x 16 slices of
        // the partial list; the output index is the inverse of the MFMA fragment layout
        n_w = n_out = PC_PE_TOTAL;
        if (o < n_out) s0 = rb_sum(slice, q.nwg, [&](int w) { return q.partial[(int64_t)w * PC_PE_TOTAL + o]; });
        red[tid] = s0;
        __syncthreads();
        if (tid < RB_OUT && o < n_out) {
            float tot = 0.f;
#pragma unroll
            for (int k = 0; k < RB_SL; ++k) tot += red[k * RB_OUT + tid];
            int t, idx;
            pc_head_partial_target(o, t, idx);
            if (t >= 0 && a.head[t]) {
                float* d = a.head[t] + idx;
                *d = q.accumulate ? *d + tot : tot;
            }
        }
        // structural zeros: the second (unused) output row of the last layer -- weight elements 64..127 and bias element 1
        if (bx == 0 && !q.accumulate) {
            if (tid >= 64 && tid < 128 && a.head[6]) a.head[6][tid] = 0.f;
            if (tid == 128 && a.head[7]) a.head[7][1] = 0.f;
        }
        return;
    }
    if (q.kind == 0) {
        const int CINC = q.src_cin < 16 ? q.src_cin : 16;
        const int EC = q.Cout * CINC * 9 + q.Cout;           // compacted partial (WgradCfg::EC)
        n_w = q.Cout * q.Cin * 9;
        n_out = n_w + q.Cout;
        if (o < n_w) {
            const int tap = o % 9, ci = (o / 9) % q.Cin + q.src_ci0, co = o / (9 * q.Cin);
            const int chunk = ci / CINC, cil = ci % CINC;
            const int oc = co * (CINC * 9) + cil * 9 + tap;
            const float* base = q.partial + (int64_t)chunk * q.nwg * EC + oc;
            s0 = rb_sum(slice, q.nwg, [&](int w) { return base[(int64_t)w * EC]; });
        } else if (o < n_out) {
            const float* base = q.partial + q.Cout * CINC * 9 + (o - n_w);
            s0 = rb_sum(slice, q.nwg, [&](int w) { return base[(int64_t)w * EC]; });
        }
    } else if (q.kind == 2) {
        // raw sum of Cin floats per partial (up_bwd.hip: composed-weight gradient accumulators + border sums; the chain-rule
        // launch that follows turns the total into parameter gradients)
        n_w = n_out = q.Cin;
        if (o < n_w) s0 = rb_sum(slice, q.nwg, [&](int w) { return q.partial[(int64_t)w * q.Cin + o]; });
    } else {
        const int C = q.Cin, NBK = C / 4, E = NBK * 256 + NBK * 64;
        n_w = C * C * 4;
        n_out = n_w + C;
        if (o < n_w) {
            const int ci = o / (4 * C), ng = o % (4 * C);
            const int e = (((ng >> 4) * 64) + (ci >> 2) * 16 + (ng & 15)) * 4 + (ci & 3);
            s0 = rb_sum(slice, q.nwg, [&](int w) { return q.partial[(int64_t)w * E + e]; });
        } else if (o < n_out) {
            // bias gradient of a transposed conv: 16 elements of every partial.  Four partials per round (64 loads in flight; eight measured slower: 34 us): as one
            // partial per round this tail was nwg / 8 = 64 dependent memory round trips of ONE workgroup -- 7 - 10 of the launch's 28 us in
            // the bf16 step, whose four transposed-conv entries the fp32 step (composed Up blocks) does not have
            const int co = o - n_w;
            auto at = [&](int w) {
                const float* pq = q.partial + (int64_t)w * E + NBK * 256;
                float t = 0.f;
#pragma unroll
                for (int ab = 0; ab < 4; ++ab) {
                    const int ng = co * 4 + ab;
#pragma unroll
                    for (int lk = 0; lk < 4; ++lk) t += pq[(ng >> 4) * 64 + lk * 16 + (ng & 15)];
                }
                return t;
            };
            float s4[4] = {0.f, 0.f, 0.f, 0.f};
            int w = slice;
            for (; w + 3 * RB_SL < q.nwg; w += 4 * RB_SL) {
#pragma unroll
                for (int k = 0; k < 4; ++k) s4[k] += at(w + RB_SL * k);
            }
            for (; w < q.nwg; w += RB_SL) s4[0] += at(w);
            s0 = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        }
    }
    red[tid] = s0;
    __syncthreads();
    if (tid < RB_OUT && o < n_out) {
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < RB_SL; ++k) tot += red[k * RB_OUT + tid];
        int64_t od = o;
        if (q.kind == 0 && q.dw_co_stride > 0 && o < n_w) od = (int64_t)(o / (q.Cin * 9)) * q.dw_co_stride + o % (q.Cin * 9);
        float* dstp = o < n_w ? q.dw + od : q.db + (o - n_w);
        if (o >= n_w && q.db == nullptr) return;
        *dstp = (q.accumulate && q.kind != 2) ? *dstp + tot : tot;
    }
}

}  // namespace

extern "C" int64_t pc_conv3x3_wgrad_ws_bytes(int Cin, int Cout) {
    // nchunk * nwg <= MAX_WG partials of E floats; E <= 2*12*256 + 128
    (void)Cin; (void)Cout;
    return (int64_t)MAX_WG * (2 * 12 * 256 + 128) * sizeof(float);
}

extern "C" int pc_wgrad_reduce_batch(int n, const pc_wgrad_reduce_desc* d, void* stream) {
    if (n < 0 || (n > 0 && !d)) return PC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    for (int base = 0; base < n; base += RB_MAX) {
        RbArgs a{};
        const int m = n - base < RB_MAX ? n - base : RB_MAX;
        bool have_head = false;
        a.n = m;
        a.blk0[0] = 0;
        for (int i = 0; i < m; ++i) {
            const pc_wgrad_reduce_desc& s = d[base + i];
            if (!s.partial || !s.dw || s.nwg < 1) return PC_EINVAL;
            int n_out;
            if (s.kind == 3) {
                // dw: HOST array of the head's 8 gradient tensors (device pointers, NULL = skip), order of pc_head_bwd's dhw
                if (have_head) return PC_EINVAL;
                have_head = true;
                float* const* hp = reinterpret_cast<float* const*>(s.dw);
                for (int t = 0; t < 8; ++t) a.head[t] = hp[t];
                a.e[i] = RbEntry{s.partial, nullptr, nullptr, s.nwg, 0, 0, 3, s.accumulate, 0, 0, 0};
                n_out = PC_PE_TOTAL;
            } else {
                const int src_cin = (s.kind == 0 && s.src_cin > 0) ? s.src_cin : s.Cin;
                if (s.kind == 0 && (s.src_ci0 < 0 || (s.src_cin > 0 && s.src_ci0 + s.Cin > s.src_cin) || (s.src_cin <= 0 && s.src_ci0 != 0))) return PC_EINVAL;
                a.e[i] = RbEntry{s.partial, s.dw, s.db, s.nwg, s.Cin, s.Cout, s.kind, s.accumulate, s.dw_co_stride, src_cin, s.kind == 0 ? s.src_ci0 : 0};
                n_out = s.kind == 0 ? s.Cout * s.Cin * 9 + s.Cout : (s.kind == 2 ? s.Cin : s.Cin * s.Cin * 4 + s.Cin);
            }
            a.blk0[i + 1] = a.blk0[i] + (n_out + RB_OUT - 1) / RB_OUT;
        }
        hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(a.blk0[m]), dim3(256), 0, st, a);
        PC_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int pc_conv3x3_wgrad_partial_group(int n, const pc_conv_wgrad_desc* d, int B, int H, int W, int Cin, int Cout,
                                              int* nwg_out, void* stream) {
    if (n < 1 || n > PC_MAX_GROUP || !d || !nwg_out) return PC_EINVAL;
    WgradArgs ps[PC_MAX_GROUP];
    void* wss[PC_MAX_GROUP];
    for (int i = 0; i < n; ++i) {
        if (!d[i].a || !d[i].g || !d[i].ws) return PC_EINVAL;
        ps[i] = WgradArgs{};
        ps[i].a = *d[i].a;
        if (d[i].b) ps[i].b = *d[i].b;
        ps[i].g = *d[i].g;
        if (ps[i].a.C + ps[i].b.C != Cin || d[i].g->C != Cout) return PC_EINVAL;
        ps[i].B = B; ps[i].H = H; ps[i].W = W;
        wss[i] = d[i].ws;
    }
    hipStream_t st = (hipStream_t)stream;
#define PC_CASE(ci, co) \
    if (Cin == ci && Cout == co) return launch_wgrad_group<cinc_of(ci), co>(ps, wss, n, Cin, st, nwg_out);
    PC_CASE(2, 8) PC_CASE(4, 8) PC_CASE(8, 8) PC_CASE(16, 8) PC_CASE(32, 8) PC_CASE(8, 16) PC_CASE(16, 16)
#undef PC_CASE
    return PC_EINVAL;
}

extern "C" int pc_conv3x3_wgrad_partial(const pc_src* a, const pc_src* b, const pc_src* g, void* ws, int B, int H, int W,
                                        int Cin, int Cout, int* nwg_out, void* stream) {
    if (!a || !g || !ws || !nwg_out) return PC_EINVAL;
    WgradArgs p{};
    p.a = *a;
    if (b) p.b = *b;
    p.g = *g;
    if (p.a.C + p.b.C != Cin || g->C != Cout) return PC_EINVAL;
    p.B = B; p.H = H; p.W = W;
    hipStream_t st = (hipStream_t)stream;
#define PC_CASE(ci, co) \
    if (Cin == ci && Cout == co) return launch_wgrad<cinc_of(ci), co>(p, Cin, nullptr, nullptr, 0, ws, st, nwg_out);
    PC_CASE(2, 8) PC_CASE(4, 8) PC_CASE(8, 8) PC_CASE(16, 8) PC_CASE(32, 8) PC_CASE(8, 16) PC_CASE(16, 16)
#undef PC_CASE
    return PC_EINVAL;
}

extern "C" int pc_conv3x3_wgrad(const pc_src* a, const pc_src* b, const pc_src* g, float* dw, float* db, int accumulate,
                                void* ws, int B, int H, int W, int Cin, int Cout, void* stream) {
    if (!a || !g || !dw || !ws) return PC_EINVAL;
    WgradArgs p{};
    p.a = *a;
    if (b) p.b = *b;
    p.g = *g;
    if (p.a.C + p.b.C != Cin || g->C != Cout) return PC_EINVAL;
    p.B = B; p.H = H; p.W = W;
    hipStream_t st = (hipStream_t)stream;
#define PC_CASE(ci, co) \
    if (Cin == ci && Cout == co) return launch_wgrad<cinc_of(ci), co>(p, Cin, dw, db, accumulate, ws, st);
    PC_CASE(2, 8) PC_CASE(4, 8) PC_CASE(8, 8) PC_CASE(16, 8) PC_CASE(32, 8) PC_CASE(8, 16) PC_CASE(16, 16)
#undef PC_CASE
    return PC_EINVAL;
}
