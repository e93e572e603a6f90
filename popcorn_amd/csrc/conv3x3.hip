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
// of every MFMA).  The 16-output layers use the plain im2col instead (N16 below: N = 16 channels, K = 4 consecutive
// (ci, dy, dx) of a chunk's 72): every issued K-slot is a tap.
//
// Data movement: a workgroup (4 waves) owns a 32 x 16 output tile; the (CHUNK x 18 x 40) input halo tile is staged
// in LDS with row stride 48 (= 16 mod 32 banks), which makes the A-operand read -- lane (i = x, k = row) ->
// lds[ci][2*rp + k][x + dx] -- bank-conflict free.  Weights live in registers as ready-made B fragments for the
// whole kernel; workgroups are persistent over tiles (XCD-aware tile order).
//
// The tile loop is a software pipeline: every 16-lane group owns one (channel, row-range) of the halo tile and moves
// its rows as aligned 16-byte segments; the loads of tile t+1 are issued into registers before the MFMA phase of tile
// t and written to LDS after it, so HBM latency hides under MFMA inside one workgroup.  The first version spent ~1200
// mostly scalar/branch instructions per tile per wave on loader index arithmetic and was instruction-issue bound at
// 4x the MFMA time (profiles/r1_v0, tools/ablate_conv.py); the hot path below is ~10 VALU per 16-byte segment.
//
// Up to PC_MAX_GROUP independent problems of identical geometry (e.g. the SAR and optical streams, or the frozen
// building extractor next to the trainable U-Net) share one launch: blockIdx.y selects the problem.
#include "common.h"
#include <type_traits>
#include "tile_loader.h"

static int g_conv_dbg = 0, g_conv_max_grid = 0;
static long long* g_conv_ts = nullptr;     // debug: per-workgroup (start, end, hw id) timestamps (tools/conv_timeline.py)
extern "C" void pc_debug_conv(int dbg, int max_grid) { g_conv_dbg = dbg; g_conv_max_grid = max_grid; }
extern "C" void pc_debug_conv_ts(void* buf) { g_conv_ts = (long long*)buf; }

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// bf16 mode (the channels-last kernel below): the MFMA is v_mfma_f32_16x16x32_bf16 with K = 32 = (4 input rows v) x (8 input
// channels): one instruction per (horizontal tap dx, 16-px block) and 8-channel chunk instead of 8 fp32 ones.  The wave's strip
// lives in LDS as [6 rows][48 slots][8 channels] bf16 (one 16-byte slot per pixel), so a pixel operand -- lane (x, v): the 8
// channels of pixel (row v, x + dx) -- is ONE ds_read_b128; weights are an image [dy plane][co][chunk][dx][8 ci] bf16
// with an all-zero plane for the (row, output row) pairs that are not a tap.  Both operands enumerate K as
// slot(lane >> 4, j) = (row, channel j), so the products pair up whatever the hardware's internal K order is.
constexpr int BSLOTS = 48;               // slots per strip row (40 used; == 0 mod 16: the two lane rows of a b128 group do not collide)
constexpr int BWAVE_F = 6 * BSLOTS * 4;  // floats (4 per 16-byte slot) of one wave's bf16 strip

constexpr int TW = 32, TH = 16;          // output tile
constexpr int RS = 48;                   // LDS row stride  (== 16 mod 32)
constexpr int COL0 = 4;                  // LDS column of tile x0 (left halo at COL0-1): keeps float4 stores aligned
constexpr int MAXG = PC_MAX_GROUP;       // problems per launch

enum { MODE_FWD = 0, MODE_DGRAD = 1 };
enum { LD_GENERIC = 0, LD_DIRECT = 1, LD_POOL = 2, LD_REFLECT = 3 };

struct ConvProb {
    pc_src a, b;          // input sources (channels a.C then b.C)
    const float* w;       // weights
    pc_bn bn;             // FWD: this layer's BN; DGRAD: BN of the layer that produced `act`
    const float* act;     // DGRAD: post-ReLU activations of the producer (NULL = plain)
    int64_t act_bstride, act_cstride;
    int act_rstride;
    int act_dtype;
    int act_xstride;
    int fast_a, fast_b;   // pc_src_fast_mode of the two sources (generic loader)
    pc_dst out;
    pc_dst pool_out;      // FWD: 2x2-max-pooled copy of the output (ptr NULL = not wanted)
    const float* dot_w;   // FWD (EPI_DOT): weights of a following 1x1 conv over this layer's 8 channels ...
    pc_dst dot_out;       // ... whose partial sum replaces the output (ptr NULL = ordinary output)
    // FWD with ZC > 0 (pc_conv3x3_up_fwd_group): the up-sampled half of an Up block's concatenated input, never materialised --
    // z is the LOW-resolution map (ZC channels, H/2 x W/2) the transposed conv would have up-sampled; wz / tb come from
    // compose_up_kernel (composed 2x2-neighbourhood weights per output parity; bias-through-the-taps table)
    const float* z; int64_t z_bs, z_cs; int z_rs;
    const float* wz; const float* tb;
    // channels-last bf16, CIN == 8: w is [COUT][w_cin][3][3] over input channels [w_ci0, w_ci0 + w_cin) (w_cin == 0: full weight)
    int w_ci0, w_cin;
    // EPI_UPT: the ConvTranspose2d(8, 8, 2, stride 2) that consumes this layer's output (Up.up, networks.py:302-306), applied in the
    // epilogue: upt_out = (B, 8, 2H, 2W) channels-last bf16
    const float* upt_w; const float* upt_b;
    pc_dst upt_out;
};

struct ConvArgs {
    ConvProb pr[MAXG];
    int w_co_stride;      // element stride between output channels in w
    int w_ci_stride;      // element stride between input channels in w
    int w_flip;           // 1: tap index 8 - t (dgrad)
    int relu;             // FWD
    int pool;             // DGRAD: max-pool backward scatter into a 2x resolution output
    int accumulate;       // DGRAD: out += instead of out =
    int vec_ok;           // outputs (and act) are 16-byte aligned with W % 4 == 0: vector epilogue allowed
    int B, H, W;
    int tiles_x, tiles_y, ntiles;
    pc_fastdiv div_tx, div_tpi;   // by tiles_x, by tiles per image
    int dbg;              // ablation switches (tools/ablate_conv.py): 1 skip loader, 2 skip MFMA, 4 skip stores
    long long* ts;        // debug timeline buffer (8 slots per workgroup) or NULL
};

// Wave-private strips.  Each wave owns a 32 x 4 output strip (4 MFMA units), stages its own (CHUNK x 6 x 40) halo
// strip in a private LDS region and runs its own software pipeline -- there is NO workgroup barrier in the loop: DS
// operations of one wave execute in order, so "ds_write strip t+1" behind "ds_read strip t" needs no synchronisation.
// Waves therefore drift apart and the hardware interleaves one wave's MFMA phase with the others' loads, LDS writes
// and epilogue stores.  (The workgroup-tile version kept all waves of a CU in lock-step through its two barriers per
// tile: MFMA pipe 42 % busy with waves parked in issue stalls, total time == sum of the phases; tools/ablate_conv.py.)
// The price is a 6/4 instead of 18/16 row halo, served by L2.
constexpr int SROWS = 6;                 // input rows of a 4-row strip
#ifndef POPCORN_CONV_P2
#define POPCORN_CONV_P2 0
#endif
constexpr bool CONV_P2 = POPCORN_CONV_P2 != 0;   // build-time A/B switch of the paired-pixel operand mapping (see the kernel)
constexpr int CSW = SROWS * RS;          // channel stride inside a wave's LDS region

// EPI: extra work of the forward vector epilogue, as separate instantiations so that the other shapes keep their register
// count.  EPI_POOL: also write the 2x2-max-pooled output (ConvProb::pool_out).  EPI_DOT: problems with ConvProb::dot_w
// write the 1x1-conv partial sum over their 8 channels instead of the feature map (ConvProb::dot_out).
enum { EPI_NONE = 0, EPI_POOL = 1, EPI_DOT = 2, EPI_POOLBWD = 3, EPI_UPT = 4 };   // EPI_UPT (channels-last bf16, 8 -> 8 forward): also the ConvTranspose2d that follows   // EPI_POOLBWD: DGRAD with the MaxPool2d(2) backward scatter
template <int CIN, int COUT, int MODE, int LD, int EPI, int ZC = 0>
__global__ __launch_bounds__(256) void conv3x3_mfma_kernel(const ConvArgs p) {
    constexpr int CHUNK = CIN < 8 ? CIN : 8;       // 8 channels per LDS stage: 36 KB per workgroup, 4 workgroups per CU
    constexpr int NCHUNK = CIN / CHUNK;
    // ZC > 0: ZC more input "channels" that exist only at HALF the resolution -- the map z that ConvTranspose2d(ZC, ZC, 2, 2) would
    // have up-sampled into the second half of torch.cat([skip, up]) (networks.py:302-318).  conv3x3(convT(z)) is a linear map
    // with a 2 x 2 low-resolution neighbourhood per output-pixel parity (pY, pX); its weights are composed once per step
    // (compose_up_kernel) and every 8 low-resolution channels are one more stage of the strip: a 4-row x 18-column piece of z
    // in the wave's LDS region (a quarter of a regular chunk's bytes) and 64 MFMAs instead of 96 -- K = the 3 low-res rows a row
    // pair touches, one MFMA per (channel, column tap tj, x parity j); the up-sampled map is neither written nor read.
    constexpr int NZ = ZC / 8;                          // composed stages per strip
    constexpr int NST = NCHUNK + NZ;                    // stages per strip
    constexpr int NB = COUT / 8;
    constexpr bool STAGED = LD != LD_GENERIC;
    constexpr int NIT = CHUNK;                           // one 16-byte segment per channel per lane
    constexpr int W_RL = CIN * 3 + 4;                    // weight image: floats per (dy, co) row (== 4 mod 8: spreads banks)
    constexpr int W_DYS = COUT * W_RL + 16;              // ... and per dy plane
    // 16 output channels: plain im2col, N = 16 channels, M = 16 x of ONE row, K = 4 consecutive (ci, dy, dx) of a chunk's 72 --
    // every issued MFMA slot is a tap (the row-pair mapping of the 8-channel blocks spends 12 K-slots on 9 taps)
    constexpr bool N16 = COUT == 16 && CHUNK == 8;
    constexpr int W16_S = CIN * 9 + 1;                   // N16 weight image [co][ci * 9 + tap], odd row stride: 16 channels on 16 banks
    // P2 ("paired pixels", the 8-output layers with full 8-channel chunks): M index i of an MFMA is pixel x = 2 i + j of the
    // 32-px strip row (j = 0, 1: two MFMAs), so that ONE aligned ds_read_b64 feeds two pixels = operands of several (j, dx)
    // MFMAs: 3 LDS reads per 6 MFMAs instead of 6 (tools/mfma_peak.hip: the fp32 matrix pipe sustains 153 TFLOP/s from
    // registers but 97-107 with one ds_read_b32 per MFMA -- the conv kernels' MFMA phase was LDS-operand-issue bound, not power
    // bound).  The strip image becomes [row][channel][40] with a row-plane stride == 32 (mod 64) dwords: the two rows a 32-lane
    // half of a b64 read touches fall on complementary bank halves.
    constexpr bool P2 = (CONV_P2 || ZC > 0) && !N16 && CHUNK == 8;
    static_assert(ZC == 0 || (P2 && MODE == MODE_FWD && LD == LD_DIRECT && EPI == EPI_NONE && COUT == 8), "composed up-sampling stage");
    constexpr int ZROW = 48, ZCH = 4 * ZROW + 16;       // low-res piece: [8 ch][4 rows][48] floats; ZROW == ZCH - 2 ZROW == 16 (mod 32)
    constexpr int WZ_L = 28;                            // composed weight image: [stage][lane][24 (+4 pad)] floats
    constexpr int LCH = P2 ? 40 : CSW;                   // floats between channels of the strip image
    constexpr int LROW = P2 ? 8 * 40 + 32 : RS;          // floats between rows
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const ConvProb& q = p.pr[blockIdx.y];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;      // A: (i, k);  B: (k, n = li);  D: (n = li, rows 4*lk + r)
    const int s_row = li >> 3, col = li & 7;
    if (p.dbg & 8) return;
    if (p.ts && tid == 0) {
        long long* t = p.ts + 8 * (blockIdx.y * gridDim.x + blockIdx.x);
        t[0] = wall_clock64();
        t[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4) | ((long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32);
    }

    constexpr int WAVE_F = P2 ? SROWS * LROW : CHUNK * CSW;   // floats of one wave's LDS region
    float* const wl = lds + wave * WAVE_F;               // this wave's LDS region

    // ---- staged loader: lane = (row r of the 6-row strip, 16-byte segment seg of the 40-float row)
    const int l_r = lane / 10, l_seg = lane - l_r * 10;
    const bool l_act = lane < 60;
    const int CA = q.a.C;
    using act_t = float;
    const act_t* const a_ptr = reinterpret_cast<const act_t*>(q.a.ptr);
    const act_t* const b_ptr = reinterpret_cast<const act_t*>(q.b.ptr);
    const int64_t a_cstr = q.a.cstride, b_cstr = q.b.cstride;
    const int64_t my_bs = q.a.bstride;
    const int my_rs = q.a.rstride;                       // launch_conv guarantees identical layouts for both sources
    // Loads are issued UNCONDITIONALLY (out-of-image segments read a clamped, always-valid address) and the validity
    // mask is applied when the registers are written to LDS: a predicated load is if-converted into load + select,
    // i.e. an s_waitcnt right behind the load and no overlap with the MFMA phase.
    f32x4 R[NIT];
    int rvalid = 0;                                       // how many of the segment's 4 pixels are inside the image (0 .. 4)
    // the sources' extents, pinned in scalar registers: read from the descriptor inside issue() they were three DEPENDENT scalar-memory
    // round trips (s_load + s_waitcnt lgkmcnt(0), which also drains the LDS queue) in front of every strip's prefetch (round 6)
    int aW = q.a.W, aH = q.a.H, bW = q.b.W, bH = q.b.H, pW = p.W, pH = p.H;
    pc_pin(aW); pc_pin(aH); pc_pin(bW); pc_pin(bH); pc_pin(pW); pc_pin(pH);
    auto issue = [&](int ch, int b, int y0, int x0) {
        const int xg = x0 - 4 + 4 * l_seg, y = y0 - 1 + l_r;
        // DIRECT: a source placed at (0, 0) may be SMALLER than the conv domain -- the up-sampled half of an Up block's concatenated input
        // when the skip map has an odd extent (zero F.pad on the bottom / right, networks.py:309-312): rows / columns beyond the extent of
        // the chunk's own source are zeros.  (A chunk never straddles the two sources: the first one has 8 or 16 channels.)
        const int sW = LD == LD_DIRECT ? (ch * CHUNK < CA ? aW : bW) : pW;
        const int sH = LD == LD_DIRECT ? (ch * CHUNK < CA ? aH : bH) : pH;
        const bool ok = l_act && xg >= 0 && xg < sW && (unsigned)y < (unsigned)sH;
        // a row whose width is not a multiple of 4 ends inside a segment: the tail of that segment is read (the row stride is a
        // multiple of 4, so the 16 bytes exist) and masked per pixel when it is written to LDS
        rvalid = ok ? (sW - xg < 4 ? sW - xg : 4) : 0;
        if (LD == LD_REFLECT) {
            // first layer: reflect padding + channel gather; handles its own bounds (segments may straddle x = 0 / W)
            rvalid = l_act ? 4 : 0;
#pragma unroll
            for (int it = 0; it < NIT; ++it)
                R[it] = l_act ? pc_fetch_reflect_seg(q.a, b, ch * CHUNK + it, y, xg, p.H, p.W) : f32x4{0.f, 0.f, 0.f, 0.f};
            // (bf16 mode: the model input is the one operand no producer has rounded -- the pack in commit() rounds it)
        } else if (LD == LD_DIRECT) {
            const int64_t off = ok ? b * my_bs + (int64_t)y * my_rs + xg : 0;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int cg = ch * CHUNK + it;
                const act_t* cp = cg < CA ? a_ptr + cg * a_cstr : b_ptr + (cg - CA) * b_cstr;
                R[it] = pc_ld4(cp + off);
            }
        } else if (LD == LD_POOL) {
            const int64_t off = ok ? b * my_bs + (int64_t)(2 * y) * my_rs + 2 * xg : 0;
            const int rs1 = ok ? my_rs : 0;
            // the second 16-byte piece of the window: beyond the source row when fewer than 3 output pixels of this segment exist
            // (ragged width) -- it is then masked anyway and reads the first piece's address instead
            const int o4 = rvalid > 2 ? 4 : 0;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const act_t* s0 = a_ptr + (ch * CHUNK + it) * a_cstr + off;
                const f32x4 a0 = pc_ld4(s0), a1 = pc_ld4(s0 + o4);
                const f32x4 b0 = pc_ld4(s0 + rs1), b1 = pc_ld4(s0 + rs1 + o4);
                f32x4 v;
                v[0] = fmaxf(fmaxf(a0[0], a0[1]), fmaxf(b0[0], b0[1]));
                v[1] = fmaxf(fmaxf(a0[2], a0[3]), fmaxf(b0[2], b0[3]));
                v[2] = fmaxf(fmaxf(a1[0], a1[1]), fmaxf(b1[0], b1[1]));
                v[3] = fmaxf(fmaxf(a1[2], a1[3]), fmaxf(b1[2], b1[3]));
                R[it] = v;
            }
        }
    };
    auto commit = [&]() {
        if (l_act) {
            float* d = wl + l_r * LROW + 4 * l_seg;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = e < rvalid ? R[it][e] : 0.f;
                *reinterpret_cast<f32x4*>(d + it * LCH) = v;
            }
        }
    };
    auto load_generic = [&](int ch, int b, int y0, int x0) {
        for (int idx = lane; idx < CHUNK * SROWS * 34; idx += 64) {
            const int c = idx % 34, r = (idx / 34) % SROWS, ci = idx / (34 * SROWS);
            const int cg = ch * CHUNK + ci;
            const float v = cg < CA ? pc_fetch(q.a, b, cg, y0 - 1 + r, x0 - 1 + c, p.H, p.W)
                                    : pc_fetch(q.b, b, cg - CA, y0 - 1 + r, x0 - 1 + c, p.H, p.W);
            wl[ci * LCH + r * LROW + (COL0 - 1) + c] = v;
        }
    };

    // ---- composed stage: 8 low-res channels x 4 rows (y0/2 - 1 ..) x 18 columns (x0/2 - 1 ..) = 576 floats: lane = (channel, row,
    // half of the 18 columns), 9 consecutive floats each -- one address per lane, immediate offsets (a generic element index per
    // load cost 50 registers of hoisted address arithmetic)
    float Rz[ZC > 0 ? 9 : 1];
    bool pend_z = false;                                 // the prefetch in flight is a low-res piece (wave-uniform)
    const int z_ch = lane >> 3, z_row = (lane >> 1) & 3, z_half = lane & 1;
    auto issue_z = [&](int zc, int b, int y0, int x0) {
        if constexpr (ZC > 0) {
            const int Hh = p.H >> 1, Wh = p.W >> 1;
            int I = (y0 >> 1) - 1 + z_row;
            I = I < 0 ? 0 : (I >= Hh ? Hh - 1 : I);                            // clamped: masked when it is written to LDS
            const int c0 = (x0 >> 1) - 1 + 9 * z_half;                          // first of the lane's 9 columns (>= -1)
            const float* rowp = q.z + b * q.z_bs + (int64_t)(zc * 8 + z_ch) * q.z_cs + (int64_t)I * q.z_rs;
            Rz[0] = rowp[c0 < 0 ? 0 : c0];
#pragma unroll
            for (int k = 1; k < 8; ++k) Rz[k] = rowp[c0 + k];
            Rz[8] = rowp[c0 + 8 < Wh ? c0 + 8 : Wh - 1];
        }
    };
    auto commit_z = [&](int y0, int x0) {        // the coordinates of the strip whose piece is pending
        if constexpr (ZC > 0) {
            const int Hh = p.H >> 1, Wh = p.W >> 1;
            const bool rok = (unsigned)((y0 >> 1) - 1 + z_row) < (unsigned)Hh;
            const int c0 = (x0 >> 1) - 1 + 9 * z_half;
            float* d = wl + z_ch * ZCH + z_row * ZROW + 9 * z_half;
#pragma unroll
            for (int k = 0; k < 9; ++k) d[k] = rok && (unsigned)(c0 + k) < (unsigned)Wh ? Rz[k] : 0.f;
        }
    };

    // strips: tile-major so that the 4 waves of a workgroup take the 4 strips of one 32 x 16 tile (shared halo rows
    // hit in L1/L2); workgroups walk the tiles in the XCD-aware order.
    int gdim = (int)gridDim.x;                           // (pinned: read from the dispatch packet inside the loop it was one more scalar round trip per strip)
    asm volatile("" : "+s"(gdim));
    const int my_tiles = p.ntiles > (int)blockIdx.x ? (p.ntiles - 1 - (int)blockIdx.x) / gdim + 1 : 0;
    const int nstages = my_tiles * NST;
    auto strip_coords = [&](int stage, int& b, int& y0, int& x0) {
        const int t = blockIdx.x + (stage / NST) * gdim;
        const int tile = pc_xcd_remap(t, p.ntiles);
        b = (int)pc_div((uint32_t)tile, p.div_tpi);
        const int rem = tile - b * p.tiles_x * p.tiles_y;
        const int ty = (int)pc_div((uint32_t)rem, p.div_tx);
        x0 = (rem - ty * p.tiles_x) * TW;
        y0 = ty * TH + 4 * wave;
    };

    // The loads of the first strip are in flight while the weights are staged.
    int b = 0, y0 = 0, x0 = 0;
    if (nstages > 0) {
        strip_coords(0, b, y0, x0);
        if (STAGED && !(p.dbg & 1)) issue(0, b, y0, x0);
    }

    // ---- B fragments: bw[ci][dx][nb] = w[co = nb*8+col][ch*CHUNK + ci][dy = lk - s_row][dx]  (0 when dy is not a tap)
    // The (strided, possibly transposed) weight tensor is copied ONCE per workgroup into an LDS image behind the wave
    // regions, laid out [dy 0..3][co][ci*3+dx] with an all-zero dy = 3 plane for the lanes whose (row, output row) pair
    // is not a tap; a lane's fragments for one 8-channel chunk are then 24 consecutive floats = 6 ds_read_b128.  Only
    // the current chunk's fragments live in registers (24 per 8 output channels whatever CIN is): holding all of them
    // cost 96 registers at CIN = 32 and left 16->8 / 32->8 / 16->16 with two or ONE wave per SIMD.
    // Prologue latency: the weight elements and the raw BN parameters are loaded into registers FIRST (one memory round
    // trip, in flight together with the first strip), the LDS image is zeroed meanwhile, and only then are they consumed
    // -- the prologue used to be three dependent round trips (zero fill + barrier, weights, BN), ~4.6 us of every launch.
    float* const w2 = lds + 4 * WAVE_F;
    float* const wzimg = w2 + (N16 ? 16 * W16_S : 4 * W_DYS);          // ZC > 0: composed weights [stage][lane][WZ_L]
    float wzreg[ZC > 0 ? NZ * 6 : 1];
    if constexpr (ZC > 0) {
#pragma unroll
        for (int k = 0; k < NZ * 6; ++k) wzreg[k] = q.wz[tid + 256 * k];          // NZ * 1536 floats, coalesced
    }
    constexpr int NWR = (COUT * CIN * 9 + 255) / 256;
    float wreg[NWR];
#pragma unroll
    for (int k = 0; k < NWR; ++k) {
        const int e = tid + k * 256;
        const int ec = e < COUT * CIN * 9 ? e : 0;
        const int tap = ec % 9, ci = (ec / 9) % CIN, co = ec / (9 * CIN);
        wreg[k] = q.w[co * p.w_co_stride + ci * p.w_ci_stride + (p.w_flip ? 8 - tap : tap)];
    }
    const bool has_bn = MODE == MODE_FWD || q.act != nullptr;
    float bn_raw[NB][5];            // {conv bias, gamma, var, mean, beta} of channel nb*8 + col
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int c = N16 ? li : nb * 8 + col;          // N16: one channel per lane (slot nb = 0 is the one that is used)
        bn_raw[nb][0] = has_bn && q.bn.conv_bias ? q.bn.conv_bias[c] : 0.f;
        bn_raw[nb][1] = has_bn && q.bn.gamma ? q.bn.gamma[c] : 1.f;
        bn_raw[nb][2] = has_bn && q.bn.gamma ? q.bn.var[c] : 1.f;
        bn_raw[nb][3] = has_bn && q.bn.gamma ? q.bn.mean[c] : 0.f;
        bn_raw[nb][4] = has_bn && q.bn.gamma ? q.bn.beta[c] : 0.f;
    }
    {
        // only the dy = 3 plane has to be zero (the pad floats of the other planes are never read): disjoint from the weight
        // writes below, so one barrier covers both
        if constexpr (!N16)
            for (int e = tid; e < W_DYS; e += 256) w2[3 * W_DYS + e] = 0.f;
        if constexpr (ZC > 0) {
#pragma unroll
            for (int k = 0; k < NZ * 6; ++k) {
                const int e = tid + 256 * k, zc = e / 1536, r = e - zc * 1536;    // global order [stage][lane][24]
                wzimg[zc * 64 * WZ_L + (r / 24) * WZ_L + (r % 24)] = wzreg[k];
            }
        }
#pragma unroll
        for (int k = 0; k < NWR; ++k) {
            const int e = tid + k * 256;
            if (e < COUT * CIN * 9) {
                const int tap = e % 9, ci = (e / 9) % CIN, co = e / (9 * CIN);
                if constexpr (N16) w2[co * W16_S + ci * 9 + tap] = wreg[k];
                else w2[(tap / 3) * W_DYS + co * W_RL + ci * 3 + (tap % 3)] = wreg[k];
            }
        }
        __syncthreads();
    }
    const float* const wlane = w2 + (((unsigned)(lk - s_row) <= 2u) ? lk - s_row : 3) * W_DYS + col * W_RL;
    float bw[CHUNK][3][NB];
    // N16: B fragment of step m = weight (co = li, K-slot 4 * m + lk of the chunk); A = pixel (row + dy, x + dx) of channel ci
    const float* const wlane16 = w2 + li * W16_S + lk;
    int aoff16[18];
#pragma unroll
    for (int m = 0; m < 18; ++m) {
        const int kk = 4 * m + lk, ci = kk / 9, t = kk - 9 * ci, dy = t / 3;
        aoff16[m] = ci * CSW + dy * RS + (t - 3 * dy);
    }
    auto load_bw = [&](int ch) {
        if constexpr (N16) {
#pragma unroll
            for (int m = 0; m < 18; ++m) bw[m / 6][(m % 6) / 2][m % 2] = wlane16[ch * 72 + 4 * m];
        } else if constexpr (CHUNK == 8) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(wlane + nb * 8 * W_RL + ch * 24 + 4 * j);
#pragma unroll
                    for (int e = 0; e < 4; ++e) bw[(4 * j + e) / 3][(4 * j + e) % 3][nb] = t[e];
                }
        } else {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int k = 0; k < CHUNK * 3; ++k) bw[k / 3][k % 3][nb] = wlane[nb * 8 * W_RL + k];
        }
    };
    // ---- per-lane epilogue constants for co = nb*8 + col
    float e_scale[NB], e_shift[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        // folded BN (pc_bn_fold): scale = gamma / sqrt(var + eps); shift = (bias - mean) * scale + beta
        if (has_bn && q.bn.gamma) {
            e_scale[nb] = bn_raw[nb][1] * (1.0f / sqrtf(bn_raw[nb][2] + q.bn.eps));
            e_shift[nb] = (bn_raw[nb][0] - bn_raw[nb][3]) * e_scale[nb] + bn_raw[nb][4];
        } else {
            e_scale[nb] = 1.f;
            e_shift[nb] = has_bn ? bn_raw[nb][0] : 0.f;
        }
    }
    // Consume the BN constants HERE.  They come from global loads issued before the strip loop and are first used in
    // the epilogue inside it; hipcc's waitcnt pass then keeps them "pending" at the loop header on every iteration and,
    // unable to count how many prefetch loads were issued since, emits `s_waitcnt vmcnt(0)` in front of the epilogue --
    // draining the just-issued prefetch loads of the next strip on EVERY strip (load, store and MFMA phases add up
    // instead of overlapping: tools/ablate_conv.py).  An empty asm use forces the one wait to happen before the loop.
    // ZC > 0: the transposed conv's bias reaches the output through every tap whose up-sampled pixel lies inside the image: the full
    // sum S joins the shift, the border rows / columns / corners subtract their missing taps (tb[co] = {R0, R2, C0, C2, T00, T02,
    // T20, T22, S}: sums over the taps of row dy = 0 / 2, column dx = 0 / 2, the four corner taps, all nine)
    float zb[ZC > 0 ? 8 : 1];
    if constexpr (ZC > 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) zb[k] = q.tb[col * 9 + k];
        e_shift[0] += q.tb[col * 9 + 8] * e_scale[0];
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("" : : "v"(zb[k]));
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) asm volatile("" : : "v"(e_scale[nb]), "v"(e_shift[nb]));
    const act_t* const act = reinterpret_cast<const act_t*>(q.act);
    act_t* const outp = reinterpret_cast<act_t*>(q.out.ptr);
    const int64_t o_bs = q.out.bstride, o_cs = q.out.cstride, a_bs = q.act_bstride, a_cs = q.act_cstride;
    const int o_rs = q.out.rstride, a_rs = q.act_rstride;

    // deferred epilogue: the results of strip t are stored while strip t+1 is in its MFMA phase
    f32x4 pacc[4][NB];
    int eb = 0, ey0 = 0, ex0 = 0;
    bool have_prev = false;
    // x offset (inside the 32-px strip row) of the four consecutive pixels a lane holds for unit u
    auto xu = [&](int u) { return P2 ? 8 * lk + 4 * (u & 1) : (u & 1) * 16 + 4 * lk; };
    auto epilogue = [&]() {
        // lane holds (co = nb*8+col, y = ey0 + 2*(u>>1) + s_row, x = ex0 + xu(u) + r), r = 0..3
        if (ey0 >= p.H) return;
        constexpr bool POOLB = MODE == MODE_DGRAD && EPI == EPI_POOLBWD;   // p.pool, as a compile-time property
        if constexpr (N16) {
            // lane holds (co = li, y = ey0 + r, x = ex0 + h*16 + 4*lk + e) in pacc[r][h][e]
            const float sc = e_scale[0], sh = e_shift[0];
            const bool vec = p.vec_ok && (ey0 + 4 <= p.H) && (ex0 + TW <= p.W);
            if (POOLB) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int y = ey0 + r, x = ex0 + h * 16 + 4 * lk;
                        if (y >= p.H || x >= p.W) continue;
                        const f32x4 v = pacc[r][h];
                        const act_t* a0 = act + eb * a_bs + li * a_cs + (int64_t)(2 * y) * a_rs + 2 * x;
                        act_t* o0 = outp + eb * o_bs + li * o_cs + (int64_t)(2 * y) * o_rs + 2 * x;
                        if (vec) {
                            f32x4 A[2][2], O[2][2];
#pragma unroll
                            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                                for (int hh = 0; hh < 2; ++hh) {
                                    A[rr][hh] = pc_ld4(a0 + rr * a_rs + 4 * hh);
                                    O[rr][hh] = pc_ld4(o0 + rr * o_rs + 4 * hh);
                                }
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int hh = e >> 1, c = (e & 1) * 2;
                                const float w00 = A[0][hh][c], w01 = A[0][hh][c + 1], w10 = A[1][hh][c], w11 = A[1][hh][c + 1];
                                int am = 0;
                                float m = w00;
                                if (w01 > m) { m = w01; am = 1; }
                                if (w10 > m) { m = w10; am = 2; }
                                if (w11 > m) { m = w11; am = 3; }
                                const float g = m > 0.f ? v[e] * sc : 0.f;
                                O[0][hh][c] += am == 0 ? g : 0.f;
                                O[0][hh][c + 1] += am == 1 ? g : 0.f;
                                O[1][hh][c] += am == 2 ? g : 0.f;
                                O[1][hh][c + 1] += am == 3 ? g : 0.f;
                            }
#pragma unroll
                            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                                for (int hh = 0; hh < 2; ++hh) pc_st4(o0 + rr * o_rs + 4 * hh, O[rr][hh]);
                        } else {
                            const act_t* a1 = a0 + a_rs;
                            act_t* o1 = o0 + o_rs;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                if (x + e >= p.W) continue;
                                const float w00 = pc_ld1(a0 + 2 * e), w01 = pc_ld1(a0 + 2 * e + 1), w10 = pc_ld1(a1 + 2 * e), w11 = pc_ld1(a1 + 2 * e + 1);
                                int am = 0;
                                float m = w00;
                                if (w01 > m) { m = w01; am = 1; }
                                if (w10 > m) { m = w10; am = 2; }
                                if (w11 > m) { m = w11; am = 3; }
                                const float g = m > 0.f ? v[e] * sc : 0.f;
                                pc_st1(o0 + 2 * e, pc_ld1(o0 + 2 * e) + (am == 0 ? g : 0.f));
                                pc_st1(o0 + 2 * e + 1, pc_ld1(o0 + 2 * e + 1) + (am == 1 ? g : 0.f));
                                pc_st1(o1 + 2 * e, pc_ld1(o1 + 2 * e) + (am == 2 ? g : 0.f));
                                pc_st1(o1 + 2 * e + 1, pc_ld1(o1 + 2 * e + 1) + (am == 3 ? g : 0.f));
                            }
                        }
                    }
                return;
            }
#pragma unroll
            for (int rp = 0; rp < 2; ++rp)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x4 vv[2];
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const int r = 2 * rp + s, y = ey0 + r, x = ex0 + h * 16 + 4 * lk;
                        f32x4 v = pacc[r][h];
                        vv[s] = v;
                        if (y >= p.H || x >= p.W) continue;
                        act_t* op = outp + eb * o_bs + li * o_cs + (int64_t)y * o_rs + x;
                        const act_t* ap = act ? act + eb * a_bs + li * a_cs + (int64_t)y * a_rs + x : nullptr;
                        if (vec) {
                            if (MODE == MODE_FWD) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const float o = v[e] * sc + sh;
                                    v[e] = p.relu ? fmaxf(o, 0.f) : o;
                                }
                            } else {
                                if (ap) {
                                    const f32x4 a4 = pc_ld4(ap);
#pragma unroll
                                    for (int e = 0; e < 4; ++e) v[e] = a4[e] > 0.f ? v[e] * sc : 0.f;
                                }
                                if (p.accumulate) {
                                    const f32x4 o4 = pc_ld4(op);
#pragma unroll
                                    for (int e = 0; e < 4; ++e) v[e] += o4[e];
                                }
                            }
                            pc_st4(op, v);
                            vv[s] = v;
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                if (x + e >= p.W) continue;
                                float o = v[e];
                                if (MODE == MODE_FWD) {
                                    o = o * sc + sh;
                                    o = p.relu ? fmaxf(o, 0.f) : o;
                                } else {
                                    if (ap) o = pc_ld1(ap + e) > 0.f ? o * sc : 0.f;
                                    if (p.accumulate) o += pc_ld1(op + e);
                                }
                                pc_st1(op + e, o);
                            }
                        }
                    }
                    if (EPI == EPI_POOL && MODE == MODE_FWD && q.pool_out.ptr && vec) {
                        // MaxPool2d(2): both rows of the pair and both x of a pair are in the lane
                        act_t* pp = reinterpret_cast<act_t*>(q.pool_out.ptr) + eb * q.pool_out.bstride + li * q.pool_out.cstride +
                                    (int64_t)((ey0 >> 1) + rp) * q.pool_out.rstride + (ex0 >> 1) + h * 8 + 2 * lk;
                        pc_st2(pp, fmaxf(fmaxf(vv[0][0], vv[0][1]), fmaxf(vv[1][0], vv[1][1])),
                               fmaxf(fmaxf(vv[0][2], vv[0][3]), fmaxf(vv[1][2], vv[1][3])));
                    }
                }
            return;
        }
        if (POOLB && p.vec_ok && (ey0 + 4 <= p.H) && (ex0 + TW <= p.W)) {
            // MaxPool2d(2) backward on an interior strip of aligned tensors: the lane's four pooled pixels cover 8 x 2
            // full-resolution pixels = two 16-byte pieces per row of `act` and of the accumulated output (the scalar path
            // below issues 12 four-byte accesses per pooled pixel; these two launches were 127 us of the step)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int y = ey0 + 2 * (u >> 1) + s_row, x = ex0 + xu(u);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int co = nb * 8 + col;
                    const f32x4 v = pacc[u][nb];
                    const act_t* a0 = act + eb * a_bs + co * a_cs + (int64_t)(2 * y) * a_rs + 2 * x;
                    act_t* o0 = outp + eb * o_bs + co * o_cs + (int64_t)(2 * y) * o_rs + 2 * x;
                    f32x4 A[2][2], O[2][2];
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            A[rr][h] = pc_ld4(a0 + rr * a_rs + 4 * h);
                            O[rr][h] = pc_ld4(o0 + rr * o_rs + 4 * h);
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
                        const float g = m > 0.f ? v[r] * e_scale[nb] : 0.f;
                        O[0][h][e] += am == 0 ? g : 0.f;
                        O[0][h][e + 1] += am == 1 ? g : 0.f;
                        O[1][h][e] += am == 2 ? g : 0.f;
                        O[1][h][e + 1] += am == 3 ? g : 0.f;
                    }
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                        for (int h = 0; h < 2; ++h) pc_st4(o0 + rr * o_rs + 4 * h, O[rr][h]);
                }
            }
            return;
        }
        const bool full = !POOLB && p.vec_ok && (ey0 + 4 <= p.H) && (ex0 + TW <= p.W);
        if (full) {
            // interior strip, aligned tensors: no bounds checks, 16-byte accesses only
            act_t* ob = outp + eb * o_bs + col * o_cs + (int64_t)(ey0 + s_row) * o_rs + ex0;
            const act_t* ab = act ? act + eb * a_bs + col * a_cs + (int64_t)(ey0 + s_row) * a_rs + ex0 : nullptr;
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    f32x4 v = pacc[u][nb];
                    act_t* op = ob + nb * 8 * o_cs + (int64_t)((u >> 1) * 2) * o_rs + xu(u);
                    if (MODE == MODE_FWD) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float o = v[r] * e_scale[nb] + e_shift[nb];
                            v[r] = p.relu ? fmaxf(o, 0.f) : o;
                        }
                    } else {
                        if (ab) {
                            const f32x4 a4 = pc_ld4(ab + nb * 8 * a_cs + (int64_t)((u >> 1) * 2) * a_rs + xu(u));
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = a4[r] > 0.f ? v[r] * e_scale[nb] : 0.f;
                        }
                        if (p.accumulate) {
                            const f32x4 o4 = pc_ld4(op);
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] += o4[r];
                        }
                    }
                    if (EPI == EPI_DOT && MODE == MODE_FWD && q.dot_w) {
                        // sum over the 8 channels = the 8 lanes `col` of a 16-lane group half; lane col == 0 stores
                        const float wl = q.dot_w[nb * 8 + col];
                        f32x4 t;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            t[r] = pc_sum8(__fmul_rn(v[r], wl));      // (rounded product: as `v * wl + dpp(..)` the first step contracts into an fma)
                        }
                        if (col == 0)
                            *reinterpret_cast<f32x4*>(q.dot_out.ptr + eb * q.dot_out.bstride +
                                                      (int64_t)(ey0 + s_row + (u >> 1) * 2) * q.dot_out.rstride + ex0 + xu(u)) = t;
                        continue;
                    }
                    pc_st4(op, v);
                    if (EPI == EPI_POOL && MODE == MODE_FWD && q.pool_out.ptr) {
                        // MaxPool2d(2): the x pairs are in the lane, the row pair (s_row 0 / 1) sits 8 lanes apart
                        float m0 = fmaxf(v[0], v[1]), m1 = fmaxf(v[2], v[3]);
                        m0 = fmaxf(m0, pc_lane_xor8(m0));
                        m1 = fmaxf(m1, pc_lane_xor8(m1));
                        if (s_row == 0) {
                            act_t* pp = reinterpret_cast<act_t*>(q.pool_out.ptr) + eb * q.pool_out.bstride + (nb * 8 + col) * q.pool_out.cstride +
                                        (int64_t)((ey0 >> 1) + (u >> 1)) * q.pool_out.rstride + (ex0 >> 1) + (xu(u) >> 1);
                            pc_st2(pp, m0, m1);
                        }
                    }
                }
            return;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int y = ey0 + 2 * (u >> 1) + s_row;
            const int x = ex0 + xu(u);
            if (y >= p.H || x >= p.W) continue;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int co = nb * 8 + col;
                const f32x4 v = pacc[u][nb];
                if (MODE == MODE_FWD) {
                    if (EPI == EPI_DOT && q.dot_w) {
                        // partial strip (ragged width / height): the 1x1 partial logit per element; the 8 lanes `col` of a group hold
                        // the same pixels, so they are active together and the cross-lane sum is complete
                        const float wl = q.dot_w[co];
                        float* dp = q.dot_out.ptr + eb * q.dot_out.bstride + (int64_t)y * q.dot_out.rstride + x;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float o = v[r] * e_scale[nb] + e_shift[nb];
                            float t = __fmul_rn(p.relu ? fmaxf(o, 0.f) : o, wl);
                            t = pc_sum8(t);
                            if (col == 0 && x + r < p.W) dp[r] = t;
                        }
                        continue;
                    }
                    act_t* op = outp + eb * o_bs + co * o_cs + (int64_t)y * o_rs + x;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float o = v[r] * e_scale[nb] + e_shift[nb];
                        if (x + r < p.W) pc_st1(op + r, p.relu ? fmaxf(o, 0.f) : o);
                    }
                } else if (!POOLB) {
                    act_t* op = outp + eb * o_bs + co * o_cs + (int64_t)y * o_rs + x;
                    const act_t* ap = act ? act + eb * a_bs + co * a_cs + (int64_t)y * a_rs + x : nullptr;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (x + r < p.W) {
                            float o = v[r];
                            if (ap) o = pc_ld1(ap + r) > 0.f ? o * e_scale[nb] : 0.f;
                            if (p.accumulate) o += pc_ld1(op + r);
                            pc_st1(op + r, o);
                        }
                    }
                } else {
                    // MaxPool2d(2) backward: (y,x) is a pooled coordinate; route to the first arg-max of the window.
                    const act_t* a0 = act + eb * a_bs + co * a_cs + (int64_t)(2 * y) * a_rs;
                    const act_t* a1 = a0 + a_rs;
                    act_t* o0 = outp + eb * o_bs + co * o_cs + (int64_t)(2 * y) * o_rs;
                    act_t* o1 = o0 + o_rs;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int xx = x + r;
                        if (xx < p.W) {
                            const float w00 = pc_ld1(a0 + 2 * xx), w01 = pc_ld1(a0 + 2 * xx + 1), w10 = pc_ld1(a1 + 2 * xx), w11 = pc_ld1(a1 + 2 * xx + 1);
                            int am = 0;
                            float m = w00;
                            if (w01 > m) { m = w01; am = 1; }
                            if (w10 > m) { m = w10; am = 2; }
                            if (w11 > m) { m = w11; am = 3; }
                            const float g = m > 0.f ? v[r] * e_scale[nb] : 0.f;
                            pc_st1(o0 + 2 * xx, pc_ld1(o0 + 2 * xx) + (am == 0 ? g : 0.f));
                            pc_st1(o0 + 2 * xx + 1, pc_ld1(o0 + 2 * xx + 1) + (am == 1 ? g : 0.f));
                            pc_st1(o1 + 2 * xx, pc_ld1(o1 + 2 * xx) + (am == 2 ? g : 0.f));
                            pc_st1(o1 + 2 * xx + 1, pc_ld1(o1 + 2 * xx + 1) + (am == 3 ? g : 0.f));
                        }
                    }
                }
            }
        }
    };

    f32x4 acc[4][NB];
    for (int stage = 0; stage < nstages; ++stage) {
        const int ch = stage % NST;
        if (!(p.dbg & 1)) {
            if (ZC > 0 && pend_z) commit_z(y0, x0);
            else if (STAGED) commit();
            else load_generic(ch, b, y0, x0);
        }
        int nb_ = b, ny0 = y0, nx0 = x0;
        if (stage + 1 < nstages) {
            const int nch = (stage + 1) % NST;
            if (nch == 0) strip_coords(stage + 1, nb_, ny0, nx0);
            if (STAGED && !(p.dbg & 1)) {
                if (ZC > 0 && nch >= NCHUNK) { issue_z(nch - NCHUNK, nb_, ny0, nx0); pend_z = true; }
                else { issue(nch, nb_, ny0, nx0); pend_z = false; }
            }
        }
        if (have_prev) {
            if (!(p.dbg & 4)) epilogue();
            have_prev = false;
        }
        if (ch == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[u][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (ZC > 0 && ch >= NCHUNK) {
            if constexpr (ZC > 0) {
                if (!(p.dbg & 2)) {
                    // composed stage: B[k = low-res row lk][n = (s, co)] of (channel ci, column tap tj, x parity j) = register 4 ci + 2 tj + j
                    // K-slot q = 4 m + lk (m = 0..5) = (channel q / 3, low-res row q % 3 of the pair's three): 24 (channel, row) slots in
                    // 6 MFMAs per (column tap tj, x parity j, row pair) = 48 per stage (one K-slot per row with a 4th unused slot took 64);
                    // the two slots a 32-lane half reads together are 16 banks apart
                    const float* wzl = wzimg + (ch - NCHUNK) * 64 * WZ_L + lane * WZ_L;
                    const float* zl = wl + li;                              // A: column li + tj + j of (channel, row) slot lk of step m
#pragma unroll
                    for (int m = 0; m < 6; ++m) {
                        const int qs = 4 * m + lk, zci = qs / 3, zv_ = qs - 3 * zci;
                        const f32x4 bz = *reinterpret_cast<const f32x4*>(wzl + 4 * m);        // [2 tj + j] of this step
                        float zv[2][3];
#pragma unroll
                        for (int rp = 0; rp < 2; ++rp)
#pragma unroll
                            for (int c3 = 0; c3 < 3; ++c3) zv[rp][c3] = zl[zci * ZCH + (rp + zv_) * ZROW + c3];
#pragma unroll
                        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                            for (int j = 0; j < 2; ++j)
#pragma unroll
                                for (int rp = 0; rp < 2; ++rp)
                                    acc[rp * 2 + j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(zv[rp][tj + j], bz[2 * tj + j], acc[rp * 2 + j][0], 0, 0, 0);
                    }
                }
            }
        } else if (!(p.dbg & 2)) {
            load_bw(ch);                    // re-read every stage, also when CIN == CHUNK: not live across the epilogue
            const float* lrow = wl + lk * RS + (COL0 - 1) + li;
            if constexpr (P2) {
                // lane (li, lk): input row 2 rp + lk, columns 2 + 2 li .. 7 + 2 li (three aligned b64) = pixels x0 - 2 + 2 li ..;
                // operand of (j, dx) = pixel x0 - 1 + 2 li + j + dx = element 1 + j + dx of the six
                const float* l2 = wl + lk * LROW + 2 + 2 * li;
#pragma unroll
                for (int ci = 0; ci < CHUNK; ++ci) {
#pragma unroll
                    for (int rp = 0; rp < 2; ++rp) {
                        const float* src = l2 + ci * LCH + rp * 2 * LROW;
                        const float2 q0 = *reinterpret_cast<const float2*>(src), q1 = *reinterpret_cast<const float2*>(src + 2),
                                     q2 = *reinterpret_cast<const float2*>(src + 4);
                        const float sv[4] = {q0.y, q1.x, q1.y, q2.x};
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                            for (int j = 0; j < 2; ++j)
#pragma unroll
                                for (int nb = 0; nb < NB; ++nb)
                                    acc[rp * 2 + j][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(sv[j + dx], bw[ci][dx][nb], acc[rp * 2 + j][nb], 0, 0, 0);
                    }
                }
            } else if constexpr (N16) {
                // acc[r][h]: output row y0 + r, x half h, channel li
                const float* l16 = wl + (COL0 - 1) + li;
#pragma unroll
                for (int m = 0; m < 18; ++m) {
                    float av[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) av[u] = l16[aoff16[m] + (u >> 1) * RS + (u & 1) * 16];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        acc[u >> 1][u & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bw[m / 6][(m % 6) / 2][m % 2], acc[u >> 1][u & 1], 0, 0, 0);
                }
            } else
#pragma unroll
            for (int ci = 0; ci < CHUNK; ++ci) {
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    float av[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) av[u] = lrow[ci * CSW + (u >> 1) * 2 * RS + (u & 1) * 16 + dx];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[u][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bw[ci][dx][nb], acc[u][nb], 0, 0, 0);
                }
            }
        }
        if (ch == NST - 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    if constexpr (P2) {
                        // acc[rp*2 + j][nb][r] = pixel 2 (4 lk + r) + j  ->  pacc[rp*2 + h][nb][e] = pixel 8 lk + 4 h + e
                        const int rp = u >> 1, h = u & 1;
                        const f32x4 a0 = acc[rp * 2][nb], a1 = acc[rp * 2 + 1][nb];
                        pacc[u][nb] = f32x4{a0[2 * h], a1[2 * h], a0[2 * h + 1], a1[2 * h + 1]};
                        if constexpr (ZC > 0) {
                            // border pixels: the taps whose up-sampled pixel falls outside the image carry no bias
                            const int Y = y0 + 2 * rp + s_row, X0 = x0 + 8 * lk + 4 * h;
                            const bool top = Y == 0, bot = Y == p.H - 1;
                            const float rowt = top ? zb[0] : (bot ? zb[1] : 0.f);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const bool lft = X0 + e == 0, rgt = X0 + e == p.W - 1;
                                float c = rowt + (lft ? zb[2] : (rgt ? zb[3] : 0.f));
                                c -= top ? (lft ? zb[4] : (rgt ? zb[5] : 0.f)) : (bot ? (lft ? zb[6] : (rgt ? zb[7] : 0.f)) : 0.f);
                                pacc[u][nb][e] -= c;
                            }
                        }
                    } else {
                        pacc[u][nb] = acc[u][nb];
                    }
                }
            eb = b; ey0 = y0; ex0 = x0;
            have_prev = true;
        }
        b = nb_; y0 = ny0; x0 = nx0;
    }
    if (have_prev && !(p.dbg & 4)) epilogue();
    if (p.ts && tid == 0) p.ts[8 * (blockIdx.y * gridDim.x + blockIdx.x) + 1] = wall_clock64();
}

template <int CIN, int COUT, int MODE, int LD, int EPI, int ZC = 0>
int launch_conv_po(ConvArgs& p, int nprob, hipStream_t stream) {
    constexpr int CHUNK = CIN < 8 ? CIN : 8;       // 8 channels per LDS stage: 36 KB per workgroup, 4 workgroups per CU
    constexpr bool N16 = COUT == 16 && CHUNK == 8;
    constexpr bool P2 = (CONV_P2 || ZC > 0) && !N16 && CHUNK == 8;
    const size_t lds = ((size_t)4 * (P2 ? SROWS * (8 * 40 + 32) : CHUNK * CSW) + (N16 ? 16 * (CIN * 9 + 1) : 4 * (COUT * (CIN * 3 + 4) + 16)) +
                        (ZC / 8) * 64 * 28) * sizeof(float);   // wave strips + weight image (+ composed weight image)
    static int resident = 0;               // workgroups of this instantiation that fit on the chip at once
    static pc_once_per_device once;
    if (once.need()) {
        const void* fn = reinterpret_cast<const void*>(&conv3x3_mfma_kernel<CIN, COUT, MODE, LD, EPI, ZC>);
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        hipFuncAttributes fa;
        e = hipFuncGetAttributes(&fa, fn);
        if (e != hipSuccess) return (int)e;
        resident = pc_resident_workgroups(fa.numRegs, lds);
        once.mark();
        if (getenv("POPCORN_CONV_DBG"))
            fprintf(stderr, "conv3x3<%d,%d,%d,%d,z%d>: %d regs, %zu B LDS -> %d resident workgroups\n", CIN, COUT, MODE, LD, ZC,
                    fa.numRegs, lds, resident);
    }
    // Persistent workgroups, all resident at once (one wave of a 256-thread workgroup per SIMD; the register count
    // decides how many workgroups a CU holds), split over the grouped problems; the grid is then shrunk to the smallest
    // one with the same number of rounds so that the last round is as full as the others.
    int max_grid = g_conv_max_grid > 0 ? g_conv_max_grid : resident / nprob;
    if (max_grid < 1) max_grid = 1;
    int grid = p.ntiles < max_grid ? p.ntiles : max_grid;
    const int rounds = (p.ntiles + grid - 1) / grid;
    grid = (p.ntiles + rounds - 1) / rounds;
    if (p.dbg & 16) {                      // ablation: the round-1 sizing (3 workgroups per CU whatever the kernel)
        grid = 768 / nprob < 128 ? 128 : 768 / nprob;
        if (grid > p.ntiles) grid = p.ntiles;
    }
    hipLaunchKernelGGL((conv3x3_mfma_kernel<CIN, COUT, MODE, LD, EPI, ZC>), dim3(grid, nprob), dim3(256), lds, stream, p);
    PC_CHECK_LAUNCH();
    return 0;
}

#include "conv3x3_fwd_s3.h"


// =====================================================================================================================
// Channels-last bf16 kernel (PC_PREC_BF16).  Activations / gradients are bf16 tensors with the channels of a pixel
// contiguous (cstride = 1, xstride = C: torch.channels_last), i.e. ONE aligned 16-byte slot per (pixel, 8-channel group)
// in HBM -- exactly the slot of the LDS strip image and of the MFMA operand:
//   * the loader is a masked copy: 6 x 34 slots of a strip = 204 16-byte pieces, 4 per lane, in 544-byte runs per row
//     (the planar layout moved 48 pieces of 80 bytes per strip and transposed them with 16 pack instructions per lane;
//     tools/layout_bw.hip: 28 us instead of 50 us for the loads + stores of the grouped 8 -> 8 @128x128 launch);
//   * the MFMA operands are swapped against the planar kernels: A = weights (M = (row of pair s, co)), B = pixels
//     (N = 16 x), so D hands every lane FOUR CONSECUTIVE CHANNELS of one pixel:
//         lane (x = lane & 15, lk = lane >> 4), register r:  s = lk >> 1,  co = 4 * (lk & 1) + r
//     = one 8-byte store per pixel, 256 contiguous bytes per 16 lanes, and every epilogue (BN + ReLU, ReLU-mask * BN scale,
//     accumulate, 2x2 max-pool copy, max-pool backward scatter, 1x1 partial logit) is per-pixel with a plain bounds
//     predicate: no separate "aligned interior" and "generic edge" paths, and no generic loader either (any placement
//     offset of a source keeps its slots aligned).
// Same wave-private strips, register-staged prefetch, deferred epilogue and persistent XCD-aware grid as above.
constexpr int CL_PX = 34;                 // pixels per strip row incl. the one-pixel halo
constexpr int CL_PIECES = SROWS * CL_PX;  // 16-byte pieces per strip and 8-channel chunk

__device__ __forceinline__ u32x4 cl_max8(u32x4 a, u32x4 b) {      // elementwise max of 8 bf16 (exact: no rounding involved)
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float lo = fmaxf(__uint_as_float(a[e] << 16), __uint_as_float(b[e] << 16));
        const float hi = fmaxf(__uint_as_float(a[e] & 0xffff0000u), __uint_as_float(b[e] & 0xffff0000u));
        o[e] = (__float_as_uint(hi) & 0xffff0000u) | (__float_as_uint(lo) >> 16);
    }
    return o;
}
template <int CIN, int COUT, int MODE, int LD, int EPI>
__global__ __launch_bounds__(256) void conv3x3_cl_kernel(const ConvArgs p) {
    constexpr int NCHUNK = CIN <= 8 ? 1 : CIN / 8;
    constexpr int NB = COUT / 8;
    constexpr int NIT = CIN < 8 ? CIN : 1;               // REFLECT loader: planar fp32 rows, one 16-byte segment per channel
    extern __shared__ __attribute__((aligned(16))) float lds[];

    // the problem's descriptor and the launch geometry, pinned in scalar registers (common.h: pc_pin -- read from the kernel arguments
    // inside the strip loop they were 15 - 32 scalar-memory round trips per strip, round 6)
    ConvProb q = p.pr[blockIdx.y];
    pc_pin(q.a); pc_pin(q.b); pc_pin(q.out);
    q.act = pc_pin_ptr(q.act); pc_pin(q.act_bstride); pc_pin(q.act_rstride); pc_pin(q.act_xstride);
    if constexpr (EPI == EPI_POOL) pc_pin(q.pool_out);
    if constexpr (EPI == EPI_DOT) { q.dot_w = pc_pin_ptr(q.dot_w); pc_pin(q.dot_out); }
    if constexpr (EPI == EPI_UPT) pc_pin(q.upt_out);
    int pH = p.H, pW = p.W, prelu = p.relu, paccum = p.accumulate, pdbg = p.dbg, ntl = p.ntiles, tlx = p.tiles_x, tly = p.tiles_y, gdim = (int)gridDim.x;
    pc_pin(pH); pc_pin(pW); pc_pin(prelu); pc_pin(paccum); pc_pin(pdbg); pc_pin(ntl); pc_pin(tlx); pc_pin(tly); pc_pin(gdim);
    pc_fastdiv dtx = p.div_tx, dtpi = p.div_tpi;
    pc_pin(dtx); pc_pin(dtpi);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    if (pdbg & 8) return;
    u32x4* const wl = reinterpret_cast<u32x4*>(lds) + wave * (SROWS * BSLOTS);       // this wave's strip: [6 rows][48 slots]

    // ---- loader: piece id = lane + 64 * i -> (strip row, pixel of the 34-pixel row)
    int l_slot[4];            // LDS slot of the piece, -1 = the lane has no such piece
    int l_r[4], l_px[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = lane + 64 * i;
        l_r[i] = id / CL_PX;
        l_px[i] = id - l_r[i] * CL_PX;
        l_slot[i] = id < CL_PIECES ? l_r[i] * BSLOTS + (COL0 - 1) + l_px[i] : -1;
    }
    const int CA = q.a.C;
    u32x4 R[LD == LD_POOL ? 16 : 4];
    f32x4 RF[LD == LD_REFLECT ? NIT : 1];
    unsigned rvalid = 0;
    const int r_r = lane / 10, r_seg = lane - r_r * 10;  // REFLECT: lane = (row, 4-pixel segment of the 40-pixel row)
    auto issue = [&](int ch, int b, int y0, int x0) {
        if constexpr (LD == LD_REFLECT) {
            rvalid = lane < 60 ? 1u : 0u;
#pragma unroll
            for (int it = 0; it < NIT; ++it)
                RF[it] = lane < 60 ? pc_fetch_reflect_seg(q.a, b, it, y0 - 1 + r_r, x0 - 4 + 4 * r_seg, pH, pW) : f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
            const bool useb = LD == LD_DIRECT && 8 * ch >= CA;
            const pc_src& s = useb ? q.b : q.a;
            const pc_bf16_t* base = reinterpret_cast<const pc_bf16_t*>(s.ptr) + b * s.bstride + (useb ? 8 * ch - CA : 8 * ch);
            const int rs = s.rstride, xs = s.xstride;
            unsigned vm = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int y = y0 - 1 + l_r[i], x = x0 - 1 + l_px[i];
                bool ok = l_slot[i] >= 0 && (unsigned)y < (unsigned)pH && (unsigned)x < (unsigned)pW;
                if constexpr (LD == LD_DIRECT) {
                    const int ys = y - s.oy, xq = x - s.ox;
                    ok = ok && (unsigned)ys < (unsigned)s.H && (unsigned)xq < (unsigned)s.W;
                    const int64_t off = ok ? (int64_t)ys * rs + (int64_t)xq * xs : 0;
                    R[i] = *reinterpret_cast<const u32x4*>(base + off);
                } else {      // LD_POOL: the 2x2 window of a source at twice the resolution (floor mode: always inside)
                    const int64_t off = ok ? (int64_t)(2 * y) * rs + (int64_t)(2 * x) * xs : 0;
                    const int rs1 = ok ? rs : 0, xs1 = ok ? xs : 0;
                    R[4 * i + 0] = *reinterpret_cast<const u32x4*>(base + off);
                    R[4 * i + 1] = *reinterpret_cast<const u32x4*>(base + off + xs1);
                    R[4 * i + 2] = *reinterpret_cast<const u32x4*>(base + off + rs1);
                    R[4 * i + 3] = *reinterpret_cast<const u32x4*>(base + off + rs1 + xs1);
                }
                vm |= (ok ? 1u : 0u) << i;
            }
            rvalid = vm;
        }
    };
    auto commit = [&]() {
        if constexpr (LD == LD_REFLECT) {
            // planar fp32 model input: round + pack here (the one operand no producer has rounded); channel slots >= CIN are zero
            if (lane < 60) {
                u32x4* d = wl + r_r * BSLOTS + 4 * r_seg;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    u32x4 t = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
                    for (int h = 0; h < (NIT + 1) / 2; ++h)
                        t[h] = pc_pack_bf16(RF[2 * h][e], 2 * h + 1 < NIT ? RF[(2 * h + 1) % NIT][e] : 0.f);
                    d[e] = t;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (l_slot[i] >= 0) {
                    u32x4 v;
                    if constexpr (LD == LD_POOL) v = cl_max8(cl_max8(R[4 * i], R[4 * i + 1]), cl_max8(R[4 * i + 2], R[4 * i + 3]));
                    else v = R[i];
                    wl[l_slot[i]] = ((rvalid >> i) & 1u) ? v : u32x4{0u, 0u, 0u, 0u};
                }
            }
        }
    };

    const int my_tiles = ntl > (int)blockIdx.x ? (ntl - 1 - (int)blockIdx.x) / gdim + 1 : 0;
    const int nstages = my_tiles * NCHUNK;
    auto strip_coords = [&](int stage, int& b, int& y0, int& x0) {
        const int t = blockIdx.x + (stage / NCHUNK) * gdim;
        const int tile = pc_xcd_remap(t, ntl);
        b = (int)pc_div((uint32_t)tile, dtpi);
        const int rem = tile - b * tlx * tly;
        const int ty = (int)pc_div((uint32_t)rem, dtx);
        x0 = (rem - ty * tlx) * TW;
        y0 = ty * TH + 4 * wave;
    };
    int b = 0, y0 = 0, x0 = 0;
    if (nstages > 0) {
        strip_coords(0, b, y0, x0);
        if (!(pdbg & 1)) issue(0, b, y0, x0);
    }

    // ---- weight image [dy plane 0..3][co][chunk][dx][8 ci] bf16 (plane 3 all zero), as in the planar bf16 path; a lane's
    // A fragment for (chunk, dx): the 8 input channels of tap (dy = lk - s, dx) of output channel co = li & 7, s = li >> 3
    unsigned short* const w2h = reinterpret_cast<unsigned short*>(lds + 4 * BWAVE_F);
    constexpr int BW_CO = NCHUNK * 24;
    constexpr int BW_DYS = COUT * BW_CO;
    constexpr int NWR = (COUT * CIN * 9 + 255) / 256;
    float wreg[NWR];
#pragma unroll
    for (int k = 0; k < NWR; ++k) {
        const int e = tid + k * 256;
        const int ec = e < COUT * CIN * 9 ? e : 0;
        const int tap = ec % 9, ci = (ec / 9) % CIN, co = ec / (9 * CIN);
        if (q.w_cin) {       // channel window of a shared 8-channel input (first layers): zero weights outside it
            const int cw = ci - q.w_ci0;
            const bool in = (unsigned)cw < (unsigned)q.w_cin;
            wreg[k] = in ? q.w[(co * q.w_cin + cw) * 9 + tap] : 0.f;
        } else {
            wreg[k] = q.w[co * p.w_co_stride + ci * p.w_ci_stride + (p.w_flip ? 8 - tap : tap)];
        }
    }
    // per-lane epilogue constants for co = nb*8 + 4*(lk&1) + r
    const bool has_bn = MODE == MODE_FWD || q.act != nullptr;
    const int c4 = 4 * (lk & 1);
    float e_scale[NB][4], e_shift[NB][4];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = nb * 8 + c4 + r;
            const float cb = has_bn && q.bn.conv_bias ? q.bn.conv_bias[c] : 0.f;
            if (has_bn && q.bn.gamma) {
                e_scale[nb][r] = q.bn.gamma[c] * (1.0f / sqrtf(q.bn.var[c] + q.bn.eps));
                e_shift[nb][r] = (cb - q.bn.mean[c]) * e_scale[nb][r] + q.bn.beta[c];
            } else {
                e_scale[nb][r] = 1.f;
                e_shift[nb][r] = cb;
            }
        }
    float dotw[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI == EPI_DOT) {
        if (q.dot_w) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dotw[r] = pc_bf16r(q.dot_w[c4 + r]);
        }
    }
    // EPI_UPT: A fragments of the transposed conv for the two rows of a pair.  The epilogue's packed output of a lane IS the B
    // fragment of v_mfma_f32_16x16x16_bf16 (N = pixel li, k-group lk = 4 channels): lanes lk = 0, 1 hold row s = 0 of the pair, lanes
    // lk = 2, 3 row s = 1 -- so the weights of row s sit in the k-groups 2 s, 2 s + 1 of A and the other two k-groups are zero
    // (M = (x parity b = li >> 3, output channel li & 7); one instruction per (row s, output-row parity a))
    typedef short upt_s4 __attribute__((ext_vector_type(4)));
    upt_s4 upt_aw[2][2];
    float upt_bias[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI == EPI_UPT) {
#pragma unroll
        for (int sr = 0; sr < 2; ++sr)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ci = 4 * (lk & 1) + e;
                    upt_aw[sr][a][e] = (lk >> 1) == sr ? (short)pc_f2bf(q.upt_w[((ci * 8 + (li & 7)) * 2 + a) * 2 + (li >> 3)]) : (short)0;
                }
#pragma unroll
        for (int r = 0; r < 4; ++r) upt_bias[r] = q.upt_b ? q.upt_b[c4 + r] : 0.f;
    }
    for (int e = tid; e < 4 * BW_DYS / 2; e += 256) reinterpret_cast<unsigned*>(w2h)[e] = 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NWR; ++k) {
        const int e = tid + k * 256;
        if (e < COUT * CIN * 9) {
            const int tap = e % 9, ci = (e / 9) % CIN, co = e / (9 * CIN);
            w2h[(tap / 3) * BW_DYS + co * BW_CO + (ci / 8) * 24 + (tap % 3) * 8 + (ci % 8)] = pc_f2bf(wreg[k]);
        }
    }
    __syncthreads();
    // (see the planar kernel: consume the constants before the loop so that no vmcnt(0) lands in front of the epilogue)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("" : : "v"(e_scale[nb][r]), "v"(e_shift[nb][r]));
    const int a_s = li >> 3, a_co = li & 7;
    const unsigned short* const wlane_h = w2h + (((unsigned)(lk - a_s) <= 2u) ? lk - a_s : 3) * BW_DYS + a_co * BW_CO;
    bf16x8 bwh[3][NB];
    auto load_bwh = [&](int ch) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
                bwh[dx][nb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wlane_h + nb * 8 * BW_CO + ch * 24 + dx * 8));
    };

    const pc_bf16_t* const act = reinterpret_cast<const pc_bf16_t*>(q.act);
    pc_bf16_t* const outp = reinterpret_cast<pc_bf16_t*>(q.out.ptr);
    const int64_t o_bs = q.out.bstride, a_bs = q.act_bstride;
    const int o_rs = q.out.rstride, o_xs = q.out.xstride, a_rs = q.act_rstride, a_xs = q.act_xstride;

    f32x4 pacc[4][NB];
    int eb = 0, ey0 = 0, ex0 = 0;
    bool have_prev = false;
    const int e_s = lk >> 1;
    auto epilogue = [&]() {
        // lane holds pixel (y = ey0 + 2*(u>>1) + e_s, x = ex0 + (u&1)*16 + li), channels nb*8 + c4 + r
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int y = ey0 + 2 * (u >> 1) + e_s, x = ex0 + (u & 1) * 16 + li;
            const bool ok = y < pH && x < pW;
            if constexpr (MODE == MODE_FWD) {
                float dsum = 0.f;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    f32x4 v = pacc[u][nb];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float o = v[r] * e_scale[nb][r] + e_shift[nb][r];
                        v[r] = pc_bf16r(prelu ? fmaxf(o, 0.f) : o);
                    }
                    if (EPI == EPI_DOT && q.dot_w) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) dsum += v[r] * dotw[r];
                        continue;
                    }
                    if (ok && (EPI != EPI_UPT || outp)) pc_st4(outp + eb * o_bs + (int64_t)y * o_rs + (int64_t)x * o_xs + nb * 8 + c4, v);
                    if constexpr (EPI == EPI_UPT) {
                        // the transposed conv of this unit's two rows: D[(b, co)][pixel] -> up-sampled pixel (2 y_s + a, 2 x + b)
                        upt_s4 bv;
                        {
                            const unsigned p0 = pc_pack_bf16(v[0], v[1]), p1 = pc_pack_bf16(v[2], v[3]);
                            bv[0] = (short)(p0 & 0xffffu); bv[1] = (short)(p0 >> 16); bv[2] = (short)(p1 & 0xffffu); bv[3] = (short)(p1 >> 16);
                        }
                        pc_bf16_t* const uo = reinterpret_cast<pc_bf16_t*>(q.upt_out.ptr) + eb * q.upt_out.bstride;
                        const int u_rs = q.upt_out.rstride, u_xs = q.upt_out.xstride;
#pragma unroll
                        for (int sr = 0; sr < 2; ++sr) {
                            const int ys = ey0 + 2 * (u >> 1) + sr;
                            const bool oks = ys < pH && x < pW;
#pragma unroll
                            for (int a = 0; a < 2; ++a) {
                                f32x4 ua = f32x4{upt_bias[0], upt_bias[1], upt_bias[2], upt_bias[3]};
                                ua = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(upt_aw[sr][a], bv, ua, 0, 0, 0);
                                if (oks) pc_st4(uo + (int64_t)(2 * ys + a) * u_rs + (int64_t)(2 * x + (lk >> 1)) * u_xs + c4, ua);
                            }
                        }
                    }
                    if (EPI == EPI_POOL && q.pool_out.ptr) {
                        // MaxPool2d(2) (full strips only, pc_conv3x3_pool_out_ok): x pair = lane ^ 1 (DPP quad permute), row pair =
                        // lane ^ 32 (v_permlane32_swap: both halves' values in every lane) -- no trip through the LDS crossbar
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float m = fmaxf(v[r], __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[r]), 0xB1, 0xF, 0xF, false)));
                            const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
                            v[r] = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
                        }
                        if ((li & 1) == 0 && e_s == 0)
                            pc_st4(reinterpret_cast<pc_bf16_t*>(q.pool_out.ptr) + eb * q.pool_out.bstride +
                                       (int64_t)((ey0 >> 1) + (u >> 1)) * q.pool_out.rstride +
                                       (int64_t)((ex0 >> 1) + (u & 1) * 8 + (li >> 1)) * q.pool_out.xstride + nb * 8 + c4, v);
                    }
                }
                if (EPI == EPI_DOT && q.dot_w) {
                    dsum = pc_xor16_sum(dsum);             // the other four channels of the pixel
                    if ((lk & 1) == 0 && ok)
                        q.dot_out.ptr[eb * q.dot_out.bstride + (int64_t)y * q.dot_out.rstride + x] = dsum;
                }
            } else if constexpr (EPI != EPI_POOLBWD) {
                if (!ok) continue;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    f32x4 v = pacc[u][nb];
                    pc_bf16_t* op = outp + eb * o_bs + (int64_t)y * o_rs + (int64_t)x * o_xs + nb * 8 + c4;
                    if (act) {
                        const f32x4 a4 = pc_ld4(act + eb * a_bs + (int64_t)y * a_rs + (int64_t)x * a_xs + nb * 8 + c4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = a4[r] > 0.f ? v[r] * e_scale[nb][r] : 0.f;
                    }
                    if (paccum) {
                        const f32x4 o4 = pc_ld4(op);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += o4[r];
                    }
                    pc_st4(op, v);
                }
            } else {
                // MaxPool2d(2) backward: (y, x) is a pooled coordinate; the gradient goes to the first arg-max of the 2x2 window
                if (!ok) continue;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const f32x4 v = pacc[u][nb];
                    const pc_bf16_t* a0 = act + eb * a_bs + (int64_t)(2 * y) * a_rs + (int64_t)(2 * x) * a_xs + nb * 8 + c4;
                    pc_bf16_t* o0 = outp + eb * o_bs + (int64_t)(2 * y) * o_rs + (int64_t)(2 * x) * o_xs + nb * 8 + c4;
                    const f32x4 A00 = pc_ld4(a0), A01 = pc_ld4(a0 + a_xs), A10 = pc_ld4(a0 + a_rs), A11 = pc_ld4(a0 + a_rs + a_xs);
                    f32x4 O00 = pc_ld4(o0), O01 = pc_ld4(o0 + o_xs), O10 = pc_ld4(o0 + o_rs), O11 = pc_ld4(o0 + o_rs + o_xs);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int am = 0;
                        float m = A00[r];
                        if (A01[r] > m) { m = A01[r]; am = 1; }
                        if (A10[r] > m) { m = A10[r]; am = 2; }
                        if (A11[r] > m) { m = A11[r]; am = 3; }
                        const float g = m > 0.f ? v[r] * e_scale[nb][r] : 0.f;
                        O00[r] += am == 0 ? g : 0.f;
                        O01[r] += am == 1 ? g : 0.f;
                        O10[r] += am == 2 ? g : 0.f;
                        O11[r] += am == 3 ? g : 0.f;
                    }
                    pc_st4(o0, O00); pc_st4(o0 + o_xs, O01); pc_st4(o0 + o_rs, O10); pc_st4(o0 + o_rs + o_xs, O11);
                }
            }
        }
    };

    f32x4 acc[4][NB];
    for (int stage = 0; stage < nstages; ++stage) {
        const int ch = stage % NCHUNK;
        if (!(pdbg & 1)) commit();
        int nb_ = b, ny0 = y0, nx0 = x0;
        if (stage + 1 < nstages) {
            if ((stage + 1) % NCHUNK == 0) strip_coords(stage + 1, nb_, ny0, nx0);
            if (!(pdbg & 1)) issue((stage + 1) % NCHUNK, nb_, ny0, nx0);
        }
        if (have_prev) {
            if (!(pdbg & 4)) epilogue();
            have_prev = false;
        }
        if (ch == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[u][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (!(pdbg & 2)) {
            load_bwh(ch);
            const u32x4* lrow = wl + lk * BSLOTS + (COL0 - 1) + li;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                bf16x8 av[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    av[u] = __builtin_bit_cast(bf16x8, lrow[(u >> 1) * 2 * BSLOTS + (u & 1) * 16 + dx]);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[u][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bwh[dx][nb], av[u], acc[u][nb], 0, 0, 0);
            }
        }
        if (ch == NCHUNK - 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) pacc[u][nb] = acc[u][nb];
            eb = b; ey0 = y0; ex0 = x0;
            have_prev = true;
        }
        b = nb_; y0 = ny0; x0 = nx0;
    }
    if (have_prev && !(pdbg & 4)) epilogue();
}

template <int CIN, int COUT, int MODE, int LD, int EPI>
int launch_conv_cl(ConvArgs& p, int nprob, hipStream_t stream) {
    constexpr int NCHUNK = CIN <= 8 ? 1 : CIN / 8;
    const size_t lds = (size_t)4 * BWAVE_F * sizeof(float) + (size_t)4 * COUT * NCHUNK * 24 * sizeof(unsigned short);
    static int resident = 0;
    static pc_once_per_device once;
    if (once.need()) {
        const void* fn = reinterpret_cast<const void*>(&conv3x3_cl_kernel<CIN, COUT, MODE, LD, EPI>);
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        hipFuncAttributes fa;
        e = hipFuncGetAttributes(&fa, fn);
        if (e != hipSuccess) return (int)e;
        resident = pc_resident_workgroups(fa.numRegs, lds);
        once.mark();
        if (getenv("POPCORN_CONV_DBG"))
            fprintf(stderr, "conv3x3_cl<%d,%d,%d,%d,%d>: %d regs, %zu B LDS -> %d resident workgroups\n", CIN, COUT, MODE, LD, EPI,
                    fa.numRegs, lds, resident);
    }
    int max_grid = g_conv_max_grid > 0 ? g_conv_max_grid : resident / nprob;
    if (max_grid < 1) max_grid = 1;
    int grid = p.ntiles < max_grid ? p.ntiles : max_grid;
    const int rounds = (p.ntiles + grid - 1) / grid;
    grid = (p.ntiles + rounds - 1) / rounds;
    hipLaunchKernelGGL((conv3x3_cl_kernel<CIN, COUT, MODE, LD, EPI>), dim3(grid, nprob), dim3(256), lds, stream, p);
    PC_CHECK_LAUNCH();
    return 0;
}

template <int CIN, int COUT, int MODE, int LD>
int launch_conv_cl_epi(ConvArgs& p, int nprob, hipStream_t stream) {
    if constexpr (MODE == MODE_FWD && CIN >= 8) {
        bool po = false;
        for (int i = 0; i < nprob; ++i) po = po || p.pr[i].pool_out.ptr != nullptr;
        if (po) return launch_conv_cl<CIN, COUT, MODE, LD, EPI_POOL>(p, nprob, stream);
    }
    if constexpr (MODE == MODE_FWD && CIN == 8 && COUT == 8) {
        bool dot = false;
        for (int i = 0; i < nprob; ++i) dot = dot || p.pr[i].dot_w != nullptr;
        if (dot) return launch_conv_cl<CIN, COUT, MODE, LD, EPI_DOT>(p, nprob, stream);
    }
    if constexpr (MODE == MODE_FWD && CIN == 8 && COUT == 8 && LD == LD_DIRECT) {
        int nupt = 0;
        for (int i = 0; i < nprob; ++i) nupt += p.pr[i].upt_w != nullptr;
        if (nupt) {
            if (nupt != nprob) return PC_EINVAL;            // all problems of the launch or none
            return launch_conv_cl<CIN, COUT, MODE, LD, EPI_UPT>(p, nprob, stream);
        }
    }
    if constexpr (MODE == MODE_DGRAD) {
        if (p.pool) return launch_conv_cl<CIN, COUT, MODE, LD, EPI_POOLBWD>(p, nprob, stream);
    }
    return launch_conv_cl<CIN, COUT, MODE, LD, EPI_NONE>(p, nprob, stream);
}

// bf16 mode: validate the channels-last descriptors and pick the loader
template <int CIN, int COUT, int MODE>
int launch_conv_bf16(ConvArgs& p, int nprob, hipStream_t stream) {
    int mode = -1;
    for (int i = 0; i < nprob; ++i) {
        ConvProb& q = p.pr[i];
        const int m = q.a.mode;
        if (mode >= 0 && m != mode) return PC_EINVAL;
        mode = m;
        if (m == PC_SRC_REFLECT) {       // the model input: planar fp32
            if (CIN > 4 || q.b.C || q.a.dtype != PC_F32 || !pc_planar(q.a)) return PC_EINVAL;
        } else {
            if (CIN < 8 || !pc_cl_ok(q.a) || q.a.C % 8 != 0 || q.a.xstride < q.a.C) return PC_EINVAL;
            if (m == PC_SRC_POOL2 && (q.b.C || CIN > 16 || q.a.W < 2 * p.W || q.a.H < 2 * p.H)) return PC_EINVAL;
            if (q.b.C && (!pc_cl_ok(q.b) || q.b.mode != PC_SRC_DIRECT || q.b.C % 8 != 0)) return PC_EINVAL;
        }
        const bool own_out = q.out.ptr != reinterpret_cast<float*>(q.dot_out.ptr);
        if (own_out && !pc_cl_ok(q.out)) return PC_EINVAL;
        if (q.pool_out.ptr && !pc_cl_ok(q.pool_out)) return PC_EINVAL;
        if (q.dot_out.ptr && (q.dot_out.dtype != PC_F32 || !pc_planar(q.dot_out))) return PC_EINVAL;
        if (q.act && !pc_cl_ok(q.act, q.act_dtype, q.act_bstride, q.act_cstride, q.act_rstride, q.act_xstride)) return PC_EINVAL;
    }
    if (mode == PC_SRC_REFLECT) {
        if constexpr (CIN <= 4) return launch_conv_cl_epi<CIN, COUT, MODE, LD_REFLECT>(p, nprob, stream);
    } else if (mode == PC_SRC_POOL2) {
        if constexpr (CIN >= 8 && CIN <= 16) return launch_conv_cl_epi<CIN, COUT, MODE, LD_POOL>(p, nprob, stream);
    } else {
        if constexpr (CIN >= 8) return launch_conv_cl_epi<CIN, COUT, MODE, LD_DIRECT>(p, nprob, stream);
    }
    return PC_EINVAL;
}

template <int CIN, int COUT, int MODE, int LD>
int launch_conv_ld(ConvArgs& p, int nprob, hipStream_t stream) {
    if constexpr (MODE == MODE_FWD && LD == LD_DIRECT && (CIN == 8 || CIN == 16) && (COUT == 8 || COUT == 16)) {
        // split-operand form (conv3x3_fwd_s3.h): every 8- / 16-channel layer whose tensors are aligned
        if (fwd_s3_ok<CIN, COUT>(p, nprob, 0)) {
            bool po = false, dot = false;
            for (int i = 0; i < nprob; ++i) {
                po = po || p.pr[i].pool_out.ptr != nullptr;
                dot = dot || p.pr[i].dot_w != nullptr;
            }
            if constexpr (CIN == 8 && COUT == 8) {
                if (dot) return launch_fwd_s3<CIN, COUT, EPI_DOT, 0>(p, nprob, stream);
            }
            return po ? launch_fwd_s3<CIN, COUT, EPI_POOL, 0>(p, nprob, stream) : launch_fwd_s3<CIN, COUT, EPI_NONE, 0>(p, nprob, stream);
        }
    }
    if constexpr (MODE == MODE_FWD && CIN >= 8) {       // the layers in front of a Down block: inc.conv.3 (8->8), down1 conv.3 (16->16)
        bool po = false;
        for (int i = 0; i < nprob; ++i) po = po || p.pr[i].pool_out.ptr != nullptr;
        if (po) return launch_conv_po<CIN, COUT, MODE, LD, EPI_POOL>(p, nprob, stream);
    }
    if constexpr (MODE == MODE_FWD && CIN == 8 && COUT == 8) {      // up1.conv.3, the layer in front of the 1x1 out-conv
        bool dot = false;
        for (int i = 0; i < nprob; ++i) dot = dot || p.pr[i].dot_w != nullptr;
        if (dot) return launch_conv_po<CIN, COUT, MODE, LD, EPI_DOT>(p, nprob, stream);
    }
    if constexpr (MODE == MODE_DGRAD) {
        // the pool-backward epilogues live in their own instantiation: in the plain one they cost a wave per SIMD
        if (p.pool) return launch_conv_po<CIN, COUT, MODE, LD, EPI_POOLBWD>(p, nprob, stream);
    }
    return launch_conv_po<CIN, COUT, MODE, LD, EPI_NONE>(p, nprob, stream);
}

// staged-loader classification of a planar fp32 source (1 = aligned DIRECT, 2 = aligned POOL2, 0 = generic)
int conv_src_mode(const pc_src& s, int H, int W) {
    if (s.C == 0) return 0;
    const bool al = ((reinterpret_cast<uintptr_t>(s.ptr) & 15) == 0) && (s.rstride % 4 == 0) && (s.cstride % 4 == 0) && (s.bstride % 4 == 0);
    if (!al) return 0;
    // (W % 4 != 0 is fine: the row stride is a multiple of 4, so every 16-byte segment that starts inside a row exists in
    // memory; the loader masks its tail per pixel.  POOL2 reads source columns 2 xg .. 2 xg + 7 of which only those below 2 W are
    // used: with 1 or 2 valid pixels in the last segment (W % 4 = 1, 2) the second 16-byte piece is not read at all, with 3
    // (W % 4 = 3) it ends at column 2 W + 1: the source row must then have at least 2 W + 2 floats.)
    // (a DIRECT source at the origin that ends before the conv domain does is zero beyond its extent: the loader masks per source)
    if (s.mode == PC_SRC_DIRECT && s.oy == 0 && s.ox == 0 && s.H <= H && s.W <= W && s.H >= 1 && s.W >= 1) return 1;
    if (s.mode == PC_SRC_POOL2 && s.W == 2 * W && s.H >= 2 * H && ((W % 4) != 3 || s.rstride >= 2 * W + 2)) return 2;
    return 0;
}

bool same_layout(const pc_src& a, const pc_src& b) { return b.C == 0 || (a.bstride == b.bstride && a.rstride == b.rstride); }

template <int CIN, int COUT, int MODE>
int launch_conv(ConvArgs& p, int nprob, hipStream_t stream) {
    constexpr int CHUNK = CIN < 8 ? CIN : 8;       // 8 channels per LDS stage: 36 KB per workgroup, 4 workgroups per CU
    p.tiles_x = (p.W + TW - 1) / TW;
    p.tiles_y = (p.H + TH - 1) / TH;
    p.ntiles = p.B * p.tiles_x * p.tiles_y;
    if (p.ntiles <= 0) return 0;
    p.div_tx = pc_make_fastdiv(p.tiles_x);
    p.div_tpi = pc_make_fastdiv(p.tiles_x * p.tiles_y);
    p.dbg = g_conv_dbg;
    p.ts = g_conv_ts;
    if (g_pc_precision == PC_PREC_BF16) return launch_conv_bf16<CIN, COUT, MODE>(p, nprob, stream);
    // loader choice: all problems of the group must qualify for a staged loader
    bool direct = true, pool = CHUNK >= 8, reflect = true;      // DIRECT also for the 2 / 4-channel first layers (pre-padded input)
    bool vec = true;      // (any width: the vector epilogues only take strips that lie completely inside the image; a ragged last strip
                          // of a row takes the per-element path)
    // fp32 mode: planar fp32 tensors everywhere
    const uintptr_t amask = 15;                        // a 4-pixel vector access: 16 bytes
    for (int i = 0; i < nprob; ++i) {
        ConvProb& q = p.pr[i];
        if (q.a.dtype != PC_F32 || !pc_planar(q.a) || (q.b.C && (q.b.dtype != PC_F32 || !pc_planar(q.b)))) return PC_EINVAL;
        if ((q.out.ptr != reinterpret_cast<float*>(q.dot_out.ptr) && (q.out.dtype != PC_F32 || !pc_planar(q.out))) ||
            (q.pool_out.ptr && (q.pool_out.dtype != PC_F32 || !pc_planar(q.pool_out))) ||
            (q.dot_out.ptr && (q.dot_out.dtype != PC_F32 || !pc_planar(q.dot_out))) || (q.act && (q.act_dtype != PC_F32 || q.act_xstride > 1)))
            return PC_EINVAL;
        q.fast_a = conv_src_mode(q.a, p.H, p.W);
        q.fast_b = conv_src_mode(q.b, p.H, p.W);
        const bool lay = same_layout(q.a, q.b);
        direct = direct && q.fast_a == 1 && (q.b.C == 0 || q.fast_b == 1) && lay && (CIN <= 16 || q.a.C == 16);
        pool = pool && q.fast_a == 2 && q.b.C == 0;
        vec = vec && (q.out.rstride % 4 == 0) && (q.out.cstride % 4 == 0) && (q.out.bstride % 4 == 0) &&
              ((reinterpret_cast<uintptr_t>(q.out.ptr) & amask) == 0);
        if (q.act) vec = vec && (q.act_rstride % 4 == 0) && (q.act_cstride % 4 == 0) && (q.act_bstride % 4 == 0) &&
                         ((reinterpret_cast<uintptr_t>(q.act) & amask) == 0);
        reflect = reflect && q.a.mode == PC_SRC_REFLECT && q.b.C == 0;
    }
    p.vec_ok = vec ? 1 : 0;
    if (direct) return launch_conv_ld<CIN, COUT, MODE, LD_DIRECT>(p, nprob, stream);
    if constexpr (CHUNK >= 8 && CIN <= 16) {
        if (pool) return launch_conv_ld<CIN, COUT, MODE, LD_POOL>(p, nprob, stream);
    }
    if constexpr (CIN <= 4) {
        if (reflect) return launch_conv_ld<CIN, COUT, MODE, LD_REFLECT>(p, nprob, stream);
    }
    return launch_conv_ld<CIN, COUT, MODE, LD_GENERIC>(p, nprob, stream);
}

template <int MODE>
int dispatch_conv(ConvArgs& p, int nprob, int Cin, int Cout, hipStream_t stream) {
#define PC_CASE(ci, co) \
    if (Cin == ci && Cout == co) return launch_conv<ci, co, MODE>(p, nprob, stream);
    PC_CASE(2, 8) PC_CASE(4, 8) PC_CASE(8, 8) PC_CASE(16, 8) PC_CASE(32, 8) PC_CASE(8, 16) PC_CASE(16, 16)
#undef PC_CASE
    return PC_EINVAL;
}

int fill_fwd(ConvProb& q, const pc_src* a, const pc_src* b, const float* w, const pc_bn* bn, const pc_dst* out, int Cin) {
    if (!a || !w || !bn) return PC_EINVAL;
    q.a = *a;
    if (b) q.b = *b;
    if (q.a.C + q.b.C != Cin) return PC_EINVAL;
    q.w = w;
    q.bn = *bn;
    if (out) q.out = *out;
    return 0;
}

int fill_dgrad(ConvProb& q, const pc_src* g, const float* w, int c0, const pc_src* act, const pc_bn* act_bn, int pool,
               const pc_dst* out, int Cg) {
    if (!g || !w || !out || g->C != Cg) return PC_EINVAL;
    if (pool && !act) return PC_EINVAL;
    q.a = *g;
    // forward weight w[cg][Cin_total][3][3]; as a conv over g producing input channel (c0 + co):
    //   weight(out = co, in = cg, tap) = w[cg][c0 + co][8 - tap]
    q.w = w + (int64_t)c0 * 9;
    if (act) {
        if (!act_bn) return PC_EINVAL;
        q.bn = *act_bn;
        q.act = act->ptr;
        q.act_bstride = act->bstride;
        q.act_cstride = act->cstride;
        q.act_rstride = act->rstride;
        q.act_dtype = act->dtype;
        q.act_xstride = act->xstride;
    }
    q.out = *out;
    return 0;
}

// the pooled second output needs every strip on the vector epilogue: full 32 x 4 strips, 16-byte aligned output
bool pool_out_geometry_ok(const pc_dst& out, int H, int W) {
    if (out.dtype == PC_BF16) return W % 32 == 0 && H % 4 == 0 && pc_cl_ok(out);      // channels-last bf16 (bf16 mode)
    return W % 32 == 0 && H % 4 == 0 && pc_planar(out) && out.rstride % 4 == 0 && out.cstride % 4 == 0 && out.bstride % 4 == 0 &&
           (reinterpret_cast<uintptr_t>(out.ptr) & 15) == 0;
}

}  // namespace

extern "C" int pc_conv3x3_pool_out_ok(const pc_dst* out, int H, int W) { return out && pool_out_geometry_ok(*out, H, W) ? 1 : 0; }

extern "C" int pc_conv3x3_bn_relu_fwd(const pc_src* a, const pc_src* b, const float* w, const pc_bn* bn, int relu,
                                      const pc_dst* out, int B, int H, int W, int Cin, int Cout, void* stream) {
    ConvArgs p{};
    const int rc = fill_fwd(p.pr[0], a, b, w, bn, out, Cin);
    if (rc) return rc;
    p.w_co_stride = Cin * 9;
    p.w_ci_stride = 9;
    p.relu = relu;
    p.B = B; p.H = H; p.W = W;
    return dispatch_conv<MODE_FWD>(p, 1, Cin, Cout, (hipStream_t)stream);
}

extern "C" int pc_conv3x3_bn_relu_fwd_group(int n, const pc_conv_fwd_desc* d, int relu, int B, int H, int W, int Cin, int Cout,
                                            void* stream) {
    if (n < 1 || n > MAXG || !d) return PC_EINVAL;
    ConvArgs p{};
    for (int i = 0; i < n; ++i) {
        const int rc = fill_fwd(p.pr[i], d[i].a, d[i].b, d[i].w, d[i].bn, d[i].out, Cin);
        if (rc) return rc;
        if (d[i].upt_w) {
            // the transposed conv of the Up block that follows, in this launch's epilogue (channels-last bf16, 8 -> 8 layers)
            if (g_pc_precision != PC_PREC_BF16 || Cin != 8 || Cout != 8 || !relu || d[i].b || d[i].pool_out || d[i].dot_w || !d[i].upt_out ||
                !d[i].upt_out->ptr || !pc_cl_ok(*d[i].upt_out) || d[i].upt_out->xstride < 8 || d[i].a->mode != PC_SRC_DIRECT)
                return PC_EINVAL;
            p.pr[i].upt_w = d[i].upt_w;
            p.pr[i].upt_b = d[i].upt_b;
            p.pr[i].upt_out = *d[i].upt_out;
        }
        if (d[i].w_cin) {
            if (g_pc_precision != PC_PREC_BF16 || Cin != 8 || d[i].b || d[i].w_ci0 < 0 || d[i].w_cin < 1 || d[i].w_ci0 + d[i].w_cin > 8 ||
                d[i].a->mode != PC_SRC_DIRECT)
                return PC_EINVAL;
            p.pr[i].w_ci0 = d[i].w_ci0;
            p.pr[i].w_cin = d[i].w_cin;
        }
        if (d[i].dot_w) {
            // the partial 1x1 sum replaces the feature map.  bf16 mode: all strips must take the vector epilogue; fp32: any geometry
            // (partial strips take the per-element form), planar fp32 output
            const bool dot_geom = d[i].dot_out && (g_pc_precision == PC_PREC_BF16 ? pool_out_geometry_ok(*d[i].dot_out, H, W)
                                                                                  : (d[i].dot_out->dtype == PC_F32 && pc_planar(*d[i].dot_out)));
            if (!d[i].dot_out || !d[i].dot_out->ptr || Cin != 8 || Cout != 8 || !relu || d[i].pool_out || !dot_geom)
                return PC_EINVAL;
            p.pr[i].dot_w = d[i].dot_w;
            p.pr[i].dot_out = *d[i].dot_out;
            if (!d[i].out) p.pr[i].out = *d[i].dot_out;     // keeps the alignment checks of the launcher meaningful
        } else if (!d[i].out) {
            return PC_EINVAL;
        }
        if (d[i].pool_out) {
            const pc_dst& po = *d[i].pool_out;
            const bool po_ok = po.dtype == PC_BF16 ? pc_cl_ok(po)
                                                   : (po.rstride % 2 == 0 && po.cstride % 2 == 0 && po.bstride % 2 == 0 &&
                                                      (reinterpret_cast<uintptr_t>(po.ptr) & 7) == 0 && pc_planar(po));
            if (Cin < 8 || !pool_out_geometry_ok(*d[i].out, H, W) || !po.ptr || !po_ok) return PC_EINVAL;
            p.pr[i].pool_out = po;
        }
    }
    p.w_co_stride = Cin * 9;
    p.w_ci_stride = 9;
    p.relu = relu;
    p.B = B; p.H = H; p.W = W;
    return dispatch_conv<MODE_FWD>(p, n, Cin, Cout, (hipStream_t)stream);
}

extern "C" int pc_conv3x3_dgrad(const pc_src* g, const float* w, int Cin_total, int c0, int Cn,
                                const pc_src* act, const pc_bn* act_bn, int pool, int accumulate,
                                const pc_dst* out, int B, int H, int W, int Cg, void* stream) {
    ConvArgs p{};
    const int rc = fill_dgrad(p.pr[0], g, w, c0, act, act_bn, pool, out, Cg);
    if (rc) return rc;
    p.w_co_stride = 9;
    p.w_ci_stride = Cin_total * 9;
    p.w_flip = 1;
    p.pool = pool;
    p.accumulate = accumulate;
    p.B = B; p.H = H; p.W = W;
    return dispatch_conv<MODE_DGRAD>(p, 1, Cg, Cn, (hipStream_t)stream);
}

extern "C" int pc_conv3x3_dgrad_group(int n, const pc_conv_dgrad_desc* d, int Cin_total, int c0, int Cn, int pool,
                                      int accumulate, int B, int H, int W, int Cg, void* stream) {
    if (n < 1 || n > MAXG || !d) return PC_EINVAL;
    ConvArgs p{};
    for (int i = 0; i < n; ++i) {
        const int rc = fill_dgrad(p.pr[i], d[i].g, d[i].w, c0, d[i].act, d[i].act_bn, pool, d[i].out, Cg);
        if (rc) return rc;         // (act is a per-problem property: the masked skip block and the plain up block of a concat layer mix)
    }
    p.w_co_stride = 9;
    p.w_ci_stride = Cin_total * 9;
    p.w_flip = 1;
    p.pool = pool;
    p.accumulate = accumulate;
    p.B = B; p.H = H; p.W = W;
    return dispatch_conv<MODE_DGRAD>(p, n, Cg, Cn, (hipStream_t)stream);
}


// =====================================================================================================================
// Up block without the up-sampled map:  conv3x3(cat[skip, ConvTranspose2d(z)]) = conv3x3(skip; W[:, :Cs]) + a parity-dependent
// 2 x 2-neighbourhood map of z + the transposed conv's bias through the taps (networks.py:302-318).
// compose_up_kernel builds, once per call (the weights change every step):
//   wz[stage][lane = 16 lk + li][4 m + 2 tj + j], K-slot q = 4 m + lk = (channel ci = q / 3, low-res row offset index v = q % 3):
//       sum over c' and the taps (dy, dx) that land on (v, sub-row a) / (column offset tj + j - 1, sub-column b) for output parity
//       (pY = li >> 3, pX = j) of  W[co = li & 7][Cs + c'][dy][dx] * Wt[8 stage + ci][c'][a][b]
//   tb[co] = {R0, R2, C0, C2, T00, T02, T20, T22, S} with T[co][dy][dx] = sum_c' W[co][Cs + c'][dy][dx] * bt[c']
namespace {
constexpr int COMPOSE_MAX = 2 * PC_MAX_GROUP;      // both Up levels of a forward pass in one launch
struct ComposeArgs {
    const float* w[COMPOSE_MAX]; const float* wt[COMPOSE_MAX]; const float* bt[COMPOSE_MAX]; float* ws[COMPOSE_MAX];
    int Cs[COMPOSE_MAX], C[COMPOSE_MAX];
};
__device__ __forceinline__ void up_rowmap(int p, int d, int& v, int& a) {      // parity p, tap d -> low-res offset index v (0..2), sub-pixel a
    const int t = p + d - 1;
    const int i = t < 0 ? -1 : (t >> 1);
    v = i + 1;
    a = t - 2 * i;
}
__global__ __launch_bounds__(256) void compose_up_kernel(const ComposeArgs a) {
    // the two small weight tensors go to LDS first (coalesced), the 64-term sums then read LDS (the direct form spent 18 us
    // of dependent L2 round trips per call)
    __shared__ float sW[8 * 16 * 9], sT[16 * 16 * 4], sB[16];
    const float* W = a.w[blockIdx.y];
    const float* Wt = a.wt[blockIdx.y];
    const float* bt = a.bt[blockIdx.y];
    float* ws = a.ws[blockIdx.y];
    const int C = a.C[blockIdx.y], Cs = a.Cs[blockIdx.y], Ct = Cs + C;
    for (int e = threadIdx.x; e < 8 * C * 9; e += 256) {
        const int co = e / (C * 9), r = e - co * C * 9;
        sW[e] = W[(co * Ct + Cs) * 9 + r];                      // [co][c'][tap] of the up half
    }
    for (int e = threadIdx.x; e < C * C * 4; e += 256) sT[e] = Wt[e];
    if (threadIdx.x < C) sB[threadIdx.x] = bt ? bt[threadIdx.x] : 0.f;
    __syncthreads();
    const int nwz = (C / 8) * 1536;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < nwz + 72 + 2048; e += gridDim.x * 256) {
        if (e >= nwz + 72) {
            // operand image of the BACKWARD data gradient (up_bwd.hip): wd[(4 (4 co + r) + c) * 16 + ci] = Kd[co][ci][r][c], the weight
            // of output pixel (2i - 1 + r, 2j - 1 + c) in dL/dz[ci][i][j]: window row r -> (pY, v) = (1,2), (0,1), (1,1), (0,0)
            const int k = e - nwz - 72, ci = k & 15, c = (k >> 4) & 3, r = (k >> 6) & 3, co = k >> 8;
            float acc = 0.f;
            if (ci < C) {
                const int pY = (r & 1) ^ 1, v = r == 0 ? 2 : (r == 3 ? 0 : 1);
                const int pX = (c & 1) ^ 1, c3 = c == 0 ? 2 : (c == 3 ? 0 : 1);
                for (int dy = 0; dy < 3; ++dy) {
                    int vv, sa;
                    up_rowmap(pY, dy, vv, sa);
                    if (vv != v) continue;
                    for (int dx = 0; dx < 3; ++dx) {
                        int cc, sb;
                        up_rowmap(pX, dx, cc, sb);
                        if (cc != c3) continue;
                        for (int cp = 0; cp < C; ++cp) acc += sW[(co * C + cp) * 9 + dy * 3 + dx] * sT[((ci * C + cp) * 2 + sa) * 2 + sb];
                    }
                }
            }
            ws[e] = acc;
        } else if (e < nwz) {
            const int stage = e / 1536, r = e - stage * 1536, lane = r / 24, qq = r - lane * 24;
            const int lk = lane >> 4, li = lane & 15, pY = li >> 3, co = li & 7;
            const int m = qq >> 2, tj = (qq >> 1) & 1, j = qq & 1;
            const int qs = 4 * m + lk, ci = stage * 8 + qs / 3, vrow = qs % 3;
            float acc = 0.f;
            for (int dy = 0; dy < 3; ++dy) {
                int v, sa;
                up_rowmap(pY, dy, v, sa);
                if (v != vrow) continue;
                for (int dx = 0; dx < 3; ++dx) {
                    int c3, sb;
                    up_rowmap(j, dx, c3, sb);
                    if (c3 != tj + j) continue;
                    for (int c = 0; c < C; ++c) acc += sW[(co * C + c) * 9 + dy * 3 + dx] * sT[((ci * C + c) * 2 + sa) * 2 + sb];
                }
            }
            ws[e] = acc;
        } else {
            const int k = e - nwz, co = k / 9, which = k - 9 * co;
            float T[3][3];
            for (int dy = 0; dy < 3; ++dy)
                for (int dx = 0; dx < 3; ++dx) {
                    float t = 0.f;
                    for (int c = 0; c < C; ++c) t += sW[(co * C + c) * 9 + dy * 3 + dx] * sB[c];
                    T[dy][dx] = t;
                }
            float v;
            switch (which) {
                case 0: v = T[0][0] + T[0][1] + T[0][2]; break;
                case 1: v = T[2][0] + T[2][1] + T[2][2]; break;
                case 2: v = T[0][0] + T[1][0] + T[2][0]; break;
                case 3: v = T[0][2] + T[1][2] + T[2][2]; break;
                case 4: v = T[0][0]; break;
                case 5: v = T[0][2]; break;
                case 6: v = T[2][0]; break;
                case 7: v = T[2][2]; break;
                default: v = T[0][0] + T[0][1] + T[0][2] + T[1][0] + T[1][1] + T[1][2] + T[2][0] + T[2][1] + T[2][2]; break;
            }
            ws[nwz + k] = v;
        }
    }
}
}  // namespace

extern "C" int64_t pc_conv3x3_up_ws_bytes(int C) { return (int64_t)((C / 8) * 1536 + 72 + 2048) * sizeof(float); }

extern "C" int pc_conv3x3_up_fwd_ok(const pc_src* skip, const pc_src* z, const pc_dst* out, int H, int W, int Cs, int C) {
    if (g_pc_precision != PC_PREC_FP32 || !skip || !z || !out) return 0;
    // (any even width with 16-byte aligned rows: the low-resolution piece is masked per column, the border bias per pixel, the ragged
    // last strip of a row is stored by the per-element epilogue)
    if (!((Cs == 8 && C == 8) || (Cs == 16 && C == 16)) || (H & 3) || (W & 1)) return 0;
    if (skip->C != Cs || z->C != C || z->H * 2 != H || z->W * 2 != W || z->mode != PC_SRC_DIRECT || z->oy || z->ox) return 0;
    if (z->dtype != PC_F32 || !pc_planar(*z) || skip->dtype != PC_F32 || !pc_planar(*skip) || out->dtype != PC_F32 || !pc_planar(*out)) return 0;
    if (conv_src_mode(*skip, H, W) != 1) return 0;
    return (out->rstride % 4 == 0) && (out->cstride % 4 == 0) && (out->bstride % 4 == 0) && ((reinterpret_cast<uintptr_t>(out->ptr) & 15) == 0);
}

// composed operand images of up to 2 * PC_MAX_GROUP Up-block convolutions (any mix of (Cs, C) = (8, 8) / (16, 16)) in ONE launch:
// d[i].ws <- compose(d[i].w, d[i].wt, d[i].bt); a following pc_conv3x3_up_fwd_group(..., relu | PC_UP_PRECOMPOSED, ...) skips its own
extern "C" int pc_conv3x3_up_compose_group(int n, const pc_conv_up_fwd_desc* d, const int* Cs, const int* C, void* stream) {
    if (n < 1 || n > COMPOSE_MAX || !d || !Cs || !C) return PC_EINVAL;
    ComposeArgs ca{};
    int cmax = 0;
    for (int i = 0; i < n; ++i) {
        if (!d[i].w || !d[i].wt || !d[i].ws || !((Cs[i] == 8 && C[i] == 8) || (Cs[i] == 16 && C[i] == 16))) return PC_EINVAL;
        ca.w[i] = d[i].w; ca.wt[i] = d[i].wt; ca.bt[i] = d[i].bt; ca.ws[i] = (float*)d[i].ws;
        ca.Cs[i] = Cs[i]; ca.C[i] = C[i];
        if (C[i] > cmax) cmax = C[i];
    }
    hipLaunchKernelGGL(compose_up_kernel, dim3(((cmax / 8) * 1536 + 72 + 2048 + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, ca);
    PC_CHECK_LAUNCH();
    return 0;
}

extern "C" int pc_conv3x3_up_fwd_group(int n, const pc_conv_up_fwd_desc* d, int relu, int B, int H, int W, int Cs, int C, void* stream) {
    if (n < 1 || n > MAXG || !d) return PC_EINVAL;
    const bool precomposed = (relu & PC_UP_PRECOMPOSED) != 0;
    relu &= 1;
    ConvArgs p{};
    ComposeArgs ca{};
    for (int i = 0; i < n; ++i) {
        if (!d[i].skip || !d[i].z || !d[i].w || !d[i].wt || !d[i].bn || !d[i].out || !d[i].ws ||
            !pc_conv3x3_up_fwd_ok(d[i].skip, d[i].z, d[i].out, H, W, Cs, C))
            return PC_EINVAL;
        ConvProb& q = p.pr[i];
        q.a = *d[i].skip;
        q.w = d[i].w;
        q.bn = *d[i].bn;
        q.out = *d[i].out;
        q.z = d[i].z->ptr; q.z_bs = d[i].z->bstride; q.z_cs = d[i].z->cstride; q.z_rs = d[i].z->rstride;
        q.wz = (const float*)d[i].ws;
        q.tb = (const float*)d[i].ws + (C / 8) * 1536;
        q.fast_a = 1;
        ca.w[i] = d[i].w; ca.wt[i] = d[i].wt; ca.bt[i] = d[i].bt; ca.ws[i] = (float*)d[i].ws;
        ca.Cs[i] = Cs; ca.C[i] = C;
    }
    hipStream_t st = (hipStream_t)stream;
    if (!precomposed) {
        hipLaunchKernelGGL(compose_up_kernel, dim3(((C / 8) * 1536 + 72 + 2048 + 255) / 256, n), dim3(256), 0, st, ca);
        PC_CHECK_LAUNCH();
    }
    p.w_co_stride = (Cs + C) * 9;
    p.w_ci_stride = 9;
    p.relu = relu;
    p.B = B; p.H = H; p.W = W;
    p.vec_ok = 1;
    p.tiles_x = (p.W + TW - 1) / TW;
    p.tiles_y = (p.H + TH - 1) / TH;
    p.ntiles = p.B * p.tiles_x * p.tiles_y;
    if (p.ntiles <= 0) return 0;
    p.div_tx = pc_make_fastdiv(p.tiles_x);
    p.div_tpi = pc_make_fastdiv(p.tiles_x * p.tiles_y);
    p.dbg = g_conv_dbg;
    p.ts = g_conv_ts;
    if (Cs == 8) {
        if (fwd_s3_ok<8, 8>(p, n, 8)) return launch_fwd_s3<8, 8, EPI_NONE, 8>(p, n, st);
        return launch_conv_po<8, 8, MODE_FWD, LD_DIRECT, EPI_NONE, 8>(p, n, st);
    }
    if (fwd_s3_ok<16, 8>(p, n, 16)) return launch_fwd_s3<16, 8, EPI_NONE, 16>(p, n, st);
    return launch_conv_po<16, 8, MODE_FWD, LD_DIRECT, EPI_NONE, 16>(p, n, st);
}
