// conv3x3_fwd_s3.h -- PC_PREC_FP32 forward conv3x3 + BN + ReLU on the bf16 matrix pipe (round 6), included by conv3x3.hip inside its
// anonymous namespace (it uses ConvArgs / ConvProb, the strip geometry and the EPI_* constants defined there).
//
// The forward counterpart of conv3x3_bwd_s3_kernel (conv3x3_bwd.hip) for every forward layer with 8 or 16 input and output channels on
// aligned planar fp32 tensors: DoubleConv / Down.conv (networks.py:259-294) incl. the pooled second output and the partial 1x1 logit, and
// the first conv of an Up block taken straight from the low-resolution map (Up: networks.py:302-318; composed weights: compose_up_kernel
// above).  In the step (B = 64, 4 problems per launch) the 16-channel and composed layers were bound by the fp32 matrix pipe (-15 .. -25 %
// each in this form); the plain 8 -> 8 layers move their bytes at the strip pattern's rate in either form and gain 10 - 20 %.
//   * planar fp32 tensors at both ends; a strip (32 x 4 outputs, 6 x 34 inputs) of 8 input channels is loaded as one aligned 16-byte
//     piece per lane and channel, prefetched one STAGE ahead, and split ONCE while it is written to LDS: every fp32 value is exactly the
//     sum of three bf16 numbers (common.h: pc_split_pair), so a lane turns 4 pixels x 8 channels into 3 planes x 4 channels-last 16-byte
//     slots; one ds_read_b128 is then the K = 32 = (4 strip rows x 8 channels) pixel operand of v_mfma_f32_16x16x32_bf16;
//   * every product is six bf16 x bf16 partial products (the three smallest of the nine, together below 2^-23 of the product, are
//     dropped), accumulated in fp32 smallest first: 6 x 16 matrix cycles per 32 K-slots against 8 x 32 on v_mfma_f32_16x16x4_f32;
//   * a strip runs NST stages through ONE wave-private image: the 8-channel chunks of the input, then (ZC > 0) the 8-channel chunks of
//     the low-resolution map z (4 rows x 18 columns); the weights live in LDS for the whole kernel (split once in the prologue) and a
//     stage reads its B fragments from there (NST == 1: once, before the loop);
//   * ZC > 0 uses the paired-pixel mapping of conv3x3_mfma_kernel's composed stage (M index i = pixel 2 i + j, one accumulator per
//     x parity j): the strip image stores even and odd pixels in two blocks of 17 slots so that every operand read is unit-stride;
//   * A = pixels, B = weights: a lane ends with FOUR CONSECUTIVE PIXELS of one output channel = the planar fp32 epilogues of
//     conv3x3_mfma_kernel (BN + ReLU, 16-byte stores, the 2 x 2 max-pooled second output, the transposed conv's bias through the taps).
// The result differs from the fp32-MFMA form by the dropped partial products only (tests/test_gpu_conv_fwd_split.py measures both against
// float64); pc_set_conv_split(0) selects the fp32-MFMA kernels everywhere.

template <int CI, int CO, int ZC>
struct FS3Cfg {
    static constexpr int NCH = CI / 8, NZ = ZC / 8, NST = NCH + NZ, NB = CO / 8;
    static constexpr int BSL = 34;                              // slots per strip row: logical slot s = pixel x0 - 1 + s
    static constexpr int ZSL = 18;                              // slots per row of the low-resolution piece: slot t = column x0 / 2 - 1 + t
    static constexpr int DUMMY = SROWS * BSL;                   // one more slot per plane: where a lane writes the pixels that are not part of the strip
    static constexpr int IMG = (SROWS * BSL + 1) * 16;          // bytes of one split plane of a wave's image
    static constexpr int WAVE_B = 3 * IMG;
    static constexpr int BW_CO = NCH * 24, BW_DYS = CO * BW_CO, WPL = 3 * BW_DYS;    // weight image, bf16 elements: [dy 0..2][co][chunk][dx][8 ci]
    static constexpr int ZW_N = NZ * 32, ZW_V = 16 * ZW_N, ZPL = 3 * ZW_V;           // composed weights: [row v 0..2][n = (s, co)][chunk][2 tj + j][8 zci]
    static constexpr size_t LDS_B = (size_t)4 * WAVE_B + (size_t)3 * (WPL + ZPL) * 2;
    static constexpr int WAVES_PER_SIMD = (size_t)3 * LDS_B <= 160 * 1024 ? 3 : 2;   // (168 registers then; the 16 + 16z -> 8 layer keeps 2)
};

// physical slot of logical slot s inside a strip row.  Plain layers: conv3x3_bwd_s3_kernel's XOR swizzle (tools/lds_bank_sim.py); composed
// layers: even pixels in slots 0..16, odd pixels in 17..33
__device__ __forceinline__ int fs3_sl(int s) { return s ^ ((s >> 3) & 3); }
__device__ __forceinline__ int fs3_sl_p2(int s) { return (s & 1) * 17 + (s >> 1); }

template <typename T>
__device__ __forceinline__ T fs3_pin(T v) {
    asm volatile("" : "+s"(v));
    return v;
}
typedef __attribute__((address_space(1))) char* fs3_gptr;
typedef __attribute__((address_space(1))) const f32x4* fs3_gld4;
typedef __attribute__((address_space(1))) f32x4* fs3_gst4;
typedef float fs3_f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) fs3_f32x2* fs3_gst2;
__device__ __forceinline__ fs3_gptr fs3_pin_global(const void* ptr) {
    uint64_t v = reinterpret_cast<uint64_t>(ptr);
    asm volatile("" : "+s"(v));
    return (fs3_gptr)v;
}
template <int N>
using fs3_int = std::integral_constant<int, N>;

template <int CI, int CO, int EPI, int ZC>
__global__ __launch_bounds__(256, (FS3Cfg<CI, CO, ZC>::WAVES_PER_SIMD)) void conv3x3_fwd_s3_kernel(const ConvArgs p) {
    using Cfg = FS3Cfg<CI, CO, ZC>;
    constexpr int NCH = Cfg::NCH, NZ = Cfg::NZ, NST = Cfg::NST, NB = Cfg::NB, BSL = Cfg::BSL, ZSL = Cfg::ZSL, IMG = Cfg::IMG;
    constexpr int BW_CO = Cfg::BW_CO, BW_DYS = Cfg::BW_DYS, WPL = Cfg::WPL, ZW_N = Cfg::ZW_N, ZPL = Cfg::ZPL;
    constexpr bool P2 = ZC > 0;
    static_assert(ZC == 0 || (CO == 8 && EPI == EPI_NONE), "composed stage: 8 outputs, plain epilogue");
    static_assert(EPI == EPI_NONE || EPI == EPI_POOL || (EPI == EPI_DOT && CI == 8 && CO == 8), "epilogues of the split-form forward kernel");
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    const ConvProb& q = p.pr[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int s_row = li >> 3, col = li & 7;
    unsigned char* const wimg = ldsb + wave * Cfg::WAVE_B;                                  // split plane pl at wimg + pl * IMG
    unsigned short* const w2h = reinterpret_cast<unsigned short*>(ldsb + 4 * Cfg::WAVE_B);    // weight image: plane pl at w2h + pl * WPL
    unsigned short* const wzh = w2h + 3 * WPL;                                              // composed weights: plane pl at wzh + pl * ZPL

    // ---- wave-uniform descriptor fields, pinned in scalar registers for the whole kernel (common.h: why)
    const fs3_gptr a_base = fs3_pin_global(q.a.ptr), o_base = fs3_pin_global(q.out.ptr);
    const unsigned a_bs = fs3_pin((unsigned)q.a.bstride), a_cs = fs3_pin((unsigned)q.a.cstride * 4u), a_rs = fs3_pin((unsigned)q.a.rstride);
    const unsigned o_bs = fs3_pin((unsigned)q.out.bstride), o_cs = fs3_pin((unsigned)q.out.cstride), o_rs = fs3_pin((unsigned)q.out.rstride);
    const fs3_gptr z_base = fs3_pin_global(ZC > 0 ? (const void*)q.z : (const void*)q.a.ptr);
    const unsigned z_bs = fs3_pin((unsigned)(ZC > 0 ? q.z_bs : 0)), z_cs = fs3_pin((unsigned)(ZC > 0 ? q.z_cs : 0) * 4u),
                   z_rs = fs3_pin((unsigned)(ZC > 0 ? q.z_rs : 0));
    const bool has_po = EPI == EPI_POOL && q.pool_out.ptr != nullptr;
    const fs3_gptr po_base = fs3_pin_global(has_po ? (const void*)q.pool_out.ptr : (const void*)q.out.ptr);
    const unsigned po_bs = fs3_pin((unsigned)(has_po ? q.pool_out.bstride : 0)), po_cs = fs3_pin((unsigned)(has_po ? q.pool_out.cstride : 0)),
                   po_rs = fs3_pin((unsigned)(has_po ? q.pool_out.rstride : 0));
    const int H = fs3_pin(p.H), W = fs3_pin(p.W), relu = fs3_pin(p.relu);
    // ablation switches (pc_debug_conv; tools/time_conv_fwd.py --ablate) exist in -DPOPCORN_CONV_ABLATE builds only (tools/build_variant.sh): as
    // run-time flags they made the commit conditional, and a path that may skip it leaves its loads pending -- the compiler then waits
    // vmcnt(0) before it re-uses their registers for the next prefetch, i.e. behind the epilogue's stores
#ifdef POPCORN_CONV_ABLATE
    const int dbg = fs3_pin(p.dbg);
#else
    constexpr int dbg = 0;
#endif
    // EPI_DOT: problems with dot_w write sum_co dot_w[co] * out[co] (the partial logit of the 1x1 out-conv that follows) instead of the map
    const bool has_dot = EPI == EPI_DOT && q.dot_w != nullptr;
    const fs3_gptr d_base = fs3_pin_global(has_dot ? (const void*)q.dot_out.ptr : (const void*)q.out.ptr);
    const unsigned d_bs = fs3_pin((unsigned)(has_dot ? q.dot_out.bstride : 0)), d_rs = fs3_pin((unsigned)(has_dot ? q.dot_out.rstride : 0));
    float dot_wl = 0.f;
    if constexpr (EPI == EPI_DOT) {
        if (has_dot) dot_wl = q.dot_w[col];
    }

    // ---- loaders.  Input chunk: lane = (row of the 6-row strip, 16-byte segment of the 40-float row x0 - 4 ..).  Low-resolution piece:
    //      lane = (row of the 4 rows y0 / 2 - 1 .., 16-byte segment of the 24-float row x0 / 2 - 4 ..).  Byte offsets inside a tensor are
    //      32-bit (the launcher checks the extents): one scalar base per channel + one vector offset per lane
    const int l_r = lane / 10, l_seg = lane - l_r * 10;
    const bool l_act = lane < 60;
    const int z_r = lane / 6, z_s = lane - z_r * 6;
    const bool z_act = ZC > 0 && lane < 24;
    f32x4 R[8];
    bool rvalid = false;
    // slot (relative to the wave's image) of each of the lane's four pixels.  Pixels that are not part of the strip (the outer three of the
    // first and last segment, idle lanes) go to a dummy slot: the LDS writes are UNCONDITIONAL, so every path consumes all eight loaded
    // registers and the compiler knows no load is pending when the next prefetch is issued (with predicated writes it waited vmcnt(0) there
    // -- behind the epilogue's stores)
    int c_sl[4], cz_sl[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int sl = 4 * l_seg + e - 3;
        c_sl[e] = (l_act && sl >= 0 && sl <= 33) ? l_r * BSL + (P2 ? fs3_sl_p2(sl) : fs3_sl(sl)) : Cfg::DUMMY;
        const int t = 4 * z_s + e - 3;
        cz_sl[e] = (z_act && t >= 0 && t <= 17) ? z_r * ZSL + t : Cfg::DUMMY;
    }
    auto issue = [&](auto CH, int b, int y0, int x0) {
        constexpr int ch = decltype(CH)::value;
        if constexpr (ch < NCH) {
            const int xg = x0 - 4 + 4 * l_seg, y = y0 - 1 + l_r;
            const bool ok = l_act && xg >= 0 && xg < W && (unsigned)y < (unsigned)H;
            rvalid = ok;
            const unsigned off = ok ? ((unsigned)b * a_bs + (unsigned)y * a_rs + (unsigned)xg) * 4u : 0u;
#pragma unroll
            for (int it = 0; it < 8; ++it) R[it] = *(fs3_gld4)(a_base + (size_t)((ch * 8 + it) * a_cs) + off);
        } else {
            const int I = (y0 >> 1) - 1 + z_r, c = (x0 >> 1) - 4 + 4 * z_s;
            const bool ok = z_act && (unsigned)I < (unsigned)(H >> 1) && c >= 0 && c < (W >> 1);     // (W / 2) % 4 == 0: whole segments
            rvalid = ok;
            const unsigned off = ok ? ((unsigned)b * z_bs + (unsigned)I * z_rs + (unsigned)c) * 4u : 0u;
#pragma unroll
            for (int it = 0; it < 8; ++it) R[it] = *(fs3_gld4)(z_base + (size_t)(((ch - NCH) * 8 + it) * z_cs) + off);
        }
    };
    // split + transpose while staging: the 8 channels of a pixel = one 16-byte slot per plane
    auto commit = [&](auto CH) {
        constexpr bool isz = decltype(CH)::value >= NCH;
        const bool act = isz ? z_act : l_act;
        // strips that touch the image border: zero what lies outside (wave-uniform test; interior strips skip the selects)
        if (__builtin_amdgcn_ballot_w64(act && !rvalid) != 0) {
#pragma unroll
            for (int it = 0; it < 8; ++it)
#pragma unroll
                for (int e = 0; e < 4; ++e) R[it][e] = rvalid ? R[it][e] : 0.f;
        }
        u32x4* const d0 = reinterpret_cast<u32x4*>(wimg);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            u32x4 q1, q2, q3;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                unsigned a1, a2, a3;
                pc_split_pair(R[2 * d][e], R[2 * d + 1][e], a1, a2, a3);
                q1[d] = a1; q2[d] = a2; q3[d] = a3;
            }
            u32x4* d = d0 + (isz ? cz_sl[e] : c_sl[e]);
            d[0] = q1;
            d[IMG / 16] = q2;
            d[2 * (IMG / 16)] = q3;
        }
    };
    const int gdim = fs3_pin((int)gridDim.x), ntl = fs3_pin(p.ntiles);
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
        if (!(dbg & 1)) issue(fs3_int<0>{}, b, y0, x0);         // in flight while the weights are staged
    }

    // ---- weights, split once: three planes of [dy 0..2][co][chunk][dx][8 ci] bf16; composed weights [v 0..2][n][chunk][2 tj + j][8 zci]
    //      from the fp32 image of compose_up_kernel (ws[stage * 1536 + lane * 24 + 4 m + 2 tj + j]: K-slot 4 m + (lane >> 4) = 3 zci + v).
    //      The lanes whose (strip row, output row) pair is not a tap (dy = lk - s_row outside 0..2; v = lk = 3) zero their B fragments
    //      in registers -- an all-zero fourth plane cost 4 KB of LDS, the third workgroup of a CU
    constexpr int NWR = (CO * CI * 9 + 255) / 256;
    float wreg[NWR];
#pragma unroll
    for (int k = 0; k < NWR; ++k) {
        const int e = tid + k * 256;
        const int ec = e < CO * CI * 9 ? e : 0;
        const int tap = ec % 9, ci = (ec / 9) % CI, co = ec / (9 * CI);
        wreg[k] = q.w[co * p.w_co_stride + ci * p.w_ci_stride + tap];
    }
    float wzreg[ZC > 0 ? NZ * 6 : 1];
    if constexpr (ZC > 0) {
#pragma unroll
        for (int k = 0; k < NZ * 6; ++k) wzreg[k] = q.wz[tid + 256 * k];
    }
    float bn_raw[NB][5];            // {conv bias, gamma, var, mean, beta} of channel nb*8 + col
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int c = nb * 8 + col;
        bn_raw[nb][0] = q.bn.conv_bias ? q.bn.conv_bias[c] : 0.f;
        bn_raw[nb][1] = q.bn.gamma ? q.bn.gamma[c] : 1.f;
        bn_raw[nb][2] = q.bn.gamma ? q.bn.var[c] : 1.f;
        bn_raw[nb][3] = q.bn.gamma ? q.bn.mean[c] : 0.f;
        bn_raw[nb][4] = q.bn.gamma ? q.bn.beta[c] : 0.f;
    }
    float zb[ZC > 0 ? 9 : 1];
    if constexpr (ZC > 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) zb[k] = q.tb[col * 9 + k];
    }
    auto split3 = [](float w, unsigned short& h1, unsigned short& h2, unsigned short& h3) {
        const float a1 = pc_bf16r(w), r1 = w - a1, a2 = pc_bf16r(r1), a3 = r1 - a2;
        h1 = (unsigned short)(__float_as_uint(a1) >> 16);
        h2 = (unsigned short)(__float_as_uint(a2) >> 16);
        h3 = pc_f2bf(a3);
    };
#pragma unroll
    for (int k = 0; k < NWR; ++k) {
        const int e = tid + k * 256;
        if (e < CO * CI * 9) {
            const int tap = e % 9, ci = (e / 9) % CI, co = e / (9 * CI);
            const int o = (tap / 3) * BW_DYS + co * BW_CO + (ci / 8) * 24 + (tap % 3) * 8 + (ci % 8);
            split3(wreg[k], w2h[o], w2h[WPL + o], w2h[2 * WPL + o]);
        }
    }
    if constexpr (ZC > 0) {
#pragma unroll
        for (int k = 0; k < NZ * 6; ++k) {
            const int e = tid + 256 * k, zc = e / 1536, r = e - zc * 1536;
            const int ln = r / 24, k24 = r - ln * 24, m = k24 >> 2, tjj = k24 & 3;
            const int qs = 4 * m + (ln >> 4), zci = qs / 3, v = qs - 3 * zci;
            const int o = (v * 16 + (ln & 15)) * ZW_N + zc * 32 + tjj * 8 + zci;
            split3(wzreg[k], wzh[o], wzh[ZPL + o], wzh[2 * ZPL + o]);
        }
    }
    // folded BN (pc_bn_fold): scale = gamma / sqrt(var + eps); shift = (bias - mean) * scale + beta
    float e_scale[NB], e_shift[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        if (q.bn.gamma) {
            e_scale[nb] = bn_raw[nb][1] * (1.0f / sqrtf(bn_raw[nb][2] + q.bn.eps));
            e_shift[nb] = (bn_raw[nb][0] - bn_raw[nb][3]) * e_scale[nb] + bn_raw[nb][4];
        } else {
            e_scale[nb] = 1.f;
            e_shift[nb] = bn_raw[nb][0];
        }
    }
    if constexpr (ZC > 0) {
        // the transposed conv's bias reaches the output through every tap whose up-sampled pixel lies inside the image: the full sum joins
        // the shift, border rows / columns / corners subtract their missing taps in the epilogue (conv3x3_mfma_kernel)
        e_shift[0] += zb[8] * e_scale[0];
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("" : : "v"(zb[k]));
    }
    // (consumed here: a value first used inside the loop keeps the compiler's vmcnt bookkeeping pessimistic for every iteration)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) asm volatile("" : : "v"(e_scale[nb]), "v"(e_shift[nb]));
    __syncthreads();

    // ---- B fragments: lane (n = (s_row, col), K group lk = strip row of the pair's four) reads plane dy = lk - s_row
    const bool tap_ok = (unsigned)(lk - s_row) <= 2u, zrow_ok = lk < 3;
    const unsigned short* const wlane = w2h + (tap_ok ? lk - s_row : 0) * BW_DYS + col * BW_CO;
    const unsigned short* const wzlane = wzh + ((zrow_ok ? lk : 0) * 16 + li) * ZW_N;
    const u32x4 zero4 = u32x4{0u, 0u, 0u, 0u};
    bf16x8 wq[3][3][NB];
    auto load_wq = [&](int ch) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                {
                    const u32x4 t = *reinterpret_cast<const u32x4*>(wlane + pl * WPL + nb * 8 * BW_CO + ch * 24 + dx * 8);
                    wq[pl][dx][nb] = __builtin_bit_cast(bf16x8, tap_ok ? t : zero4);
                }
    };
    constexpr bool WQ_HOIST = NST == 1 && NB == 1;             // the B fragments of the only stage stay in registers for the whole kernel
    constexpr bool WQ_PER_DX = !P2 && !WQ_HOIST;               // 16 outputs / several stages: 3 x NB fragments at a time (72 registers otherwise)
    if constexpr (WQ_HOIST) load_wq(0);

    // ---- pixel-operand slots
    int dg_sl[3][2];                                            // plain mapping: M index li = pixel li + 16 h
    int p2_sl[4];                                               // paired mapping: M index li = pixel 2 li + j; operand (j, dx) = logical slot 2 li + j + dx
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int h = 0; h < 2; ++h) dg_sl[dx][h] = fs3_sl(li + dx + 16 * h);
#pragma unroll
    for (int t = 0; t < 4; ++t) p2_sl[t] = fs3_sl_p2(2 * li + t);

    f32x4 acc[4][NB];
    auto matrix = [&](auto CH) {
        constexpr int ch = decltype(CH)::value;
        if constexpr (ch < NCH) {
            if constexpr (!WQ_HOIST && !WQ_PER_DX) load_wq(ch);
            if constexpr (!P2) {
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    if constexpr (WQ_PER_DX) {
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb) {
                                const u32x4 t = *reinterpret_cast<const u32x4*>(wlane + pl * WPL + nb * 8 * BW_CO + ch * 24 + dx * 8);
                                wq[pl][dx][nb] = __builtin_bit_cast(bf16x8, tap_ok ? t : zero4);
                            }
                    }
                    bf16x8 av[3][4];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        const u32x4* lrow = reinterpret_cast<const u32x4*>(wimg + pl * IMG) + lk * BSL;
#pragma unroll
                        for (int u = 0; u < 4; ++u) av[pl][u] = __builtin_bit_cast(bf16x8, lrow[(u >> 1) * 2 * BSL + dg_sl[dx][u & 1]]);
                    }
#pragma unroll
                    for (int pw = 2; pw >= 0; --pw)           // weight split index; pixel split indices 2 - pw .. 0: smallest products first
#pragma unroll
                        for (int pa = 2 - pw; pa >= 0; --pa)
#pragma unroll
                            for (int u = 0; u < 4; ++u)
#pragma unroll
                                for (int nb = 0; nb < NB; ++nb)
                                    acc[u][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[pa][u], wq[pw][dx][nb], acc[u][nb], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int rp = 0; rp < 2; ++rp) {
                    bf16x8 av[3][4];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        const u32x4* lrow = reinterpret_cast<const u32x4*>(wimg + pl * IMG) + (2 * rp + lk) * BSL;
#pragma unroll
                        for (int t = 0; t < 4; ++t) av[pl][t] = __builtin_bit_cast(bf16x8, lrow[p2_sl[t]]);
                    }
#pragma unroll
                    for (int pw = 2; pw >= 0; --pw)
#pragma unroll
                        for (int pa = 2 - pw; pa >= 0; --pa)
#pragma unroll
                            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                                for (int j = 0; j < 2; ++j)
                                    acc[rp * 2 + j][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[pa][j + dx], wq[pw][dx][0], acc[rp * 2 + j][0], 0, 0, 0);
                }
            }
        } else {
            // composed stage: K group lk = low-resolution row v of the three a row pair touches (piece row rp + v; lk = 3 carries zero
            // weights and re-reads row 3), operand (tj, j) = column li + tj + j of the piece
            constexpr int zc = ch - NCH;
            bf16x8 wz[3][4];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const u32x4 w4 = *reinterpret_cast<const u32x4*>(wzlane + pl * ZPL + zc * 32 + t * 8);
                    wz[pl][t] = __builtin_bit_cast(bf16x8, zrow_ok ? w4 : zero4);
                }
#pragma unroll
            for (int rp = 0; rp < 2; ++rp) {
                const int zr = rp + lk > 3 ? 3 : rp + lk;
                bf16x8 zv[3][3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const u32x4* lrow = reinterpret_cast<const u32x4*>(wimg + pl * IMG) + zr * ZSL + li;
#pragma unroll
                    for (int t = 0; t < 3; ++t) zv[pl][t] = __builtin_bit_cast(bf16x8, lrow[t]);
                }
#pragma unroll
                for (int pw = 2; pw >= 0; --pw)
#pragma unroll
                    for (int pa = 2 - pw; pa >= 0; --pa)
#pragma unroll
                        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                            for (int j = 0; j < 2; ++j)
                                acc[rp * 2 + j][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zv[pa][tj + j], wz[pw][2 * tj + j], acc[rp * 2 + j][0], 0, 0, 0);
            }
        }
    };

    // Deferred epilogue: finish() turns the accumulators of strip k into final values (pacc), store_prev() writes them in strip k + 1,
    // BETWEEN its first commit and the prefetch it issues.  With the stores at the end of their own strip they sat behind the prefetch in
    // the memory queue; they are predicated (execz branches), so the compiler cannot count them and the commit's wait for the prefetched
    // loads became s_waitcnt vmcnt(0) -- a store round trip in series with every strip (conv3x3_mfma_kernel defers for the same reason).
    f32x4 pacc[4][NB];
    int eb = 0, ey0 = 0, ex0 = 0;
    bool have_prev = false;
    auto finish = [&](int fb, int fy0, int fx0) {
        // lane holds (co = nb*8 + col, y = fy0 + 2*(u>>1) + s_row, x = fx0 + xu(u) + r), r = 0..3
        eb = fb; ey0 = fy0; ex0 = fx0;
        have_prev = true;
        if constexpr (P2) {
            // acc[rp*2 + j][r] = pixel 2 (4 lk + r) + j  ->  pacc[rp*2 + h][e] = pixel 8 lk + 4 h + e
#pragma unroll
            for (int rp = 0; rp < 2; ++rp)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 a0 = acc[rp * 2][0], a1 = acc[rp * 2 + 1][0];
                    f32x4 v = f32x4{a0[2 * h], a1[2 * h], a0[2 * h + 1], a1[2 * h + 1]};
                    // border pixels: the taps whose up-sampled pixel falls outside the image carry no bias
                    const int Y = ey0 + 2 * rp + s_row, X0 = ex0 + 8 * lk + 4 * h;
                    const bool top = Y == 0, bot = Y == H - 1;
                    const float rowt = top ? zb[0] : (bot ? zb[1] : 0.f);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool lft = X0 + e == 0, rgt = X0 + e == W - 1;
                        float c = rowt + (lft ? zb[2] : (rgt ? zb[3] : 0.f));
                        c -= top ? (lft ? zb[4] : (rgt ? zb[5] : 0.f)) : (bot ? (lft ? zb[6] : (rgt ? zb[7] : 0.f)) : 0.f);
                        v[e] -= c;
                    }
                    pacc[rp * 2 + h][0] = v;
                }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) pacc[u][nb] = acc[u][nb];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float o = pacc[u][nb][r] * e_scale[nb] + e_shift[nb];
                    pacc[u][nb][r] = relu ? fmaxf(o, 0.f) : o;
                }
    };
    auto store_prev = [&]() {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int y = ey0 + 2 * (u >> 1) + s_row, x = ex0 + (P2 ? 8 * lk + 4 * (u & 1) : (u & 1) * 16 + 4 * lk);
            const bool in = y < H && x < W;                     // W % 4 == 0: a 4-pixel piece is inside or outside as a whole
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const f32x4 v = pacc[u][nb];
                if constexpr (EPI == EPI_DOT) {
                    if (has_dot) {
                        // sum over the 8 channels = the 8 lanes `col` of a 16-lane group half (they hold the same pixels); lane col == 0 stores
                        f32x4 t;
#pragma unroll
                        for (int r = 0; r < 4; ++r) t[r] = pc_sum8(__fmul_rn(v[r], dot_wl));      // (rounded product: no fma contraction into the first step)
                        if (in && col == 0) *(fs3_gst4)(d_base + (size_t)(((unsigned)eb * d_bs + (unsigned)y * d_rs + (unsigned)x) * 4u)) = t;
                        continue;
                    }
                }
                if (in)
                    *(fs3_gst4)(o_base + (size_t)(((unsigned)eb * o_bs + (unsigned)(nb * 8 + col) * o_cs + (unsigned)y * o_rs + (unsigned)x) * 4u)) = v;
                if constexpr (EPI == EPI_POOL) {
                    // MaxPool2d(2) (full strips only: pool_out_geometry_ok): the x pairs are in the lane, the row pair (s_row 0 / 1) sits 8 lanes apart
                    if (has_po) {
                        float m0 = fmaxf(v[0], v[1]), m1 = fmaxf(v[2], v[3]);
                        m0 = fmaxf(m0, pc_lane_xor8(m0));
                        m1 = fmaxf(m1, pc_lane_xor8(m1));
                        if (s_row == 0)
                            *(fs3_gst2)(po_base + (size_t)(((unsigned)eb * po_bs + (unsigned)(nb * 8 + col) * po_cs + (unsigned)(y >> 1) * po_rs +
                                                             (unsigned)(x >> 1)) * 4u)) = fs3_f32x2{m0, m1};
                    }
                }
            }
        }
    };

    for (int k = 0; k < my_tiles; ++k) {
        int nb_ = b, ny0 = y0, nx0 = x0;
        const bool more = k + 1 < my_tiles;
        if (more) strip_coords(k + 1, nb_, ny0, nx0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[u][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto stage = [&](auto CH) {
            constexpr int ch = decltype(CH)::value;
            if (!(dbg & (1 | 32))) commit(CH);                  // ablation (pc_debug_conv): 1 no loads / LDS writes, 32 no split + LDS writes, 64 no loads,
            if constexpr (ch == 0) {
                // (fences: the scheduler otherwise hoists the stores between the commit's per-pixel blocks, whose waits then cover them)
                __builtin_amdgcn_sched_barrier(0);
                if (have_prev && !(dbg & 4)) store_prev();
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (ch + 1 < NST) {                       //                           2 no matrix phase, 4 no epilogue
                if (!(dbg & (1 | 64))) issue(fs3_int<ch + 1>{}, b, y0, x0);
            } else {
                if (more && !(dbg & (1 | 64))) issue(fs3_int<0>{}, nb_, ny0, nx0);
            }
            if (!(dbg & 2)) matrix(CH);
        };
        stage(fs3_int<0>{});
        if constexpr (NST > 1) stage(fs3_int<1>{});
        if constexpr (NST > 2) stage(fs3_int<2>{});
        if constexpr (NST > 3) stage(fs3_int<3>{});
        finish(b, y0, x0);
        b = nb_; y0 = ny0; x0 = nx0;
    }
    if (have_prev && !(dbg & 4)) store_prev();
}

// every problem of the group qualifies for the split-form kernel: one aligned planar fp32 source of exactly the conv domain, aligned planar
// output, 16-byte rows, 32-bit element offsets inside every tensor
template <int CI, int CO>
bool fwd_s3_ok(const ConvArgs& p, int nprob, int ZC) {
    if (g_pc_precision != PC_PREC_FP32 || pc_conv_split_on() == 0 || p.W % 4 != 0 || p.W < 4) return false;
    static const bool off = [] { const char* e = getenv("POPCORN_FWD_SPLIT"); return e && e[0] == '0'; }();   // A/B runs: the backward kernels keep their form
    if (off) return false;
    if (ZC > 0 && (p.W % 8 != 0 || p.H % 4 != 0)) return false;
    auto fits32 = [&](int64_t bs) { return bs > 0 && (int64_t)p.B * bs < ((int64_t)1 << 30); };
    auto al = [](const void* ptr, int64_t bs, int64_t cs, int64_t rs) {
        return (reinterpret_cast<uintptr_t>(ptr) & 15) == 0 && bs % 4 == 0 && cs % 4 == 0 && rs % 4 == 0;
    };
    for (int i = 0; i < nprob; ++i) {
        const ConvProb& q = p.pr[i];
        if (q.b.C != 0 || q.a.C != CI || q.a.dtype != PC_F32 || !pc_planar(q.a) || q.a.mode != PC_SRC_DIRECT || q.a.oy || q.a.ox ||
            q.a.H != p.H || q.a.W != p.W || !al(q.a.ptr, q.a.bstride, q.a.cstride, q.a.rstride) || !fits32(q.a.bstride))
            return false;
        if (q.dot_w) {
            // (launch_conv: without an `out` the descriptor's out IS dot_out)
            if (CI != 8 || CO != 8 || ZC > 0 || !q.dot_out.ptr || q.dot_out.dtype != PC_F32 || !pc_planar(q.dot_out) ||
                !al(q.dot_out.ptr, q.dot_out.bstride, 4, q.dot_out.rstride) || !fits32(q.dot_out.bstride))
                return false;
        } else if (!q.out.ptr || q.out.dtype != PC_F32 || !pc_planar(q.out) || !al(q.out.ptr, q.out.bstride, q.out.cstride, q.out.rstride) ||
                   !fits32(q.out.bstride)) {
            return false;
        }
        if ((q.dot_out.ptr && !q.dot_w) || q.upt_w || q.act) return false;
        if (q.pool_out.ptr && (q.pool_out.dtype != PC_F32 || !pc_planar(q.pool_out) || !fits32(q.pool_out.bstride) ||
                               (reinterpret_cast<uintptr_t>(q.pool_out.ptr) & 7) != 0 || q.pool_out.rstride % 2 != 0 || q.pool_out.cstride % 2 != 0 ||
                               q.pool_out.bstride % 2 != 0 || p.W % 32 != 0 || p.H % 4 != 0))
            return false;
        if (ZC > 0 && (!q.z || !q.wz || !q.tb || !al(q.z, q.z_bs, q.z_cs, q.z_rs) || !fits32(q.z_bs))) return false;
    }
    return true;
}

template <int CI, int CO, int EPI, int ZC>
int launch_fwd_s3(ConvArgs& p, int nprob, hipStream_t stream) {
    using Cfg = FS3Cfg<CI, CO, ZC>;
    static int resident = 0;
    static pc_once_per_device once;
    if (once.need()) {
        const void* fn = reinterpret_cast<const void*>(&conv3x3_fwd_s3_kernel<CI, CO, EPI, ZC>);
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_B);
        if (e != hipSuccess) return (int)e;
        hipFuncAttributes fa;
        e = hipFuncGetAttributes(&fa, fn);
        if (e != hipSuccess) return (int)e;
        resident = pc_resident_workgroups(fa.numRegs, Cfg::LDS_B);
        once.mark();
        if (getenv("POPCORN_CONV_DBG"))
            fprintf(stderr, "conv3x3_fwd_s3<%d,%d,%d,z%d>: %d regs, %zu B LDS -> %d resident workgroups\n", CI, CO, EPI, ZC, fa.numRegs,
                    (size_t)Cfg::LDS_B, resident);
    }
    int max_grid = g_conv_max_grid > 0 ? g_conv_max_grid : resident / nprob;
    if (max_grid < 1) max_grid = 1;
    int grid = p.ntiles < max_grid ? p.ntiles : max_grid;
    const int rounds = (p.ntiles + grid - 1) / grid;
    grid = (p.ntiles + rounds - 1) / rounds;
    hipLaunchKernelGGL((conv3x3_fwd_s3_kernel<CI, CO, EPI, ZC>), dim3(grid, nprob), dim3(256), Cfg::LDS_B, stream, p);
    PC_CHECK_LAUNCH();
    return 0;
}
