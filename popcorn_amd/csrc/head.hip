// head.hip -- POPCORN's sparse occupancy head on fp32 MFMA, plus the small per-pixel kernels around it.
//
// Replaces (reference model/popcorn.py):
//   :80-85,161-164  head = Conv1x1(16,64) ReLU Conv1x1(64,64) ReLU Conv1x1(64,64) ReLU Conv1x1(64,2)[:,0]
//   :195-228        sparse_module_forward (gather by mask -> head -> index_put)   -> masked in place, no gather
//   :155,271-276    revert_padding (crop)                                          -> crop offsets in the loader
//   :170-190        scale = relu(out); popdensemap = scale * building; popcount = masked sum over the census region
//   :301,317-320    fusion_out_conv + sigmoid + crop (building score)
//   :361-377        get_sparsity_mask
//
// MLP mapping: pixels ride on N (16 pixels per wave-step), hidden units on M, so the accumulator (D) layout of
// layer l -- lane (n = pixel, lk): rows 16*mb + 4*lk + r -- is *already* the B-operand layout of layer l+1 when
// its K-steps are enumerated as (mb, r) -> hidden index 16*mb + 4*k + r.  The whole 16->64->64->64 chain therefore
// stays in registers: no LDS transposes, 144 MFMAs per 16 pixels at 100 % useful MACs.  Weights are staged once per
// workgroup in LDS as ready-made A fragments (lane-linear, conflict-free); the 64->1 last layer is 16 VALU FMAs + 2
// cross-lane adds.
#include "common.h"
#include <cstdlib>

namespace {

constexpr int HID = 64;
// LDS layout (floats)
constexpr int L_A1 = 0;                         // [4 mb][4 ks][64]        layer-1 A fragments  (W0: 64 x 16)
constexpr int L_A2 = L_A1 + 16 * 64;            // [4 mb2][16 ks][64]      layer-2 A fragments  (W2: 64 x 64)
constexpr int L_A3 = L_A2 + 64 * 64;            // [4 mb2][16 ks][64]      layer-3 A fragments  (W4: 64 x 64)
constexpr int L_B0 = L_A3 + 64 * 64;            // biases b0, b2, b4 (64 each), w6 row 0 (64), b6[0]
constexpr int L_B2 = L_B0 + 64;
constexpr int L_B4 = L_B2 + 64;
constexpr int L_W6 = L_B4 + 64;
constexpr int L_END = L_W6 + 64 + 4;

struct HeadArgs {
    pc_src feat;
    int py, px;
    const float* w0; const float* b0; const float* w2; const float* b2;
    const float* w4; const float* b4; const float* w6; const float* b6;
    const uint8_t* mask;
    const float* building;
    const float* admin;
    const int64_t* census;
    float* scale_map;
    float* popdense;
    float* partial;        // [B][nchunk][2]  {popcount partial, scale-sum partial}
    const void* wimage;    // the kernel's LDS weight image, assembled once per call by head_pack_kernel (workgroups copy it)
    int B, H, W;
    int groups, nchunk, groups_per_wave;
    pc_fastdiv div_w, div_groups;
    int bf;                // PC_PREC_BF16 (popcorn_hip.h): weights rounded to bf16 when staged, hidden activations after their
                           // ReLU, the gradients G3 / G2 / G1 before they are used as operands, g_feat when stored
};

// Fill the LDS weight image.  A fragments are packed so that one lane's operands for 4 consecutive K-steps are 16
// contiguous bytes: [m-block][k-group][lane][4] -> one ds_read_b128 feeds four MFMAs (a quarter of the LDS
// instructions of a per-K-step ds_read_b32; with one wave per SIMD in the backward kernel every LDS instruction
// sits on the MFMA issue path).
//   layer 1 (W0 64x16):  [mb][lane][j]            = W0[16*mb + i][4*j + k]
//   64x64 layers:        [mb2][mb][lane][r]       = W[16*mb2 + i][16*mb + 4*k + r]      (lane = k*16 + i)
__device__ __forceinline__ void head_stage_a(float* lds, const HeadArgs& p, int tid, int nt) {
    for (int e = tid; e < 16 * 64; e += nt) {
        const int j = e & 3, lane = (e >> 2) & 63, mb = e >> 8;
        const float t = p.w0[(16 * mb + (lane & 15)) * 16 + 4 * j + (lane >> 4)];
        lds[L_A1 + e] = t;
    }
    for (int e = tid; e < 64 * 64; e += nt) {
        const int r = e & 3, lane = (e >> 2) & 63, f = e >> 8, mb = f & 3, mb2 = f >> 2;
        const int col = 16 * mb + 4 * (lane >> 4) + r;
        const int row = 16 * mb2 + (lane & 15);
        const float t2 = p.w2[row * HID + col], t4 = p.w4[row * HID + col];
        lds[L_A2 + e] = t2;
        lds[L_A3 + e] = t4;
    }
}
__device__ __forceinline__ void head_stage_weights(float* lds, const HeadArgs& p, int tid, int nt) {
    head_stage_a(lds, p, tid, nt);
    for (int e = tid; e < 64; e += nt) {
        lds[L_B0 + e] = p.b0[e];
        lds[L_B2 + e] = p.b2[e];
        lds[L_B4 + e] = p.b4[e];
        lds[L_W6 + e] = p.w6[e];   // row 0 of the [2][64] last layer: only channel 0 is used (popcorn.py:162,164)
    }
    if (tid == 0) lds[L_W6 + 64] = p.b6[0];
}
// every workgroup starts by copying the ready-made image (16-byte pieces, coalesced) instead of gathering ~9,000 weights
// itself: 1024 workgroups doing that gather at once cost the bf16 forward kernel 10 of its 47 us
__device__ __forceinline__ void head_copy_image(void* lds, const void* img, int nbytes) {
    for (int e = threadIdx.x; e < nbytes / 16; e += blockDim.x) reinterpret_cast<uint4*>(lds)[e] = reinterpret_cast<const uint4*>(img)[e];
}

// One 64-wide layer: acc[mb2] = bias + W * h  (h in D layout of the previous layer), ReLU applied by the caller.
__device__ __forceinline__ void head_layer64(const float* lds, int a_off, int b_off, int lane, int lk,
                                             const f32x4 (&h)[4], f32x4 (&acc)[4]) {
#pragma unroll
    for (int mb2 = 0; mb2 < 4; ++mb2) acc[mb2] = *reinterpret_cast<const f32x4*>(&lds[b_off + 16 * mb2 + 4 * lk]);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        f32x4 a4[4];
#pragma unroll
        for (int mb2 = 0; mb2 < 4; ++mb2) a4[mb2] = *reinterpret_cast<const f32x4*>(&lds[a_off + ((mb2 * 4 + mb) * 64 + lane) * 4]);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mb2 = 0; mb2 < 4; ++mb2)
                acc[mb2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[mb2][r], h[mb][r], acc[mb2], 0, 0, 0);
    }
}

// first layer (16 -> 64): h[mb] = b0 + W0 * x
__device__ __forceinline__ void head_layer1(const float* lds, int a_off, int b_off, int lane, int lk, const float (&xv)[4],
                                            f32x4 (&h)[4]) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        h[mb] = *reinterpret_cast<const f32x4*>(&lds[b_off + 16 * mb + 4 * lk]);
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(&lds[a_off + (mb * 64 + lane) * 4]);
#pragma unroll
        for (int j = 0; j < 4; ++j) h[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], xv[j], h[mb], 0, 0, 0);
    }
}

__device__ __forceinline__ void relu4(f32x4 (&h)[4]) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) h[mb][r] = fmaxf(h[mb][r], 0.f);
}

// The same two contractions for the producer waves of head_bwd_pc_kernel, software-pipelined by hand: with ONE wave of that role
// per SIMD nothing hides an LDS round trip, and hipcc sinks the fragment reads to just in front of their first use (a
// lgkmcnt wait of ~150 cycles per 16 MFMAs) and re-serialises the 4 accumulator chains in places (a dependent MFMA issues
// 40 cycles after its predecessor, not 32).  Here the fragments of K-block mb+1 are in flight during the 16 MFMAs of block
// mb, and a scheduling barrier per K-step pins the 4-way accumulator interleave.
// hook(step), step = 0..15: called after the 4 MFMAs of every K-step, in front of its scheduling barrier -- the caller's slice of
// non-matrix work (a piece of a ring-slot write, of the next group's prefetch) that is to execute in the shadow of those MFMAs.
// pre: the fragments of K-block 0, read by the caller one phase earlier; nx (out): the block-0 fragments of the NEXT contraction
// (at nx_off, nx_stride K-blocks between its output blocks), read during this one's last block -- no contraction starts with an
// exposed LDS round trip.
template <class Hook>
__device__ __forceinline__ void head_mm64_pf(const float* lds, int off, int lane, const f32x4 (&in)[4], f32x4 (&acc)[4],
                                             const f32x4 (&pre)[4], int nx_off, int nx_stride, f32x4 (&nx)[4], Hook&& hook) {
    f32x4 a4[2][4];
#pragma unroll
    for (int mo = 0; mo < 4; ++mo) a4[0][mo] = pre[mo];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        if (mb < 3) {
#pragma unroll
            for (int mo = 0; mo < 4; ++mo)
                a4[(mb + 1) & 1][mo] = *reinterpret_cast<const f32x4*>(&lds[off + ((mo * 4 + mb + 1) * 64 + lane) * 4]);
        } else {
#pragma unroll
            for (int mo = 0; mo < 4; ++mo) nx[mo] = *reinterpret_cast<const f32x4*>(&lds[nx_off + (mo * nx_stride * 64 + lane) * 4]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int mo = 0; mo < 4; ++mo)
                acc[mo] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[mb & 1][mo][r], in[mb][r], acc[mo], 0, 0, 0);
            hook(mb * 4 + r);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
__device__ __forceinline__ void head_frag0(const float* lds, int off, int stride, int lane, f32x4 (&f)[4]) {
#pragma unroll
    for (int mo = 0; mo < 4; ++mo) f[mo] = *reinterpret_cast<const f32x4*>(&lds[off + (mo * stride * 64 + lane) * 4]);
}
// ReLU in ONE instruction (fmaxf on a value the compiler cannot prove canonical costs a canonicalising v_max x, x in front)
__device__ __forceinline__ void relu_block(f32x4& h) {
    // as a signed-integer maximum with 0 (the order of non-negative floats is the order of their bit patterns; anything with the sign
    // bit, -0 included, is a negative integer): fmaxf() -- and v_med3(x, 0, inf), which hipcc folds back into it -- costs a
    // canonicalising v_max x, x in front of the v_max.  +NaN passes through (the reference's ReLU propagates NaN as well).
#pragma unroll
    for (int r = 0; r < 4; ++r) h[r] = __int_as_float(max(__float_as_int(h[r]), 0));
}
__device__ __forceinline__ void mask_block(f32x4& g, const f32x4& h) {
#pragma unroll
    for (int r = 0; r < 4; ++r) g[r] = h[r] > 0.f ? g[r] : 0.f;
}
// first layer with the four output blocks interleaved (head_layer1 issues the 4 K-steps of a block back to back: dependent)
__device__ __forceinline__ void head_layer1_il(const float* lds, int a_off, int b_off, int lane, int lk, const float (&xv)[4],
                                               f32x4 (&h)[4]) {
    f32x4 a4[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        h[mb] = *reinterpret_cast<const f32x4*>(&lds[b_off + 16 * mb + 4 * lk]);
        a4[mb] = *reinterpret_cast<const f32x4*>(&lds[a_off + (mb * 64 + lane) * 4]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) h[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[mb][j], xv[j], h[mb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}
__device__ __forceinline__ void head_bias4(const float* lds, int b_off, int lk, f32x4 (&acc)[4]) {
#pragma unroll
    for (int mb2 = 0; mb2 < 4; ++mb2) acc[mb2] = *reinterpret_cast<const f32x4*>(&lds[b_off + 16 * mb2 + 4 * lk]);
}

// fp32 kernels; PC_PREC_BF16 has its own (head_fwd_bf16_kernel / head_bwd_bf16_coop4_kernel below, channels-last bf16 feature map)
__global__ __launch_bounds__(256) void head_fwd_kernel(const HeadArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    head_copy_image(lds, p.wimage, L_END * (int)sizeof(float));
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int b = blockIdx.y;
    const int HW = p.H * p.W;
    const float cid = p.census ? (float)p.census[b] : 0.f;
    float pc_sum = 0.f, sc_sum = 0.f;

    const int g_begin = (blockIdx.x * 4 + wave) * p.groups_per_wave;
    int g_end = g_begin + p.groups_per_wave;
    if (g_end > p.groups) g_end = p.groups;
    // The inputs of group g + 1 (mask byte, 4 feature values per lane) are loaded while group g runs its MFMA chain: the
    // gather is unconditional (clamped pixel) so that it does not sit behind the selection branch.
    // (the epilogue's building score and admin id come with the same prefetch: read where they are used they are one more exposed
    // round trip at the tail of every group)
    // NOTHING in the prefetch may use a loaded value (round 5: `sel = valid && mask[..] != 0` inside it made every group wait for its
    // mask byte -- s_waitcnt vmcnt(0), the preceding stores included -- before the remaining loads were even issued; the raw byte and
    // the raw feature values travel to the next iteration instead), and the optional inputs are read unconditionally (from the
    // building map when absent, value ignored) so that the number of loads in flight does not depend on a branch
    float bld_n = 0.f, adm_n = 0.f;
    unsigned msk_n = 1;
    const uint8_t* const mask_or_dummy = p.mask ? p.mask : reinterpret_cast<const uint8_t*>(p.building);
    const float* const admin_or_dummy = p.admin ? p.admin : p.building;
    auto fetch = [&](int g, bool& valid, float (&xv)[4]) {
        const int q = g * 16 + li;
        valid = q < HW && g < g_end;
        const int qc = valid ? q : 0;
        msk_n = mask_or_dummy[(int64_t)b * HW + qc];
        bld_n = p.building[(int64_t)b * HW + qc];
        adm_n = admin_or_dummy[(int64_t)b * HW + qc];
        const int y = (int)pc_div((uint32_t)qc, p.div_w), x = qc - y * p.W;
        const float* fp = p.feat.ptr + b * p.feat.bstride + (int64_t)(p.py + y) * p.feat.rstride + p.px + x;
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[j] = fp[(4 * j + lk) * p.feat.cstride];
    };
    bool valid_n = false;
    float xv_n[4] = {0.f, 0.f, 0.f, 0.f};
    if (g_begin < g_end) fetch(g_begin, valid_n, xv_n);
    f32x4 w6f[4];                  // the last layer's row and bias stay in registers (as in the backward's producer waves)
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) w6f[mb] = *reinterpret_cast<const f32x4*>(&lds[L_W6 + 16 * mb + 4 * lk]);
    const float b6v = lds[L_W6 + 64];
    for (int g = g_begin; g < g_end; ++g) {
        const int q = g * 16 + li;
        const bool valid = q < HW;
        const int64_t pix = (int64_t)b * HW + q;
        const bool sel = valid_n && (p.mask ? msk_n != 0 : true);
        const float bld = bld_n, adm = adm_n;
        float xv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[j] = valid_n ? xv_n[j] : 0.f;
        fetch(g + 1, valid_n, xv_n);
        float outv = 0.f;
        if (__any(sel)) {
            // software-pipelined chain (the helpers of the backward's producer waves): every contraction's first weight fragments are
            // read one phase ahead, the ReLU of its input blocks 1-3 happens inside its own first K-steps
            f32x4 h[4], acc[4], h3[4], fa[4], fb[4];
            head_frag0(lds, L_A2, 4, lane, fa);
            head_layer1_il(lds, L_A1, L_B0, lane, lk, xv, h);
            head_bias4(lds, L_B2, lk, acc);
            relu_block(h[0]);
            head_mm64_pf(lds, L_A2, lane, h, acc, fa, L_A3, 4, fb, [&](int st) { if (!(st & 3) && st < 12) relu_block(h[(st >> 2) + 1]); });
            head_bias4(lds, L_B4, lk, h3);
            relu_block(acc[0]);
            head_mm64_pf(lds, L_A3, lane, acc, h3, fb, L_W6, 0, fa, [&](int st) { if (!(st & 3) && st < 12) relu_block(acc[(st >> 2) + 1]); });
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) relu_block(h3[mb]);
            float s = 0.f;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) s = fmaf(w6f[mb][r], h3[mb][r], s);
            s = pc_xor16_sum(s);
            s = pc_xor32_sum(s);
            outv = sel ? s + b6v : 0.f;
        }
        if (valid && lk == 0) {
            // NaN-propagating ReLU like torch's (v_max_f32 returns the non-NaN operand): a NaN head output must reach the
            // loss, where the trainer's guard sees it (run_train.py:224-227).  The hidden ReLUs use the plain maximum.
            const float scale = outv > 0.f ? outv : (outv != outv ? outv : 0.f);
            const float pd = scale * bld;
            if (p.scale_map) p.scale_map[pix] = scale;
            p.popdense[pix] = pd;
            const bool region = p.admin ? (adm == cid) : true;
            pc_sum += region ? pd : 0.f;
            sc_sum += scale;
        }
    }
    // deterministic per-workgroup partial: lanes 0..15 of each wave hold the sums
    __syncthreads();
    float* red = lds;   // weights no longer needed
    if (lk == 0) { red[wave * 16 + li] = pc_sum; red[64 + wave * 16 + li] = sc_sum; }
    __syncthreads();
    if (tid < 2) {
        float t = 0.f;
        for (int i = 0; i < 64; ++i) t += red[tid * 64 + i];
        p.partial[((int64_t)b * p.nchunk + blockIdx.x) * 2 + tid] = t;
    }
}

// popcount[b] = sum of the sample's partials (fixed order).  stats (optional, double[2]) = {Nsel, sum of scale over
// the batch}: the two scalars the scale-regularisation term needs (utils/losses.py:74) -- and the only two numbers a
// data-parallel run has to exchange before the backward pass.
// The per-sample sums of the head forward's chunk partials, by ONE block of 256 threads, in ONE fixed order shared by every kernel that
// finishes them (the data-parallel reduce launch below and the single-process popcount + loss launch: a forced one-rank data-parallel
// step must equal the plain step bit for bit, tests/test_gpu_dp.py).  tpb threads per sample (a power of two, 256 / B at most): each
// sums every tpb-th partial, then a fixed tree inside the group.  (One thread per sample walked a census region's thousands of chunks
// serially: 50 us at B = 2 x 0.45 Mpx for 20 KB of partials.)  Returns this thread's share of sum(scale) as a double (non-zero in the
// threads that own a sample).
__device__ __forceinline__ double pc_popcount_sums_block(const float* partial, int nchunk, int B, float* popcount, float* tsum, float* usum) {
    double sc = 0.0;
    int tpb = 1;
    while (2 * tpb * B <= 256) tpb *= 2;
    for (int b0 = 0; b0 < B; b0 += 256 / tpb) {
        const int b = b0 + (int)threadIdx.x / tpb, j = (int)threadIdx.x & (tpb - 1);
        float t = 0.f, u = 0.f;
        if (b < B)
            for (int c = j; c < nchunk; c += tpb) {
                t += partial[((int64_t)b * nchunk + c) * 2];
                u += partial[((int64_t)b * nchunk + c) * 2 + 1];
            }
        tsum[threadIdx.x] = t; usum[threadIdx.x] = u;
        __syncthreads();
        for (int off = tpb >> 1; off > 0; off >>= 1) {
            if (j < off) { tsum[threadIdx.x] += tsum[threadIdx.x + off]; usum[threadIdx.x] += usum[threadIdx.x + off]; }
            __syncthreads();
        }
        if (j == 0 && b < B) {
            popcount[b] = tsum[threadIdx.x];
            sc += (double)usum[threadIdx.x];
        }
        __syncthreads();
    }
    return sc;
}

// popcount[b] = sum of the sample's partials (fixed order).  stats (optional, double[2]) = {Nsel, sum of scale over
// the batch}: the two scalars the scale-regularisation term needs (utils/losses.py:74) -- and the only two numbers a
// data-parallel run has to exchange before the backward pass.  ONE block of 256 threads.
__global__ __launch_bounds__(256) void head_popcount_reduce_kernel(const float* partial, float* popcount, int B, int nchunk, double* stats,
                                                                   const int32_t* nsel_counts, double dense_count) {
    __shared__ double red[256];
    __shared__ float tsum[256], usum[256];
    const double sc = pc_popcount_sums_block(partial, nchunk, B, popcount, tsum, usum);
    if (!stats) return;
    red[threadIdx.x] = sc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        stats[0] = nsel_counts ? (double)nsel_counts[0] : dense_count;
        stats[1] = red[0];
    }
}

// ---- head backward ---------------------------------------------------------------------------------------------
// Recomputes the forward chain in registers (cheaper than saving 3 x 64 hidden channels per pixel: 164 MB per
// layer at B=64), then walks back:  g3 = W6^T g_out . relu'   ->  g2 = W4^T g3 . relu'  ->  g1 = W2^T g2 . relu'
// -> g_x = W0^T g1, with the transposed-weight A fragments staged in LDS.  Because hidden units stay on M and
// pixels on N, every intermediate keeps the forward's register layout.  The weight gradients dW = g . h^T contract
// over pixels, which needs (hidden x pixel) matrices as A and B operands: each wave transposes g and h through a
// private LDS scratch (row stride 18 floats: conflict-free fragment reads) and accumulates dW in registers over a
// persistent loop of pixel groups.  One partial per workgroup, fixed-order second-stage reduction (deterministic).
constexpr int LB_A1 = 0;                         // fwd fragments as in the forward kernel
constexpr int LB_A2 = LB_A1 + 1024;
constexpr int LB_A3 = LB_A2 + 4096;
constexpr int LB_T3 = LB_A3 + 4096;              // [4 mi][16 ks][64]: lane(i,k) = W4[16mb+4k+r][16mi+i]
constexpr int LB_T2 = LB_T3 + 4096;              // same for W2
constexpr int LB_T1 = LB_T2 + 4096;              // [16 ks][64]:       lane(i=c,k) = W0[16mb+4k+r][c]
constexpr int LB_B0 = LB_T1 + 1024;
constexpr int LB_B2 = LB_B0 + 64;
constexpr int LB_B4 = LB_B2 + 64;
constexpr int LB_W6 = LB_B4 + 64;
constexpr int LB_SCR = LB_W6 + 64 + 4;           // per wave: Gm[64][18] + Hm[64][18]
constexpr int SCR_LD = 20;    // multiple of 4 (ds_read_b128) and 5 x 16 B: the 16 rows of a fragment hit 16 distinct slots
constexpr int SCR_WAVE = 2 * 64 * SCR_LD;
constexpr int LB_END = LB_SCR + 4 * SCR_WAVE;
// weight image of the fp32 backward kernels: forward fragments, transposed fragments, biases  (LB_A1 .. LB_W6; LB_SCR floats)
__device__ __forceinline__ void head_stage_weights_bwd(float* lds, const HeadArgs& p, int tid, int nt) {
    head_stage_a(lds, p, tid, nt);                      // LB_A1..LB_A3 coincide with L_A1..L_A3
    for (int e = tid; e < 64 * 64; e += nt) {
        const int r = e & 3, l = (e >> 2) & 63, f = e >> 8, mb = f & 3, mi = f >> 2;
        const int row = 16 * mb + 4 * (l >> 4) + r;                   // o
        const int col = 16 * mi + (l & 15);                           // i
        const float t4 = p.w4[row * HID + col], t2 = p.w2[row * HID + col];
        lds[LB_T3 + e] = t4;
        lds[LB_T2 + e] = t2;
    }
    for (int e = tid; e < 16 * 64; e += nt) {
        const int r = e & 3, l = (e >> 2) & 63, mb = e >> 8;
        const int row = 16 * mb + 4 * (l >> 4) + r;
        const float t0 = p.w0[row * 16 + (l & 15)];
        lds[LB_T1 + e] = t0;
    }
    for (int e = tid; e < 64; e += nt) {
        lds[LB_B0 + e] = p.b0[e];
        lds[LB_B2 + e] = p.b2[e];
        lds[LB_B4 + e] = p.b4[e];
        lds[LB_W6 + e] = p.w6[e];
    }
    if (tid == 0) lds[LB_W6 + 64] = p.b6[0];
}

// workgroup partial layout (common.h: the batched reduction of conv3x3_wgrad.hip can finish these partials too)
constexpr int PE_W4 = PC_PE_W4, PE_W2 = PC_PE_W2, PE_W0 = PC_PE_W0, PE_W6 = PC_PE_W6, PE_B0 = PC_PE_B0, PE_B2 = PC_PE_B2, PE_B4 = PC_PE_B4,
              PE_B6 = PC_PE_B6;
constexpr int PE_TOTAL = PC_PE_TOTAL;

struct HeadBwdArgs {
    HeadArgs f;
    const float* g_popcount;     // [B] or NULL
    const float* g_popdense;     // [B][H][W] or NULL
    const float* g_scale_map;    // [B][H][W] or NULL
    const float* g_scale_const;  // device scalar or NULL: added on every selected pixel
    pc_dst g_feat;
    pc_bn fbn[2];                // BN of the layers that produced feat channels 0-7 / 8-15 (fuse_feat_bn)
    int fuse_feat_bn;            // 1: g_feat *= (feat > 0) * bn_scale  (ReLU + frozen-BN backward of those layers)
    int dbg;                     // ablation (tools/ablate_head.py): 1 consumer idle, 2 no hand-off
    float* partial;              // [nwg][PE_TOTAL]
    int total_groups;
    int Hp, Wp;                  // extent of the padded feature / gradient maps
    int zero_in_kernel;          // producer / consumer kernel: g_feat is zeroed by the kernel itself (border + skipped groups)
};

__device__ __forceinline__ float lane_sum16(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    return v;
}

// dgrad through one 64x64 layer: out[mi] = sum_o W[o][16mi+i] * g[o]   (transposed fragments at t_off, float4-packed)
__device__ __forceinline__ void head_dgrad64(const float* lds, int t_off, int lane, const f32x4 (&g)[4], f32x4 (&out)[4]) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) out[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        f32x4 a4[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) a4[mi] = *reinterpret_cast<const f32x4*>(&lds[t_off + ((mi * 4 + mb) * 64 + lane) * 4]);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                out[mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[mi][r], g[mb][r], out[mi], 0, 0, 0);
    }
}

// scatter a D-layout (hidden x pixel) tile into the wave's LDS scratch as a [64][SCR_LD] matrix whose columns are
// permuted so that the 4 K-steps (pixels 4*ks + k, ks = 0..3) of a lane are contiguous: column(px) = (px & 3)*4 + (px >> 2)
__device__ __forceinline__ void head_store_mat(float* m, int li, int lk, const f32x4 (&v)[4]) {
    const int c = (li & 3) * 4 + (li >> 2);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) m[(16 * mb + 4 * lk + r) * SCR_LD + c] = v[mb][r];
}

// dW[o][i] += sum_px G[o][px] * Hm[i][px]   (64 x 64, 16 pixels = 4 k-steps; one ds_read_b128 per 16-row block)
__device__ __forceinline__ void head_wgrad64(const float* gm, const float* hm, int li, int lk, f32x4 (&dw)[4][4]) {
    f32x4 af[4], bf[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        af[q] = *reinterpret_cast<const f32x4*>(&gm[(16 * q + li) * SCR_LD + 4 * lk]);
        bf[q] = *reinterpret_cast<const f32x4*>(&hm[(16 * q + li) * SCR_LD + 4 * lk]);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
                dw[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mb][ks], bf[nb][ks], dw[mb][nb], 0, 0, 0);
}

__global__ __launch_bounds__(256, 1) void head_bwd_kernel(const HeadBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const HeadArgs& p = a.f;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    head_copy_image(lds, p.wimage, LB_SCR * (int)sizeof(float));     // forward + transposed fragments, biases (head_stage_weights_bwd)
    __syncthreads();

    float* gm = lds + LB_SCR + wave * SCR_WAVE;
    float* hm = gm + 64 * SCR_LD;
    const int HW = p.H * p.W;
    const float gsc = a.g_scale_const ? *a.g_scale_const : 0.f;
    float fscale[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        fscale[r] = 1.f;
        if (a.fuse_feat_bn) {
            const int c = 4 * lk + r;
            float sh;
            pc_bn_fold(a.fbn[c >> 3], c & 7, fscale[r], sh);
        }
    }

    f32x4 dW4[4][4], dW2[4][4], dW0[4], dw6[4], db0[4], db2[4], db4[4];
    float db6 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        dW0[i] = dw6[i] = db0[i] = db2[i] = db4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) dW4[i][j] = dW2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    for (int gg = blockIdx.x * 4 + wave; gg < a.total_groups; gg += gridDim.x * 4) {
        const int b = (int)pc_div((uint32_t)gg, p.div_groups), g = gg - b * p.groups;
        const int q = g * 16 + li;
        const bool valid = q < HW;
        const int64_t pix = (int64_t)b * HW + q;
        const bool sel = valid && (p.mask ? p.mask[pix] != 0 : true);
        if (!__any(sel)) continue;
        const int y = valid ? (int)pc_div((uint32_t)q, p.div_w) : 0, x = valid ? q - y * p.W : 0;
        const float* fp = p.feat.ptr + b * p.feat.bstride + (int64_t)(p.py + y) * p.feat.rstride + p.px + x;
        float xv[4], fvv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            xv[j] = valid ? fp[(4 * j + lk) * p.feat.cstride] : 0.f;
            fvv[j] = (valid && a.fuse_feat_bn) ? fp[(4 * lk + j) * p.feat.cstride] : 1.f;   // issued with xv: not on the tail
        }
        // upstream gradient of relu(out) at this pixel
        float gup = 0.f;
        if (sel) {
            const float bld = p.building[pix];
            const bool region = p.admin ? (p.admin[pix] == (float)p.census[b]) : true;
            gup = gsc;
            if (a.g_popcount && region) gup += a.g_popcount[b] * bld;
            if (a.g_popdense) gup += a.g_popdense[pix] * bld;
            if (a.g_scale_map) gup += a.g_scale_map[pix];
        }
        // ---- forward recompute
        f32x4 h1[4], h2[4], h3[4];
        head_layer1(lds, LB_A1, LB_B0, lane, lk, xv, h1);
        relu4(h1);
        head_layer64(lds, LB_A2, LB_B2, lane, lk, h1, h2);
        relu4(h2);
        head_layer64(lds, LB_A3, LB_B4, lane, lk, h2, h3);
        relu4(h3);
        float s = 0.f;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(&lds[LB_W6 + 16 * mb + 4 * lk]);
#pragma unroll
            for (int r = 0; r < 4; ++r) s = fmaf(w[r], h3[mb][r], s);
        }
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        const float outv = s + lds[LB_W6 + 64];
        const float gout = (sel && outv > 0.f) ? gup : 0.f;
        if (!__any(gout != 0.f)) continue;

        // ---- layer 4 (64 -> 1)
        f32x4 g3[4], g2[4], g1[4];
        if (lk == 0) db6 += gout;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(&lds[LB_W6 + 16 * mb + 4 * lk]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dw6[mb][r] = fmaf(gout, h3[mb][r], dw6[mb][r]);
                g3[mb][r] = h3[mb][r] > 0.f ? w[r] * gout : 0.f;
                db4[mb][r] += g3[mb][r];
            }
        }
        // ---- layer 3 (W4)
        __builtin_amdgcn_wave_barrier();
        head_store_mat(gm, li, lk, g3);
        head_store_mat(hm, li, lk, h2);
        __builtin_amdgcn_wave_barrier();
        head_wgrad64(gm, hm, li, lk, dW4);
        head_dgrad64(lds, LB_T3, lane, g3, g2);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                g2[mb][r] = h2[mb][r] > 0.f ? g2[mb][r] : 0.f;
                db2[mb][r] += g2[mb][r];
            }
        // ---- layer 2 (W2)
        __builtin_amdgcn_wave_barrier();
        head_store_mat(gm, li, lk, g2);
        head_store_mat(hm, li, lk, h1);
        __builtin_amdgcn_wave_barrier();
        head_wgrad64(gm, hm, li, lk, dW2);
        head_dgrad64(lds, LB_T2, lane, g2, g1);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                g1[mb][r] = h1[mb][r] > 0.f ? g1[mb][r] : 0.f;
                db0[mb][r] += g1[mb][r];
            }
        // ---- layer 1 (W0: 64 x 16)
        __builtin_amdgcn_wave_barrier();
        head_store_mat(gm, li, lk, g1);
        {
            const int c = (li & 3) * 4 + (li >> 2);
#pragma unroll
            for (int j = 0; j < 4; ++j) hm[(4 * j + lk) * SCR_LD + c] = xv[j];
        }
        __builtin_amdgcn_wave_barrier();
        {
            const f32x4 bf = *reinterpret_cast<const f32x4*>(&hm[li * SCR_LD + 4 * lk]);
            f32x4 af[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) af[mb] = *reinterpret_cast<const f32x4*>(&gm[(16 * mb + li) * SCR_LD + 4 * lk]);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
                    dW0[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mb][ks], bf[ks], dW0[mb], 0, 0, 0);
        }
        f32x4 gx = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const f32x4 t4 = *reinterpret_cast<const f32x4*>(&lds[LB_T1 + (mb * 64 + lane) * 4]);
#pragma unroll
            for (int r = 0; r < 4; ++r) gx = __builtin_amdgcn_mfma_f32_16x16x4f32(t4[r], g1[mb][r], gx, 0, 0, 0);
        }
        if (valid) {
            float* op = a.g_feat.ptr + b * a.g_feat.bstride + (int64_t)(p.py + y) * a.g_feat.rstride + p.px + x;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float o = gx[r];
                // xv[r'] holds feat channel 4*j + lk; channel 4*lk + r is held by lane group lk' = r at j = lk
                if (a.fuse_feat_bn) {
                    const float fv = fvv[r];
                    o = fv > 0.f ? o * fscale[r] : 0.f;
                }
                op[(4 * lk + r) * a.g_feat.cstride] = o;
            }
        }
    }

    // ---- per-lane vectors: reduce over the 16 pixel lanes
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dw6[mb][r] = lane_sum16(dw6[mb][r]);
            db0[mb][r] = lane_sum16(db0[mb][r]);
            db2[mb][r] = lane_sum16(db2[mb][r]);
            db4[mb][r] = lane_sum16(db4[mb][r]);
        }
    db6 = lane_sum16(db6);

    // ---- cross-wave reduction (fixed order) -> one partial per workgroup
    float* part = a.partial + (int64_t)blockIdx.x * PE_TOTAL;
    __syncthreads();
#pragma unroll
    for (int stage = 0; stage < 2; ++stage) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
                *reinterpret_cast<f32x4*>(&lds[wave * 4096 + ((mb * 4 + nb) * 64 + lane) * 4]) = stage == 0 ? dW4[mb][nb] : dW2[mb][nb];
        __syncthreads();
        for (int e = tid; e < 4096; e += 256)
            part[(stage == 0 ? PE_W4 : PE_W2) + e] = ((lds[e] + lds[4096 + e]) + lds[8192 + e]) + lds[12288 + e];
        __syncthreads();
    }
    // small stuff: per wave [dW0 1024][dw6 64][db0 64][db2 64][db4 64][db6 1] -> stride 1344
    {
        float* w = lds + wave * 1344;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) *reinterpret_cast<f32x4*>(&w[(mb * 64 + lane) * 4]) = dW0[mb];
        if (li == 0) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int h = 16 * mb + 4 * lk + r;
                    w[1024 + h] = dw6[mb][r];
                    w[1088 + h] = db0[mb][r];
                    w[1152 + h] = db2[mb][r];
                    w[1216 + h] = db4[mb][r];
                }
            if (lk == 0) w[1280] = db6;
        }
        __syncthreads();
        for (int e = tid; e < 1281; e += 256) {
            const float t = ((lds[e] + lds[1344 + e]) + lds[2688 + e]) + lds[4032 + e];
            part[PE_W0 + e] = t;     // PE_W0.. contiguous: W0(1024) W6(64) B0 B2 B4 (64 each) B6(1)
        }
    }
}

// Phase profile of the producer waves (debug builds only: make CXXFLAGS+=-DPOPCORN_HEAD_PROF; tools/ablate_head.py prints it)
#ifdef POPCORN_HEAD_PROF
__device__ long long g_head_prof[2048 * 16];
#define HP_DECL long long hp_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long hp_t = 0, hp_a = 0; int hp_last = -1
#define HP_ACQ0 hp_a = (long long)__builtin_readcyclecounter()
#define HP_ACQ1 hp_acc[9] += (long long)__builtin_readcyclecounter() - hp_a
#define HP_MARK(k) do { __builtin_amdgcn_sched_barrier(0); const long long t_ = (long long)__builtin_readcyclecounter(); \
                        if (hp_last >= 0) hp_acc[hp_last] += t_ - hp_t; hp_t = t_; hp_last = (k); } while (0)
#define HP_DUMP do { HP_MARK(8); if (wave == 0 && lane == 0) for (int k_ = 0; k_ < 10; ++k_) g_head_prof[blockIdx.x * 16 + k_] = hp_acc[k_]; } while (0)
// the bf16 kernel's stamps: seven scalar accumulators with compile-time bucket indices (hp_acc[hp_last] above is a dynamically indexed array =
// scratch memory, and every scratch access waits for vmcnt(0): the stamps then absorb the latency of the prefetch loads in flight)
#define HQ_DECL long long hq0 = 0, hq1 = 0, hq2 = 0, hq3 = 0, hq4 = 0, hq5 = 0, hq6 = 0, hq_t = 0, hq_n = 0
#define HQ_NOW(v) do { __builtin_amdgcn_sched_barrier(0); v = (long long)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define HQ_CLOSE(acc) do { HQ_NOW(hq_n); acc += hq_n - hq_t; hq_t = hq_n; } while (0)
#define HQ_DUMP do { if (lane == 0) { long long* d_ = &g_head_prof[(blockIdx.x * 4 + wave) * 16]; d_[0] = hq0; d_[1] = hq1; d_[2] = hq2; d_[3] = hq3; \
                                      d_[4] = hq4; d_[5] = hq5; d_[6] = hq6; } } while (0)
#define HR_DECL long long hr[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, hr_t = 0, hr_n = 0
#define HR_CLOSE(k) do { HQ_NOW(hr_n); hr[k] += hr_n - hr_t; hr_t = hr_n; } while (0)
#define HR_DUMP do { if (lane == 0) for (int k_ = 0; k_ < 12; ++k_) g_head_prof[(blockIdx.x * 4 + wave) * 16 + k_] = hr[k_]; } while (0)
#else
#define HR_DECL
#define HR_CLOSE(k)
#define HR_DUMP
#define HQ_DECL
#define HQ_NOW(v)
#define HQ_CLOSE(acc)
#define HQ_DUMP
#define HP_DECL
#define HP_ACQ0
#define HP_ACQ1
#define HP_MARK(k)
#define HP_DUMP
#endif

// ---- bf16 mode (PC_PREC_BF16): the head on v_mfma_f32_16x16x32_bf16 / 16x16x16_bf16 ---------------------------------------
// The 64-wide contractions take 2 instructions of K = 32 instead of 16 fp32 ones, so the matrix pipe stops being the limit.
// A layer computed as D = mfma(A = W fragment, B = packed activations) leaves lane = pixel, registers = hidden units -- the
// operand layout of the NEXT layer, so the forward and backward chains stay in registers; the weight gradients (contraction
// over pixels) take their operands through an LDS exchange and transposing reads (head_bwd_bf16_coop4_kernel).
// K-slot conventions (identical for both operands, so the hardware's internal K order is irrelevant):
//   64-wide contraction, instruction t of 2:  slot (lk, j) -> hidden unit 16*(2t + (j >> 2)) + 4*lk + (j & 3)
//       (= D-layout registers h[2t][0..3], h[2t+1][0..3] of the lane, packed in order)
//   16-wide feature contraction (layer 1):    slot (lk, j) -> feature channel 4*lk + j       (one 8-byte load per lane: the four
//       channels arrive packed, as the operand -- round 5; rounds 2-4 gathered channel 4*j + lk with four 2-byte loads and converted them
//       to fp32 inside the prefetch, which made every prefetch wait for its own loads: profiles/r5_head_bwd_bf16_phases.json)
//   32-pixel contraction (weight gradients):  slot (lk, j) -> pixel 8*lk + j of a pair of 16-pixel groups
typedef __bf16 hbf16x8 __attribute__((ext_vector_type(8)));
typedef short hs16x4 __attribute__((ext_vector_type(4)));
typedef unsigned hu32x4 __attribute__((ext_vector_type(4)));
// LDS image (bytes): ready-made fragments, lane-linear
constexpr int HB_A1 = 0;                          // [4 mb][64 lanes][4 bf16]   W0[16mb+i][4lk+j]
constexpr int HB_A2 = HB_A1 + 4 * 64 * 8;         // [4 mb2][2 t][64][8 bf16]   W2[16mb2+i][unit(t,lk,j)]
constexpr int HB_A3 = HB_A2 + 8 * 64 * 16;        // same for W4
constexpr int HB_T3 = HB_A3 + 8 * 64 * 16;        // [4 mi][2 t][64][8]         W4[unit(t,lk,j)][16mi+i]
constexpr int HB_T2 = HB_T3 + 8 * 64 * 16;        // same for W2
constexpr int HB_T1 = HB_T2 + 8 * 64 * 16;        // [2 t][64][8]               W0[unit(t,lk,j)][c = i]
constexpr int HB_F32 = HB_T1 + 2 * 64 * 16;       // floats: b0[64] b2[64] b4[64] w6[64] (rounded) b6
constexpr int HB_END = HB_F32 + (4 * 64 + 4) * 4;

__device__ __forceinline__ int hb_unit(int t, int lk, int j) { return 16 * (2 * t + (j >> 2)) + 4 * lk + (j & 3); }

__device__ __forceinline__ void head_stage_weights_bf16(unsigned char* lds, const HeadArgs& p, bool backward, int tid, int nt) {
    unsigned short* h = reinterpret_cast<unsigned short*>(lds);
    auto bits = [](float x) { return (unsigned short)(__float_as_uint(pc_bf16r(x)) >> 16); };
    for (int e = tid; e < 4 * 64 * 4; e += nt) {
        const int j = e & 3, lane = (e >> 2) & 63, mb = e >> 8;
        h[HB_A1 / 2 + e] = bits(p.w0[(16 * mb + (lane & 15)) * 16 + 4 * (lane >> 4) + j]);
    }
    for (int e = tid; e < 8 * 64 * 8; e += nt) {
        const int j = e & 7, lane = (e >> 3) & 63, f = e >> 9, t = f & 1, mb2 = f >> 1;
        const int u = hb_unit(t, lane >> 4, j), i = lane & 15;
        h[HB_A2 / 2 + e] = bits(p.w2[(16 * mb2 + i) * HID + u]);
        h[HB_A3 / 2 + e] = bits(p.w4[(16 * mb2 + i) * HID + u]);
        if (backward) {
            h[HB_T3 / 2 + e] = bits(p.w4[u * HID + 16 * mb2 + i]);
            h[HB_T2 / 2 + e] = bits(p.w2[u * HID + 16 * mb2 + i]);
        }
    }
    if (backward)
        for (int e = tid; e < 2 * 64 * 8; e += nt) {
            const int j = e & 7, lane = (e >> 3) & 63, t = e >> 9;
            h[HB_T1 / 2 + e] = bits(p.w0[hb_unit(t, lane >> 4, j) * 16 + (lane & 15)]);
        }
    float* f = reinterpret_cast<float*>(lds + HB_F32);
    for (int e = tid; e < 64; e += nt) {
        f[e] = p.b0[e];
        f[64 + e] = p.b2[e];
        f[128 + e] = p.b4[e];
        f[192 + e] = pc_bf16r(p.w6[e]);
    }
    if (tid == 0) f[256] = p.b6[0];
}

__device__ __forceinline__ hbf16x8 hb_pack8(const f32x4& a, const f32x4& b) {
    const hu32x4 q = {pc_pack_bf16(a[0], a[1]), pc_pack_bf16(a[2], a[3]), pc_pack_bf16(b[0], b[1]), pc_pack_bf16(b[2], b[3])};
    return __builtin_bit_cast(hbf16x8, q);
}
__device__ __forceinline__ hs16x4 hb_pack4(float a, float b, float c, float d) {
    const uint2 q = make_uint2(pc_pack_bf16(a, b), pc_pack_bf16(c, d));
    return __builtin_bit_cast(hs16x4, q);
}
// per 16-bit half: 0xffff where the bf16 is non-zero (packed min with 1, then 0 - x), and the AND of a packed vector with such a mask
// (two VOP3P instructions per dword; written as asm because hipcc lowers the vector form to 16-bit compares + selects + perms)
__device__ __forceinline__ unsigned hb_nz32(unsigned x) {
    unsigned nz, r;
    asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(nz) : "v"(x));
    asm("v_pk_sub_u16 %0, 0, %1 op_sel_hi:[0,1]" : "=v"(r) : "v"(nz));
    return r;
}
__device__ __forceinline__ hu32x4 hb_nzmask(const hbf16x8& v) {
    const hu32x4 q = __builtin_bit_cast(hu32x4, v);
    return hu32x4{hb_nz32(q[0]), hb_nz32(q[1]), hb_nz32(q[2]), hb_nz32(q[3])};
}
__device__ __forceinline__ hbf16x8 hb_and(const hbf16x8& v, const hu32x4& m) {
    return __builtin_bit_cast(hbf16x8, __builtin_bit_cast(hu32x4, v) & m);
}
// `lane` may carry an opaque zero (see the group loops): the fragment reads must stay INSIDE the loop -- hoisted, the 36 KB
// of loop-invariant weight fragments would occupy ~150 registers per lane and spill the accumulators
__device__ __forceinline__ hbf16x8 hb_frag8(const unsigned char* lds, int off, int blk, int t, int lane) {
    return __builtin_bit_cast(hbf16x8, *reinterpret_cast<const hu32x4*>(lds + off + (blk * 2 + t) * 1024 + lane * 16));
}
__device__ __forceinline__ hs16x4 hb_frag4(const unsigned char* lds, int off, int blk, int lane) {
    return __builtin_bit_cast(hs16x4, *reinterpret_cast<const uint2*>(lds + off + blk * 512 + lane * 8));
}

// one 64 -> 64 layer.  hb[t]: packed input (lane = pixel).  o1[mb2]: D = W . h (lane = pixel, regs = hidden 16*mb2 + 4*lk + r),
// initialised with the bias.
__device__ __forceinline__ void hb_layer64(const unsigned char* lds, int a_off, const float* bias, int lane, int lk,
                                           const hbf16x8 (&hb)[2], f32x4 (&o1)[4]) {
#pragma unroll
    for (int mb2 = 0; mb2 < 4; ++mb2) {
        o1[mb2] = *reinterpret_cast<const f32x4*>(&bias[16 * mb2 + 4 * lk]);
#pragma unroll
        for (int t = 0; t < 2; ++t)
            o1[mb2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hb_frag8(lds, a_off, mb2, t, lane), hb[t], o1[mb2], 0, 0, 0);
    }
}

// ReLU of an MFMA result in ONE instruction: the signed-integer maximum with 0 (see relu_block; fmaxf() -- and v_med3(x, 0, inf),
// which hipcc folds back into it -- costs a canonicalising v_max x, x in front).  (Not inline asm: the compiler does not place
// the MFMA-result hazard wait in front of an asm statement -- measured: non-deterministic losses.)
__device__ __forceinline__ float hb_relu(float x) { return __int_as_float(max(__float_as_int(x), 0)); }
__device__ __forceinline__ void hb_relu4(f32x4 (&h)[4]) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) h[mb][r] = hb_relu(h[mb][r]);
}
// ReLU + round to bf16, kept as fp32 values: one pack per PAIR and two unpacks (instead of a pack and a shift per element)
__device__ __forceinline__ void hb_relu_round(f32x4 (&h)[4]) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
            const unsigned q = pc_pack_bf16(hb_relu(h[mb][r]), hb_relu(h[mb][r + 1]));
            h[mb][r] = __uint_as_float(q << 16);
            h[mb][r + 1] = __uint_as_float(q & 0xffff0000u);
        }
}

__global__ __launch_bounds__(256) void head_fwd_bf16_kernel(const HeadArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    head_copy_image(ldsb, p.wimage, HB_END);
    __syncthreads();
    const float* lf = reinterpret_cast<const float*>(ldsb + HB_F32);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    f32x4 w6f[4];                  // the last layer's (rounded) row and bias stay in registers
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) w6f[mb] = *reinterpret_cast<const f32x4*>(&lf[192 + 16 * mb + 4 * lk]);
    const float b6v = lf[256];
    const int b = blockIdx.y;
    const int HW = p.H * p.W;
    const float cid = p.census ? (float)p.census[b] : 0.f;
    float pc_sum = 0.f, sc_sum = 0.f;
    const int g_begin = (blockIdx.x * 4 + wave) * p.groups_per_wave;
    int g_end = g_begin + p.groups_per_wave;
    if (g_end > p.groups) g_end = p.groups;
    // (the epilogue's building score and admin id come with the same prefetch: read where they are used they are one more exposed
    // round trip at the tail of every group)
    float bld_n = 0.f, adm_n = 0.f;
    // (nothing in the prefetch may USE a loaded value -- a conversion, a select on it: the compiler then waits for the load right where
    // it was issued and the prefetch hides nothing; the raw bits travel to the next iteration)
    // the optional inputs are read unconditionally (from the building map when absent, value ignored): with a load count that depends
    // on a branch the compiler's s_waitcnt for the PREVIOUS group's values must assume the fewest new loads in flight and ends up
    // waiting for the first of the new ones
    unsigned msk_n = 1;
    const uint8_t* const mask_or_dummy = p.mask ? p.mask : reinterpret_cast<const uint8_t*>(p.building);
    const float* const admin_or_dummy = p.admin ? p.admin : p.building;
    auto fetch = [&](int g, bool& valid, uint2& xq) {
        const int q = g * 16 + li;
        valid = q < HW && g < g_end;
        const int qc = valid ? q : 0;
        msk_n = mask_or_dummy[(int64_t)b * HW + qc];
        bld_n = p.building[(int64_t)b * HW + qc];
        adm_n = admin_or_dummy[(int64_t)b * HW + qc];
        const int y = (int)pc_div((uint32_t)qc, p.div_w), x = qc - y * p.W;
        // channels-last feature map: the 16 channels of the pixel are contiguous; this lane's K-slots are channels 4*lk .. +3
        const pc_bf16_t* fp = reinterpret_cast<const pc_bf16_t*>(p.feat.ptr) + b * p.feat.bstride + (int64_t)(p.py + y) * p.feat.rstride +
                              (int64_t)(p.px + x) * p.feat.xstride;
        xq = *reinterpret_cast<const uint2*>(fp + 4 * lk);
    };
    bool valid_n = false;
    uint2 xq_n = make_uint2(0u, 0u);
    if (g_begin < g_end) fetch(g_begin, valid_n, xq_n);
    for (int g = g_begin; g < g_end; ++g) {
        const int q = g * 16 + li;
        const bool valid = q < HW;
        const int64_t pix = (int64_t)b * HW + q;
        const bool sel = valid_n && (p.mask ? msk_n != 0 : true);
        const float bld = bld_n, adm = adm_n;
        const hs16x4 xb = __builtin_bit_cast(hs16x4, valid_n ? xq_n : make_uint2(0u, 0u));
        fetch(g + 1, valid_n, xq_n);
        float outv = 0.f;
        if (__any(sel)) {
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));          // opaque: keeps the fragment reads in the loop
            f32x4 h[4], acc[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                h[mb] = *reinterpret_cast<const f32x4*>(&lf[16 * mb + 4 * lk]);
                h[mb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(hb_frag4(ldsb, HB_A1, mb, lane_o), xb, h[mb], 0, 0, 0);
            }
            hb_relu4(h);                                   // the pack rounds to bf16
            hbf16x8 hb[2] = {hb_pack8(h[0], h[1]), hb_pack8(h[2], h[3])};
            hb_layer64(ldsb, HB_A2, lf + 64, lane_o, lk, hb, acc);
            hb_relu4(acc);
            hb[0] = hb_pack8(acc[0], acc[1]); hb[1] = hb_pack8(acc[2], acc[3]);
            hb_layer64(ldsb, HB_A3, lf + 128, lane_o, lk, hb, h);
            hb_relu_round(h);
            float s = 0.f;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) s = fmaf(w6f[mb][r], h[mb][r], s);
            s = pc_xor16_sum(s);
            s = pc_xor32_sum(s);
            outv = sel ? s + b6v : 0.f;
        }
        if (valid && lk == 0) {
            const float scale = outv > 0.f ? outv : (outv != outv ? outv : 0.f);
            const float pd = scale * bld;
            if (p.scale_map) p.scale_map[pix] = scale;
            p.popdense[pix] = pd;
            const bool region = p.admin ? (adm == cid) : true;
            pc_sum += region ? pd : 0.f;
            sc_sum += scale;
        }
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(ldsb);
    if (lk == 0) { red[wave * 16 + li] = pc_sum; red[64 + wave * 16 + li] = sc_sum; }
    __syncthreads();
    if (tid < 2) {
        float t = 0.f;
        for (int i = 0; i < 64; ++i) t += red[tid * 64 + i];
        p.partial[((int64_t)b * p.nchunk + blockIdx.x) * 2 + tid] = t;
    }
}


// transposing LDS read ([4 rows][16 columns] of bf16 -> lane c receives column c) and the pairing of two of them into one operand
__device__ __forceinline__ hs16x4 hc_tr(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) hs16x4*)(p));
}
__device__ __forceinline__ hbf16x8 hc_pair(hs16x4 a, hs16x4 b) {
    return __builtin_bit_cast(hbf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

// ---- fp32 results on the bf16 matrix pipe: 3-way operand splits (round 5) -----------------------------------------------------------
// An fp32 number is EXACTLY the sum of three bf16 numbers (a1 = rn(a), a2 = rn(a - a1), a3 = rn(a - a1 - a2): 8 + 8 + 8 mantissa bits,
// same exponent range; each difference is exact in fp32).  A product a * b is then the sum of nine bf16 x bf16 products, each exact in
// the fp32 accumulator of v_mfma_f32_16x16x32_bf16; the three smallest (a2 b3, a3 b2, a3 b3: <= 2^-23 of the product together) are
// dropped, the other six are accumulated smallest first.  Per product the error is of the size of ONE fp32 rounding -- the class of
// error the fp32 MFMA's own fma chain makes at every step -- and the contraction runs at 6 x 16 cycles per 32 K-slots instead of
// 8 x 32 on v_mfma_f32_16x16x4_f32: 2.67x the matrix rate for the price of 11 VALU instructions per pair of activations (weights are
// split once per step, by the pack launch).  The head is the one kernel class of the fp32 step that is bound by the matrix pipe
// (forward 0.72 of the fp32 MFMA peak in issued work), which is where this pays; the conv class is half memory-bound and gained 10 %
// in round 2's prototype.  POPCORN_HEAD_SPLIT=0 selects the fp32-MFMA kernel (A/B and tests).
//   LDS image: three planes (split index) of the bf16 forward fragments A1 | A2 | A3 (offsets HB_A1 / HB_A2 / HB_A3 inside a plane),
//   then fp32 b0[64] b2[64] b4[64] w6[64] b6.
constexpr int HS_PLANE = HB_T3;
constexpr int HS_F32 = 3 * HS_PLANE;
constexpr int HS_END = HS_F32 + (4 * 64 + 4) * 4;

__device__ __forceinline__ void hs_split3(float x, unsigned short (&o)[3]) {
    const float a1 = pc_bf16r(x), r1 = x - a1, a2 = pc_bf16r(r1), r2 = r1 - a2, a3 = pc_bf16r(r2);
    o[0] = (unsigned short)(__float_as_uint(a1) >> 16);
    o[1] = (unsigned short)(__float_as_uint(a2) >> 16);
    o[2] = (unsigned short)(__float_as_uint(a3) >> 16);
}

__device__ __forceinline__ void head_stage_weights_split(unsigned char* img, const HeadArgs& p, int tid, int nt) {
    unsigned short* h = reinterpret_cast<unsigned short*>(img);
    unsigned short o[3];
    for (int e = tid; e < 4 * 64 * 4; e += nt) {
        const int j = e & 3, lane = (e >> 2) & 63, mb = e >> 8;
        hs_split3(p.w0[(16 * mb + (lane & 15)) * 16 + 4 * (lane >> 4) + j], o);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) h[(pl * HS_PLANE + HB_A1) / 2 + e] = o[pl];
    }
    for (int e = tid; e < 8 * 64 * 8; e += nt) {
        const int j = e & 7, lane = (e >> 3) & 63, f = e >> 9, t = f & 1, mb2 = f >> 1;
        const int u = hb_unit(t, lane >> 4, j), i = lane & 15;
        hs_split3(p.w2[(16 * mb2 + i) * HID + u], o);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) h[(pl * HS_PLANE + HB_A2) / 2 + e] = o[pl];
        hs_split3(p.w4[(16 * mb2 + i) * HID + u], o);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) h[(pl * HS_PLANE + HB_A3) / 2 + e] = o[pl];
    }
    float* f = reinterpret_cast<float*>(img + HS_F32);
    for (int e = tid; e < 64; e += nt) {
        f[e] = p.b0[e];
        f[64 + e] = p.b2[e];
        f[128 + e] = p.b4[e];
        f[192 + e] = p.w6[e];
    }
    if (tid == 0) f[256] = p.b6[0];
}

// one pair of fp32 values -> the three packed bf16 pairs of its split
__device__ __forceinline__ void hs_split_pair(float x0, float x1, unsigned& q1, unsigned& q2, unsigned& q3) {
    q1 = pc_pack_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(q1 << 16), r1 = x1 - __uint_as_float(q1 & 0xffff0000u);
    q2 = pc_pack_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(q2 << 16), s1 = r1 - __uint_as_float(q2 & 0xffff0000u);
    q3 = pc_pack_bf16(s0, s1);
}
// the 16 values of a lane (D layout: h[mb][r] = hidden 16*mb + 4*lk + r) -> packed operands o[split][t] of the next 64-wide contraction
__device__ __forceinline__ void hs_split_pack(const f32x4 (&h)[4], hbf16x8 (&o)[3][2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        hu32x4 q[3];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const f32x4& v = h[2 * t + (d >> 1)];
            unsigned q1, q2, q3;
            hs_split_pair(v[2 * (d & 1)], v[2 * (d & 1) + 1], q1, q2, q3);
            q[0][d] = q1; q[1][d] = q2; q[2][d] = q3;
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) o[pl][t] = __builtin_bit_cast(hbf16x8, q[pl]);
    }
}
// 64 -> 64 layer on split operands: o1[mb2] = bias + W . h, six partial products per K-step, smallest first; the four accumulators
// of the layer take turns (independent MFMA chains)
__device__ __forceinline__ void hs_layer64(const unsigned char* lds, int a_off, const float* bias, int lane, int lk,
                                           const hbf16x8 (&hb)[3][2], f32x4 (&o1)[4]) {
#pragma unroll
    for (int mb2 = 0; mb2 < 4; ++mb2) o1[mb2] = *reinterpret_cast<const f32x4*>(&bias[16 * mb2 + 4 * lk]);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int pl = 2; pl >= 0; --pl) {                 // weight split index; activation split indices 0 .. 2 - pl
            hbf16x8 fr[4];
#pragma unroll
            for (int mb2 = 0; mb2 < 4; ++mb2) fr[mb2] = hb_frag8(lds, pl * HS_PLANE + a_off, mb2, t, lane);
#pragma unroll
            for (int q = 2 - pl; q >= 0; --q)
#pragma unroll
                for (int mb2 = 0; mb2 < 4; ++mb2)
                    o1[mb2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[mb2], hb[q][t], o1[mb2], 0, 0, 0);
        }
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void head_fwd_split_kernel(const HeadArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    head_copy_image(ldsb, p.wimage, HS_END);
    __syncthreads();
    const float* lf = reinterpret_cast<const float*>(ldsb + HS_F32);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    f32x4 w6f[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) w6f[mb] = *reinterpret_cast<const f32x4*>(&lf[192 + 16 * mb + 4 * lk]);
    const float b6v = lf[256];
    const int b = blockIdx.y;
    const int HW = p.H * p.W;
    const float cid = p.census ? (float)p.census[b] : 0.f;
    float pc_sum = 0.f, sc_sum = 0.f;
    const int g_begin = (blockIdx.x * NW + wave) * p.groups_per_wave;
    int g_end = g_begin + p.groups_per_wave;
    if (g_end > p.groups) g_end = p.groups;
    // prefetch of the next group's inputs: raw values only, unconditional loads (see head_fwd_kernel)
    float bld_n = 0.f, adm_n = 0.f;
    unsigned msk_n = 1;
    const uint8_t* const mask_or_dummy = p.mask ? p.mask : reinterpret_cast<const uint8_t*>(p.building);
    const float* const admin_or_dummy = p.admin ? p.admin : p.building;
    auto fetch = [&](int g, bool& valid, float (&xv)[4]) {
        const int q = g * 16 + li;
        valid = q < HW && g < g_end;
        const int qc = valid ? q : 0;
        msk_n = mask_or_dummy[(int64_t)b * HW + qc];
        bld_n = p.building[(int64_t)b * HW + qc];
        adm_n = admin_or_dummy[(int64_t)b * HW + qc];
        const int y = (int)pc_div((uint32_t)qc, p.div_w), x = qc - y * p.W;
        const float* fp = p.feat.ptr + b * p.feat.bstride + (int64_t)(p.py + y) * p.feat.rstride + p.px + x;
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[j] = fp[(4 * lk + j) * p.feat.cstride];          // K-slots of this lane: channels 4*lk .. +3
    };
    bool valid_n = false;
    float xv_n[4] = {0.f, 0.f, 0.f, 0.f};
    if (g_begin < g_end) fetch(g_begin, valid_n, xv_n);
    for (int g = g_begin; g < g_end; ++g) {
        const int q = g * 16 + li;
        const bool valid = q < HW;
        const int64_t pix = (int64_t)b * HW + q;
        const bool sel = valid_n && (p.mask ? msk_n != 0 : true);
        const float bld = bld_n, adm = adm_n;
        float xv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[j] = valid_n ? xv_n[j] : 0.f;
        fetch(g + 1, valid_n, xv_n);
        float outv = 0.f;
        if (__any(sel)) {
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));          // opaque: keeps the fragment reads in the loop
            f32x4 h[4], acc[4];
            {
                unsigned x1[2], x2[2], x3[2];
                hs_split_pair(xv[0], xv[1], x1[0], x2[0], x3[0]);
                hs_split_pair(xv[2], xv[3], x1[1], x2[1], x3[1]);
                const hs16x4 xb[3] = {__builtin_bit_cast(hs16x4, make_uint2(x1[0], x1[1])), __builtin_bit_cast(hs16x4, make_uint2(x2[0], x2[1])),
                                      __builtin_bit_cast(hs16x4, make_uint2(x3[0], x3[1]))};
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) h[mb] = *reinterpret_cast<const f32x4*>(&lf[16 * mb + 4 * lk]);
#pragma unroll
                for (int pl = 2; pl >= 0; --pl) {
                    hs16x4 fr[4];
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) fr[mb] = hb_frag4(ldsb, pl * HS_PLANE + HB_A1, mb, lane_o);
#pragma unroll
                    for (int qq = 2 - pl; qq >= 0; --qq)
#pragma unroll
                        for (int mb = 0; mb < 4; ++mb) h[mb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(fr[mb], xb[qq], h[mb], 0, 0, 0);
                }
            }
            hb_relu4(h);
            hbf16x8 hb[3][2];
            hs_split_pack(h, hb);
            hs_layer64(ldsb, HB_A2, lf + 64, lane_o, lk, hb, acc);
            hb_relu4(acc);
            hs_split_pack(acc, hb);
            hs_layer64(ldsb, HB_A3, lf + 128, lane_o, lk, hb, h);
            hb_relu4(h);
            float s = 0.f;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) s = fmaf(w6f[mb][r], h[mb][r], s);
            s = pc_xor16_sum(s);
            s = pc_xor32_sum(s);
            outv = sel ? s + b6v : 0.f;
        }
        if (valid && lk == 0) {
            const float scale = outv > 0.f ? outv : (outv != outv ? outv : 0.f);        // NaN-propagating ReLU (see head_fwd_kernel)
            const float pd = scale * bld;
            if (p.scale_map) p.scale_map[pix] = scale;
            p.popdense[pix] = pd;
            const bool region = p.admin ? (adm == cid) : true;
            pc_sum += region ? pd : 0.f;
            sc_sum += scale;
        }
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(ldsb);
    if (lk == 0) { red[wave * 16 + li] = pc_sum; red[16 * NW + wave * 16 + li] = sc_sum; }
    __syncthreads();
    if (tid < 2) {
        float t = 0.f;
        for (int i = 0; i < 16 * NW; ++i) t += red[tid * 16 * NW + i];
        p.partial[((int64_t)b * p.nchunk + blockIdx.x) * 2 + tid] = t;
    }
}

// ---- head backward, producer / consumer form ----------------------------------------------------------------------
// The single-role kernel above needs 209 accumulator registers per wave for the three weight-gradient GEMMs, which
// pins it at ONE wave per SIMD: every global-load, LDS round trip and VALU stretch of that wave idles the matrix pipe
// (measured 0.50 of the fp32 MFMA peak).  Here a 512-thread workgroup runs two roles with <= 256 registers each, i.e.
// two waves per SIMD:
//   producers (waves 0-3): per 16-pixel group, forward recompute + the data-gradient chain (288 MFMAs, ~130 registers);
//       after each layer they hand the (G = pre-activation gradient, H = layer input) pair to their consumer through
//       a 2-slot LDS ring as (hidden x pixel) matrices;
//   consumers (waves 4-7): the weight-gradient GEMMs dW += G . H^T (144 MFMAs per group) and the bias gradients (row
//       sums of G), accumulated in registers for the whole kernel.
// Hand-off: per pair two LDS counters (produced / consumed) + a done flag, polled with s_sleep; LDS operations of a wave
// are performed in order, so "data writes, then counter write" needs no fence beyond a compiler barrier.  Slots carry
// no tag: every processed group emits exactly three slots (layer 3, 2, 1), so slot k is of kind k % 3.
constexpr int PC_LKS = 80;                                // floats per (block, lk) row of a slot matrix: 16 lanes x 4 + 16 (the 4 lk rows
                                                          // of a block start 16 banks apart: conflict-free transposing reads)
constexpr int PC_MBS = 4 * PC_LKS;                        // per 16-row block
constexpr int PC_MAT = 4 * PC_MBS;                        // one 64 x 16 matrix (1280 floats)
constexpr int PC_SLOT = 2 * PC_MAT;                       // G + H
constexpr int PC_NSLOT = 2;
constexpr int LP_RING = LB_W6 + 64 + 4;                  // weights image is shared with the single-role kernel
constexpr int LP_FLAGS = LP_RING + 4 * PC_NSLOT * PC_SLOT;
constexpr int LP_END = LP_FLAGS + 16;

// ---- split-operand producer of head_bwd_pc_kernel<DBG, true> (round 5) ---------------------------------------------------------------
// The producer's chain (forward recompute + data gradients: 288 of the kernel's 432 fp32 MFMAs per 16 pixels) on the bf16 matrix pipe
// with 3-way split operands, six partial products per K-step (see head_fwd_split_kernel): 228 instructions of 16 cycles instead of 288
// of 32.  Its results have the D layout of the fp32 instruction (both are 16 x 16 tiles in 4 registers), so the ring slots, the consumer
// waves and the reductions are untouched.  Weight image: three split planes of the ROW-MAJOR matrices W2 | W4 | W0 -- one image serves
// the forward fragments (A = W: ds_read_b128 of the lane's 8 K-slots) AND the data-gradient fragments (A = W^T: two
// ds_read_b64_tr_b16 per fragment), which is what lets three planes fit next to the ring (55 KB instead of 3 x 37).  Columns of a
// 64-wide row are stored in K-slot order (position 32 t + 8 lk + j <-> unit 16 (2t + (j >> 2)) + 4 lk + (j & 3)) and the 16-byte pieces of
// a row are XOR-swizzled with (row >> 1) & 7: the forward reads of a 16-lane service group (8 rows of one lk + the other 8 rows of its
// neighbour) then hit all 64 banks once; the transposing reads take a 2-way conflict (4 LDS cycles instead of 2: the LDS array is not
// what this kernel waits for).
constexpr int LS_W2 = 0, LS_W4 = 8192, LS_W0 = 16384, LS_PLANE = 18432;     // bytes inside a plane
constexpr int LS_F32 = 3 * LS_PLANE;                                        // floats: b0[64] b2[64] b4[64] w6[64] b6
constexpr int LS_WEND = LS_F32 + (4 * 64 + 4) * 4;                          // 56,336 bytes
constexpr int LPS_RING = LS_WEND / 4;                                       // (floats) ring and flags as in the fp32 form
constexpr int LPS_FLAGS = LPS_RING + 4 * PC_NSLOT * PC_SLOT;
constexpr int LPS_END = LPS_FLAGS + 16;
static_assert(LS_WEND % 16 == 0 && LPS_END * 4 <= 160 * 1024 && 4 * 4096 <= LPS_FLAGS, "split-producer LDS map");

__device__ __forceinline__ void head_stage_weights_bsplit(unsigned char* img, const HeadArgs& p, int tid, int nt) {
    unsigned short o[3];
    for (int e = tid; e < 64 * 64; e += nt) {
        const int row = e >> 6, c = e & 63;
        const int pos = (c >> 5) * 32 + ((c >> 2) & 3) * 8 + ((c >> 4) & 1) * 4 + (c & 3);
        const int byte = row * 128 + (((pos >> 3) ^ ((row >> 1) & 7)) * 16) + (pos & 7) * 2;
        hs_split3(p.w2[e], o);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<unsigned short*>(img + pl * LS_PLANE + LS_W2 + byte) = o[pl];
        hs_split3(p.w4[e], o);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<unsigned short*>(img + pl * LS_PLANE + LS_W4 + byte) = o[pl];
    }
    for (int e = tid; e < 64 * 16; e += nt) {
        hs_split3(p.w0[e], o);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<unsigned short*>(img + pl * LS_PLANE + LS_W0 + e * 2) = o[pl];
    }
    float* f = reinterpret_cast<float*>(img + LS_F32);
    for (int e = tid; e < 64; e += nt) {
        f[e] = p.b0[e];
        f[64 + e] = p.b2[e];
        f[128 + e] = p.b4[e];
        f[192 + e] = p.w6[e];
    }
    if (tid == 0) f[256] = p.b6[0];
}

// per-lane byte offsets of the fragment reads (computed once per wave)
struct LsLane {
    int a[2];        // forward fragment (t): row i of a 16-row block, the lane's 16-byte K-slot piece
    int t[2];        // data-gradient fragment, input-unit block pair mi >> 1 = 0 / 1: row 4 lk + (li >> 2) of a 16-row block, quad li & 3
    int a0, t0;      // W0: forward fragment (8 bytes: channels 4 lk .. + 3 of row i) / transposed (row 4 lk + (li >> 2), quad li & 3)
};
__device__ __forceinline__ LsLane ls_lane(int lane) {
    const int li = lane & 15, lk = lane >> 4;
    LsLane L;
#pragma unroll
    for (int t = 0; t < 2; ++t) L.a[t] = li * 128 + (((t * 4 + lk) ^ ((li >> 1) & 7)) * 16);
    const int sw = (2 * lk + (li >> 3)) & 7;
    L.t[0] = (4 * lk + (li >> 2)) * 128 + (((li & 3) ^ sw) * 16);
    L.t[1] = L.t[0] ^ 64;
    L.a0 = li * 32 + lk * 8;
    L.t0 = (4 * lk + (li >> 2)) * 32 + (li & 3) * 8;
    return L;
}
__device__ __forceinline__ hbf16x8 ls_frag_a(const unsigned char* w, const LsLane& L, int mb2, int t) {
    return __builtin_bit_cast(hbf16x8, *reinterpret_cast<const hu32x4*>(w + mb2 * 2048 + L.a[t]));
}
__device__ __forceinline__ hbf16x8 ls_frag_t(const unsigned char* w, const LsLane& L, int mi, int t) {
    const unsigned char* q = w + L.t[mi >> 1] + (mi & 1) * 8;
    return hc_pair(hc_tr(q + (2 * t) * 2048), hc_tr(q + (2 * t + 1) * 2048));
}
// o1[mb2] += W . h  (SPLIT forward fragments) or W^T . g (transposed fragments), six partial products per K-step, smallest first
template <bool TR>
__device__ __forceinline__ void ls_layer64(const unsigned char* planes, int w_off, const LsLane& L, const hbf16x8 (&hb)[3][2], f32x4 (&o1)[4]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int pl = 2; pl >= 0; --pl) {
            hbf16x8 fr[4];
#pragma unroll
            for (int mo = 0; mo < 4; ++mo)
                fr[mo] = TR ? ls_frag_t(planes + pl * LS_PLANE + w_off, L, mo, t) : ls_frag_a(planes + pl * LS_PLANE + w_off, L, mo, t);
#pragma unroll
            for (int q = 2 - pl; q >= 0; --q)
#pragma unroll
                for (int mo = 0; mo < 4; ++mo) o1[mo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[mo], hb[q][t], o1[mo], 0, 0, 0);
        }
}

// The same contraction with its fragment reads one step ahead of the instructions that use them (the producer is ONE wave of its role
// per SIMD: an LDS round trip in front of every 4 - 12 instructions was half of its forward phase, profiles/r5_head_bwd_split_phases.json).
// Six steps (K-step t, weight plane pl) of 4 / 8 / 12 instructions; fr: the fragments of step 0, loaded by the caller one phase earlier;
// nxt(fr): issues the reads of the first fragments of whatever contraction FOLLOWS, in the shadow of this one's last step.
template <bool TR>
__device__ __forceinline__ void ls_load4(const unsigned char* planes, int w_off, const LsLane& L, int t, int pl, hbf16x8 (&fr)[4]) {
#pragma unroll
    for (int mo = 0; mo < 4; ++mo) fr[mo] = TR ? ls_frag_t(planes + pl * LS_PLANE + w_off, L, mo, t) : ls_frag_a(planes + pl * LS_PLANE + w_off, L, mo, t);
}
template <bool TR, class Next>
__device__ __forceinline__ void ls_layer64_pf(const unsigned char* planes, int w_off, const LsLane& L, const hbf16x8 (&hb)[3][2], f32x4 (&o1)[4],
                                              hbf16x8 (&fr)[4], Next&& nxt) {
    hbf16x8 fb[4];
#pragma unroll
    for (int st = 0; st < 6; ++st) {
        const int t = st / 3, pl = 2 - st % 3;
        hbf16x8 (&cur)[4] = (st & 1) ? fb : fr;
        hbf16x8 (&oth)[4] = (st & 1) ? fr : fb;
        if (st < 5) ls_load4<TR>(planes, w_off, L, (st + 1) / 3, 2 - (st + 1) % 3, oth);
        else nxt(oth);                                   // (step 5 is odd: `oth` is fr)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 2 - pl; q >= 0; --q)
#pragma unroll
            for (int mo = 0; mo < 4; ++mo) o1[mo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[mo], hb[q][t], o1[mo], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}
// two blocks (one K-step's worth of the next contraction's operand) -> their three split planes
template <int T>
__device__ __forceinline__ void hs_split_t(const f32x4& a, const f32x4& b, hbf16x8 (&o)[3][2]) {
    hu32x4 q[3];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const f32x4& v = d < 2 ? a : b;
        unsigned q1, q2, q3;
        hs_split_pair(v[2 * (d & 1)], v[2 * (d & 1) + 1], q1, q2, q3);
        q[0][d] = q1; q[1][d] = q2; q[2][d] = q3;
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) o[pl][T] = __builtin_bit_cast(hbf16x8, q[pl]);
}
__device__ __forceinline__ void relu_block2(f32x4& a, f32x4& b) { relu_block(a); relu_block(b); }
// four fp32 values (the four K-slots of a lane in v_mfma_f32_16x16x16_bf16) -> their three split planes
__device__ __forceinline__ void hs_split4(const f32x4& v, hs16x4 (&o)[3]) {
    unsigned a1, a2, a3, b1, b2, b3;
    hs_split_pair(v[0], v[1], a1, a2, a3);
    hs_split_pair(v[2], v[3], b1, b2, b3);
    o[0] = __builtin_bit_cast(hs16x4, make_uint2(a1, b1));
    o[1] = __builtin_bit_cast(hs16x4, make_uint2(a2, b2));
    o[2] = __builtin_bit_cast(hs16x4, make_uint2(a3, b3));
}
// consumer side of the split form: dW[mb][nb] += G block mb . (H block nb)^T over the slot's 16 pixels, six partial products
template <int NB>
__device__ __forceinline__ void hs_wgrad16(const f32x4 (&af)[4], const f32x4 (&bf)[4], f32x4 (&dW)[4][4]) {
    hs16x4 as[4][3];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) hs_split4(af[mb], as[mb]);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        hs16x4 bs[3];
        hs_split4(bf[nb], bs);
#pragma unroll
        for (int pl = 2; pl >= 0; --pl)
#pragma unroll
            for (int qq = 2 - pl; qq >= 0; --qq)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
                    dW[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(as[mb][pl], bs[qq], dW[mb][nb], 0, 0, 0);
    }
}

template <int DBG, bool SPL>   // SPL: the producer's chain on split bf16 operands (above).  DBG: ablation builds (tools/ablate_head.py): 1 consumer idle, 2 no hand-off; 0 = the product kernel (as run-time flags the
                          // two switches put a branch around every ring write of the producer: 14 extra basic blocks in its group loop)
__global__ __launch_bounds__(512, 2) void head_bwd_pc_kernel(const HeadBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const HeadArgs& p = a.f;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int pair = wave & 3;
    const bool producer = wave < 4;
    // fp32 form: forward + transposed fragments, biases (head_stage_weights_bwd); SPL: the three split planes (head_stage_weights_bsplit)
    head_copy_image(lds, p.wimage, SPL ? LS_WEND : LB_SCR * (int)sizeof(float));
    int* flags = reinterpret_cast<int*>(lds + (SPL ? LPS_FLAGS : LP_FLAGS));
    if (tid < 16) flags[tid] = 0;
    __syncthreads();

    float* ring = lds + (SPL ? LPS_RING : LP_RING) + pair * PC_NSLOT * PC_SLOT;
    // LDS-address-space pointers: through a generic `volatile int*` the polls and counter updates become FLAT instructions, whose
    // completion is waited for with vmcnt(0) -- i.e. behind every global load and store the wave has in flight
    typedef __attribute__((address_space(3))) volatile int lds_flag;
    lds_flag* prod_p = (lds_flag*)(flags + pair * 4 + 0);
    lds_flag* cons_p = (lds_flag*)(flags + pair * 4 + 1);
    lds_flag* done_p = (lds_flag*)(flags + pair * 4 + 2);
    float* part = a.partial + (int64_t)blockIdx.x * PE_TOTAL;

    // Role-private accumulators: declared here (they feed the common reduction below) but initialised ONLY inside the
    // role that owns them, so that their live ranges do not extend through the other role's code (a shared
    // zero-initialisation made the allocator keep all 160 accumulator registers live in the producer: 157 spills).
    f32x4 dW4[4][4], dW2[4][4], dW0[4], dw6[4];
    float dbs[3][4];          // consumer: bias-gradient partial sums (row 16q + li, pixels = lk mod 4)
    float db6;

    if (__builtin_amdgcn_readfirstlane(wave) < 4) __builtin_amdgcn_s_setprio(2);   // producers are the critical path: they win the
                                                                                    // MFMA arbitration, the consumer fills their bubbles
    if (producer) {
        db6 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) dw6[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int HW = p.H * p.W;
        const float gsc = a.g_scale_const ? *a.g_scale_const : 0.f;
        float fscale[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            fscale[r] = 1.f;
            if (a.fuse_feat_bn) {
                const int c = 4 * lk + r;
                float sh;
                pc_bn_fold(a.fbn[c >> 3], c & 7, fscale[r], sh);
            }
        }
        int nprod = 0;
        HP_DECL;
        HR_DECL;
        HQ_NOW(hr_t);
        // c_early: the consumer's counter as read one phase earlier (it only grows: a stale value is merely conservative) -- the
        // LDS round trip of the poll is then hidden behind that phase instead of sitting in front of every slot
        auto acquire = [&](int c_early) -> float* {
            HP_ACQ0;
            if (nprod - __builtin_amdgcn_readfirstlane(c_early) >= PC_NSLOT) {
                while (true) {
                    const int c = __builtin_amdgcn_readfirstlane(*cons_p);
                    if (nprod - c < PC_NSLOT) break;
                    if (!SPL) __builtin_amdgcn_s_sleep(2);        // (SPL: the phases between two hand-offs are a third as long: poll back to back)
                }
            }
            asm volatile("" ::: "memory");
            HP_ACQ1;
            return ring + (nprod % PC_NSLOT) * PC_SLOT;
        };
        auto release = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            ++nprod;
            if (lane == 0) *prod_p = nprod;
        };
        // Per-group inputs are fetched one group ahead (raw loads only).  The group index is wave-uniform, so the sample index and
        // everything per sample live in scalar registers; optional inputs get a valid dummy address (the building map) and a
        // select instead of a branch around their load, and the last iteration prefetches a clamped (valid) group: the loop top is
        // straight-line code -- written with a branch per optional load and the (b, y, x) arithmetic done twice it cost ~2,300
        // cycles of a ~15,000-cycle group with the matrix pipe idle.
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        int vzero;
        asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
        const bool has_msk = p.mask != nullptr, has_adm = p.admin != nullptr, has_gpc = a.g_popcount != nullptr;
        const bool has_gpd = a.g_popdense != nullptr, has_gsm = a.g_scale_map != nullptr, has_fbn = a.fuse_feat_bn != 0;
        const uint8_t* msk_p = has_msk ? p.mask : reinterpret_cast<const uint8_t*>(p.building);
        const float* adm_p = has_adm ? p.admin : p.building;
        const float* gpd_p = has_gpd ? a.g_popdense : p.building;
        const float* gsm_p = has_gsm ? a.g_scale_map : p.building;
        const unsigned fcs = (unsigned)p.feat.cstride, gcs = (unsigned)a.g_feat.cstride;
        float n_xv[4], n_fv[4], n_bld = 0.f, n_adm = 0.f, n_gpd = 0.f, n_gsm = 0.f, n_gpc = 0.f;
        long long n_cen = 0;          // raw: the int64 -> float conversion at the point of USE (inside fetch it is a wait for the loads just issued)
        unsigned n_msk = 1, n_go = 0;
        bool n_valid = false;
        int n_b = 0;
        // The prefetch in five stages: inside the group loop they run one per K-step in the shadow of the second layer's MFMAs
        // (head_mm64_pf's hook); stage 4 also sends the PREVIOUS group's gradient out -- behind the loads: stored at the end of its
        // own iteration it sat in front of the next iteration's wait for the prefetched inputs (vmcnt counts loads and stores in one
        // queue).
        float pend_o[4] = {0.f, 0.f, 0.f, 0.f};
        float* pend_p = nullptr;
        const float* f_fb = nullptr;
        unsigned f_fo = 0, f_qq = 0;
        int64_t f_pb = 0;
        int f_b = 0;
        auto fetch_stage = [&](int st, int gg) {
            if (st == 0) {
                const int b = (int)pc_div((uint32_t)gg, p.div_groups), g = gg - b * p.groups;
                const int q = g * 16 + li;
                n_valid = q < HW;
                n_b = f_b = b;
                f_qq = n_valid ? (unsigned)q : 0u;
                const unsigned y = pc_div(f_qq, p.div_w), x = f_qq - y * (unsigned)p.W;
                f_fb = p.feat.ptr + b * p.feat.bstride;
                f_fo = (unsigned)(p.py + (int)y) * (unsigned)p.feat.rstride + (unsigned)p.px + x;
                n_go = (unsigned)(p.py + (int)y) * (unsigned)a.g_feat.rstride + (unsigned)p.px + x;
                f_pb = (int64_t)b * HW;
            } else if (st == 1) {
                // SPL: K-slot (lk, j) of the first layer = channel 4 lk + j (as in the split forward kernel): the operand IS the lane's
                // own four channels, which the fused ReLU backward of the feature layers reads as well (no second set of loads)
#pragma unroll
                for (int j = 0; j < 4; ++j) n_xv[j] = f_fb[f_fo + (unsigned)(SPL ? 4 * lk + j : 4 * j + lk) * fcs];
            } else if (st == 2) {
                if constexpr (!SPL) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) n_fv[j] = f_fb[f_fo + (unsigned)(4 * lk + j) * fcs];   // (only used with fuse_feat_bn; n_xv's cache lines)
                }
            } else if (st == 3) {
                n_msk = msk_p[f_pb + f_qq];
                n_bld = p.building[f_pb + f_qq];
                n_adm = adm_p[f_pb + f_qq];
                n_gpd = gpd_p[f_pb + f_qq];
                n_gsm = gsm_p[f_pb + f_qq];
            } else if (st == 4) {
                // the two per-sample scalars as well: read at their point of use they were two DEPENDENT, fully exposed memory
                // round trips at the top of every group
                // (indexed with an opaque per-lane zero: a provably uniform load result is moved to scalar registers by the compiler
                // with v_readfirstlane RIGHT HERE, behind a vmcnt(0) wait for everything the prefetch has just issued)
                if (has_adm) n_cen = p.census[f_b + vzero];
                if (has_gpc) n_gpc = a.g_popcount[f_b + vzero];
                if (pend_p) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) pend_p[(unsigned)(4 * lk + r) * gcs] = pend_o[r];
                    pend_p = nullptr;
                }
            }
        };
        auto fetch = [&](int gg) {
#pragma unroll
            for (int st = 0; st < 5; ++st) fetch_stage(st, gg);
        };
        const int gstep = gridDim.x * 4;
        constexpr bool handoff = !(DBG & 2);
        // Slot matrices are stored as the producer HOLDS them (D layout: lane (pixel li, lk), register block mb = rows 16 mb + 4 lk
        // + r): one 16-byte write per block, [mb][lk][PC_LKS] floats with the lane's 4 values at li * 4 -- the consumer does the
        // transposing reads.  (The first form scattered every element into a [row][pixel] matrix: 32 ds_write_b32 per slot on the
        // producer, the critical path, where every LDS instruction costs the wave ~30-40 cycles of MFMA issue --
        // profiles/r3_mfma_peak.json, lds variants at 1 wave per SIMD.)
        const int sbase = lk * PC_LKS + li * 4;
        auto put = [&](float* sl, int mb, const f32x4 (&G)[4], const f32x4 (&Hm)[4]) {
            *reinterpret_cast<f32x4*>(&sl[mb * PC_MBS + sbase]) = G[mb];
            *reinterpret_cast<f32x4*>(&sl[PC_MAT + mb * PC_MBS + sbase]) = Hm[mb];
        };
        // the last layer's row (this lane's 16 hidden units) and bias stay in registers: read at their point of use they are 9 LDS round
        // trips per group in a stretch with no matrix instruction to hide them
        f32x4 w6f[4], a1f[4], b0f[4];         // (and the first layer's fragments + bias: the group loop started with their round trip)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            w6f[mb] = *reinterpret_cast<const f32x4*>(&lds[(SPL ? LS_F32 / 4 + 192 : LB_W6) + 16 * mb + 4 * lk]);
            a1f[mb] = SPL ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(&lds[LB_A1 + (mb * 64 + lane) * 4]);
            b0f[mb] = *reinterpret_cast<const f32x4*>(&lds[(SPL ? LS_F32 / 4 : LB_B0) + 16 * mb + 4 * lk]);
        }
        const float b6v = lds[SPL ? LS_F32 / 4 + 256 : LB_W6 + 64];
        const LsLane LL = ls_lane(lane);
        int gg = blockIdx.x * 4 + wv;
        if (gg < a.total_groups) fetch(gg);
        for (; gg < a.total_groups; gg += gstep) {
            const bool valid = n_valid;
            const bool sel = valid && (!has_msk || n_msk != 0);
            float* const gp = a.g_feat.ptr + n_b * a.g_feat.bstride + n_go;       // this lane's pixel of the gradient map, channel 0
            float xv[4], fvv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { xv[j] = valid ? n_xv[j] : 0.f; fvv[j] = (valid && has_fbn) ? (SPL ? n_xv[j] : n_fv[j]) : 1.f; }
            float gup = 0.f;
            {
                const bool region = !has_adm || n_adm == (float)n_cen;
                float u = gsc;
                u += (has_gpc && region) ? n_gpc * n_bld : 0.f;
                u += has_gpd ? n_gpd * n_bld : 0.f;
                u += has_gsm ? n_gsm : 0.f;
                gup = sel ? u : 0.f;
            }
            const int gnx = gg + gstep < a.total_groups ? gg + gstep : a.total_groups - 1;   // (clamped: the last prefetch is a dummy)
            // g_feat is written exactly once per element by this kernel (no zero-fill pass in front of it): crop pixels
            // by the producer that owns their group -- zeros when the group is skipped --, the border by the consumers
            auto store_zero = [&]() {
                if (a.zero_in_kernel && valid) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) gp[(unsigned)(4 * lk + r) * gcs] = 0.f;
                }
            };
            if (!__any(sel)) { fetch(gnx); store_zero(); continue; }
            if constexpr (SPL) {
                // ---- the chain on split bf16 operands (straight-line: with 16-cycle instructions the layer boundaries are VALU work --
                // the splits -- that the consumer wave's fp32 MFMAs fill)
                fetch_stage(0, gnx);                      // (index arithmetic; the loads go out in the shadow of the layers' last steps)
                HR_CLOSE(0);                              // loop top (consume the prefetch)
                const unsigned char* const wpl = reinterpret_cast<const unsigned char*>(lds);
                const float* const lf = reinterpret_cast<const float*>(wpl + LS_F32);
                f32x4 h1[4], h2[4], h3[4];
                hbf16x8 ob[3][2];
                hbf16x8 fr[4];
                {
                    unsigned x1[2], x2[2], x3[2];
                    hs_split_pair(xv[0], xv[1], x1[0], x2[0], x3[0]);
                    hs_split_pair(xv[2], xv[3], x1[1], x2[1], x3[1]);
                    const hs16x4 xb[3] = {__builtin_bit_cast(hs16x4, make_uint2(x1[0], x1[1])), __builtin_bit_cast(hs16x4, make_uint2(x2[0], x2[1])),
                                          __builtin_bit_cast(hs16x4, make_uint2(x3[0], x3[1]))};
                    hs16x4 f1[3][4];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                        for (int mb = 0; mb < 4; ++mb)
                            f1[pl][mb] = __builtin_bit_cast(hs16x4, *reinterpret_cast<const uint2*>(wpl + pl * LS_PLANE + LS_W0 + mb * 512 + LL.a0));
                    ls_load4<false>(wpl, LS_W2, LL, 0, 2, fr);                 // the second layer's first fragments
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) h1[mb] = *reinterpret_cast<const f32x4*>(&lf[16 * mb + 4 * lk]);
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) h2[mb] = *reinterpret_cast<const f32x4*>(&lf[64 + 16 * mb + 4 * lk]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int pl = 2; pl >= 0; --pl)
#pragma unroll
                        for (int qq = 2 - pl; qq >= 0; --qq)
#pragma unroll
                            for (int mb = 0; mb < 4; ++mb) h1[mb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(f1[pl][mb], xb[qq], h1[mb], 0, 0, 0);
                }
                hb_relu4(h1);
                hs_split_pack(h1, ob);
                ls_layer64_pf<false>(wpl, LS_W2, LL, ob, h2, fr, [&](hbf16x8 (&f)[4]) {
                    ls_load4<false>(wpl, LS_W4, LL, 0, 2, f);
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) h3[mb] = *reinterpret_cast<const f32x4*>(&lf[128 + 16 * mb + 4 * lk]);
                    fetch_stage(1, gnx);
                    fetch_stage(2, gnx);
                });
                hb_relu4(h2);
                hs_split_pack(h2, ob);
                ls_layer64_pf<false>(wpl, LS_W4, LL, ob, h3, fr, [&](hbf16x8 (&f)[4]) {
                    ls_load4<true>(wpl, LS_W4, LL, 0, 2, f);
                    fetch_stage(3, gnx);
                    fetch_stage(4, gnx);
                });
                hb_relu4(h3);
                HR_CLOSE(1);                              // forward chain
                const int c_early0 = *cons_p;
                float s = 0.f;
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s = fmaf(w6f[mb][r], h3[mb][r], s);
                s = pc_xor16_sum(s);
                s = pc_xor32_sum(s);
                const float outv = s + b6v;
                const float gout = (sel && outv > 0.f) ? gup : 0.f;
                if (!__any(gout != 0.f)) { store_zero(); continue; }
                f32x4 g3[4], g2[4], g1[4];
                if (lk == 0) db6 += gout;
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        dw6[mb][r] = fmaf(gout, h3[mb][r], dw6[mb][r]);
                        g3[mb][r] = h3[mb][r] > 0.f ? w6f[mb][r] * gout : 0.f;
                    }
                HR_CLOSE(2);                              // output layer + G3
                float* sl = handoff ? acquire(c_early0) : nullptr;
                HR_CLOSE(3);                              // wait for slot 0
                const int c_early1 = *cons_p;
                if (handoff) {
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) put(sl, mb, g3, h2);
                    release();
                }
                HR_CLOSE(4);                              // slot 0 writes
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) g2[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
                hs_split_pack(g3, ob);
                ls_layer64_pf<true>(wpl, LS_W4, LL, ob, g2, fr, [&](hbf16x8 (&f)[4]) { ls_load4<true>(wpl, LS_W2, LL, 0, 2, f); });
                mask_block(g2[0], h2[0]);
                mask_block(g2[1], h2[1]);
                mask_block(g2[2], h2[2]);
                mask_block(g2[3], h2[3]);
                HR_CLOSE(5);                              // data gradient of layer 3
                sl = handoff ? acquire(c_early1) : nullptr;
                HR_CLOSE(6);
                const int c_early2 = *cons_p;
                if (handoff) {
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) put(sl, mb, g2, h1);
                    release();
                }
                HR_CLOSE(7);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) g1[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
                hs_split_pack(g2, ob);
                // (what follows is the 16-channel data gradient: its fragments of planes 2 and 1 -- [plane][K-step] -- come with the last step)
                ls_layer64_pf<true>(wpl, LS_W2, LL, ob, g1, fr, [&](hbf16x8 (&f)[4]) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const unsigned char* q0 = wpl + (2 - (k >> 1)) * LS_PLANE + LS_W0 + LL.t0 + (k & 1) * 32 * 32;
                        f[k] = hc_pair(hc_tr(q0), hc_tr(q0 + 16 * 32));
                    }
                });
                mask_block(g1[0], h1[0]);
                mask_block(g1[1], h1[1]);
                mask_block(g1[2], h1[2]);
                mask_block(g1[3], h1[3]);
                HR_CLOSE(8);                              // data gradient of layer 2
                sl = handoff ? acquire(c_early2) : nullptr;
                HR_CLOSE(9);
                if (handoff) {
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) *reinterpret_cast<f32x4*>(&sl[mb * PC_MBS + sbase]) = g1[mb];
                    *reinterpret_cast<f32x4*>(&sl[PC_MAT + sbase]) = f32x4{xv[0], xv[1], xv[2], xv[3]};
                    release();
                }
                HR_CLOSE(10);
                hbf16x8 fz[2];                            // plane 0 of the 16-channel data gradient's fragments
                {
                    const unsigned char* q0 = wpl + LS_W0 + LL.t0;
                    fz[0] = hc_pair(hc_tr(q0), hc_tr(q0 + 16 * 32));
                    fz[1] = hc_pair(hc_tr(q0 + 32 * 32), hc_tr(q0 + 48 * 32));
                }
                hs_split_pack(g1, ob);
                f32x4 gx = f32x4{0.f, 0.f, 0.f, 0.f}, gx2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int pl = 2; pl >= 0; --pl) {
                    const hbf16x8 f0 = pl == 0 ? fz[0] : fr[2 * (2 - pl)], f1 = pl == 0 ? fz[1] : fr[2 * (2 - pl) + 1];
#pragma unroll
                    for (int qq = 2 - pl; qq >= 0; --qq) {
                        gx = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f0, ob[qq][0], gx, 0, 0, 0);
                        gx2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1, ob[qq][1], gx2, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) gx[r] += gx2[r];
                if (valid) {
                    pend_p = gp;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float o = gx[r];
                        if (a.fuse_feat_bn) o = fvv[r] > 0.f ? o * fscale[r] : 0.f;
                        pend_o[r] = o;
                    }
                }
                HR_CLOSE(11);                             // data gradient of layer 1 (features)
            } else {
            HP_MARK(0);
            // Every contraction gets its first fragments from the phase before it (fa / fb, alternating), and the ReLU / ReLU' of its
            // input blocks 1-3 happens in the shadow of its own first K-steps (block mb + 1 in step 4 mb): a layer boundary costs the
            // ReLU of ONE block, not an LDS round trip + 16-32 VALU instructions with the matrix pipe idle.
            f32x4 h1[4], h2[4], h3[4], fa[4], fb[4];
            head_frag0(lds, LB_A2, 4, lane, fa);
            head_bias4(lds, LB_B2, lk, h2);          // (in flight during the first layer: the scheduling barriers below keep it up here)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) h1[mb] = b0f[mb];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) h1[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1f[mb][j], xv[j], h1[mb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            relu_block(h1[0]);
            head_mm64_pf(lds, LB_A2, lane, h1, h2, fa, LB_A3, 4, fb, [&](int st) {
                if (!(st & 3) && st < 12) relu_block(h1[(st >> 2) + 1]);
                if ((st & 1) && st < 10) fetch_stage(st >> 1, gnx);
                if (st == 12) head_bias4(lds, LB_B4, lk, h3);
            });
            relu_block(h2[0]);
            head_mm64_pf(lds, LB_A3, lane, h2, h3, fb, LB_T3, 4, fa, [&](int st) { if (!(st & 3) && st < 12) relu_block(h2[(st >> 2) + 1]); });
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) relu_block(h3[mb]);
            HP_MARK(1);
            const int c_early0 = *cons_p;
            float s = 0.f;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) s = fmaf(w6f[mb][r], h3[mb][r], s);
            s = pc_xor16_sum(s);
            s = pc_xor32_sum(s);
            const float outv = s + b6v;
            const float gout = (sel && outv > 0.f) ? gup : 0.f;
            if (!__any(gout != 0.f)) { store_zero(); continue; }

            f32x4 g3[4], g2[4], g1[4];
            if (lk == 0) db6 += gout;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dw6[mb][r] = fmaf(gout, h3[mb][r], dw6[mb][r]);
                    g3[mb][r] = h3[mb][r] > 0.f ? w6f[mb][r] * gout : 0.f;
                }
            HP_MARK(2);
            // The three hand-offs (slot kind 0: (G3, H2) -> dW4, db4; 1: (G2, H1) -> dW2, db2; 2: (G1, X) -> dW0, db0) are written
            // one element pair per K-step in the shadow of the NEXT contraction's MFMAs -- as a block in front of it, the 32 scattered
            // LDS writes + their wait were ~600-1,000 cycles per slot with the matrix pipe idle.
            float* sl = handoff ? acquire(c_early0) : nullptr;
            HP_MARK(3);
            const int c_early1 = *cons_p;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) g2[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
            head_mm64_pf(lds, LB_T3, lane, g3, g2, fa, LB_T2, 4, fb, [&](int st) { if (handoff && (st & 3) == 1) put(sl, st >> 2, g3, h2); });
            if (handoff) release();
            mask_block(g2[0], h2[0]);
            HP_MARK(4);
            sl = handoff ? acquire(c_early1) : nullptr;
            HP_MARK(5);
            const int c_early2 = *cons_p;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) g1[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
            head_mm64_pf(lds, LB_T2, lane, g2, g1, fb, LB_T1, 1, fa, [&](int st) {
                if (handoff && (st & 3) == 1) put(sl, st >> 2, g2, h1);           // (block mb was masked in step 4 (mb - 1))
                if (!(st & 3) && st < 12) mask_block(g2[(st >> 2) + 1], h2[(st >> 2) + 1]);
            });
            if (handoff) release();
            mask_block(g1[0], h1[0]);
            mask_block(g1[1], h1[1]);
            HP_MARK(6);
            sl = handoff ? acquire(c_early2) : nullptr;
            HP_MARK(7);
            f32x4 gx = f32x4{0.f, 0.f, 0.f, 0.f}, gx2 = f32x4{0.f, 0.f, 0.f, 0.f};   // two chains: the MFMA dependent latency
#pragma unroll                                                                       // (40 cyc) exceeds the issue interval (32)
            for (int mb = 0; mb < 4; mb += 2) {
                const f32x4 t4 = fa[mb], u4 = fa[mb + 1];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    gx = __builtin_amdgcn_mfma_f32_16x16x4f32(t4[r], g1[mb][r], gx, 0, 0, 0);
                    gx2 = __builtin_amdgcn_mfma_f32_16x16x4f32(u4[r], g1[mb + 1][r], gx2, 0, 0, 0);
                    if (mb == 0 && r < 2) mask_block(g1[2 + r], h1[2 + r]);
                    if (handoff) {            // G1 block r in step (0, r) -- blocks 2, 3 are masked in steps (0, 0), (0, 1) --, X in (2, 0)
                        if (mb == 0) *reinterpret_cast<f32x4*>(&sl[r * PC_MBS + sbase]) = g1[r];
                        else if (r == 0) *reinterpret_cast<f32x4*>(&sl[PC_MAT + sbase]) = f32x4{xv[0], xv[1], xv[2], xv[3]};
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (handoff) release();
#pragma unroll
            for (int r = 0; r < 4; ++r) gx[r] += gx2[r];
            if (valid) {
                pend_p = gp;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float o = gx[r];
                    if (a.fuse_feat_bn) o = fvv[r] > 0.f ? o * fscale[r] : 0.f;
                    pend_o[r] = o;
                }
            }
            }      // (!SPL)
        }
        if (pend_p) {
#pragma unroll
            for (int r = 0; r < 4; ++r) pend_p[(unsigned)(4 * lk + r) * gcs] = pend_o[r];
        }
        if constexpr (SPL) { HR_DUMP; } else { HP_DUMP; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) *done_p = 1;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dW0[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            dbs[0][i] = dbs[1][i] = dbs[2][i] = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) dW4[i][j] = dW2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (a.zero_in_kernel) {
            // The consumers have nothing to do until the first slot arrives: they write the zero border of g_feat (the
            // padding frame around the H x W crop: 39 % of a 128 x 128 map for 100 x 100 tiles) -- the only part of the
            // gradient map the producers never touch.  One (b, c, row) job per half-wave: rows above / below the crop in
            // full (16-byte stores when the row allows), crop rows only left and right of it.
            const int Hp = a.Hp, Wp = a.Wp;
            const int l32 = tid & 31;
            const int nhw = gridDim.x * 8, hw = blockIdx.x * 8 + ((tid - 256) >> 5);
            const int njobs = p.B * 16 * Hp;
            const bool v4 = (Wp & 3) == 0;
            const int right0 = p.px + p.W;
            for (int j = hw; j < njobs; j += nhw) {
                const int row = j % Hp;
                float* rp = a.g_feat.ptr + (int64_t)(j / Hp) * a.g_feat.cstride + (int64_t)row * a.g_feat.rstride;
                if (row < p.py || row >= p.py + p.H) {
                    if (v4) for (int x4 = 4 * l32; x4 < Wp; x4 += 128) *reinterpret_cast<f32x4*>(rp + x4) = f32x4{0.f, 0.f, 0.f, 0.f};
                    else for (int x1 = l32; x1 < Wp; x1 += 32) rp[x1] = 0.f;
                } else if (p.px <= 16 && Wp - right0 <= 16) {
                    // both side strips in one store: lanes 0-15 the left one, lanes 16-31 the right one
                    const int xs = l32 < 16 ? l32 : right0 + (l32 - 16);
                    if (l32 < 16 ? l32 < p.px : xs < Wp) rp[xs] = 0.f;
                } else {
                    for (int x1 = l32; x1 < p.px; x1 += 32) rp[x1] = 0.f;
                    for (int x1 = right0 + l32; x1 < Wp; x1 += 32) rp[x1] = 0.f;
                }
            }
        }
        int ncons = 0;
        // wait until slot `ncons` has been produced; returns false when the producer has finished without producing it
        auto wait_slot = [&]() -> bool {
            while (true) {
                if (__builtin_amdgcn_readfirstlane(*prod_p) > ncons) return true;
                if (__builtin_amdgcn_readfirstlane(*done_p)) return __builtin_amdgcn_readfirstlane(*prod_p) > ncons;
                __builtin_amdgcn_s_sleep(2);
            }
        };
        // fetch the (G, H) fragments of the current slot and hand the slot back
        auto take = [&](f32x4 (&af)[4], f32x4 (&bf)[4], bool x_only) {
            asm volatile("" ::: "memory");
            // transposing reads: fragment (q, ks) = element (row 16 q + li, pixel 4 ks + lk) = register li & 3 of the producer lane
            // (pixel, lk' = li >> 2), block q; for X (16 channels x 16 pixels, channel 4 j + lk' in register j) the two are swapped.
            // Lanes hit 16 (li >> 2) + 4 lk + (li & 3) (mod 32) = every bank twice: conflict-free 4-byte reads, paired by the compiler
            // into ds_read2_b32 (ks, ks + 1 are 16 floats apart).
            const float* gm = ring + (ncons % PC_NSLOT) * PC_SLOT;
            const float* hm = gm + PC_MAT;
            const int cb = (li >> 2) * PC_LKS + lk * 4 + (li & 3), xb = (li & 3) * PC_LKS + lk * 4 + (li >> 2);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    af[q][ks] = gm[q * PC_MBS + cb + 16 * ks];
                    bf[q][ks] = x_only ? hm[(SPL ? cb : xb) + 16 * ks] : hm[q * PC_MBS + cb + 16 * ks];     // (SPL: X holds channel 4 lk' + j in register j)
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            ++ncons;
            if (lane == 0) *cons_p = ncons;          // slot is free again: everything needed is in registers
        };
        // every processed group emits exactly three slots, in this order: no tags, no dynamic dispatch
        while (wait_slot()) {
            f32x4 af[4], bf[4];
            take(af, bf, false);                                             // (G3, H2) -> dW4, db4
            if (DBG & 1) { wait_slot(); take(af, bf, false); wait_slot(); take(af, bf, true); continue; }
#pragma unroll
            for (int q = 0; q < 4; ++q) dbs[2][q] += (af[q][0] + af[q][1]) + (af[q][2] + af[q][3]);
            if constexpr (SPL) hs_wgrad16<4>(af, bf, dW4);
            else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb)
                        dW4[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mb][ks], bf[nb][ks], dW4[mb][nb], 0, 0, 0);
            }
            wait_slot();
            take(af, bf, false);                                             // (G2, H1) -> dW2, db2
#pragma unroll
            for (int q = 0; q < 4; ++q) dbs[1][q] += (af[q][0] + af[q][1]) + (af[q][2] + af[q][3]);
            if constexpr (SPL) hs_wgrad16<4>(af, bf, dW2);
            else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb)
                        dW2[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mb][ks], bf[nb][ks], dW2[mb][nb], 0, 0, 0);
            }
            wait_slot();
            take(af, bf, true);                                              // (G1, X) -> dW0, db0
#pragma unroll
            for (int q = 0; q < 4; ++q) dbs[0][q] += (af[q][0] + af[q][1]) + (af[q][2] + af[q][3]);
            if constexpr (SPL) {
                hs16x4 xs[3];
                hs_split4(bf[0], xs);
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    hs16x4 as[3];
                    hs_split4(af[mb], as);
#pragma unroll
                    for (int pl = 2; pl >= 0; --pl)
#pragma unroll
                        for (int qq = 2 - pl; qq >= 0; --qq)
                            dW0[mb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(as[pl], xs[qq], dW0[mb], 0, 0, 0);
                }
            } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
                    dW0[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mb][ks], bf[0][ks], dW0[mb], 0, 0, 0);
            }
        }
    }

    // ---- reductions: producers own dw6 / db6, consumers own dW4, dW2, dW0 and the bias sums
    if (producer) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) dw6[mb][r] = lane_sum16(dw6[mb][r]);
        db6 = lane_sum16(db6);
    } else {
#pragma unroll
        for (int l3 = 0; l3 < 3; ++l3)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v = dbs[l3][q];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                dbs[l3][q] = v;
            }
    }
    __syncthreads();
#pragma unroll
    for (int stage = 0; stage < 2; ++stage) {
        if (!producer) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
                    *reinterpret_cast<f32x4*>(&lds[pair * 4096 + ((mb * 4 + nb) * 64 + lane) * 4]) = stage == 0 ? dW4[mb][nb] : dW2[mb][nb];
        }
        __syncthreads();
        for (int e = tid; e < 4096; e += 512)
            part[(stage == 0 ? PE_W4 : PE_W2) + e] = ((lds[e] + lds[4096 + e]) + lds[8192 + e]) + lds[12288 + e];
        __syncthreads();
    }
    {
        float* w = lds + pair * 1344;       // [dW0 1024][dw6 64][db0 64][db2 64][db4 64][db6 1]
        if (!producer) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) *reinterpret_cast<f32x4*>(&w[(mb * 64 + lane) * 4]) = dW0[mb];
            if (lk == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    w[1088 + 16 * q + li] = dbs[0][q];
                    w[1152 + 16 * q + li] = dbs[1][q];
                    w[1216 + 16 * q + li] = dbs[2][q];
                }
            }
        } else if (li == 0) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) w[1024 + 16 * mb + 4 * lk + r] = dw6[mb][r];
            if (lk == 0) w[1280] = db6;
        }
        __syncthreads();
        for (int e = tid; e < 1281; e += 512) {
            const float t = ((lds[e] + lds[1344 + e]) + lds[2688 + e]) + lds[4032 + e];
            part[PE_W0 + e] = t;
        }
    }
}

// ---- bf16 backward, cooperative form -------------------------------------------------------------------------------------
// The first bf16 kernel kept all 160 weight-gradient accumulator registers in every wave and computed every chain twice (a
// second, transposed orientation: mfma with swapped operands) to get their operands: one wave per SIMD, 116 MFMAs and two sets
// of bias / ReLU / pack epilogues per 16 pixels, nothing to overlap with (198 us per launch at B = 64).
// Here the waves of a workgroup SHARE the weight gradients: every wave runs the forward + backward chain of its own 16-pixel
// group in the first orientation only (lane = pixel), writes the packed operands H1, H2, G3, G2, G1, X as [pixel][channel] rows
// into its slot of an LDS exchange area, and after a barrier accumulates only ITS blocks of dW4 / dW2 / dW0 over all the
// workgroup's groups: the contraction over pixels takes both operands through ds_read_b64_tr_b16 ([4 pixels][16 channels] ->
// lane = channel, 4 pixels), K = 32 pixels per v_mfma_f32_16x16x32_bf16.  Bias gradients are one more MFMA against an all-ones
// operand (every column of the block holds the row sums).
// Workgroup shape: FOUR waves, TWO workgroups per CU (round 2 went 8 waves x 1 workgroup, 118 us -> this, 101 us: the 8-wave form
// waited 53 % of its wave cycles at the two barriers of an iteration with nothing else resident on the CU).  To fit twice into
// 160 KiB the exchange rows have no padding (128 bytes per pixel; 32 for the features); the 8-byte pieces of a row are
// XOR-swizzled with the row index instead -- on the writes and on the transposing reads alike -- so that the 16 pixel rows of a
// lane group still hit 16 different bank pairs.  A wave owns the row block mb = wave of dW4 / dW2 (4 column blocks each) and of
// dW0: 38 chain + 15 weight-gradient MFMAs per wave and group, 48 accumulator registers.
constexpr int HC_EX = (HB_END + 15) & ~15;        // exchange area behind the weight images


constexpr int H4_ROW = 128, H4_T = 16 * H4_ROW;
constexpr int H4_H1 = 0, H4_H2 = H4_T, H4_G3 = 2 * H4_T, H4_G2 = 3 * H4_T, H4_G1 = 4 * H4_T, H4_X = 5 * H4_T;
constexpr int H4_XROW = 32;
constexpr int H4_SLOT = H4_X + 16 * H4_XROW;      // 10752 bytes per group
constexpr int H4_WAVES = 4;
constexpr int H4_END = HC_EX + H4_WAVES * H4_SLOT;
static_assert(2 * H4_END <= 160 * 1024, "two workgroups per CU");
__global__ __launch_bounds__(256, 2) void head_bwd_bf16_coop4_kernel(const HeadBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    const HeadArgs& p = a.f;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    head_copy_image(ldsb, p.wimage, HB_END);
    // the exchange area starts as zeros: a slot that was never written must not feed NaN bit patterns into 0 * x
    for (int e = tid; e < H4_WAVES * H4_SLOT / 16; e += 256) reinterpret_cast<uint4*>(ldsb + HC_EX)[e] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    const float* lf = reinterpret_cast<const float*>(ldsb + HB_F32);
    float* part = a.partial + (int64_t)blockIdx.x * PE_TOTAL;
    f32x4 w6f[4];                  // the last layer's (rounded) row and bias stay in registers
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) w6f[mb] = *reinterpret_cast<const f32x4*>(&lf[192 + 16 * mb + 4 * lk]);
    const float b6v = lf[256];
    const int HW = p.H * p.W;

    if (a.zero_in_kernel) {
        // padding frame of the channels-last gradient map (the crop is written below, zeros included): one (b, row) job per half-wave
        const int Hp = a.Hp, Wp = a.Wp;
        const int l32 = tid & 31;
        const int nhw = gridDim.x * 8, hw = blockIdx.x * 8 + (tid >> 5);
        const int njobs = p.B * Hp;
        const int right0 = p.px + p.W;
        const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
        for (int j = hw; j < njobs; j += nhw) {
            const int row = j % Hp;
            pc_bf16_t* rp = reinterpret_cast<pc_bf16_t*>(a.g_feat.ptr) + (int64_t)(j / Hp) * a.g_feat.bstride + (int64_t)row * a.g_feat.rstride;
            const int xs = a.g_feat.xstride;
            if (row < p.py || row >= p.py + p.H) {
                for (int i = l32; i < 2 * Wp; i += 32) *reinterpret_cast<uint4*>(rp + (int64_t)(i >> 1) * xs + 8 * (i & 1)) = z4;
            } else {
                const int nl = 2 * p.px, nr = 2 * (Wp - right0);
                for (int i = l32; i < nl + nr; i += 32) {
                    const int k = i < nl ? i : i - nl;
                    const int xpix = i < nl ? (k >> 1) : right0 + (k >> 1);
                    *reinterpret_cast<uint4*>(rp + (int64_t)xpix * xs + 8 * (k & 1)) = z4;
                }
            }
        }
    }

    // ---- this wave's blocks: row block mb = wave of dW4 / dW2 (column blocks 0..3) and of dW0
    const int my_mb = wave;
    f32x4 dW4[4], dW2[4], dW0 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) dW4[i] = dW2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // bias gradients = row sums of G3 / G2 / G1 over the pixels: one more MFMA against an all-ones operand (every column of the block
    // then holds the row sums) instead of 7 VALU per transposed read -- the matrix pipe is 13 % busy, the VALU is the limiter
    f32x4 dB4 = f32x4{0.f, 0.f, 0.f, 0.f}, dB2 = dB4, dB0 = dB4;
    const hbf16x8 ones8 = __builtin_bit_cast(hbf16x8, hu32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});
    f32x4 dw6[4];
    float db6 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) dw6[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float gsc = a.g_scale_const ? *a.g_scale_const : 0.f;
    float fscale[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        fscale[r] = 1.f;
        if (a.fuse_feat_bn) {
            const int c = 4 * lk + r;
            float sh;
            pc_bn_fold(a.fbn[c >> 3], c & 7, fscale[r], sh);
        }
    }
    unsigned char* const ex = ldsb + HC_EX;
    unsigned char* const my = ex + wave * H4_SLOT;
    // transposing reads: lane supplies pixel row r0 = 8 * (lk & 1) + (li >> 2) (second read: r0 + 4) and column quad q = li & 3 of a
    // [4 pixels][16 channels] block; k-group lk = pixels 8 * lk .. + 7 of a 32-pixel pair of groups.  Physical 8-byte piece of
    // (row r, piece p) = p ^ (r & 15): for p = 4 * mb + q the low two bits are q ^ (r & 3), the block index mb ^ (r >> 2).
    const int t_r0 = 8 * (lk & 1) + (li >> 2);
    const int t_off = (lk >> 1) * H4_SLOT + t_r0 * H4_ROW + 8 * ((li & 3) ^ (t_r0 & 3));
    const int t_sel = 32 * (t_r0 >> 2);                       // XOR on the 32-byte block offset: first read; second read: ^ 32
    // features: 4 pieces per row, physical piece = p ^ ((r ^ (r >> 2)) & 3)
    const int t_offx = (lk >> 1) * H4_SLOT + H4_X + t_r0 * H4_XROW + 8 * ((li & 3) ^ ((t_r0 ^ (t_r0 >> 2)) & 3));
    const int t_offx2 = (lk >> 1) * H4_SLOT + H4_X + (t_r0 + 4) * H4_XROW + 8 * ((li & 3) ^ (((t_r0 + 4) ^ ((t_r0 + 4) >> 2)) & 3));

    // per-group inputs, fetched one group ahead
    float n_bld = 0.f, n_adm = 0.f, n_gpd = 0.f, n_gsm = 0.f, n_gpc = 0.f;
    long long n_cen = 0;          // the two per-sample scalars ride with the prefetch (raw; converted where used): read at their point of
                                  // use they are two dependent, fully exposed memory round trips per group (see head_bwd_pc_kernel)
    uint2 n_f4 = make_uint2(0u, 0u);
    unsigned n_msk = 1;
    auto fetch = [&](int gg) {
        const int b = (int)pc_div((uint32_t)gg, p.div_groups), g = gg - b * p.groups;
        const int q = g * 16 + li;
        const bool valid = q < HW;
        const int64_t pix = (int64_t)b * HW + (valid ? q : 0);
        const int y = valid ? (int)pc_div((uint32_t)q, p.div_w) : 0, x = valid ? q - y * p.W : 0;
        const pc_bf16_t* fp = reinterpret_cast<const pc_bf16_t*>(p.feat.ptr) + b * p.feat.bstride + (int64_t)(p.py + y) * p.feat.rstride +
                              (int64_t)(p.px + x) * p.feat.xstride;
        n_f4 = *reinterpret_cast<const uint2*>(fp + 4 * lk);          // channels 4*lk .. +3: the layer-1 operand (K-slots of this lane), the
                                                                      // mask of the fused ReLU backward AND the dW0 operand -- raw bits:
                                                                      // nothing here may use a loaded value (see head_fwd_bf16_kernel)
        n_msk = p.mask ? p.mask[pix] : 1;
        n_bld = p.building[pix];
        if (p.admin) { n_adm = p.admin[pix]; n_cen = p.census[b]; }
        if (a.g_popdense) n_gpd = a.g_popdense[pix];
        if (a.g_scale_map) n_gsm = a.g_scale_map[pix];
        if (a.g_popcount) n_gpc = a.g_popcount[b];
    };
    const int gstep = gridDim.x * H4_WAVES;
    const int niter = (a.total_groups + gstep - 1) / gstep;
    int gg = blockIdx.x * H4_WAVES + wave;
    if (gg < a.total_groups) fetch(gg);
    HQ_DECL;
    HQ_NOW(hq_t);
    for (int it = 0; it < niter; ++it, gg += gstep) {
        const bool live = gg < a.total_groups;
        bool wrote = false;
        if (live) {
            const int b = (int)pc_div((uint32_t)gg, p.div_groups), g = gg - b * p.groups;
            const int q = g * 16 + li;
            const bool valid = q < HW;
            const int y = valid ? (int)pc_div((uint32_t)q, p.div_w) : 0, x = valid ? q - y * p.W : 0;
            const bool sel = valid && n_msk != 0;
            const uint2 f4 = valid ? n_f4 : make_uint2(0u, 0u);
            float gup = 0.f;
            if (sel) {
                const bool region = p.admin ? (n_adm == (float)n_cen) : true;
                gup = gsc;
                if (a.g_popcount && region) gup += n_gpc * n_bld;
                if (a.g_popdense) gup += n_gpd * n_bld;
                if (a.g_scale_map) gup += n_gsm;
            }
            if (gg + gstep < a.total_groups) fetch(gg + gstep);
            pc_bf16_t* const gxp = reinterpret_cast<pc_bf16_t*>(a.g_feat.ptr) + b * a.g_feat.bstride + (int64_t)(p.py + y) * a.g_feat.rstride +
                                   (int64_t)(p.px + x) * a.g_feat.xstride + 4 * lk;
            bool active = __any(sel);
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));              // opaque: keeps the weight-fragment reads inside the loop
            hbf16x8 hb1[2], hb2[2];
            hu32x4 m1[2], m2[2];                          // 0xffff per bf16 half of hb1 / hb2 that is non-zero (= passed its ReLU)
            f32x4 h3[4];
            float gout = 0.f;
            HQ_CLOSE(hq0);                            // phase 0: loop top (consume the prefetch, issue the next one)
            if (active) {
                // ---- forward chain (lane = pixel, registers = hidden 16*mb + 4*lk + r)
                const hs16x4 xb = __builtin_bit_cast(hs16x4, f4);
                {
                    f32x4 h1[4];
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) {
                        h1[mb] = *reinterpret_cast<const f32x4*>(&lf[16 * mb + 4 * lk]);
                        h1[mb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(hb_frag4(ldsb, HB_A1, mb, lane_o), xb, h1[mb], 0, 0, 0);
                    }
                    hb_relu4(h1);
                    hb1[0] = hb_pack8(h1[0], h1[1]); hb1[1] = hb_pack8(h1[2], h1[3]);            // the pack rounds to bf16
                    m1[0] = hb_nzmask(hb1[0]); m1[1] = hb_nzmask(hb1[1]);
                }
                {
                    f32x4 h2[4];
                    hb_layer64(ldsb, HB_A2, lf + 64, lane_o, lk, hb1, h2);
                    hb_relu4(h2);
                    hb2[0] = hb_pack8(h2[0], h2[1]); hb2[1] = hb_pack8(h2[2], h2[3]);
                    m2[0] = hb_nzmask(hb2[0]); m2[1] = hb_nzmask(hb2[1]);
                    hb_layer64(ldsb, HB_A3, lf + 128, lane_o, lk, hb2, h3);
                    hb_relu_round(h3);
                }
                float s = 0.f;
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s = fmaf(w6f[mb][r], h3[mb][r], s);
                s = pc_xor16_sum(s);
                s = pc_xor32_sum(s);
                const float outv = s + b6v;
                gout = (sel && outv > 0.f) ? gup : 0.f;
                active = __any(gout != 0.f);
            }
            HQ_CLOSE(hq1);                            // phase 1: forward chain
            if (active) {
                if (lk == 0) db6 += gout;
                // ---- backward chain: G3 = relu'(h3) . w6 . gout;  G2 = relu'(h2) . (W4^T G3);  G1 = relu'(h1) . (W2^T G2);  g_x = W0^T G1
                hbf16x8 gb3[2], gb2[2], gb1[2];
                {
                    f32x4 g3[4];
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            dw6[mb][r] = fmaf(gout, h3[mb][r], dw6[mb][r]);
                            g3[mb][r] = h3[mb][r] > 0.f ? w6f[mb][r] * gout : 0.f;
                        }
                    gb3[0] = hb_pack8(g3[0], g3[1]); gb3[1] = hb_pack8(g3[2], g3[3]);
                }
                {
                    f32x4 g2[4];
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) {
                        g2[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int t = 0; t < 2; ++t)
                            g2[mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hb_frag8(ldsb, HB_T3, mi, t, lane_o), gb3[t], g2[mi], 0, 0, 0);
                    }
                    // the ReLU mask on the PACKED gradient: one AND per two elements
                    gb2[0] = hb_and(hb_pack8(g2[0], g2[1]), m2[0]); gb2[1] = hb_and(hb_pack8(g2[2], g2[3]), m2[1]);
                }
                {
                    f32x4 g1[4];
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) {
                        g1[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int t = 0; t < 2; ++t)
                            g1[mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hb_frag8(ldsb, HB_T2, mi, t, lane_o), gb2[t], g1[mi], 0, 0, 0);
                    }
                    gb1[0] = hb_and(hb_pack8(g1[0], g1[1]), m1[0]); gb1[1] = hb_and(hb_pack8(g1[2], g1[3]), m1[1]);
                }
                f32x4 gx = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 2; ++t) gx = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hb_frag8(ldsb, HB_T1, 0, t, lane_o), gb1[t], gx, 0, 0, 0);
                if (valid) {
                    const f32x4 fv = f32x4{__uint_as_float(f4.x << 16), __uint_as_float(f4.x & 0xffff0000u), __uint_as_float(f4.y << 16),
                                           __uint_as_float(f4.y & 0xffff0000u)};
                    f32x4 o4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float o = gx[r];
                        if (a.fuse_feat_bn) o = fv[r] > 0.f ? o * fscale[r] : 0.f;
                        o4[r] = o;
                    }
                    pc_st4(gxp, o4);
                }
                HQ_CLOSE(hq2);                        // phase 2: backward chain + the gradient store
                // ---- operands of the weight gradients into this wave's slot: rows = pixels, 8-byte pieces of 4 channels
                // (pack t holds hidden 16*(2t) + 4*lk .. +3 and 16*(2t+1) + 4*lk .. +3)
                // piece p = lk + 4 * k of row li goes to physical piece p ^ li: low bits lk ^ (li & 3), block k ^ (li >> 2)
                auto put = [&](int tensor, const hbf16x8 (&v)[2]) {
                    unsigned char* d = my + tensor + li * H4_ROW + 8 * (lk ^ (li & 3));
                    const int bs = 32 * (li >> 2);
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const hu32x4 q4 = __builtin_bit_cast(hu32x4, v[t]);
                        *reinterpret_cast<uint2*>(d + ((32 * (2 * t)) ^ bs)) = make_uint2(q4[0], q4[1]);
                        *reinterpret_cast<uint2*>(d + ((32 * (2 * t + 1)) ^ bs)) = make_uint2(q4[2], q4[3]);
                    }
                };
                put(H4_H1, hb1); put(H4_H2, hb2); put(H4_G3, gb3); put(H4_G2, gb2); put(H4_G1, gb1);
                *reinterpret_cast<uint2*>(my + H4_X + li * H4_XROW + 8 * (lk ^ ((li ^ (li >> 2)) & 3))) = f4;
                wrote = true;
            } else if (a.zero_in_kernel && valid) {
                *reinterpret_cast<uint2*>(gxp) = make_uint2(0u, 0u);
            }
        }
        if (!wrote) {
            // nothing to contribute: zero gradients in the slot (the H / X rows of an earlier group stay: finite values times zero)
            const uint2 z2 = make_uint2(0u, 0u);
#pragma unroll
            for (int tz = 0; tz < 3; ++tz) {
                unsigned char* d = my + H4_G3 + tz * H4_T + li * H4_ROW + 8 * lk;      // all 16 pieces of the row: no swizzle needed
#pragma unroll
                for (int k = 0; k < 4; ++k) *reinterpret_cast<uint2*>(d + 32 * k) = z2;
            }
        }
        HQ_CLOSE(hq3);                                // phase 3: exchange writes (inactive groups: their zero rows)
        __syncthreads();
        HQ_CLOSE(hq4);                                // phase 4: first barrier (exchange complete)
        // ---- weight gradients over the 4 groups: 2 K-steps of 32 pixels (slots 2*ks, 2*ks + 1)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const unsigned char* base = ex + 2 * ks * H4_SLOT + t_off;
            auto frag = [&](int tensor, int blk) {           // operand block `blk` (16 channels) of a tensor: two transposed reads
                const int o = (32 * blk) ^ t_sel;
                return hc_pair(hc_tr(base + tensor + o), hc_tr(base + tensor + 4 * H4_ROW + (o ^ 32)));
            };
            {
                const hbf16x8 av = frag(H4_G3, my_mb);
                dB4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, ones8, dB4, 0, 0, 0);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) dW4[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, frag(H4_H2, nb), dW4[nb], 0, 0, 0);
            }
            {
                const hbf16x8 av = frag(H4_G2, my_mb);
                dB2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, ones8, dB2, 0, 0, 0);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) dW2[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, frag(H4_H1, nb), dW2[nb], 0, 0, 0);
            }
            {
                const hbf16x8 av = frag(H4_G1, my_mb);
                dB0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, ones8, dB0, 0, 0, 0);
                const unsigned char* xb = ex + 2 * ks * H4_SLOT;
                dW0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, hc_pair(hc_tr(xb + t_offx), hc_tr(xb + t_offx2)), dW0, 0, 0, 0);
            }
        }
        HQ_CLOSE(hq5);                                // phase 5: weight gradients (transposing reads + MFMAs)
        __syncthreads();
        HQ_CLOSE(hq6);                                // phase 6: second barrier (slots free again)
    }
    HQ_DUMP;

    // ---- workgroup partial (layout of head_bwd_pc_kernel): every 16x16 block has ONE owner; dw6 / db6 are summed over the waves
    // D layout of a block: lane = (m >> 2) * 16 + n, register = m & 3  (m = row = gradient's hidden unit, n = column)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        *reinterpret_cast<f32x4*>(&part[PE_W4 + ((my_mb * 4 + nb) * 64 + lane) * 4]) = dW4[nb];
        *reinterpret_cast<f32x4*>(&part[PE_W2 + ((my_mb * 4 + nb) * 64 + lane) * 4]) = dW2[nb];
    }
    *reinterpret_cast<f32x4*>(&part[PE_W0 + (wave * 64 + lane) * 4]) = dW0;
    // bias gradients: every column of the ones-block holds the row sums; column 0 = lanes li == 0, row m = 4 * lk + register
    if (li == 0) {
        *reinterpret_cast<f32x4*>(&part[PE_B4 + 16 * my_mb + 4 * lk]) = dB4;
        *reinterpret_cast<f32x4*>(&part[PE_B2 + 16 * my_mb + 4 * lk]) = dB2;
        *reinterpret_cast<f32x4*>(&part[PE_B0 + 16 * my_mb + 4 * lk]) = dB0;
    }
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) dw6[mb][r] = lane_sum16(dw6[mb][r]);
    db6 = lane_sum16(db6);
    float* red = reinterpret_cast<float*>(ldsb);
    __syncthreads();
    if (li == 0) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave * 65 + 16 * mb + 4 * lk + r] = dw6[mb][r];
        if (lk == 0) red[wave * 65 + 64] = db6;
    }
    __syncthreads();
    if (tid < 65) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < H4_WAVES; ++w) t += red[w * 65 + tid];
        part[(tid < 64 ? PE_W6 + tid : PE_B6)] = t;
    }
}

// assembles the LDS weight image of the head kernel that follows, once, in global memory (kind 0: fp32 forward, 1: fp32
// backward, 2: bf16 forward, 3: bf16 backward)
// blocks 0..7 assemble image `kind` into img; blocks 8..15 (PC_HEAD_FWD_PACK_BOTH: a 16-block launch) the backward's image of the same
// mode (kind + 1) into img2
// (kind 4: the fp32 forward on 3-way split bf16 operands, head_fwd_split_kernel; kind 5: its backward partner, the split planes of the
// producer waves of head_bwd_pc_kernel<., true>)
__global__ __launch_bounds__(256) void head_pack_kernel(const HeadArgs p, void* img, int kind, void* img2) {
    const int second = blockIdx.x >= 8 ? 1 : 0;
    const int tid = (blockIdx.x - 8 * second) * blockDim.x + threadIdx.x, nt = 8 * blockDim.x;
    if (second) { img = img2; kind = kind == 4 ? 5 : kind + 1; }
    if (kind == 0) head_stage_weights(reinterpret_cast<float*>(img), p, tid, nt);
    else if (kind == 1) head_stage_weights_bwd(reinterpret_cast<float*>(img), p, tid, nt);
    else if (kind == 4) head_stage_weights_split(reinterpret_cast<unsigned char*>(img), p, tid, nt);
    else if (kind == 5) head_stage_weights_bsplit(reinterpret_cast<unsigned char*>(img), p, tid, nt);
    else head_stage_weights_bf16(reinterpret_cast<unsigned char*>(img), p, kind == 3, tid, nt);
}
constexpr int HEAD_IMG_BYTES = 96 * 1024;      // room for the largest image (fp32 backward: LB_SCR floats = 72 KB)
static_assert(LB_SCR * 4 <= HEAD_IMG_BYTES && HB_END <= HEAD_IMG_BYTES && L_END * 4 <= HEAD_IMG_BYTES && HS_END <= HEAD_IMG_BYTES && LS_WEND <= HEAD_IMG_BYTES,
              "weight image slot");
// the images live in the unused tail of the backward partial area of the workspace (pc_head_ws_bytes reserves 512 x 12288
// floats, the backward uses at most 512 x PE_TOTAL = 395 x 12288): slot 0 forward, slot 1 backward
__host__ inline void* head_image_slot(void* ws, int B, int H, int W, int slot) {
    const int groups = (H * W + 15) / 16;
    const int64_t nchunk = (groups + 31) / 32 + 1;
    return reinterpret_cast<char*>(ws) + (int64_t)B * nchunk * 2 * sizeof(float) + (int64_t)490 * 12288 * sizeof(float) + (int64_t)slot * HEAD_IMG_BYTES;
}

struct HeadReduceArgs {
    const float* partial;
    int nwg;
    float* dhw[8];
    int accumulate;
};

// outputs: w0 (1024) b0 (64) w2 (4096) b2 (64) w4 (4096) b4 (64) w6 (128) b6 (2)  = 9538
__global__ __launch_bounds__(256) void head_bwd_reduce_kernel(const HeadReduceArgs p) {
    // 32 CONSECUTIVE partial elements x 8 slices of the partial list per workgroup: the loads of a lane group are one 128-byte
    // line per partial (indexing by output element instead scattered them over the MFMA fragment layout: 16x the sectors); the
    // output index is the inverse of the fragment layout.  The sums are short dependent-latency chains: sized for memory-level
    // parallelism.
    __shared__ float red[256];
    const int tid = threadIdx.x, slice = tid >> 5;
    const int e = blockIdx.x * 32 + (tid & 31);
    int t, idx;                        // tensor id, index within the tensor
    pc_head_partial_target(e, t, idx);
    float s = 0.f;
    if (t >= 0) {
#pragma unroll 8
        for (int w = slice; w < p.nwg; w += 8) s += p.partial[(int64_t)w * PE_TOTAL + e];
    }
    red[tid] = s;
    __syncthreads();
    if (tid < 32 && t >= 0 && p.dhw[t]) {
        const float tot = (((red[tid] + red[32 + tid]) + (red[64 + tid] + red[96 + tid])) +
                           ((red[128 + tid] + red[160 + tid]) + (red[192 + tid] + red[224 + tid])));
        float* d = p.dhw[t] + idx;
        *d = p.accumulate ? *d + tot : tot;
    }
    // structural zeros: the second (unused) output row of the last layer -- weight elements 64..127 and bias element 1
    if (blockIdx.x == 0 && !p.accumulate) {
        if (tid < 64 && p.dhw[6]) p.dhw[6][64 + tid] = 0.f;
        if (tid == 64 && p.dhw[7]) p.dhw[7][1] = 0.f;
    }
}

// ---- fusion_out_conv (1x1, 16->1) + sigmoid + crop ------------------------------------------------------------
__global__ __launch_bounds__(256) void outconv_sigmoid_crop_kernel(pc_src feat, const float* w, const float* bias,
                                                                   pc_dst out, int B, int H, int W, int py, int px, int bf) {
    const int64_t n = (int64_t)B * H * W;
    float wv[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {      // C = 16 (fusion_out_conv) or 8 (sar/optical_out_conv); bf16 mode: operand rounding
        const float t = c < feat.C ? w[c] : 0.f;
        wv[c] = bf ? pc_bf16r(t) : t;
    }
    const float bv = bias[0];
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)n; i += gridDim.x * blockDim.x) {
        const unsigned row = i / (unsigned)W;
        const int x = (int)(i - row * (unsigned)W), y = (int)(row % (unsigned)H), b = (int)(row / (unsigned)H);
        const int64_t fo = b * feat.bstride + (int64_t)(py + y) * feat.rstride + (int64_t)(px + x) * pc_xs(feat);
        float s = bv;
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (c < feat.C) s = fmaf(pc_src_at(feat, fo + c * feat.cstride), wv[c], s);
        out.ptr[b * out.bstride + (int64_t)y * out.rstride + x] = 1.f / (1.f + expf(-s));
    }
}

// ---- sparsity mask ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sparsity_mask_kernel(const float* building, const float* admin, const int64_t* census,
                                                            const uint8_t* rowsel, const uint8_t* colsel, int occ,
                                                            uint8_t* mask, int32_t* counts, int B, int H, int W) {
    const int64_t n = (int64_t)B * H * W;
    int nsel = 0, nreg = 0;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)n; i += gridDim.x * blockDim.x) {
        const unsigned row = i / (unsigned)W;
        const int x = (int)(i - row * (unsigned)W), y = (int)(row % (unsigned)H), b = (int)(row / (unsigned)H);
        const bool region = admin[i] == (float)census[b];
        // popcorn.py:365-372: ((building>0)*region | grid) & region  [occupancymodel]   /   region | grid) & region
        const bool base = occ ? (building[i] > 0.f) : true;
        const bool m = region && (base || (rowsel[y] && colsel[x]));
        mask[i] = m ? 1 : 0;
        nsel += m;
        nreg += region;
    }
    // integer counts: order-independent, atomics are exact.  One atomic pair per BLOCK (a per-wave atomic on two words
    // serialised 16 k atomics: 188 us for a 640 k-pixel batch, profiles/r1_v0).
    __shared__ int red[2][4];
    for (int off = 32; off > 0; off >>= 1) { nsel += __shfl_down(nsel, off); nreg += __shfl_down(nreg, off); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = nsel; red[1][threadIdx.x >> 6] = nreg; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int a = red[0][0] + red[0][1] + red[0][2] + red[0][3], b2 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        if (a) atomicAdd(&counts[0], a);
        if (b2) atomicAdd(&counts[1], b2);
    }
}

// popcorn.py:374-375: an empty selection falls back to the region mask
__global__ __launch_bounds__(256) void sparsity_mask_fallback_kernel(const float* admin, const int64_t* census, uint8_t* mask,
                                                                     int32_t* counts, int B, int H, int W) {
    if (counts[0] != 0) return;
    const int64_t n = (int64_t)B * H * W;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)n; i += gridDim.x * blockDim.x) {
        const int b = (int)(i / (unsigned)(W * H));
        mask[i] = admin[i] == (float)census[b] ? 1 : 0;
    }
}

__global__ void sparsity_mask_fix_count_kernel(int32_t* counts) {
    if (counts[0] == 0) counts[0] = counts[1];
}

// ---- building score + sparsity mask in ONE launch -----------------------------------------------------------------
// outconv_sigmoid_crop_kernel + sparsity_mask_kernel + the empty-selection fallback + the count fix-up (a memset and four
// dependent launches, ~38 us between the U-Net forward and the head) as one kernel: every block accumulates {nsel, nregion}
// into a scratch pair and takes a ticket; the block that draws the last ticket publishes the counts, applies the
// fallback of popcorn.py:374-375 if the whole batch selected nothing (rare; done by that one block).  The accumulators live
// in a library-owned scratch that a one-wave kernel zeroes in front of every launch.
// Zero fills inside the train step are kernels, not hipMemsetAsync: memset nodes captured into the step's HIP graph were
// not reliably re-executed / ordered on replay once the node sequence of the graph changed (a 16- or 32-byte one never
// replayed; with it gone the 67 MB one of the head backward went wrong too: garbage gradients from the second replay on,
// eager launches always correct).
__global__ __launch_bounds__(256) void zero_fill_kernel(float* p, int64_t n4, int64_t rem) {
    const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) reinterpret_cast<f32x4*>(p)[i] = z;
    if (blockIdx.x == 0 && (int64_t)threadIdx.x < rem) p[4 * n4 + threadIdx.x] = 0.f;
}

__global__ void zero_words_kernel(uint32_t* p, int n) {
    if ((int)threadIdx.x < n) p[threadIdx.x] = 0u;
}

struct ScoreMaskArgs {
    pc_src feat; const float* w; const float* bias; pc_dst out;      // 1x1 conv + sigmoid + crop (as outconv_sigmoid_crop)
    const float* admin; const int64_t* census; const uint8_t* rowsel; const uint8_t* colsel;
    int occ; uint8_t* mask; int32_t* counts; unsigned* scratch;      // scratch: one packed 64-bit accumulator {nsel, nregion, ticket} (see the kernel), zero between launches
    int B, H, W, py, px;
    int bf;
};

constexpr int SM_THREADS = 1024;       // 16 waves per block, at most one block per CU: B = 64 tiles (160 k four-pixel items) in ONE round of loads
__global__ __launch_bounds__(SM_THREADS) void score_mask_kernel(const ScoreMaskArgs a) {
    const int64_t n = (int64_t)a.B * a.H * a.W;
    float wv[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const float t = c < a.feat.C ? a.w[c] : 0.f;
        wv[c] = a.bf ? pc_bf16r(t) : t;
    }
    const float bv = a.bias[0];
    int nsel = 0, nreg = 0;
    const bool vec4 = a.feat.dtype == PC_F32 && pc_planar(a.feat) &&      // (a bf16 feature map -- the non-dot fallback of bf16 mode -- takes the scalar loop)
                      (a.W & 3) == 0 && (a.out.rstride & 3) == 0 && (a.out.bstride & 3) == 0 &&
                      ((reinterpret_cast<uintptr_t>(a.out.ptr) | reinterpret_cast<uintptr_t>(a.admin) |
                        reinterpret_cast<uintptr_t>(a.mask)) & 15) == 0;
    if (vec4) {
        // four consecutive pixels of a row per thread: 16-byte accesses (the feature read is 4-byte aligned only: the crop
        // offset px is arbitrary), a quarter of the dependent iterations of the scalar loop
        const unsigned n4 = (unsigned)(n >> 2), w4 = (unsigned)a.W >> 2;
        for (unsigned i4 = blockIdx.x * blockDim.x + threadIdx.x; i4 < n4; i4 += gridDim.x * blockDim.x) {
            const unsigned row = i4 / w4;
            const int x = (int)(i4 - row * w4) * 4, y = (int)(row % (unsigned)a.H), b = (int)(row / (unsigned)a.H);
            const float* fp = a.feat.ptr + b * a.feat.bstride + (int64_t)(a.py + y) * a.feat.rstride + a.px + x;
            f32x4 s = f32x4{bv, bv, bv, bv};
#pragma unroll
            for (int c = 0; c < 16; ++c)
                if (c < a.feat.C) {
                    const f32x4u f = *reinterpret_cast<const f32x4u*>(fp + c * a.feat.cstride);
#pragma unroll
                    for (int e = 0; e < 4; ++e) s[e] = fmaf(f[e], wv[c], s[e]);
                }
            const f32x4 adm = *reinterpret_cast<const f32x4*>(a.admin + 4 * (int64_t)i4);
            const float cid = (float)a.census[b];
            const bool rs = a.rowsel[y] != 0;
            f32x4 bld;
            unsigned mbits = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bld[e] = 1.f / (1.f + expf(-s[e]));
                const bool region = adm[e] == cid;
                const bool base = a.occ ? (bld[e] > 0.f) : true;
                const bool m = region && (base || (rs && a.colsel[x + e]));
                mbits |= (m ? 1u : 0u) << (8 * e);
                nsel += m;
                nreg += region;
            }
            *reinterpret_cast<f32x4*>(a.out.ptr + b * a.out.bstride + (int64_t)y * a.out.rstride + x) = bld;
            *reinterpret_cast<unsigned*>(a.mask + 4 * (int64_t)i4) = mbits;
        }
    } else
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)n; i += gridDim.x * blockDim.x) {
        const unsigned row = i / (unsigned)a.W;
        const int x = (int)(i - row * (unsigned)a.W), y = (int)(row % (unsigned)a.H), b = (int)(row / (unsigned)a.H);
        const int64_t fo = b * a.feat.bstride + (int64_t)(a.py + y) * a.feat.rstride + (int64_t)(a.px + x) * pc_xs(a.feat);
        float s = bv;
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (c < a.feat.C) s = fmaf(pc_src_at(a.feat, fo + c * a.feat.cstride), wv[c], s);
        const float building = 1.f / (1.f + expf(-s));
        a.out.ptr[b * a.out.bstride + (int64_t)y * a.out.rstride + x] = building;
        const bool region = a.admin[i] == (float)a.census[b];
        const bool base = a.occ ? (building > 0.f) : true;
        const bool m = region && (base || (a.rowsel[y] && a.colsel[x]));
        a.mask[i] = m ? 1 : 0;
        nsel += m;
        nreg += region;
    }
    __shared__ int red[2][SM_THREADS / 64];
    __shared__ unsigned last;
    for (int off = 32; off > 0; off >>= 1) { nsel += __shfl_down(nsel, off); nreg += __shfl_down(nreg, off); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = nsel; red[1][threadIdx.x >> 6] = nreg; }
    __syncthreads();
    __shared__ unsigned long long tot_sh;
    if (threadIdx.x == 0) {
        int s0 = 0, s1 = 0;
#pragma unroll
        for (int w = 0; w < SM_THREADS / 64; ++w) { s0 += red[0][w]; s1 += red[1][w]; }
        // integer counts: order-independent, exact.  ONE 64-bit atomic per block carries {nsel : 27 | nregion : 27 | ticket : 10}
        // (device-scope atomics on one address retire at ~13 ns each, and every dependent one is a round trip to the memory side:
        // counts, ticket and the last block's read of the totals were three of them); the block that draws the last ticket has the
        // totals in the returned value
        __threadfence();
        const unsigned long long add = (unsigned long long)(unsigned)s0 | ((unsigned long long)(unsigned)s1 << 27) | (1ull << 54);
        const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long*>(a.scratch), add);
        last = (unsigned)(old >> 54) == gridDim.x - 1 ? 1u : 0u;
        tot_sh = old + add;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    const unsigned tot_sel = (unsigned)(tot_sh & ((1ull << 27) - 1)), tot_reg = (unsigned)((tot_sh >> 27) & ((1ull << 27) - 1));
    if (tot_sel == 0) {
        // an empty selection falls back to the region mask (popcorn.py:374-375)
        for (unsigned i = threadIdx.x; i < (unsigned)n; i += blockDim.x) {
            const int b = (int)(i / (unsigned)(a.W * a.H));
            a.mask[i] = a.admin[i] == (float)a.census[b] ? 1 : 0;
        }
    }
    if (threadIdx.x == 0) {
        a.counts[0] = (int32_t)(tot_sel ? tot_sel : tot_reg);
        a.counts[1] = (int32_t)tot_reg;
        // this block is the last one alive: leave the accumulator and the ticket at zero for the next call (which is ordered
        // behind this kernel on the stream), instead of a zeroing launch in front of every call
        a.scratch[0] = 0u; a.scratch[1] = 0u;
        __threadfence();
    }
}

// ---- ordered compaction: out[rank(i)] = src[i] for mask[i] != 0 (row-major order) --------------------------------
constexpr int CBLK = 1024;   // elements per block

__global__ __launch_bounds__(256) void compact_count_kernel(const uint8_t* mask, int32_t* block_counts, int64_t n) {
    __shared__ int red[4];
    const int64_t base = (int64_t)blockIdx.x * CBLK;
    int c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t i = base + k * 256 + threadIdx.x;
        c += (i < n && mask[i]) ? 1 : 0;
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// single block exclusive scan of block_counts (nblocks <= a few thousand)
__global__ __launch_bounds__(1024) void compact_scan_kernel(int32_t* block_counts, int nblocks, int32_t* n_out) {
    __shared__ int sh[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < nblocks ? block_counts[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int t = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nblocks) block_counts[i] = carry + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += sh[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_out = carry;
}

__global__ __launch_bounds__(256) void compact_write_kernel(const float* src, const uint8_t* mask, const int32_t* block_off,
                                                            float* out, int64_t n) {
    __shared__ int wave_tot[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * CBLK;
    // element order inside a block: k-major (k*256 + tid) keeps the global order row-major
    bool m[4];
    unsigned long long bal[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t i = base + k * 256 + threadIdx.x;
        m[k] = i < n && mask[i];
        bal[k] = __ballot(m[k]);
        if (lane == 0) wave_tot[k][wave] = __popcll(bal[k]);
    }
    __syncthreads();
    int off = block_off[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int pre = 0;
        for (int w = 0; w < wave; ++w) pre += wave_tot[k][w];
        if (m[k]) {
            const int rank = __popcll(bal[k] & ((1ull << lane) - 1ull));
            out[off + pre + rank] = src[base + k * 256 + threadIdx.x];
        }
        off += wave_tot[k][0] + wave_tot[k][1] + wave_tot[k][2] + wave_tot[k][3];
    }
}

// inverse of compact_write_kernel: out[i] = mask[i] ? src[rank(i)] : 0  (autograd of the boolean-index gather, popcorn.py:173)
__global__ __launch_bounds__(256) void scatter_masked_kernel(const float* src, const uint8_t* mask, const int32_t* block_off,
                                                             float* out, int64_t n) {
    __shared__ int wave_tot[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * CBLK;
    bool m[4];
    unsigned long long bal[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t i = base + k * 256 + threadIdx.x;
        m[k] = i < n && mask[i];
        bal[k] = __ballot(m[k]);
        if (lane == 0) wave_tot[k][wave] = __popcll(bal[k]);
    }
    __syncthreads();
    int off = block_off[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int pre = 0;
        for (int w = 0; w < wave; ++w) pre += wave_tot[k][w];
        const int64_t i = base + k * 256 + threadIdx.x;
        if (i < n) out[i] = m[k] ? src[off + pre + __popcll(bal[k] & ((1ull << lane) - 1ull))] : 0.f;
        off += wave_tot[k][0] + wave_tot[k][1] + wave_tot[k][2] + wave_tot[k][3];
    }
}

// get_sparsity_mask(sparse_unet=True), popcorn.py:336-359: one workgroup per sample.
//   bmask = building > thresh;  mask = (bmask | grid) & region;  ratio = #(region & ~bmask) / (#(grid & region & ~bmask) + 1e-5)
__global__ __launch_bounds__(256) void sparsity_mask_unet_kernel(const float* building, const float* admin, const int64_t* census,
                                                                 const uint8_t* rowsel, const uint8_t* colsel, float thresh,
                                                                 uint8_t* mask, float* ratio, int H, int W) {
    __shared__ int red[2][256];
    const int b = blockIdx.x;
    const float cid = (float)census[b];
    const int64_t base = (int64_t)b * H * W;
    int n_empty = 0, n_sub = 0;
    for (int i = threadIdx.x; i < H * W; i += 256) {
        const int y = i / W, x = i - y * W;
        const bool region = admin[base + i] == cid;
        const bool bm = building[base + i] > thresh;
        const bool grid = rowsel[y] && colsel[x];
        mask[base + i] = (uint8_t)((bm || grid) && region);
        n_empty += (region && !bm) ? 1 : 0;
        n_sub += (grid && region && !bm) ? 1 : 0;
    }
    red[0][threadIdx.x] = n_empty;
    red[1][threadIdx.x] = n_sub;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) { red[0][threadIdx.x] += red[0][threadIdx.x + off]; red[1][threadIdx.x] += red[1][threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) ratio[b] = (float)red[0][0] / ((float)red[1][0] + 1e-5f);
}

// F.pad(x, (left, right, top, bottom), mode="reflect") for NCHW planes (add_padding, popcorn.py:231-258)
__global__ __launch_bounds__(256) void reflect_pad_kernel(const float* in, float* out, int64_t planes, int H, int W, int Hp, int Wp,
                                                          int top, int left) {
    const int64_t n = planes * Hp * Wp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % Wp), y = (int)((i / Wp) % Hp);
        const int64_t pl = i / ((int64_t)Wp * Hp);
        out[i] = in[(pl * H + pc_reflect(y - top, H)) * W + pc_reflect(x - left, W)];
    }
}

// the same with a channel gather: out[b][j] = pad(in[b][sel[j]]); one thread = 4 consecutive x of one output row
// NORM: also (x - mean[j]) / std[j] per output plane -- band selection + apply_normalize (utils/utils.py:105-127) + add_padding in
// ONE pass over the raw tile (pc_select_normalize_pad); the division is the one pc_select_normalize performs (same bits)
struct PadSel { int sel[8]; float mean[8]; float stdv[8]; };
template <bool NORM>
__global__ __launch_bounds__(256) void reflect_pad_select_kernel(const float* in, float* out, PadSel ps, int Cin, int nsel, int H, int W,
                                                                 int Hp, int Wp, int top, int left, int nrows, int wq, int rows_per_block) {
    // a block = rows_per_block output rows x wq 4-pixel pieces (wq * rows_per_block <= 256)
    const int tr = threadIdx.x / wq, piece = threadIdx.x - tr * wq;
    const int row = blockIdx.x * rows_per_block + tr;
    if (tr >= rows_per_block || row >= nrows) return;
    const int pl = row / Hp, y = row - pl * Hp;             // pl = b * nsel + j
    const int b = pl / nsel, j = pl - b * nsel;
    const float* src = in + ((int64_t)(b * Cin + ps.sel[j]) * H + pc_reflect(y - top, H)) * W;
    float* dst = out + (int64_t)row * Wp + 4 * piece;
    const int x0 = 4 * piece, xs = x0 - left;
    f32x4 v;
    if (xs >= 0 && xs + 3 < W) {
        if (((xs | W) & 1) == 0) {
            // even pad and even width (14 / 100 for the training tiles): the piece is 8-byte aligned -- two 8-byte loads (the
            // 4-byte-aligned vector type below is split into four dword loads by the compiler)
            typedef float f32x2a __attribute__((ext_vector_type(2)));
            const f32x2a t0 = *reinterpret_cast<const f32x2a*>(src + xs), t1 = *reinterpret_cast<const f32x2a*>(src + xs + 2);
            v = f32x4{t0[0], t0[1], t1[0], t1[1]};
        } else {
            const f32x4u t = *reinterpret_cast<const f32x4u*>(src + xs);   // interior: one (unaligned) 16-byte load
            v = f32x4{t[0], t[1], t[2], t[3]};
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = x0 + e < Wp ? src[pc_reflect(xs + e, W)] : 0.f;
    }
    if (NORM) {
        const float mu = ps.mean[j], sd = ps.stdv[j];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (v[e] - mu) / sd;
    }
    if ((Wp & 3) == 0) {
        *reinterpret_cast<f32x4*>(dst) = v;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (x0 + e < Wp) dst[e] = v[e];
    }
}

// PC_PREC_BF16 ingest (pc_ingest_cl8): one thread = one pixel of the padded domain = ONE aligned 16-byte channels-last slot:
// band select + normalise + reflect padding + stream order + round to bf16; channel slots >= nsel are zero
template <bool NORM>
__global__ __launch_bounds__(256) void ingest_cl8_kernel(const float* __restrict__ in, uint4* __restrict__ out, PadSel ps, int Cin, int nsel,
                                                         int H, int W, int Hp, int Wp, int top, int left, int npix) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    const int x = i % Wp, r = i / Wp;
    const int y = r % Hp, b = r / Hp;
    const int64_t o = (int64_t)pc_reflect(y - top, H) * W + pc_reflect(x - left, W);
    const int64_t plane = (int64_t)H * W;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float t = 0.f;
        if (j < nsel) {
            t = in[((int64_t)b * Cin + ps.sel[j]) * plane + o];
            if (NORM) t = (t - ps.mean[j]) / ps.stdv[j];
        }
        v[j] = t;
    }
    out[i] = make_uint4(pc_pack_bf16(v[0], v[1]), pc_pack_bf16(v[2], v[3]), pc_pack_bf16(v[4], v[5]), pc_pack_bf16(v[6], v[7]));
}

// the same ingest from the two tensors a loader ships (pc_ingest_split): S2 reflectances as UINT16 digital numbers (planar, C2 bands) and
// S1 backscatter as fp32 (planar, C1 bands); channel index sel[j] < C2 -> s2, else s1[sel[j] - C2].  One thread = one pixel of the padded
// domain, all nsel channels: the planar fp32 form writes nsel coalesced words, the channels-last bf16 form one 16-byte slot.
template <bool CL8>
__global__ __launch_bounds__(256) void ingest_split_kernel(const uint16_t* __restrict__ s2, const float* __restrict__ s1, void* __restrict__ out, PadSel ps,
                                                           int C2, int C1, int nsel, int H, int W, int Hp, int Wp, int top, int left, int npix) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    const int x = i % Wp, r = i / Wp;
    const int y = r % Hp, b = r / Hp;
    const int64_t o = (int64_t)pc_reflect(y - top, H) * W + pc_reflect(x - left, W);
    const int64_t plane = (int64_t)H * W;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float t = 0.f;
        if (j < nsel) {
            const int c = ps.sel[j];
            t = c < C2 ? (float)s2[((int64_t)b * C2 + c) * plane + o] : s1[((int64_t)b * C1 + (c - C2)) * plane + o];
            t = (t - ps.mean[j]) / ps.stdv[j];
        }
        v[j] = t;
    }
    if (CL8) {
        reinterpret_cast<uint4*>(out)[i] = make_uint4(pc_pack_bf16(v[0], v[1]), pc_pack_bf16(v[2], v[3]), pc_pack_bf16(v[4], v[5]), pc_pack_bf16(v[6], v[7]));
    } else {
        float* op = reinterpret_cast<float*>(out) + (int64_t)b * nsel * Hp * Wp + (int64_t)y * Wp + x;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < nsel) op[(int64_t)j * Hp * Wp] = v[j];
    }
}


// pc_ingest_pad_strided: the three ingests above for rows of ANY width and an output whose rows are `rs` >= Wp floats apart (the native
// step executor's arena pads rows to 16 bytes): one thread = one 16-byte piece of an output row.  KIND 0: planar fp32 source (model input or
// raw tile), 2: uint16 S2 + fp32 S1 (channel index sel < C2 -> s2).  Same arithmetic as the kernels above ((x - mean) / std), so the same bits.
template <int KIND, bool NORM>
__global__ __launch_bounds__(256) void ingest_pad_strided_kernel(const void* __restrict__ data, const void* __restrict__ data2, float* __restrict__ out, PadSel ps,
                                                                 int Cin, int C2, int nsel, int H, int W, int Hp, int Wp, int rs, int top, int left,
                                                                 int64_t npieces, int wq) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npieces) return;
    const int64_t row = i / wq;
    const int piece = (int)(i - row * wq);
    const int pl = (int)(row / Hp), y = (int)(row - (int64_t)pl * Hp);
    const int b = pl / nsel, j = pl - b * nsel;
    const int ys = pc_reflect(y - top, H);
    const int x0 = 4 * piece, xs = x0 - left;
    const int c = ps.sel[j];
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (KIND == 2 && c < C2) {
        const uint16_t* src = reinterpret_cast<const uint16_t*>(data) + ((int64_t)(b * C2 + c) * H + ys) * W;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (x0 + e < Wp) v[e] = (float)src[pc_reflect(xs + e, W)];
    } else {
        const float* src = KIND == 2 ? reinterpret_cast<const float*>(data2) + ((int64_t)(b * (Cin - C2) + (c - C2)) * H + ys) * W
                                     : reinterpret_cast<const float*>(data) + ((int64_t)(b * Cin + c) * H + ys) * W;
        if (xs >= 0 && xs + 3 < W) {
            const f32x4u t = *reinterpret_cast<const f32x4u*>(src + xs);
            v = f32x4{t[0], t[1], t[2], t[3]};
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (x0 + e < Wp) v[e] = src[pc_reflect(xs + e, W)];
        }
    }
    if (NORM) {
        const float mu = ps.mean[j], sd = ps.stdv[j];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = x0 + e < Wp ? (v[e] - mu) / sd : 0.f;
    }
    *reinterpret_cast<f32x4*>(out + ((int64_t)pl * Hp + y) * rs + x0) = v;      // (rs % 4 == 0: the pad columns of the row get zeros)
}

}  // namespace

static int launch_pad_select(const float* in, float* out, int B, int Cin, int nsel, const int* sel, const float* mean, const float* stdv,
                             int H, int W, int top, int bottom, int left, int right, void* stream) {
    if (!in || !out || !sel || B < 1 || nsel < 1 || nsel > 8 || top >= H || bottom >= H || left >= W || right >= W || top < 0 ||
        bottom < 0 || left < 0 || right < 0)
        return PC_EINVAL;
    PadSel ps{};
    for (int j = 0; j < nsel; ++j) {
        if (sel[j] < 0 || sel[j] >= Cin) return PC_EINVAL;
        ps.sel[j] = sel[j];
        ps.mean[j] = mean ? mean[j] : 0.f;
        ps.stdv[j] = stdv ? stdv[j] : 1.f;
    }
    const int Hp = H + top + bottom, Wp = W + left + right;
    const int64_t nrows = (int64_t)B * nsel * Hp;
    const int wq = (Wp + 3) >> 2;
    if (wq > 256 || nrows > 0x7fffffff) return PC_EINVAL;          // rows of up to 1024 pixels (the training tiles; windows are not padded)
    const int rpb = 256 / wq;
    const dim3 grid((unsigned)((nrows + rpb - 1) / rpb));
    if (mean)
        hipLaunchKernelGGL(reflect_pad_select_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, in, out, ps, Cin, nsel, H, W, Hp, Wp,
                           top, left, (int)nrows, wq, rpb);
    else
        hipLaunchKernelGGL(reflect_pad_select_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, in, out, ps, Cin, nsel, H, W, Hp, Wp,
                           top, left, (int)nrows, wq, rpb);
    PC_CHECK_LAUNCH();
    return 0;
}

extern "C" int pc_reflect_pad_select(const float* in, float* out, int B, int Cin, int nsel, const int* sel, int H, int W, int top,
                                     int bottom, int left, int right, void* stream) {
    return launch_pad_select(in, out, B, Cin, nsel, sel, nullptr, nullptr, H, W, top, bottom, left, right, stream);
}

extern "C" int pc_select_normalize_pad(const float* raw, float* out, int B, int Craw, int nsel, const int* band, const float* mean,
                                       const float* stdv, int H, int W, int top, int bottom, int left, int right, void* stream) {
    if (!mean || !stdv) return PC_EINVAL;
    return launch_pad_select(raw, out, B, Craw, nsel, band, mean, stdv, H, W, top, bottom, left, right, stream);
}

extern "C" int pc_ingest_cl8(const float* raw, void* out, int B, int Craw, int nsel, const int* band, const float* mean, const float* stdv,
                             int H, int W, int top, int bottom, int left, int right, void* stream) {
    if (!raw || !out || !band || B < 1 || nsel < 1 || nsel > 8 || top >= H || bottom >= H || left >= W || right >= W || top < 0 ||
        bottom < 0 || left < 0 || right < 0 || (mean == nullptr) != (stdv == nullptr) || (reinterpret_cast<uintptr_t>(out) & 15))
        return PC_EINVAL;
    PadSel ps{};
    for (int j = 0; j < nsel; ++j) {
        if (band[j] < 0 || band[j] >= Craw) return PC_EINVAL;
        ps.sel[j] = band[j];
        ps.mean[j] = mean ? mean[j] : 0.f;
        ps.stdv[j] = stdv ? stdv[j] : 1.f;
    }
    const int Hp = H + top + bottom, Wp = W + left + right;
    const int64_t npix = (int64_t)B * Hp * Wp;
    if (npix > 0x7fffffff) return PC_EINVAL;
    const dim3 grid((unsigned)((npix + 255) / 256));
    if (mean)
        hipLaunchKernelGGL(ingest_cl8_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, raw, reinterpret_cast<uint4*>(out), ps, Craw, nsel,
                           H, W, Hp, Wp, top, left, (int)npix);
    else
        hipLaunchKernelGGL(ingest_cl8_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, raw, reinterpret_cast<uint4*>(out), ps, Craw, nsel,
                           H, W, Hp, Wp, top, left, (int)npix);
    PC_CHECK_LAUNCH();
    return 0;
}

extern "C" int pc_ingest_split(const uint16_t* s2, int C2, const float* s1, int C1, void* out, int cl8, int B, int nsel, const int* band,
                               const float* mean, const float* stdv, int H, int W, int top, int bottom, int left, int right, void* stream) {
    if (!s2 || !s1 || !out || !band || !mean || !stdv || B < 1 || C2 < 1 || C1 < 1 || nsel < 1 || nsel > 8 || top >= H || bottom >= H ||
        left >= W || right >= W || top < 0 || bottom < 0 || left < 0 || right < 0 || (reinterpret_cast<uintptr_t>(out) & 15))
        return PC_EINVAL;
    PadSel ps{};
    for (int j = 0; j < nsel; ++j) {
        if (band[j] < 0 || band[j] >= C2 + C1) return PC_EINVAL;
        ps.sel[j] = band[j];
        ps.mean[j] = mean[j];
        ps.stdv[j] = stdv[j];
    }
    const int Hp = H + top + bottom, Wp = W + left + right;
    const int64_t npix = (int64_t)B * Hp * Wp;
    if (npix > 0x7fffffff) return PC_EINVAL;
    const dim3 grid((unsigned)((npix + 255) / 256));
    if (cl8)
        hipLaunchKernelGGL(ingest_split_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, s2, s1, out, ps, C2, C1, nsel, H, W, Hp, Wp, top,
                           left, (int)npix);
    else
        hipLaunchKernelGGL(ingest_split_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, s2, s1, out, ps, C2, C1, nsel, H, W, Hp, Wp, top,
                           left, (int)npix);
    PC_CHECK_LAUNCH();
    return 0;
}

extern "C" int pc_ingest_pad_strided(int kind, const void* data, const void* data2, int Cin, float* out, int out_rstride, int B, int nsel,
                                     const int* sel, const float* mean, const float* stdv, int H, int W, int top, int bottom, int left, int right,
                                     void* stream) {
    if (!data || !out || !sel || B < 1 || nsel < 1 || nsel > 8 || top >= H || bottom >= H || left >= W || right >= W || top < 0 || bottom < 0 ||
        left < 0 || right < 0 || (mean == nullptr) != (stdv == nullptr) || (kind == PC_DATA_SPLIT && (!data2 || !mean)) ||
        (kind != PC_DATA_INPUT && kind != PC_DATA_RAW && kind != PC_DATA_SPLIT))
        return PC_EINVAL;
    const int Hp = H + top + bottom, Wp = W + left + right;
    if (out_rstride < Wp || (out_rstride & 3) || (reinterpret_cast<uintptr_t>(out) & 15)) return PC_EINVAL;
    const int C2 = kind == PC_DATA_SPLIT ? 4 : 0;
    const int Ctot = kind == PC_DATA_SPLIT ? 6 : Cin;
    PadSel ps{};
    for (int j = 0; j < nsel; ++j) {
        if (sel[j] < 0 || sel[j] >= Ctot) return PC_EINVAL;
        ps.sel[j] = sel[j];
        ps.mean[j] = mean ? mean[j] : 0.f;
        ps.stdv[j] = stdv ? stdv[j] : 1.f;
    }
    const int wq = (Wp + 3) >> 2;
    const int64_t npieces = (int64_t)B * nsel * Hp * wq;
    const int64_t nblk = (npieces + 255) / 256;
    if (nblk > 0x7fffffff) return PC_EINVAL;
    const dim3 grid((unsigned)nblk);
    hipStream_t st = (hipStream_t)stream;
    if (kind == PC_DATA_SPLIT)
        hipLaunchKernelGGL((ingest_pad_strided_kernel<2, true>), grid, dim3(256), 0, st, data, data2, out, ps, 6, C2, nsel, H, W, Hp, Wp, out_rstride, top,
                           left, npieces, wq);
    else if (mean)
        hipLaunchKernelGGL((ingest_pad_strided_kernel<0, true>), grid, dim3(256), 0, st, data, data2, out, ps, Cin, 0, nsel, H, W, Hp, Wp, out_rstride, top,
                           left, npieces, wq);
    else
        hipLaunchKernelGGL((ingest_pad_strided_kernel<0, false>), grid, dim3(256), 0, st, data, data2, out, ps, Cin, 0, nsel, H, W, Hp, Wp, out_rstride, top,
                           left, npieces, wq);
    PC_CHECK_LAUNCH();
    return 0;
}

// zero fill by a kernel (not a memset node: see zero_fill_kernel); p 16-byte aligned
extern "C" int pc_zero_fill(float* p, int64_t n, void* stream) {
    if (!p || n < 0 || (reinterpret_cast<uintptr_t>(p) & 15)) return PC_EINVAL;
    if (n == 0) return 0;
    const int64_t n4 = n / 4, rem = n - 4 * n4;
    int grid = (int)((n4 + 255) / 256);
    grid = grid < 1 ? 1 : (grid > 2048 ? 2048 : grid);
    hipLaunchKernelGGL(zero_fill_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, n4, rem);
    PC_CHECK_LAUNCH();
    return 0;
}

extern "C" int pc_sparsity_mask_unet(const float* building, const float* admin_mask, const int64_t* census_idx,
                                     const uint8_t* rowsel, const uint8_t* colsel, float threshold, uint8_t* mask, float* ratio,
                                     int B, int H, int W, void* stream) {
    if (!building || !admin_mask || !census_idx || !rowsel || !colsel || !mask || !ratio || B < 1) return PC_EINVAL;
    hipLaunchKernelGGL(sparsity_mask_unet_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, building, admin_mask, census_idx,
                       rowsel, colsel, threshold, mask, ratio, H, W);
    PC_CHECK_LAUNCH();
    return 0;
}

extern "C" int pc_reflect_pad(const float* in, float* out, int64_t planes, int H, int W, int top, int bottom, int left, int right,
                              void* stream) {
    if (!in || !out || top >= H || bottom >= H || left >= W || right >= W || top < 0 || bottom < 0 || left < 0 || right < 0) return PC_EINVAL;
    const int Hp = H + top + bottom, Wp = W + left + right;
    int64_t g = (planes * Hp * Wp + 255) / 256;
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(reflect_pad_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, in, out, planes, H, W, Hp, Wp, top, left);
    PC_CHECK_LAUNCH();
    return 0;
}

extern "C" int pc_scatter_masked(const float* src, const uint8_t* mask, float* out, void* ws, int64_t n, void* stream) {
    if (!src || !mask || !out || !ws) return PC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int nblocks = (int)((n + CBLK - 1) / CBLK);
    if (nblocks == 0) return 0;
    int32_t* bc = reinterpret_cast<int32_t*>(ws);
    hipLaunchKernelGGL(compact_count_kernel, dim3(nblocks), dim3(256), 0, st, mask, bc, n);
    PC_CHECK_LAUNCH();
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, st, bc, nblocks, bc + nblocks);
    PC_CHECK_LAUNCH();
    hipLaunchKernelGGL(scatter_masked_kernel, dim3(nblocks), dim3(256), 0, st, src, mask, bc, out, n);
    PC_CHECK_LAUNCH();
    return 0;
}

extern "C" int64_t pc_head_ws_bytes(int B, int H, int W) {
    const int groups = (H * W + 15) / 16;
    const int64_t nchunk = (groups + 31) / 32 + 1;
    // fwd partials [B][nchunk]; the backward needs workgroup partials of the 9.5k weight gradients
    return (int64_t)B * nchunk * 2 * sizeof(float) + 512 * 12288 * (int64_t)sizeof(float);
}

// One round of workgroups: every workgroup stages the 36 KB of weights, so the launch holds exactly the workgroups that are resident at
// once (B images x chunks-per-image <= resident) and each wave walks enough 16-pixel groups to cover its image chunk -- with a fixed 8
// groups per wave a B = 64 batch of 100 x 100 tiles was 1280 workgroups on 1024 slots, i.e. a second, quarter-full round.
constexpr int HS_NW = 8;              // waves per workgroup of head_fwd_split_kernel (two workgroups per CU: four waves per SIMD)
static int g_head_split = -1;
static int head_split_on() {
    if (g_head_split < 0) {
        const char* ev = getenv("POPCORN_HEAD_SPLIT");
        const char* sr = getenv("POPCORN_HEAD_BWD_SINGLE_ROLE");          // (the ablation's single-role backward is an fp32-MFMA kernel:
        g_head_split = ((ev && ev[0] == '0') || (sr && sr[0] == '1')) ? 0 : 1;      // both head kernels then keep the fp32 images)
    }
    return g_head_split;
}
extern "C" int pc_get_head_split(void) { return head_split_on(); }
extern "C" int pc_set_head_split(int on) {
    const int prev = head_split_on();
    g_head_split = on ? 1 : 0;
    return prev;
}
static int head_fwd_chunks(int B, int H, int W, bool split, int* groups_per_wave, int* nchunk) {
    static int resident[2] = {0, 0};
    static pc_once_per_device once;
    if (once.need()) {
        hipFuncAttributes fa;
        hipError_t e = hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&head_fwd_kernel));
        if (e != hipSuccess) return (int)e;
        resident[0] = pc_resident_workgroups(fa.numRegs, L_END * sizeof(float));
        e = hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&head_fwd_split_kernel<HS_NW>));
        if (e != hipSuccess) return (int)e;
        // pc_resident_workgroups counts 256-thread workgroups: an HS_NW-wave workgroup takes HS_NW / 4 of those register slots
        resident[1] = pc_resident_workgroups(fa.numRegs * (HS_NW / 4), HS_END);
        once.mark();
    }
    const int nw = split ? HS_NW : 4;
    const int groups = (H * W + 15) / 16;
    int chunks = resident[split ? 1 : 0] / (B > 0 ? B : 1);       // chunks per image that fit in one round
    const int max_chunks = (groups + 8 * nw - 1) / (8 * nw);      // never fewer than 8 groups per wave (workspace bound)
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks < 1) chunks = 1;
    int gpw = (groups + nw * chunks - 1) / (nw * chunks);
    if (gpw < 8) gpw = 8;
    *groups_per_wave = gpw;
    *nchunk = (groups + nw * gpw - 1) / (nw * gpw);
    return 0;
}

extern "C" int pc_head_fwd(const pc_src* feat, int py, int px, const float* const* hw, const uint8_t* mask,
                           const float* building, const float* admin_mask, const int64_t* census_idx,
                           float* scale_map, float* popdensemap, float* popcount, double* stats,
                           const int32_t* nsel_counts, void* ws, int B, int H, int W, int flags, void* stream) {
    if (!feat || !hw || !building || !popdensemap || !popcount || !ws) return PC_EINVAL;
    if (admin_mask && !census_idx) return PC_EINVAL;
    HeadArgs p{};
    p.feat = *feat; p.py = py; p.px = px;
    p.w0 = hw[0]; p.b0 = hw[1]; p.w2 = hw[2]; p.b2 = hw[3]; p.w4 = hw[4]; p.b4 = hw[5]; p.w6 = hw[6]; p.b6 = hw[7];
    p.mask = mask; p.building = building; p.admin = admin_mask; p.census = census_idx;
    p.scale_map = scale_map; p.popdense = popdensemap;
    p.partial = reinterpret_cast<float*>(ws);
    p.B = B; p.H = H; p.W = W; p.bf = g_pc_precision == PC_PREC_BF16;
    p.groups = (H * W + 15) / 16;
    p.div_w = pc_make_fastdiv(W);
    p.div_groups = pc_make_fastdiv(p.groups);
    const bool split = !p.bf && head_split_on();
    {
        const int rc = head_fwd_chunks(B, H, W, split, &p.groups_per_wave, &p.nchunk);
        if (rc) return rc;
    }
    hipStream_t st = (hipStream_t)stream;
    // bf16 mode: the feature map is a channels-last bf16 tensor (16 contiguous channels per pixel); fp32 mode: planar fp32
    if (p.bf ? !(pc_cl_ok(*feat) && feat->xstride >= 16) : !(feat->dtype == PC_F32 && pc_planar(*feat))) return PC_EINVAL;
    void* img = head_image_slot(ws, B, H, W, 0);
    p.wimage = img;
    const bool both = (flags & PC_HEAD_FWD_PACK_BOTH) != 0;
    hipLaunchKernelGGL(head_pack_kernel, dim3(both ? 16 : 8), dim3(256), 0, st, p, img, p.bf ? 2 : (split ? 4 : 0), head_image_slot(ws, B, H, W, 1));
    PC_CHECK_LAUNCH();
    if (p.bf) hipLaunchKernelGGL(head_fwd_bf16_kernel, dim3(p.nchunk, B), dim3(256), HB_END, st, p);
    else if (split) hipLaunchKernelGGL(head_fwd_split_kernel<HS_NW>, dim3(p.nchunk, B), dim3(64 * HS_NW), HS_END, st, p);
    else hipLaunchKernelGGL(head_fwd_kernel, dim3(p.nchunk, B), dim3(256), L_END * sizeof(float), st, p);
    PC_CHECK_LAUNCH();
    if (flags & PC_HEAD_FWD_DEFER_REDUCE) return 0;               // pc_head_popcount_loss finishes popcount / stats (and the loss)
    hipLaunchKernelGGL(head_popcount_reduce_kernel, dim3(1), dim3(256), 0, st, p.partial, popcount, B,
                       p.nchunk, stats, nsel_counts, (double)B * H * W);
    PC_CHECK_LAUNCH();
    return 0;
}

// popcount / stats reduction of pc_head_fwd(PC_HEAD_FWD_DEFER_REDUCE) AND the loss forward + backward (pc_loss_fwd_bwd) in ONE
// single-block launch: the two 4.7 us launches of a single-process training step (a data-parallel step all-reduces the stats in between
// and keeps them apart)
__global__ __launch_bounds__(256) void head_popcount_loss_kernel(const float* partial, int nchunk, const int32_t* nsel_counts, double dense_count,
                                                                 float* popcount, double* stats, pc_loss_args a) {
    __shared__ double red[256];
    __shared__ double s_stats[2];
    __shared__ float tsum[256], usum[256];
    const double sc = pc_popcount_sums_block(partial, nchunk, a.B, popcount, tsum, usum);
    // (the same orders as head_popcount_reduce_kernel: per-sample sums by the shared routine, sum(scale) by the same 256-leaf tree)
    red[threadIdx.x] = sc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        s_stats[0] = nsel_counts ? (double)nsel_counts[0] : dense_count;
        s_stats[1] = red[0];
        if (stats) { stats[0] = s_stats[0]; stats[1] = s_stats[1]; }
    }
    __syncthreads();
    pc_loss_block(a, popcount, s_stats, red);
}

extern "C" int pc_head_popcount_loss(void* ws, int B, int H, int W, const int32_t* nsel_counts, const float* y, const float* lam4,
                                     float scale_regularization, float lam_weak, float inv_B, float* popcount, double* stats,
                                     float* loss_out, float* g_popcount, float* g_scale_const, void* stream) {
    if (!ws || !y || !lam4 || !popcount || !loss_out || !g_popcount || !g_scale_const || B < 1) return PC_EINVAL;
    int gpw = 0, nchunk = 0;
    const int rc = head_fwd_chunks(B, H, W, g_pc_precision != PC_PREC_BF16 && head_split_on(), &gpw, &nchunk);      // (as pc_head_fwd chose)
    if (rc) return rc;
    pc_loss_args a{};
    a.y = y;
    for (int i = 0; i < 4; ++i) a.lam[i] = lam4[i];
    a.sreg = scale_regularization; a.lam_weak = lam_weak; a.inv_B = inv_B; a.B = B;
    a.loss_out = loss_out; a.g_popcount = g_popcount; a.g_scale_const = g_scale_const;
    hipLaunchKernelGGL(head_popcount_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float*>(ws), nchunk,
                       nsel_counts, (double)B * H * W, popcount, stats, a);
    PC_CHECK_LAUNCH();
    return 0;
}

extern "C" int pc_outconv_sigmoid_crop(const pc_src* feat, const float* w, const float* bias, const pc_dst* out,
                                       int B, int H, int W, int py, int px, void* stream) {
    if (!feat || !w || !bias || !out || feat->C < 1 || feat->C > 16) return PC_EINVAL;
    const int64_t n = (int64_t)B * H * W;
    int grid = (int)((n + 255) / 256);
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(outconv_sigmoid_crop_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, *feat, w, bias, *out,
                       B, H, W, py, px, (int)(g_pc_precision == PC_PREC_BF16));
    PC_CHECK_LAUNCH();
    return 0;
}

extern "C" int pc_building_score_mask(const pc_src* feat, const float* w, const float* bias, const pc_dst* building_out,
                                      const float* admin_mask, const int64_t* census_idx, const uint8_t* rowsel,
                                      const uint8_t* colsel, int occupancymodel, uint8_t* mask, int32_t* counts,
                                      int B, int H, int W, int py, int px, void* stream) {
    if (!feat || !w || !bias || !building_out || !admin_mask || !census_idx || !rowsel || !colsel || !mask || !counts ||
        feat->C < 1 || feat->C > 16)
        return PC_EINVAL;
    if ((int64_t)B * H * W >= ((int64_t)1 << 27)) return PC_EINVAL;      // the packed 64-bit accumulator holds two 27-bit counts
    static unsigned* scratch = nullptr;     // one 64-bit word {nsel : 27 | nregion : 27 | ticket : 10}: device-scope atomics only; zero
                                            // between calls (the kernel's last block resets it), zeroed once here
    if (!scratch) {
        hipError_t e = hipMalloc(&scratch, 4 * sizeof(unsigned));
        if (e != hipSuccess) return (int)e;
        e = hipMemset(scratch, 0, 4 * sizeof(unsigned));
        if (e != hipSuccess) return (int)e;
        e = hipDeviceSynchronize();            // the first kernel may run on a non-blocking stream
        if (e != hipSuccess) return (int)e;
    }
    ScoreMaskArgs a{};
    a.feat = *feat; a.w = w; a.bias = bias; a.out = *building_out; a.admin = admin_mask; a.census = census_idx;
    a.rowsel = rowsel; a.colsel = colsel; a.occ = occupancymodel; a.mask = mask; a.counts = counts; a.scratch = scratch;
    a.B = B; a.H = H; a.W = W; a.py = py; a.px = px; a.bf = g_pc_precision == PC_PREC_BF16;
    const int64_t n = (int64_t)B * H * W;
    int grid = (int)((n + SM_THREADS - 1) / SM_THREADS);
    if (grid > 256) grid = 256;            // one block per CU: the per-block atomics are the serial part
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(score_mask_kernel, dim3(grid), dim3(SM_THREADS), 0, (hipStream_t)stream, a);
    PC_CHECK_LAUNCH();
    return 0;
}

extern "C" int pc_sparsity_mask(const float* building, const float* admin_mask, const int64_t* census_idx,
                                const uint8_t* rowsel, const uint8_t* colsel, int occupancymodel,
                                uint8_t* mask, int32_t* counts, int B, int H, int W, void* stream) {
    if (!building || !admin_mask || !census_idx || !rowsel || !colsel || !mask || !counts) return PC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(64), 0, st, reinterpret_cast<uint32_t*>(counts), 2);
    PC_CHECK_LAUNCH();
    const int64_t n = (int64_t)B * H * W;
    int grid = (int)((n + 255) / 256);
    if (grid > 512) grid = 512;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(sparsity_mask_kernel, dim3(grid), dim3(256), 0, st, building, admin_mask, census_idx, rowsel, colsel,
                       occupancymodel, mask, counts, B, H, W);
    PC_CHECK_LAUNCH();
    hipLaunchKernelGGL(sparsity_mask_fallback_kernel, dim3(grid), dim3(256), 0, st, admin_mask, census_idx, mask, counts, B, H, W);
    PC_CHECK_LAUNCH();
    hipLaunchKernelGGL(sparsity_mask_fix_count_kernel, dim3(1), dim3(1), 0, st, counts);
    PC_CHECK_LAUNCH();
    return 0;
}

extern "C" int64_t pc_compact_ws_bytes(int64_t n) { return ((n + CBLK - 1) / CBLK + 1) * (int64_t)sizeof(int32_t); }

extern "C" int pc_compact_masked(const float* src, const uint8_t* mask, float* out, int32_t* n_out, void* ws, int64_t n,
                                 void* stream) {
    if (!src || !mask || !out || !n_out || !ws) return PC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int nblocks = (int)((n + CBLK - 1) / CBLK);
    int32_t* bc = reinterpret_cast<int32_t*>(ws);
    if (nblocks == 0) {           // (a kernel, not a memset node: see zero_fill_kernel)
        hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(64), 0, st, reinterpret_cast<uint32_t*>(n_out), 1);
        PC_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(compact_count_kernel, dim3(nblocks), dim3(256), 0, st, mask, bc, n);
    PC_CHECK_LAUNCH();
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, st, bc, nblocks, n_out);
    PC_CHECK_LAUNCH();
    hipLaunchKernelGGL(compact_write_kernel, dim3(nblocks), dim3(256), 0, st, src, mask, bc, out, n);
    PC_CHECK_LAUNCH();
    return 0;
}

// workgroup partials of the backward call: behind the forward's partials in ws; one per workgroup of the kernel the mode launches
// (fp32: one 8-wave workgroup per CU, 4 groups in flight each; bf16: two 4-wave workgroups per CU, 2 x 79 KB of LDS)
static void head_bwd_partial_geometry(void* ws, int B, int H, int W, bool bf, float** partial, int* nwg) {
    const int groups = (H * W + 15) / 16, total_groups = B * groups;
    const int64_t nchunk = (groups + 31) / 32 + 1;
    *partial = reinterpret_cast<float*>(ws) + B * nchunk * 2;
    const int per = bf ? H4_WAVES : 4, cap = bf ? 512 : 256;
    int n = (total_groups + per - 1) / per;
    if (n > cap) n = cap;
    if (n < 1) n = 1;
    *nwg = n;
}
extern "C" int pc_head_bwd_partials(void* ws, int B, int H, int W, const float** partial, int* nwg) {
    if (!ws || !partial || !nwg || B < 1 || H < 1 || W < 1) return PC_EINVAL;
    float* pp;
    head_bwd_partial_geometry(ws, B, H, W, g_pc_precision == PC_PREC_BF16, &pp, nwg);
    *partial = pp;
    return 0;
}

extern "C" int pc_head_bwd(const pc_src* feat, int py, int px, const float* const* hw, const uint8_t* mask,
                           const float* building, const float* admin_mask, const int64_t* census_idx,
                           const float* g_popcount, const float* g_popdense, const float* g_scale_map,
                           const float* g_scale_const, float* const* dhw, int accumulate,
                           const pc_dst* g_feat, const pc_bn* feat_bn_sar, const pc_bn* feat_bn_opt,
                           int Hp, int Wp, void* ws, int B, int H, int W, int flags, void* stream) {
    if (!feat || !hw || !building || !dhw || !g_feat || !ws) return PC_EINVAL;
    if (admin_mask && !census_idx) return PC_EINVAL;
    const bool bfmode = g_pc_precision == PC_PREC_BF16;          // bf16 mode: feat and g_feat are channels-last bf16 tensors
    static int use_pc = -1;
    if (use_pc < 0) {
        const char* ev = getenv("POPCORN_HEAD_BWD_SINGLE_ROLE");
        use_pc = (ev && ev[0] == '1') ? 0 : 1;
    }
    const bool split = !bfmode && use_pc && head_split_on();     // fp32 mode: the producer waves' chain on split bf16 operands
    if (bfmode) {
        if (!pc_cl_ok(*feat) || feat->xstride < 16 || !pc_cl_ok(*g_feat) || g_feat->xstride != 16 || g_feat->rstride != 16 * Wp ||
            g_feat->bstride != (int64_t)16 * Hp * Wp)
            return PC_EINVAL;
    } else {
        if (feat->dtype != PC_F32 || g_feat->dtype != PC_F32 || !pc_planar(*feat) || !pc_planar(*g_feat)) return PC_EINVAL;
        if (g_feat->cstride != (int64_t)Hp * Wp || g_feat->bstride != (int64_t)16 * Hp * Wp || g_feat->rstride != Wp) return PC_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    if ((reinterpret_cast<uintptr_t>(g_feat->ptr) & 15) != 0) return PC_EINVAL;
    const bool zero_launch = !bfmode && !use_pc;                  // (the single-role debug kernel only: the product kernels zero in-kernel)
    if (zero_launch) {
        // zero fill by a kernel, not a memset node (see zero_fill_kernel)
        const int64_t n4 = (int64_t)B * 16 * Hp * Wp / 4, rem = (int64_t)B * 16 * Hp * Wp - 4 * n4;
        hipLaunchKernelGGL(zero_fill_kernel, dim3(2048), dim3(256), 0, st, g_feat->ptr, n4, rem);
        PC_CHECK_LAUNCH();
    }
    HeadBwdArgs a{};
    a.Hp = Hp; a.Wp = Wp; a.zero_in_kernel = zero_launch ? 0 : 1;
    HeadArgs& p = a.f;
    p.feat = *feat; p.py = py; p.px = px;
    p.w0 = hw[0]; p.b0 = hw[1]; p.w2 = hw[2]; p.b2 = hw[3]; p.w4 = hw[4]; p.b4 = hw[5]; p.w6 = hw[6]; p.b6 = hw[7];
    p.mask = mask; p.building = building; p.admin = admin_mask; p.census = census_idx;
    p.B = B; p.H = H; p.W = W; p.bf = g_pc_precision == PC_PREC_BF16;
    p.groups = (H * W + 15) / 16;
    p.div_w = pc_make_fastdiv(W);
    p.div_groups = pc_make_fastdiv(p.groups);
    a.g_popcount = g_popcount; a.g_popdense = g_popdense; a.g_scale_map = g_scale_map; a.g_scale_const = g_scale_const;
    a.g_feat = *g_feat;
    if (feat_bn_sar && feat_bn_opt) { a.fbn[0] = *feat_bn_sar; a.fbn[1] = *feat_bn_opt; a.fuse_feat_bn = 1; }
    a.total_groups = B * p.groups;
    int nwg;
    head_bwd_partial_geometry(ws, B, H, W, p.bf, &a.partial, &nwg);
    {
        const char* dv = getenv("POPCORN_HEAD_DBG");
        a.dbg = dv ? atoi(dv) : 0;
    }
    static pc_once_per_device once;
    if (once.need()) {
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&head_bwd_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LB_END * sizeof(float)));
        if (e2 != hipSuccess) return (int)e2;
        for (const void* f : {reinterpret_cast<const void*>(&head_bwd_pc_kernel<0, false>), reinterpret_cast<const void*>(&head_bwd_pc_kernel<1, false>),
                              reinterpret_cast<const void*>(&head_bwd_pc_kernel<2, false>)}) {
            e2 = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LP_END * sizeof(float)));
            if (e2 != hipSuccess) return (int)e2;
        }
        for (const void* f : {reinterpret_cast<const void*>(&head_bwd_pc_kernel<0, true>), reinterpret_cast<const void*>(&head_bwd_pc_kernel<1, true>),
                              reinterpret_cast<const void*>(&head_bwd_pc_kernel<2, true>)}) {
            e2 = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LPS_END * sizeof(float)));
            if (e2 != hipSuccess) return (int)e2;
        }
        once.mark();
    }
    {
        void* img = head_image_slot(ws, B, H, W, 1);
        p.wimage = img;
        if (!(flags & PC_HEAD_BWD_PACKED)) {
            hipLaunchKernelGGL(head_pack_kernel, dim3(8), dim3(256), 0, st, p, img, p.bf ? 3 : (split ? 5 : 1), (void*)nullptr);
            PC_CHECK_LAUNCH();
        }
    }
    if (p.bf) {
        static pc_once_per_device once4;
        if (once4.need()) {
            hipError_t e5 = hipFuncSetAttribute(reinterpret_cast<const void*>(&head_bwd_bf16_coop4_kernel),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, H4_END);
            if (e5 != hipSuccess) return (int)e5;
            if (getenv("POPCORN_CONV_DBG")) {
                int nb = 0;
                (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, head_bwd_bf16_coop4_kernel, 256, H4_END);
                fprintf(stderr, "head_bwd_bf16_coop4: %d bytes of LDS -> %d workgroups per CU\n", H4_END, nb);
            }
            once4.mark();
        }
        hipLaunchKernelGGL(head_bwd_bf16_coop4_kernel, dim3(nwg), dim3(256), H4_END, st, a);
#ifdef POPCORN_HEAD_PROF
        if (getenv("POPCORN_HEAD_PROF")) {
            static long long hp[2048 * 16];
            (void)hipStreamSynchronize(st);
            (void)hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_head_prof), sizeof(hp));
            double tot[10] = {0};
            const int nw = nwg * 4;
            for (int w = 0; w < nw && w < 2048; ++w) for (int k = 0; k < 10; ++k) tot[k] += (double)hp[w * 16 + k];
            const double nit = (double)((a.total_groups + nw - 1) / nw);
            fprintf(stderr, "head_bwd_bf16_coop4 phases, cycles per iteration and wave (loop top, forward chain, backward chain + store, exchange writes, barrier 1, weight gradients, barrier 2; %d workgroups, %.0f iterations):", nwg, nit);
            for (int k = 0; k < 7; ++k) fprintf(stderr, " %.0f", tot[k] / nw / nit);
            fprintf(stderr, "\n");
        }
#endif
    }
    else if (use_pc && split) {
        if (a.dbg == 1) hipLaunchKernelGGL((head_bwd_pc_kernel<1, true>), dim3(nwg), dim3(512), LPS_END * sizeof(float), st, a);
        else if (a.dbg == 2) hipLaunchKernelGGL((head_bwd_pc_kernel<2, true>), dim3(nwg), dim3(512), LPS_END * sizeof(float), st, a);
        else hipLaunchKernelGGL((head_bwd_pc_kernel<0, true>), dim3(nwg), dim3(512), LPS_END * sizeof(float), st, a);
#ifdef POPCORN_HEAD_PROF
        if (getenv("POPCORN_HEAD_PROF")) {
            static long long hp[2048 * 16];
            (void)hipStreamSynchronize(st);
            (void)hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_head_prof), sizeof(hp));
            double tot[12] = {0};
            const int nw = nwg * 4;
            for (int w = 0; w < nw && w < 2048; ++w) for (int k = 0; k < 12; ++k) tot[k] += (double)hp[w * 16 + k];
            const double ngr = (double)a.total_groups / nw;
            fprintf(stderr, "head_bwd_pc split producer phases, cycles per group (loop top, forward, out+g3, wait0, put0, dgrad3, wait1, put1, dgrad2, wait2, put2, gx+store):");
            for (int k = 0; k < 12; ++k) fprintf(stderr, " %.0f", tot[k] / nw / ngr);
            fprintf(stderr, "\n");
        }
#endif
    }
    else if (use_pc) {
        if (a.dbg == 1) hipLaunchKernelGGL((head_bwd_pc_kernel<1, false>), dim3(nwg), dim3(512), LP_END * sizeof(float), st, a);
        else if (a.dbg == 2) hipLaunchKernelGGL((head_bwd_pc_kernel<2, false>), dim3(nwg), dim3(512), LP_END * sizeof(float), st, a);
        else hipLaunchKernelGGL((head_bwd_pc_kernel<0, false>), dim3(nwg), dim3(512), LP_END * sizeof(float), st, a);
#ifdef POPCORN_HEAD_PROF
        if (getenv("POPCORN_HEAD_PROF")) {
            static long long hp[256 * 16];
            (void)hipStreamSynchronize(st);
            (void)hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_head_prof), sizeof(hp));
            double tot[10] = {0};
            for (int w = 0; w < nwg; ++w) for (int k = 0; k < 10; ++k) tot[k] += (double)hp[w * 16 + k];
            const double ngr = (double)a.total_groups / (nwg * 4);
            fprintf(stderr, "head_bwd_pc producer phases, cycles per group (fwd, out+g3, slot0, dgrad3, slot1, dgrad2, slot2, gx+store, loop top; of which ring waits):");
            for (int k = 0; k < 10; ++k) fprintf(stderr, " %.0f", tot[k] / nwg / ngr);
            fprintf(stderr, "\n");
        }
#endif
    }
    else hipLaunchKernelGGL(head_bwd_kernel, dim3(nwg), dim3(256), LB_END * sizeof(float), st, a);
    PC_CHECK_LAUNCH();
    if (flags & PC_HEAD_BWD_DEFER_REDUCE) return 0;      // the caller's batched reduction finishes the partials (pc_head_bwd_partials)
    HeadReduceArgs r{};
    r.partial = a.partial; r.nwg = nwg; r.accumulate = accumulate;
    for (int i = 0; i < 8; ++i) r.dhw[i] = dhw[i];
    hipLaunchKernelGGL(head_bwd_reduce_kernel, dim3((PE_TOTAL + 31) / 32), dim3(256), 0, st, r);
    PC_CHECK_LAUNCH();
    return 0;
}
