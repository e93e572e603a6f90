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
    float* partial;        // [B][nchunk]
    int B, H, W;
    int groups, nchunk, groups_per_wave;
};

// Fill the LDS weight image.  Fragment (lane = k*16 + i) of k-step ks, m-block mb:  W[16*mb + i][col(ks, k)]
__device__ __forceinline__ void head_stage_weights(float* lds, const HeadArgs& p) {
    const int tid = threadIdx.x;
    for (int e = tid; e < 16 * 64; e += blockDim.x) {
        const int lane = e & 63, f = e >> 6, ks = f & 3, mb = f >> 2;
        lds[L_A1 + e] = p.w0[(16 * mb + (lane & 15)) * 16 + 4 * ks + (lane >> 4)];
    }
    for (int e = tid; e < 64 * 64; e += blockDim.x) {
        const int lane = e & 63, f = e >> 6, ks = f & 15, mb2 = f >> 4;
        // k-step ks = (mb, r): hidden input index 16*mb + 4*k + r
        const int col = 16 * (ks >> 2) + 4 * (lane >> 4) + (ks & 3);
        const int row = 16 * mb2 + (lane & 15);
        lds[L_A2 + e] = p.w2[row * HID + col];
        lds[L_A3 + e] = p.w4[row * HID + col];
    }
    for (int e = tid; e < 64; e += blockDim.x) {
        lds[L_B0 + e] = p.b0[e];
        lds[L_B2 + e] = p.b2[e];
        lds[L_B4 + e] = p.b4[e];
        lds[L_W6 + e] = p.w6[e];          // row 0 of the [2][64] last layer: only channel 0 is used (popcorn.py:162,164)
    }
    if (tid == 0) lds[L_W6 + 64] = p.b6[0];
}

// One 64-wide layer: acc[mb2] = bias + W * h  (h in D layout of the previous layer), ReLU applied by the caller.
__device__ __forceinline__ void head_layer64(const float* lds, int a_off, int b_off, int lane, int lk,
                                             const f32x4 (&h)[4], f32x4 (&acc)[4]) {
#pragma unroll
    for (int mb2 = 0; mb2 < 4; ++mb2) acc[mb2] = *reinterpret_cast<const f32x4*>(&lds[b_off + 16 * mb2 + 4 * lk]);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ks = mb * 4 + r;
#pragma unroll
            for (int mb2 = 0; mb2 < 4; ++mb2)
                acc[mb2] = __builtin_amdgcn_mfma_f32_16x16x4f32(lds[a_off + (mb2 * 16 + ks) * 64 + lane], h[mb][r],
                                                                acc[mb2], 0, 0, 0);
        }
}

__device__ __forceinline__ void relu4(f32x4 (&h)[4]) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) h[mb][r] = fmaxf(h[mb][r], 0.f);
}

__global__ __launch_bounds__(256) void head_fwd_kernel(const HeadArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    head_stage_weights(lds, p);
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int b = blockIdx.y;
    const int HW = p.H * p.W;
    const float cid = p.census ? (float)p.census[b] : 0.f;
    float pc_sum = 0.f;

    const int g_begin = (blockIdx.x * 4 + wave) * p.groups_per_wave;
    int g_end = g_begin + p.groups_per_wave;
    if (g_end > p.groups) g_end = p.groups;
    for (int g = g_begin; g < g_end; ++g) {
        const int q = g * 16 + li;
        const bool valid = q < HW;
        const int64_t pix = (int64_t)b * HW + q;
        const bool sel = valid && (p.mask ? p.mask[pix] != 0 : true);
        float outv = 0.f;
        if (__any(sel)) {
            const int y = valid ? q / p.W : 0, x = valid ? q - (q / p.W) * p.W : 0;
            const float* fp = p.feat.ptr + b * p.feat.bstride + (int64_t)(p.py + y) * p.feat.rstride + p.px + x;
            float xv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) xv[j] = valid ? fp[(4 * j + lk) * p.feat.cstride] : 0.f;
            f32x4 h[4], acc[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) h[mb] = *reinterpret_cast<const f32x4*>(&lds[L_B0 + 16 * mb + 4 * lk]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
                    h[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(lds[L_A1 + (mb * 4 + j) * 64 + lane], xv[j], h[mb], 0, 0, 0);
            relu4(h);
            head_layer64(lds, L_A2, L_B2, lane, lk, h, acc);
            relu4(acc);
            head_layer64(lds, L_A3, L_B4, lane, lk, acc, h);
            relu4(h);
            float s = 0.f;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(&lds[L_W6 + 16 * mb + 4 * lk]);
#pragma unroll
                for (int r = 0; r < 4; ++r) s = fmaf(w[r], h[mb][r], s);
            }
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            outv = sel ? s + lds[L_W6 + 64] : 0.f;
        }
        if (valid && lk == 0) {
            const float scale = fmaxf(outv, 0.f);
            const float pd = scale * p.building[pix];
            if (p.scale_map) p.scale_map[pix] = scale;
            p.popdense[pix] = pd;
            const bool region = p.admin ? (p.admin[pix] == cid) : true;
            pc_sum += region ? pd : 0.f;
        }
    }
    // deterministic per-workgroup partial: lanes 0..15 of each wave hold the sums
    __syncthreads();
    float* red = lds;   // weights no longer needed
    if (lk == 0) red[wave * 16 + li] = pc_sum;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int i = 0; i < 64; ++i) t += red[i];
        p.partial[(int64_t)b * p.nchunk + blockIdx.x] = t;
    }
}

__global__ void head_popcount_reduce_kernel(const float* partial, float* popcount, int B, int nchunk) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float t = 0.f;
    for (int c = 0; c < nchunk; ++c) t += partial[(int64_t)b * nchunk + c];
    popcount[b] = t;
}

// ---- fusion_out_conv (1x1, 16->1) + sigmoid + crop ------------------------------------------------------------
__global__ __launch_bounds__(256) void outconv_sigmoid_crop_kernel(pc_src feat, const float* w, const float* bias,
                                                                   pc_dst out, int B, int H, int W, int py, int px) {
    const int64_t n = (int64_t)B * H * W;
    float wv[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) wv[c] = w[c];
    const float bv = bias[0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H), b = (int)(i / ((int64_t)W * H));
        const float* fp = feat.ptr + b * feat.bstride + (int64_t)(py + y) * feat.rstride + px + x;
        float s = bv;
#pragma unroll
        for (int c = 0; c < 16; ++c) s = fmaf(fp[c * feat.cstride], wv[c], s);
        out.ptr[b * out.bstride + (int64_t)y * out.rstride + x] = 1.f / (1.f + expf(-s));
    }
}

// ---- sparsity mask ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sparsity_mask_kernel(const float* building, const float* admin, const int64_t* census,
                                                            const uint8_t* rowsel, const uint8_t* colsel, int occ,
                                                            uint8_t* mask, int32_t* counts, int B, int H, int W) {
    const int64_t n = (int64_t)B * H * W;
    int nsel = 0, nreg = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H), b = (int)(i / ((int64_t)W * H));
        const bool region = admin[i] == (float)census[b];
        // popcorn.py:365-372: ((building>0)*region | grid) & region  [occupancymodel]   /   region | grid) & region
        const bool base = occ ? (building[i] > 0.f) : true;
        const bool m = region && (base || (rowsel[y] && colsel[x]));
        mask[i] = m ? 1 : 0;
        nsel += m;
        nreg += region;
    }
    // integer counts: order-independent, atomics are exact
    for (int off = 32; off > 0; off >>= 1) { nsel += __shfl_down(nsel, off); nreg += __shfl_down(nreg, off); }
    if ((threadIdx.x & 63) == 0) {
        if (nsel) atomicAdd(&counts[0], nsel);
        if (nreg) atomicAdd(&counts[1], nreg);
    }
}

// popcorn.py:374-375: an empty selection falls back to the region mask
__global__ __launch_bounds__(256) void sparsity_mask_fallback_kernel(const float* admin, const int64_t* census, uint8_t* mask,
                                                                     int32_t* counts, int B, int H, int W) {
    if (counts[0] != 0) return;
    const int64_t n = (int64_t)B * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / ((int64_t)W * H));
        mask[i] = admin[i] == (float)census[b] ? 1 : 0;
    }
}

__global__ void sparsity_mask_fix_count_kernel(int32_t* counts) {
    if (counts[0] == 0) counts[0] = counts[1];
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

}  // namespace

extern "C" int64_t pc_head_ws_bytes(int B, int H, int W) {
    const int groups = (H * W + 15) / 16;
    const int64_t nchunk = (groups + 31) / 32 + 1;
    // fwd partials [B][nchunk]; the backward needs workgroup partials of the 9.5k weight gradients
    return (int64_t)B * nchunk * sizeof(float) + 512 * 12288 * (int64_t)sizeof(float);
}

extern "C" int pc_head_fwd(const pc_src* feat, int py, int px, const float* const* hw, const uint8_t* mask,
                           const float* building, const float* admin_mask, const int64_t* census_idx,
                           float* scale_map, float* popdensemap, float* popcount, void* ws,
                           int B, int H, int W, void* stream) {
    if (!feat || !hw || !building || !popdensemap || !popcount || !ws) return PC_EINVAL;
    if (admin_mask && !census_idx) return PC_EINVAL;
    HeadArgs p{};
    p.feat = *feat; p.py = py; p.px = px;
    p.w0 = hw[0]; p.b0 = hw[1]; p.w2 = hw[2]; p.b2 = hw[3]; p.w4 = hw[4]; p.b4 = hw[5]; p.w6 = hw[6]; p.b6 = hw[7];
    p.mask = mask; p.building = building; p.admin = admin_mask; p.census = census_idx;
    p.scale_map = scale_map; p.popdense = popdensemap;
    p.partial = reinterpret_cast<float*>(ws);
    p.B = B; p.H = H; p.W = W;
    p.groups = (H * W + 15) / 16;
    p.groups_per_wave = 8;
    p.nchunk = (p.groups + 4 * p.groups_per_wave - 1) / (4 * p.groups_per_wave);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(head_fwd_kernel, dim3(p.nchunk, B), dim3(256), L_END * sizeof(float), st, p);
    PC_CHECK_LAUNCH();
    hipLaunchKernelGGL(head_popcount_reduce_kernel, dim3((B + 63) / 64), dim3(64), 0, st, p.partial, popcount, B, p.nchunk);
    PC_CHECK_LAUNCH();
    return 0;
}

extern "C" int pc_outconv_sigmoid_crop(const pc_src* feat, const float* w, const float* bias, const pc_dst* out,
                                       int B, int H, int W, int py, int px, void* stream) {
    if (!feat || !w || !bias || !out || feat->C != 16) return PC_EINVAL;
    const int64_t n = (int64_t)B * H * W;
    int grid = (int)((n + 255) / 256);
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(outconv_sigmoid_crop_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, *feat, w, bias, *out,
                       B, H, W, py, px);
    PC_CHECK_LAUNCH();
    return 0;
}

extern "C" int pc_sparsity_mask(const float* building, const float* admin_mask, const int64_t* census_idx,
                                const uint8_t* rowsel, const uint8_t* colsel, int occupancymodel,
                                uint8_t* mask, int32_t* counts, int B, int H, int W, void* stream) {
    if (!building || !admin_mask || !census_idx || !rowsel || !colsel || !mask || !counts) return PC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(counts, 0, 2 * sizeof(int32_t), st);
    if (e != hipSuccess) return (int)e;
    const int64_t n = (int64_t)B * H * W;
    int grid = (int)((n + 255) / 256);
    if (grid > 2048) grid = 2048;
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
    if (nblocks == 0) return (int)hipMemsetAsync(n_out, 0, sizeof(int32_t), st);
    hipLaunchKernelGGL(compact_count_kernel, dim3(nblocks), dim3(256), 0, st, mask, bc, n);
    PC_CHECK_LAUNCH();
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, st, bc, nblocks, n_out);
    PC_CHECK_LAUNCH();
    hipLaunchKernelGGL(compact_write_kernel, dim3(nblocks), dim3(256), 0, st, src, mask, bc, out, n);
    PC_CHECK_LAUNCH();
    return 0;
}
