// census.hip -- census-region aggregation (the true scatter-add of the path) and the sliding-window stitcher.
//
// Replaces (reference):
//   data/PopulationDataset.py:675-712  convert_popmap_to_census: a Python loop over every census row doing bbox-crop +
//       `boundary == cidx` mask + sum, O(regions x bbox pixels)                      -> one segment-sum pass
//   data/PopulationDataset.py:823-852  adjust_map_to_census: same loop, then `pred[mask] *= POP20 / sum`  -> one rescale pass
//   run_eval.py:84-154                 per-window D2H + CPU masked `+=` into (h,w) accumulators, then mean / std
//                                      -> accumulators stay on the device
// All HBM-bound integer-indexed streaming kernels: coalesced 16-byte reads, LDS-privatised accumulation, no MFMA.
#include "common.h"

namespace {

// ---- segment sum: sums[id] += pred[i] for boundary[i] == id -------------------------------------------------------------
// fp64 accumulation (LDS atomics per workgroup, then one global atomic per touched id): the result rounded to fp32 is
// independent of the accumulation order for any realistic map, so the output is reproducible although atomics are
// used, and more accurate than the reference's fp32 masked sum.
constexpr int SEG_LDS_IDS = 4096;

__global__ __launch_bounds__(256) void census_sum_kernel(const float* __restrict__ pred, const int32_t* __restrict__ boundary,
                                                         int64_t n, int num_ids, double* sums, int32_t* counts) {
    __shared__ double acc[SEG_LDS_IDS];
    __shared__ int cnt[SEG_LDS_IDS];
    const bool use_lds = num_ids <= SEG_LDS_IDS;
    if (use_lds) {
        for (int i = threadIdx.x; i < num_ids; i += 256) { acc[i] = 0.0; cnt[i] = 0; }
        __syncthreads();
    }
    // any 4-byte aligned band of a larger map (a rank's row band starts at byte r0 * w * 4): a scalar head up to pred's first
    // 16-byte boundary, 16-byte reads from there (boundary too when it happens to be aligned at the same element), scalar tail
    int64_t head = (int64_t)(((16 - (reinterpret_cast<uintptr_t>(pred) & 15)) & 15) >> 2);
    if (head > n) head = n;
    const bool b_vec = ((reinterpret_cast<uintptr_t>(boundary + head)) & 15) == 0;
    const float* pv = pred + head;
    const int32_t* bv = boundary + head;
    const int64_t n4 = (n - head) >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(pv + 4 * i);
        int4 b;
        if (b_vec) b = *reinterpret_cast<const int4*>(bv + 4 * i);
        else { b.x = bv[4 * i]; b.y = bv[4 * i + 1]; b.z = bv[4 * i + 2]; b.w = bv[4 * i + 3]; }
        const int ids[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int id = ids[e];
            if ((unsigned)id < (unsigned)num_ids) {
                if (use_lds) { atomicAdd(&acc[id], (double)p[e]); atomicAdd(&cnt[id], 1); }
                else { atomicAdd(&sums[id], (double)p[e]); if (counts) atomicAdd(&counts[id], 1); }
            }
        }
    }
    // head + tail
    if (blockIdx.x == 0) {
        for (int64_t j = threadIdx.x; j < head + (n - head - 4 * n4); j += 256) {
            const int64_t i = j < head ? j : head + 4 * n4 + (j - head);
            const int id = boundary[i];
            if ((unsigned)id < (unsigned)num_ids) {
                if (use_lds) { atomicAdd(&acc[id], (double)pred[i]); atomicAdd(&cnt[id], 1); }
                else { atomicAdd(&sums[id], (double)pred[i]); if (counts) atomicAdd(&counts[id], 1); }
            }
        }
    }
    if (use_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < num_ids; i += 256) {
            if (cnt[i]) {
                atomicAdd(&sums[i], acc[i]);
                if (counts) atomicAdd(&counts[i], cnt[i]);
            }
        }
    }
}

// ---- dasymetric rescale: pred[i] *= pop[id] / sum[id]  (regions with sum == 0 or no census entry untouched) ------------
__global__ __launch_bounds__(256) void census_adjust_kernel(float* pred, const int32_t* __restrict__ boundary, int64_t n,
                                                            int num_ids, const double* __restrict__ sums,
                                                            const float* __restrict__ pop, const uint8_t* __restrict__ has_entry) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int id = boundary[i];
        if ((unsigned)id < (unsigned)num_ids && (!has_entry || has_entry[id])) {
            const float s = (float)sums[id];          // the reference's fp32 region total (PopulationDataset.py:843)
            if (s != 0.f) pred[i] *= pop[id] / s;     // adj_scale = POP20 / pred_census_count, fp32 (:846-847)
        }
    }
}

// ---- stitcher ------------------------------------------------------------------------------------------------------------
// One window of an M-member ensemble: interior pixels (overlap border excluded, PopulationDataset.py:656-672) of
//   out_sum  += sum_m pd_m      out_sq  += sum_m pd_m^2      (same for scale)      count += M        (run_eval.py:108-135)
struct StitchArgs {
    const float* pd; const float* sc;       // [M][ps_y][ps_x] (sc may be NULL)
    float* out_sum; float* out_sq; float* sc_sum; float* sc_sq; int16_t* count;
    int M, psy, psx, overlap, yl, xl, H, W;
};

__global__ __launch_bounds__(256) void stitch_accumulate_kernel(const StitchArgs a) {
    const int iy = a.psy - 2 * a.overlap, ix = a.psx - 2 * a.overlap;
    const int n = iy * ix;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int ry = i / ix, rx = i - ry * ix;
        const int py = a.overlap + ry, px = a.overlap + rx;
        const int gy = a.yl + py, gx = a.xl + px;
        if (gy >= a.H || gx >= a.W) continue;
        float s = 0.f, s2 = 0.f, t = 0.f, t2 = 0.f;
        for (int m = 0; m < a.M; ++m) {
            const float v = a.pd[((int64_t)m * a.psy + py) * a.psx + px];
            s += v; s2 += v * v;
            if (a.sc) { const float w = a.sc[((int64_t)m * a.psy + py) * a.psx + px]; t += w; t2 += w * w; }
        }
        const int64_t o = (int64_t)gy * a.W + gx;
        a.out_sum[o] += s;
        a.out_sq[o] += s2;
        if (a.sc) { a.sc_sum[o] += t; a.sc_sq[o] += t2; }
        a.count[o] = (int16_t)(a.count[o] + a.M);
    }
}

// run_eval.py:140-154: where count > 1: mean = sum / n;  std = sqrt((sq - mean^2 * n) / (n - 1)); elsewhere untouched
__global__ __launch_bounds__(256) void stitch_finalize_kernel(float* out_sum, float* out_sq, float* sc_sum, float* sc_sq,
                                                              const int16_t* count, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = count[i];
        if (c > 1) {
            const float fc = (float)c;
            const float m = out_sum[i] / fc;
            out_sum[i] = m;
            out_sq[i] = sqrtf((out_sq[i] - (m * m) * fc) / (float)(c - 1));
            if (sc_sum) {
                const float ms = sc_sum[i] / fc;
                sc_sum[i] = ms;
                sc_sq[i] = sqrtf((sc_sq[i] - (ms * ms) * fc) / (float)(c - 1));
            }
        }
    }
}

// zero fill as a kernel (memset nodes were unreliable under HIP-graph replay on this stack)
__global__ __launch_bounds__(256) void census_zero_kernel(double* sums, int32_t* counts, int n) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        sums[i] = 0.0;
        if (counts) counts[i] = 0;
    }
}

// visit counts of MANY windows in one pass (windows other ranks computed: only their count, no data).  One workgroup per (row, 1024-column
// segment): the windows whose interior covers the row are collected into LDS once (wave-uniform test), every thread then tests its four
// columns against that short list -- no atomics on the count map, no scratch plane, any overlap between windows (catch-up windows).
constexpr int CW_CAP = 1024;
__global__ __launch_bounds__(256) void stitch_count_windows_kernel(const int32_t* __restrict__ win, int nwin, int M, int16_t* __restrict__ count,
                                                                   int H, int W) {
    __shared__ int ly0[CW_CAP], ly1[CW_CAP];
    __shared__ int ln;
    const int c0 = blockIdx.x * 1024 + threadIdx.x * 4;
    for (int r = blockIdx.y; r < H; r += gridDim.y) {
    int acc[4] = {0, 0, 0, 0};
    for (int base = 0; base < nwin; base += CW_CAP) {
        if (threadIdx.x == 0) ln = 0;
        __syncthreads();
        for (int k = base + threadIdx.x; k < nwin && k < base + CW_CAP; k += 256) {
            const int x0 = win[4 * k], x1 = win[4 * k + 1], y0 = win[4 * k + 2], y1 = win[4 * k + 3];
            if (x0 <= r && r < x1 && y1 > y0) {
                const int slot = atomicAdd(&ln, 1);
                ly0[slot] = y0; ly1[slot] = y1;
            }
        }
        __syncthreads();
        const int n = ln;
        for (int k = 0; k < n; ++k) {
            const int y0 = ly0[k], y1 = ly1[k];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += (y0 <= c0 + e && c0 + e < y1) ? 1 : 0;
        }
        __syncthreads();
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (c0 + e < W && acc[e]) {
            int16_t* p = count + (int64_t)r * W + c0 + e;
            *p = (int16_t)(*p + M * acc[e]);
        }
    }
}

int stream_grid(int64_t n, int per_thread = 4) {
    int64_t g = (n / per_thread + 255) / 256;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" int pc_census_sum(const float* pred, const int32_t* boundary, int64_t n, int num_ids, double* sums,
                             int32_t* counts, void* stream) {
    if (!pred || !boundary || !sums || num_ids <= 0) return PC_EINVAL;
    if ((reinterpret_cast<uintptr_t>(pred) & 3) || (reinterpret_cast<uintptr_t>(boundary) & 3)) return PC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(census_zero_kernel, dim3((num_ids + 255) / 256 > 64 ? 64 : (num_ids + 255) / 256), dim3(256), 0, st, sums, counts, num_ids);
    PC_CHECK_LAUNCH();
    int grid = stream_grid(n, 16);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(census_sum_kernel, dim3(grid), dim3(256), 0, st, pred, boundary, n, num_ids, sums, counts);
    PC_CHECK_LAUNCH();
    return 0;
}

extern "C" int pc_census_adjust(float* pred, const int32_t* boundary, int64_t n, int num_ids, const double* sums,
                                const float* pop, const uint8_t* has_entry, void* stream) {
    if (!pred || !boundary || !sums || !pop || num_ids <= 0) return PC_EINVAL;
    hipLaunchKernelGGL(census_adjust_kernel, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, pred, boundary, n,
                       num_ids, sums, pop, has_entry);
    PC_CHECK_LAUNCH();
    return 0;
}

extern "C" int pc_stitch_accumulate(const float* popdense, const float* scale, int M, int ps_y, int ps_x, int overlap,
                                    int yl, int xl, float* out_sum, float* out_sq, float* scale_sum, float* scale_sq,
                                    int16_t* count, int H, int W, void* stream) {
    if (!popdense || !out_sum || !out_sq || !count || M < 1) return PC_EINVAL;
    if (scale && (!scale_sum || !scale_sq)) return PC_EINVAL;
    if (ps_y <= 2 * overlap || ps_x <= 2 * overlap) return 0;
    StitchArgs a{};
    a.pd = popdense; a.sc = scale; a.out_sum = out_sum; a.out_sq = out_sq; a.sc_sum = scale_sum; a.sc_sq = scale_sq;
    a.count = count; a.M = M; a.psy = ps_y; a.psx = ps_x; a.overlap = overlap; a.yl = yl; a.xl = xl; a.H = H; a.W = W;
    const int64_t n = (int64_t)(ps_y - 2 * overlap) * (ps_x - 2 * overlap);
    hipLaunchKernelGGL(stitch_accumulate_kernel, dim3(stream_grid(n, 2)), dim3(256), 0, (hipStream_t)stream, a);
    PC_CHECK_LAUNCH();
    return 0;
}

extern "C" int pc_stitch_finalize(float* out_sum, float* out_sq, float* scale_sum, float* scale_sq, const int16_t* count,
                                  int64_t n, void* stream) {
    if (!out_sum || !out_sq || !count) return PC_EINVAL;
    hipLaunchKernelGGL(stitch_finalize_kernel, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, out_sum, out_sq,
                       scale_sum, scale_sq, count, n);
    PC_CHECK_LAUNCH();
    return 0;
}

extern "C" int pc_stitch_count_windows(const int32_t* win, int nwin, int M, int16_t* count, int H, int W, void* stream) {
    if (!win || !count || nwin < 0 || H < 1 || W < 1) return PC_EINVAL;
    if (nwin == 0) return 0;
    hipLaunchKernelGGL(stitch_count_windows_kernel, dim3((W + 1023) / 1024, H < 65535 ? H : 65535), dim3(256), 0, (hipStream_t)stream, win, nwin, M, count, H, W);
    PC_CHECK_LAUNCH();
    return 0;
}
