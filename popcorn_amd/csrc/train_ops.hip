// train_ops.hip -- the O(batch) / O(parameters) kernels of a training step: loss forward+backward, global gradient
// norm, fused clip + Adam, and the loader-side band select + normalise.
//
// Replaces (reference): utils/losses.py:49-76 (population loss + scale regulariser) and its autograd;
// torch.nn.utils.clip_grad_norm_ + optim.Adam (run_train.py:82-90,233-238); utils/utils.py:105-127 (apply_normalize)
// + the band selection of data/PopulationDataset.py:566-568.
#include "common.h"

namespace {

// ---- loss ---------------------------------------------------------------------------------------------------------
// loss = sum_k lam[k] * L_k(popcount, y) + sreg * sum(scale) / Nsel, with L_k in {l1, log_l1, mse, log_mse} taken as
// a mean over the GLOBAL batch (inv_B = 1 / (world * B)), so that summing every rank's gradient reproduces the
// single-process gradient exactly.  Outputs d(lam_weak * loss)/d popcount[b] and the constant
// d(lam_weak * loss)/d scale on selected pixels.
__global__ __launch_bounds__(256) void loss_fwd_bwd_kernel(const pc_loss_args a, const float* popcount, const double* stats) {
    __shared__ double red[256];
    pc_loss_block(a, popcount, stats, red);
}

// ---- gradient norm (deterministic two-level tree over a flat buffer) ---------------------------------------------------
__global__ __launch_bounds__(1024) void grad_norm_kernel(const float* g, int n, float* norm_out) {
    __shared__ double red[1024];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) acc += (double)g[i] * (double)g[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) *norm_out = (float)sqrt(red[0]);
}

// ---- clip + Adam over the flat parameter buffer --------------------------------------------------------------------
// torch.optim.Adam semantics (L2 weight decay added to the gradient, bias-corrected, eps outside the sqrt), with the
// clip_grad_norm_ coefficient min(1, max_norm / (norm + 1e-6)) folded in.  Elements [0, n_decay) get weight decay, the
// tail (head.6.*) does not (run_train.py:82-90).  hyper (device): {lr}.  step (device int32) is incremented here so a
// captured graph advances the bias correction on every replay.
// Parameter groups (pc_adam_groups): the reference leaves `.grad = None` on the encoder (limit1) or the whole U-Net
// (limit2) for large samples (run_train.py:191-198, networks.py:124-132), and torch.optim.Adam then SKIPS those
// parameters: no weight decay, no moment update, no per-parameter step increment.  Elements of an inactive group are left
// untouched here, and every group keeps its own step counter (torch's per-parameter `step`), so bias correction of a group
// only advances on the steps that updated it.
struct AdamArgs {
    float* p; const float* g; float* m; float* v;
    int n, n_decay;
    const float* hyper; float wd, beta1, beta2, eps, max_norm;
    const float* norm; int32_t* step;
    pc_adam_groups grp;        // nseg == 0: one group, always active, counter step[0]
};

__device__ __forceinline__ int adam_group_of(const AdamArgs& a, int i) {
    int g = 0;
#pragma unroll
    for (int s = PC_ADAM_MAX_SEG - 1; s >= 0; --s)
        if (s < a.grp.nseg && i < a.grp.seg_end[s]) g = a.grp.seg_group[s];
    return g;
}

// per-group {step size, sqrt(bias correction 2)} into shared memory; thread t < PC_ADAM_GROUPS handles group t
__device__ __forceinline__ void adam_group_constants(const AdamArgs& a, float (*sh)[2], int tid) {
    if (tid < (a.grp.nseg ? PC_ADAM_GROUPS : 1)) {
        const int t = a.step[tid] + 1;
        const double bc1 = 1.0 - pow((double)a.beta1, (double)t);
        const double bc2 = 1.0 - pow((double)a.beta2, (double)t);
        sh[tid][0] = (float)((double)a.hyper[0] / bc1);          // step size
        sh[tid][1] = (float)sqrt(bc2);
    }
}

__device__ __forceinline__ void adam_update_one(const AdamArgs& a, int i, float coef, const float (*sh)[2]) {
    const int grp = a.grp.nseg ? adam_group_of(a, i) : 0;
    if (a.grp.nseg && !((a.grp.active_mask >> grp) & 1)) return;      // grad is None: parameter and state untouched
    float g = a.g[i] * coef;
    const float p = a.p[i];
    if (i < a.n_decay && a.wd != 0.f) g = fmaf(a.wd, p, g);
    const float m = a.beta1 * a.m[i] + (1.f - a.beta1) * g;
    const float v = a.beta2 * a.v[i] + (1.f - a.beta2) * g * g;
    a.m[i] = m;
    a.v[i] = v;
    const float denom = sqrtf(v) / sh[grp][1] + a.eps;
    a.p[i] = p - sh[grp][0] * (m / denom);
}

__global__ __launch_bounds__(256) void adam_clip_kernel(const AdamArgs a) {
    __shared__ float sh[PC_ADAM_GROUPS][2];
    adam_group_constants(a, sh, threadIdx.x);
    __syncthreads();
    float coef = 1.f;
    if (a.max_norm > 0.f && a.norm) {
        coef = a.max_norm / (*a.norm + 1e-6f);
        coef = coef > 1.f ? 1.f : coef;
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < a.n) adam_update_one(a, i, coef, sh);
}

__global__ void adam_step_inc_kernel(int32_t* step, int ngroups, int active_mask) {
    if ((int)threadIdx.x < ngroups && ((active_mask >> threadIdx.x) & 1)) step[threadIdx.x] += 1;
}

// One-launch variant: every workgroup first computes the total gradient norm itself (the flat buffer is ~157 KB and sits
// in L2; all workgroups add in the same order, so they all get the same bits), then updates its 256 elements; the
// workgroup that finishes last advances the step counter (every workgroup has read it by then).  Replaces the
// single-workgroup norm kernel + the update + the one-thread counter kernel (3 dependent launches at the end of a step).
constexpr int ADAM_NT = 1024;
__global__ __launch_bounds__(ADAM_NT) void adam_clip_fused_kernel(const AdamArgs a, float* norm_out, unsigned* ticket) {
    // 39 k parameters: the kernel is a chain of memory round trips, not work.  1024-thread workgroups (every workgroup reads the
    // whole gradient for the norm: 10 16-byte loads per thread in two batches instead of 39 in five, and 39 workgroups re-read it
    // instead of 154); the thread's own parameter / state / gradient element and the per-group bias corrections (two double pow()
    // on three threads) are in flight UNDER the norm phase instead of behind it; one barrier instead of an 8-step tree.
    __shared__ double wred[ADAM_NT / 64];
    __shared__ float sh[PC_ADAM_GROUPS][2];
    const int tid = threadIdx.x;
    const int i = blockIdx.x * ADAM_NT + tid;
    int grp = 0;
    bool act = i < a.n;
    if (act && a.grp.nseg) {
        grp = adam_group_of(a, i);
        act = (a.grp.active_mask >> grp) & 1;                 // grad is None: parameter and state untouched
    }
    const int ic = act ? i : 0;
    const float g0 = a.g[ic], p0 = a.p[ic], m0 = a.m[ic], v0 = a.v[ic];
    adam_group_constants(a, sh, tid);
    float coef = 1.f;
    if (a.max_norm > 0.f) {
        double acc = 0.0;
        const int n4 = a.n >> 2;
        const f32x4* g4 = reinterpret_cast<const f32x4*>(a.g);
        int k = tid;
        for (; k + 7 * ADAM_NT < n4; k += 8 * ADAM_NT) {
            f32x4 g[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) g[u] = g4[k + u * ADAM_NT];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                acc += ((double)g[u][0] * g[u][0] + (double)g[u][1] * g[u][1]) + ((double)g[u][2] * g[u][2] + (double)g[u][3] * g[u][3]);
        }
        {
            f32x4 g[8];                                          // the tail batch: still all loads first
#pragma unroll
            for (int u = 0; u < 8; ++u) g[u] = k + u * ADAM_NT < n4 ? g4[k + u * ADAM_NT] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 8; ++u)
                acc += ((double)g[u][0] * g[u][0] + (double)g[u][1] * g[u][1]) + ((double)g[u][2] * g[u][2] + (double)g[u][3] * g[u][3]);
        }
        for (int j = 4 * n4 + tid; j < a.n; j += ADAM_NT) acc += (double)a.g[j] * a.g[j];
        // wave total in every lane (xor butterfly: the partner sums are commutative pairs, all lanes hold the same bits), then the
        // 16 wave totals in fixed order -- every workgroup computes the identical norm
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if ((tid & 63) == 0) wred[tid >> 6] = acc;
        __syncthreads();
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < ADAM_NT / 64; ++w) tot += wred[w];
        const float norm = (float)sqrt(tot);
        if (blockIdx.x == 0 && tid == 0 && norm_out) *norm_out = norm;
        coef = a.max_norm / (norm + 1e-6f);
        coef = coef > 1.f ? 1.f : coef;
    } else {
        __syncthreads();
    }
    if (act) {
        float g = g0 * coef;
        if (i < a.n_decay && a.wd != 0.f) g = fmaf(a.wd, p0, g);
        const float m = a.beta1 * m0 + (1.f - a.beta1) * g;
        const float v = a.beta2 * v0 + (1.f - a.beta2) * g * g;
        a.m[i] = m;
        a.v[i] = v;
        const float denom = sqrtf(v) / sh[grp][1] + a.eps;
        a.p[i] = p0 - sh[grp][0] * (m / denom);
    }
    __syncthreads();
    if (tid == 0) {
        __threadfence();
        if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
            const int ng = a.grp.nseg ? PC_ADAM_GROUPS : 1;
            for (int g = 0; g < ng; ++g)
                if (!a.grp.nseg || ((a.grp.active_mask >> g) & 1)) a.step[g] += 1;
            *ticket = 0u;
        }
    }
}

// ---- band select + normalise -------------------------------------------------------------------------------------------
// raw tile (B, Craw, H, W) -> model input (B, 6, H, W) = [(raw[band[c]] - mean[c]) / std[c]]
struct SelNormArgs { const float* raw; float* out; int band[6]; float mean[6], stdv[6]; int Craw; int64_t hw, n; };

__global__ __launch_bounds__(256) void select_normalize_scalar_kernel(const SelNormArgs a) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * 256) {
        const int64_t pix = i % a.hw;
        const int c = (int)((i / a.hw) % 6);
        const int64_t b = i / (a.hw * 6);
        a.out[i] = (a.raw[(b * a.Craw + a.band[c]) * a.hw + pix] - a.mean[c]) / a.stdv[c];
    }
}

__global__ __launch_bounds__(256) void select_normalize_kernel(const SelNormArgs a) {
    // one float4 of 4 consecutive pixels per thread (hw % 4 == 0 is checked on the host, else scalar path)
    const unsigned hw4 = (unsigned)(a.hw >> 2);
    const unsigned n4 = (unsigned)(a.n >> 2);
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
        const unsigned plane = i / hw4, pix4 = i - plane * hw4;
        const unsigned c = plane % 6u, b = plane / 6u;
        const f32x4 v = *reinterpret_cast<const f32x4*>(a.raw + ((int64_t)b * a.Craw + a.band[c]) * a.hw + 4 * (int64_t)pix4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[e] - a.mean[c]) / a.stdv[c];
        *reinterpret_cast<f32x4*>(a.out + 4 * (int64_t)i) = o;
    }
}

}  // namespace

extern "C" int pc_loss_fwd_bwd(const float* popcount, const float* y, const double* stats, const float* lam4,
                               float scale_regularization, float lam_weak, float inv_B, int B,
                               float* loss_out, float* g_popcount, float* g_scale_const, void* stream) {
    if (!popcount || !y || !lam4 || !loss_out || !g_popcount || !g_scale_const) return PC_EINVAL;
    pc_loss_args a{};
    a.y = y;
    for (int i = 0; i < 4; ++i) a.lam[i] = lam4[i];
    a.sreg = scale_regularization; a.lam_weak = lam_weak; a.inv_B = inv_B; a.B = B;
    a.loss_out = loss_out; a.g_popcount = g_popcount; a.g_scale_const = g_scale_const;
    hipLaunchKernelGGL(loss_fwd_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a, popcount, stats);
    PC_CHECK_LAUNCH();
    return 0;
}

extern "C" int pc_grad_norm(const float* g, int n, float* norm_out, void* stream) {
    if (!g || !norm_out) return PC_EINVAL;
    hipLaunchKernelGGL(grad_norm_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, g, n, norm_out);
    PC_CHECK_LAUNCH();
    return 0;
}

static int fill_groups(AdamArgs& a, const pc_adam_groups* groups, int n) {
    if (!groups) return 0;
    if (groups->nseg < 1 || groups->nseg > PC_ADAM_MAX_SEG || groups->seg_end[groups->nseg - 1] != n) return PC_EINVAL;
    for (int s = 0; s < groups->nseg; ++s) {
        if (groups->seg_group[s] < 0 || groups->seg_group[s] >= PC_ADAM_GROUPS) return PC_EINVAL;
        if (s && groups->seg_end[s] < groups->seg_end[s - 1]) return PC_EINVAL;
    }
    a.grp = *groups;
    return 0;
}

extern "C" int pc_adam_clip_step(float* p, const float* g, float* m, float* v, int n, int n_decay, const float* hyper_dev,
                                 float weight_decay, float beta1, float beta2, float eps, float max_norm,
                                 const float* norm_dev, int32_t* step_dev, const pc_adam_groups* groups, void* stream) {
    if (!p || !g || !m || !v || !hyper_dev || !step_dev) return PC_EINVAL;
    AdamArgs a{};
    if (int rc = fill_groups(a, groups, n)) return rc;
    a.p = p; a.g = g; a.m = m; a.v = v; a.n = n; a.n_decay = n_decay; a.hyper = hyper_dev; a.wd = weight_decay;
    a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.max_norm = max_norm; a.norm = norm_dev; a.step = step_dev;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_clip_kernel, dim3((n + 255) / 256), dim3(256), 0, st, a);
    PC_CHECK_LAUNCH();
    hipLaunchKernelGGL(adam_step_inc_kernel, dim3(1), dim3(64), 0, st, step_dev, groups ? PC_ADAM_GROUPS : 1,
                       groups ? groups->active_mask : 1);
    PC_CHECK_LAUNCH();
    return 0;
}

extern "C" int pc_adam_clip_step_fused(float* p, const float* g, float* m, float* v, int n, int n_decay,
                                       const float* hyper_dev, float weight_decay, float beta1, float beta2, float eps,
                                       float max_norm, float* norm_out_dev, int32_t* step_dev,
                                       const pc_adam_groups* groups, void* stream) {
    if (!p || !g || !m || !v || !hyper_dev || !step_dev) return PC_EINVAL;
    if ((reinterpret_cast<uintptr_t>(g) & 15) != 0) return PC_EINVAL;
    static unsigned* ticket = nullptr;      // zero-initialised once; the kernel leaves it at zero
    if (!ticket) {
        hipError_t e = hipMalloc(&ticket, sizeof(unsigned));
        if (e != hipSuccess) return (int)e;
        // synchronous memset + device-wide sync: the first kernel may run on a non-blocking stream, which a null-stream
        // memset alone does not order against
        e = hipMemset(ticket, 0, sizeof(unsigned));
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) return (int)e;
    }
    AdamArgs a{};
    if (int rc = fill_groups(a, groups, n)) return rc;
    a.p = p; a.g = g; a.m = m; a.v = v; a.n = n; a.n_decay = n_decay; a.hyper = hyper_dev; a.wd = weight_decay;
    a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.max_norm = max_norm; a.norm = nullptr; a.step = step_dev;
    hipLaunchKernelGGL(adam_clip_fused_kernel, dim3((n + ADAM_NT - 1) / ADAM_NT), dim3(ADAM_NT), 0, (hipStream_t)stream, a, norm_out_dev, ticket);
    PC_CHECK_LAUNCH();
    return 0;
}

extern "C" int pc_select_normalize(const float* raw, int Craw, const int* band6, const float* mean6, const float* std6,
                                   float* out, int B, int H, int W, void* stream) {
    if (!raw || !band6 || !mean6 || !std6 || !out) return PC_EINVAL;
    SelNormArgs a{};
    a.raw = raw; a.out = out; a.Craw = Craw; a.hw = (int64_t)H * W; a.n = (int64_t)B * 6 * H * W;
    for (int i = 0; i < 6; ++i) {
        if (band6[i] < 0 || band6[i] >= Craw) return PC_EINVAL;
        a.band[i] = band6[i]; a.mean[i] = mean6[i]; a.stdv[i] = std6[i];
    }
    const bool vec = (a.hw % 4 == 0) && ((reinterpret_cast<uintptr_t>(raw) & 15) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0) &&
                     a.n < (int64_t)1 << 33;
    int grid = (int)(((vec ? a.n / 4 : a.n) + 255) / 256);
    if (grid > 8192) grid = 8192;
    if (grid < 1) grid = 1;
    if (vec) hipLaunchKernelGGL(select_normalize_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(select_normalize_scalar_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    PC_CHECK_LAUNCH();
    return 0;
}


// ---- the trainer's augmentations in ONE pass (round 6) ----------------------------------------------------------------------------------
// run_train.py:386-402 / utils/transform.py: RandomBrightness -> RandomGamma on the raw Sentinel-2 digital numbers, then (after the
// concatenation [S2, S1]) RandomVerticalFlip -> RandomHorizontalFlip -> RandomRotationTransform(90 / 180 / 270) jointly on input and
// admin_mask, all with ONE coin per batch.  The coins and factors are drawn on the host with the reference's generators in its order
// (popcorn_amd/utils/transform.py: draw_fused_params); this kernel applies them while it assembles the 6-channel raw tile the step's ingest
// normalises -- one launch instead of ~35 elementwise / flip / rot90 / cat launches per step.  Per element the S2 arithmetic is the
// reference's, operation by operation in fp32: x / 10000 * beta clamped to [0, 1] * 10000; max(x, 0) / 10000 to the power gamma clamped * 10000.
namespace {
struct AugArgs {
    const float* s2; const float* s1; const float* admin;
    float* raw; float* admin_out;
    int B, H, W, Ho, Wo;
    int vflip, hflip, rot;             // rot = number of counter-clockwise quarter turns (torch.rot90 k)
    int bright, gam;
    float beta, gamma;
};
__global__ __launch_bounds__(256) void augment_raw_kernel(const AugArgs a) {
    const int64_t npx = (int64_t)a.Ho * a.Wo;
    const int64_t n = (int64_t)a.B * npx;
    const int64_t hw = (int64_t)a.H * a.W;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * 256) {
        const int b = (int)(idx / npx);
        const int64_t r = idx - (int64_t)b * npx;
        const int i = (int)(r / a.Wo), j = (int)(r - (int64_t)i * a.Wo);
        // out = rot90^k(hflip(vflip(in))): undo the rotation, then the flips
        int y, x;
        switch (a.rot & 3) {
            case 1: y = j; x = a.W - 1 - i; break;
            case 2: y = a.H - 1 - i; x = a.W - 1 - j; break;
            case 3: y = a.H - 1 - j; x = i; break;
            default: y = i; x = j; break;
        }
        if (a.hflip) x = a.W - 1 - x;
        if (a.vflip) y = a.H - 1 - y;
        const int64_t src = (int64_t)y * a.W + x;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = a.s2[((int64_t)b * 4 + c) * hw + src];
            if (a.bright) v = fminf(fmaxf(__fmul_rn(__fdiv_rn(v, 10000.f), a.beta), 0.f), 1.f) * 10000.f;
            if (a.gam) v = fminf(fmaxf(powf(__fdiv_rn(fmaxf(v, 0.f), 10000.f), a.gamma), 0.f), 1.f) * 10000.f;
            a.raw[((int64_t)b * 6 + c) * npx + r] = v;
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) a.raw[((int64_t)b * 6 + 4 + c) * npx + r] = a.s1[((int64_t)b * 2 + c) * hw + src];
        a.admin_out[(int64_t)b * npx + r] = a.admin[(int64_t)b * hw + src];
    }
}
// odd rotations: a 32 x 32 tile of the output is a 32 x 32 tile of the source read ACROSS its rows -- staged through LDS so that both the
// reads (along source rows) and the writes (along output rows) are coalesced (the direct form above touches one cache line per thread)
__global__ __launch_bounds__(256) void augment_raw_transposed_kernel(const AugArgs a) {
    __shared__ float tile[7][32][33];
    const int tiles_j = (a.Wo + 31) / 32, tiles_i = (a.Ho + 31) / 32;
    const int64_t hw = (int64_t)a.H * a.W, npx = (int64_t)a.Ho * a.Wo;
    const int tx = threadIdx.x & 31, ty0 = threadIdx.x >> 5;            // 32 x 8 threads, four rows each
    for (int t = blockIdx.x; t < a.B * tiles_i * tiles_j; t += gridDim.x) {
        const int b = t / (tiles_i * tiles_j), r = t - b * tiles_i * tiles_j;
        const int i0 = (r / tiles_j) * 32, j0 = (r % tiles_j) * 32;
        // source coordinates of output (i, j): rot 1: (y, x) = (j, W - 1 - i); rot 3: (H - 1 - j, i); then the flips.  Along a source ROW x
        // follows i: the tile is read with tx -> i (x direction), ty -> j (y direction)
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int dj = ty0 + 8 * q, di = tx;
            const int i = i0 + di, j = j0 + dj;
            if (i < a.Ho && j < a.Wo) {
                int y, x;
                if ((a.rot & 3) == 1) { y = j; x = a.W - 1 - i; } else { y = a.H - 1 - j; x = i; }
                if (a.hflip) x = a.W - 1 - x;
                if (a.vflip) y = a.H - 1 - y;
                const int64_t src = (int64_t)y * a.W + x;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float v = a.s2[((int64_t)b * 4 + c) * hw + src];
                    if (a.bright) v = fminf(fmaxf(__fmul_rn(__fdiv_rn(v, 10000.f), a.beta), 0.f), 1.f) * 10000.f;
                    if (a.gam) v = fminf(fmaxf(powf(__fdiv_rn(fmaxf(v, 0.f), 10000.f), a.gamma), 0.f), 1.f) * 10000.f;
                    tile[c][dj][di] = v;
                }
#pragma unroll
                for (int c = 0; c < 2; ++c) tile[4 + c][dj][di] = a.s1[((int64_t)b * 2 + c) * hw + src];
                tile[6][dj][di] = a.admin[(int64_t)b * hw + src];
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int di = ty0 + 8 * q, dj = tx;                 // output row i0 + di, column j0 + dj: tx along the output row
            const int i = i0 + di, j = j0 + dj;
            if (i < a.Ho && j < a.Wo) {
                const int64_t o = (int64_t)i * a.Wo + j;
#pragma unroll
                for (int c = 0; c < 6; ++c) a.raw[((int64_t)b * 6 + c) * npx + o] = tile[c][dj][di];
                a.admin_out[(int64_t)b * npx + o] = tile[6][dj][di];
            }
        }
    }
}
}  // namespace

extern "C" int pc_augment_raw(const float* s2, const float* s1, const float* admin, float* raw, float* admin_out, int B, int H, int W,
                              int vflip, int hflip, int rot, int bright, float beta, int gam, float gamma, void* stream) {
    if (!s2 || !s1 || !admin || !raw || !admin_out || B < 1 || H < 1 || W < 1 || rot < 0 || rot > 3) return PC_EINVAL;
    AugArgs a{};
    a.s2 = s2; a.s1 = s1; a.admin = admin; a.raw = raw; a.admin_out = admin_out;
    a.B = B; a.H = H; a.W = W;
    a.Ho = (rot & 1) ? W : H; a.Wo = (rot & 1) ? H : W;
    a.vflip = vflip; a.hflip = hflip; a.rot = rot; a.bright = bright; a.gam = gam; a.beta = beta; a.gamma = gamma;
    const int64_t n = (int64_t)B * H * W;
    int grid = (int)((n + 255) / 256);
    if (grid > 16384) grid = 16384;
    if (rot & 1) {
        int tiles = B * ((a.Ho + 31) / 32) * ((a.Wo + 31) / 32);
        if (tiles > 8192) tiles = 8192;
        hipLaunchKernelGGL(augment_raw_transposed_kernel, dim3(tiles), dim3(256), 0, (hipStream_t)stream, a);
    } else {
        hipLaunchKernelGGL(augment_raw_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    }
    PC_CHECK_LAUNCH();
    return 0;
}
