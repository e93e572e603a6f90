// Shared device helpers for libpopcorn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include "popcorn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte access at 4-byte alignment (global memory only)

// bf16 mixed precision (PC_PREC_BF16, popcorn_hip.h): round-to-nearest-even of an fp32 value to the nearest bf16, kept in an
// fp32 register (what torch's `.to(torch.bfloat16).to(torch.float32)` gives; NaN payloads aside).
__device__ __forceinline__ float pc_bf16r(float x) {
    return (float)(__bf16)x;            // v_cvt_pk_bf16_f32 (round-to-nearest-even in hardware) + a shift
}
// two floats -> one dword of two bf16 (lo in bits 0-15), rounding to nearest even: ONE v_cvt_pk_bf16_f32
typedef __bf16 pc_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pc_pack_bf16(float lo, float hi) {
    const pc_bf16x2 p = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, p);
}
extern int g_pc_precision;      // api.hip: pc_set_precision()

// ---- fp32 results on the bf16 matrix pipe: exact 3-way operand splits (head.hip round 5, conv3x3_bwd.hip / conv3x3_s3.hip round 6) ----
// a = a1 + a2 + a3 exactly, with a1 = rn_bf16(a), a2 = rn_bf16(a - a1), a3 = a - a1 - a2 (8 + 8 + 8 mantissa bits, every difference exact
// in fp32).  One PAIR of fp32 values -> the three packed bf16 pairs of its split (lo half = x0): 11 VALU instructions.
__device__ __forceinline__ void pc_split_pair(float x0, float x1, unsigned& q1, unsigned& q2, unsigned& q3) {
    q1 = pc_pack_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(q1 << 16), r1 = x1 - __uint_as_float(q1 & 0xffff0000u);
    q2 = pc_pack_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(q2 << 16), s1 = r1 - __uint_as_float(q2 & 0xffff0000u);
    q3 = pc_pack_bf16(s0, s1);
}
// conv kernels' multiplication form in PC_PREC_FP32 (api.hip: pc_set_conv_split): 1 = split-operand kernels where they exist
int pc_conv_split_on();

// ---- wave-uniform descriptor fields pinned in scalar registers (round 6) --------------------------------------------------------------------
// Kernel arguments are read with s_load from the kernarg segment.  hipcc treats those loads as free to REMATERIALISE: under SGPR pressure it
// re-issues them wherever a field is used -- inside the strip loops that meant up to 34 scalar-memory round trips per iteration
// (s_load + s_waitcnt lgkmcnt(0), which also drains the wave's LDS queue; tools/isa_scalar_loads.py lists them per kernel), 1,800 of
// 10,000 cycles per strip in the first version of conv3x3_bwd_s3_kernel.  A value that went through an empty asm is opaque to that: it
// stays in an SGPR (or is spilled to a VGPR lane, a v_readlane away).  Pointers come back GLOBAL (through an address_space(1) cast): a
// pointer that lost its provenance would be dereferenced with flat_load, which counts on vmcnt AND lgkmcnt.
__device__ __forceinline__ void pc_pin(int& v) { asm volatile("" : "+s"(v)); }
__device__ __forceinline__ void pc_pin(unsigned& v) { asm volatile("" : "+s"(v)); }
__device__ __forceinline__ void pc_pin(int64_t& v) { asm volatile("" : "+s"(v)); }
__device__ __forceinline__ void pc_pin(float& v) { asm volatile("" : "+s"(v)); }
template <typename T>
__device__ __forceinline__ T* pc_pin_ptr(T* ptr) {
    uint64_t v = reinterpret_cast<uint64_t>(ptr);
    asm volatile("" : "+s"(v));
    return (T*)(__attribute__((address_space(1))) T*)v;
}
__device__ __forceinline__ void pc_pin(pc_src& s) {
    s.ptr = pc_pin_ptr(s.ptr);
    pc_pin(s.bstride); pc_pin(s.cstride); pc_pin(s.rstride); pc_pin(s.xstride); pc_pin(s.H); pc_pin(s.W); pc_pin(s.oy); pc_pin(s.ox); pc_pin(s.C);
}
__device__ __forceinline__ void pc_pin(pc_dst& d) {
    d.ptr = pc_pin_ptr(d.ptr);
    pc_pin(d.bstride); pc_pin(d.cstride); pc_pin(d.rstride); pc_pin(d.xstride);
}

// ---- element access for fp32 / bf16 containers (bf16 = unsigned short bits) -------------------------------------------------
typedef unsigned short pc_bf16_t;
__device__ __forceinline__ float pc_bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned short pc_f2bf(float x) { return (unsigned short)(pc_pack_bf16(x, 0.f) & 0xffffu); }
__device__ __forceinline__ float pc_ld1(const float* p) { return *p; }
__device__ __forceinline__ float pc_ld1(const pc_bf16_t* p) { return pc_bf2f(*p); }
__device__ __forceinline__ void pc_st1(float* p, float v) { *p = v; }
// four consecutive elements (bf16: four channels of a channels-last pixel); the bf16 forms need 8-byte alignment, the fp32 forms 16-byte
__device__ __forceinline__ f32x4 pc_ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 pc_ld4(const pc_bf16_t* p) {
    const uint2 q = *reinterpret_cast<const uint2*>(p);
    return f32x4{__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16), __uint_as_float(q.y & 0xffff0000u)};
}
__device__ __forceinline__ void pc_st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void pc_st4(pc_bf16_t* p, f32x4 v) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pc_pack_bf16(v[0], v[1]), pc_pack_bf16(v[2], v[3]));
}
__device__ __forceinline__ void pc_st2(float* p, float a, float b) { *reinterpret_cast<float2*>(p) = make_float2(a, b); }
// element i of a source described by a pc_src (run-time dtype: generic / fallback paths only)
// x stride of a descriptor (0 / 1 = planar rows), and whether it describes a planar tensor -- the only form the fp32 kernels take
__host__ __device__ __forceinline__ int pc_xs(const pc_src& s) { return s.xstride > 1 ? s.xstride : 1; }
__host__ __device__ __forceinline__ int pc_xs(const pc_dst& d) { return d.xstride > 1 ? d.xstride : 1; }
__host__ __device__ __forceinline__ bool pc_planar(const pc_src& s) { return s.xstride <= 1; }
__host__ __device__ __forceinline__ bool pc_planar(const pc_dst& d) { return d.xstride <= 1; }
// channels-last bf16: one aligned 16-byte slot per (pixel, 8-channel group)
__host__ __device__ __forceinline__ bool pc_cl_ok(const void* ptr, int dtype, int64_t bstride, int64_t cstride, int rstride, int xstride) {
    return dtype == PC_BF16 && cstride == 1 && xstride >= 8 && xstride % 8 == 0 && rstride % 8 == 0 && bstride % 8 == 0 &&
           (reinterpret_cast<uintptr_t>(ptr) & 15) == 0;
}
__host__ __device__ __forceinline__ bool pc_cl_ok(const pc_src& s) { return pc_cl_ok(s.ptr, s.dtype, s.bstride, s.cstride, s.rstride, s.xstride); }
__host__ __device__ __forceinline__ bool pc_cl_ok(const pc_dst& d) { return pc_cl_ok(d.ptr, d.dtype, d.bstride, d.cstride, d.rstride, d.xstride); }

__device__ __forceinline__ float pc_src_at(const pc_src& s, int64_t i) {
    return s.dtype == PC_BF16 ? pc_bf2f(reinterpret_cast<const pc_bf16_t*>(s.ptr)[i]) : s.ptr[i];
}

#define PC_CHECK_LAUNCH()                         \
    do {                                          \
        hipError_t e__ = hipGetLastError();       \
        if (e__ != hipSuccess) return (int)e__;   \
    } while (0)

// One-time launcher setup (hipFuncSetAttribute(MaxDynamicSharedMemorySize), resident-grid probes) is per DEVICE, not per process:
// the guard is a bit mask over the current device ordinal, set with an atomic OR (a process that launches on a second device
// repeats the setup there; two host threads racing through it both do the idempotent setup).
struct pc_once_per_device {
    std::atomic<uint64_t> done{0};
    static uint64_t bit() {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess) d = 0;
        return 1ull << (d & 63);
    }
    bool need() const { return !(done.load(std::memory_order_acquire) & bit()); }
    void mark() { done.fetch_or(bit(), std::memory_order_release); }
};

// Workgroups of 256 threads (one wave per SIMD) that are resident on the whole chip at once for a kernel using `regs`
// unified VGPR+AGPR registers per lane (512 per SIMD lane, allocated in blocks of 8) and `lds` bytes of LDS (160 KB / CU).
static inline int pc_resident_workgroups(int regs, size_t lds) {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) cus = pr.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    const int r = regs < 8 ? 8 : (regs + 7) & ~7;
    int per_cu = 512 / r;
    if (lds > 0 && (int)((160u * 1024u) / lds) < per_cu) per_cu = (int)((160u * 1024u) / lds);
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    return per_cu * cus;
}

// Division by a launch-invariant divisor without the ~40-instruction VALU sequence the compiler emits for a runtime
// integer divide (there is no hardware integer divide; in the persistent tile loops those sequences were ~1000
// instructions per stage per wave -- tools/ablate_conv.py).  q = umulhi(n, ceil(2^32 / d)) is exact for n * d < 2^32; pc_div
// corrects the estimate beyond that.
struct pc_fastdiv {
    uint32_t d, m;
};
__device__ __forceinline__ void pc_pin(pc_fastdiv& f) { pc_pin(f.d); pc_pin(f.m); }
static inline pc_fastdiv pc_make_fastdiv(uint32_t d) {
    pc_fastdiv f;
    f.d = d ? d : 1;
    f.m = f.d == 1 ? 0u : (uint32_t)((((uint64_t)1 << 32) + f.d - 1) / f.d);
    return f;
}
// Round 4: m = ceil(2^32 / d) OVER-estimates the quotient by one for some n once n * d reaches 2^32 -- 100 x 100 tiles never get
// there, a 2 x 2100 x 2150 census region does (group index 564 k / 282 k groups per image -> sample index 2 of 2 = a fault in the head
// backward).  The estimate is never more than one too large for n < 2^32 (n * (m - 2^32 / d) / 2^32 < 1): one multiply + compare
// makes the quotient exact for every n, d with n + d < 2^32.
__device__ __forceinline__ uint32_t pc_div(uint32_t n, const pc_fastdiv& f) {
    if (f.d == 1) return n;
    const uint32_t q = __umulhi(n, f.m);
    return q - (uint32_t)(q * f.d > n);
}

// XCD-aware block remap (8 XCDs, block b runs on XCD b % 8): give every XCD a contiguous slice of the tile
// space so that neighbouring tiles (which share halo rows and the same weights) hit the same private L2.
// Bijective for any nwg (cdna_hip_programming.md T1).
__device__ __forceinline__ int pc_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// Per-channel folded BN: scale = gamma / sqrt(var + eps); shift = (bias - mean) * scale + beta.
__device__ __forceinline__ void pc_bn_fold(const pc_bn& bn, int c, float& scale, float& shift) {
    const float cb = bn.conv_bias ? bn.conv_bias[c] : 0.f;
    if (bn.gamma) {
        scale = bn.gamma[c] * (1.0f / sqrtf(bn.var[c] + bn.eps));
        shift = (cb - bn.mean[c]) * scale + bn.beta[c];
    } else {
        scale = 1.f;
        shift = cb;
    }
}

__device__ __forceinline__ int pc_reflect(int i, int n) {
    // F.pad(mode='reflect'): -1 -> 1, n -> n-2
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

// Fetch conv-domain element (b, c, y, x) of a source; y/x may lie outside the conv domain [0,H)x[0,W) -> 0
// (the conv's own zero padding).
__device__ __forceinline__ float pc_fetch(const pc_src& s, int b, int c, int y, int x, int H, int W) {
    if ((unsigned)y >= (unsigned)H || (unsigned)x >= (unsigned)W) return 0.f;
    if (s.mode == PC_SRC_DIRECT) {
        const int ys = y - s.oy, xs = x - s.ox;
        if ((unsigned)ys >= (unsigned)s.H || (unsigned)xs >= (unsigned)s.W) return 0.f;
        return pc_src_at(s, b * s.bstride + c * s.cstride + (int64_t)ys * s.rstride + (int64_t)xs * pc_xs(s));
    } else if (s.mode == PC_SRC_POOL2) {
        const int xst = pc_xs(s);
        const int64_t o = b * s.bstride + c * s.cstride + (int64_t)(2 * y) * s.rstride + (int64_t)(2 * x) * xst;
        // nn.MaxPool2d(2): floor mode, windows never cross the source extent for y < H/2, x < W/2
        return fmaxf(fmaxf(pc_src_at(s, o), pc_src_at(s, o + xst)), fmaxf(pc_src_at(s, o + s.rstride), pc_src_at(s, o + s.rstride + xst)));
    } else {
        const int ys = pc_reflect(y - s.oy, s.H), xs = pc_reflect(x - s.ox, s.W);
        return pc_src_at(s, b * s.bstride + s.chmap[c & 3] * s.cstride + (int64_t)ys * s.rstride + (int64_t)xs * pc_xs(s));
    }
}

// One 16-byte segment [xg, xg+4) of conv-domain row y of a REFLECT source (reflect padding + channel gather fused into
// the first conv, popcorn.py:231-258,130-134): interior segments are one (possibly unaligned) 16-byte load, segments that
// touch a reflected border fall back to four element loads; everything outside the conv domain is the conv's zero padding.
__device__ __forceinline__ f32x4 pc_fetch_reflect_seg(const pc_src& s, int b, int c, int y, int xg, int H, int W) {
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if ((unsigned)y >= (unsigned)H || xg + 3 < 0 || xg >= W) return v;
    const int ys = pc_reflect(y - s.oy, s.H);
    const float* row = s.ptr + b * s.bstride + s.chmap[c & 3] * s.cstride + (int64_t)ys * s.rstride;
    const int xs = xg - s.ox;
    if (xg >= 0 && xg + 3 < W && xs >= 0 && xs + 3 < s.W) {
        const f32x4u t = *reinterpret_cast<const f32x4u*>(row + xs);
        v = f32x4{t[0], t[1], t[2], t[3]};
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if ((unsigned)(xg + e) < (unsigned)W) v[e] = row[pc_reflect(xs + e, s.W)];
    }
    return v;
}

// ---- workgroup partials of the head backward (head.hip) -----------------------------------------------------------------------------
// [nwg][PC_PE_TOTAL] floats; the weight blocks are MFMA D fragments (written as the accumulators are held).  Shared with
// conv3x3_wgrad.hip: the batched reduction of a backward pass can finish them in the same launch (pc_wgrad_reduce_desc kind 3).
constexpr int PC_PE_W4 = 0, PC_PE_W2 = 4096, PC_PE_W0 = 8192, PC_PE_W6 = 9216, PC_PE_B0 = 9280, PC_PE_B2 = 9344, PC_PE_B4 = 9408, PC_PE_B6 = 9472;
constexpr int PC_PE_TOTAL = 9480;
// element e of a partial -> (gradient tensor t in the order {w0, b0, w2, b2, w4, b4, w6, b6}, index within it); t = -1: padding
__device__ __forceinline__ void pc_head_partial_target(int e, int& t, int& idx) {
    t = -1; idx = 0;
    if (e < PC_PE_W0) {                   // dW4 / dW2: D[(mb, nb) block][lane = (m >> 2) * 16 + n][reg = m & 3]
        const int e2 = e & 4095, l = e2 >> 2, blk = l >> 6, lane = l & 63;
        t = e < PC_PE_W2 ? 4 : 2;
        idx = (16 * (blk >> 2) + 4 * (lane >> 4) + (e2 & 3)) * 64 + 16 * (blk & 3) + (lane & 15);
    } else if (e < PC_PE_W6) {            // dW0: block mb, 16 feature columns
        const int e2 = e - PC_PE_W0, l = e2 >> 2, lane = l & 63;
        t = 0;
        idx = (16 * (l >> 6) + 4 * (lane >> 4) + (e2 & 3)) * 16 + (lane & 15);
    } else if (e < PC_PE_B0) { t = 6; idx = e - PC_PE_W6; }
    else if (e < PC_PE_B2) { t = 1; idx = e - PC_PE_B0; }
    else if (e < PC_PE_B4) { t = 3; idx = e - PC_PE_B2; }
    else if (e < PC_PE_B6) { t = 5; idx = e - PC_PE_B4; }
    else if (e == PC_PE_B6) { t = 7; idx = 0; }
}

// ---- cross-lane exchanges on the VALU -----------------------------------------------------------------------------------------------
// hipcc lowers EVERY __shfl_xor to ds_bpermute_b32 -- a trip through the LDS crossbar with its own lgkmcnt wait (192 of them per strip
// pair in the fp32 1x1-dot epilogue) -- although lane ^ 1, ^ 2, ^ 8 are DPP controls (quad_perm, row_ror:8) and lane ^ 16 / ^ 32 are
// gfx950's v_permlane16_swap / v_permlane32_swap.  All lanes of the 16-lane row (of the wave, for the swaps) must be active.
template <int CTRL>
__device__ __forceinline__ float pc_dpp(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float pc_lane_xor1(float x) { return pc_dpp<0xB1>(x); }      // quad_perm [1, 0, 3, 2]
__device__ __forceinline__ float pc_lane_xor2(float x) { return pc_dpp<0x4E>(x); }      // quad_perm [2, 3, 0, 1]
__device__ __forceinline__ float pc_lane_xor8(float x) { return pc_dpp<0x128>(x); }     // row_ror:8
// sum over the 8 lanes that share lane >> 3, in every lane, in the association order of the xor-1, -2, -4 butterfly (after the two quad
// steps the four lanes of a quad hold the same bits, so the mirror image within the half row IS the lane ^ 4 partner's value)
__device__ __forceinline__ float pc_sum8(float x) {
    x += pc_lane_xor1(x);
    x += pc_lane_xor2(x);
    x += pc_dpp<0x141>(x);                                                              // row_half_mirror
    return x;
}
// v + v of lane ^ 16 (lane ^ 32) in every lane: v_permlane16_swap exchanges the odd 16-lane rows of its first operand with the even rows
// of its second (v_permlane32_swap: the upper half with the lower half), so with both operands = v the two results hold the partner
// pair of every row
__device__ __forceinline__ float pc_xor16_sum(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float pc_xor32_sum(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}


// ---- loss forward + backward for one block of 256 threads (train_ops.hip: pc_loss_fwd_bwd; head.hip: pc_head_popcount_loss) ------------
// loss = sum_k lam[k] * L_k(popcount, y) + sreg * sum(scale) / Nsel, with L_k in {l1, log_l1, mse, log_mse} taken as a mean over the GLOBAL
// batch (inv_B = 1 / (world * B)); outputs d(lam_weak * loss) / d popcount[b] and the constant d(lam_weak * loss) / d scale on selected pixels.
struct pc_loss_args {
    const float* y;
    float lam[4]; float sreg, lam_weak, inv_B; int B;
    float* loss_out;        // [2]: {optimisation loss (local part of the batch mean + regulariser), regulariser}
    float* g_popcount; float* g_scale_const;
};
__device__ __forceinline__ void pc_loss_block(const pc_loss_args& a, const float* popcount, const double* stats, double* red /* [256] shared */) {
    double acc = 0.0;
    for (int b = threadIdx.x; b < a.B; b += 256) {
        const float pc = popcount[b], y = a.y[b];
        const float d = pc - y;
        const float lp = logf(pc + 1.f), ly = logf(y + 1.f);
        const float dl = lp - ly;
        const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        const float sgl = dl > 0.f ? 1.f : (dl < 0.f ? -1.f : 0.f);
        const float l = a.lam[0] * fabsf(d) + a.lam[1] * fabsf(dl) + a.lam[2] * d * d + a.lam[3] * dl * dl;
        const float g = a.lam[0] * sg + a.lam[1] * sgl / (pc + 1.f) + a.lam[2] * 2.f * d + a.lam[3] * 2.f * dl / (pc + 1.f);
        a.g_popcount[b] = a.lam_weak * a.inv_B * g;
        acc += (double)l;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double nsel = stats ? stats[0] : 0.0, ssum = stats ? stats[1] : 0.0;
        const double reg = (a.sreg > 0.f && nsel > 0.0) ? (double)a.sreg * ssum / nsel : 0.0;
        a.loss_out[0] = (float)(red[0] * (double)a.inv_B + reg);
        a.loss_out[1] = (float)reg;
        *a.g_scale_const = (a.sreg > 0.f && nsel > 0.0) ? (float)((double)a.lam_weak * (double)a.sreg / nsel) : 0.f;
    }
}
