// tile_loader.h -- stage a (NCH x 18 x 34) halo tile of a conv input into LDS.
//
// Work split: a "row job" is one (channel, row) of the tile = 34 floats.  16 lanes serve one job: lanes 0-7 move the
// 32 interior floats as one aligned 16-byte load each, lanes 8/9 fetch the two halo columns, so a 256-thread
// workgroup retires 16 rows per pass with no per-element index arithmetic (the generic per-element path costs ~60
// VALU instructions per float and made the conv kernels loader-bound: profiles/r1_v0).  The interior lands at LDS
// column COL0 (a multiple of 4) so it can be written with one ds_write_b128 per lane.
//
// Fast paths (decided once on the host, see pc_src_fast_mode): 1 = aligned DIRECT source, 2 = aligned POOL2 source
// (2x2 max taken on the fly from two rows of two float4).  Everything else (reflect padding, odd sizes, offset
// sources) goes through pc_fetch per element.
#pragma once
#include "common.h"

// host side: can the vector path be used for this source on a conv domain of H x W?
static inline int pc_src_fast_mode(const pc_src& s, int H, int W) {
    if (s.C == 0 || s.dtype != PC_F32) return 0;      // bf16 containers: staged paths have their own check, the rest is pc_fetch
    const bool al = ((reinterpret_cast<uintptr_t>(s.ptr) & 15) == 0) && (s.rstride % 4 == 0) && (s.cstride % 4 == 0) &&
                    (s.bstride % 4 == 0);
    if (!al) return 0;
    if (s.mode == PC_SRC_DIRECT && s.oy == 0 && s.ox == 0 && s.H == H && s.W == W && (W % 4) == 0) return 1;
    if (s.mode == PC_SRC_POOL2 && s.W == 2 * W && s.H >= 2 * H && (W % 4) == 0) return 2;
    return 0;
}

template <int RS, int CS, int COL0, bool VEC_STORE>
__device__ __forceinline__ void pc_load_row_job(float* lds, const pc_src& s, int fast, int b, int c, int ci, int r,
                                                int y0, int x0, int H, int W, int lane16) {
    const int y = y0 - 1 + r;
    float* dst = lds + ci * CS + r * RS + COL0;
    if (lane16 < 8) {
        const int x = x0 + 4 * lane16;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (fast == 1) {
            if ((unsigned)y < (unsigned)H && x < W)
                v = *reinterpret_cast<const f32x4*>(s.ptr + b * s.bstride + c * s.cstride + (int64_t)y * s.rstride + x);
        } else if (fast == 2) {
            if ((unsigned)y < (unsigned)H && x < W) {
                const float* p = s.ptr + b * s.bstride + c * s.cstride + (int64_t)(2 * y) * s.rstride + 2 * x;
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(p), a1 = *reinterpret_cast<const f32x4*>(p + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(p + s.rstride), b1 = *reinterpret_cast<const f32x4*>(p + s.rstride + 4);
                v[0] = fmaxf(fmaxf(a0[0], a0[1]), fmaxf(b0[0], b0[1]));
                v[1] = fmaxf(fmaxf(a0[2], a0[3]), fmaxf(b0[2], b0[3]));
                v[2] = fmaxf(fmaxf(a1[0], a1[1]), fmaxf(b1[0], b1[1]));
                v[3] = fmaxf(fmaxf(a1[2], a1[3]), fmaxf(b1[2], b1[3]));
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = pc_fetch(s, b, c, y, x + e, H, W);
        }
        if (VEC_STORE) {
            *reinterpret_cast<f32x4*>(dst + 4 * lane16) = v;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[4 * lane16 + e] = v[e];
        }
    } else if (lane16 == 8) {
        const float t = pc_fetch(s, b, c, y, x0 - 1, H, W);
        dst[-1] = t;
    } else if (lane16 == 9) {
        const float t = pc_fetch(s, b, c, y, x0 + 32, H, W);
        dst[32] = t;
    }
}

// channels [c_begin, c_begin + NCH) of cat[A, B] -> lds[ci][r][COL0 + (x - x0)], rows y0-1 .. y0+16, cols x0-1 .. x0+32
template <int NCH, int RS, int CS, int COL0, bool VEC_STORE>
__device__ __forceinline__ void pc_load_halo_tile(float* lds, const pc_src& A, const pc_src& Bs, int fastA, int fastB,
                                                  int c_begin, int b, int y0, int x0, int H, int W, int tid) {
    const int lane16 = tid & 15, grp = tid >> 4;
    const int CA = A.C;
    for (int job = grp; job < NCH * 18; job += 16) {
        const int ci = job / 18, r = job - ci * 18;
        const int cg = c_begin + ci;
        if (cg < CA) pc_load_row_job<RS, CS, COL0, VEC_STORE>(lds, A, fastA, b, cg, ci, r, y0, x0, H, W, lane16);
        else pc_load_row_job<RS, CS, COL0, VEC_STORE>(lds, Bs, fastB, b, cg - CA, ci, r, y0, x0, H, W, lane16);
    }
}

// ---- register-staged variant of the aligned DIRECT path -------------------------------------------------------------
// All of a thread's 16-byte loads for a tile are issued back to back into registers (pc_halo_issue), and written to LDS
// later (pc_halo_commit).  That (a) keeps NIT independent loads in flight per lane instead of one load -> wait ->
// ds_write per loop trip (the per-trip form cost ~1 us of exposed memory latency per trip: tools/ablate_conv.py), and
// (b) lets the caller issue the loads of tile t+1 *before* the MFMA phase of tile t, so HBM latency hides under MFMA
// inside a single workgroup.  A row is moved as 10 aligned float4 segments x0-4 .. x0+35 (both halo columns ride in
// the outer segments), so every lane runs the same code: no divergence between interior and halo lanes.
// Requires COL0 == 4 and both sources in fast mode 1.
template <int NCH>
struct pc_halo_regs {
    static constexpr int NIT = (NCH * 18 + 15) / 16;
    f32x4 v[NIT];
};

template <int NCH>
__device__ __forceinline__ void pc_halo_issue(pc_halo_regs<NCH>& R, const pc_src& A, const pc_src& Bs, int c_begin,
                                              int b, int y0, int x0, int H, int W, int tid) {
    const int lane16 = tid & 15, grp = tid >> 4;
    const int x = x0 - 4 + 4 * lane16;
    const bool xin = lane16 < 10 && x >= 0 && x < W;
    const int CA = A.C;
#pragma unroll
    for (int it = 0; it < pc_halo_regs<NCH>::NIT; ++it) {
        const int job = grp + 16 * it;
        const int ci = job / 18, r = job - ci * 18;
        const int cg = c_begin + ci;
        const int y = y0 - 1 + r;
        const float* base = cg < CA ? A.ptr + b * A.bstride + cg * A.cstride + (int64_t)y * A.rstride
                                    : Bs.ptr + b * Bs.bstride + (cg - CA) * Bs.cstride + (int64_t)y * Bs.rstride;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (xin && job < NCH * 18 && (unsigned)y < (unsigned)H) v = *reinterpret_cast<const f32x4*>(base + x);
        R.v[it] = v;
    }
}

template <int NCH, int RS, int CS>
__device__ __forceinline__ void pc_halo_commit(float* lds, const pc_halo_regs<NCH>& R, int tid) {
    const int lane16 = tid & 15, grp = tid >> 4;
#pragma unroll
    for (int it = 0; it < pc_halo_regs<NCH>::NIT; ++it) {
        const int job = grp + 16 * it;
        const int ci = job / 18, r = job - ci * 18;
        if (lane16 < 10 && job < NCH * 18) *reinterpret_cast<f32x4*>(lds + ci * CS + r * RS + 4 * lane16) = R.v[it];
    }
}
