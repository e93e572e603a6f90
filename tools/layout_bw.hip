// Memory-side microbenchmark behind the bf16 activation layout decision: the conv kernels' strip
// traffic WITHOUT the arithmetic -- 4 problems x 64 images x 8 channels x 128 x 128 bf16 in, the same out, persistent
// waves that own 32 x 4 output strips (6 x 34 input halo), one strip of prefetch, stores of strip t-1 behind the loads of
// strip t+1 -- in two HBM layouts:
//   A  planar NCHW            (pieces: 80 B per (channel, row) in, 64 B per (channel, row) out)
//   B  channel-packed NHWC8   (pieces: 544 B per row in, 512 B per row out; one 16-byte slot per pixel)
// each as loads only / stores only / both.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/layout_bw tools/layout_bw.hip && /tmp/layout_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

constexpr int NP = 4, NB = 64, C = 8, H = 128, W = 128;
constexpr int TILES_X = W / 32, TILES_Y = H / 16, NTILES = NP * NB * TILES_X * TILES_Y;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// fp32 variants (LAYOUT 2 = NCHW fp32: 16-byte pieces per lane as conv3x3_mfma_kernel<..., false> issues them;
// LAYOUT 3 = NHWC8 fp32: 32-byte pixel slots)
template <int LAYOUT, int MODE>
__global__ __launch_bounds__(256) void strips32(const float* __restrict__ in, float* __restrict__ out, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    unsigned acc = 0;
    u32x4 R[8];
    auto coords = [&](int t, int64_t& img, int& y0, int& x0) {
        img = t / (TILES_X * TILES_Y);
        const int rem = t % (TILES_X * TILES_Y);
        y0 = (rem / TILES_X) * 16 + 4 * wave;
        x0 = (rem % TILES_X) * 32;
    };
    auto issue = [&](int t) {
        int64_t img; int y0, x0;
        coords(t, img, y0, x0);
        if (LAYOUT == 2) {
            const int r = lane / 10, seg = lane % 10;
            int y = y0 - 1 + r, x = x0 - 4 + 4 * seg;
            const bool ok = lane < 60 && y >= 0 && y < H && x >= 0 && x < W;
            const int64_t off = ok ? (img * C * H + y) * W + x : 0;
#pragma unroll
            for (int c = 0; c < 8; ++c) R[c] = *reinterpret_cast<const u32x4*>(in + off + (int64_t)c * H * W);
        } else {
#pragma unroll
            for (int i = 0; i < 7; ++i) {          // 6 x 34 slots x 2 halves = 408 16-byte pieces
                const int id = lane + 64 * i;
                const int r = id / 68, px = (id % 68) >> 1, hf = id & 1;
                int y = y0 - 1 + r, x = x0 - 1 + px;
                const bool ok = id < 408 && y >= 0 && y < H && x >= 0 && x < W;
                const int64_t off = ok ? ((img * H + y) * W + x) * 8 + 4 * hf : 0;
                R[i] = *reinterpret_cast<const u32x4*>(in + off);
            }
        }
    };
    auto consume = [&]() {
#pragma unroll
        for (int c = 0; c < (LAYOUT == 2 ? 8 : 7); ++c) acc += R[c][0] ^ R[c][1] ^ R[c][2] ^ R[c][3];
    };
    auto store = [&](int t, unsigned v) {
        int64_t img; int y0, x0;
        coords(t, img, y0, x0);
        if (LAYOUT == 2) {
            const int s_row = li >> 3, col = li & 7;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int y = y0 + 2 * (u >> 1) + s_row, x = x0 + (u & 1) * 16 + 4 * lk;
                *reinterpret_cast<u32x4*>(out + ((img * C + col) * H + y) * W + x) = u32x4{v, v + u, v, v};
            }
        } else {
            const int s = lk >> 1, half = lk & 1;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int y = y0 + 2 * (u >> 1) + s, x = x0 + (u & 1) * 16 + li;
                *reinterpret_cast<u32x4*>(out + ((img * H + y) * W + x) * 8 + 4 * half) = u32x4{v, v + u, v, v};
            }
        }
    };
    int t = blockIdx.x;
    if (t < NTILES && (MODE & 1)) issue(t);
    int prev = -1;
    for (; t < NTILES; t += gridDim.x) {
        if (MODE & 1) consume();
        const int nt = t + gridDim.x;
        if (nt < NTILES && (MODE & 1)) issue(nt);
        if (prev >= 0 && (MODE & 2)) store(prev, acc);
        prev = t;
    }
    if (prev >= 0 && (MODE & 2)) store(prev, acc);
    if (acc == 0x12345678u) sink[0] = acc;
}

// NCHW fp32 with other strip shapes (SW x SH output pixels per wave; the workgroup stacks its 4 waves vertically),
// tiles walked in the XCD-aware order of the real kernels when XCD is set
template <int SW, int SH, int MODE, bool XCD>
__global__ __launch_bounds__(256) void strips_shape(const float* __restrict__ in, float* __restrict__ out, unsigned* sink) {
    constexpr int TX = W / SW, TY = H / (4 * SH), NT = NP * NB * TX * TY;
    constexpr int SEGS = SW / 4 + 2, NSEG = (SH + 2) * SEGS, NL = (NSEG + 63) / 64;
    constexpr int NST = SH * SW / 4 * 8 / 64;        // 16-byte stores per lane
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned acc = 0;
    u32x4 R[NL][8];
    auto coords = [&](int t, int64_t& img, int& y0, int& x0) {
        if (XCD) t = (t % 8) * (NT / 8) + t / 8;
        img = t / (TX * TY);
        const int rem = t % (TX * TY);
        y0 = (rem / TX) * 4 * SH + SH * wave;
        x0 = (rem % TX) * SW;
    };
    auto issue = [&](int t) {
        int64_t img; int y0, x0;
        coords(t, img, y0, x0);
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int id = lane + 64 * i;
            const int r = id / SEGS, seg = id % SEGS;
            int y = y0 - 1 + r, x = x0 - 4 + 4 * seg;
            const bool ok = id < NSEG && y >= 0 && y < H && x >= 0 && x < W;
            const int64_t off = ok ? (img * C * H + y) * W + x : 0;
#pragma unroll
            for (int c = 0; c < 8; ++c) R[i][c] = *reinterpret_cast<const u32x4*>(in + off + (int64_t)c * H * W);
        }
    };
    auto consume = [&]() {
#pragma unroll
        for (int i = 0; i < NL; ++i)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc += R[i][c][0] ^ R[i][c][1] ^ R[i][c][2] ^ R[i][c][3];
    };
    auto store = [&](int t, unsigned v) {
        int64_t img; int y0, x0;
        coords(t, img, y0, x0);
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int id = lane + 64 * u;                 // (channel, row, 4-pixel segment), x fastest
            const int seg = id % (SW / 4), r = (id / (SW / 4)) % SH, c = id / (SW / 4 * SH);
            *reinterpret_cast<u32x4*>(out + ((img * C + c) * H + y0 + r) * W + x0 + 4 * seg) = u32x4{v, v + u, v, v};
        }
    };
    int t = blockIdx.x;
    if (t < NT && (MODE & 1)) issue(t);
    int prev = -1;
    for (; t < NT; t += gridDim.x) {
        if (MODE & 1) consume();
        const int nt = t + gridDim.x;
        if (nt < NT && (MODE & 1)) issue(nt);
        if (prev >= 0 && (MODE & 2)) store(prev, acc);
        prev = t;
    }
    if (prev >= 0 && (MODE & 2)) store(prev, acc);
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int SW, int SH, bool XCD>
static void run_shape(const char* tag, uint16_t** ins, uint16_t** outs, unsigned* sink, int nsets) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto go = [&](int mode, int s) {
        if (mode == 1) hipLaunchKernelGGL((strips_shape<SW, SH, 1, XCD>), dim3(1024), dim3(256), 0, 0, (const float*)ins[s], (float*)outs[s], sink);
        if (mode == 2) hipLaunchKernelGGL((strips_shape<SW, SH, 2, XCD>), dim3(1024), dim3(256), 0, 0, (const float*)ins[s], (float*)outs[s], sink);
        if (mode == 3) hipLaunchKernelGGL((strips_shape<SW, SH, 3, XCD>), dim3(1024), dim3(256), 0, 0, (const float*)ins[s], (float*)outs[s], sink);
    };
    for (int mode = 1; mode <= 3; ++mode) {
        for (int s = 0; s < nsets; ++s) go(mode, s);
        hipDeviceSynchronize();
        const int reps = 5;
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r)
            for (int s = 0; s < nsets; ++s) go(mode, s);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / (reps * nsets);
        const double bytes = (double)NP * NB * C * H * W * 4 * (mode == 3 ? 2 : 1);
        printf("%-22s xcd %d  %s  %7.1f us  %6.2f TB/s\n", tag, (int)XCD, mode == 1 ? "loads " : mode == 2 ? "stores" : "both  ", us, bytes / us / 1e6);
    }
}

template <int LAYOUT, int MODE>   // MODE bit 0: loads, bit 1: stores
__global__ __launch_bounds__(256) void strips(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    unsigned acc = 0;
    uint2 RA[8];
    u32x4 RB[4];
    auto coords = [&](int t, int64_t& img, int& y0, int& x0) {
        img = t / (TILES_X * TILES_Y);
        const int rem = t % (TILES_X * TILES_Y);
        y0 = (rem / TILES_X) * 16 + 4 * wave;
        x0 = (rem % TILES_X) * 32;
    };
    auto issue = [&](int t) {
        int64_t img; int y0, x0;
        coords(t, img, y0, x0);
        if (LAYOUT == 0) {
            const int r = lane / 10, seg = lane % 10;
            int y = y0 - 1 + r, x = x0 - 4 + 4 * seg;
            const bool ok = lane < 60 && y >= 0 && y < H && x >= 0 && x < W;
            const int64_t off = ok ? (img * C * H + y) * W + x : 0;
#pragma unroll
            for (int c = 0; c < 8; ++c) RA[c] = *reinterpret_cast<const uint2*>(in + off + (int64_t)c * H * W);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int id = lane + 64 * i;
                const int r = id / 34, px = id % 34;
                int y = y0 - 1 + r, x = x0 - 1 + px;
                const bool ok = id < 204 && y >= 0 && y < H && x >= 0 && x < W;
                const int64_t off = ok ? ((img * H + y) * W + x) * 8 : 0;
                RB[i] = *reinterpret_cast<const u32x4*>(in + off);
            }
        }
    };
    auto consume = [&]() {
        if (LAYOUT == 0) {
#pragma unroll
            for (int c = 0; c < 8; ++c) acc += RA[c].x ^ RA[c].y;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc += RB[i][0] ^ RB[i][1] ^ RB[i][2] ^ RB[i][3];
        }
    };
    auto store = [&](int t, unsigned v) {
        int64_t img; int y0, x0;
        coords(t, img, y0, x0);
        if (LAYOUT == 0) {
            const int s_row = li >> 3, col = li & 7;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int y = y0 + 2 * (u >> 1) + s_row, x = x0 + (u & 1) * 16 + 4 * lk;
                *reinterpret_cast<uint2*>(out + ((img * C + col) * H + y) * W + x) = make_uint2(v, v + u);
            }
        } else {
            const int s = lk >> 1, half = lk & 1;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int y = y0 + 2 * (u >> 1) + s, x = x0 + (u & 1) * 16 + li;
                *reinterpret_cast<uint2*>(out + ((img * H + y) * W + x) * 8 + 4 * half) = make_uint2(v, v + u);
            }
        }
    };
    int t = blockIdx.x;
    if (t < NTILES && (MODE & 1)) issue(t);
    int prev = -1;
    for (; t < NTILES; t += gridDim.x) {
        if (MODE & 1) consume();
        const int nt = t + gridDim.x;
        if (nt < NTILES && (MODE & 1)) issue(nt);
        if (prev >= 0 && (MODE & 2)) store(prev, acc);
        prev = t;
    }
    if (prev >= 0 && (MODE & 2)) store(prev, acc);
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int LAYOUT, int MODE>
static void launch(int grid, uint16_t* in, uint16_t* out, unsigned* sink) {
    if (LAYOUT < 2) hipLaunchKernelGGL((strips<LAYOUT, MODE>), dim3(grid), dim3(256), 0, 0, in, out, sink);
    else hipLaunchKernelGGL((strips32<LAYOUT, MODE>), dim3(grid), dim3(256), 0, 0, (const float*)in, (float*)out, sink);
}

template <int LAYOUT, int MODE>
static void run(const char* tag, uint16_t** ins, uint16_t** outs, unsigned* sink, int nsets) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {1024}) {
        for (int s = 0; s < nsets; ++s) launch<LAYOUT, MODE>(grid, ins[s], outs[s], sink);
        hipDeviceSynchronize();
        const int reps = 5;
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r)
            for (int s = 0; s < nsets; ++s) launch<LAYOUT, MODE>(grid, ins[s], outs[s], sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / (reps * nsets);
        const double bytes = (double)NP * NB * C * H * W * (LAYOUT < 2 ? 2 : 4) * (((MODE & 1) ? 1 : 0) + ((MODE & 2) ? 1 : 0));
        printf("%-28s grid %4d  %7.1f us  %6.2f TB/s\n", tag, grid, us, bytes / us / 1e6);
    }
}

int main() {
    const size_t n = (size_t)NP * NB * C * H * W;
    const int nsets = 4;
    uint16_t *ins[nsets], *outs[nsets];
    unsigned* sink;
    hipMalloc(&sink, 4);
    for (int s = 0; s < nsets; ++s) {
        hipMalloc(&ins[s], n * 4 + 4096);
        hipMalloc(&outs[s], n * 4 + 4096);
        hipMemset(ins[s], 1, n * 4);
        hipMemset(outs[s], 0, n * 4);
    }
    run<0, 1>("A NCHW   loads", ins, outs, sink, nsets);
    run<0, 2>("A NCHW   stores", ins, outs, sink, nsets);
    run<0, 3>("A NCHW   loads+stores", ins, outs, sink, nsets);
    run<1, 1>("B NHWC8  loads", ins, outs, sink, nsets);
    run<1, 2>("B NHWC8  stores", ins, outs, sink, nsets);
    run<1, 3>("B NHWC8  loads+stores", ins, outs, sink, nsets);
    run<2, 1>("A32 NCHW fp32  loads", ins, outs, sink, nsets);
    run<2, 2>("A32 NCHW fp32  stores", ins, outs, sink, nsets);
    run<2, 3>("A32 NCHW fp32  loads+stores", ins, outs, sink, nsets);
    run<3, 1>("B32 NHWC8 fp32 loads", ins, outs, sink, nsets);
    run<3, 2>("B32 NHWC8 fp32 stores", ins, outs, sink, nsets);
    run<3, 3>("B32 NHWC8 fp32 loads+stores", ins, outs, sink, nsets);
    run_shape<32, 4, false>("NCHW fp32 32x4", ins, outs, sink, nsets);
    run_shape<32, 4, true>("NCHW fp32 32x4", ins, outs, sink, nsets);
    run_shape<64, 2, true>("NCHW fp32 64x2", ins, outs, sink, nsets);
    run_shape<128, 1, true>("NCHW fp32 128x1", ins, outs, sink, nsets);
    run_shape<128, 2, true>("NCHW fp32 128x2", ins, outs, sink, nsets);
    run_shape<64, 4, true>("NCHW fp32 64x4", ins, outs, sink, nsets);
    return 0;
}
