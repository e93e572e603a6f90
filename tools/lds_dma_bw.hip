// LDS-DMA loader experiment for the planar fp32 conv class.
//
// The strip pipeline of conv3x3_mfma_kernel<8,8,fwd> (grouped x4, B = 64, 128 x 128: 1024 workgroups x 4 waves, every wave owns 32 x 4
// output strips and stages an [8 ch][6 rows][48 floats] halo image in its private LDS region) with a STAND-IN compute phase -- the
// kernel's 96 v_mfma_f32_16x16x4_f32 per strip, each fed by one ds_read_b32 of the image -- and its 8 x 4 16-byte stores, in three
// loader forms:
//   S   shipped: one aligned 16-byte global load per (channel, lane) into registers, issued BEFORE the compute phase of the previous
//       strip, written to LDS (8 ds_write_b128 per lane) after it                                         [36.9 KB of LDS / workgroup]
//   D1  global_load_lds_dwordx4 (LDS-DMA), single image: the 9,216-byte image is 576 contiguous 16-byte pieces = 9 wave-wide DMA
//       instructions (12 lanes per 48-float row keep the conflict-free row stride; lanes outside the image read a zero page); no
//       staging registers, no ds_write -- but the DMA overwrites the image the compute phase reads, so it can only be issued AFTER
//       that phase and its latency is covered by the other waves of the SIMD only
//   D2  LDS-DMA, two images per wave (the DMA of strip t+1 runs under the compute phase of strip t)     [73.7 KB of LDS / workgroup:
//       2 workgroups per CU instead of 4]
// each as: loads only | loads + compute | loads + compute + stores.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_dma_bw tools/lds_dma_bw.hip && /tmp/lds_dma_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

constexpr int NP = 4, NB = 64, C = 8, H = 128, W = 128;
constexpr int TILES_X = W / 32, TILES_Y = H / 16, NTILES = NP * NB * TILES_X * TILES_Y;
constexpr int RS = 48, CS = 6 * RS, IMG = C * CS;              // floats: row stride, channel stride, strip image (2304 floats)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

template <int FORM, int MODE>        // FORM 0 = S, 1 = D1, 2 = D2;  MODE bit 0: compute phase, bit 1: stores
__global__ __launch_bounds__(256) void strips(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ zeros, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    constexpr int NBUF = FORM == 2 ? 2 : 1;
    float* const img0 = lds + wave * NBUF * IMG;
    f32x4 acc[4] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    f32x4 R[8];
    auto coords = [&](int t, int64_t& img, int& y0, int& x0) {
        // XCD-aware order of the real kernels (pc_xcd_remap): block b runs on XCD b % 8 and gets a contiguous slice of the tile space, so
        // neighbouring tiles (shared halo rows) hit the same private L2
        t = (t & 7) * (NTILES / 8) + (t >> 3);
        img = t / (TILES_X * TILES_Y);
        const int rem = t % (TILES_X * TILES_Y);
        y0 = (rem / TILES_X) * 16 + 4 * wave;
        x0 = (rem % TILES_X) * 32;
    };
    // S: lane = (row, 4-pixel segment of the 40-float row)
    const int s_r = lane / 10, s_seg = lane % 10;
    bool s_ok = false;
    auto issue_regs = [&](int t) {
        int64_t img; int y0, x0;
        coords(t, img, y0, x0);
        const int y = y0 - 1 + s_r, x = x0 - 4 + 4 * s_seg;
        s_ok = lane < 60 && y >= 0 && y < H && x >= 0 && x < W;
        const int64_t off = s_ok ? (img * C * H + y) * W + x : 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) R[c] = *reinterpret_cast<const f32x4*>(in + off + (int64_t)c * H * W);
    };
    auto commit_regs = [&](float* im) {
        if (lane < 60) {
#pragma unroll
            for (int c = 0; c < 8; ++c)
                *reinterpret_cast<f32x4*>(im + c * CS + s_r * RS + 4 * s_seg) = s_ok ? R[c] : f32x4{0, 0, 0, 0};
        }
    };
    // D: piece id = lane + 64 i -> (channel, row, 16-byte piece of the 48-float row); pieces 10, 11 of a row are padding
    int d_ch[9], d_r[9], d_seg[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int id = lane + 64 * i;
        d_ch[i] = id / 72; d_r[i] = (id % 72) / 12; d_seg[i] = id % 12;
    }
    auto issue_dma = [&](int t, float* im) {
        int64_t img; int y0, x0;
        coords(t, img, y0, x0);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int y = y0 - 1 + d_r[i], x = x0 - 4 + 4 * d_seg[i];
            const bool ok = d_seg[i] < 10 && y >= 0 && y < H && x >= 0 && x < W;
            const float* src = ok ? in + ((img * C + d_ch[i]) * H + y) * W + x : zeros;
            __builtin_amdgcn_global_load_lds(src, (lds_void*)(im + i * 256), 16, 0, 0);
        }
    };
    float wreg[24];
#pragma unroll
    for (int k = 0; k < 24; ++k) wreg[k] = 0.001f * (float)(k + lane);
    auto compute = [&](const float* im) {
        // the A-operand read pattern of the real kernel: lane (x = li, input row lk) of unit u, tap column dx
        const float* lrow = im + lk * RS + 3 + li;
#pragma unroll
        for (int ci = 0; ci < 8; ++ci)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(lrow[ci * CS + (u >> 1) * 2 * RS + (u & 1) * 16 + dx], wreg[ci * 3 + dx], acc[u], 0, 0, 0);
    };
    auto store = [&](int t) {
        int64_t img; int y0, x0;
        coords(t, img, y0, x0);
        const int s_row = li >> 3, col = li & 7;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int y = y0 + 2 * (u >> 1) + s_row, x = x0 + (u & 1) * 16 + 4 * lk;
            *reinterpret_cast<f32x4*>(out + ((img * C + col) * H + y) * W + x) = acc[u];
        }
    };
    unsigned tok = 0;
    int t = blockIdx.x, prev = -1, cur = 0;
    if (FORM == 0) { if (t < NTILES) issue_regs(t); }
    else if (FORM == 2) { if (t < NTILES) issue_dma(t, img0); }
    for (; t < NTILES; t += gridDim.x) {
        const int nt = t + gridDim.x;
        float* im = img0 + cur * IMG;
        if (FORM == 0) {
            commit_regs(im);                                   // (waits for the loads issued one strip ago)
            if (nt < NTILES) issue_regs(nt);
        } else if (FORM == 1) {
            issue_dma(t, im);                                  // the image is free only now
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this strip's DMA (issued one strip ago)
            if (nt < NTILES) issue_dma(nt, img0 + (cur ^ 1) * IMG);
        }
        if (prev >= 0 && (MODE & 2)) store(prev);
        if (MODE & 1) compute(im);
        else tok += __float_as_uint(im[lane]);                 // loads only: consume one word of the image
        prev = t;
        if (FORM == 2) cur ^= 1;
    }
    if (prev >= 0 && (MODE & 2)) store(prev);
    tok += __float_as_uint(acc[0][0]) ^ __float_as_uint(acc[1][1]) ^ __float_as_uint(acc[2][2]) ^ __float_as_uint(acc[3][3]);
    if (tok == 0x12345678u) sink[0] = tok;
}

template <int FORM, int MODE>
static void run(const char* tag, float** ins, float** outs, const float* zeros, unsigned* sink, int nsets) {
    const size_t ldsb = (size_t)4 * (FORM == 2 ? 2 : 1) * IMG * sizeof(float);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&strips<FORM, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = FORM == 2 ? 512 : 1024;                   // resident workgroups: 2 / 4 per CU
    for (int s = 0; s < nsets; ++s) hipLaunchKernelGGL((strips<FORM, MODE>), dim3(grid), dim3(256), ldsb, 0, ins[s], outs[s], zeros, sink);
    hipDeviceSynchronize();
    const int reps = 5;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r)
        for (int s = 0; s < nsets; ++s) hipLaunchKernelGGL((strips<FORM, MODE>), dim3(grid), dim3(256), ldsb, 0, ins[s], outs[s], zeros, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / (reps * nsets);
    printf("{\"form\": \"%s\", \"phase\": \"%s\", \"us\": %.1f, \"lds_bytes_per_wg\": %zu, \"grid\": %d}\n", tag,
           MODE == 0 ? "loads" : MODE == 1 ? "loads+compute" : "loads+compute+stores", us, ldsb, grid);
}

int main() {
    const size_t n = (size_t)NP * NB * C * H * W;
    const int nsets = 4;
    float *ins[nsets], *outs[nsets], *zeros;
    unsigned* sink;
    hipMalloc(&sink, 4);
    hipMalloc(&zeros, 4096);
    hipMemset(zeros, 0, 4096);
    for (int s = 0; s < nsets; ++s) {
        hipMalloc(&ins[s], n * 4 + 4096);
        hipMalloc(&outs[s], n * 4 + 4096);
        hipMemset(ins[s], 1, n * 4);
        hipMemset(outs[s], 0, n * 4);
    }
    run<0, 0>("S  registers + ds_write (shipped)", ins, outs, zeros, sink, nsets);
    run<1, 0>("D1 lds-dma, one image", ins, outs, zeros, sink, nsets);
    run<2, 0>("D2 lds-dma, two images", ins, outs, zeros, sink, nsets);
    run<0, 1>("S  registers + ds_write (shipped)", ins, outs, zeros, sink, nsets);
    run<1, 1>("D1 lds-dma, one image", ins, outs, zeros, sink, nsets);
    run<2, 1>("D2 lds-dma, two images", ins, outs, zeros, sink, nsets);
    run<0, 3>("S  registers + ds_write (shipped)", ins, outs, zeros, sink, nsets);
    run<1, 3>("D1 lds-dma, one image", ins, outs, zeros, sink, nsets);
    run<2, 3>("D2 lds-dma, two images", ins, outs, zeros, sink, nsets);
    return 0;
}
