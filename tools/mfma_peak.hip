// Matrix-pipe peak microbenchmark (SURVEY.md section 8d "Hardware peaks ... must confirm").
//
// Register-resident MFMA loops on every CU of the chip -- no memory traffic inside the loop -- for
//   f32   v_mfma_f32_16x16x4_f32     (the fp32 conv / head kernels' instruction)
//   bf16  v_mfma_f32_16x16x32_bf16   (the bf16 mode's instruction)
// at 1 / 2 / 4 waves per SIMD, on random and on all-zero operands, ~2 s each, plus the operand patterns of the conv kernel:
//   lds32   one ds_read_b32 A operand per MFMA (conflict-free, the planar fp32 conv kernel's pattern)
//   lds128  one ds_read_b128 per MFMA (the judge's variant: 4x the LDS bytes per MFMA)
//   lds128q one ds_read_b128 per FOUR MFMAs (same LDS bytes as lds32 in a quarter of the DS instructions)
// Every run reports: TFLOP/s, shader cycles per MFMA per SIMD (s_memtime over the loop), the effective shader clock
// (s_memtime ticks / wall_clock64 time, 100 MHz reference) and -- when readable -- sclk / power sampled from sysfs by a host
// thread while the kernel runs.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_peak.bin tools/mfma_peak.hip -lpthread && tools/mfma_peak.bin [seconds]
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <unistd.h>
#include <string>
#include <thread>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct Stamp { unsigned long long cyc0, cyc1, wall0, wall1; };

constexpr int NACC = 8;            // independent accumulators per wave: 8 x 32 cycles covers the 40-cycle dependent latency

// OPER 0: operands in registers; 1: one ds_read_b32 A operand per MFMA; 2: one ds_read_b128 per MFMA; 3: one ds_read_b128 per
// four MFMAs.  LDS operands are software-pipelined by construction: the reads of iteration it+1 are issued before the MFMAs
// of iteration it (two register sets, loop unrolled by two), so a single wave does not expose the LDS latency.
// STREAM: every iteration each lane also loads 16 bytes from / every second iteration stores 16 bytes to a large buffer
// (coalesced, advancing): ~1 KiB per wave per 8 MFMAs = the conv layers' ~18 flop per byte.
// All operands pass through a VALU multiply by a run-time 1.0f before the loop so that hipcc's s_waitcnt for their loads
// sits in front of the loop, not between the MFMAs.
template <int OPER, bool STREAM>
__global__ __launch_bounds__(256) void f32_loop(const float* __restrict__ src, float* __restrict__ sink, Stamp* st, int iters, float one,
                                                u32x4* __restrict__ big, unsigned big_mask) {
    __shared__ __attribute__((aligned(16))) float lds[4 * 2048];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* my = lds + wave * 2048;
    for (int i = lane; i < 2048; i += 64) my[i] = src[(i * 7 + lane) & 1023] * one;
    __syncthreads();
    float a[NACC], b[NACC];
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        a[i] = src[(lane * NACC + i) & 1023] * one;
        b[i] = src[(lane * NACC + i + 517) & 1023] * one;
        acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    unsigned gpos = (blockIdx.x * 256 + threadIdx.x) & big_mask;
    const unsigned gstep = (gridDim.x * 256) & big_mask;
    u32x4 gv = u32x4{0, 0, 0, 0}, gt = u32x4{0, 0, 0, 0};
    auto rd = [&](float (&av)[NACC], f32x4 (&aq)[NACC], int it) {
        if (OPER == 1) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) av[i] = my[((it + i) & 15) * 64 + lane];            // unit stride: conflict-free
        } else if (OPER == 2) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) aq[i] = *reinterpret_cast<const f32x4*>(my + ((it + i) & 7) * 256 + lane * 4);
        } else if (OPER == 3) {
#pragma unroll
            for (int i = 0; i < NACC / 4; ++i) aq[i] = *reinterpret_cast<const f32x4*>(my + ((it + i) & 7) * 256 + lane * 4);
        }
    };
    auto mm = [&](const float (&av)[NACC], const f32x4 (&aq)[NACC]) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            const float x = OPER == 0 ? a[i] : OPER == 1 ? av[i] : OPER == 2 ? aq[i][i & 3] : aq[i >> 2][i & 3];
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, b[i], acc[i], 0, 0, 0);
        }
    };
    float av0[NACC], av1[NACC];
    f32x4 aq0[NACC], aq1[NACC];
    rd(av0, aq0, 0);
    const unsigned long long w0 = wall_clock64(), c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it += 2) {
        rd(av1, aq1, it + 1);
        u32x4 tn = gt;
        if (STREAM) { tn = big[gpos]; gpos = (gpos + gstep) & big_mask; }      // consumed one iteration (16 MFMAs) later
        mm(av0, aq0);
        rd(av0, aq0, it + 2);
        if (STREAM) { gv ^= gt; big[gpos] = gv; gpos = (gpos + gstep) & big_mask; gt = tn; }
        mm(av1, aq1);
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < NACC; ++i) s += acc[i];
    if (s[0] + s[1] + s[2] + s[3] + av0[0] + aq0[0][0] == 12345.678f) sink[0] = s[0] + (float)gv[0];   // keeps everything alive
    if (threadIdx.x == 0) st[blockIdx.x] = Stamp{c0, c1, w0, w1};
}

template <int OPER, bool STREAM>     // OPER 0: registers; 2: one ds_read_b128 (= one operand of 8 bf16) per MFMA
__global__ __launch_bounds__(256) void bf16_loop(const float* __restrict__ src, float* __restrict__ sink, Stamp* st, int iters, float one,
                                                 u32x4* __restrict__ big, unsigned big_mask) {
    __shared__ __attribute__((aligned(16))) float lds[4 * 2048];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* my = lds + wave * 2048;
    for (int i = lane; i < 2048; i += 64) my[i] = src[(i * 7 + lane) & 1023] * one;
    __syncthreads();
    bf16x8 a[NACC], b[NACC];
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a[i][j] = (__bf16)(src[(lane * 64 + i * 8 + j) & 1023] * one);
            b[i][j] = (__bf16)(src[(lane * 64 + i * 8 + j + 331) & 1023] * one);
        }
        acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    unsigned gpos = (blockIdx.x * 256 + threadIdx.x) & big_mask;
    const unsigned gstep = (gridDim.x * 256) & big_mask;
    u32x4 gv = u32x4{0, 0, 0, 0}, gt = u32x4{0, 0, 0, 0};
    auto rd = [&](u32x4 (&aq)[NACC], int it) {
        if (OPER == 2) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) aq[i] = *reinterpret_cast<const u32x4*>(my + ((it + i) & 7) * 256 + lane * 4);
        }
    };
    auto mm = [&](const u32x4 (&aq)[NACC]) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(OPER == 0 ? a[i] : __builtin_bit_cast(bf16x8, aq[i]), b[i], acc[i], 0, 0, 0);
    };
    u32x4 aq0[NACC], aq1[NACC];
    rd(aq0, 0);
    const unsigned long long w0 = wall_clock64(), c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it += 2) {
        rd(aq1, it + 1);
        u32x4 tn = gt;
        if (STREAM) { tn = big[gpos]; gpos = (gpos + gstep) & big_mask; }
        mm(aq0);
        rd(aq0, it + 2);
        if (STREAM) { gv ^= gt; big[gpos] = gv; gpos = (gpos + gstep) & big_mask; gt = tn; }
        mm(aq1);
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < NACC; ++i) s += acc[i];
    if (s[0] + s[1] + s[2] + s[3] + (float)aq0[0][0] == 12345.678f) sink[0] = s[0] + (float)gv[0];
    if (threadIdx.x == 0) st[blockIdx.x] = Stamp{c0, c1, w0, w1};
}

// ---------------------------------------------------------------------------------------------- sysfs sampler (best effort)
static std::string g_pci;        // "0000:xx:yy.z" of the HIP device: only the matching drm card is sampled
static std::string find_hwmon(const char* leaf) {
    for (int card = 0; card < 64; ++card) {
        const std::string dev = "/sys/class/drm/card" + std::to_string(card) + "/device";
        char link[512];
        const ssize_t n = readlink(dev.c_str(), link, sizeof(link) - 1);
        if (n <= 0) continue;
        link[n] = 0;
        if (!g_pci.empty() && !strcasestr(link, g_pci.c_str())) continue;
        std::string base = dev + "/hwmon";
        DIR* d = opendir(base.c_str());
        if (!d) continue;
        while (dirent* e = readdir(d)) {
            if (strncmp(e->d_name, "hwmon", 5)) continue;
            std::string p = base + "/" + e->d_name + "/" + leaf;
            if (FILE* f = fopen(p.c_str(), "r")) { fclose(f); closedir(d); return p; }
        }
        closedir(d);
    }
    return "";
}
static double read_num(const std::string& p) {
    if (p.empty()) return -1;
    FILE* f = fopen(p.c_str(), "r");
    if (!f) return -1;
    double v = -1;
    if (fscanf(f, "%lf", &v) != 1) v = -1;
    fclose(f);
    return v;
}

struct Sampler {
    std::string fpower, fsclk;
    std::atomic<bool> stop{false};
    std::vector<double> pw, ck;
    std::thread th;
    Sampler() {
        fpower = find_hwmon("power1_average");
        if (fpower.empty()) fpower = find_hwmon("power1_input");
        fsclk = find_hwmon("freq1_input");
    }
    void start() {
        stop = false; pw.clear(); ck.clear();
        th = std::thread([this] {
            while (!stop) {
                double p = read_num(fpower), c = read_num(fsclk);
                if (p >= 0) pw.push_back(p * 1e-6);
                if (c >= 0) ck.push_back(c * 1e-6);
                std::this_thread::sleep_for(std::chrono::milliseconds(50));
            }
        });
    }
    void finish(double& p, double& c) {
        stop = true; th.join();
        p = c = -1;
        // skip the first quarter (ramp)
        if (!pw.empty()) { double s = 0; size_t n0 = pw.size() / 4; for (size_t i = n0; i < pw.size(); ++i) s += pw[i]; p = s / (pw.size() - n0); }
        if (!ck.empty()) { double s = 0; size_t n0 = ck.size() / 4; for (size_t i = n0; i < ck.size(); ++i) s += ck[i]; c = s / (ck.size() - n0); }
    }
};

typedef void (*kern_t)(const float*, float*, Stamp*, int, float, u32x4*, unsigned);

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 2.0;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    char pci[64] = {0};
    if (hipDeviceGetPCIBusId(pci, sizeof(pci), 0) == hipSuccess) g_pci = pci;
    Sampler smp;
    printf("{\"device\": \"%s\", \"arch\": \"%s\", \"pci\": \"%s\", \"cus\": %d, \"clock_mhz_max\": %d, \"seconds_per_run\": %.1f, "
           "\"sysfs_power\": \"%s\", \"sysfs_sclk\": \"%s\", \"runs\": [\n", prop.name, prop.gcnArchName, pci, cus, prop.clockRate / 1000,
           seconds, smp.fpower.c_str(), smp.fsclk.c_str());
    float *src_rand, *src_zero, *sink;
    Stamp* st;
    u32x4* big;
    const size_t big_elems = (size_t)1 << 26;                  // 64 M x 16 B = 1 GiB: streaming, not cache-resident
    CK(hipMalloc(&src_rand, 4096)); CK(hipMalloc(&src_zero, 4096)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&st, sizeof(Stamp) * cus * 8));
    CK(hipMalloc(&big, big_elems * sizeof(u32x4)));
    std::vector<float> h(1024);
    srand(1600);
    for (auto& v : h) v = (float)(rand() & 0xffff) / 32768.f - 1.f;
    CK(hipMemcpy(src_rand, h.data(), 4096, hipMemcpyHostToDevice));
    CK(hipMemset(src_zero, 0, 4096));
    struct Variant { const char* name; kern_t k; double flop_per_mfma; bool stream; };
    const Variant vars[] = {
        {"f32_16x16x4 regs", f32_loop<0, false>, 2048.0, false},
        {"f32_16x16x4 lds32 (1 ds_read_b32 / MFMA, pipelined)", f32_loop<1, false>, 2048.0, false},
        {"f32_16x16x4 lds128 (1 ds_read_b128 / MFMA, pipelined)", f32_loop<2, false>, 2048.0, false},
        {"f32_16x16x4 lds128q (1 ds_read_b128 / 4 MFMA, pipelined)", f32_loop<3, false>, 2048.0, false},
        {"f32_16x16x4 regs + HBM stream (1 KiB / wave / 8 MFMA)", f32_loop<0, true>, 2048.0, true},
        {"f32_16x16x4 lds32 + HBM stream", f32_loop<1, true>, 2048.0, true},
        {"bf16_16x16x32 regs", bf16_loop<0, false>, 16384.0, false},
        {"bf16_16x16x32 lds128 (1 ds_read_b128 / MFMA, pipelined)", bf16_loop<2, false>, 16384.0, false},
        {"bf16_16x16x32 regs + HBM stream (1 KiB / wave / 8 MFMA)", bf16_loop<0, true>, 16384.0, true},
    };
    bool first = true;
    for (const Variant& v : vars) {
        for (int wps : {1, 2, 4}) {
            for (int zero = 0; zero < 2; ++zero) {
                if (zero && wps == 4) continue;
                const float* src = zero ? src_zero : src_rand;
                if (v.stream) {            // stream contents follow the operand kind (zero buffer = low switching activity)
                    if (zero) CK(hipMemset(big, 0, big_elems * sizeof(u32x4)));
                    else CK(hipMemset(big, 0x5a, big_elems * sizeof(u32x4)));
                }
                const int grid = cus * wps;
                hipEvent_t e0, e1;
                CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                int iters = 20000;         // calibrate
                hipLaunchKernelGGL(v.k, dim3(grid), dim3(256), 0, 0, src, sink, st, iters, 1.0f, big, (unsigned)(big_elems - 1));
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(v.k, dim3(grid), dim3(256), 0, 0, src, sink, st, iters, 1.0f, big, (unsigned)(big_elems - 1));
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                iters = ((int)(iters * (seconds * 1e3 / ms)) + 1) & ~1;
                smp.start();
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(v.k, dim3(grid), dim3(256), 0, 0, src, sink, st, iters, 1.0f, big, (unsigned)(big_elems - 1));
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                double pw, ck;
                smp.finish(pw, ck);
                CK(hipEventElapsedTime(&ms, e0, e1));
                std::vector<Stamp> hs(grid);
                CK(hipMemcpy(hs.data(), st, sizeof(Stamp) * grid, hipMemcpyDeviceToHost));
                double cyc = 0, wall = 0;
                for (auto& s : hs) { cyc += (double)(s.cyc1 - s.cyc0); wall += (double)(s.wall1 - s.wall0); }
                cyc /= grid; wall /= grid;
                const double mfma_per_wave = (double)iters * NACC;
                const double total = mfma_per_wave * 4.0 * grid;
                const double tflops = total * v.flop_per_mfma / (ms * 1e-3) / 1e12;
                const double eff_ghz = cyc / (wall / 100e6) / 1e9;      // wall_clock64: 100 MHz
                // matrix-pipe cycles per MFMA on one SIMD, from the event time and the measured clock (all SIMDs busy all the time)
                const double cyc_per_mfma = eff_ghz * 1e9 * (ms * 1e-3) / (total / (cus * 4.0));
                const double gbs = v.stream ? (double)iters * 0.75 * 1024.0 * 4.0 * grid / (ms * 1e-3) / 1e9 : 0.0;
                printf("%s {\"variant\": \"%s\", \"waves_per_simd\": %d, \"operands\": \"%s\", \"ms\": %.1f, \"tflops\": %.1f, "
                       "\"cycles_per_mfma_per_simd\": %.2f, \"effective_clock_ghz\": %.3f, \"hbm_gbps\": %.0f, \"sclk_sysfs_mhz\": %.0f, \"power_w\": %.0f}",
                       first ? " " : ",\n ", v.name, wps, zero ? "zero" : "random", ms, tflops, cyc_per_mfma, eff_ghz, gbs, ck, pw);
                first = false;
                fflush(stdout);
                CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
            }
        }
    }
    printf("\n]}\n");
    return 0;
}
