// api.hip -- library-level entry points of libpopcorn_hip.so.
#include "common.h"

int g_pc_precision = PC_PREC_FP32;

extern "C" int pc_abi_version(void) { return PC_ABI_VERSION; }

extern "C" int pc_set_precision(int mode) {
    if (mode != PC_PREC_FP32 && mode != PC_PREC_BF16) return PC_EINVAL;
    const int prev = g_pc_precision;
    g_pc_precision = mode;
    return prev;
}

extern "C" int pc_get_precision(void) { return g_pc_precision; }

// ---- multiplication form of the fp32 conv kernels (popcorn_hip.h: pc_set_conv_split) --------------------------------------------------
static int g_conv_split = -1;
int pc_conv_split_on() {
    if (g_conv_split < 0) {
        const char* ev = getenv("POPCORN_CONV_SPLIT");
        g_conv_split = (ev && ev[0] == '0') ? 0 : 1;
    }
    return g_conv_split;
}
extern "C" int pc_get_conv_split(void) { return pc_conv_split_on(); }
extern "C" int pc_set_conv_split(int on) {
    const int prev = pc_conv_split_on();
    g_conv_split = on ? 1 : 0;
    return prev;
}

extern "C" int pc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" const char* pc_error_string(int code) {
    switch (code) {
        case 0: return "ok";
        case PC_EINVAL: return "invalid argument / unsupported shape";
        case PC_ENOGPU: return "no HIP device";
        case PC_ENOMEM: return "pc_train_step: arena too small (pc_step_io.arena_needed)";
        case PC_ENOTSUP: return "pc_train_step: variant not covered by the native executor";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}

// sizeof() of the ABI structs as the compiler laid them out: 0 = pc_src, 1 = pc_dst, 2 = pc_bn,
// 3 = pc_conv_fwd_desc, 4 = pc_adam_groups, 5 = pc_level2_fwd_desc, 6 = pc_step_plan, 7 = pc_step_io (binding self-check)
extern "C" int pc_sizeof(int which) {
    switch (which) {
        case 0: return (int)sizeof(pc_src);
        case 1: return (int)sizeof(pc_dst);
        case 2: return (int)sizeof(pc_bn);
        case 3: return (int)sizeof(pc_conv_fwd_desc);
        case 4: return (int)sizeof(pc_adam_groups);
        case 5: return (int)sizeof(pc_level2_fwd_desc);
        case 6: return (int)sizeof(pc_step_plan);
        case 7: return (int)sizeof(pc_step_io);
        default: return -1;
    }
}

// ---- test hook: the cross-lane helpers of common.h (DPP / v_permlane swaps instead of ds_bpermute) on one wave ----------------------
__global__ __launch_bounds__(64) void lane_ops_kernel(const float* in, float* out) {
    const int l = threadIdx.x;
    const float x = in[l];
    out[l] = pc_sum8(x);
    out[64 + l] = fmaxf(x, pc_lane_xor8(x));
    out[128 + l] = pc_xor16_sum(x);
    out[192 + l] = pc_xor32_sum(x);
    out[256 + l] = pc_lane_xor1(x);
    out[320 + l] = pc_lane_xor2(x);
}
extern "C" int pc_debug_lane_ops(const float* in, float* out, void* stream) {
    if (!in || !out) return PC_EINVAL;
    hipLaunchKernelGGL(lane_ops_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, in, out);
    PC_CHECK_LAUNCH();
    return 0;
}
