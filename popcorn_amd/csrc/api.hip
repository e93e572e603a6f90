// api.hip -- library-level entry points of libpopcorn_hip.so.
#include "common.h"

extern "C" int pc_abi_version(void) { return PC_ABI_VERSION; }

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
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}
