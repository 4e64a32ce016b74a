// common.h -- host-side helpers shared by the C-ABI translation units.
#ifndef ORBX_COMMON_H
#define ORBX_COMMON_H

#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <string>

#include "../../include/orbslam_hip.h"

namespace orbx {

// Last error text, per thread (the ABI never throws).
inline std::string &last_error()
{
    static thread_local std::string s;
    return s;
}

inline int set_error(int code, const char *what, const char *file, int line)
{
    char buf[512];
    snprintf(buf, sizeof(buf), "%s (%s:%d)", what, file, line);
    last_error() = buf;
    return code;
}

// True when a HIP device is usable.  Probed once; there is no CPU fallback.
inline bool device_ok()
{
    static int state = -1;
    if (state < 0) {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        state = (e == hipSuccess && n > 0) ? 1 : 0;
        (void)hipGetLastError();
    }
    return state == 1;
}

} // namespace orbx

#define ORBX_FAIL(code, msg) return orbx::set_error((code), (msg), __FILE__, __LINE__)

#define ORBX_HIP(call)                                                                  \
    do {                                                                                \
        hipError_t e__ = (call);                                                        \
        if (e__ != hipSuccess)                                                          \
            return orbx::set_error(ORBX_ERR_HIP, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

#define ORBX_NEED_DEVICE()                                                              \
    do {                                                                                \
        if (!orbx::device_ok()) ORBX_FAIL(ORBX_ERR_NO_DEVICE, "no usable HIP device (no CPU fallback exists)"); \
    } while (0)

#endif // ORBX_COMMON_H
