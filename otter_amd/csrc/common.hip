// common.hip — error channel and library-wide entry points of libotter_hip.so.
#include <stdarg.h>

#include "otter_common.h"

namespace otter {

char* err_buf() {
    static thread_local char buf[1024] = {0};
    return buf;
}

int set_err(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 1024, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace otter

extern "C" {

const char* otter_last_error(void) { return otter::err_buf(); }

int otter_version(void) { return 1; }

int otter_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return otter::set_err(OTTER_EHIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    return n;
}

}  // extern "C"
