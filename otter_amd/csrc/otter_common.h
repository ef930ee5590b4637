// otter_common.h — shared host-side helpers of libotter_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/otter_hip.h"

namespace otter {

// thread-local message behind otter_last_error()
char* err_buf();
int set_err(int code, const char* fmt, ...);

#define OTTER_HIP_TRY(expr)                                                              \
    do {                                                                                 \
        hipError_t e__ = (expr);                                                         \
        if (e__ != hipSuccess)                                                           \
            return ::otter::set_err(e__ == hipErrorOutOfMemory ? OTTER_ENOMEM : OTTER_EHIP, \
                                    "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,        \
                                    hipGetErrorString(e__));                             \
    } while (0)

// RAII: make `device` current for the scope of one C-ABI call
struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// grow-only device buffer
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int reserve(size_t need) {
        if (need <= bytes) return OTTER_OK;
        if (p) { OTTER_HIP_TRY(hipFree(p)); p = nullptr; bytes = 0; }
        size_t want = need + need / 8;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            (void)hipGetLastError();
            OTTER_HIP_TRY(hipMalloc(&p, need));
            want = need;
        }
        bytes = want;
        return OTTER_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
static inline int64_t round_up64(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

}  // namespace otter
