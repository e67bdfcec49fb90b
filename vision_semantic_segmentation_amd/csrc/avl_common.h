// Shared host-side helpers of libavl_hip.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "avl_hip.h"

namespace avl {

char* err_buf();                       // thread-local, 512 bytes
int set_error(int code, const char* fmt, ...);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace avl

#define AVL_REQUIRE(cond, ...)                                         \
    do {                                                               \
        if (!(cond)) return avl::set_error(AVL_E_ARG, __VA_ARGS__);    \
    } while (0)

#define AVL_HIP_CHECK(expr)                                                                      \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return avl::set_error(AVL_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                  __FILE__, __LINE__);                                           \
    } while (0)

#define AVL_LAUNCH_CHECK() AVL_HIP_CHECK(hipGetLastError())
