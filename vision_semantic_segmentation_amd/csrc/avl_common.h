// Shared host-side helpers of libavl_hip.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "avl_hip.h"

namespace avl {

char* err_buf();                       // thread-local, 512 bytes
int set_error(int code, const char* fmt, ...);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace avl

// Experiment switches (A/B schedules, timing probes whose results may be garbage).  The RELEASE library reads no environment
// variable at all: AVL_EXP_INT / AVL_EXP_STR fold to their defaults and the probe template instantiations are not compiled.
// `make experiments` builds libavl_hip_exp.so with -DAVL_EXPERIMENTS, which the tools/ scripts load through AVL_HIP_LIB
// (Python side); only there do the AVL_* variables exist.  tests/test_abi.py asserts that the release build holds no "AVL_" string.
#ifdef AVL_EXPERIMENTS
constexpr bool kStamps = true;          // in-kernel s_memtime stamp code is compiled (it runs only where a probe hands in a buffer)
#define AVL_EXP_INT(name, dflt) ([] { static const int v = getenv(name) ? atoi(getenv(name)) : (dflt); return v; }())
#define AVL_EXP_STR(name) ([] { static const char* v = getenv(name); return v; }())
#else
constexpr bool kStamps = false;
#define AVL_EXP_INT(name, dflt) (dflt)
#define AVL_EXP_STR(name) (static_cast<const char*>(nullptr))
#endif

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
