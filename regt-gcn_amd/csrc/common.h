// Shared helpers for the regtgcn HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define REGT_OK 0
#define REGT_ERR_ARG 1
#define REGT_ERR_HIP 2

namespace regt {

void set_error(const char* fmt, ...);

#define REGT_CHECK_ARG(cond, ...)                      \
    do {                                               \
        if (!(cond)) {                                 \
            regt::set_error(__VA_ARGS__);              \
            return REGT_ERR_ARG;                       \
        }                                              \
    } while (0)

#define REGT_CHECK_HIP(expr)                                                             \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            regt::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return REGT_ERR_HIP;                                                         \
        }                                                                                \
    } while (0)

#define REGT_CHECK_LAUNCH() REGT_CHECK_HIP(hipGetLastError())

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

}  // namespace regt
