// Shared helpers for libemp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/emp_hip.h"

#define EMP_WAVE 64

extern thread_local char emp_err_buf[512];

#define EMP_FAIL(code, ...)                                         \
    do {                                                            \
        snprintf(emp_err_buf, sizeof(emp_err_buf), __VA_ARGS__);    \
        return (code);                                              \
    } while (0)

#define EMP_REQUIRE(cond, ...)                         \
    do {                                               \
        if (!(cond)) EMP_FAIL(EMP_EINVAL, __VA_ARGS__); \
    } while (0)

#define EMP_CHECK_LAUNCH(name)                                                         \
    do {                                                                               \
        hipError_t e_ = hipGetLastError();                                             \
        if (e_ != hipSuccess)                                                          \
            EMP_FAIL(EMP_ELAUNCH, "%s: launch failed: %s", name, hipGetErrorString(e_)); \
    } while (0)

static inline hipStream_t emp_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

static inline int64_t emp_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Memory-bound grids: cap at 256 CUs x 8 blocks and grid-stride the rest.
static inline int emp_grid(int64_t work_items, int block, int max_blocks = 2048)
{
    int64_t g = emp_cdiv(work_items, block);
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (int)g;
}
