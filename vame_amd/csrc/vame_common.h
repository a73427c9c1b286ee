// Host-side helpers shared by the C-ABI translation units.
#pragma once
#include "vame_device.h"
#include "../../include/vame_hip.h"
#include <stdarg.h>
#include <stdio.h>

void vame_set_error(const char* fmt, ...);

#define VAME_CHECK_ARG(cond, code, ...)                 \
    do {                                                \
        if (!(cond)) {                                  \
            vame_set_error(__VA_ARGS__);                \
            return (code);                              \
        }                                               \
    } while (0)

#define VAME_LAUNCH_CHECK(what)                                                     \
    do {                                                                            \
        hipError_t e_ = hipGetLastError();                                          \
        if (e_ != hipSuccess) {                                                     \
            vame_set_error("%s: launch failed: %s", (what), hipGetErrorString(e_)); \
            return VAME_E_HIP;                                                      \
        }                                                                           \
    } while (0)

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
