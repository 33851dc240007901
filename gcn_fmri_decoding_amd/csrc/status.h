// Status plumbing shared by device and host-only translation units of libchebgcn.so.
#pragma once
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/chebgcn.h"

namespace chebgcn {

// thread-local message behind chebgcn_last_error()
char* err_buf();
int fail(int code, const char* fmt, ...);

}  // namespace chebgcn

#define CG_REQUIRE(cond, ...)                                           \
    do {                                                                \
        if (!(cond)) return chebgcn::fail(CHEBGCN_EINVAL, __VA_ARGS__); \
    } while (0)
