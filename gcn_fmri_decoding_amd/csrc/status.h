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

// thread-local record behind chebgcn_last_dispatch(): the kernel templates the calling thread's last launching entry point
// enqueued.  Names must have static storage (string literals, or one string built once per template instantiation).
void note_dispatch(const char* name);         // first kernel of a call: starts a new record
void note_dispatch_more(const char* name);    // further kernels of the same call

}  // namespace chebgcn

#define CG_REQUIRE(cond, ...)                                           \
    do {                                                                \
        if (!(cond)) return chebgcn::fail(CHEBGCN_EINVAL, __VA_ARGS__); \
    } while (0)
