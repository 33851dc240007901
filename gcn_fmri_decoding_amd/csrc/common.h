// Internal definitions shared by the libchebgcn.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include "status.h"

namespace chebgcn {

#define CG_HIP(expr)                                                                     \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess)                                                            \
            return chebgcn::fail(CHEBGCN_EHIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr, \
                                 hipGetErrorString(e_));                                 \
    } while (0)

inline int plane_stride(int M) { return (M + 31) & ~31; }

// Length-sorted sliced ELL image of one sparse operator (device memory).
// Rows are ranked by descending length; rank r lives in group r/64, lane r%64.
// Group g owns slots [goff[g], goff[g+1]) x 64 lanes, slot-major: entry (s, lane)
// of the group is at goff[g]*64 + s*64 + lane.  Padding entries have val = 0 and
// col = M (the always-zero slot of the LDS image).
struct Ell {
    int ngroups = 0;
    int max_len = 0;
    int64_t nslots = 0;          // sum of group lengths (x64 = padded entries)
    int32_t* goff = nullptr;     // [ngroups+1]
    uint16_t* col16 = nullptr;   // [nslots*64]   (only when M < 65535)
    float* val = nullptr;        // [nslots*64]
    int32_t* rowid = nullptr;    // [ngroups*64]  rank -> row, -1 for padding ranks
    // plain CSR for the out-of-LDS fallback
    int32_t* rowptr = nullptr;   // [M+1]
    int32_t* col32 = nullptr;    // [nnz]
    float* cval = nullptr;       // [nnz]
};

}  // namespace chebgcn

struct chebgcn_graph {
    int M = 0;
    int Mp = 0;
    int64_t nnz = 0;
    int device = 0;
    int num_cus = 0;
    int lds_ok = 0;              // on-chip recurrence usable (LDS image fits)
    chebgcn::Ell fwd;            // L~
    chebgcn::Ell adj;            // L~^T
};
