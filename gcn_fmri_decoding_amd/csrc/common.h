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

// Length-sorted sliced ELL image of one sparse operator (device memory), laid out for the
// on-chip recurrence kernel.
//  * Rows are ranked by descending length; rank r lives in group r/64, lane r%64.
//  * Vertices are renumbered into LDS *slots* by a bank-aware colouring (graph.hip): slot =
//    colour + 32*index, so vertices gathered by the same half-wave instruction tend to sit in
//    different LDS bank pairs.  `nodeslot` maps vertex -> slot for the linear passes.
//  * Group g owns quads [ginfo[g].x, +ceil(len/4)); a quad is 4 consecutive entries of each
//    of the 64 rows: colq[quad*64 + lane] = 4 packed 16-bit slot ids, valq[...] = 4 values.
//    ginfo[g].y = the group's length rounded up to even.  Padding entries have val = 0 and
//    point at `zero_slot`, an LDS entry that always holds 0.
struct Ell {
    int ngroups = 0;
    int max_len = 0;
    int64_t nquads = 0;
    int lds_entries = 0;          // float2 entries of the LDS image (incl. zero slot; multiple of 4)
    int zero_slot = 0;
    int2* ginfo = nullptr;        // [ngroups] {quad offset, even length}
    uint2* colq = nullptr;        // [nquads*64]
    float4* valq = nullptr;       // [nquads*64]
    uint16_t* rowslot = nullptr;  // [ngroups*64]  rank -> slot of that row, 0xFFFF for padding ranks
    uint16_t* nodeslot = nullptr; // [Mp + 4]      vertex -> slot, 0xFFFF for i >= M
    double est_cycles = 0;        // modelled LDS cycles per half-wave gather (1 = conflict free)
    double est_cycles_naive = 0;  // same with the identity numbering, for reference
    // plain CSR for the out-of-LDS fallback
    int32_t* rowptr = nullptr;    // [M+1]
    int32_t* col32 = nullptr;     // [nnz]
    float* cval = nullptr;        // [nnz]
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
