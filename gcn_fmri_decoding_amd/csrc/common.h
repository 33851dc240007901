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
constexpr int kQuadPad = 4;
constexpr int kQuadMin = 3;      // quads stored (zero-padded) for every group, = QMAX of recurrence.hip

// Length-sorted sliced ELL image of one sparse operator (device memory), laid out for the
// on-chip recurrence kernel (recurrence.hip).
//  * `planes` (P) = planes a workgroup carries through the recurrence at once: 4 when the
//    LDS image of the ACTIVE vertices fits with 16 B per vertex, else 2 (8 B per vertex).
//    A vertex is active when its row or its column of the operator is non-empty; with P = 4
//    only active vertices get an LDS slot (isolated ones obey T_k = -T_{k-2} and are patched
//    in by the streaming code), with P = 2 every vertex has one.  Slots are numbered
//    component-major (vertices 4q first, then 4q+1, ...; see build_ell).
//  * Ranked rows (all rows for P = 2, active rows for P = 4) are sorted by descending length;
//    rank r lives in group r/64, lane r%64.  Group g owns quads [ginfo[g].x, +ceil(len/4)); a
//    quad holds 4 consecutive entries of each of the 64 rows: colq[quad*64 + lane] = 4 packed
//    16-bit LDS slot ids, valq[quad*64 + lane] = 4 values; ginfo[g].y = the group's length
//    rounded up to even.  Padding entries have val = 0 and point at zero_slot, an LDS entry
//    that always holds 0.  Every group stores an even number of quads (>= kQuadMin); kQuadPad
//    spare quads follow the last group so that a kernel may request a fixed number per group.
struct Ell {
    int planes = 2;
    int ngroups = 0;
    int max_len = 0;
    int nranked = 0;              // rows handled by the gather
    int lds_entries = 0;          // P-float entries of the LDS image incl. the zero slot
    int zero_slot = 0;
    int64_t nslots = 0;           // sum of (even) group lengths
    int64_t nquads = 0;
    int64_t cost_before = 0, cost_after = 0, cost_ideal = 0;   // bank-conflict statistics of build_ell (graph_query 9, 10)
    int iso_max512 = 0;           // P = 4: most isolated vertices (inside the graph, no slot) among the linear pieces of one thread
                                  // of a 512-thread workgroup (piece q belongs to thread q % 512); recurrence4.hip keeps up to 4 in registers
    int2* ginfo = nullptr;        // [ngroups] {quad offset, even length}
    uint2* colq = nullptr;        // [(nquads + kQuadPad)*64]
    uint4* colo = nullptr;        // [(nquads + kQuadPad)/2*64]  the same ids, quads 2o and 2o+1 in one record
    float4* valq = nullptr;       // [(nquads + kQuadPad)*64]
    float4* valp = nullptr;       // the same, with the two ids of a short third quad in .z / .w (graph.hip)
    // Fixed-stride image of the first 12 entries of every row group (P = 4, recurrence4.hip): offsets are compile-time
    // constants of the group's slot, so the gather needs no per-group table lookups.  uval[g*4 + r][lane]: r = 0, 1 the
    // values of quads 0 and 1; r = 2 {value 8, value 9, slot ids (8, 9), slot ids (10, 11)}; r = 3 {value 10, value 11, 0, 0};
    // uids[g][lane] = the eight 16-bit slot ids of quads 0 and 1.  Entries beyond 12 stay in colo / valq.
    float4* uval = nullptr;       // [ngroups*4*64]
    uint4* uids = nullptr;        // [ngroups*64]
    // Ordered image (recurrence_ord.hip; graphs whose rows come sorted by descending length): thread t of a workgroup of
    // ord_NT threads owns ord_NQ vertex quads (rows 4q..4q+3, the same 16 bytes it loads and stores per plane),
    // vertex v has LDS slot (v & 3)*ord_SQ + (v >> 2) while (v >> 2) < ord_SQ (an entry = `planes` floats: 4 or 2); group (4u + i)*(ord_NT/64) + w, lane l is row
    // 4*(64*blkmap[w*ord_NQ + u] + l) + i.  Only ginfo / colo / valq / uval / uids / blkmap are built.
    // Which 64-quad block a wave works on at level u is a table (blkmap[w*ord_NQ + u], ascending in u): the rows are sorted,
    // so block b holds longer rows than block b + 1 -- dealt out in order, wave 0 would get the longest rows of every level
    // (measured: its gather 27k cycles against 20-22k for the others, and every step waits for it); the host balances the
    // gather cost of the waves instead.
    int ord_NT = 0, ord_NQ = 0, ord_NG = 0, ord_SQ = 0;
    // first vertex the ordered kernel's ord_NQ quad levels do NOT reach (more than one level of isolated / padding vertices behind
    // the rows: the fake vertices of a deep coarsening); 0 = none.  [ord_tail, Mp) holds no rows: cheb_ord_tail_kernel streams it
    int ord_tail = 0;
    int32_t* blkmap = nullptr;    // [NT/64 * ord_NQ]
    uint16_t* rowslot = nullptr;  // [ngroups*64]  rank -> LDS slot of that row, 0xFFFF for padding ranks
    uint16_t* nodeslot = nullptr; // [Mp + 4]      vertex -> LDS slot, 0xFFFF = none (isolated / pad)
    // plain CSR for the out-of-LDS fallback
    int32_t* rowptr = nullptr;    // [M+1]
    int32_t* col32 = nullptr;     // [nnz]
    float* cval = nullptr;        // [nnz]
    // atlas-sized graphs (fused_small.hip): one 128-byte record per vertex, [Mp][32] dwords -- 0..9 the row's neighbour vertices
    // two per dword (16 bits each, caller's entry order), 10 the row length, 12..31 the values -- so that a lane has its whole
    // operator row after ONE memory round trip (through the row pointers it takes two); NULL where rows are longer than 20 entries
    uint32_t* fs_rec = nullptr;
};

// what the on-chip kernels take by value
struct EllView {
    const int2* ginfo;
    const uint2* colq;
    const uint4* colo;
    const float4* valq;
    const float4* valp;
    const uint16_t* rowslot;
    const uint16_t* nodeslot;
    int ngroups, zero_slot;
    const float4* uval;
    const uint4* uids;
    const int32_t* blkmap;
};

static inline EllView view(const Ell& e) {
    return EllView{e.ginfo, e.colq, e.colo, e.valq, e.valp, e.rowslot, e.nodeslot, e.ngroups, e.zero_slot, e.uval, e.uids, e.blkmap};
}

}  // namespace chebgcn

struct chebgcn_graph;
namespace chebgcn {
// recurrence.hip: the generic on-chip kernel carries four planes for graphs of at most 2048 ranked
// rows and 768 linear pieces (its largest 256-thread shapes: 8 rows and 3 pieces per thread)
inline bool generic4_fits(int rows, int Mq) { return rows <= 2048 && Mq <= 768; }
// recurrence4.hip: four-plane kernel beyond
bool onchip4_fits(int lds_entries, int rows, int Mq);
template <bool ADJ>
int dispatch_onchip4(const chebgcn_graph* g, const Ell& ell, const float* src, float* dst, int nplanes, int K, int copy_t0,
                     hipStream_t stream);
// recurrence_ord.hip: shape {NQ, NG, planes per workgroup} of the ordered kernel for a graph of Mq vertex quads of which the
// first SQ have rows; false = not served.  NQ * NT may be less than Mq (NQ is capped at NG + 1): the vertices behind are the tail.  ordered_fits: the launch is addressable by the ordered kernels
// (one 32-bit buffer descriptor per slab), else the caller takes the regular images
bool ordered_shape(int Mq, int SQ, int* NT, int* NQ, int* NG, int* planes);
bool ordered_fits(const chebgcn_graph* g, int nplanes);
template <bool ADJ>
int dispatch_ordered(const chebgcn_graph* g, const Ell& ell, const float* src, float* dst, int nplanes, int K, int copy_t0,
                     hipStream_t stream);
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
    // Automatic plane choice on a big graph (four planes, recurrence4.hip): the two-plane images as well.
    // Four planes win once a workgroup has several plane groups to work through (batch 256 at Fin = 32:
    // 8 per CU); with one or two (batch 64) the two-plane kernel's lighter group turn-over wins.
    int has_alt2 = 0;
    chebgcn::Ell fwd2, adj2;
    // Rows (of L~ and of L~^T) sorted by descending length and a shape recurrence_ord.hip serves: the ordered images.  The
    // automatic plane choice then runs every launch on them (one 16-byte piece of a plane = the four rows a thread owns:
    // planes go from HBM to registers and back without a pass through LDS).
    int ord_ok = 0;
    chebgcn::Ell ofwd, oadj;
};

#ifndef CG_PICK4_GROUPS_PER_CU
#define CG_PICK4_GROUPS_PER_CU 4     // four planes per workgroup from this many plane groups per CU (pick_ell)
#endif
namespace chebgcn {
// the operator image a launch over `nplanes` planes uses
// (`adjoint`: the image of L~^T; `adjoint_kernel`: the Clenshaw kernel runs on it -- chebgcn_recurrence_fwd_t runs the FORWARD
// kernel on the image of L~^T)
inline const Ell& pick_ell(const chebgcn_graph* g, bool adjoint, int nplanes, int adjoint_kernel = -1) {
    const Ell& e = adjoint ? g->adj : g->fwd;
    const bool adjk = adjoint_kernel < 0 ? adjoint : adjoint_kernel != 0;
    if (!g->has_alt2 || e.planes != 4) return e;
    // Beyond 10752 vertices the two-plane kernel has only its 512-thread shapes with 24..40 rows per thread (0.23-0.26 of the
    // HBM roofline at any launch size): four planes win -- except in the forward direction on a graph with more than four
    // isolated vertices per thread (iso_max512: the coarsening's fake vertices), where the four-plane kernel patches them in
    // from memory once per order.  Measured on the level-0 graph of the six-level pooling network (M = 12672, 2672 fake
    // vertices, batch 64, K = 20 and 10): forward 0.97 ms on two planes, 1.49 ms on four; adjoint 0.61 ms on two, 0.43 on four.
    // Small launches (fewer plane groups than CUs: predict() tails, B*Fin/4 < 256) keep the two-plane image there as well --
    // measured in round 6 at K = 10, Fin = 32 (profiles/r06_pick_ell_small_launches.txt): adjoint at batch 2 / 4 / 8 0.155 / 0.163 /
    // 0.169 ms on four planes against 0.143 / 0.146 / 0.152 ms on two; batch 1 the same either way.
    if (g->M > 10752)
        return ((adjk || e.iso_max512 <= 4) && (nplanes + 3) / 4 >= g->num_cus) ? e : (adjoint ? g->adj2 : g->fwd2);
    // few groups per CU: the two-plane image keeps more workgroups in flight.  Measured in the configs[1] step
    // (M = 10466, 2048 planes per launch): both directions on two planes 4.235 ms, adjoint on four 4.27, both on
    // four 4.37 -- although the isolated adjoint launch is 5 % faster on four planes (tools/kbench.py)
    if ((nplanes + 3) / 4 < CG_PICK4_GROUPS_PER_CU * g->num_cus) return adjoint ? g->adj2 : g->fwd2;
    return e;
}
}  // namespace chebgcn
