// Chebyshev recurrence on chip for graphs whose vertices come SORTED BY DESCENDING ROW LENGTH (gfx950).
//
//   forward  (lib_new/models_gcn.py:598-610):  T_0 = x, T_1 = L T_0, T_k = 2 L T_{k-1} - T_{k-2}
//   adjoint  (TF autodiff of the above):       c_{K-1} = G_{K-1}, c_j = G_j + 2 L^T c_{j+1} - c_{j+2},
//                                              dx = G_0 + L^T c_1 - c_2
//
// Same algorithm and operator records as recurrence4.hip (four planes per workgroup, T_{k-1} of the ACTIVE vertices in
// LDS, T_{k-2} -> T_k of the own rows in registers, the operator streamed once per step through a ring of records).  What
// is different is WHO OWNS WHICH ROWS.  The kernels of recurrence.hip / recurrence4.hip hand rows to lanes in length order
// (64 rows of equal length per wave instruction) while planes travel in vertex order, so every plane piece crosses LDS
// between the two orders: the input is scattered into the image through a slot table, every slab is gathered back out of
// it (K-1 passes over the image per plane group), the adjoint adds G_j in a pass of its own behind an extra barrier, and
// isolated vertices (no LDS slot) are patched in by separate code.  Phase stamps of round 3: of a plane group's 131k cycles
// (forward, K = 5) 70k are the four gathers; the rest is that traffic.
//
// When the vertex order IS the length order (the Python host relabels the graph once: graph.length_order, P L P^T, and
// keeps activations, per-vertex biases and the first FC layer's rows in that order), one 16-byte piece of a plane -- four
// consecutive vertices -- is four rows of (nearly) equal length.  Thread t owns the quads q = u*512 + t:
//   * the four rows of a quad are four "slices" of the gather (slice 4u + i = rows 4q + i of the 64 lanes of a wave: 64
//     rows that span 256 consecutive ranks, still of equal length up to the class boundaries);
//   * a plane piece is loaded straight into the registers that hold the state of its own rows and stored straight from
//     them: slab k leaves as soon as the gather has produced it, G_j is added in registers, no pass over the image;
//   * LDS sees the image only: staged once per plane group, rotated once per step (own slots, conflict-free);
//   * vertex v has slot (v & 3)*SQ + (v >> 2) (component-major: for a fixed row-in-quad the lanes of a wave touch
//     consecutive entries), computed, not looked up: no slot tables, no id registers;
//   * an isolated vertex (empty row and column: sorted behind all others, no slot) keeps x in its state registers and goes
//     out as T_k = 0 for odd k, (-1)^(k/2) x for even k; adjoint: the registers accumulate dx = G_0 - G_2 + G_4 - ...
//
// PL = planes a workgroup carries at once = floats per LDS entry: 4 while the image of the active vertices fits with 16 bytes
// each (up to 10238 active vertices), 2 beyond (8 bytes each, up to 20478).  A plane PIECE is 16 bytes -- four consecutive
// vertices of one plane -- for either; an ENTRY is the PL planes of one vertex.
//
// This header holds the kernel template; recurrence_ord.hip (PL = 4) and recurrence_ord2.hip (PL = 2) instantiate it.
#pragma once
#include <string>
#include <type_traits>

#include "common.h"

namespace chebgcn {

extern int g_stagger;

#ifndef CG_ORD_WG2
#define CG_ORD_WG2 1         // cap the registers of the smallest shape so that two workgroups share a CU (adjoint: 20 spilled registers,
                             // 0.202 -> 0.186 ms at N = 2600 and 0.229 -> 0.218 ms at N = 3800, batch 256; forward fits anyway)
#endif
#ifndef CG_ORD_RING4
#define CG_ORD_RING4 2       // operator ring depth (slices ahead) of the four-plane shapes: what the registers allow
#endif
#ifndef CG_ORD_RING2
#define CG_ORD_RING2 2       // ... of the two-plane shapes
#endif
#ifndef CG_ORD_PRIO
#define CG_ORD_PRIO 1
#endif
#ifndef CG_X
#define CG_X 0               // 64: in-kernel phase stamps (tools/vbuild.sh, tools/kbench.py --stamps); 0 in production
#endif
static __device__ long long g_dbgo[16 * 64];     // (one per translation unit)
#define CG_STAMP(id)                                                                          \
    do {                                                                                      \
        if ((CG_X & 64) && (id) < 64 && lane == 0 && blockIdx.x == 37 && grp == (int)(blockIdx.x + (ngrp > (int)gridDim.x ? gridDim.x : 0))) \
            g_dbgo[wave * 64 + (id)] = (long long)__builtin_readcyclecounter();               \
    } while (0)

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

// plane accesses: one descriptor per slab (SGPRs), a uniform byte offset (plane + piece run) and ONE per-thread offset
// register; streaming (nt): planes are read once and written once and must not push the operator out of the XCD's L2
__device__ __forceinline__ rsrc_t slab_rsrc(const float* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
}
#ifndef CG_ORD_ST_AUX
#define CG_ORD_ST_AUX 2      // cache policy of the slab stores (2 = nt); experiment: 0 (do they stay in the Infinity Cache for the contraction?)
#endif
__device__ __forceinline__ float4 ldp(rsrc_t r, unsigned voff, unsigned soff) {
    const f32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 2);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void stp(rsrc_t r, unsigned voff, unsigned soff, float4 v) {
    const f32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(t, r, voff, soff, CG_ORD_ST_AUX);
    // a 16-byte buffer store with an SGPR offset still reads its data registers while the following instructions issue
    // (hipcc pads the hazard for the immediate-offset form only): the asm keeps the data live across the wait states
    asm volatile("s_nop 1" : : "v"(t) : "memory");
}
// byte offset of the LDS entry (16 bytes for PL = 4, 8 for PL = 2) named by the low / high 16 bits of w
template <int PL>
__device__ __forceinline__ unsigned ofs_lo(unsigned w) {
    unsigned r;
    if constexpr (PL == 4)
        asm("v_lshlrev_b32_sdwa %0, 4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(w));
    else
        asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(w));
    return r;
}
template <int PL>
__device__ __forceinline__ unsigned ofs_hi(unsigned w) {
    unsigned r;
    if constexpr (PL == 4)
        asm("v_lshlrev_b32_sdwa %0, 4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(w));
    else
        asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(w));
    return r;
}
__device__ __forceinline__ float comp(const float4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
__device__ __forceinline__ float comp(const float2& v, int i) { return i == 0 ? v.x : v.y; }
// an LDS entry / the state of one row: its PL planes
template <int PL> struct entry_of;
template <> struct entry_of<4> { typedef float4 type; };
template <> struct entry_of<2> { typedef float2 type; };
// uniform values through an opaque identity: what is derived from them is recomputed at its use instead of being hoisted out
// of the step loop into (spilled) SGPRs
__device__ __forceinline__ unsigned opaque_v(unsigned x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ float4 opaque_v(float4 x) { asm volatile("" : "+v"(x.x), "+v"(x.y), "+v"(x.z), "+v"(x.w)); return x; }
__device__ __forceinline__ float2 opaque_v(float2 x) { asm volatile("" : "+v"(x.x), "+v"(x.y)); return x; }
__device__ __forceinline__ unsigned opaque_s(unsigned x) { asm volatile("" : "+s"(x)); return x; }
__device__ __forceinline__ int opaque_s(int x) { asm volatile("" : "+s"(x)); return x; }
__device__ __forceinline__ float4 fma4(float s, const float4& t, const float4& a) {
    return make_float4(fmaf(s, t.x, a.x), fmaf(s, t.y, a.y), fmaf(s, t.z, a.z), fmaf(s, t.w, a.w));
}
__device__ __forceinline__ float2 fma4(float s, const float2& t, const float2& a) { return make_float2(fmaf(s, t.x, a.x), fmaf(s, t.y, a.y)); }
__device__ __forceinline__ float4 sel4(bool c, const float4& a, const float4& b) {
    return make_float4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w);
}
__device__ __forceinline__ float2 sel4(bool c, const float2& a, const float2& b) { return make_float2(c ? a.x : b.x, c ? a.y : b.y); }
__device__ __forceinline__ float4 scale4(float s, const float4& a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }
__device__ __forceinline__ float2 scale4(float s, const float2& a) { return make_float2(s * a.x, s * a.y); }
__device__ __forceinline__ float4 sub4(const float4& a, const float4& b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float2 sub4(const float2& a, const float2& b) { return make_float2(a.x - b.x, a.y - b.y); }
// f*acc + sg*old per plane
__device__ __forceinline__ float4 step4(float f, const float4& acc, float sg, const float4& old) {
    return make_float4(fmaf(f, acc.x, sg * old.x), fmaf(f, acc.y, sg * old.y), fmaf(f, acc.z, sg * old.z), fmaf(f, acc.w, sg * old.w));
}
__device__ __forceinline__ float2 step4(float f, const float2& acc, float sg, const float2& old) {
    return make_float2(fmaf(f, acc.x, sg * old.x), fmaf(f, acc.y, sg * old.y));
}
template <class E> __device__ __forceinline__ E zero_entry();
template <> __device__ __forceinline__ float4 zero_entry<float4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }
template <> __device__ __forceinline__ float2 zero_entry<float2>() { return make_float2(0.f, 0.f); }
// pieces <-> entries of one quad: m[p] = the four rows of plane p  ->  the PL planes of row i ...
__device__ __forceinline__ float4 row_of(const float4 (&m)[4], int i) { return make_float4(comp(m[0], i), comp(m[1], i), comp(m[2], i), comp(m[3], i)); }
__device__ __forceinline__ float2 row_of(const float4 (&m)[2], int i) { return make_float2(comp(m[0], i), comp(m[1], i)); }
// ... and e[i] = the PL planes of row i  ->  the four rows of plane p
template <class E>
__device__ __forceinline__ float4 piece_of(const E (&e)[4], int p) { return make_float4(comp(e[0], p), comp(e[1], p), comp(e[2], p), comp(e[3], p)); }

// Workgroup barrier for the LDS image only.  __syncthreads() also waits for every vector-memory operation of the wave
// (s_waitcnt vmcnt(0)): the slab stores issued during the gather and the plane requests issued in front of the barrier would
// have to complete before a wave may even ARRIVE -- the HBM latency and the drain of 168 KB of stores per step exposed at
// every barrier.  Here only the LDS operations are waited for; plane traffic stays in flight across the barrier.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int FIRST, int LAST, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (FIRST < LAST) {
        f(std::integral_constant<int, FIRST>{});
        static_for<FIRST + 1, LAST>(f);
    }
}

// PL = planes per workgroup = floats per LDS entry (4 or 2); ENT = LDS entries; NQ = vertex quads per thread
// (ceil(Mp/4 / NT)); NG = leading quad levels that hold rows (ceil(SQ / NT)): levels NG..NQ-1 are isolated vertices and
// padding only.
template <int PL, int ENT, int NQ, int NG, int NT, bool ADJ>
// (two workgroups per CU where two images fit; HIP's second launch bound counts waves per SIMD; the 256-thread shapes of the
// graphs below 2049 vertices: four workgroups per CU)
__global__ void __launch_bounds__(NT, NT <= 256 ? 4 : (CG_ORD_WG2 && NQ == NG && 2 * ENT * 4 * PL <= 160 * 1024 && 2 * NT <= 1024) ? 2 * NT / 256 : NT / 256)
cheb_ord_kernel(EllView e, const float* __restrict__ src, float* __restrict__ dst, int M, int Mp, int nplanes, int K, size_t slab,
                int SQ, int flags) {
    typedef typename entry_of<PL>::type ent_t;
    __shared__ ent_t T[ENT];                         // slot-indexed: the PL planes of one vertex
    constexpr int NW = NT / 64;
    constexpr int NJ = 4 * NG;                       // gather slices per thread
    static_assert((PL == 4 || PL == 2) && NG >= 1 && NG <= NQ && NJ <= 64, "shape");
    const int copy_t0 = flags & 1;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Mq = Mp >> 2;
    const int ngrp = (nplanes + PL - 1) / PL;
    const ent_t zero4 = zero_entry<ent_t>();
    const float4 zero_piece = make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned zslot = (unsigned)e.zero_slot;    // 4*SQ; zslot + 1 = trash (written, never read as data)
    auto lds = [&](unsigned byteoff) -> ent_t {
        return *reinterpret_cast<const ent_t*>(reinterpret_cast<const char*>(T) + byteoff);
    };
    // own quads: level u of this wave is block blk[u] (64 quads, common.h `blkmap`): q(u) = 64 blk[u] + lane.  A quad has LDS
    // slots while q < SQ; the host puts the only block of a wave that may not be full of rows at its level NG - 1.
    int blk[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) blk[u] = __builtin_amdgcn_readfirstlane(e.blkmap[wave * NQ + u]);
    // (the lane goes through an opaque identity at every use: slot addresses, range tests and piece offsets derived from it are
    // loop invariant, and hipcc would otherwise keep some sixty of them in registers across the whole kernel -- spills)
    auto quad_of = [&](int u) __attribute__((always_inline)) -> int { return blk[u] * 64 + (int)opaque_v((unsigned)lane); };
    auto has_slot = [&](int u) __attribute__((always_inline)) -> bool { return u < NG - 1 || (u < NG && quad_of(u) < SQ); };
    auto own_slot = [&](int u, int i, bool wr) __attribute__((always_inline)) -> unsigned {
        return has_slot(u) ? (unsigned)(i * SQ + quad_of(u)) : zslot + (wr ? 1u : 0u);
    };

    int2 gtab = make_int2(0, 0);                     // lane j: {quad offset, length} of group j*NW + wave
    if (lane < NJ) gtab = e.ginfo[lane * NW + wave];
    // slices of a wave are sorted by length: the first nB have more than 10 entries, the first nA more than 8; bit j of mC:
    // slice j has more than 12 (uniform: SGPRs)
    const int nA = __popcll(__ballot(lane < NJ && gtab.y > 8));
    const int nB = __popcll(__ballot(lane < NJ && gtab.y > 10));
    const unsigned long long mC64 = __ballot(lane < NJ && gtab.y > 12);
    const unsigned mC = (unsigned)mC64, mC_hi = (unsigned)(mC64 >> 32);      // (slices 32.. exist for NG > 8 only)
    const rsrc_t uval_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)e.uval, 0, 0x7FFFFFFF, 0x00020000);
    const rsrc_t uids_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)e.uids, 0, 0x7FFFFFFF, 0x00020000);
    if (tid == 0) T[zslot] = zero4;                  // never written again
    {   // one-off stagger of the workgroups of an XCD (see recurrence.hip)
        const int gpw = (ngrp + (int)gridDim.x - 1) / (int)gridDim.x;
        const int sx = (flags >> 20) & 0xFF;
        const int m8 = gpw <= 4 ? 8 : gpw >= 16 ? 16 : 8 + (8 * (gpw - 4)) / 12;
        const int reps = ((blockIdx.x >> 3) & 31) * (sx ? sx - 1 : m8) / 8;
        for (int i = 0; i < reps; ++i) __builtin_amdgcn_s_sleep(10);
    }
    const unsigned slab_bytes = (unsigned)(slab * sizeof(float));     // < 4 GB (checked by the dispatcher)
    const unsigned vb = (unsigned)lane * 16u;
    auto plane_off = [&](int g, int p, int u) __attribute__((always_inline)) -> unsigned {   // uniform per wave: plane p of group g, level u
        const int i = g * PL + p;
        // (a partial last group computes its missing planes as copies of the last one and stores them to the same addresses)
        // (opaque: the 24 sums of a group would otherwise sit in SGPRs across the gather, spilled to VGPR lanes)
        return (unsigned)(i < nplanes ? i : nplanes - 1) * (unsigned)Mp * 4u + opaque_s((unsigned)blk[u]) * 1024u;
    };
    auto in_range = [&](int u) __attribute__((always_inline)) -> bool { return quad_of(u) < Mq; };
    // the PL planes of quad level u of group g: d[p] = rows 4q..4q+3 of plane p (zeros beyond the plane)
    auto load_level = [&](rsrc_t rs, int g, int u, float4 (&d)[PL]) __attribute__((always_inline)) {
        const unsigned vo = in_range(u) ? vb : 0x80000000u;                 // out of range: no memory access, zeros
#pragma unroll
        for (int p = 0; p < PL; ++p) d[p] = ldp(rs, vo, plane_off(g, p, u));
    };
    // quad level u of group g from the state of its four rows (r0..r3 = the PL planes of rows 4q..4q+3)
    auto store_level = [&](rsrc_t rs, int g, int u, const ent_t& r0, const ent_t& r1, const ent_t& r2, const ent_t& r3)
                           __attribute__((always_inline)) {
        if (in_range(u)) {
            const ent_t rows[4] = {r0, r1, r2, r3};
#pragma unroll
            for (int p = 0; p < PL; ++p) stp(rs, vb, plane_off(g, p, u), piece_of(rows, p));
        }
    };
    // coefficient of slab m in what an isolated vertex holds: T_m = c(m) x (forward), dx = sum_m c(m) G_m (adjoint)
    auto iso_coef = [&](int m) -> float { return (m & 1) ? 0.f : ((m & 2) ? -1.f : 1.f); };

    ent_t st[4 * NQ];                                // rows with a slot: T_{k-2} -> T_k; isolated: x (adjoint: the running dx)
    // Plane pieces on their way in: the next group's input, G_j of the adjoint.  HBM requests are issued by a wave when its own
    // gather is over and its operator ring is empty -- loads return in order, so a plane request in front of an operator
    // request holds the ring up for a whole memory latency (measured with G_j requested level by level inside the gather:
    // 41k cycles per step instead of 27k) -- and their latency runs under the wait for the slower waves and the rotate.
    float4 gin[NQ][PL];
    // adjoint: G_{K-2} of the NEXT group, requested with its G_{K-1} when a group ends (behind the dx stores, whose registers
    // it takes over): the first step of a group then starts without a memory round trip of its own
    float4 gin2[ADJ ? NQ : 1][PL];
    auto request_into = [&](float4 (*d)[PL], const float* base, int g) __attribute__((always_inline)) {
        const rsrc_t rs = slab_rsrc(base, slab_bytes);
#pragma unroll
        for (int u = 0; u < NQ; ++u) load_level(rs, g, u, d[u]);
    };
    auto request_in = [&](const float* base, int g) __attribute__((always_inline)) { request_into(gin, base, g); };
    auto clear = [&](float4 (*d)[PL]) __attribute__((always_inline)) {     // ends a live range (see the group end)
#pragma unroll
        for (int u = 0; u < NQ; ++u)
#pragma unroll
            for (int p = 0; p < PL; ++p) d[u][p] = zero_piece;
    };
    // adjoint: c_j = G_j + f L^T c_{j+1} - c_{j+2}: the state takes G_j - c_{j+2} when G_j arrives, the gather adds the rest
    // (isolated rows: dx += c(j) G_j)
    auto consume = [&](float4 (*d)[PL], int jm) __attribute__((always_inline)) {
        const float ck = iso_coef(jm);
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const bool hs = has_slot(u);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const ent_t gr = row_of(d[u], i), old = st[4 * u + i];
                const ent_t a = sub4(gr, old), b = fma4(ck, gr, old);
                st[4 * u + i] = u < NG - 1 ? a : u < NG ? sel4(hs, a, b) : b;
            }
        }
    };

    // ---- a plane group's input (in `gin`) -> LDS image + row state ------------------------------------------------------
    // forward: x; adjoint: G_{K-1} = c_{K-1}
    auto stage = [&](int g) __attribute__((always_inline)) {
        const rsrc_t rs_t0 = slab_rsrc(dst, slab_bytes);
        const float c0 = ADJ ? iso_coef(K - 1) : 1.f;
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            if (!ADJ && copy_t0 && in_range(u)) {                          // T_0 = x goes straight to slab 0
#pragma unroll
                for (int p = 0; p < PL; ++p) stp(rs_t0, vb, plane_off(g, p, u), gin[u][p]);
            }
            const bool hs = has_slot(u);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const ent_t ent = row_of(gin[u], i);                        // the PL planes of row 4q + i
                if (u < NG) T[own_slot(u, i, true)] = ent;
                st[4 * u + i] = (u < NG - 1) ? zero4 : sel4(hs, zero4, ADJ ? scale4(c0, ent) : ent);
            }
        }
    };

    int grp = blockIdx.x;
    const float* in0 = src + (ADJ ? (size_t)(K - 1) * slab : 0);
    if (grp < ngrp) {
        request_in(in0, grp);
        if (ADJ) request_into(gin2, src + (size_t)(K - 2) * slab, grp);    // (K >= 2 here)
    }
    lds_barrier();                                    // the zero slot
    for (; grp < ngrp; grp += gridDim.x) {
        CG_STAMP(0);
        stage(grp);
        if (ADJ) {
            consume(gin2, K - 2);
            clear(gin2);
        }
        CG_STAMP(1);
        lds_barrier();                                // the image of this group is complete
        CG_STAMP(2);
        for (int step = 1; step < K; ++step) {
            const int jm = K - 1 - step;             // adjoint: this step computes c_jm
            const bool last = step == K - 1;
            const float f = ADJ ? (last ? 1.f : 2.f) : (step == 1 ? 1.f : 2.f);
            const rsrc_t rs_out = slab_rsrc(ADJ ? dst : dst + (size_t)step * slab, slab_bytes);
            const float ck = iso_coef(ADJ ? jm : step);
            __builtin_amdgcn_sched_barrier(0);      // (what precedes stays in front of the gather: its registers are the ring's)
            CG_STAMP(3);

            // ---- gather: st <- f * (A T_{k-1})[own rows] - st  (adjoint: + st, which holds G_j - c_{j+2}) ------------------
            // Operator records of slice j (group j*NW + wave) sit at compile-time offsets behind one per-wave base; they
            // travel through a ring two slices deep: values of quads 0 / 1, the record of entries 8..11, the eight ids of
            // quads 0 and 1.  The two optional requests are skipped by scalar-only tests (j is a constant, nA / nB SGPRs).
            constexpr int RD = PL == 4 ? CG_ORD_RING4 : CG_ORD_RING2;      // slices of operator records in flight per wave
            float4 uq[3][RD];
            float2 ub[RD];
            uint4 uo[RD];
            const unsigned vsoff = (unsigned)wave * 4096u, isoff = (unsigned)wave * 1024u;
            auto urequest = [&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                constexpr unsigned vo = (unsigned)j * NW * 4096u, io = (unsigned)j * NW * 1024u;
                const unsigned vs = opaque_s(vsoff), is = opaque_s(isoff);
                const f32x4 a = __builtin_amdgcn_raw_buffer_load_b128(uval_rsrc, lane * 16, vs + vo, 0);
                const f32x4 b = __builtin_amdgcn_raw_buffer_load_b128(uval_rsrc, lane * 16, vs + (vo + 1024u), 0);
                const u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(uids_rsrc, lane * 16, is + io, 0);
                uq[0][j % RD] = make_float4(a.x, a.y, a.z, a.w);
                uq[1][j % RD] = make_float4(b.x, b.y, b.z, b.w);
                uo[j % RD] = make_uint4(c.x, c.y, c.z, c.w);
                if (j < opaque_s(nA)) {
                    const f32x4 d = __builtin_amdgcn_raw_buffer_load_b128(uval_rsrc, lane * 16, vs + (vo + 2048u), 0);
                    uq[2][j % RD] = make_float4(d.x, d.y, d.z, d.w);
                }
                if (__builtin_expect(j < opaque_s(nB), 0)) {
                    const f32x2 h = __builtin_amdgcn_raw_buffer_load_b64(uval_rsrc, lane * 16, vs + (vo + 3072u), 0);
                    ub[j % RD] = make_float2(h.x, h.y);
                }
            };
            auto lds_quad = [&](const uint2 c, const float4 v, ent_t& acc) __attribute__((always_inline)) {
                const unsigned a0 = ofs_lo<PL>(c.x), a1 = ofs_hi<PL>(c.x), a2 = ofs_lo<PL>(c.y), a3 = ofs_hi<PL>(c.y);
                const ent_t t0 = lds(a0), t1 = lds(a1), t2 = lds(a2), t3 = lds(a3);
                acc = fma4(v.x, t0, acc);
                acc = fma4(v.y, t1, acc);
                acc = fma4(v.z, t2, acc);
                acc = fma4(v.w, t3, acc);
            };
            urequest(std::integral_constant<int, 0>{});
            static_for<1, (RD < NJ ? RD : NJ)>([&](auto jc) { urequest(jc); });
            // levels without rows: nothing to gather, their pieces go out right away (forward: c(step) x; adjoint: dx)
            static_for<NG, NQ>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                if (!ADJ) {
                    if (!last)
                        store_level(rs_out, grp, u, scale4(ck, st[4 * u]), scale4(ck, st[4 * u + 1]), scale4(ck, st[4 * u + 2]),
                                    scale4(ck, st[4 * u + 3]));
                }
            });
            static_for<0, NJ>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                constexpr int u = j >> 2, i = j & 3;
                // waves that are ahead yield to the ones behind (see recurrence.hip).  The two waves of a SIMD tie at equal
                // progress and the arbiter then serves the older one: waves 0-3 ended every gather 4-5k cycles ahead of waves
                // 4-7, and the step waited for those.  The younger wave of a SIMD therefore stays one level above the older
                // one at equal progress (CG_ORD_PRIO).
                if constexpr (j == 0 || (4 * j) / NJ != (4 * (j - 1)) / NJ) {
                    constexpr int stage = (4 * j) / NJ;
                    const int pr = (CG_ORD_PRIO && wave >= NW / 2) ? (stage == 0 ? 3 : 4 - stage) : 3 - stage;
                    if (pr == 3) __builtin_amdgcn_s_setprio(3);
                    else if (pr == 2) __builtin_amdgcn_s_setprio(2);
                    else if (pr == 1) __builtin_amdgcn_s_setprio(1);
                    else __builtin_amdgcn_s_setprio(0);
                }
                const uint4 o01 = uo[j % RD];
                const float4 v0 = uq[0][j % RD], v1 = uq[1][j % RD], v2 = uq[2][j % RD];
                const float2 v3 = ub[j % RD];
                ent_t acc = zero4;
                lds_quad(make_uint2(o01.x, o01.y), v0, acc);
                lds_quad(make_uint2(o01.z, o01.w), v1, acc);
                // (the optional records pass through an opaque identity at their use: hipcc otherwise hoists the copy that
                // prepares a component as a v_pk_fma operand into the block of the CONDITIONAL request two slices earlier,
                // with `s_waitcnt vmcnt(0)` right behind the load -- the whole ring drained at every slice)
                if (j < opaque_s(nA)) {
                    const float4 w2 = opaque_v(v2);
                    const unsigned c4 = __float_as_uint(w2.z);
                    const ent_t t0 = lds(ofs_lo<PL>(c4)), t1 = lds(ofs_hi<PL>(c4));
                    acc = fma4(w2.x, t0, acc);
                    acc = fma4(w2.y, t1, acc);
                    if (j < opaque_s(nB)) {
                        const float2 w3 = opaque_v(v3);
                        const unsigned c5 = __float_as_uint(w2.w);
                        const ent_t t2 = lds(ofs_lo<PL>(c5)), t3 = lds(ofs_hi<PL>(c5));
                        acc = fma4(w3.x, t2, acc);
                        acc = fma4(w3.y, t3, acc);
                    }
                }
                if constexpr (j + RD < NJ) urequest(std::integral_constant<int, j + RD>{});    // refill the ring slots just consumed
                // rows beyond 12 entries (rare; sorted: the first slices of a wave): their further quads from the
                // variable-stride image
                if ((opaque_s(j < 32 ? mC : mC_hi) >> (j & 31)) & 1u) {
                    const int qoff = __builtin_amdgcn_readlane(gtab.x, j), len = __builtin_amdgcn_readlane(gtab.y, j);
                    for (int q = 3; 4 * q < len; ++q) {
                        const uint4 o = e.colo[(size_t)((qoff >> 1) + (q >> 1)) * 64 + lane];
                        const float4 v = e.valq[(size_t)(qoff + q) * 64 + lane];
                        lds_quad((q & 1) ? make_uint2(o.z, o.w) : make_uint2(o.x, o.y), v, acc);
                    }
                }
                const ent_t old = st[j];
                const float sg = ADJ ? 1.f : -1.f;
                ent_t nw = step4(f, acc, sg, old);
                if constexpr (u == NG - 1) nw = sel4(has_slot(u), nw, old);         // isolated rows keep their state
                st[j] = nw;
                // ---- the level's pieces go out as soon as its four slices are done: slab `step` (forward), dx (adjoint) ----
                if constexpr (i == 3) {
                    if (!ADJ && !last) {       // (the last slab / dx: behind the next group's request, see the group end)
                        if constexpr (u == NG - 1 && !ADJ) {
                            // mixed level: isolated rows hold x and go out as c(step) x
                            const float sc = has_slot(u) ? 1.f : ck;
                            store_level(rs_out, grp, u, scale4(sc, st[4 * u]), scale4(sc, st[4 * u + 1]), scale4(sc, st[4 * u + 2]),
                                        scale4(sc, st[4 * u + 3]));
                        } else {
                            store_level(rs_out, grp, u, st[4 * u], st[4 * u + 1], st[4 * u + 2], st[4 * u + 3]);
                        }
                    }
                }
            });
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);      // plane requests stay BEHIND the gather (registers, and the in-order ring)
            CG_STAMP(4 * step + 0);
            if (!last) {
                if (ADJ) request_in(src + (size_t)(jm - 1) * slab, grp);          // G of the next step
                lds_barrier();                        // every gather of this step is done
                CG_STAMP(4 * step + 1);
                // rotate: LDS <- T_k, registers <- T_{k-1} of the own rows (own slots: no other thread touches them)
                static_for<0, NG>([&](auto uc) {
                    constexpr int u = decltype(uc)::value;
                    ent_t prev[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) prev[i] = T[own_slot(u, i, false)];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        T[own_slot(u, i, true)] = st[4 * u + i];
                        if constexpr (u == NG - 1) st[4 * u + i] = sel4(has_slot(u), prev[i], st[4 * u + i]);
                        else st[4 * u + i] = prev[i];
                    }
                });
                CG_STAMP(4 * step + 2);
                if (ADJ) {
                    consume(gin, jm - 1);            // G of the next step, in front of the barrier: the wait overlaps the other waves' rotate
                    clear(gin);
                }
                lds_barrier();   
                CG_STAMP(4 * step + 3);
            }
        }
        // ---- the next group's input takes the image over -----------------------------------------------------------------
        CG_STAMP(40);
        // The CU's memory pipeline is in order: requested behind the 168 KB of stores of the last slab, the next group's input
        // was not even ISSUED until those had drained at the CU's share of HBM write bandwidth (17-22k cycles at this barrier,
        // phase stamps of round 4).  Forward: request first, then the last slab from the registers it is still in.
        const bool more = grp + (int)gridDim.x < ngrp;
        if (more) request_in(in0, grp + gridDim.x);
        else clear(gin);                             // (ends the live range of the old pieces: without it they stay allocated through every gather)
        __builtin_amdgcn_sched_barrier(0);
        if (K > 1) {                                 // forward: slab K-1; adjoint: dx -- from the registers it is still in
            const rsrc_t rs_last = slab_rsrc(ADJ ? dst : dst + (size_t)(K - 1) * slab, slab_bytes);
            const float ck = iso_coef(K - 1);
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                const float sc = (ADJ || has_slot(u)) ? 1.f : ck;
                store_level(rs_last, grp, u, scale4(sc, st[4 * u]), scale4(sc, st[4 * u + 1]), scale4(sc, st[4 * u + 2]),
                            scale4(sc, st[4 * u + 3]));
            }
        }
        if (ADJ) {
            __builtin_amdgcn_sched_barrier(0);
            if (more) request_into(gin2, src + (size_t)(K - 2) * slab, grp + gridDim.x);
            else clear(gin2);
        }
        lds_barrier();                                // every gather of the last step is done: the image may be overwritten
        CG_STAMP(41);
    }
}

// LDS entries of a shape: what NG levels of NT quads can hold (+ the zero and the trash slot), capped by the 160 KB of a CU
template <int PL, int NG, int NT>
struct ord_entries {
    static constexpr int cap = 160 * 1024 / (4 * PL);
    static constexpr int value = (4 * NT * NG + 16) < cap ? (4 * NT * NG + 16) : cap;
};

template <int PL, int NT, int NQ, int NG, bool ADJ>
int launch_ord(const chebgcn_graph* g, const Ell& ell, const float* src, float* dst, int nplanes, int K, int copy_t0,
               hipStream_t stream) {
    constexpr int ENT = ord_entries<PL, NG, NT>::value;
    auto kernel = cheb_ord_kernel<PL, ENT, NQ, NG, NT, ADJ>;
    // workgroups a CU holds (LDS and registers of the shape; small graphs: more than one), asked once per shape
    static int per_cu = 0;
    if (per_cu == 0) {
        int n = 0;
        CG_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, NT, 0));
        per_cu = n < 1 ? 1 : n;
    }
    const int ngrp = (nplanes + PL - 1) / PL;
    int grid = g->num_cus * per_cu;
    if (grid > ngrp) grid = ngrp;
    const size_t slab = (size_t)nplanes * g->Mp;
    // (four planes: the name tests and profiles of round 4 know)
    static const std::string name = std::string(PL == 4 ? "cheb_ord_kernel<" : "cheb_ord2_kernel<") + std::to_string(ENT) + "," +
                                    std::to_string(NQ) + "," + std::to_string(NG) + "," + std::to_string(NT) + "," +
                                    (ADJ ? "true" : "false") + ">";
    note_dispatch(name.c_str());
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(NT), 0, stream, view(ell), src, dst, g->M, g->Mp, nplanes, K, slab, ell.ord_SQ,
                       copy_t0 | (g_stagger << 20));
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

// the shapes of one plane count: NG in [NG0, NG1], NQ in {NG, NG + 1}
template <int PL, int NT, int NG, int NG1, bool ADJ>
int launch_ord_shape(const chebgcn_graph* g, const Ell& ell, const float* src, float* dst, int nplanes, int K, int copy_t0,
                     hipStream_t stream) {
    if (ell.ord_NG == NG) {
        if (ell.ord_NQ == NG) return launch_ord<PL, NT, NG, NG, ADJ>(g, ell, src, dst, nplanes, K, copy_t0, stream);
        if (ell.ord_NQ == NG + 1) return launch_ord<PL, NT, NG + 1, NG, ADJ>(g, ell, src, dst, nplanes, K, copy_t0, stream);
    }
    if constexpr (NG < NG1) return launch_ord_shape<PL, NT, NG + 1, NG1, ADJ>(g, ell, src, dst, nplanes, K, copy_t0, stream);
    return fail(CHEBGCN_EUNSUPPORTED, "recurrence: no ordered kernel shape {%d planes, %d, %d}", PL, ell.ord_NQ, ell.ord_NG);
}

}  // namespace

// shapes built: four planes, 512 threads, NG = 2..5; two planes, kOrd2NT threads, every NG from the first that does not fit
// four planes to 20476 active vertices; NQ = NG or NG + 1 each.  (Three planes -- 12-byte entries, up to 13650 active vertices --
// were built and measured in round 5: ds_read_b96 at a 12-byte stride runs the gather at a quarter of the two-plane rate,
// 0.10 of the HBM roofline at N = 10242; EXPERIMENTS.md)
#ifndef CG_ORD2_NT
#define CG_ORD2_NT 512
#endif
constexpr int kOrd4NT = 512, kOrd4NG0 = 2, kOrd4NG1 = 5;
// graphs of 1025 ... 2048 vertices (more than 256 quads per plane, at most 512 with rows): 256 threads, one or two quad levels
// with rows (recurrence_ord_small.hip)
constexpr int kOrdSNT = 256, kOrdSNG0 = 1, kOrdSNG1 = 2;
template <bool ADJ>
int launch_ordered_small(const chebgcn_graph* g, const Ell& ell, const float* src, float* dst, int nplanes, int K, int copy_t0,
                         hipStream_t stream);
constexpr int kOrd2NT = CG_ORD2_NT, kOrd2NG0 = (2560 + kOrd2NT - 1) / kOrd2NT, kOrd2NG1 = (5119 + kOrd2NT - 1) / kOrd2NT;
template <bool ADJ>
int launch_ordered2(const chebgcn_graph* g, const Ell& ell, const float* src, float* dst, int nplanes, int K, int copy_t0,
                    hipStream_t stream);       // recurrence_ord2.hip / recurrence_ord2a.hip

}  // namespace chebgcn
