// Chebyshev recurrence on chip, FOUR planes per workgroup (gfx950) -- graphs of roughly 2k..10k
// active vertices, where 16 bytes per ACTIVE vertex still fit the 160 KB of LDS.  This is the
// kernel the benchmark graph (N = 10000 -> M = 10466, 10000 active vertices) runs.
//
//   forward  (lib_new/models_gcn.py:598-610):  T_0 = x, T_1 = L T_0, T_k = 2 L T_{k-1} - T_{k-2}
//   adjoint  (TF autodiff of the above):       c_{K-1} = G_{K-1}, c_j = G_j + 2 L^T c_{j+1} - c_{j+2},
//                                              dx = G_0 + L^T c_1 - c_2
//
// Same algorithm, operator format (common.h) and phase structure as the two-plane kernel of
// recurrence.hip; why four planes: what paces the two-plane kernel is the operator stream --
// every workgroup re-reads all (value, slot id) entries from L2 once per step, 551 KB against
// 84 KB of plane payload, and the L2 -> CU path delivers 64 B per clock and CU.  One operator
// entry and one 16-byte ds_read_b128 serve four planes here: half the operator bytes, half the
// address arithmetic and half the LDS instructions per plane.  What it costs, and how it is paid:
//   * LDS: only ACTIVE vertices (non-empty row or column of the operator) have a slot.  An
//     isolated vertex ("fake" vertices of the coarsening) obeys T_k = -T_{k-2}: T_k = 0 for odd k
//     and (-1)^(k/2) x for even k.  Its even slabs are written once, when the input is staged
//     (4-byte stores); copy-outs of even slabs store the other vertices of such a piece
//     component-wise, so nothing in the gather loop ever loads (adjoint: dx = G_0 - G_2 + ...
//     is gathered from the gradient slabs when dx goes out).
//   * registers: 512 threads (two waves per SIMD) with up to 256 VGPRs: 20 rows x 4 planes of
//     T_{k-2} state per thread (80), the operator ring (40), and -- only while those are idle -- the
//     96 staging registers of the next input.  The rotate exchanges registers and LDS in chunks
//     of five rows (a slot is touched by its own thread only).
//   * no spare LDS for a second image: the copy-out of a group's last slab and the staging of the
//     next group's input are ONE pass -- a thread reads the old entries of its linear pieces and
//     overwrites them with the new input, no barrier in between; the input was requested behind
//     the barrier that ended the last gather, so its HBM latency hides under the last rotate.
#include <type_traits>

#include <string>

#include "common.h"

namespace chebgcn {

extern int g_ablate;
extern int g_stagger;

#ifndef CG_X
#define CG_X 0               // 64: in-kernel phase stamps (tools/vbuild.sh, tools/kbench.py --stamps); 0 in production
#endif
// In-kernel phase stamps (CG_X & 64, tools/xbuild.sh): lane 0 of every wave of workgroup 37 records the
// cycle counter at the phase boundaries of its SECOND plane group (tools/kbench.py --stamps).
__device__ long long g_dbg4[16 * 64];
#define CG_STAMP(id)                                                                          \
    do {                                                                                      \
        if ((CG_X & 64) && (id) < 64 && lane == 0 && blockIdx.x == 37 && grp == (int)(blockIdx.x + gridDim.x)) \
            g_dbg4[wave * 64 + (id)] = (long long)__builtin_readcyclecounter();                \
    } while (0)

namespace {


typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Plane accesses are buffer instructions: the descriptor of one slab (SGPRs), a uniform byte offset
// (plane + piece, an SGPR) and ONE per-thread offset register (tid * 16) -- no 64-bit per-lane
// address arithmetic, no address registers per (piece, plane).  Planes are read once and written
// once: streaming (nt) accesses keep them from pushing the operator out of the XCD's L2.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t slab_rsrc(const float* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 ldp(rsrc_t r, unsigned voff, unsigned soff) {
    const f32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 2);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void stp(rsrc_t r, unsigned voff, unsigned soff, float4 v) {
    const f32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(t, r, voff, soff, 2);
    // gfx950 (measured: lanes 12..15 of every 16 stored the NEXT value of the first data register): a
    // 16-byte buffer store still reads its data registers while the following instructions issue.
    // hipcc pads that hazard for the immediate-offset form only, not for an SGPR soffset.  The asm
    // reads the data (so nothing overwrites the registers before it) and supplies the wait states.
    asm volatile("s_nop 1" : : "v"(t) : "memory");
}
__device__ __forceinline__ void stp1(rsrc_t r, unsigned voff, unsigned soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 2);      // (the builtin's data operand is an integer)
}
// byte offset of the 16-byte LDS entry named by the low / high 16 bits of w: one VALU op
__device__ __forceinline__ unsigned ofs_lo(unsigned w) {
    unsigned r;
    asm("v_lshlrev_b32_sdwa %0, 4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(w));
    return r;
}
__device__ __forceinline__ unsigned ofs_hi(unsigned w) {
    unsigned r;
    asm("v_lshlrev_b32_sdwa %0, 4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(w));
    return r;
}
__device__ __forceinline__ float comp(const float4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
__device__ __forceinline__ void add_comp(float4& v, int i, float x) {
    if (i == 0) v.x += x; else if (i == 1) v.y += x; else if (i == 2) v.z += x; else v.w += x;
}
// Identity the optimiser cannot see through: keeps values DERIVED from the small per-thread
// tables (unpacked slot ids) from being hoisted out of the plane-group loop, where they would
// occupy registers for the whole kernel.
__device__ __forceinline__ unsigned opaque(unsigned x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ uint2 opaque(uint2 x) { asm volatile("" : "+v"(x.x), "+v"(x.y)); return x; }
__device__ __forceinline__ float4 opaque(float4 x) { asm volatile("" : "+v"(x.x), "+v"(x.y), "+v"(x.z), "+v"(x.w)); return x; }
// the same for a uniform value (SGPR): what is derived from it is recomputed at its use (one scalar add or compare)
// instead of being hoisted out of the step loop -- 20 slots x 5 offsets and 3 conditions would occupy over a hundred
// SGPRs, spilled to VGPR lanes and read back with v_readlane + wait states in front of every load
__device__ __forceinline__ unsigned opaque_s(unsigned x) { asm volatile("" : "+s"(x)); return x; }
__device__ __forceinline__ int opaque_s(int x) { asm volatile("" : "+s"(x)); return x; }
__device__ __forceinline__ float4 fma4(float s, const float4& t, const float4& a) {
    return make_float4(fmaf(s, t.x, a.x), fmaf(s, t.y, a.y), fmaf(s, t.z, a.z), fmaf(s, t.w, a.w));
}
// plane p of the four vertices whose entries are t[0..3]
__device__ __forceinline__ float4 plane_of_entries(const float4 (&t)[4], int p) {
    return make_float4(comp(t[0], p), comp(t[1], p), comp(t[2], p), comp(t[3], p));
}

// compile-time loop: f(std::integral_constant<int, J>) for J = FIRST .. LAST-1 (the slot number of a row group must be a
// constant inside the asm statement of its gather)
template <int FIRST, int LAST, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (FIRST < LAST) {
        f(std::integral_constant<int, FIRST>{});
        static_for<FIRST + 1, LAST>(f);
    }
}

// ENT = LDS entries (16 B each), NJ = row slices per thread (ceil(groups / 8)), NQ = linear
// 16-byte pieces per thread and plane (ceil(Mp/4 / 512)).
// ISOREG: every thread holds at most NISO isolated vertices among its linear pieces (graph.hip checks): their values ride
// in registers for the whole plane group instead of being re-read from HBM when slabs go out (see `xi` below).
constexpr int NISO = 4;
template <int ENT, int NJ, int NQ, int NT4, bool ADJ, bool ISOREG>
__global__ void __launch_bounds__(NT4)
cheb4_kernel(EllView e, const float* __restrict__ src, float* __restrict__ dst, int M, int Mp, int nplanes, int K,
             size_t slab, int flags) {
    __shared__ float4 T[ENT];                        // slot-indexed: the four planes of one vertex
    constexpr int NW4 = NT4 / 64;
    static_assert(NJ <= 64 && NQ <= NJ && (NJ % 2) == 0, "shape");
    static_assert(NJ * (NT4 / 64) <= 160, "the fixed-stride operator image is padded to 160 row groups (graph.hip)");
    const int copy_t0 = flags & 1;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Mq = Mp >> 2;
    const int ngrp = (nplanes + 3) >> 2;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned zslot = (unsigned)e.zero_slot;    // always 0; zslot + 1 = trash (written, never read as data)
    auto lds = [&](unsigned byteoff) -> float4 {
        return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(T) + byteoff);
    };

    // ---- per-thread tables, loaded once -----------------------------------------------------
    unsigned rowreg[NJ / 2];                         // LDS slot of the own row of slice j (two per register, 0xFFFF = none)
#pragma unroll
    for (int j2 = 0; j2 < NJ / 2; ++j2) {
        unsigned r = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int g = (2 * j2 + h) * NW4 + wave;
            const unsigned id = g < e.ngroups ? e.rowslot[g * 64 + lane] : 0xFFFFu;
            r |= id << (16 * h);
        }
        rowreg[j2] = r;
    }
    uint2 nsreg[NQ];                                 // LDS slots of the 4 vertices of every linear piece of this thread
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        const int q = tid + u * NT4;
        nsreg[u] = (q < Mq) ? reinterpret_cast<const uint2*>(e.nodeslot)[q] : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
    }
    unsigned isomask = 0;                            // bit 4u+i: vertex i of piece u is isolated (inside the graph, no slot)
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        const int q = tid + u * NT4;
        const unsigned w[2] = {nsreg[u].x, nsreg[u].y};
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (((w[i >> 1] >> (16 * (i & 1))) & 0xFFFFu) == 0xFFFFu && 4 * q + i < M) isomask |= 1u << (4 * u + i);
    }
    // ISOREG (forward) -- isolated vertices in registers.  T_k of an isolated vertex is 0 for odd k and (-1)^(k/2) x for even
    // k; x is in this thread's staging registers when the group's input is staged, so the (at most NISO) values are kept
    // (xi[n][plane], n = rank of the vertex among the thread's isolated ones) and every even slab goes out with sign * x
    // selected into place -- no re-read of x, no 4-byte patch stores, no second round trip when the group is turned over
    // (round 2 re-read x there: 10-13k of the turn-over's 35k cycles were one exposed memory latency, plus ~190 mostly-empty
    // store instructions per wave; the first gather of a group ran 7k cycles behind that traffic).  Measured at the north-star
    // shape: 0.555 against 0.580 ms in place, 0.54 against 0.59 ms with the copy of x; K = 25: 1.47 against 1.60 ms.
    // Tried and rejected: fetching x (adjoint: the even G_j) at the isolated vertices with 4-byte loads beside the plane
    // loads and storing sign * x behind the copy-out with 4-byte stores (no selects: 0.585 ms forward, 0.64 adjoint); the
    // register scheme for the adjoint (accumulating the isolated components of every even G_j: 0.597 against 0.578 ms).
    float xi[ISOREG ? NISO : 1][4];
#pragma unroll
    for (int n = 0; n < (ISOREG ? NISO : 1); ++n)
#pragma unroll
        for (int p = 0; p < 4; ++p) xi[n][p] = 0.f;
    int2 gtab = make_int2(0, 0);                     // lane j: {quad offset, length} of group j*8 + wave
    if (lane < NJ && lane * NW4 + wave < e.ngroups) gtab = e.ginfo[lane * NW4 + wave];
    // slots of this wave are sorted by length: the first nB have more than 10 entries, the first nA more than 8; bit j of
    // mC: slot j has more than 12 (uniform values: SGPRs)
    const int nA = __popcll(__ballot(lane < NJ && gtab.y > 8));
    const int nB = __popcll(__ballot(lane < NJ && gtab.y > 10));
    const unsigned mC = (unsigned)__ballot(lane < NJ && gtab.y > 12);
    const __amdgpu_buffer_rsrc_t uval_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)e.uval, 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t uids_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)e.uids, 0, 0x7FFFFFFF, 0x00020000);
    if (tid == 0) T[zslot] = zero4;                  // never written again
    {   // one-off stagger of the workgroups of an XCD (see recurrence.hip)
        const int gpw = (ngrp + (int)gridDim.x - 1) / (int)gridDim.x;
        const int sx = (flags >> 20) & 0xFF;
        const int m8 = gpw <= 4 ? 8 : gpw >= 16 ? 16 : 8 + (8 * (gpw - 4)) / 12;
        const int reps = ((blockIdx.x >> 3) & 31) * (sx ? sx - 1 : m8) / 8;
        for (int i = 0; i < reps; ++i) __builtin_amdgcn_s_sleep(10);
    }
    const size_t in_base = ADJ ? (size_t)(K - 1) * slab : 0;

    // plane offsets of a group (uniform); planes beyond nplanes alias the last one and are never stored
    const unsigned slab_bytes = (unsigned)(slab * sizeof(float));     // < 4 GB (checked by the dispatcher)
    const unsigned vb = (unsigned)tid * 16u;                            // this thread's byte offset inside a run of 512 pieces
    auto plane_off = [&](int g, int p, int u) __attribute__((always_inline)) -> unsigned {             // uniform: plane p of group g, piece run u
        const int i = g * 4 + p;
        return (unsigned)(i < nplanes ? i : nplanes - 1) * (unsigned)Mp * 4u + (unsigned)u * (NT4 * 16u);
    };
    const rsrc_t rs_t0 = slab_rsrc(dst, slab_bytes);                    // forward: slab 0 of the stack (T_0)
    // (a partial last group computes its missing planes as copies of the last one and stores them to the
    // same addresses: identical values, no branch per plane)
    auto ids_of_piece = [&](int u, unsigned (&id)[4]) {
        const uint2 nq = opaque(nsreg[u]);
        id[0] = nq.x & 0xFFFFu; id[1] = nq.x >> 16; id[2] = nq.y & 0xFFFFu; id[3] = nq.y >> 16;
    };

    // linear staging registers: the next group's input, G_j of the adjoint.  Live only while the
    // operator ring and (between groups) the row state are idle.
    float4 gx[NQ][4];
    auto load_planes = [&](const float* base, int g, int u0 = 0, int u1 = NQ) __attribute__((always_inline)) {
        const rsrc_t rs = slab_rsrc(base, slab_bytes);
#pragma unroll
        for (int u = u0; u < u1; ++u) {
            const int q = tid + u * NT4;
#pragma unroll
            for (int p = 0; p < 4; ++p) gx[u][p] = zero4;
            if (q < Mq) {
#pragma unroll
                for (int p = 0; p < 4; ++p) gx[u][p] = ldp(rs, vb, plane_off(g, p, u));
            }
        }
    };
    auto clear_planes = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NQ; ++u)
#pragma unroll
            for (int p = 0; p < 4; ++p) gx[u][p] = zero4;
    };

    // Forward, isolated vertices (no LDS slot): T_k = 0 for odd k, (-1)^(k/2) x for even k.  The copy-outs
    // below write what the zero slot gives (0) everywhere; the even slabs of a group are put right in ONE
    // pass when the group is turned over: the pieces that hold an isolated vertex re-read their 16 bytes
    // of x (other lanes address beyond the slab: zeros, no memory access; all requests of a batch
    // together: one latency) and store +-x over the zeros, 4 bytes at a time, in every even slab.
    // (Patching each slab as it goes out cost one exposed memory latency per even slab and batch.)
    auto load_patch = [&](float4 (*px)[4], int g, int u0, int u1) {
        const rsrc_t rs = slab_rsrc(src, slab_bytes);
#pragma unroll
        for (int u = u0; u < u1; ++u) {
            const unsigned vo = ((isomask >> (4 * u)) & 15u) ? vb : 0x80000000u;
#pragma unroll
            for (int p = 0; p < 4; ++p) px[u - u0][p] = ldp(rs, vo, plane_off(g, p, u));
        }
    };
    auto fix_isolated = [&](float4 (*px)[4], int g, int u0, int u1) {
        float sgn = -1.f;
        for (int k = 2; k < K; k += 2, sgn = -sgn) {
            const rsrc_t rs = slab_rsrc(dst + (size_t)k * slab, slab_bytes);
#pragma unroll
            for (int u = u0; u < u1; ++u) {
                const unsigned iso = (isomask >> (4 * u)) & 15u;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (iso & (1u << i)) {
#pragma unroll
                        for (int p = 0; p < 4; ++p) stp1(rs, vb + 4u * i, plane_off(g, p, u), sgn * comp(px[u - u0][p], i));
                    }
            }
        }
    };
    // One slab of the forward stack from the LDS image, pieces [u0, u1): t[i] = LDS entry of vertex 4q+i
    // (zero for a vertex without a slot: pads and isolated vertices)
    // The slot ids of the linear pieces (12 registers) are not kept across a gather in the ISOREG kernels: with the isolated
    // values in registers hipcc spilled them and reloaded each one with `s_waitcnt vmcnt(0)` in front of its piece -- every
    // copy-out drained its own stores six times.  They are fetched again (L2, all requests together) when a linear phase
    // starts; the opaque thread id keeps the loads from being hoisted back out of the step loop.
    auto reload_piece_slots = [&]() __attribute__((always_inline)) {
        const int t = (int)opaque((unsigned)tid);
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int q = t + u * NT4;
            nsreg[u] = (q < Mq) ? reinterpret_cast<const uint2*>(e.nodeslot)[q] : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
        }
    };
    // One slab of the forward stack (or dx) from the LDS image, pieces [u0, u1): t[i] = LDS entry of vertex 4q+i (zero for
    // a vertex without a slot: pads and isolated vertices).  ISOREG: iso_sign = what an isolated vertex holds in this slab
    // relative to x (0 in odd slabs -- what the zero slot gave anyway --, -1 / +1 in even ones).
    auto copy_out = [&](rsrc_t out, int g, int u0, int u1, float iso_sign = 0.f) __attribute__((always_inline)) {
        if (ISOREG && u0 == 0) reload_piece_slots();
#pragma unroll
        for (int u = u0; u < u1; ++u) {
            const int q = tid + u * NT4;
            if (q < Mq) {
                unsigned id[4];
                ids_of_piece(u, id);
                float4 t[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) t[i] = T[id[i] == 0xFFFFu ? zslot : id[i]];
                bool is[4] = {false, false, false, false};
                int rk[4] = {0, 0, 0, 0};
                const bool do_iso = ISOREG && iso_sign != 0.f;      // uniform
                if (do_iso) {
                    const unsigned im = opaque(isomask);            // (opaque: see iso_take)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        is[i] = (im >> (4 * u + i)) & 1u;
                        rk[i] = __popc(im & ((1u << (4 * u + i)) - 1u));       // rank among this thread's isolated vertices
                    }
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    float4 o = plane_of_entries(t, p);
                    if (do_iso) {
                        float v[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            v[i] = xi[0][p];
#pragma unroll
                            for (int n = 1; n < (ISOREG ? NISO : 1); ++n) v[i] = (rk[i] == n) ? xi[n][p] : v[i];
                            v[i] *= iso_sign;
                        }
                        o = make_float4(is[0] ? v[0] : o.x, is[1] ? v[1] : o.y, is[2] ? v[2] : o.z, is[3] ? v[3] : o.w);
                    }
                    stp(out, vb, plane_off(g, p, u), o);
                }
            }
            if (ISOREG) __builtin_amdgcn_sched_barrier(0);      // one piece at a time: interleaved, their selects spill
        }
    };
    // xi = the isolated components of the staging registers gx (straight-line selects: a branch per vertex made hipcc spill
    // hundreds of registers; the mask goes through an opaque identity: ranks and comparisons derived from it are loop
    // invariant, and hipcc would otherwise keep them in SGPR pairs across the whole kernel)
    auto iso_take = [&]() __attribute__((always_inline)) {
        const unsigned im = opaque(isomask);
#pragma unroll
        for (int u = 0; u < NQ; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool is = (im >> (4 * u + i)) & 1u;
                const int r = __popc(im & ((1u << (4 * u + i)) - 1u));
#pragma unroll
                for (int n = 0; n < NISO; ++n) {
                    const bool hit = is && r == n;
#pragma unroll
                    for (int p = 0; p < 4; ++p) xi[ISOREG ? n : 0][p] = hit ? comp(gx[u][p], i) : xi[ISOREG ? n : 0][p];
                }
            }
    };
    auto slab_iso_sign = [&](int k) -> float { return (k & 1) ? 0.f : ((k & 2) ? -1.f : 1.f); };

    // The pass between two groups: the final image of group `pg` (forward: slab K-1; adjoint: dx)
    // goes out and the input of group `ng` (already in gx) takes its place.  Every linear piece is read
    // and then overwritten by the thread that owns it in every linear phase: no barrier in between.
    auto turn_over = [&](bool have_prev, int pg, bool have_next, int ng) __attribute__((always_inline)) {
        const rsrc_t rs_out = slab_rsrc(ADJ ? dst : dst + (size_t)(K - 1) * slab, slab_bytes);   // dx / slab K-1
        if (have_prev && !ADJ) {
            // Requests, in the order their data is needed (the vector memory pipeline returns in order): x
            // of the isolated vertices of the first half of the pieces, then the next group's input.  The
            // patch registers overlay the row state, which died with the last rotate.
            constexpr int NB = (NQ + 1) / 2;
            float4 px[NB][4];
            const int grp = pg;                          // (for the stamps)
            if (!ISOREG && K > 2) load_patch(px, pg, 0, NB);
            if (have_next) load_planes(src, ng); else clear_planes();
            CG_STAMP(42);
            copy_out(rs_out, pg, 0, NQ, slab_iso_sign(K - 1));
            CG_STAMP(43);
            if (!ISOREG && K > 2) {
                fix_isolated(px, pg, 0, NB);
                CG_STAMP(44);
                load_patch(px, pg, NB, NQ);
                fix_isolated(px, pg, NB, NQ);
                CG_STAMP(45);
            }
        }
        if (have_prev && ADJ) {
            // dx = the final image; an isolated vertex (no slot) has dx = G_0 - G_2 + G_4 - ...: the pieces
            // that hold one re-read their 16 bytes of those gradient slabs (other lanes address out of
            // range: no memory access).  Two pieces at a time, up to three slabs per round trip, all
            // requests of a round issued together into the idle staging registers.
            constexpr int NB = 2, NS = NQ / NB >= 3 ? 3 : NQ / NB;
            static_assert(NB * NS <= NQ, "the staging registers hold a round of patch requests");
#pragma unroll
            for (int u0 = 0; u0 < NQ; u0 += NB) {
                float4 acc[NB][4];
#pragma unroll
                for (int uu = 0; uu < NB; ++uu)
#pragma unroll
                    for (int p = 0; p < 4; ++p) acc[uu][p] = zero4;
                float sgn = 1.f;
                for (int m0 = 0; m0 < K; m0 += 2 * NS) {
#pragma unroll
                    for (int sidx = 0; sidx < NS; ++sidx) {
                        const int m = m0 + 2 * sidx;
                        const rsrc_t rs_g = slab_rsrc(src + (size_t)(m < K ? m : 0) * slab, slab_bytes);
#pragma unroll
                        for (int uu = 0; uu < NB; ++uu) {
                            const int u = u0 + uu;
                            const bool want = u < NQ && m < K && ((isomask >> (4 * (u < NQ ? u : 0))) & 15u) != 0;
                            const unsigned vo = want ? vb : 0x80000000u;
#pragma unroll
                            for (int p = 0; p < 4; ++p) gx[sidx * NB + uu][p] = ldp(rs_g, vo, plane_off(pg, p, u < NQ ? u : 0));
                        }
                    }
#pragma unroll
                    for (int sidx = 0; sidx < NS; ++sidx) {
                        const float sg = (sidx & 1) ? -sgn : sgn;          // out-of-range requests returned zeros
#pragma unroll
                        for (int uu = 0; uu < NB; ++uu)
#pragma unroll
                            for (int p = 0; p < 4; ++p) {
                                acc[uu][p].x = fmaf(sg, gx[sidx * NB + uu][p].x, acc[uu][p].x);
                                acc[uu][p].y = fmaf(sg, gx[sidx * NB + uu][p].y, acc[uu][p].y);
                                acc[uu][p].z = fmaf(sg, gx[sidx * NB + uu][p].z, acc[uu][p].z);
                                acc[uu][p].w = fmaf(sg, gx[sidx * NB + uu][p].w, acc[uu][p].w);
                            }
                    }
                    if (NS & 1) sgn = -sgn;
                }
#pragma unroll
                for (int uu = 0; uu < NB; ++uu) {
                    const int u = u0 + uu;
                    const int q = tid + u * NT4;
                    if (u < NQ && q < Mq) {
                        unsigned id[4];
                        ids_of_piece(u, id);
                        float4 t[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) t[i] = T[id[i] == 0xFFFFu ? zslot : id[i]];
                        const unsigned iso = (isomask >> (4 * u)) & 15u;
#pragma unroll
                        for (int p = 0; p < 4; ++p) {
                            float4 o = plane_of_entries(t, p);
                            if (iso & 1u) o.x = acc[uu][p].x;
                            if (iso & 2u) o.y = acc[uu][p].y;
                            if (iso & 4u) o.z = acc[uu][p].z;
                            if (iso & 8u) o.w = acc[uu][p].w;
                            stp(rs_out, vb, plane_off(pg, p, u), o);
                        }
                    }
                }
            }
        }
        if (have_next) {
            // adjoint (and the very first group): requested only now -- the staging registers were the
            // patch buffer of the dx copy-out above; forward: requested behind the last rotate
            if (ADJ || !have_prev) load_planes(src + in_base, ng);
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                const int q = tid + u * NT4;
                if (q < Mq) {
                    unsigned id[4];
                    ids_of_piece(u, id);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        T[id[i] == 0xFFFFu ? zslot + 1u : id[i]] =
                            make_float4(comp(gx[u][0], i), comp(gx[u][1], i), comp(gx[u][2], i), comp(gx[u][3], i));
                    if (!ADJ && copy_t0) {           // T_0 = x goes straight to slab 0
#pragma unroll
                        for (int p = 0; p < 4; ++p) stp(rs_t0, vb, plane_off(ng, p, u), gx[u][p]);
                    }
                }
            }
            if (ISOREG) iso_take();                  // x of this thread's isolated vertices, for the even slabs of the group
            // isolated vertices: forward x itself; adjoint the first term G_{K-1} of dx = G_0 - G_2 + ... (0 if K-1 is odd)

        }
    };

    int grp = blockIdx.x;
    __syncthreads();
    turn_over(false, 0, grp < ngrp, grp);
    for (; grp < ngrp; grp += gridDim.x) {
        CG_STAMP(0);
        CG_STAMP(1);
        __syncthreads();                             // the image of this group is complete
        CG_STAMP(2);

        float4 st[NJ];                               // T_{k-2} of the own rows, replaced by T_k in place
#pragma unroll
        for (int j = 0; j < NJ; ++j) st[j] = zero4;

        // Closes step `sdone`: once every wave has finished its gather, LDS <- T_k and the registers
        // <- T_{k-1} of the own rows (adjoint: then c_j += G_j).  Runs at the top of the next step and,
        // for the last step (where the next group's input is requested), after the loop -- kept out of
        // the loop so that the staging registers are not live across a gather.
        auto finish_step = [&](int sdone, bool last) __attribute__((always_inline)) {
            CG_STAMP(4 * sdone + 0);
            __syncthreads();                         // every gather (and copy-out read) of this step is done
            CG_STAMP(4 * sdone + 1);
            // HBM requests only now, behind the barrier: queued while other waves still gather they
            // would hold up those waves' operator loads (the vector memory pipeline returns in order)
            // Registers: the row state (4*NJ) and all 16*NQ staging registers do not fit together, so
            // the first half of G_j is requested before the rotate and the second half behind it; after
            // the last step the old T_{k-1} of the own rows is not needed any more (write-only rotate),
            // and the next group's input is requested once the state is dead.
            constexpr int NH = NQ / 3;                     // a third before the rotate (its chunk registers are live), the rest behind
            const float* gj = src + (size_t)(K - 1 - sdone) * slab;                // G_j of the finished step (adjoint)
            if (ADJ) load_planes(gj, grp, 0, NH);
            // rotate: LDS <- T_k, registers <- T_{k-1} of the own rows; a slot belongs to one thread
            constexpr int RC = 5;
#pragma unroll
            for (int j0 = 0; j0 < NJ; j0 += RC) {
                float4 prev[RC];
                if (!last) {
#pragma unroll
                    for (int j = j0; j < j0 + RC && j < NJ; ++j) {
                        const unsigned r = (opaque(rowreg[j >> 1]) >> (16 * (j & 1))) & 0xFFFFu;
                        prev[j - j0] = T[r == 0xFFFFu ? zslot : r];
                    }
                }
#pragma unroll
                for (int j = j0; j < j0 + RC && j < NJ; ++j) {
                    const unsigned r = (opaque(rowreg[j >> 1]) >> (16 * (j & 1))) & 0xFFFFu;
                    T[r == 0xFFFFu ? zslot + 1u : r] = st[j];
                    if (!last) st[j] = prev[j - j0];
                }
            }
            if (ADJ) load_planes(gj, grp, NH, NQ);

            CG_STAMP(4 * sdone + 2);
            __syncthreads();
            CG_STAMP(4 * sdone + 3);
            if (ADJ) {
                // c_j += G_j: every linear piece by its owner
#pragma unroll
                for (int u = 0; u < NQ; ++u) {
                    const int q = tid + u * NT4;
                    if (q < Mq) {
                        unsigned id[4];
                        ids_of_piece(u, id);
                        float4 t[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) t[i] = T[id[i] == 0xFFFFu ? zslot : id[i]];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            t[i].x += comp(gx[u][0], i);
                            t[i].y += comp(gx[u][1], i);
                            t[i].z += comp(gx[u][2], i);
                            t[i].w += comp(gx[u][3], i);
                            T[id[i] == 0xFFFFu ? zslot + 1u : id[i]] = t[i];
                        }
                    }
                }
                __syncthreads();
            }
        };
        for (int step = 1; step < K; ++step) {
            if (step > 1) finish_step(step - 1, false);
            const float f = ADJ ? (step == K - 1 ? 1.f : 2.f) : (step == 1 ? 1.f : 2.f);
            // forward: slab step-1 is written out while this step gathers
            const bool do_out = !ADJ && step > 1;
            const rsrc_t out_slab = slab_rsrc(dst + (size_t)(step - 1) * slab, slab_bytes);

            // ---- gather: st <- f * (A T_{k-1})[own rows] - st -----------------------------------------------
            // Operator records of slot j (group j*NW4 + wave) sit at compile-time offsets behind one per-wave base; they
            // travel through a ring two slots deep: values of quads 0 / 1, the record of entries 8..11, the eight ids of
            // quads 0 and 1.  The two optional requests are skipped by scalar-only tests (j is a constant, nA / nB SGPRs).
            float4 uq[3][2];
            float2 ub[2];
            uint4 uo[2];
            const unsigned vsoff = (unsigned)wave * 4096u, isoff = (unsigned)wave * 1024u;
            auto urequest = [&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                constexpr unsigned vo = (unsigned)j * NW4 * 4096u, io = (unsigned)j * NW4 * 1024u;
                const unsigned vs = opaque_s(vsoff), is = opaque_s(isoff);
                const f32x4 a = __builtin_amdgcn_raw_buffer_load_b128(uval_rsrc, lane * 16, vs + vo, 0);
                const f32x4 b = __builtin_amdgcn_raw_buffer_load_b128(uval_rsrc, lane * 16, vs + (vo + 1024u), 0);
                const u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(uids_rsrc, lane * 16, is + io, 0);
                uq[0][j & 1] = make_float4(a.x, a.y, a.z, a.w);
                uq[1][j & 1] = make_float4(b.x, b.y, b.z, b.w);
                uo[j & 1] = make_uint4(c.x, c.y, c.z, c.w);
                if (j < opaque_s(nA)) {
                    const f32x4 d = __builtin_amdgcn_raw_buffer_load_b128(uval_rsrc, lane * 16, vs + (vo + 2048u), 0);
                    uq[2][j & 1] = make_float4(d.x, d.y, d.z, d.w);
                }
                if (__builtin_expect(j < opaque_s(nB), 0)) {
                    const f32x2 h = __builtin_amdgcn_raw_buffer_load_b64(uval_rsrc, lane * 16, vs + (vo + 3072u), 0);
                    ub[j & 1] = make_float2(h.x, h.y);
                }
            };
            auto lds_quad = [&](const uint2 c, const float4 v, float4& acc) __attribute__((always_inline)) {
                const unsigned a0 = ofs_lo(c.x), a1 = ofs_hi(c.x), a2 = ofs_lo(c.y), a3 = ofs_hi(c.y);
                const float4 t0 = lds(a0), t1 = lds(a1), t2 = lds(a2), t3 = lds(a3);
                acc = fma4(v.x, t0, acc);
                acc = fma4(v.y, t1, acc);
                acc = fma4(v.z, t2, acc);
                acc = fma4(v.w, t3, acc);
            };
            urequest(std::integral_constant<int, 0>{});
            if constexpr (NJ > 1) urequest(std::integral_constant<int, 1>{});
            static_for<0, NJ>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                // waves that are ahead yield to the ones behind (see recurrence.hip)
                if constexpr (j == 0 || (4 * j) / NJ != (4 * (j - 1)) / NJ) {
                    constexpr int pr = 3 - (4 * j) / NJ;
                    if constexpr (pr == 3) __builtin_amdgcn_s_setprio(3);
                    else if constexpr (pr == 2) __builtin_amdgcn_s_setprio(2);
                    else if constexpr (pr == 1) __builtin_amdgcn_s_setprio(1);
                    else __builtin_amdgcn_s_setprio(0);
                }
                const uint4 o01 = uo[j & 1];
                const float4 v0 = uq[0][j & 1], v1 = uq[1][j & 1], v2 = uq[2][j & 1];
                const float2 v3 = ub[j & 1];
                // the same records gathered by compiler-scheduled C++ (no fixed register block): length classes chosen by
                // scalar-only tests of the slot number against nA / nB
                float4 acc = zero4;
                lds_quad(make_uint2(o01.x, o01.y), v0, acc);
                lds_quad(make_uint2(o01.z, o01.w), v1, acc);
                if (j < opaque_s(nA)) {
                    const unsigned c4 = __float_as_uint(v2.z);
                    const float4 t0 = lds(ofs_lo(c4)), t1 = lds(ofs_hi(c4));
                    acc = fma4(v2.x, t0, acc);
                    acc = fma4(v2.y, t1, acc);
                    if (j < opaque_s(nB)) {
                        const unsigned c5 = __float_as_uint(v2.w);
                        const float4 t2 = lds(ofs_lo(c5)), t3 = lds(ofs_hi(c5));
                        acc = fma4(v3.x, t2, acc);
                        acc = fma4(v3.y, t3, acc);
                    }
                }
                if constexpr (j + 2 < NJ) urequest(std::integral_constant<int, j + 2>{});      // refill the ring slots just consumed
                st[j] = make_float4(fmaf(f, acc.x, -st[j].x), fmaf(f, acc.y, -st[j].y), fmaf(f, acc.z, -st[j].z),
                                    fmaf(f, acc.w, -st[j].w));
            });
            // Rows beyond 12 entries (rare; rows are sorted, so they sit in the first slots of a wave): the sum over their
            // further quads, from the variable-stride image the round-2 way, is added afterwards -- st = f * (sum) - st_old
            // is linear in the sum.  One test per step when there are none.
            if (opaque_s(mC) != 0u) {
                static_for<0, NJ>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    if ((opaque_s(mC) >> j) & 1u) {
                        const int qoff = __builtin_amdgcn_readlane(gtab.x, j), len = __builtin_amdgcn_readlane(gtab.y, j);
                        float4 acc = zero4;
                        for (int q = 3; 4 * q < len; ++q) {
                            const uint4 o = e.colo[(size_t)((qoff >> 1) + (q >> 1)) * 64 + lane];
                            const float4 v = e.valq[(size_t)(qoff + q) * 64 + lane];
                            lds_quad((q & 1) ? make_uint2(o.z, o.w) : make_uint2(o.x, o.y), v, acc);
                        }
                        st[j] = make_float4(fmaf(f, acc.x, st[j].x), fmaf(f, acc.y, st[j].y), fmaf(f, acc.z, st[j].z), fmaf(f, acc.w, st[j].w));
                    }
                });
            }
            CG_STAMP(24 + step);
            // forward: slab step-1 (the image the gather just read) goes out now, before the barrier --
            // waves that finish their rows early stream while the others still gather; kept out of the
            // gather loop: its 40-odd temporaries do not fit next to the row state and the operator ring
            if (do_out) copy_out(out_slab, grp, 0, NQ, slab_iso_sign(step - 1));
        }
        finish_step(K - 1, true);
        // ---- the final image goes out, the next group's input comes in -------------------------
        CG_STAMP(40);
        turn_over(true, grp, grp + (int)gridDim.x < ngrp, grp + (int)gridDim.x);
        CG_STAMP(41);
    }
}

template <int ENT, int NJ, int NQ, int NT4, bool ADJ>
int launch4(const chebgcn_graph* g, const Ell& ell, const float* src, float* dst, int nplanes, int K, int copy_t0,
            hipStream_t stream) {
    static_assert(NT4 == 512, "Ell::iso_max512 counts isolated vertices per thread of a 512-thread workgroup");
    const int ngrp = (nplanes + 3) / 4;
    int grid = g->num_cus * ((160 * 1024) / (ENT * 16) < 1 ? 1 : (160 * 1024) / (ENT * 16));
    if (grid > ngrp) grid = ngrp;
    const size_t slab = (size_t)nplanes * g->Mp;
    bool launched = false;
    static const std::string stem = "cheb4_kernel<" + std::to_string(ENT) + "," + std::to_string(NJ) + "," + std::to_string(NQ) + "," +
                                    std::to_string(NT4) + "," + (ADJ ? "true" : "false");
    static const std::string name_iso = stem + ",true>", name_plain = stem + ",false>";
    if constexpr (!ADJ) {                            // (the adjoint keeps the round-2 scheme: see `xi` in the kernel)
        if (ell.iso_max512 <= NISO) {
            note_dispatch(name_iso.c_str());
            hipLaunchKernelGGL((cheb4_kernel<ENT, NJ, NQ, NT4, ADJ, true>), dim3(grid), dim3(NT4), 0, stream, view(ell), src, dst, g->M,
                               g->Mp, nplanes, K, slab, copy_t0 | (g_stagger << 20));
            launched = true;
        }
    }
    if (!launched) note_dispatch(name_plain.c_str());
    if (!launched)
        hipLaunchKernelGGL((cheb4_kernel<ENT, NJ, NQ, NT4, ADJ, false>), dim3(grid), dim3(NT4), 0, stream, view(ell), src, dst, g->M,
                           g->Mp, nplanes, K, slab, copy_t0 | (g_stagger << 20));
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

// shape 0 = none; shapes: {entries, threads, rows per thread, linear pieces per thread}.  512 threads (two
// waves per SIMD, up to 256 registers each).  A 768-thread shape (three waves per SIMD, 168 registers,
// 14 rows per thread) was measured on the same box: forward 0.64-0.66 ms against 0.61 ms, adjoint
// 0.69 ms against 0.61 ms at the north-star shape -- its row state spills inside the gather.
int shape4(int lds_entries, int rows, int Mq) {
    if (generic4_fits(rows, Mq)) return 0;           // small graphs: the generic kernel of recurrence.hip
    if (lds_entries <= 5120 && rows <= 10 * 512 && (Mq + 511) / 512 <= 4) return (Mq + 511) / 512 <= 3 ? 1 : 2;
    if (lds_entries <= 10240 && rows <= 20 * 512 && (Mq + 511) / 512 <= 7) return (Mq + 511) / 512 <= 6 ? 3 : 4;
    return 0;
}

}  // namespace

bool onchip4_fits(int lds_entries, int rows, int Mq) { return shape4(lds_entries, rows, Mq) != 0; }

template <bool ADJ>
int dispatch_onchip4(const chebgcn_graph* g, const Ell& ell, const float* src, float* dst, int nplanes, int K, int copy_t0,
                     hipStream_t stream) {
    // one slab is addressed through one buffer descriptor (32-bit offsets): larger batches go in chunks of planes
    const int max_planes = (int)((0xFFFF0000ull / ((size_t)g->Mp * sizeof(float))) & ~3ull);
    if (nplanes > max_planes) return fail(CHEBGCN_EUNSUPPORTED, "recurrence: %d planes of %d vertices exceed 4 GB per slab", nplanes, g->Mp);
    switch (shape4(ell.lds_entries, ell.ngroups * 64, g->Mp / 4)) {
        case 1: return launch4<5120, 10, 3, 512, ADJ>(g, ell, src, dst, nplanes, K, copy_t0, stream);
        case 2: return launch4<5120, 10, 4, 512, ADJ>(g, ell, src, dst, nplanes, K, copy_t0, stream);
        case 3: return launch4<10240, 20, 6, 512, ADJ>(g, ell, src, dst, nplanes, K, copy_t0, stream);
        case 4: return launch4<10240, 20, 7, 512, ADJ>(g, ell, src, dst, nplanes, K, copy_t0, stream);
        default: break;
    }
    return fail(CHEBGCN_EUNSUPPORTED, "recurrence: no four-plane kernel shape for %d rows", ell.ngroups * 64);
}

template int dispatch_onchip4<false>(const chebgcn_graph*, const Ell&, const float*, float*, int, int, int, hipStream_t);
template int dispatch_onchip4<true>(const chebgcn_graph*, const Ell&, const float*, float*, int, int, int, hipStream_t);

}  // namespace chebgcn

#ifdef CG_EXPERIMENT
extern "C" int chebgcn_debug_stamps4(long long* out) {      // CG_X & 64 builds only (tools/kbench.py --stamps)
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(chebgcn::g_dbg4), sizeof(long long) * 16 * 64) == hipSuccess ? 0 : -1;
}
#endif
