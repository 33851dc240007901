// Chebyshev three-term recurrence over a fixed sparse operator, forward and adjoint.
//
//   forward  (lib_new/models_gcn.py:598-610):  T_0 = x, T_1 = L T_0, T_k = 2 L T_{k-1} - T_{k-2}
//   adjoint  (TF autodiff of the above):       c_{K-1} = G_{K-1}, c_j = G_j + 2 L^T c_{j+1} - c_{j+2},
//                                              dx = G_0 + L^T c_1 - c_2
//
// Data layout: planes.  A plane is one (window, feature) column of the reference's
// x0[M, Fin*N] matrix: Mp contiguous floats, vertex-fastest.  Planes are independent under
// the recurrence, so a workgroup owns P planes (P = 4 or 2, one P-float entry per vertex) and
// runs all K-1 steps for them on chip:
//   * T_{k-1} of the P planes lives in LDS (the gather source),
//   * T_{k-2}, overwritten in place by the freshly computed T_k, of the rows a thread owns
//     lives in VGPRs,
//   * HBM sees each plane exactly once per slab: x is read once, every T_k written once
//     (compulsory traffic 4*M*Fin*K bytes per window instead of 4*M*Fin*(3K-4) for a
//     kernel-per-step SpMM).
// What bounds it: every workgroup re-streams the operator (6 B per entry: 16-bit LDS slot +
// fp32 value) from L2 once per step -- measured, that stream (not HBM, not LDS) set the pace
// with 2 planes per workgroup.  More planes per workgroup amortise it, hence P = 4 whenever
// 16 B per *active* vertex fit the 160 KiB LDS (graph.hip); vertices whose operator row and
// column are both empty ("fake" vertices added by the coarsening) have no LDS slot: they obey
// T_k = -T_{k-2} and are patched in while streaming.
// The operator comes as a length-sorted sliced ELL (graph.hip): the 64 rows handled by one
// wave have (nearly) equal length, so the entry loop has a wave-uniform trip count and no
// divergence; column/value loads are coalesced 128 B / 256 B per wave, all of a group's
// entries are requested before the first LDS gather (one L2 round trip per group), and the
// gathers are issued in batches of four.
// Because rows are handed to lanes in length order, results are scattered back into the LDS
// image and streamed out linearly, which keeps every HBM access fully coalesced.
// Software pipeline (one workgroup per CU, so nothing else hides latency):
//   * the linear copy-out of T_{k-1} (LDS -> HBM) is interleaved with the gather of step k,
//     each piece issued right after a group's operator loads so that later waits on those
//     loads never cover the stores;
//   * workgroups of one XCD start staggered so that they are in different phases.
#include <string>

#include "common.h"

namespace chebgcn {

int g_stagger = 0;          // chebgcn_tune(4, x): 0 = automatic
int g_wide = 0;            // chebgcn_tune(3, 1): prefer the 1024-thread shape (experiment)

#ifndef CG_X
#define CG_X 0               // 64: in-kernel phase stamps (tools/xbuild.sh, tools/kbench.py --stamps); 0 in production
#endif
constexpr int QMAX = 3;      // quads (4 operator entries each) requested per group, always, one group ahead
static_assert(QMAX <= kQuadPad && QMAX == kQuadMin, "the operator arrays are padded for the unconditional requests");

// planes are read once and written once: streaming (non-temporal) accesses keep them from
// pushing the operator, which every step re-reads, out of the XCD's L2
typedef float f32x4n __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldg4(const float* p) {
    const f32x4n v = __builtin_nontemporal_load(reinterpret_cast<const f32x4n*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void stg4(float* p, float4 v) {
    f32x4n t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4n*>(p));
}
__device__ __forceinline__ unsigned slot_of(uint2 c, int i) {        // i-th 16-bit id of a packed quad
    const unsigned w = (i & 2) ? c.y : c.x;
    return (i & 1) ? (w >> 16) : (w & 0xFFFFu);
}
__device__ __forceinline__ float comp(float4 v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
__device__ __forceinline__ void set_comp(float4& v, int i, float x) {
    if (i == 0) v.x = x; else if (i == 1) v.y = x; else if (i == 2) v.z = x; else v.w = x;
}

// Identity the optimiser cannot see through: keeps values DERIVED from the small per-thread
// tables (unpacked slot ids, flags) from being hoisted out of the plane-group loop, where they
// would occupy registers for the whole kernel and spill.
__device__ __forceinline__ unsigned opaque(unsigned x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ uint2 opaque(uint2 x) { asm volatile("" : "+v"(x.x), "+v"(x.y)); return x; }
__device__ __forceinline__ float4 opaque(float4 x) { asm volatile("" : "+v"(x.x), "+v"(x.y), "+v"(x.z), "+v"(x.w)); return x; }

// LDS entries a workgroup shape can hold: one per ranked row (+ zero slot, rounded), capped by
// the 160 KB of a CU.  For P = 4 the image holds only active vertices, all of them ranked.
__host__ __device__ constexpr int lds_capacity(int nj, int nthr, int planes) {
    return (nj * nthr + 4) * planes * 4 <= 160 * 1024 ? nj * nthr + 4 : 160 * 1024 / (planes * 4);
}

// In-kernel phase stamps (CG_X & 64, tools/xbuild.sh): lane 0 of every wave of workgroup
// g_dbg_block records the cycle counter at phase boundaries of its SECOND plane group (its only one on small launches).
__device__ long long g_dbg[16 * 64];
#define CG_STAMP(id)                                                                          \
    do {                                                                                      \
        if ((CG_X & 64) && (id) < 64 && lane == 0 && blockIdx.x == 37 && grp == blockIdx.x + (ngrp > (int)gridDim.x ? (int)gridDim.x : 0)) \
            g_dbg[wave * 64 + (id)] = (long long)__builtin_readcyclecounter();                \
    } while (0)

// levels whose id records (2 per group) stay in LDS: big shapes only (one workgroup per CU anyway)
__host__ __device__ constexpr int lds_id_levels(int nj, int nthr, int planes) {
    if (nthr < 512) return 0;
    const int spare = 160 * 1024 - lds_capacity(nj, nthr, planes) * planes * 4;
    const int lv = spare / ((nthr >> 6) * 2 * 1024);
    return lv > nj ? nj : lv < 0 ? 0 : lv;
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int P> struct Ent { float x[P]; };       // one LDS entry: P planes of one vertex

template <int P>
__device__ __forceinline__ Ent<P> lds_get(const float* T, unsigned slot) {
    Ent<P> r;
    if constexpr (P == 4) {
        const float4 t = *reinterpret_cast<const float4*>(T + slot * 4);
        r.x[0] = t.x; r.x[1] = t.y; r.x[2] = t.z; r.x[3] = t.w;
    } else {
        const float2 t = *reinterpret_cast<const float2*>(T + slot * 2);
        r.x[0] = t.x; r.x[1] = t.y;
    }
    return r;
}
template <int P>
__device__ __forceinline__ Ent<P> lds_at(const float* T, unsigned byteoff) {
    Ent<P> r;
    const char* p = reinterpret_cast<const char*>(T) + byteoff;
    if constexpr (P == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        r.x[0] = t.x; r.x[1] = t.y; r.x[2] = t.z; r.x[3] = t.w;
    } else {
        const float2 t = *reinterpret_cast<const float2*>(p);
        r.x[0] = t.x; r.x[1] = t.y;
    }
    return r;
}
// byte offset of the LDS entry named by the low (H = 0) / high (H = 1) 16 bits of w: one VALU op
template <int P, int H>
__device__ __forceinline__ unsigned entry_ofs(unsigned w) {
    unsigned r;
    if constexpr (P == 4 && H == 0)
        asm("v_lshlrev_b32_sdwa %0, 4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(w));
    else if constexpr (P == 4)
        asm("v_lshlrev_b32_sdwa %0, 4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(w));
    else if constexpr (H == 0)
        asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(w));
    else
        asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(w));
    return r;
}
template <int P>
__device__ __forceinline__ void lds_put(float* T, unsigned slot, const Ent<P>& v) {
    if constexpr (P == 4) *reinterpret_cast<float4*>(T + slot * 4) = make_float4(v.x[0], v.x[1], v.x[2], v.x[3]);
    else *reinterpret_cast<float2*>(T + slot * 2) = make_float2(v.x[0], v.x[1]);
}

// One workgroup = all ranked rows x P planes.  NJ = row slices per thread
// (ceil(ngroups*64 / NTHR)), NQ = 16-byte linear pieces per thread and plane (ceil(Mp/4 / NTHR) <= NJ).
template <int P, int NJ, int NQ, int NTHR, bool ADJ>
__global__ void __launch_bounds__(NTHR)
cheb_onchip_kernel(EllView e, const float* __restrict__ src, float* __restrict__ dst,
                   int M, int Mp, int nplanes, int K, size_t slab, int flags) {
    // static image: its address folds into the LDS instructions (no base add per access)
    __shared__ __attribute__((aligned(16))) float T[lds_capacity(NJ, NTHR, P) * P];    // [entries][P], slot-indexed
    // What the image leaves free of the 160 KB holds the id records of the first JL levels of every
    // wave (1 KB each): an LDS read instead of a vector-memory instruction per step and record.
    constexpr int JL = lds_id_levels(NJ, NTHR, P);
    __shared__ uint4 idrec[JL > 0 ? JL * (NTHR >> 6) * 2 * 64 : 1];
    constexpr int nthr = NTHR;
    constexpr int nwaves = NTHR >> 6;
    const int copy_t0 = flags & 1;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Mq = Mp >> 2;                                       // float4 pieces per plane
    const int ngrp = (nplanes + P - 1) / P;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    // LDS slot of the own row of every slice, two 16-bit ids per register (0xFFFF = none)
    constexpr int NJ2 = (NJ + 1) / 2;
    unsigned rowreg[NJ2];
#pragma unroll
    for (int j2 = 0; j2 < NJ2; ++j2) {
        unsigned r = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int g = (2 * j2 + h) * nwaves + wave;
            const unsigned id = (2 * j2 + h < NJ && g < e.ngroups) ? e.rowslot[g * 64 + lane] : 0xFFFFu;
            r |= id << (16 * h);
        }
        rowreg[j2] = r;
    }
    // LDS slots of the vertices of the linear pieces this thread moves (4 ids per piece)
    uint2 nsreg[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        const int q = tid + u * nthr;
        nsreg[u] = (q < Mq) ? reinterpret_cast<const uint2*>(e.nodeslot)[q] : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
    }
    // {quad offset, even length} of this wave's groups: lane j holds group j*nwaves + wave.  The
    // gather reads them with v_readlane -- a load inside the loop would have to be waited for
    // with vmcnt(0), draining the operator requests that are in flight.
    int2 gtab = make_int2(0, 0);
    if (lane < NJ && lane * nwaves + wave < e.ngroups) gtab = e.ginfo[lane * nwaves + wave];
    const __amdgpu_buffer_rsrc_t colo_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)e.colo, 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t valq_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)e.valp, 0, 0x7FFFFFFF, 0x00020000);
    if (tid < P) T[e.zero_slot * P + tid] = 0.f;      // never written again
    if constexpr (JL > 0) {
#pragma unroll
        for (int j = 0; j < JL; ++j) {
            const int qoff = __builtin_amdgcn_readlane(gtab.x, j);
#pragma unroll
            for (int o = 0; o < 2; ++o)
                idrec[((j * nwaves + wave) * 2 + o) * 64 + lane] = e.colo[(size_t)((qoff >> 1) + o) * 64 + lane];
        }
    }
    // Vertices / ranks without an LDS slot (id 0xFFFF) read the zero slot and write a trash slot:
    // straight-line code whose LDS accesses can all be in flight together, instead of a branch
    // and a round trip per access.
    auto rd_slot = [&](unsigned id) -> unsigned { return id == 0xFFFFu ? (unsigned)e.zero_slot : id; };
    auto wr_slot = [&](unsigned id) -> unsigned { return id == 0xFFFFu ? (unsigned)e.zero_slot + 1u : id; };
    // One-off stagger of the workgroups of an XCD (blockIdx % 8 selects the XCD): started in
    // lock-step, all CUs would stream from L2, gather from LDS and write to HBM at the same
    // times; spread over roughly one step they overlap each other's phases instead.
    {
        // ... scaled with the number of groups a workgroup works through
        const int gpw = (ngrp + (int)gridDim.x - 1) / (int)gridDim.x;
        const int sx = (flags >> 20) & 0xFF;             // chebgcn_tune(4, x): stagger in 1/8 units of 640 cycles per rank (experiment)
        // measured best: 640 cycles per rank at 4 groups per workgroup, twice that from 16 groups on
        const int m8 = gpw <= 4 ? 8 : gpw >= 16 ? 16 : 8 + (8 * (gpw - 4)) / 12;
        // small shapes (several workgroups per CU, about one group each) are not staggered: -10 % there
        const int reps = ((blockIdx.x >> 3) & 31) * (sx ? sx - 1 : NTHR >= 512 ? m8 : 0) / 8;
        for (int i = 0; i < reps; ++i) __builtin_amdgcn_s_sleep(10);
    }

    auto plane_of = [&](int grp, int p) { const int i = grp * P + p; return i < nplanes ? i : nplanes - 1; };

    // linear staging registers: next group's input, G_j of the adjoint
    float4 pre[NQ][P];
    auto fetch = [&](const float* base, int grp) {                // request P planes, linear
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int q = tid + u * nthr;
#pragma unroll
            for (int p = 0; p < P; ++p) pre[u][p] = zero4;
            if (q < Mq) {
#pragma unroll
                for (int p = 0; p < P; ++p) pre[u][p] = ldg4(base + (size_t)plane_of(grp, p) * Mp + 4 * q);
            }
        }
    };
    // LDS -> P planes of slab `out`; isolated vertices get sign * xiso (forward) ------------
    auto copy_out_piece = [&](int u, float* out, int grp, float iso_sign, const float* xiso) {
        const int q = tid + u * nthr;
        if (q < Mq) {
            const uint2 nq = opaque(nsreg[u]);
            if constexpr (P == 4) {
                // two plane pairs one after the other (8-byte LDS reads): half the live registers
                unsigned sl[4];
                bool none[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned id = slot_of(nq, i);
                    none[i] = id == 0xFFFFu;
                    sl[i] = none[i] ? (unsigned)e.zero_slot : id;
                }
                // an isolated vertex (no slot) takes sign * x; the pad beyond M (no slot either) stays 0
                bool patch = false;
#pragma unroll
                for (int i = 0; i < 4; ++i) patch |= none[i] && 4 * q + i < M;
                patch = patch && iso_sign != 0.f;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float4 o[2];
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        o[p] = zero4;
                        if (patch) {
                            const float4 x = ldg4(xiso + (size_t)plane_of(grp, 2 * h + p) * Mp + 4 * q);
                            o[p] = make_float4(iso_sign * x.x, iso_sign * x.y, iso_sign * x.z, iso_sign * x.w);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float2 t = *reinterpret_cast<const float2*>(T + sl[i] * 4 + 2 * h);
                        const bool pad = 4 * q + i >= M;
                        set_comp(o[0], i, none[i] ? (pad ? 0.f : comp(o[0], i)) : t.x);
                        set_comp(o[1], i, none[i] ? (pad ? 0.f : comp(o[1], i)) : t.y);
                    }
#pragma unroll
                    for (int p = 0; p < 2; ++p)
                        if (grp * P + 2 * h + p < nplanes) stg4(out + (size_t)(grp * P + 2 * h + p) * Mp + 4 * q, o[p]);
                }
            } else {
                float4 o[P];
#pragma unroll
                for (int i = 0; i < 4; ++i) {                    // no branches: a vertex without a slot reads 0
                    const Ent<P> t = lds_get<P>(T, rd_slot(slot_of(nq, i)));
#pragma unroll
                    for (int p = 0; p < P; ++p) set_comp(o[p], i, t.x[p]);
                }
#pragma unroll
                for (int p = 0; p < P; ++p)
                    if (grp * P + p < nplanes) stg4(out + (size_t)(grp * P + p) * Mp + 4 * q, o[p]);
            }
        }
    };

    // operator ring (see the gather below).  Small graphs (NJ <= 2: every row group of a wave fits the ring) request their
    // records ONCE per workgroup and keep them for every step of every plane group: the ring is never refilled there,
    // and an atlas-sized layer (N = 360: 376 rows, K = 10) otherwise pays one exposed L2 round trip in each of its nine steps.
    constexpr int RING = 2 * QMAX;
    constexpr int QO = (QMAX + 1) / 2;       // id records (two quads each) per group in the ring
    constexpr int ORING = 2 * QO;
    uint4 ro[ORING];
    float4 rv[RING];
    bool ring_resident = false;
    int grp = blockIdx.x;
    const size_t in_base = ADJ ? (size_t)(K - 1) * slab : 0;
    __syncthreads();
    // The input of a plane group is requested while the previous group finishes (its last rotate
    // and copy-out), so the HBM latency and the burst of one workgroup's 2*P*Mp bytes are hidden.
    if (grp < ngrp) fetch(src + in_base, grp);
    for (; grp < ngrp; grp += gridDim.x) {
        // ---- input planes -> LDS image ----------------------------------------------------------
        CG_STAMP(0);
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int q = tid + u * nthr;
            if (q < Mq) {
                const uint2 nq = opaque(nsreg[u]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    Ent<P> t;
#pragma unroll
                    for (int p = 0; p < P; ++p) t.x[p] = comp(pre[u][p], i);
                    lds_put<P>(T, wr_slot(slot_of(nq, i)), t);
                }
                if (!ADJ && copy_t0) {            // T_0 = x goes straight to slab 0
#pragma unroll
                    for (int p = 0; p < P; ++p)
                        if (grp * P + p < nplanes) stg4(dst + (size_t)(grp * P + p) * Mp + 4 * q, pre[u][p]);
                }
            }
        }
        CG_STAMP(1);
        __syncthreads();
        CG_STAMP(2);

        Ent<P> st[NJ];                        // T_{k-2} of the own rows, replaced by T_k in place
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int p = 0; p < P; ++p) st[j].x[p] = 0.f;

        // Closes step `sdone`: once every wave has finished its gather, LDS <- T_k and the
        // registers <- T_{k-1} of the own rows (adjoint: then c_j += G_j).  Runs at the top of the
        // next step and, for the last step, after the loop.
        // Requests the next group's input (the else branch ends the live range of the old staging registers)
        auto fetch_next = [&]() {
            if (grp + (int)gridDim.x < ngrp) fetch(src + in_base, grp + gridDim.x);
            else {
#pragma unroll
                for (int u = 0; u < NQ; ++u)
#pragma unroll
                    for (int p = 0; p < P; ++p) pre[u][p] = zero4;
            }
        };
        auto finish_step = [&](int sdone, bool last) {
            CG_STAMP(4 * sdone + 0);
            // adjoint: G_j of the finished step (added after the rotate) is requested as soon as THIS wave's gather is done,
            // in front of the barrier: its latency runs under the wait for the slower waves and the rotate (0.652 -> 0.634 ms
            // at batch 256, 0.176 -> 0.172 ms inside the step).  Requested earlier still, with the last levels of the gather
            // (the wave's own operator requests are all out by then): 0.72 ms.
            if (ADJ) fetch(src + (size_t)(K - 1 - sdone) * slab, grp);
            __syncthreads();                                    // every gather (and copy-out read) of this step is done
            CG_STAMP(4 * sdone + 1);
            // forward: only now, behind the barrier -- HBM loads queued while other waves still
            // gather would hold up their operator loads (the vector memory pipeline returns in order)
            if (!ADJ && last) fetch_next();
            // ---- rotate: LDS <- T_k, registers <- T_{k-1} of the own rows -----------------
            {
                Ent<P> prev[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const unsigned r = (opaque(rowreg[j >> 1]) >> (16 * (j & 1))) & 0xFFFFu;
                    prev[j] = lds_get<P>(T, rd_slot(r));
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const unsigned r = (opaque(rowreg[j >> 1]) >> (16 * (j & 1))) & 0xFFFFu;
                    lds_put<P>(T, wr_slot(r), st[j]);
                    st[j] = prev[j];
                }
            }
            CG_STAMP(4 * sdone + 2);
            __syncthreads();
            CG_STAMP(4 * sdone + 3);
            if (ADJ) {
                // ---- c_j += G_j, linear --------------------------------------------------------
#pragma unroll
                for (int u = 0; u < NQ; ++u) {
                    const int q = tid + u * nthr;
                    if (q < Mq) {
                        const uint2 nq = opaque(nsreg[u]);
                        Ent<P> t[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) t[i] = lds_get<P>(T, rd_slot(slot_of(nq, i)));
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
#pragma unroll
                            for (int p = 0; p < P; ++p) t[i].x[p] += comp(pre[u][p], i);
                            lds_put<P>(T, wr_slot(slot_of(nq, i)), t[i]);
                        }
                    }
                }
                __syncthreads();
            }
                };
        for (int step = 1; step < K; ++step) {
            if (ADJ && step > 1) finish_step(step - 1, false);      // (forward: below, behind the first operator requests)
            const float f = ADJ ? (step == K - 1 ? 1.f : 2.f) : (step == 1 ? 1.f : 2.f);
            // forward: slab step-1 is written out while this step gathers; an isolated vertex has
            // T_k = 0 for odd k and (-1)^(k/2) x for even k
            const bool do_out = !ADJ && step > 1;
            float* out_slab = dst + (size_t)(step - 1) * slab;
            const int ko = step - 1;
            const float iso_sign = (ko & 1) ? 0.f : ((ko & 2) ? -1.f : 1.f);

            // ---- gather: st <- f * (L T_{k-1})[own rows] - st ---------------------------------
            // Operator entries travel through a ring of RING quads (4 entries of each of the 64
            // rows): every group stores QMAX zero-padded quads, quad n = QMAX*j + q lives in ring
            // slot n % RING and is requested RING quads (two groups) before it is gathered -- one
            // group of lead does not cover the L2 latency.  Requests are unconditional, so the
            // compiler can count the loads in flight (s_waitcnt vmcnt(N), N > 0).
            auto group_info = [&](int j, int& qoff, int& len) {
                // {quad offset, length} of group j*nwaves + wave, from lane j of the wave's table
                qoff = __builtin_amdgcn_readlane(gtab.x, j);
                len = __builtin_amdgcn_readlane(gtab.y, j);
            };
            // buffer loads: descriptor + uniform offset in SGPRs, lane offset in one VGPR -- no 64-bit
            // address arithmetic on the vector ALU.  Ids come eight at a time (one record per two
            // quads): the vector-memory path costs ~16 cycles per wave instruction whatever its width.
            auto request_ids = [&](int j, int o) {   // id record o of group j -> its ring slot
                int qoff, len;
                group_info(j, qoff, len);
                if (o >= 1 && len <= 10) return;               // the second record only beyond 10 entries (9..10: ids in the value record)
                if (j < JL && o < 2) {      // resident in LDS (j is a compile-time constant here)
                    ro[(QO * j + o) % ORING] = idrec[((j * nwaves + wave) * 2 + o) * 64 + lane];
                    return;
                }
                const u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(colo_rsrc, lane * 16, ((qoff >> 1) + o) * 1024, 0);
                ro[(QO * j + o) % ORING] = make_uint4(c.x, c.y, c.z, c.w);
            };
            auto request = [&](int j, int q) {       // values of quad q of group j -> their ring slot
                int qoff, len;
                group_info(j, qoff, len);
                if (q >= 2 && len <= 8) return;                // the third quad only where a row needs it
                const f32x4 v = __builtin_amdgcn_raw_buffer_load_b128(valq_rsrc, lane * 16, (qoff + q) * 1024, 0);
                rv[(QMAX * j + q) % RING] = make_float4(v.x, v.y, v.z, v.w);
            };
            auto ids_of = [&](int j, int q) {        // the four ids of quad q of group j
                const uint4 o = ro[(QO * j + (q >> 1)) % ORING];
                return (q & 1) ? make_uint2(o.z, o.w) : make_uint2(o.x, o.y);
            };
            auto pair = [&](const unsigned c, const float v0, const float v1, float (&acc)[P]) {
                const Ent<P> t0 = lds_at<P>(T, entry_ofs<P, 0>(c)), t1 = lds_at<P>(T, entry_ofs<P, 1>(c));
#pragma unroll
                for (int p = 0; p < P; ++p) acc[p] = fmaf(v0, t0.x[p], acc[p]);
#pragma unroll
                for (int p = 0; p < P; ++p) acc[p] = fmaf(v1, t1.x[p], acc[p]);
            };
            auto quad = [&](const uint2 c, const float4 v, float (&acc)[P]) {
                Ent<P> t[4];
                unsigned at[4] = {entry_ofs<P, 0>(c.x), entry_ofs<P, 1>(c.x), entry_ofs<P, 0>(c.y), entry_ofs<P, 1>(c.y)};
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    t[i] = lds_at<P>(T, at[i]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int p = 0; p < P; ++p) acc[p] = fmaf(comp(v, i), t[i].x[p], acc[p]);
            };
            if (NJ > 2 || !ring_resident) {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    if (jj >= NJ) break;
#pragma unroll
                    for (int o = 0; o < QO; ++o) request_ids(jj, o);
#pragma unroll
                    for (int q = 0; q < QMAX; ++q) request(jj, q);
                }
                ring_resident = true;
            }
            // forward: the operator records of the first two row groups are requested BEFORE the previous step is closed
            // (barrier, rotate, barrier), their L2 latency runs under it: 0.576 -> 0.568 ms at batch 256, 0.127 -> 0.125 ms
            // inside the step.  The adjoint, whose staging registers are live across that phase, spills 14 registers
            // with it (0.61 -> 0.645 ms) and closes the step first.
            if (!ADJ && step > 1) finish_step(step - 1, false);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                // the linear pieces of the previous slab go out with the last NQ levels, when the
                // operator ring is no longer being refilled (measured against spreading them over the
                // gather or sending them first: +1.5 % at B=256)
                if (!ADJ && do_out && j >= NJ - NQ) copy_out_piece(j - (NJ - NQ), out_slab, grp, iso_sign, src);
                // waves that are ahead yield to the ones behind, so that the waves of a SIMD reach
                // the barrier together instead of the oldest finishing early (fewer waves = less overlap)
                if (NJ >= 4 && (j == 0 || (4 * j) / NJ != (4 * (j - 1)) / NJ)) {
                    const int pr = 3 - (4 * j) / NJ;
                    if (pr == 3) __builtin_amdgcn_s_setprio(3);
                    else if (pr == 2) __builtin_amdgcn_s_setprio(2);
                    else if (pr == 1) __builtin_amdgcn_s_setprio(1);
                    else __builtin_amdgcn_s_setprio(0);
                }
                int qoff, len;
                group_info(j, qoff, len);
                float acc[P];
#pragma unroll
                for (int p = 0; p < P; ++p) acc[p] = 0.f;
#pragma unroll
                for (int q = 0; q < QMAX; ++q) {
                    // the first two quads always (zero-padded), the third for rows beyond 8 entries
                    // (the ring value passes through an opaque identity at its use: without it hipcc hoists the
                    // copy that prepares .w as a v_pk_fma operand into the block of the CONDITIONAL third-quad
                    // request, where it needs `s_waitcnt vmcnt(0)` right behind the load -- the whole operator
                    // ring drained once per level)
                    if (q == 2 && len > 8 && len <= 10) {
                        // a third quad of two entries: their ids are the .z word of the value record
                        const float4 v = opaque(rv[(QMAX * j + q) % RING]);
                        pair(__float_as_uint(v.z), v.x, v.y, acc);
                    } else if (q < 2 || len > 10) quad(ids_of(j, q), opaque(rv[(QMAX * j + q) % RING]), acc);
                    if (j + 2 < NJ) {
                        request(j + 2, q);                       // refill the slots just consumed
                        if ((q & 1) || q == QMAX - 1) request_ids(j + 2, q >> 1);
                    }
                }
                if (len > 4 * QMAX) {
                    for (int q = QMAX; 4 * q < len; ++q) {       // rows longer than 4*QMAX entries (rare)
                        const uint4 o = e.colo[(size_t)((qoff >> 1) + (q >> 1)) * 64 + lane];
                        const float4 v = e.valq[(size_t)(qoff + q) * 64 + lane];
                        quad((q & 1) ? make_uint2(o.z, o.w) : make_uint2(o.x, o.y), v, acc);
                    }
                }
#pragma unroll
                for (int p = 0; p < P; ++p) st[j].x[p] = fmaf(f, acc[p], -st[j].x[p]);
            }
        }
        finish_step(K - 1, true);
        if (ADJ) fetch_next();

        // ---- stream the last image out ---------------------------------------------------------
        if (!ADJ) {
            const int ko = K - 1;
            const float iso_sign = (ko & 1) ? 0.f : ((ko & 2) ? -1.f : 1.f);
#pragma unroll
            for (int u = 0; u < NQ; ++u) copy_out_piece(u, dst + (size_t)ko * slab, grp, iso_sign, src);
        } else {
            // dx; an isolated vertex has dx = G_0 - G_2 + G_4 - ...
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                const int q = tid + u * nthr;
                if (q < Mq) {
                    float4 o[P];
                    unsigned iso = 0;
                    const uint2 nq = opaque(nsreg[u]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const unsigned sl = slot_of(nq, i);
                        const Ent<P> t = lds_get<P>(T, rd_slot(sl));
                        if (sl == 0xFFFFu && 4 * q + i < M) iso |= 1u << i;
#pragma unroll
                        for (int p = 0; p < P; ++p) set_comp(o[p], i, t.x[p]);
                    }
                    if (P == 4 && iso != 0) {
                        float sgn = 1.f;
                        for (int m = 0; m < K; m += 2, sgn = -sgn) {
#pragma unroll
                            for (int p = 0; p < P; ++p) {
                                const float4 x = ldg4(src + (size_t)m * slab + (size_t)plane_of(grp, p) * Mp + 4 * q);
#pragma unroll
                                for (int i = 0; i < 4; ++i)
                                    if (iso & (1u << i)) set_comp(o[p], i, comp(o[p], i) + sgn * comp(x, i));
                            }
                        }
                    }
#pragma unroll
                    for (int p = 0; p < P; ++p)
                        if (grp * P + p < nplanes) stg4(dst + (size_t)(grp * P + p) * Mp + 4 * q, o[p]);
                }
            }
        }
        CG_STAMP(40);
        __syncthreads();                      // LDS reads done before the image is overwritten
        CG_STAMP(41);
    }
}

// Fallback for graphs whose LDS image does not fit (M > ~20k): one launch per step,
// one thread per (vertex, plane), gathers served by L1/L2.  out = f * (A src) - sub + add.
__global__ void __launch_bounds__(256)
cheb_step_global_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                        const float* __restrict__ val, const float* src, const float* sub,
                        const float* add, float* out, int M, int Mp, float f) {
    // `out` may alias `sub` (element-wise, same thread), never `src`
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t plane = (size_t)blockIdx.y * Mp;
    if (r >= M) return;
    float acc = 0.f;
    for (int e = rowptr[r]; e < rowptr[r + 1]; ++e) acc = fmaf(val[e], src[plane + col[e]], acc);
    float v = f * acc;
    if (sub) v -= sub[plane + r];
    if (add) v += add[plane + r];
    out[plane + r] = v;
}

template <int P, int NJ, int NQ, int NTHR, bool ADJ>
static int launch_onchip(const chebgcn_graph* g, const Ell& ell, const float* src, float* dst,
                         int nplanes, int K, int copy_t0, hipStream_t stream) {
    const size_t lds = (size_t)lds_capacity(NJ, NTHR, P) * P * sizeof(float) +
                       (size_t)lds_id_levels(NJ, NTHR, P) * (NTHR >> 6) * 2 * 1024;
    static_assert(NQ <= NJ, "a linear piece is issued per group at most");
    auto kern = cheb_onchip_kernel<P, NJ, NQ, NTHR, ADJ>;
    static const std::string name = "cheb_onchip_kernel<" + std::to_string(P) + "," + std::to_string(NJ) + "," + std::to_string(NQ) +
                                    "," + std::to_string(NTHR) + "," + (ADJ ? "true" : "false") + ">";
    note_dispatch(name.c_str());
    int per_cu = (int)((160 * 1024) / lds);
    per_cu = per_cu < 1 ? 1 : per_cu;
    if (per_cu > 2048 / NTHR) per_cu = 2048 / NTHR;
    if (per_cu > 8) per_cu = 8;
    const int ngrp = (nplanes + P - 1) / P;
    int grid = g->num_cus * per_cu;
    if (grid > ngrp) grid = ngrp;
    const size_t slab = (size_t)nplanes * g->Mp;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHR), 0, stream, view(ell), src, dst, g->M, g->Mp,
                       nplanes, K, slab, copy_t0 | (g_stagger << 20));
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

// Workgroup shape: as many waves as the register budget allows.  Per thread the kernel keeps
// P VGPRs per row slice (NJ) and 4P per linear piece (NQ = ceil(NJ/4)) next to ~70 for the
// operator entries and temporaries; 1024 / 768 / 512 threads may use 128 / 168 / 256 VGPRs.
template <int P, bool ADJ>
static int dispatch_onchip(const chebgcn_graph* g, const Ell& ell, const float* src, float* dst, int nplanes, int K,
                           int copy_t0, hipStream_t stream) {
    const int rows = ell.ngroups * 64;
    const int Mq = g->Mp / 4;
    auto fits = [&](int nj, int nq, int nthr) {
        return nj * nthr >= rows && nq * nthr >= Mq && ell.lds_entries <= lds_capacity(nj, nthr, P);
    };
#define CG_TRY(NJ, NQ, NTHR) if (fits(NJ, NQ, NTHR)) return launch_onchip<P, NJ, NQ, NTHR, ADJ>(g, ell, src, dst, nplanes, K, copy_t0, stream)
    // <= 2048 rows (the shapes with more linear pieces than a quarter of the rows serve four planes only: there the isolated
    // vertices have pieces but no rows; with two planes every vertex is a row and the shape before them always fits)
    CG_TRY(1, 1, 256); CG_TRY(2, 1, 256); CG_TRY(4, 1, 256);
    if constexpr (P == 4) { CG_TRY(4, 2, 256); }
    CG_TRY(8, 2, 256);
    if constexpr (P == 4) { CG_TRY(8, 3, 256); }
    if constexpr (P == 4) {
        // beyond 2048 rows: the dedicated kernel of recurrence4.hip
        return dispatch_onchip4<ADJ>(g, ell, src, dst, nplanes, K, copy_t0, stream);
    } else {
        CG_TRY(8, 3, 512); CG_TRY(8, 3, 768); CG_TRY(11, 4, 768);
        if (g_wide) { CG_TRY(11, 3, 1024); }                                                  // experiment: 16 waves
        CG_TRY(14, 4, 768);                                                                   // <= 10752
        CG_TRY(24, 7, 512); CG_TRY(32, 9, 512); CG_TRY(40, 11, 512);                         // <= 20480
    }
#undef CG_TRY
    return fail(CHEBGCN_EUNSUPPORTED, "recurrence: no kernel shape for %d rows x %d planes", rows, P);
}

static int step_global(const chebgcn_graph* g, const Ell& ell, const float* src, const float* sub,
                       const float* add, float* out, int nplanes, float f, hipStream_t stream) {
    dim3 grid((g->M + 255) / 256, nplanes);
    note_dispatch("cheb_step_global_kernel");
    hipLaunchKernelGGL(cheb_step_global_kernel, grid, dim3(256), 0, stream, ell.rowptr, ell.col32,
                       ell.cval, src, sub, add, out, g->M, g->Mp, f);
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

}  // namespace chebgcn

using namespace chebgcn;

#ifdef CG_EXPERIMENT
// Experiment builds only (tools/xbuild.sh, -DCG_EXPERIMENT): knobs for tools/kbench.py.  The shipped
// library does not export them (tests/test_abi_and_host.py checks the export list both ways).
extern "C" int chebgcn_debug_stamps(long long* out) {       // CG_X & 64 builds only (tools/kbench.py --stamps)
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(chebgcn::g_dbg), sizeof(long long) * 16 * 64) == hipSuccess ? 0 : -1;
}

extern "C" int chebgcn_tune(int key, int value) {
    if (key == 3) { g_wide = value; return 0; }
    if (key == 4) { g_stagger = value & 0xFF; return 0; }
    return -1;
}
#endif

extern "C" int chebgcn_recurrence_fwd(const chebgcn_graph* g, const float* x, float* stack, int B,
                                      int Fin, int K, chebgcn_stream stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CG_REQUIRE(g && x && stack, "recurrence_fwd: NULL argument");
    CG_REQUIRE(B > 0 && Fin > 0 && K >= 1, "recurrence_fwd: bad shape B=%d Fin=%d K=%d", B, Fin, K);
    CG_REQUIRE((int64_t)B * Fin < (1 << 30), "recurrence_fwd: too many planes");
    const int nplanes = B * Fin;
    const size_t slab = (size_t)nplanes * g->Mp;
    const int copy_t0 = (x != stack);
    if (K > 1 && g->ord_ok && ordered_fits(g, nplanes)) return dispatch_ordered<false>(g, g->ofwd, x, stack, nplanes, K, copy_t0, stream);
    if (K == 1 || !g->lds_ok) {
        if (copy_t0) CG_HIP(hipMemcpyAsync(stack, x, slab * sizeof(float), hipMemcpyDeviceToDevice, stream));
        if (K == 1) return CHEBGCN_OK;
        int rc = step_global(g, g->fwd, stack, nullptr, nullptr, stack + slab, nplanes, 1.f, stream);
        for (int k = 2; k < K && rc == CHEBGCN_OK; ++k)
            rc = step_global(g, g->fwd, stack + (k - 1) * slab, stack + (k - 2) * slab, nullptr,
                             stack + k * slab, nplanes, 2.f, stream);
        return rc;
    }
    const Ell& ell = pick_ell(g, false, nplanes);
    if (ell.planes == 4) return dispatch_onchip<4, false>(g, ell, x, stack, nplanes, K, copy_t0, stream);
    return dispatch_onchip<2, false>(g, ell, x, stack, nplanes, K, copy_t0, stream);
}

// T_k(L~^T) x: the forward recurrence on the TRANSPOSED operator (the images every handle carries for the adjoint).  With it the
// gradient of a layer wrt its input is  dx = sum_k [T_k(L~^T) dy] W_k^T : recurrence on the Fout planes of dy, then a contraction
// of the stack of dy with the re-indexed W -- the same sum as the Clenshaw form (chebgcn_contract_bwd_x + chebgcn_recurrence_bwd),
// associated the other way round: for Fout <= Fin it moves no more bytes and runs on the two faster kernels.
extern "C" int chebgcn_recurrence_fwd_t(const chebgcn_graph* g, const float* x, float* stack, int B,
                                        int Fin, int K, chebgcn_stream stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CG_REQUIRE(g && x && stack, "recurrence_fwd_t: NULL argument");
    CG_REQUIRE(B > 0 && Fin > 0 && K >= 1, "recurrence_fwd_t: bad shape B=%d Fin=%d K=%d", B, Fin, K);
    CG_REQUIRE((int64_t)B * Fin < (1 << 30), "recurrence_fwd_t: too many planes");
    const int nplanes = B * Fin;
    const size_t slab = (size_t)nplanes * g->Mp;
    const int copy_t0 = (x != stack);
    if (K > 1 && g->ord_ok && ordered_fits(g, nplanes)) return dispatch_ordered<false>(g, g->oadj, x, stack, nplanes, K, copy_t0, stream);
    if (K == 1 || !g->lds_ok) {
        if (copy_t0) CG_HIP(hipMemcpyAsync(stack, x, slab * sizeof(float), hipMemcpyDeviceToDevice, stream));
        if (K == 1) return CHEBGCN_OK;
        int rc = step_global(g, g->adj, stack, nullptr, nullptr, stack + slab, nplanes, 1.f, stream);
        for (int k = 2; k < K && rc == CHEBGCN_OK; ++k)
            rc = step_global(g, g->adj, stack + (k - 1) * slab, stack + (k - 2) * slab, nullptr,
                             stack + k * slab, nplanes, 2.f, stream);
        return rc;
    }
    const Ell& ell = pick_ell(g, true, nplanes, 0);
    if (ell.planes == 4) return dispatch_onchip<4, false>(g, ell, x, stack, nplanes, K, copy_t0, stream);
    return dispatch_onchip<2, false>(g, ell, x, stack, nplanes, K, copy_t0, stream);
}

extern "C" int chebgcn_recurrence_bwd(const chebgcn_graph* g, const float* gstack, float* dx, int B,
                                      int Fin, int K, chebgcn_stream stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CG_REQUIRE(g && gstack && dx, "recurrence_bwd: NULL argument");
    CG_REQUIRE(B > 0 && Fin > 0 && K >= 1, "recurrence_bwd: bad shape B=%d Fin=%d K=%d", B, Fin, K);
    CG_REQUIRE((int64_t)B * Fin < (1 << 30), "recurrence_bwd: too many planes");
    const int nplanes = B * Fin;
    const size_t slab = (size_t)nplanes * g->Mp;
    if (K == 1) {
        CG_HIP(hipMemcpyAsync(dx, gstack, slab * sizeof(float), hipMemcpyDeviceToDevice, stream));
        return CHEBGCN_OK;
    }
    // (graphs of at most 2048 active vertices -- the 256-thread ordered shapes: the Clenshaw adjoint of this file is the faster one there,
    // 0.063 against 0.072 ms at N = 1000, 0.161 against 0.175 at N = 2000, batch 128, K = 10; it runs in any vertex order)
    const bool small_ord = g->ord_ok && g->oadj.ord_NT <= 256;
    if (g->ord_ok && ordered_fits(g, nplanes) && !small_ord)
        return dispatch_ordered<true>(g, g->oadj, gstack, dx, nplanes, K, 0, stream);
    if (g->lds_ok) {
        const Ell& ell = pick_ell(g, true, nplanes);
        if (ell.planes == 4) return dispatch_onchip<4, true>(g, ell, gstack, dx, nplanes, K, 0, stream);
        return dispatch_onchip<2, true>(g, ell, gstack, dx, nplanes, K, 0, stream);
    }
    // fallback: Clenshaw with two scratch slabs; c_{j} = G_j + f L^T c_{j+1} - c_{j+2}
    float* scratch = nullptr;
    CG_HIP(hipMallocAsync((void**)&scratch, 2 * slab * sizeof(float), stream));
    const float* c1 = gstack + (size_t)(K - 1) * slab;   // c_{j+1}
    const float* c2 = nullptr;                           // c_{j+2}
    int rc = CHEBGCN_OK;
    for (int j = K - 2; j >= 0 && rc == CHEBGCN_OK; --j) {
        float* out = (j == 0) ? dx : scratch + (size_t)(j & 1) * slab;
        rc = step_global(g, g->adj, c1, c2, gstack + (size_t)j * slab, out, nplanes, j == 0 ? 1.f : 2.f, stream);
        c2 = c1;
        c1 = out;
    }
    (void)hipFreeAsync(scratch, stream);
    return rc;
}
