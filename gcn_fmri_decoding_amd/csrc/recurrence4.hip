// Chebyshev recurrence on chip, FOUR planes per workgroup (gfx950) -- the variant for graphs of
// roughly 2k..10k active vertices, where 16 bytes per active vertex still fit the 160 KB LDS.
//
// Same algorithm and data formats as recurrence.hip (see there and common.h); what differs:
//   * an LDS entry holds the four planes of one vertex, so one operator entry (2 B slot id +
//     4 B value, streamed from L2) and one 16-byte ds_read_b128 serve four planes: half the
//     operator stream, half the address arithmetic and half the LDS instructions per plane;
//   * only ACTIVE vertices (non-empty row or column of the operator) have an LDS slot.  An
//     isolated vertex obeys T_k = -T_{k-2}: T_k = 0 for odd k and (-1)^(k/2) x for even k; the
//     copy-out code patches those lanes from x (adjoint: dx = G_0 - G_2 + G_4 - ...);
//   * 512 threads with up to 256 VGPRs each: 20 rows x 4 planes of T_{k-2} state per thread;
//   * the LDS image is a static array (its address folds into the ds_read, the slot id becomes
//     a byte offset with one SDWA shift), all streaming addresses are uniform base + 32-bit
//     lane offset.
// Reference semantics: lib_new/models_gcn.py:587-617 (chebyshev5), forward; the adjoint is the
// Clenshaw form of its gradient (oracle/layers_ref.py chebyshev5_bwd).
#include "common.h"

namespace chebgcn {

extern int g_ablate;

namespace {

#ifndef CG_X
#define CG_X 0               // timing experiments only (tools/xbuild.sh); results are wrong when non-zero
#endif

constexpr int QMAX = kQuadMin;   // quads stored for every group and requested one group ahead

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
// uniform base + 32-bit byte offset: global_load/store with an SGPR base
__device__ __forceinline__ float4 ldg4(const float* ubase, unsigned byteoff) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(ubase) + byteoff);
}
__device__ __forceinline__ void stg4(float* ubase, unsigned byteoff, float4 v) {
    *reinterpret_cast<float4*>(reinterpret_cast<char*>(ubase) + byteoff) = v;
}
__device__ __forceinline__ unsigned slot_of(uint2 c, int i) {
    const unsigned w = (i & 2) ? c.y : c.x;
    return (i & 1) ? (w >> 16) : (w & 0xFFFFu);
}
__device__ __forceinline__ float comp(float4 v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
__device__ __forceinline__ void set_comp(float4& v, int i, float x) {
    if (i == 0) v.x = x; else if (i == 1) v.y = x; else if (i == 2) v.z = x; else v.w = x;
}
// Identity the optimiser cannot see through: keeps values DERIVED from the small per-thread
// tables (unpacked slot ids, flags) from being hoisted out of the plane-group loop, where they
// would occupy ~80 registers for the whole kernel and spill.
__device__ __forceinline__ unsigned opaque(unsigned x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ uint2 opaque(uint2 x) { asm volatile("" : "+v"(x.x), "+v"(x.y)); return x; }
__device__ __forceinline__ float4 fma4(float s, float4 t, float4 a) {
    return make_float4(fmaf(s, t.x, a.x), fmaf(s, t.y, a.y), fmaf(s, t.z, a.z), fmaf(s, t.w, a.w));
}

// ENT = LDS entries (16 B each), NJ = row slices per thread, NQ = linear 4-vertex pieces per
// thread and plane, NTHR = workgroup size.
template <int ENT, int NJ, int NQ, int NTHR, bool ADJ>
__global__ void __launch_bounds__(NTHR)
cheb4_kernel(EllView e, const float* __restrict__ src, float* __restrict__ dst, int M, int Mp, int nplanes,
             int K, size_t slab, int flags) {
    __shared__ float4 T[ENT];                        // slot-indexed: the four planes of one vertex
    constexpr int nwaves = NTHR >> 6;
    constexpr int QS = NJ / NQ;                      // a linear piece is copied out every QS groups
    static_assert(QS >= 1 && NJ <= 64, "shape");
    const int copy_t0 = flags & 1;
    const int abl = flags >> 8;                      // tools/kbench.py: 1 no stores, 2 no gather, 16 no loads
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Mq = Mp >> 2;
    const int ngrp = (nplanes + 3) >> 2;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto lds = [&](unsigned byteoff) -> float4 {
        return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(T) + byteoff);
    };

    // ---- per-thread tables, loaded once -----------------------------------------------------
    constexpr int NJ2 = (NJ + 1) / 2;
    unsigned rowreg[NJ2];                            // LDS slot of the own row of slice j (two per register)
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
    uint2 nsreg[NQ];                                 // LDS slots of the 4 vertices of linear piece u
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        const int q = tid + u * NTHR;
        nsreg[u] = (q < Mq) ? reinterpret_cast<const uint2*>(e.nodeslot)[q] : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
    }
    int2 gtab = make_int2(0, 0);                     // lane j: {quad offset, length} of group j*nwaves + wave
    if (lane < NJ && lane * nwaves + wave < e.ngroups) gtab = e.ginfo[lane * nwaves + wave];
    if (tid == 0) T[e.zero_slot] = zero4;            // never written again
    if (!(abl & 32)) {                               // spread the workgroups of an XCD over one step
        const int reps = 2 * ((blockIdx.x >> 3) & 31);
        for (int i = 0; i < reps; ++i) __builtin_amdgcn_s_sleep(10);
    }
    const size_t in_base = ADJ ? (size_t)(K - 1) * slab : 0;
    __syncthreads();

    for (int grp = blockIdx.x; grp < ngrp; grp += gridDim.x) {
        // uniform plane offsets; planes beyond nplanes alias the last one and are never stored
        size_t pl[4];
        bool pv[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int i = grp * 4 + p;
            pv[p] = i < nplanes;
            pl[p] = (size_t)(pv[p] ? i : nplanes - 1) * Mp;
        }
        float4 pre[NQ][4];                           // linear staging: input planes, G_j of the adjoint
        auto fetch = [&](const float* base) {
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                const int q = tid + u * NTHR;
                const unsigned qb = opaque((unsigned)q * 16u);     // not hoisted, not folded into a 64-bit base
#pragma unroll
                for (int p = 0; p < 4; ++p) pre[u][p] = zero4;
                if (q < Mq && !(abl & 16)) {
#pragma unroll
                    for (int p = 0; p < 4; ++p) pre[u][p] = ldg4(base + pl[p], qb);
                }
            }
        };
        // LDS image -> the four planes of slab `out` (forward), two plane pairs one after the other
        // (8-byte LDS reads: half the live registers).  Isolated vertices have no slot: in an odd
        // slab they are 0 like the zero slot; their values in the even slabs were stored when the
        // input was staged, so there (`keep`) a piece that contains one stores its other
        // components one by one.
        auto copy_out = [&](int u, float* out, bool keep) {
            const int q = tid + u * NTHR;
            const unsigned qb = opaque((unsigned)q * 16u);     // not hoisted, not folded into a 64-bit base
            if (q < Mq && !(abl & 1)) {
                const uint2 nq = opaque(nsreg[u]);
                unsigned at[4];
                bool iso[4];
                bool any_iso = false;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned id = slot_of(nq, i);
                    const bool none = id == 0xFFFFu;
                    iso[i] = none && 4 * q + i < M;
                    any_iso |= iso[i];
                    at[i] = (none ? (unsigned)e.zero_slot : id) * 16u;
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float4 o[2];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float2 t = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(T) + at[i] + 8 * h);
                        set_comp(o[0], i, t.x);
                        set_comp(o[1], i, t.y);
                    }
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        if (!pv[2 * h + p]) continue;
                        float* plane = out + pl[2 * h + p];
                        if (keep && any_iso) {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (!iso[i]) *reinterpret_cast<float*>(reinterpret_cast<char*>(plane) + qb + 4 * i) = comp(o[p], i);
                        } else {
                            stg4(plane, qb, o[p]);
                        }
                    }
                }
            }
        };

        // ---- input planes -> LDS image (forward: and straight to slab 0) --------------------
        fetch(src + in_base);
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int q = tid + u * NTHR;
                const unsigned qb = opaque((unsigned)q * 16u);     // not hoisted, not folded into a 64-bit base
            if (q < Mq) {
                const uint2 nq = opaque(nsreg[u]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned id = slot_of(nq, i);
                    if (id != 0xFFFFu)
                        T[id] = make_float4(comp(pre[u][0], i), comp(pre[u][1], i), comp(pre[u][2], i), comp(pre[u][3], i));
                }
                if (!ADJ && !(abl & 1)) {
                    if (copy_t0) {
#pragma unroll
                        for (int p = 0; p < 4; ++p)
                            if (pv[p]) stg4(dst + pl[p], qb, pre[u][p]);
                    }
                    // an isolated vertex has T_k = (-1)^(k/2) x in the even slabs: stored here, once
                    bool iso[4];
                    bool any_iso = false;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        iso[i] = slot_of(nq, i) == 0xFFFFu && 4 * q + i < M;
                        any_iso |= iso[i];
                    }
                    if (any_iso) {
                        float sgn = -1.f;
                        for (int k = 2; k < K; k += 2, sgn = -sgn) {
#pragma unroll
                            for (int p = 0; p < 4; ++p) {
                                if (!pv[p]) continue;
                                char* plane = reinterpret_cast<char*>(dst + (size_t)k * slab + pl[p]) + qb;
#pragma unroll
                                for (int i = 0; i < 4; ++i)
                                    if (iso[i]) *reinterpret_cast<float*>(plane + 4 * i) = sgn * comp(pre[u][p], i);
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();

        float4 st[NJ];                               // T_{k-2} of the own rows, replaced by T_k in place
#pragma unroll
        for (int j = 0; j < NJ; ++j) st[j] = zero4;

        for (int step = 1; step < K; ++step) {
            const float f = ADJ ? (step == K - 1 ? 1.f : 2.f) : (step == 1 ? 1.f : 2.f);
            // forward: slab step-1 is written out while this step gathers
            const bool do_out = !ADJ && step > 1;
            float* out_slab = dst + (size_t)(step - 1) * slab;
            const bool keep = ((step - 1) & 1) == 0;           // even slab: isolated vertices already stored

            // ---- gather: st <- f * (A T_{k-1})[own rows] - st -----------------------------------
            // Operator entries travel through a ring of RING quads (4 entries of each of the 64
            // rows): every group stores QMAX zero-padded quads, quad n = QMAX*j + q lives in ring
            // slot n % RING and is requested RING quads (two groups) before it is gathered -- one
            // group of lead does not cover the L2 latency.  Requests are unconditional, so the
            // compiler can count the loads in flight (s_waitcnt vmcnt(N), N > 0).
            constexpr int RING = 2 * QMAX;
            uint2 rc[RING];
            float4 rv[RING];
            auto group_info = [&](int j, int& qoff, int& len) {
                qoff = __builtin_amdgcn_readlane(gtab.x, j);
                len = __builtin_amdgcn_readlane(gtab.y, j);
                if (abl & 2) { qoff = 0; len = 0; }
                if (CG_X & 1) qoff = 0;
            };
            auto request = [&](int j, int q) {                   // quad q of group j -> its ring slot
                int qoff, len;
                group_info(j, qoff, len);
                rc[(QMAX * j + q) % RING] = e.colq[(size_t)(qoff + q) * 64 + lane];
                rv[(QMAX * j + q) % RING] = e.valq[(size_t)(qoff + q) * 64 + lane];
            };
            auto quad = [&](const uint2 c, const float4 v, float4& acc) {
                unsigned a0 = ofs_lo(c.x), a1 = ofs_hi(c.x), a2 = ofs_lo(c.y), a3 = ofs_hi(c.y);
                if (CG_X & 2) { a0 = lane * 16; a1 = a0 + 1024; a2 = a0 + 2048; a3 = a0 + 3072; }
                const float4 t0 = lds(a0), t1 = lds(a1), t2 = lds(a2), t3 = lds(a3);
                acc = fma4(v.x, t0, acc);
                acc = fma4(v.y, t1, acc);
                acc = fma4(v.z, t2, acc);
                acc = fma4(v.w, t3, acc);
            };
#pragma unroll
            for (int n = 0; n < RING; ++n)
                if (n / QMAX < NJ) request(n / QMAX, n % QMAX);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (do_out && (j % QS) == 0 && (j / QS) < NQ) copy_out(j / QS, out_slab, keep);
                int qoff, len;
                group_info(j, qoff, len);
                float4 acc = zero4;
                const bool gather = !(abl & 2);
#pragma unroll
                for (int q = 0; q < QMAX; ++q) {
                    // the first two quads always (zero-padded), the third for rows beyond 8 entries
                    if (gather && (q < 2 || len > 8)) quad(rc[(QMAX * j + q) % RING], rv[(QMAX * j + q) % RING], acc);
                    if (j + 2 < NJ) request(j + 2, q);           // refill the slot just consumed
                }
                if (gather && len > 4 * QMAX) {
                    for (int q = QMAX; 4 * q < len; ++q) {           // rows longer than 4*QMAX entries (rare)
                        const uint2 c = e.colq[(size_t)(qoff + q) * 64 + lane];
                        const float4 v = e.valq[(size_t)(qoff + q) * 64 + lane];
                        quad(c, v, acc);
                    }
                }
                st[j] = make_float4(fmaf(f, acc.x, -st[j].x), fmaf(f, acc.y, -st[j].y), fmaf(f, acc.z, -st[j].z),
                                    fmaf(f, acc.w, -st[j].w));
            }
            if (ADJ) fetch(src + (size_t)(K - 1 - step) * slab);      // G_j, added after the rotate
            __syncthreads();                         // every gather (and copy-out read) of this step is done
            // ---- rotate: LDS <- T_k, registers <- T_{k-1} of the own rows ------------------------
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const unsigned r = (opaque(rowreg[j >> 1]) >> (16 * (j & 1))) & 0xFFFFu;
                if (r != 0xFFFFu) {
                    const float4 old = T[r];
                    T[r] = st[j];
                    st[j] = old;
                }
            }
            __syncthreads();
            if (ADJ) {
                // ---- c_j += G_j, linear ---------------------------------------------------------
#pragma unroll
                for (int u = 0; u < NQ; ++u) {
                    const int q = tid + u * NTHR;
                    if (q < Mq) {
                        const uint2 nq = opaque(nsreg[u]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const unsigned id = slot_of(nq, i);
                            if (id != 0xFFFFu) {
                                float4 t = T[id];
                                t.x += comp(pre[u][0], i);
                                t.y += comp(pre[u][1], i);
                                t.z += comp(pre[u][2], i);
                                t.w += comp(pre[u][3], i);
                                T[id] = t;
                            }
                        }
                    }
                }
                __syncthreads();
            }
        }

        // ---- stream the last image out ---------------------------------------------------------
        if (!ADJ) {
            const int ko = K - 1;
#pragma unroll
            for (int u = 0; u < NQ; ++u) copy_out(u, dst + (size_t)ko * slab, (ko & 1) == 0);
        } else {
            // dx; an isolated vertex has dx = G_0 - G_2 + G_4 - ...
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                const int q = tid + u * NTHR;
                const unsigned qb = opaque((unsigned)q * 16u);     // not hoisted, not folded into a 64-bit base
                if (q < Mq && !(abl & 1)) {
                    const uint2 nq = opaque(nsreg[u]);
                    float4 t[4];
                    bool iso[4];
                    bool patch = false;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const unsigned id = slot_of(nq, i);
                        const bool none = id == 0xFFFFu;
                        iso[i] = none && 4 * q + i < M;
                        patch |= iso[i];
                        t[i] = T[none ? (unsigned)e.zero_slot : id];
                    }
                    float4 o[4];
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        o[p] = make_float4(comp(t[0], p), comp(t[1], p), comp(t[2], p), comp(t[3], p));
                    if (patch) {
                        float sgn = 1.f;
                        for (int m = 0; m < K; m += 2, sgn = -sgn) {
#pragma unroll
                            for (int p = 0; p < 4; ++p) {
                                const float4 x = ldg4(src + (size_t)m * slab + pl[p], qb);
#pragma unroll
                                for (int i = 0; i < 4; ++i)
                                    if (iso[i]) set_comp(o[p], i, comp(o[p], i) + sgn * comp(x, i));
                            }
                        }
                    }
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        if (pv[p]) stg4(dst + pl[p], qb, o[p]);
                }
            }
        }
        __syncthreads();                             // LDS reads done before the image is overwritten
    }
}

template <int ENT, int NJ, int NQ, int NTHR, bool ADJ>
int launch4(const chebgcn_graph* g, const Ell& ell, const float* src, float* dst, int nplanes, int K, int copy_t0,
            hipStream_t stream) {
    const int per_cu = (160 * 1024) / (ENT * 16);
    const int ngrp = (nplanes + 3) / 4;
    int grid = g->num_cus * (per_cu < 1 ? 1 : per_cu);
    if (grid > ngrp) grid = ngrp;
    const size_t slab = (size_t)nplanes * g->Mp;
    hipLaunchKernelGGL((cheb4_kernel<ENT, NJ, NQ, NTHR, ADJ>), dim3(grid), dim3(NTHR), 0, stream, view(ell), src, dst,
                       g->M, g->Mp, nplanes, K, slab, copy_t0 | (g_ablate << 8));
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

// shape 0 = none, 1 = 5120 entries, 512 threads x 10 rows, 2 = 10240 entries, 768 threads x 14 rows
int shape4(int lds_entries, int rows, int Mq) {
    if (rows <= 2048) return 0;                      // small graphs: the generic kernel of recurrence.hip
    if (lds_entries <= 5120 && rows <= 10 * 512 && Mq <= 3 * 512) return 1;
    if (lds_entries <= 10240 && rows <= 14 * 768 && Mq <= 4 * 768) return 2;
    return 0;
}

}  // namespace

bool onchip4_fits(int lds_entries, int rows, int Mq) { return shape4(lds_entries, rows, Mq) != 0; }

template <bool ADJ>
int dispatch_onchip4(const chebgcn_graph* g, const float* src, float* dst, int nplanes, int K, int copy_t0,
                     hipStream_t stream) {
    const Ell& ell = ADJ ? g->adj : g->fwd;
    switch (shape4(ell.lds_entries, ell.ngroups * 64, g->Mp / 4)) {
        case 1: return launch4<5120, 10, 3, 512, ADJ>(g, ell, src, dst, nplanes, K, copy_t0, stream);
        case 2: return launch4<10240, 14, 4, 768, ADJ>(g, ell, src, dst, nplanes, K, copy_t0, stream);
        default: break;
    }
    return fail(CHEBGCN_EUNSUPPORTED, "recurrence: no four-plane kernel shape for %d rows", ell.ngroups * 64);
}

template int dispatch_onchip4<false>(const chebgcn_graph*, const float*, float*, int, int, int, hipStream_t);
template int dispatch_onchip4<true>(const chebgcn_graph*, const float*, float*, int, int, int, hipStream_t);

}  // namespace chebgcn
