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
// Clenshaw form of its gradient (see recurrence.hip).
#include "common.h"

namespace chebgcn {

extern int g_ablate;
__device__ long long g_dbg4[16 * 64];            // phase stamps of CG_X & 64 builds (tools/kbench.py --stamps)

namespace {

#ifndef CG_X
#define CG_X 0               // timing experiments only (tools/xbuild.sh); results are wrong when non-zero
#endif
#ifndef CG_ABL
#define CG_ABL 0
#endif

constexpr int QMAX = kQuadMin;   // quads stored for every group and requested one group ahead

#define CG_STAMP(id)                                                                          \
    do {                                                                                      \
        if ((CG_X & 64) && (id) < 64 && lane == 0 && blockIdx.x == 37 && grp == (int)(blockIdx.x + gridDim.x)) \
            g_dbg4[wave * 64 + (id)] = (long long)__builtin_readcyclecounter();                \
    } while (0)

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
__device__ __forceinline__ float ldg1(const float* ubase, unsigned byteoff) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(ubase) + byteoff);
}
__device__ __forceinline__ void stg1(float* ubase, unsigned byteoff, float v) {
    *reinterpret_cast<float*>(reinterpret_cast<char*>(ubase) + byteoff) = v;
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

// ENT = LDS entries (16 B each), NJ = row slices per thread, NV = vertices per thread in the
// linear (streaming) phases, NTHR = workgroup size.
//
// Streaming phases: lane l of the workgroup owns the vertices v = tid + u*NTHR (u < NV).  A wave
// therefore touches 64 consecutive vertices of a plane with one 4-byte access per lane (256 B,
// two full cache lines) and -- slots being numbered in vertex order -- 64 consecutive LDS
// entries with one conflict-free 16-byte access per lane.  No transposes, no per-component
// branches: a vertex without a slot is handled by predicating its lane.
template <int ENT, int NJ, int NV, int NTHR, bool ADJ>
__global__ void __launch_bounds__(NTHR)
cheb4_kernel(EllView e, const float* __restrict__ src, float* __restrict__ dst, int M, int Mp, int nplanes,
             int K, size_t slab, int flags) {
    __shared__ float4 T[ENT];                        // slot-indexed: the four planes of one vertex
    constexpr int nwaves = NTHR >> 6;
    static_assert(NJ <= 64 && NV <= 2 * NJ, "shape");
    const int copy_t0 = flags & 1;
    const int abl = CG_ABL ? flags >> 8 : 0;         // tools/kbench.py (xbuild only): 1 no stores, 2 no gather, 16 no loads
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    constexpr int NV2 = (NV + 1) / 2;
    unsigned vsreg[NV2];                             // LDS slot of the own vertex u (two per register), 0xFFFF = none
#pragma unroll
    for (int u2 = 0; u2 < NV2; ++u2) {
        unsigned r = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int v = tid + (2 * u2 + h) * NTHR;
            const unsigned id = (2 * u2 + h < NV && v < Mp) ? e.nodeslot[v] : 0xFFFFu;
            r |= id << (16 * h);
        }
        vsreg[u2] = r;
    }
    auto vslot = [&](int u) -> unsigned { return (opaque(vsreg[u >> 1]) >> (16 * (u & 1))) & 0xFFFFu; };
    int2 gtab = make_int2(0, 0);                     // lane j: {quad offset, length} of group j*nwaves + wave
    if (lane < NJ && lane * nwaves + wave < e.ngroups) gtab = e.ginfo[lane * nwaves + wave];
    if (tid == 0) T[e.zero_slot] = zero4;            // never written again
    // no slot (id 0xFFFF): read the zero slot, write the trash slot -- straight-line LDS code
    auto rd_slot = [&](unsigned id) -> unsigned { return id == 0xFFFFu ? (unsigned)e.zero_slot : id; };
    auto wr_slot = [&](unsigned id) -> unsigned { return id == 0xFFFFu ? (unsigned)e.zero_slot + 1u : id; };
    if (!(abl & 32)) {                               // spread the workgroups of an XCD over one step
        // ... scaled with the number of groups a workgroup works through
        const int gpw = (ngrp + (int)gridDim.x - 1) / (int)gridDim.x;
        const int m8 = gpw <= 4 ? 8 : gpw >= 16 ? 16 : 8 + (8 * (gpw - 4)) / 12;
        const int reps = ((blockIdx.x >> 3) & 31) * m8 / 8;
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
        // LDS image -> the four planes of slab `out` (forward).  A vertex without a slot reads
        // the zero slot; isolated vertices were stored into the even slabs when the input was
        // staged, so there (`keep`) their lanes do not store.
        auto copy_out = [&](int u, float* out, bool keep) {
            const int v = tid + u * NTHR;
            const unsigned vb = opaque((unsigned)v * 4u);
            if (v < Mp && !(abl & 1)) {
                const unsigned id = vslot(u);
                const bool none = id == 0xFFFFu;
                const float4 t = T[none ? (unsigned)e.zero_slot : id];
                if (!(keep && none && v < M)) {
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        if (pv[p]) stg1(out + pl[p], vb, comp(t, p));
                }
            }
        };

        // ---- input planes -> LDS image (forward: and straight to slab 0) --------------------
        CG_STAMP(0);
        constexpr int NVH = (NV + 1) / 2;            // two batches: half the staging registers
#pragma unroll
        for (int u0 = 0; u0 < NV; u0 += NVH) {
            float4 x[NVH];
#pragma unroll
            for (int uu = 0; uu < NVH; ++uu) {
                const int v = tid + (u0 + uu) * NTHR;
                const unsigned vb = opaque((unsigned)v * 4u);
                x[uu] = zero4;
                if (u0 + uu < NV && v < Mp && !(abl & 16))
                    x[uu] = make_float4(ldg1(src + in_base + pl[0], vb), ldg1(src + in_base + pl[1], vb),
                                        ldg1(src + in_base + pl[2], vb), ldg1(src + in_base + pl[3], vb));
            }
#pragma unroll
            for (int uu = 0; uu < NVH; ++uu) {
                const int v = tid + (u0 + uu) * NTHR;
                const unsigned vb = opaque((unsigned)v * 4u);
                if (u0 + uu < NV && v < Mp) {
                    const unsigned id = vslot(u0 + uu);
                    T[wr_slot(id)] = x[uu];
                    if (!ADJ && !(abl & 1)) {
                        if (copy_t0) {
#pragma unroll
                            for (int p = 0; p < 4; ++p)
                                if (pv[p]) stg1(dst + pl[p], vb, comp(x[uu], p));
                        }
                        // an isolated vertex has T_k = (-1)^(k/2) x in the even slabs: stored here, once
                        if (id == 0xFFFFu && v < M) {
                            float sgn = -1.f;
                            for (int k = 2; k < K; k += 2, sgn = -sgn) {
#pragma unroll
                                for (int p = 0; p < 4; ++p)
                                    if (pv[p]) stg1(dst + (size_t)k * slab + pl[p], vb, sgn * comp(x[uu], p));
                            }
                        }
                    }
                }
            }
        }
        CG_STAMP(1);
        __syncthreads();
        CG_STAMP(2);

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
                if ((CG_X & 4) && j > 1) return;               // experiment: no operator loads after the prologue
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
                if (do_out) {
                    // the vertices u in [j*NV/NJ, (j+1)*NV/NJ) go out with this group: small store
                    // packets spread over the whole gather
                    const int ua = j * NV / NJ, ub = (j + 1) * NV / NJ;
                    if (ua < ub) copy_out(ua, out_slab, keep);
                    if (ua + 1 < ub) copy_out(ua + 1, out_slab, keep);
                }
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
            float4 gj[ADJ ? NV : 1];                 // G_j of the adjoint, added after the rotate
            if (ADJ) {
                const float* gs = src + (size_t)(K - 1 - step) * slab;
#pragma unroll
                for (int u = 0; u < NV; ++u) {
                    const int v = tid + u * NTHR;
                    const unsigned vb = opaque((unsigned)v * 4u);
                    gj[u] = zero4;
                    if (v < Mp && !(abl & 16))
                        gj[u] = make_float4(ldg1(gs + pl[0], vb), ldg1(gs + pl[1], vb), ldg1(gs + pl[2], vb),
                                            ldg1(gs + pl[3], vb));
                }
            }
            CG_STAMP(4 * step + 0);
            __syncthreads();                         // every gather (and copy-out read) of this step is done
            CG_STAMP(4 * step + 1);
            // ---- rotate: LDS <- T_k, registers <- T_{k-1} of the own rows ------------------------
            {
                float4 prev[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const unsigned r = (opaque(rowreg[j >> 1]) >> (16 * (j & 1))) & 0xFFFFu;
                    prev[j] = T[rd_slot(r)];
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const unsigned r = (opaque(rowreg[j >> 1]) >> (16 * (j & 1))) & 0xFFFFu;
                    T[wr_slot(r)] = st[j];
                    st[j] = prev[j];
                }
            }
            CG_STAMP(4 * step + 2);
            __syncthreads();
            CG_STAMP(4 * step + 3);
            if (ADJ) {
                // ---- c_j += G_j, linear ---------------------------------------------------------
#pragma unroll
                for (int u = 0; u < NV; ++u) {
                    const unsigned id = vslot(u);
                    float4 t = T[rd_slot(id)];
                    t.x += gj[u].x;
                    t.y += gj[u].y;
                    t.z += gj[u].z;
                    t.w += gj[u].w;
                    T[wr_slot(id)] = t;
                }
                __syncthreads();
            }
        }

        // ---- stream the last image out ---------------------------------------------------------
        if (!ADJ) {
            const int ko = K - 1;
#pragma unroll
            for (int u = 0; u < NV; ++u) copy_out(u, dst + (size_t)ko * slab, (ko & 1) == 0);
        } else {
            // dx; an isolated vertex has dx = G_0 - G_2 + G_4 - ...
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int v = tid + u * NTHR;
                const unsigned vb = opaque((unsigned)v * 4u);
                if (v < Mp && !(abl & 1)) {
                    const unsigned id = vslot(u);
                    const bool none = id == 0xFFFFu;
                    float4 t = T[none ? (unsigned)e.zero_slot : id];
                    if (none && v < M) {
                        float sgn = 1.f;
                        for (int m = 0; m < K; m += 2, sgn = -sgn) {
                            const float* gs = src + (size_t)m * slab;
                            t.x += sgn * ldg1(gs + pl[0], vb);
                            t.y += sgn * ldg1(gs + pl[1], vb);
                            t.z += sgn * ldg1(gs + pl[2], vb);
                            t.w += sgn * ldg1(gs + pl[3], vb);
                        }
                    }
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        if (pv[p]) stg1(dst + pl[p], vb, comp(t, p));
                }
            }
        }
        CG_STAMP(40);
        __syncthreads();                             // LDS reads done before the image is overwritten
        CG_STAMP(41);
    }
}

template <int ENT, int NJ, int NV, int NTHR, bool ADJ>
int launch4(const chebgcn_graph* g, const Ell& ell, const float* src, float* dst, int nplanes, int K, int copy_t0,
            hipStream_t stream) {
    const int per_cu = (160 * 1024) / (ENT * 16);
    const int ngrp = (nplanes + 3) / 4;
    int grid = g->num_cus * (per_cu < 1 ? 1 : per_cu);
    if (grid > ngrp) grid = ngrp;
    const size_t slab = (size_t)nplanes * g->Mp;
    hipLaunchKernelGGL((cheb4_kernel<ENT, NJ, NV, NTHR, ADJ>), dim3(grid), dim3(NTHR), 0, stream, view(ell), src, dst,
                       g->M, g->Mp, nplanes, K, slab, copy_t0 | (g_ablate << 8));
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

// shape 0 = none, 1 = 5120 entries, 512 threads x 10 rows, 2 = 10240 entries, 768 threads x 14 rows
int shape4(int lds_entries, int rows, int Mq) {
    if (rows <= 2048) return 0;                      // small graphs: the generic kernel of recurrence.hip
    if (lds_entries <= 5120 && rows <= 10 * 512 && Mq * 4 <= 12 * 512) return 1;
    if (lds_entries <= 10240 && rows <= 14 * 768 && Mq * 4 <= 14 * 768) return 2;
    return 0;
}

}  // namespace

bool onchip4_fits(int lds_entries, int rows, int Mq) { return shape4(lds_entries, rows, Mq) != 0; }

template <bool ADJ>
int dispatch_onchip4(const chebgcn_graph* g, const float* src, float* dst, int nplanes, int K, int copy_t0,
                     hipStream_t stream) {
    const Ell& ell = ADJ ? g->adj : g->fwd;
    switch (shape4(ell.lds_entries, ell.ngroups * 64, g->Mp / 4)) {
        case 1: return launch4<5120, 10, 12, 512, ADJ>(g, ell, src, dst, nplanes, K, copy_t0, stream);
        case 2: return launch4<10240, 14, 14, 768, ADJ>(g, ell, src, dst, nplanes, K, copy_t0, stream);
        default: break;
    }
    return fail(CHEBGCN_EUNSUPPORTED, "recurrence: no four-plane kernel shape for %d rows", ell.ngroups * 64);
}

template int dispatch_onchip4<false>(const chebgcn_graph*, const float*, float*, int, int, int, hipStream_t);
template int dispatch_onchip4<true>(const chebgcn_graph*, const float*, float*, int, int, int, hipStream_t);

}  // namespace chebgcn

extern "C" int chebgcn_debug_stamps4(long long* out) {      // CG_X & 64 builds only
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(chebgcn::g_dbg4), sizeof(long long) * 16 * 64) == hipSuccess ? 0 : -1;
}
