// Chebyshev three-term recurrence over a fixed sparse operator, forward and adjoint.
//
//   forward  (lib_new/models_gcn.py:598-610):  T_0 = x, T_1 = L T_0, T_k = 2 L T_{k-1} - T_{k-2}
//   adjoint  (TF autodiff of the above):       c_{K-1} = G_{K-1}, c_j = G_j + 2 L^T c_{j+1} - c_{j+2},
//                                              dx = G_0 + L^T c_1 - c_2
//
// Data layout: planes.  A plane is one (window, feature) column of the reference's
// x0[M, Fin*N] matrix: Mp contiguous floats, vertex-fastest.  Planes are independent
// under the recurrence, so a workgroup owns TWO planes (one float2 per vertex) and
// runs all K-1 steps for them on chip:
//   * T_{k-1} of both planes lives in LDS (8 B per vertex; the gather source),
//   * T_{k-2}, overwritten in place by the freshly computed T_k, of the rows a thread owns
//     lives in VGPRs,
//   * HBM sees each plane exactly once per slab: x is read once, every T_k written once
//     (compulsory traffic 4*M*Fin*K bytes per window instead of 4*M*Fin*(3K-4) for a
//     kernel-per-step SpMM).
// The operator comes as a length-sorted sliced ELL (graph.hip): the 64 rows handled by
// one wave have (nearly) equal length, so the slot loop has a wave-uniform trip count
// and no divergence; column/value loads are coalesced 128 B / 256 B per wave.  All
// operator entries of a group are requested before the first LDS gather so that one L2
// round trip covers the whole group.
// Because rows are handed to lanes in length order, results are scattered back into the
// LDS image and streamed out linearly, which keeps every HBM access fully coalesced.
// Software pipeline (one workgroup per CU, so nothing else hides latency):
//   * the linear copy-out of T_{k-1} (LDS -> HBM) is interleaved with the gather of step k,
//     each piece issued right after a group's operator loads so that later waits on those
//     loads never cover the stores;
//   * in the adjoint, the G_j planes needed after the gather are requested the same way;
//   * the next plane pair's input is requested before the final copy-out of the current one.
#include "common.h"

namespace chebgcn {

int g_ablate = 0;   // set through chebgcn_tune(0, bits) by tools/kbench.py; 0 in production

struct EllView {
    const int2* ginfo;
    const uint2* colq;
    const float4* valq;
    const uint16_t* rowslot;
    const uint16_t* nodeslot;
    int ngroups, zero_slot;
};

static inline EllView view(const Ell& e) {
    return EllView{e.ginfo, e.colq, e.valq, e.rowslot, e.nodeslot, e.ngroups, e.zero_slot};
}

constexpr int QMAX = 3;      // quads (4 operator entries each) requested ahead per group

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void stg4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// i-th 16-bit slot id of a packed quad
__device__ __forceinline__ unsigned slot_of(uint2 c, int i) {
    const unsigned w = (i & 2) ? c.y : c.x;
    return (i & 1) ? (w >> 16) : (w & 0xFFFFu);
}
__device__ __forceinline__ float comp(float4 v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

// operator entries of one 64-row group, as requested from L2
struct Ops {
    uint2 c[QMAX];
    float4 v[QMAX];
    int len;                 // even length of the group (wave-uniform), 0 if the group does not exist
    int qoff;
};

// One workgroup = all rows x 2 planes.  NJ = row slices per thread (ceil(ngroups*64 / blockDim)),
// NQ = 16-byte linear pieces per thread and plane (ceil(Mp/4 / blockDim) <= ceil(NJ/4)).
template <int NJ, int NTHR, bool ADJ>
__global__ void __launch_bounds__(NTHR)
cheb_onchip_kernel(EllView e, const float* __restrict__ src, float* __restrict__ dst,
                   int M, int Mp, int nplanes, int K, size_t slab, int flags, int lds_entries) {
    extern __shared__ __attribute__((aligned(16))) float2 T[];   // [lds_entries] slot-indexed image, then tables
    constexpr int NQ = (NJ + 3) / 4;
    const int copy_t0 = flags & 1;
    // ablation bits for tools/kbench.py (always 0 in production):
    // 1 = no global stores, 2 = no gather, 16 = no global loads, 32 = stagger start (x (abl>>6)&7)
    const int abl = flags >> 8;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    constexpr int nthr = NTHR;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int nwaves = NTHR >> 6;
    const int Mq = Mp >> 2;                                       // float4 pieces per plane
    const int npairs = (nplanes + 1) >> 1;
    const float2 zero2 = make_float2(0.f, 0.f);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    // Index tables, copied into LDS once per workgroup (they would cost ~12 VGPRs otherwise):
    //   rs[g*64 + lane] = LDS slot of the row that lane owns in group g (0xFFFF = none)
    //   nsq[q]          = LDS slots of vertices 4q..4q+3 (4 x 16 bit; 0xFFFF = pad)
    uint16_t* rs = reinterpret_cast<uint16_t*>(T + lds_entries);
    uint2* nsq = reinterpret_cast<uint2*>(rs + e.ngroups * 64);
    for (int i = tid; i < e.ngroups * 32; i += nthr)
        reinterpret_cast<unsigned*>(rs)[i] = reinterpret_cast<const unsigned*>(e.rowslot)[i];
    for (int q = tid; q < Mq; q += nthr) nsq[q] = reinterpret_cast<const uint2*>(e.nodeslot)[q];
    if (tid == 0) T[e.zero_slot] = zero2;     // never written again
    __syncthreads();
    // De-correlate the CUs of an XCD (blockIdx % 8 selects the XCD): every workgroup streams the
    // same operator image from L2; started in lock-step they all hit the same L2 channel at the
    // same time.  A one-off stagger spreads them over one step's worth of time.
    if (abl & 32) {
        const int k = (blockIdx.x >> 3) & 31;
        const int reps = k * (1 + ((abl >> 6) & 7));      // x ~0.3 us each
        for (int i = 0; i < reps; ++i) __builtin_amdgcn_s_sleep(10);
    }

    // linear staging registers: next pair's input (both modes), G_j of the adjoint
    float4 pa[NQ], pb[NQ];

    auto fetch = [&](const float* a, const float* b) {           // request 2 planes, linear
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int q = tid + u * nthr;
            pa[u] = zero4;
            pb[u] = zero4;
            if (q < Mq && !(abl & 16)) {
                pa[u] = ldg4(a + 4 * q);
                pb[u] = ldg4(b + 4 * q);
            }
        }
    };
    auto copy_out_piece = [&](int u, float* o0, float* o1, bool has1) {   // LDS -> 2 planes
        const int q = tid + u * nthr;
        if (q < Mq && !(abl & 1)) {
            float2 t[4];
            const uint2 nq = nsq[q];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned sl = slot_of(nq, i);
                t[i] = (sl != 0xFFFFu) ? T[sl] : zero2;
            }
            stg4(o0 + 4 * q, make_float4(t[0].x, t[1].x, t[2].x, t[3].x));
            if (has1) stg4(o1 + 4 * q, make_float4(t[0].y, t[1].y, t[2].y, t[3].y));
        }
    };
    auto load_ops = [&](Ops& o, int g) {
        o.len = 0;
        o.qoff = 0;
        if (g < e.ngroups && !(abl & 2)) {
            const int2 gi = e.ginfo[g];
            o.qoff = gi.x;
            o.len = gi.y;
#pragma unroll
            for (int q = 0; q < QMAX; ++q)
                if (4 * q < gi.y) {
                    o.c[q] = e.colq[(size_t)(gi.x + q) * 64 + lane];
                    o.v[q] = e.valq[(size_t)(gi.x + q) * 64 + lane];
                }
        }
    };
    auto pair_fma = [&](float2& acc, uint2 c, float4 v, int i) {   // entries i, i+1 of a quad
        const float2 t0 = T[slot_of(c, i)], t1 = T[slot_of(c, i + 1)];
        const float v0 = comp(v, i), v1 = comp(v, i + 1);
        acc.x = fmaf(v0, t0.x, acc.x); acc.y = fmaf(v0, t0.y, acc.y);
        acc.x = fmaf(v1, t1.x, acc.x); acc.y = fmaf(v1, t1.y, acc.y);
    };
    auto consume = [&](const Ops& o) -> float2 {
        float2 acc = zero2;
        const int len = o.len;
        if (len >= 8) {                                   // the common case: one batch of eight
            float2 t[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] = T[slot_of(o.c[i >> 2], i & 3)];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float vv = comp(o.v[i >> 2], i & 3);
                acc.x = fmaf(vv, t[i].x, acc.x); acc.y = fmaf(vv, t[i].y, acc.y);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; i += 2)
                if (i < len) pair_fma(acc, o.c[i >> 2], o.v[i >> 2], i & 3);
        }
#pragma unroll
        for (int i = 8; i < 4 * QMAX; i += 2)
            if (i < len) pair_fma(acc, o.c[i >> 2], o.v[i >> 2], i & 3);
        for (int q = QMAX; 4 * q < len; ++q) {            // rows longer than 4*QMAX entries (rare)
            const uint2 c = e.colq[(size_t)(o.qoff + q) * 64 + lane];
            const float4 v = e.valq[(size_t)(o.qoff + q) * 64 + lane];
            pair_fma(acc, c, v, 0);
            if (4 * q + 2 < len) pair_fma(acc, c, v, 2);
        }
        return acc;
    };

    int pair = blockIdx.x;
    if (pair < npairs) {
        const size_t base = ADJ ? (size_t)(K - 1) * slab : 0;
        const int p0 = 2 * pair, p1 = (p0 + 1 < nplanes) ? p0 + 1 : p0;
        fetch(src + base + (size_t)p0 * Mp, src + base + (size_t)p1 * Mp);
    }
    for (; pair < npairs; pair += gridDim.x) {
        const int p0 = 2 * pair;
        const bool has1 = (p0 + 1) < nplanes;
        const int p1 = has1 ? p0 + 1 : p0;

        // ---- staged input -> LDS image ----------------------------------------------------------
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int q = tid + u * nthr;
            if (q < Mq) {
                const float4 a = pa[u], b = pb[u];
                const uint2 nq = nsq[q];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned sl = slot_of(nq, i);
                    if (sl != 0xFFFFu) T[sl] = make_float2(comp(a, i), comp(b, i));
                }
            }
        }
        __syncthreads();

        float2 st[NJ];                        // T_{k-2} of the own rows, replaced by T_k in place
#pragma unroll
        for (int j = 0; j < NJ; ++j) st[j] = zero2;

        for (int step = 1; step < K; ++step) {
            const float f = ADJ ? (step == K - 1 ? 1.f : 2.f) : (step == 1 ? 1.f : 2.f);
            // linear traffic interleaved with this step's gather
            const bool do_out = !ADJ && (step > 1 || copy_t0);    // forward: write T_{step-1}
            float* o0 = dst + (size_t)(step - 1) * slab + (size_t)p0 * Mp;
            float* o1 = dst + (size_t)(step - 1) * slab + (size_t)p1 * Mp;
            const float* g0 = src + (size_t)(K - 1 - step) * slab + (size_t)p0 * Mp;   // adjoint: G_j
            const float* g1 = src + (size_t)(K - 1 - step) * slab + (size_t)p1 * Mp;

            // ---- gather: st <- f * (L T_{k-1})[own rows] - st ---------------------------------
            // the operator entries of group j+1 are requested before group j is consumed
            Ops ops[2];
            load_ops(ops[0], wave);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (j + 1 < NJ) load_ops(ops[(j + 1) & 1], (j + 1) * nwaves + wave);
                if ((j & 3) == 0 && (j >> 2) < NQ) {       // one linear piece per four groups
                    const int u = j >> 2;
                    if (ADJ) {
                        const int q = tid + u * nthr;
                        pa[u] = zero4;
                        pb[u] = zero4;
                        if (q < Mq && !(abl & 16)) {
                            pa[u] = ldg4(g0 + 4 * q);
                            pb[u] = ldg4(g1 + 4 * q);
                        }
                    } else if (do_out) {
                        copy_out_piece(u, o0, o1, has1);
                    }
                }
                const float2 acc = consume(ops[j & 1]);
                st[j].x = fmaf(f, acc.x, -st[j].x);
                st[j].y = fmaf(f, acc.y, -st[j].y);
                __builtin_amdgcn_sched_barrier(0);          // keep the prefetch distance at one group
            }
            __syncthreads();                  // every gather (and copy-out read) of this step is done
            // ---- rotate: LDS <- T_k, registers <- T_{k-1} of the own rows -----------------
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int g = j * nwaves + wave;
                const unsigned r = (g < e.ngroups) ? rs[g * 64 + lane] : 0xFFFFu;
                if (r != 0xFFFFu) {
                    const float2 old = T[r];
                    T[r] = st[j];
                    st[j] = old;
                }
            }
            __syncthreads();
            if (ADJ) {
                // ---- c_j += G_j, linear --------------------------------------------------------
#pragma unroll
                for (int u = 0; u < NQ; ++u) {
                    const int q = tid + u * nthr;
                    if (q < Mq) {
                        const float4 a = pa[u], b = pb[u];
                        const uint2 nq = nsq[q];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const unsigned sl = slot_of(nq, i);
                            if (sl != 0xFFFFu) {
                                const float2 t = T[sl];
                                T[sl] = make_float2(t.x + comp(a, i), t.y + comp(b, i));
                            }
                        }
                    }
                }
                __syncthreads();
            }
        }

        // ---- request the next pair's input, then stream the last image out -----------------
        const int npair = pair + gridDim.x;
        if (npair < npairs) {
            const size_t base = ADJ ? (size_t)(K - 1) * slab : 0;
            const int q0 = 2 * npair, q1 = (q0 + 1 < nplanes) ? q0 + 1 : q0;
            fetch(src + base + (size_t)q0 * Mp, src + base + (size_t)q1 * Mp);
        }
        {
            float* o0 = dst + (ADJ ? 0 : (size_t)(K - 1) * slab) + (size_t)p0 * Mp;
            float* o1 = dst + (ADJ ? 0 : (size_t)(K - 1) * slab) + (size_t)p1 * Mp;
#pragma unroll
            for (int u = 0; u < NQ; ++u) copy_out_piece(u, o0, o1, has1);
        }
        __syncthreads();                      // LDS reads done before the image is overwritten
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

template <int NJ, int NTHR, bool ADJ>
static int launch_onchip(const chebgcn_graph* g, const Ell& ell, const float* src, float* dst,
                         int nplanes, int K, int copy_t0, hipStream_t stream) {
    const size_t lds = (size_t)ell.lds_entries * sizeof(float2) + (size_t)ell.ngroups * 64 * 2 + (size_t)(g->Mp / 4) * 8;
    auto kern = cheb_onchip_kernel<NJ, NTHR, ADJ>;
    CG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = (int)((160 * 1024) / lds);
    per_cu = per_cu < 1 ? 1 : per_cu;
    if (per_cu > 2048 / NTHR) per_cu = 2048 / NTHR;
    if (per_cu > 8) per_cu = 8;
    const int npairs = (nplanes + 1) / 2;
    int grid = g->num_cus * per_cu;
    if (grid > npairs) grid = npairs;
    const size_t slab = (size_t)nplanes * g->Mp;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHR), lds, stream, view(ell), src, dst, g->M, g->Mp,
                       nplanes, K, slab, copy_t0 | (g_ablate << 8), ell.lds_entries);
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

// Workgroup shape: as many waves as the register budget allows.  Per thread the kernel keeps
// 2 VGPRs per row slice (NJ) and 8 per linear piece (NQ = ceil(NJ/4)) next to ~80 for the
// operator prefetch and temporaries; 1024 / 768 / 512 threads may use 128 / 168 / 256 VGPRs.
template <bool ADJ>
static int dispatch_onchip(const chebgcn_graph* g, const float* src, float* dst, int nplanes, int K,
                           int copy_t0, hipStream_t stream) {
    const Ell& ell = ADJ ? g->adj : g->fwd;
    const int rows = ell.ngroups * 64;
    const int Mq = g->Mp / 4;
    auto fits = [&](int nj, int nthr) { return nj * nthr >= rows && ((nj + 3) / 4) * nthr >= Mq; };
#define CG_TRY(NJ, NTHR) if (fits(NJ, NTHR)) return launch_onchip<NJ, NTHR, ADJ>(g, ell, src, dst, nplanes, K, copy_t0, stream)
    CG_TRY(1, 256); CG_TRY(2, 256); CG_TRY(4, 256); CG_TRY(8, 256);      // M <= 2048
    CG_TRY(8, 512); CG_TRY(8, 1024);                                        // M <= 8192
    CG_TRY(11, 768); CG_TRY(14, 768);                                       // M <= 10752
    CG_TRY(24, 512); CG_TRY(32, 512); CG_TRY(40, 512);                      // M <= 20480
#undef CG_TRY
    return fail(CHEBGCN_EUNSUPPORTED, "recurrence: no kernel shape for %d rows", rows);
}

static int step_global(const chebgcn_graph* g, const Ell& ell, const float* src, const float* sub,
                       const float* add, float* out, int nplanes, float f, hipStream_t stream) {
    dim3 grid((g->M + 255) / 256, nplanes);
    hipLaunchKernelGGL(cheb_step_global_kernel, grid, dim3(256), 0, stream, ell.rowptr, ell.col32,
                       ell.cval, src, sub, add, out, g->M, g->Mp, f);
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

}  // namespace chebgcn

using namespace chebgcn;

// Undeclared tuning hook for tools/kbench.py (not part of the ABI in include/chebgcn.h).
extern "C" int chebgcn_tune(int key, int value) {
    if (key == 0) { g_ablate = value; return 0; }
    return -1;
}

extern "C" int chebgcn_recurrence_fwd(const chebgcn_graph* g, const float* x, float* stack, int B,
                                      int Fin, int K, chebgcn_stream stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CG_REQUIRE(g && x && stack, "recurrence_fwd: NULL argument");
    CG_REQUIRE(B > 0 && Fin > 0 && K >= 1, "recurrence_fwd: bad shape B=%d Fin=%d K=%d", B, Fin, K);
    CG_REQUIRE((int64_t)B * Fin < (1 << 30), "recurrence_fwd: too many planes");
    const int nplanes = B * Fin;
    const size_t slab = (size_t)nplanes * g->Mp;
    const int copy_t0 = (x != stack);
    if (K == 1 || !g->lds_ok) {
        if (copy_t0) CG_HIP(hipMemcpyAsync(stack, x, slab * sizeof(float), hipMemcpyDeviceToDevice, stream));
        if (K == 1) return CHEBGCN_OK;
        int rc = step_global(g, g->fwd, stack, nullptr, nullptr, stack + slab, nplanes, 1.f, stream);
        for (int k = 2; k < K && rc == CHEBGCN_OK; ++k)
            rc = step_global(g, g->fwd, stack + (k - 1) * slab, stack + (k - 2) * slab, nullptr,
                             stack + k * slab, nplanes, 2.f, stream);
        return rc;
    }
    return dispatch_onchip<false>(g, x, stack, nplanes, K, copy_t0, stream);
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
    if (g->lds_ok) return dispatch_onchip<true>(g, gstack, dx, nplanes, K, 0, stream);
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
