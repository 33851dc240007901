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
//   * T_{k-2} and the freshly computed T_k of the rows a thread owns live in VGPRs,
//   * HBM sees each plane exactly once per slab: x is read once, every T_k written once
//     (compulsory traffic 4*M*Fin*K bytes per window instead of 4*M*Fin*(3K-4) for a
//     kernel-per-step SpMM).
// The operator comes as a length-sorted sliced ELL (graph.hip): the 64 rows handled by
// one wave have (nearly) equal length, so the slot loop has a wave-uniform trip count
// and no divergence; column/value loads are coalesced 128 B / 256 B per wave.
// Because rows are handed to lanes in length order, results are scattered back into the
// LDS image and streamed out linearly, which keeps every HBM access fully coalesced.
#include "common.h"

namespace chebgcn {

struct EllView {
    const int32_t* goff;
    const uint16_t* col16;
    const float* val;
    const int32_t* rowid;
    int ngroups;
};

static inline EllView view(const Ell& e) { return EllView{e.goff, e.col16, e.val, e.rowid, e.ngroups}; }

// One workgroup = all rows x 2 planes.  NJ = row slices per thread (ceil(ngroups*64 / blockDim)).
template <int NJ, bool ADJ>
__global__ void __launch_bounds__(1024)
cheb_onchip_kernel(EllView e, const float* __restrict__ src, float* __restrict__ dst,
                   int M, int Mp, int nplanes, int K, size_t slab, int copy_t0) {
    extern __shared__ __attribute__((aligned(16))) float2 T[];   // [Mp + 4]; T[M..] == 0
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int nthr = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = nthr >> 6;
    const int Mq = Mp >> 2;                                       // float4 groups per plane
    const int npairs = (nplanes + 1) >> 1;

    int row[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        int g = j * nwaves + wave;
        row[j] = (g < e.ngroups) ? e.rowid[g * 64 + lane] : -1;
    }

    for (int pair = blockIdx.x; pair < npairs; pair += gridDim.x) {
        const int p0 = 2 * pair;
        const bool has1 = (p0 + 1) < nplanes;
        const int p1 = has1 ? p0 + 1 : p0;
        // fwd: src = x;  adj: src = gstack, start from slab K-1
        const float* s0 = src + (ADJ ? (size_t)(K - 1) * slab : 0) + (size_t)p0 * Mp;
        const float* s1 = src + (ADJ ? (size_t)(K - 1) * slab : 0) + (size_t)p1 * Mp;

        __syncthreads();                      // previous pair's LDS reads are done
        for (int q = tid; q < Mq + 1; q += nthr) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
            if (q < Mq) {
                a = *reinterpret_cast<const float4*>(s0 + 4 * q);
                b = *reinterpret_cast<const float4*>(s1 + 4 * q);
            }
            const int i = 4 * q;
            T[i + 0] = (i + 0 < M) ? make_float2(a.x, b.x) : make_float2(0.f, 0.f);
            T[i + 1] = (i + 1 < M) ? make_float2(a.y, b.y) : make_float2(0.f, 0.f);
            T[i + 2] = (i + 2 < M) ? make_float2(a.z, b.z) : make_float2(0.f, 0.f);
            T[i + 3] = (i + 3 < M) ? make_float2(a.w, b.w) : make_float2(0.f, 0.f);
            if (!ADJ && copy_t0 && q < Mq) {
                *reinterpret_cast<float4*>(dst + (size_t)p0 * Mp + 4 * q) = a;
                if (has1) *reinterpret_cast<float4*>(dst + (size_t)p1 * Mp + 4 * q) = b;
            }
        }
        __syncthreads();

        float2 tm2[NJ], tnew[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) tm2[j] = make_float2(0.f, 0.f);

        for (int step = 1; step < K; ++step) {
            const float f = ADJ ? (step == K - 1 ? 1.f : 2.f) : (step == 1 ? 1.f : 2.f);
            // ---- gather: acc = (L T_{k-1})[own rows] -------------------------------------
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int g = j * nwaves + wave;            // wave-uniform
                float2 acc = make_float2(0.f, 0.f);
                if (g < e.ngroups) {
                    const int off = e.goff[g];
                    const int len = e.goff[g + 1] - off;
                    const uint16_t* cp = e.col16 + (size_t)off * 64 + lane;
                    const float* vp = e.val + (size_t)off * 64 + lane;
                    int s = 0;
                    for (; s + 4 <= len; s += 4) {
                        const int c0 = cp[(s + 0) * 64], c1 = cp[(s + 1) * 64];
                        const int c2 = cp[(s + 2) * 64], c3 = cp[(s + 3) * 64];
                        const float v0 = vp[(s + 0) * 64], v1 = vp[(s + 1) * 64];
                        const float v2 = vp[(s + 2) * 64], v3 = vp[(s + 3) * 64];
                        const float2 t0 = T[c0], t1 = T[c1], t2 = T[c2], t3 = T[c3];
                        acc.x = fmaf(v0, t0.x, acc.x); acc.y = fmaf(v0, t0.y, acc.y);
                        acc.x = fmaf(v1, t1.x, acc.x); acc.y = fmaf(v1, t1.y, acc.y);
                        acc.x = fmaf(v2, t2.x, acc.x); acc.y = fmaf(v2, t2.y, acc.y);
                        acc.x = fmaf(v3, t3.x, acc.x); acc.y = fmaf(v3, t3.y, acc.y);
                    }
                    for (; s < len; ++s) {
                        const int c0 = cp[s * 64];
                        const float v0 = vp[s * 64];
                        const float2 t0 = T[c0];
                        acc.x = fmaf(v0, t0.x, acc.x); acc.y = fmaf(v0, t0.y, acc.y);
                    }
                }
                tnew[j].x = fmaf(f, acc.x, -tm2[j].x);
                tnew[j].y = fmaf(f, acc.y, -tm2[j].y);
            }
            __syncthreads();                  // every gather of this step has read LDS
            // ---- rotate: LDS <- T_k, registers <- T_{k-1} of the own rows -----------------
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (row[j] >= 0) {
                    tm2[j] = T[row[j]];
                    T[row[j]] = tnew[j];
                }
            }
            __syncthreads();
            if (!ADJ) {
                // ---- stream T_k out, linear and coalesced ---------------------------------
                float* o0 = dst + (size_t)step * slab + (size_t)p0 * Mp;
                float* o1 = dst + (size_t)step * slab + (size_t)p1 * Mp;
                for (int q = tid; q < Mq; q += nthr) {
                    const float2 t0 = T[4 * q], t1 = T[4 * q + 1], t2 = T[4 * q + 2], t3 = T[4 * q + 3];
                    *reinterpret_cast<float4*>(o0 + 4 * q) = make_float4(t0.x, t1.x, t2.x, t3.x);
                    if (has1) *reinterpret_cast<float4*>(o1 + 4 * q) = make_float4(t0.y, t1.y, t2.y, t3.y);
                }
            } else {
                // ---- c_j += G_j, linear and coalesced -------------------------------------
                const float* g0 = src + (size_t)(K - 1 - step) * slab + (size_t)p0 * Mp;
                const float* g1 = src + (size_t)(K - 1 - step) * slab + (size_t)p1 * Mp;
                for (int q = tid; q < Mq; q += nthr) {
                    const float4 a = *reinterpret_cast<const float4*>(g0 + 4 * q);
                    const float4 b = *reinterpret_cast<const float4*>(g1 + 4 * q);
                    const int i = 4 * q;
                    if (i + 0 < M) { float2 t = T[i + 0]; T[i + 0] = make_float2(t.x + a.x, t.y + b.x); }
                    if (i + 1 < M) { float2 t = T[i + 1]; T[i + 1] = make_float2(t.x + a.y, t.y + b.y); }
                    if (i + 2 < M) { float2 t = T[i + 2]; T[i + 2] = make_float2(t.x + a.z, t.y + b.z); }
                    if (i + 3 < M) { float2 t = T[i + 3]; T[i + 3] = make_float2(t.x + a.w, t.y + b.w); }
                }
                __syncthreads();
            }
        }
        if (ADJ) {
            float* o0 = dst + (size_t)p0 * Mp;
            float* o1 = dst + (size_t)p1 * Mp;
            for (int q = tid; q < Mq; q += nthr) {
                const float2 t0 = T[4 * q], t1 = T[4 * q + 1], t2 = T[4 * q + 2], t3 = T[4 * q + 3];
                *reinterpret_cast<float4*>(o0 + 4 * q) = make_float4(t0.x, t1.x, t2.x, t3.x);
                if (has1) *reinterpret_cast<float4*>(o1 + 4 * q) = make_float4(t0.y, t1.y, t2.y, t3.y);
            }
        }
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

template <int NJ, bool ADJ>
static int launch_onchip(const chebgcn_graph* g, const Ell& ell, const float* src, float* dst,
                         int nplanes, int K, int copy_t0, int nthr, hipStream_t stream) {
    const size_t lds = (size_t)(g->Mp + 4) * sizeof(float2);
    auto kern = cheb_onchip_kernel<NJ, ADJ>;
    CG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = (int)((160 * 1024) / lds);
    per_cu = per_cu < 1 ? 1 : per_cu;
    if (per_cu > 2048 / nthr) per_cu = 2048 / nthr;
    if (per_cu > 8) per_cu = 8;
    const int npairs = (nplanes + 1) / 2;
    int grid = g->num_cus * per_cu;
    if (grid > npairs) grid = npairs;
    const size_t slab = (size_t)nplanes * g->Mp;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(nthr), lds, stream, view(ell), src, dst, g->M, g->Mp,
                       nplanes, K, slab, copy_t0);
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

template <bool ADJ>
static int dispatch_onchip(const chebgcn_graph* g, const float* src, float* dst, int nplanes, int K,
                           int copy_t0, hipStream_t stream) {
    const Ell& ell = ADJ ? g->adj : g->fwd;
    const int nthr = g->M <= 2048 ? 256 : (g->M <= 4096 ? 512 : 1024);
    const int nj = (ell.ngroups * 64 + nthr - 1) / nthr;
#define CG_CASE(N) if (nj <= N) return launch_onchip<N, ADJ>(g, ell, src, dst, nplanes, K, copy_t0, nthr, stream)
    CG_CASE(1); CG_CASE(2); CG_CASE(4); CG_CASE(6); CG_CASE(8); CG_CASE(12); CG_CASE(16); CG_CASE(20);
#undef CG_CASE
    return fail(CHEBGCN_EUNSUPPORTED, "recurrence: %d row slices per thread", nj);
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
