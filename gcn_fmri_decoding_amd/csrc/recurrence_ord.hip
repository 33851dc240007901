// Ordered Chebyshev recurrence (recurrence_ord_kernel.h): the four-plane shapes, the shape table and the dispatcher.
#include <algorithm>

#include "recurrence_ord_kernel.h"

namespace chebgcn {

// Shape of the ordered kernel for a graph of Mq vertex quads of which the first SQ have rows: NT = threads, NQ = quads per thread,
// NG = quad levels with rows, planes = 4 while 16 bytes per active vertex fit the LDS (+ zero and trash slot), else 2.
// Served: 2049 ... 20476 active vertices on 512 threads; round 6: planes of 1025 ... 2048 vertices (at most 2048 active) on 256
// threads -- there the on-chip kernel of recurrence.hip needs two linear pieces per thread and plane and drops to 0.23 of the
// roofline (N = 1000, M = 1044, K = 10, batch 128: forward 0.094 -> 0.065 ms, N = 2000: 0.156 -> 0.112 ms; below, in the caller's
// order, it is the faster one: N = 800 0.055 against 0.059 ms, N = 500 0.031 against 0.046 -- EXPERIMENTS 8.8).  Isolated / padding
// vertices beyond one quad level behind the rows: the tail kernel.  CHEBGCN_ORD_SMALL=0: the 512-thread shapes only.
bool ordered_shape(int Mq, int SQ, int* NT, int* NQ, int* NG, int* planes) {
    static const int small_on = [] { const char* e = getenv("CHEBGCN_ORD_SMALL"); return e ? atoi(e) : 1; }();
    if (SQ <= 512 && (Mq <= 256 || !small_on)) return false;
    auto shape = [&](int nt, int ng0, int ng1, int cap) {
        const int ng = (SQ + nt - 1) / nt;
        // quad levels per thread: the levels with rows + at most ONE level of isolated / padding vertices, whose state sits in
        // registers like a row's; what lies behind (the fake vertices of a graph coarsened many times: 2672 of 12672 at six
        // levels) holds no rows and nobody gathers from it -- the tail kernel streams it (T_k = c(k) x there)
        const int nq = std::min((Mq + nt - 1) / nt, ng + 1);
        if (nq < ng || ng < ng0 || ng > ng1 || 4 * SQ + 2 > cap) return false;
        *NT = nt;
        *NQ = nq;
        *NG = ng;
        return true;
    };
    if (SQ <= 512) {
        if (!shape(kOrdSNT, kOrdSNG0, kOrdSNG1, ord_entries<4, 1, kOrdSNT>::cap)) return false;
        *planes = 4;
        return true;
    }
    if (shape(kOrd4NT, kOrd4NG0, kOrd4NG1, ord_entries<4, 1, kOrd4NT>::cap)) *planes = 4;
    // (two planes: up to 10752 vertices the 768-thread kernel of recurrence.hip on the caller's order is faster -- N = 10242, batch
    // 256: 0.596 / 0.630 ms against 0.678 / 0.699 ms; beyond, the ordered kernel wins by 1.2x (N = 13000) to 2.9x (N = 19000))
    else if (4 * Mq > 10752 && shape(kOrd2NT, kOrd2NG0, kOrd2NG1, ord_entries<2, 1, kOrd2NT>::cap)) *planes = 2;
    else return false;
    return true;
}

// Vertices [tail0, Mp) of every plane: isolated (no row, no column entry) or padding.  Forward: T_0 = x, T_k = c(k) x with
// c = 1, 0, -1, 0, ... (T_1 = L x = 0, T_k = -T_{k-2}); adjoint: dx = sum_m c(m) G_m.  One thread = four vertices of one plane.
template <bool ADJ>
__global__ void __launch_bounds__(256)
cheb_ord_tail_kernel(const float* __restrict__ src, float* __restrict__ dst, int Mp, int tail0, int nplanes, int K, size_t slab,
                     int copy_t0) {
    const int tq = (Mp - tail0) >> 2;
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    const int plane = (int)(id / tq), q = (int)(id - (long long)plane * tq);
    if (plane >= nplanes) return;
    const size_t off = (size_t)plane * Mp + tail0 + 4 * q;
    if (!ADJ) {
        const float4 x = *reinterpret_cast<const float4*>(src + off);
        if (copy_t0) *reinterpret_cast<float4*>(dst + off) = x;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f), nx = make_float4(-x.x, -x.y, -x.z, -x.w);
        for (int k = 1; k < K; ++k) *reinterpret_cast<float4*>(dst + (size_t)k * slab + off) = (k & 1) ? z : ((k & 2) ? nx : x);
    } else {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int m = 0; m < K; m += 2) {
            const float4 gm = *reinterpret_cast<const float4*>(src + (size_t)m * slab + off);
            const float c = (m & 2) ? -1.f : 1.f;
            acc.x = fmaf(c, gm.x, acc.x); acc.y = fmaf(c, gm.y, acc.y); acc.z = fmaf(c, gm.z, acc.z); acc.w = fmaf(c, gm.w, acc.w);
        }
        *reinterpret_cast<float4*>(dst + off) = acc;
    }
}

// planes one launch may hold: a slab is addressed through one buffer descriptor (32-bit offsets)
bool ordered_fits(const chebgcn_graph* g, int nplanes) {
    return (size_t)nplanes * g->Mp * sizeof(float) <= 0xFFFF0000ull;
}

template <bool ADJ>
int dispatch_ordered(const chebgcn_graph* g, const Ell& ell, const float* src, float* dst, int nplanes, int K, int copy_t0,
                     hipStream_t stream) {
    if (!ordered_fits(g, nplanes))
        return fail(CHEBGCN_EUNSUPPORTED, "recurrence: %d planes of %d vertices exceed 4 GB per slab", nplanes, g->Mp);
    if constexpr (ADJ) {
        if (ell.ord_NT <= kOrdSNT) return fail(CHEBGCN_EUNSUPPORTED, "recurrence: the 256-thread ordered shapes are forward kernels");
    }
    const int rc = (!ADJ && ell.ord_NT <= kOrdSNT) ? launch_ordered_small<false>(g, ell, src, dst, nplanes, K, copy_t0, stream)
                   : ell.planes == 4 ? launch_ord_shape<4, kOrd4NT, kOrd4NG0, kOrd4NG1, ADJ>(g, ell, src, dst, nplanes, K, copy_t0, stream)
                                     : launch_ordered2<ADJ>(g, ell, src, dst, nplanes, K, copy_t0, stream);
    if (rc != CHEBGCN_OK || !ell.ord_tail || ell.ord_tail >= g->Mp) return rc;
    // the vertices behind the kernel's quad levels: no rows, no column entries -- T_k = c(k) x, dx = sum_m c(m) G_m
    const int tq = (g->Mp - ell.ord_tail) / 4;
    const long long n = (long long)nplanes * tq;
    note_dispatch_more(ADJ ? "cheb_ord_tail_kernel<true>" : "cheb_ord_tail_kernel<false>");
    hipLaunchKernelGGL(cheb_ord_tail_kernel<ADJ>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, dst, g->Mp, ell.ord_tail,
                       nplanes, K, (size_t)nplanes * g->Mp, copy_t0);
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

template int dispatch_ordered<false>(const chebgcn_graph*, const Ell&, const float*, float*, int, int, int, hipStream_t);
template int dispatch_ordered<true>(const chebgcn_graph*, const Ell&, const float*, float*, int, int, int, hipStream_t);

}  // namespace chebgcn

#ifdef CG_EXPERIMENT
extern "C" int chebgcn_debug_stampso(long long* out) {      // CG_X & 64 builds only (tools/kbench.py --stamps)
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(chebgcn::g_dbgo), sizeof(long long) * 16 * 64) == hipSuccess ? 0 : -1;
}
#endif
