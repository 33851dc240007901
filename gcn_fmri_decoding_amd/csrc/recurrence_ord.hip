// Ordered Chebyshev recurrence (recurrence_ord_kernel.h): the four-plane shapes, the shape table and the dispatcher.
#include "recurrence_ord_kernel.h"

namespace chebgcn {

// Shape of the ordered kernel for a graph of Mq vertex quads of which the first SQ have rows: NQ = quads per thread (512
// threads), NG = quad levels with rows, planes = 4 while 16 bytes per active vertex fit the LDS (+ zero and trash slot), else 2.
// Served: 2049 ... 20476 active vertices (below, the generic on-chip kernel and the fused atlas layer work in the caller's
// order and gain nothing from a relabelling), at most one quad level of isolated / padding vertices behind the rows.
bool ordered_shape(int Mq, int SQ, int* NT, int* NQ, int* NG, int* planes) {
    if (SQ <= 512) return false;
    auto shape = [&](int nt, int ng0, int ng1, int cap) {
        const int nq = (Mq + nt - 1) / nt, ng = (SQ + nt - 1) / nt;
        if ((nq != ng && nq != ng + 1) || ng < ng0 || ng > ng1 || 4 * SQ + 2 > cap) return false;
        *NT = nt;
        *NQ = nq;
        *NG = ng;
        return true;
    };
    if (shape(kOrd4NT, kOrd4NG0, kOrd4NG1, ord_entries<4, 1, kOrd4NT>::cap)) *planes = 4;
    // (two planes: up to 10752 vertices the 768-thread kernel of recurrence.hip on the caller's order is faster -- N = 10242, batch
    // 256: 0.596 / 0.630 ms against 0.678 / 0.699 ms; beyond, the ordered kernel wins by 1.2x (N = 13000) to 2.9x (N = 19000))
    else if (4 * Mq > 10752 && shape(kOrd2NT, kOrd2NG0, kOrd2NG1, ord_entries<2, 1, kOrd2NT>::cap)) *planes = 2;
    else return false;
    return true;
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
    if (ell.planes == 4) return launch_ord_shape<4, kOrd4NT, kOrd4NG0, kOrd4NG1, ADJ>(g, ell, src, dst, nplanes, K, copy_t0, stream);
    return launch_ordered2<ADJ>(g, ell, src, dst, nplanes, K, copy_t0, stream);
}

template int dispatch_ordered<false>(const chebgcn_graph*, const Ell&, const float*, float*, int, int, int, hipStream_t);
template int dispatch_ordered<true>(const chebgcn_graph*, const Ell&, const float*, float*, int, int, int, hipStream_t);

}  // namespace chebgcn

#ifdef CG_EXPERIMENT
extern "C" int chebgcn_debug_stampso(long long* out) {      // CG_X & 64 builds only (tools/kbench.py --stamps)
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(chebgcn::g_dbgo), sizeof(long long) * 16 * 64) == hipSuccess ? 0 : -1;
}
#endif
