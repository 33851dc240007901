// Ordered Chebyshev recurrence (recurrence_ord_kernel.h): the 256-thread shapes of graphs of 1025 ... 2048 vertices.
#include "recurrence_ord_kernel.h"

namespace chebgcn {

template <bool ADJ>
int launch_ordered_small(const chebgcn_graph* g, const Ell& ell, const float* src, float* dst, int nplanes, int K, int copy_t0,
                         hipStream_t stream) {
    return launch_ord_shape<4, kOrdSNT, kOrdSNG0, kOrdSNG1, ADJ>(g, ell, src, dst, nplanes, K, copy_t0, stream);
}
template int launch_ordered_small<false>(const chebgcn_graph*, const Ell&, const float*, float*, int, int, int, hipStream_t);
// (forward only: the Clenshaw adjoint of these graphs runs the on-chip kernel of recurrence.hip -- chebgcn_recurrence_bwd)

}  // namespace chebgcn

#ifdef CG_EXPERIMENT
extern "C" int chebgcn_debug_stampsos(long long* out) {     // CG_X & 64 builds only
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(chebgcn::g_dbgo), sizeof(long long) * 16 * 64) == hipSuccess ? 0 : -1;
}
#endif
