// Ordered Chebyshev recurrence (recurrence_ord_kernel.h): the two-plane shapes (10239 ... 20476 active vertices), forward.
#include "recurrence_ord_kernel.h"

namespace chebgcn {

template <>
int launch_ordered2<false>(const chebgcn_graph* g, const Ell& ell, const float* src, float* dst, int nplanes, int K, int copy_t0,
                       hipStream_t stream) {
    return launch_ord_shape<2, kOrd2NT, kOrd2NG0, kOrd2NG1, false>(g, ell, src, dst, nplanes, K, copy_t0, stream);
}

}  // namespace chebgcn

#ifdef CG_EXPERIMENT
extern "C" int chebgcn_debug_stampso2(long long* out) {     // CG_X & 64 builds only (forward kernels of this file)
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(chebgcn::g_dbgo), sizeof(long long) * 16 * 64) == hipSuccess ? 0 : -1;
}
#endif
