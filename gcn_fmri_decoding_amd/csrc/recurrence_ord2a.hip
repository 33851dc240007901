// Ordered Chebyshev recurrence (recurrence_ord_kernel.h): the two-plane shapes (10239 ... 20476 active vertices), adjoint.
#include "recurrence_ord_kernel.h"

namespace chebgcn {

template <>
int launch_ordered2<true>(const chebgcn_graph* g, const Ell& ell, const float* src, float* dst, int nplanes, int K, int copy_t0,
                       hipStream_t stream) {
    return launch_ord_shape<2, kOrd2NT, kOrd2NG0, kOrd2NG1, true>(g, ell, src, dst, nplanes, K, copy_t0, stream);
}

}  // namespace chebgcn
