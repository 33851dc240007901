// Fully connected layers of the head on atlas-sized inputs (models_gcn.py:650-656, 674-682):
//   y[b][o] = act( sum_i x[b][i] * W[i][o] + bias[o] )
// for products too small for the vendor library's heuristics (hipBLASLt runs 128 x 512 x 256 in 60 us on ONE workgroup;
// tools/probes/fc_small_probe.py).  One workgroup of eight waves owns a 32 x 32 tile of y; the waves split the reduction
// in chunks of eight input features (32x32x2 fp32 matrix instructions), their partial tiles are added in wave order in
// LDS -- fixed order, deterministic.
#include <hip/hip_runtime.h>
#include "../../include/chebgcn.h"
#include "status.h"
#include "contract_common.h"

namespace chebgcn {

constexpr int FC_WAVES = 8;
constexpr int FC_U = 8;        // chunks of eight input features a wave keeps in flight

__global__ void __launch_bounds__(FC_WAVES * 64)
fc_fwd_kernel(const float* __restrict__ x, long long ldx, const float* __restrict__ W, const float* __restrict__ bias,
              float* __restrict__ y, int B, int I, int O, int relu) {
    __shared__ float part[FC_WAVES][32][33];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int o0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
    const int brow = min(b0 + c, B - 1), ocol = min(o0 + c, O - 1);
    const float* xr = x + (size_t)brow * ldx;
    const float* wc = W + ocol;
    const int nchunks = (I + 7) >> 3;
    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    // chunk q of this wave covers input features 8*(wave + FC_WAVES*q) + 4*h .. + 3 for this half-wave
    for (int q0 = 0; wave + FC_WAVES * q0 < nchunks; q0 += FC_U) {
        f32x4 av[FC_U];
        float bv[FC_U][4];
#pragma unroll
        for (int u = 0; u < FC_U; ++u) {
            const int k = 8 * (wave + FC_WAVES * (q0 + u)) + 4 * h;
            const int kc = k < I ? k : 0;               // I is a multiple of four: a group of four is inside or outside
            av[u] = *reinterpret_cast<const f32x4*>(xr + kc);
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[u][j] = wc[(size_t)(kc + j) * O];
        }
#pragma unroll
        for (int u = 0; u < FC_U; ++u) {
            const int k = 8 * (wave + FC_WAVES * (q0 + u)) + 4 * h;
            const bool ok = k < I;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ok ? av[u][j] : 0.f, ok ? bv[u][j] : 0.f, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) part[wave][acc_row(j, h)][c] = acc[j];
    __syncthreads();
    for (int e = threadIdx.x; e < 32 * 32; e += FC_WAVES * 64) {
        const int r = e >> 5, cc = e & 31;
        float s = part[0][r][cc];
#pragma unroll
        for (int w = 1; w < FC_WAVES; ++w) s += part[w][r][cc];
        if (b0 + r < B && o0 + cc < O) {
            s += bias ? bias[o0 + cc] : 0.f;
            y[(size_t)(b0 + r) * O + o0 + cc] = relu ? fmaxf(s, 0.f) : s;
        }
    }
}

}  // namespace chebgcn

using namespace chebgcn;

extern "C" int chebgcn_fc_fwd_supported(int B, int I, int O) {
    return B > 0 && I > 0 && O > 0 && (I & 3) == 0 && I <= 4096 && (long long)B * O <= (1 << 20);
}

extern "C" int chebgcn_fc_fwd(const float* x, int64_t ldx, const float* W, const float* bias, float* y, int B, int I,
                              int O, int relu, chebgcn_stream stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CG_REQUIRE(x && W && y && B > 0 && I > 0 && O > 0 && ldx >= I, "fc_fwd: bad argument");
    if (!chebgcn_fc_fwd_supported(B, I, O) || (ldx & 3) || ((uintptr_t)x & 15))
        return CHEBGCN_EUNSUPPORTED;
    dim3 grid((O + 31) / 32, (B + 31) / 32);
    hipLaunchKernelGGL(fc_fwd_kernel, grid, dim3(FC_WAVES * 64), 0, stream, x, (long long)ldx, W, bias, y, B, I, O, relu);
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}
