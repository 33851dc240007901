// ReluGrad + bias gradient of a pool == 1 layer from (dout, ReLU bit mask): the body of bias_grad_relu_kernel (pointwise.hip), shared
// with the launch that adds the per-workgroup partials of a small weight gradient (contract.hip reduce_partials_small_bias_kernel).
#pragma once
#include "common.h"

#ifndef CG_DY_NT
#define CG_DY_NT 0      // see contract.hip
#endif

namespace chebgcn {

// workgroups along the vertex axis and batch subsets per workgroup of bias_grad_relu_kernel: 64 quads x 4 batch subsets, or 16 x 16
// where that leaves the chip short of work
static inline int bias_grad_blocks(int M, int F, int* parts_out) {
    const int Mp = plane_stride(M);
    const bool fine = ((Mp / 4 + 63) / 64) * F < 512;
    if (parts_out) *parts_out = fine ? 16 : 4;
    return fine ? (Mp / 4 + 15) / 16 : (Mp / 4 + 63) / 64;
}

template <int BIAS, int NP, bool DY16, bool MASKED>
__device__ __forceinline__ void bias_grad_relu_body(const float* __restrict__ dout, const uint8_t* __restrict__ mask, float* __restrict__ dy,
                                                    float* __restrict__ dbias, float* __restrict__ fpart, int B, int M, int Mp, int F,
                                                    size_t d_bstride, size_t d_fstride,      // element strides of dout: F*Mp and Mp, or Mp and 0 (one plane per window)
                                                    int bx, int f, int nbx) {                // the job: block bx of nbx along the vertices, filter f
    __shared__ float4 psum[256];
    constexpr int QL = 256 / NP;                        // quads per workgroup
    const int ql = threadIdx.x % QL, part = threadIdx.x / QL;
    const int Mq = Mp >> 2;
    const int q = bx * QL + ql;
    const bool live = q < Mq;
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        const float* gp = dout + (size_t)f * d_fstride + 4 * q;
        const uint8_t* mp = MASKED ? mask + (size_t)f * Mq + q : nullptr;
#pragma unroll 4
        for (int b = part; b < B; b += NP) {
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            const f32x4 g = CG_DY_NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(gp + (size_t)b * d_bstride))
                                     : *reinterpret_cast<const f32x4*>(gp + (size_t)b * d_bstride);
            const int bits = MASKED ? mp[(size_t)b * F * Mq] : 15;
            const float4 d = make_float4((bits & 1) ? g.x : 0.f, (bits & 2) ? g.y : 0.f, (bits & 4) ? g.z : 0.f,
                                         (bits & 8) ? g.w : 0.f);
            if (DY16) {
                typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                const bf16x4 h = {(__bf16)d.x, (__bf16)d.y, (__bf16)d.z, (__bf16)d.w};
                *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(dy) + ((size_t)b * F + f) * Mp + 4 * q) = h;
            } else if (MASKED && dy) {
                *reinterpret_cast<float4*>(dy + ((size_t)b * F + f) * Mp + 4 * q) = d;
            }
            sum.x += d.x; sum.y += d.y; sum.z += d.z; sum.w += d.w;
        }
        const int m = 4 * q;                            // the padding of the plane takes no gradient
        sum.x = m + 0 < M ? sum.x : 0.f;
        sum.y = m + 1 < M ? sum.y : 0.f;
        sum.z = m + 2 < M ? sum.z : 0.f;
        sum.w = m + 3 < M ? sum.w : 0.f;
    }
    if (BIAS == CHEBGCN_BIAS_NONE) return;
    psum[threadIdx.x] = sum;
    __syncthreads();
    if (part == 0) {
        float4 t = psum[ql];
#pragma unroll
        for (int p = 1; p < NP; ++p) {
            const float4 o = psum[p * QL + ql];
            t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
        }
        if (BIAS == CHEBGCN_BIAS_VERTEX) {
            if (live) *reinterpret_cast<float4*>(dbias + (size_t)f * Mp + 4 * q) = t;
        } else {
            float s = (t.x + t.y) + (t.z + t.w);
            for (int d = QL / 2; d > 0; d >>= 1) s += __shfl_xor(s, d);      // the QL lanes of part 0 (QL <= 64: one wave)
            if (ql == 0) fpart[(size_t)f * nbx + bx] = s;
        }
    }
}


}  // namespace chebgcn
