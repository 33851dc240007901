// Shared by the contraction kernels (contract.hip: f32 MFMA, contract_bf16.hip: bf16 MFMA).
#pragma once
#include "common.h"

namespace chebgcn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// streaming (non-temporal) 16-byte load: the stack / gradient planes are read once
__device__ __forceinline__ float4 ld_stream(const float* p) {
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

// row of accumulator register j for a lane in half h (32x32 C/D layout, every input dtype)
__device__ __forceinline__ int acc_row(int j, int h) { return (j & 3) + 8 * (j >> 2) + 4 * h; }

struct FwdArgs {
    const float* stack; const float* W; const float* bias; float* out; uint8_t* argmax;
    int B, M, Mp, Fin, K, Fout, FinK;
    int pool, pool_kind, relu, bias_kind;
    int Mo, Mpo;
    size_t slab;                 // B*Fin*Mp
    // out_K > 0: the output rows are the rows fin*out_K + k of a Chebyshev (gradient) stack and go to
    // out[k][b][fin][m] instead of out[b][row][m] (chebgcn_contract_bwd_x_bf16: the kernel computes W^T dy)
    int out_K = 0;
    // pool == 1 with ReLU: one byte per four vertices, bit r = (result of vertex 4i+r > 0), planes
    // [B][Fout][Mp/4] -- all the gradient kernels need of the forward result (the ReluGrad of the
    // reference's autodiff folded into chebgcn_contract_bwd_*_relu)
    uint8_t* relu_mask = nullptr;
    // mean over the filters of the (bias + ReLU) result, [B][Mp] (tf.reduce_mean(x, -1), models_gcn.py:673, fused into the
    // last layer's epilogue: contract_fwd_ring_kernel only); `out` may then be NULL (the layer output itself is not stored)
    float* mean_out = nullptr;
    // gate (contract_fwd_ring_kernel<true, true> only: chebgcn_contract_fwd_gated): a ReLU mask in the layout of `relu_mask`,
    // [B][Fout][Mp/4] -- the result is stored as bit ? value : 0.  The kernel is then the last step of the gradient wrt a
    // layer's input in forward form, and the mask is the one the layer BELOW left: what is stored is that layer's dy
    const uint8_t* gate = nullptr;
};

// Epilogue of one filter row for the four vertices n0..n0+3 held by lane c of a half-wave:
// bias (models_gcn.py:619-629), ReLU, graph pooling over p consecutive vertices (:631-648),
// store; `argmax` receives the arg-max byte (max pooling) or the ReLU mask (average pooling).
// `bbv`: the per-vertex bias of this row when the caller fetched it ahead (have_bb), so that the
// rows of a tile do not pay one memory round trip each.
// `plane`: index of the output plane when the caller tracks it (out_K scatter without the divisions), else -1.
__device__ __forceinline__ void fwd_epilogue_row(const FwdArgs& a, int b, int fo, float (&v)[4], int n0, bool valid,
                                                 int c, bool have_bb = false, float4 bbv = make_float4(0.f, 0.f, 0.f, 0.f),
                                                 long long plane = -1) {
    const bool fo_ok = fo < a.Fout;
    const int p = a.pool;
    const int lanes_per_win = p > 4 ? (p >> 2) : 1;     // lanes sharing one pooling window
    if (a.bias_kind == CHEBGCN_BIAS_FILTER) {
        const float bb = fo_ok ? a.bias[fo] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bb;
    } else if (a.bias_kind == CHEBGCN_BIAS_VERTEX) {
        float4 bb = bbv;
        if (!have_bb) {
            bb = make_float4(0.f, 0.f, 0.f, 0.f);
            if (fo_ok && valid) bb = *reinterpret_cast<const float4*>(a.bias + (size_t)fo * a.Mp + n0);
        }
        v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
    }
    if (a.relu) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
    }
    const int foc = fo_ok ? fo : 0;
    float* orow = a.out + (plane >= 0 ? (size_t)plane
                           : a.out_K > 0 ? ((size_t)(foc % a.out_K) * a.B * (a.Fout / a.out_K) + (size_t)b * (a.Fout / a.out_K) + foc / a.out_K)
                                         : ((size_t)b * a.Fout + foc)) * a.Mpo;
    uint8_t* arow = a.argmax ? a.argmax + ((size_t)b * a.Fout + (fo_ok ? fo : 0)) * a.Mpo : nullptr;
    if (p == 1) {
        if (fo_ok && valid) {
            if (a.out) *reinterpret_cast<float4*>(orow + n0) = make_float4(v[0], v[1], v[2], v[3]);
            if (a.relu_mask)
                a.relu_mask[((size_t)b * a.Fout + fo) * (a.Mpo >> 2) + (n0 >> 2)] =
                    (uint8_t)((v[0] > 0.f ? 1 : 0) | (v[1] > 0.f ? 2 : 0) | (v[2] > 0.f ? 4 : 0) | (v[3] > 0.f ? 8 : 0));
        }
    } else if (a.pool_kind == CHEBGCN_POOL_MAX) {
        if (p == 2) {
            const int no = n0 >> 1;
            if (fo_ok && no < a.Mpo) {
                const bool s0 = v[1] > v[0], s1 = v[3] > v[2];
                *reinterpret_cast<float2*>(orow + no) = make_float2(s0 ? v[1] : v[0], s1 ? v[3] : v[2]);
                if (arow) *reinterpret_cast<uchar2*>(arow + no) = make_uchar2(s0 ? 1 : 0, s1 ? 1 : 0);
            }
        } else {
            float m = v[0];
            int idx = 0;
#pragma unroll
            for (int r = 1; r < 4; ++r)
                if (v[r] > m) { m = v[r]; idx = r; }
            idx += 4 * (c & (lanes_per_win - 1));
            for (int d = 1; d < lanes_per_win; d <<= 1) {
                const float om = __shfl_xor(m, d);
                const int oi = __shfl_xor(idx, d);
                if (om > m || (om == m && oi < idx)) { m = om; idx = oi; }
            }
            const int no = n0 / p;
            if (fo_ok && (c & (lanes_per_win - 1)) == 0 && no < a.Mpo) {
                orow[no] = m;
                if (arow) arow[no] = (uint8_t)idx;
            }
        }
    } else {
        // average pooling; the argmax buffer receives the ReLU mask of the window
        // (bit i set <=> element i of the window is > 0), which is all the
        // gradient needs (pool <= 8)
        const int mask = (v[0] > 0.f ? 1 : 0) | (v[1] > 0.f ? 2 : 0) | (v[2] > 0.f ? 4 : 0) | (v[3] > 0.f ? 8 : 0);
        if (p == 2) {
            const int no = n0 >> 1;
            if (fo_ok && no < a.Mpo) {
                *reinterpret_cast<float2*>(orow + no) = make_float2(0.5f * (v[0] + v[1]), 0.5f * (v[2] + v[3]));
                if (arow) *reinterpret_cast<uchar2*>(arow + no) = make_uchar2(mask & 3, mask >> 2);
            }
        } else {
            float s = (v[0] + v[1]) + (v[2] + v[3]);
            int mk = mask << (4 * (c & (lanes_per_win - 1) & 1));
            for (int d = 1; d < lanes_per_win; d <<= 1) {
                s += __shfl_xor(s, d);
                mk |= __shfl_xor(mk, d);
            }
            const int no = n0 / p;
            if (fo_ok && (c & (lanes_per_win - 1)) == 0 && no < a.Mpo) {
                orow[no] = s / (float)p;
                if (arow) arow[no] = (uint8_t)mk;
            }
        }
    }
}

}  // namespace chebgcn
