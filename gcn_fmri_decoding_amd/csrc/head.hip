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
constexpr int FC_CHUNK = 32;   // input features per chunk of the forward's reduction
constexpr int FC_U = 2;        // chunks a wave keeps in flight

// gridDim.z > 1: the reduction is also split across workgroups (chunks [z*cps, (z+1)*cps) of FC_CHUNK input features); the
// partial tiles go to part_out[z][b][o] and fc_fwd_reduce_kernel adds them in order.
__global__ void __launch_bounds__(FC_WAVES * 64)
fc_fwd_kernel(const float* __restrict__ x, long long ldx, const float* __restrict__ W, const float* __restrict__ bias,
              float* __restrict__ y, float* __restrict__ part_out, int B, int I, int O, int relu, int cps) {
    __shared__ float part[FC_WAVES][32][33];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    const int o0 = bx * 32, b0 = by * 32;
    const int brow = min(b0 + c, B - 1), ocol = min(o0 + c, O - 1);
    const float* xr = x + (size_t)brow * ldx;
    const float* wc = W + ocol;
    const int nchunks = (I + FC_CHUNK - 1) / FC_CHUNK;
    const int q_lo = bz * cps, q_hi = min(q_lo + cps, nchunks);
    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    // chunk q covers input features 32*q .. 32*q + 31: this half-wave takes 16 of them, i.e. every lane reads one whole
    // 64-byte sector of its OWN row of x in four 16-byte requests (with 16 bytes per lane and sector the L1 fills were four
    // times the bytes used: 64 x 10466 x 512 23.4 -> 19.4 us); this wave takes chunks q_lo + wave, + FC_WAVES, ...
    for (int q0 = q_lo + wave; q0 < q_hi; q0 += FC_WAVES * FC_U) {
        f32x4 av[FC_U][4];
        float bv[FC_U][4][4];
#pragma unroll
        for (int u = 0; u < FC_U; ++u) {
            const int q = q0 + FC_WAVES * u;
            const int k = (q < q_hi ? FC_CHUNK * q : 0) + 16 * h;    // beyond the range: any readable address, multiplied by zero
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                av[u][t] = *reinterpret_cast<const f32x4*>(xr + (k + 4 * t + 3 < ldx ? k + 4 * t : 0));
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[u][t][j] = wc[(size_t)(k + 4 * t + j < I ? k + 4 * t + j : 0) * O];
            }
        }
#pragma unroll
        for (int u = 0; u < FC_U; ++u) {
            const int q = q0 + FC_WAVES * u;
            const int k = FC_CHUNK * q + 16 * h;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool ok = q < q_hi && k + 4 * t + j < I;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ok ? av[u][t][j] : 0.f, ok ? bv[u][t][j] : 0.f, acc, 0, 0, 0);
                }
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
            if (part_out) {
                part_out[((size_t)bz * B + b0 + r) * O + o0 + cc] = s;
            } else {
                s += bias ? bias[o0 + cc] : 0.f;
                y[(size_t)(b0 + r) * O + o0 + cc] = relu ? fmaxf(s, 0.f) : s;
            }
        }
    }
}

__global__ void __launch_bounds__(256)
fc_fwd_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias, float* __restrict__ y, int BO, int O,
                     int S, int relu) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= BO) return;
    float v[16];
    float s = 0.f;
    for (int z0 = 0; z0 < S; z0 += 16) {                 // sixteen loads in flight, added in split order
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = part[(size_t)min(z0 + u, S - 1) * BO + e];
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (z0 + u < S) s += v[u];
    }
    s += bias ? bias[e % O] : 0.f;
    y[e] = relu ? fmaxf(s, 0.f) : s;
}

// dW[i][o] = sum_b x[b][i] * gm[b][o],  db[o] = sum_b gm[b][o],  gm = g gated by y > 0 (ReluGrad) where y is given.
// One workgroup per 32 x 32 tile of dW; the eight waves split the batch in chunks of eight windows and are summed in wave
// order.  The workgroups of the first row tile also reduce db (per lane over its windows, then half-waves and waves in order).
__global__ void __launch_bounds__(FC_WAVES * 64)
fc_bwd_w_kernel(const float* __restrict__ x, long long ldx, const float* __restrict__ g, const float* __restrict__ y,
                float* __restrict__ dW, float* __restrict__ db, int B, int I, int O) {
    __shared__ float part[FC_WAVES][32][33];
    __shared__ float bpart[FC_WAVES][2][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int o0 = blockIdx.x * 32, i0 = blockIdx.y * 32;
    const int icol = min(i0 + c, I - 1), ocol = min(o0 + c, O - 1);
    const int nchunks = (B + 7) >> 3;
    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    float bsum = 0.f;
    constexpr int U = 2;
    for (int q0 = 0; wave + FC_WAVES * q0 < nchunks; q0 += U) {
        float av[U][4], gv[U][4], yv[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int b = 8 * (wave + FC_WAVES * (q0 + u)) + 4 * h + j;
                const int bc = b < B ? b : 0;
                av[u][j] = x[(size_t)bc * ldx + icol];
                gv[u][j] = g[(size_t)bc * O + ocol];
                yv[u][j] = y ? y[(size_t)bc * O + ocol] : 1.f;
            }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int b = 8 * (wave + FC_WAVES * (q0 + u)) + 4 * h + j;
                const bool ok = b < B;
                const float gm = (ok && yv[u][j] > 0.f) ? gv[u][j] : 0.f;
                bsum += gm;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ok ? av[u][j] : 0.f, gm, acc, 0, 0, 0);
            }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) part[wave][acc_row(j, h)][c] = acc[j];
    bpart[wave][h][c] = bsum;
    __syncthreads();
    for (int e = threadIdx.x; e < 32 * 32; e += FC_WAVES * 64) {
        const int r = e >> 5, cc = e & 31;
        float s = part[0][r][cc];
#pragma unroll
        for (int w = 1; w < FC_WAVES; ++w) s += part[w][r][cc];
        if (i0 + r < I && o0 + cc < O) dW[(size_t)(i0 + r) * O + o0 + cc] = s;
    }
    if (db && blockIdx.y == 0 && threadIdx.x < 32 && o0 + threadIdx.x < O) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < FC_WAVES; ++w) s += bpart[w][0][threadIdx.x] + bpart[w][1][threadIdx.x];
        db[o0 + threadIdx.x] = s;
    }
}

// dx[b][i] = sum_o gm[b][o] * W[i][o]: 32 x 32 tiles of dx, the eight waves split the outputs in chunks of eight.
template <bool VEC>
__global__ void __launch_bounds__(FC_WAVES * 64)
fc_bwd_x_kernel(const float* __restrict__ g, const float* __restrict__ y, const float* __restrict__ W,
                float* __restrict__ dx, long long lddx, int B, int I, int O) {
    __shared__ float part[FC_WAVES][32][33];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int i0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
    const int brow = min(b0 + c, B - 1), irow = min(i0 + c, I - 1);
    const float* gr = g + (size_t)brow * O;
    const float* yr = y ? y + (size_t)brow * O : nullptr;
    const float* wr = W + (size_t)irow * O;
    const int nchunks = (O + 7) >> 3;
    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    constexpr int U = 4;
    for (int q0 = 0; wave + FC_WAVES * q0 < nchunks; q0 += U) {
        f32x4 gv[U], yv[U], wv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = 8 * (wave + FC_WAVES * (q0 + u)) + 4 * h;
            if (VEC) {
                const int kc = k < O ? k : 0;
                gv[u] = *reinterpret_cast<const f32x4*>(gr + kc);
                wv[u] = *reinterpret_cast<const f32x4*>(wr + kc);
                if (yr) yv[u] = *reinterpret_cast<const f32x4*>(yr + kc);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int kc = k + j < O ? k + j : 0;
                    gv[u][j] = gr[kc];
                    wv[u][j] = wr[kc];
                    if (yr) yv[u][j] = yr[kc];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = 8 * (wave + FC_WAVES * (q0 + u)) + 4 * h;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = k + j < O;
                const float gm = (ok && (!yr || yv[u][j] > 0.f)) ? gv[u][j] : 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(gm, ok ? wv[u][j] : 0.f, acc, 0, 0, 0);
            }
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
        if (b0 + r < B && i0 + cc < I) dx[(size_t)(b0 + r) * lddx + i0 + cc] = s;
    }
}

}  // namespace chebgcn

namespace chebgcn {
// Softmax cross-entropy of the logits and its gradient in one launch (tf.nn.sparse_softmax_cross_entropy_with_logits +
// tf.reduce_mean, models_gcn.py:257-259, and what TensorFlow's autodiff derives for them):
//   loss = mean_b( logsumexp(z_b) - z_b[y_b] ),   dz[b][c] = (softmax(z_b)[c] - [c == y_b]) / B.
// One workgroup: thread t takes the rows t, t + 256, ... in order, the 256 partial sums are added in a fixed tree --
// deterministic.  (torch's cross_entropy + backward are six launches; at the reference's shapes the head of the step is
// launch-bound.)
constexpr int XENT_REG = 32;      // classes held in registers (the reference's tasks: 21-23 cognitive states)
template <typename LabelT>
__global__ void __launch_bounds__(256)
softmax_xent_kernel(const float* __restrict__ z, const LabelT* __restrict__ y, float* __restrict__ loss,
                    float* __restrict__ dz, int B, int C) {
    __shared__ float part[256];
    float acc = 0.f;
    const float invB = 1.f / (float)B;
    for (int b = threadIdx.x; b < B; b += 256) {
        const float* row = z + (size_t)b * C;
        long long t = (long long)y[b];
        // a label outside [0, C) is the caller's error.  TensorFlow's GPU kernel answers with NaN for that row's loss and
        // gradient (its CPU kernel raises); so does this one -- the loss of the step is NaN and the error surfaces -- and the
        // row is never read past its end
        const bool bad = t < 0 || t >= C;
        t = t < 0 ? 0 : t >= C ? C - 1 : t;
        float* drow = dz + (size_t)b * C;
        if (bad) {
            acc += __builtin_nanf("");
            for (int c = 0; c < C; ++c) drow[c] = __builtin_nanf("");
            continue;
        }
        if (C <= XENT_REG) {
            // the row in registers: ONE memory round trip (a loop over row[c] is a dependent load per class and pass: 12 us for
            // 128 x 22 logits)
            float v[XENT_REG];
#pragma unroll
            for (int c = 0; c < XENT_REG; ++c) v[c] = c < C ? row[c] : -__builtin_inff();
            float m = v[0];
#pragma unroll
            for (int c = 1; c < XENT_REG; ++c) m = fmaxf(m, v[c]);
            float s = 0.f, so = 0.f, et = 0.f;
#pragma unroll
            for (int c = 0; c < XENT_REG; ++c) {
                v[c] = c < C ? expf(v[c] - m) : 0.f;
                s += v[c];
                so += c == t ? 0.f : v[c];
                et = c == t ? v[c] : et;
            }
            acc += et > 1e-30f ? log1pf(so / et) : logf(s) - logf(fmaxf(et, 1e-45f));
            const float inv = invB / s;
#pragma unroll
            for (int c = 0; c < XENT_REG; ++c)
                if (c < C) drow[c] = c == t ? -so * inv : v[c] * inv;
            continue;
        }
        float m = row[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, row[c]);
        float s = 0.f, so = 0.f;                              // all classes; all but the labelled one
        for (int c = 0; c < C; ++c) {
            const float e = expf(row[c] - m);
            s += e;
            so += c == t ? 0.f : e;
        }
        // -log softmax at the labelled class = log(1 + so / e_t): no cancellation when the prediction is confident and right
        const float et = expf(row[t] - m);
        acc += et > 1e-30f ? log1pf(so / et) : logf(s) - (row[t] - m);
        const float inv = invB / s;
        // softmax - 1 at the labelled class is -(sum of the others) / s: no cancellation when the prediction is confident
        for (int c = 0; c < C; ++c) drow[c] = c == t ? -so * inv : expf(row[c] - m) * inv;
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) part[threadIdx.x] += part[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = part[0] * invB;
}
}  // namespace chebgcn

using namespace chebgcn;

// splits of the reduction across workgroups: enough workgroups for two per CU, at least 64 chunks (512 input features) each
static int fc_splits(int B, int I, int O) {
    const int tiles = ((O + 31) / 32) * ((B + 31) / 32);
    int s = 512 / tiles;
    const int by_len = (I + 511) / 512;
    if (s > by_len) s = by_len;
    return s < 1 ? 1 : s;
}

extern "C" int chebgcn_fc_fwd_supported(int B, int I, int O) {
    return B > 0 && I > 0 && O > 0 && I <= (1 << 20) && (long long)B * O <= (1 << 20);
}

extern "C" size_t chebgcn_fc_fwd_workspace(int B, int I, int O) {
    if (!chebgcn_fc_fwd_supported(B, I, O)) return 0;
    const int S = fc_splits(B, I, O);
    return S > 1 ? (size_t)S * B * O * sizeof(float) : 0;
}

extern "C" int chebgcn_fc_fwd(const float* x, int64_t ldx, const float* W, const float* bias, float* y, void* workspace,
                              size_t workspace_bytes, int B, int I, int O, int relu, chebgcn_stream stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CG_REQUIRE(x && W && y && B > 0 && I > 0 && O > 0 && ldx >= I, "fc_fwd: bad argument");
    if (!chebgcn_fc_fwd_supported(B, I, O) || (ldx & 3) || ((uintptr_t)x & 15))
        return CHEBGCN_EUNSUPPORTED;
    const int S = fc_splits(B, I, O);
    const int nchunks = (I + FC_CHUNK - 1) / FC_CHUNK, cps = (nchunks + S - 1) / S;
    CG_REQUIRE(S == 1 || (workspace && workspace_bytes >= chebgcn_fc_fwd_workspace(B, I, O)), "fc_fwd: workspace too small");
    dim3 grid((O + 31) / 32, (B + 31) / 32, S);
    note_dispatch(S > 1 ? "fc_fwd_kernel<split>" : "fc_fwd_kernel");
    if (S > 1) note_dispatch_more("fc_fwd_reduce_kernel");
    hipLaunchKernelGGL(fc_fwd_kernel, grid, dim3(FC_WAVES * 64), 0, stream, x, (long long)ldx, W, bias, y,
                       S > 1 ? (float*)workspace : nullptr, B, I, O, relu, cps);
    if (S > 1)
        hipLaunchKernelGGL(fc_fwd_reduce_kernel, dim3((B * O + 255) / 256), dim3(256), 0, stream, (const float*)workspace, bias,
                           y, B * O, O, S, relu);
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

extern "C" int chebgcn_fc_bwd(const float* x, int64_t ldx, const float* W, const float* g, const float* y, float* dW,
                              float* db, float* dx, int64_t lddx, int B, int I, int O, chebgcn_stream stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CG_REQUIRE(x && W && g && B > 0 && I > 0 && O > 0 && ldx >= I && (!dx || lddx >= I), "fc_bwd: bad argument");
    if (!chebgcn_fc_fwd_supported(B, I, O)) return CHEBGCN_EUNSUPPORTED;
    note_dispatch(dW ? "fc_bwd_w_kernel" : "");
    if (dW) {
        dim3 grid((O + 31) / 32, (I + 31) / 32);
        hipLaunchKernelGGL(fc_bwd_w_kernel, grid, dim3(FC_WAVES * 64), 0, stream, x, (long long)ldx, g, y, dW, db, B, I, O);
    }
    if (dx) {
        dim3 grid((I + 31) / 32, (B + 31) / 32);
        const bool vec = (O & 3) == 0 && ((uintptr_t)g & 15) == 0 && ((uintptr_t)W & 15) == 0 && (!y || ((uintptr_t)y & 15) == 0);
        note_dispatch_more(vec ? "fc_bwd_x_kernel<true>" : "fc_bwd_x_kernel<false>");
        if (vec)
            hipLaunchKernelGGL(fc_bwd_x_kernel<true>, grid, dim3(FC_WAVES * 64), 0, stream, g, y, W, dx, (long long)lddx, B, I, O);
        else
            hipLaunchKernelGGL(fc_bwd_x_kernel<false>, grid, dim3(FC_WAVES * 64), 0, stream, g, y, W, dx, (long long)lddx, B, I, O);
    }
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

extern "C" int chebgcn_softmax_xent(const float* logits, const void* labels, int labels_int64, float* loss, float* dlogits,
                                    int B, int C, chebgcn_stream stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CG_REQUIRE(logits && labels && loss && dlogits, "softmax_xent: NULL argument");
    CG_REQUIRE(B > 0 && C > 0, "softmax_xent: bad shape");
    note_dispatch(labels_int64 ? "softmax_xent_kernel<int64>" : "softmax_xent_kernel<int32>");
    if (labels_int64)
        hipLaunchKernelGGL(softmax_xent_kernel<long long>, dim3(1), dim3(256), 0, stream, logits, (const long long*)labels, loss,
                           dlogits, B, C);
    else
        hipLaunchKernelGGL(softmax_xent_kernel<int32_t>, dim3(1), dim3(256), 0, stream, logits, (const int32_t*)labels, loss,
                           dlogits, B, C);
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}
