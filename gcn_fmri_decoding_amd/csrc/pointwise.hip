// HBM-bound streaming kernels around the two hot kernels: gradient of pooling/ReLU/bias,
// input staging (perm_data_3d gather + layout change), the feature mean in front of the
// FC head, and the Adam update.  All accesses are coalesced along the vertex axis.
#include <algorithm>

#include "common.h"
#include "bias_grad_body.h"

namespace chebgcn {

// ---- d(pool o relu o bias): MaxPoolGrad + ReluGrad + bias reductions -------------------
// thread = one pre-pool element (f, m) and one of PARTS interleaved subsets of the batch; it loops
// over its windows so that the per-vertex bias gradient of b2relu (models_gcn.py:625-629) is a
// private register sum, and the PARTS partial sums of a vertex are added in LDS in a fixed order.
// PARTS > 1 is for small graphs (the reference's atlases have 246..1000 nodes), where M*F threads
// alone leave most of the chip idle.
template <int BIAS, int PARTS>
__global__ void __launch_bounds__(256)
brelu_pool_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                      const uint8_t* __restrict__ argmax, float* __restrict__ dy,
                      float* __restrict__ dbias, float* __restrict__ fpart, int B, int M, int Mp, int F, int pool,
                      int pool_kind, int relu, int Mpo) {
    constexpr int VB = 256 / PARTS;                  // vertices per block
    __shared__ float red[4];
    __shared__ float psum[PARTS > 1 ? 256 : 1];
    const int mloc = threadIdx.x % VB, part = threadIdx.x / VB;
    const int m = blockIdx.x * VB + mloc;
    const int f = blockIdx.y;
    const bool live = m < M;
    const int mo = live ? m / pool : 0;
    const int pos = m - mo * pool;
    const float inv = 1.0f / (float)pool;
    float sum = 0.f;
    if (live) {
        for (int b = part; b < B; b += PARTS) {
            const size_t oi = ((size_t)b * F + f) * Mpo + mo;
            // read once: streaming loads leave the L2 to dy, which the next two kernels read
            float g = __builtin_nontemporal_load(dout + oi);
            if (pool == 1) {
                if (relu) {
                    // argmax given: the ReLU mask contract_fwd leaves at pool == 1 (a bit per vertex)
                    const bool pos_out = argmax ? ((argmax[((size_t)b * F + f) * (Mp >> 2) + (m >> 2)] >> (m & 3)) & 1) != 0
                                                : __builtin_nontemporal_load(out + oi) > 0.f;
                    if (!pos_out) g = 0.f;
                }
            } else if (pool_kind == CHEBGCN_POOL_MAX) {
                const bool sel = argmax[oi] == pos;
                if (!sel || (relu && !(out[oi] > 0.f))) g = 0.f;
            } else {
                g *= inv;
                if (relu && !((argmax[oi] >> pos) & 1)) g = 0.f;
            }
            if (dy) dy[((size_t)b * F + f) * Mp + m] = g;     // NULL: bias gradient only (the contraction gradients gate dout themselves)
            sum += g;
        }
    }
    if (BIAS == CHEBGCN_BIAS_VERTEX) {
        if (PARTS > 1) {
            psum[threadIdx.x] = sum;
            __syncthreads();
            if (part == 0 && live) {
                float t = psum[mloc];
#pragma unroll
                for (int p = 1; p < PARTS; ++p) t += psum[p * VB + mloc];
                dbias[(size_t)f * Mp + m] = t;
            }
        } else if (live) {
            dbias[(size_t)f * Mp + m] = sum;
        }
    } else if (BIAS == CHEBGCN_BIAS_FILTER) {
        for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
        __syncthreads();
        // per-filter sum (b1relu): one partial per workgroup, summed in a fixed order by bias_filter_reduce_kernel
        if (threadIdx.x == 0) fpart[(size_t)f * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

// ---- ReluGrad + bias gradient of a pool == 1 layer from (dout, ReLU bit mask).  dy == NULL: the
// contraction gradients gate dout themselves (contract_bwd_*_relu) and nothing but dbias is written.
// thread = four consecutive vertices (one mask byte, one 16-byte load per window) of filter f and
// one of four interleaved subsets of the batch; fixed-order LDS sum of the four subsets.
// NP = 4 subsets of the batch x 64 quads per workgroup, or (small graphs: an atlas-sized layer would give the chip 64 workgroups
// that each walk the batch in 8 serial rounds) 16 subsets x 16 quads.
// DY16: the gated gradient is written as bf16 (RNE), [B][F][Mp] -- the operand of the bf16 contraction gradients, which
// round it to bf16 anyway (chebgcn_relu_grad_bf16)
// MASKED = false: dout is a gated gradient already (the layer above stored it gated: chebgcn_contract_fwd_gated) -- no mask is
// read, nothing but dbias is written
template <int BIAS, int NP = 4, bool DY16 = false, bool MASKED = true>
__global__ void __launch_bounds__(256)
bias_grad_relu_kernel(const float* __restrict__ dout, const uint8_t* __restrict__ mask, float* __restrict__ dy,
                      float* __restrict__ dbias, float* __restrict__ fpart, int B, int M, int Mp, int F,
                      size_t d_bstride, size_t d_fstride) {      // element strides of dout: F*Mp and Mp, or Mp and 0 (one plane per window)
    bias_grad_relu_body<BIAS, NP, DY16, MASKED>(dout, mask, dy, dbias, fpart, B, M, Mp, F, d_bstride, d_fstride, (int)blockIdx.x,
                                                (int)blockIdx.y, (int)gridDim.x);
}

// second stage of the per-filter bias gradient (b1relu, models_gcn.py:619-623): one wave per filter adds the
// per-workgroup partials in a fixed order (lane l takes partials l, l+64, ... in sequence, then a fixed
// butterfly) -- the result does not depend on the order the workgroups of the first stage ran in
__global__ void __launch_bounds__(64)
bias_filter_reduce_kernel(const float* __restrict__ fpart, float* __restrict__ dbias, int nblk) {
    const int f = blockIdx.x;
    float s = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 64) s += fpart[(size_t)f * nblk + i];
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
    if (threadIdx.x == 0) dbias[f] = s;
}

// ---- pooling between two vertex orders, and its gradient, through LDS --------------------------------------------------
// mpool1 / apool1 (lib_new/models_gcn.py:631-648) pool p consecutive vertices of the coarsening's tree order.  A model that
// keeps the vertices of a level in another order (cgcnn.vertex_order = 'length': the ordered recurrence kernels) pools through
// an index map: pooled vertex j (in ITS level's internal order) takes the source vertices pmap[j*p + i], i < p (positions in
// the source level's internal order, listed in tree order so that ties resolve as in the reference).  One workgroup = one
// plane (window, filter): the plane is read once, coalesced, into LDS and gathered from there.
//   sel (optional): what the gradient needs of the forward -- max: the winning i, 0xFF where `relu` and the maximum is not
//   positive (the ReLU in front of the pooling passes no gradient); average: bit i = member i was positive.
__global__ void __launch_bounds__(512)
pool_gather_fwd_kernel(const float* __restrict__ y, const int32_t* __restrict__ pmap, float* __restrict__ out,
                       uint8_t* __restrict__ sel, int Mp, int pool, int pool_kind, int relu, int Mo, int Mpo) {
    extern __shared__ __attribute__((aligned(16))) float pg_plane[];          // [Mp]
    const size_t pl = blockIdx.x;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const f32x4* src = reinterpret_cast<const f32x4*>(y + pl * Mp);
    // (eight 16-byte loads in flight per thread: one load and one LDS store per iteration compiles to a full wait per piece)
    const int Mq = Mp >> 2, nt = (int)blockDim.x;
    for (int i0 = threadIdx.x; i0 < Mq; i0 += 8 * nt) {
        f32x4 r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) r[u] = __builtin_nontemporal_load(src + min(i0 + u * nt, Mq - 1));
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i0 + u * nt < Mq) reinterpret_cast<f32x4*>(pg_plane)[i0 + u * nt] = r[u];
    }
    __syncthreads();
    const float inv = 1.0f / (float)pool;
    if (pool == 4 && pmap && (reinterpret_cast<uintptr_t>(pmap) & 15) == 0) {
        // the common case (p = 4: one 16-byte record of the map per pooled vertex), four pooled vertices per thread in flight
        for (int m0 = threadIdx.x; m0 < Mpo; m0 += 4 * nt) {
            int4 pm[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) pm[u] = reinterpret_cast<const int4*>(pmap)[min(m0 + u * nt, Mo - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int mo = m0 + u * nt;
                if (mo >= Mpo) continue;
                const float v[4] = {pg_plane[pm[u].x], pg_plane[pm[u].y], pg_plane[pm[u].z], pg_plane[pm[u].w]};
                float best = v[0], sum = v[0];
                int arg = 0, mask = v[0] > 0.f ? 1 : 0;
#pragma unroll
                for (int i = 1; i < 4; ++i) {
                    if (v[i] > best) { best = v[i]; arg = i; }
                    sum += v[i];
                    if (v[i] > 0.f) mask |= 1 << i;
                }
                const bool live = mo < Mo;
                const bool is_max = pool_kind == CHEBGCN_POOL_MAX;
                out[pl * Mpo + mo] = live ? (is_max ? best : sum * inv) : 0.f;
                if (sel) sel[pl * Mpo + mo] = (uint8_t)(!live ? 0 : is_max ? ((relu && !(best > 0.f)) ? 0xFF : arg) : mask);
            }
        }
        return;
    }
    for (int mo = threadIdx.x; mo < Mpo; mo += blockDim.x) {
        float o = 0.f;
        int s = 0;
        if (mo < Mo) {
            float best = 0.f, sum = 0.f;
            int arg = 0, mask = 0;
            for (int i = 0; i < pool; ++i) {
                const int idx = pmap ? pmap[(size_t)mo * pool + i] : mo * pool + i;
                const float v = pg_plane[idx];
                if (i == 0 || v > best) { best = v; arg = i; }
                sum += v;
                if (v > 0.f && i < 8) mask |= 1 << i;
            }
            if (pool_kind == CHEBGCN_POOL_MAX) {
                o = best;
                s = (relu && !(best > 0.f)) ? 0xFF : arg;
            } else {
                o = sum * inv;
                s = mask;
            }
        }
        out[pl * Mpo + mo] = o;
        if (sel) sel[pl * Mpo + mo] = (uint8_t)s;
    }
}

// MaxPoolGrad / AvgPoolGrad + ReluGrad + bias gradient with 16-byte stores (and, with `smap`, across two vertex orders):
//   dy[b][f][v] = d(loss)/d(pre-bias activation of source vertex v),  v in the source level's internal order,
//   smap[v] = j*pool + i : v is member i of pooled vertex j (NULL: v itself, the tree order); -1: no gradient (padding).
// Workgroup (pb, f) walks the windows of batch part pb: per window it stages {dout, sel} of the pooled plane in LDS (8 bytes per
// pooled vertex, coalesced reads, double-buffered: one barrier per window), every thread then looks its own source quads up
// and stores whole 16-byte pieces of dy.  The bias gradient of a vertex is a register sum over the part's windows; the NPB
// partial sums are added in part order by pool_bias_reduce_kernel (fixed order: bit-reproducible, no atomics).
// `out` (optional): the forward result, read where `sel` carries no dead flag (the fused contraction epilogue's argmax byte).
// NT = 512 (two workgroups per CU) or 1024 (one: planes of more than 8192 vertices -- with 512 threads their source quads were
// split between two workgroups that each staged the whole pooled plane: 1.4x the algorithmic HBM traffic by the counters)
template <int BIAS, bool HAS_OUT, int EPT, int NT>
__global__ void __launch_bounds__(NT, 4)         // (four waves per SIMD: 128 registers)
pool_scatter_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out, const uint8_t* __restrict__ sel,
                        const int32_t* __restrict__ smap, float* __restrict__ dy, float* __restrict__ part, int B, int M, int Mp,
                        int F, int lgp, int pool_kind, int relu, int Mo, int Mpo) {
    extern __shared__ __attribute__((aligned(16))) float2 ps_ent[];            // [2][Mpo]
    constexpr int QPT = 4;                                 // source quads per thread and pass
    // EPT: pooled vertices a thread stages per window (prefetched one window ahead while Mpo <= EPT * NT): 2, 4 or 8
    const int f = blockIdx.y, pb = blockIdx.x, NPB = gridDim.x;
    const int b0 = (int)((long long)B * pb / NPB), b1 = (int)((long long)B * (pb + 1) / NPB);
    const int Mq = Mp >> 2, pmask = (1 << lgp) - 1;
    // blockIdx.z: the source quads are split between gridDim.z workgroups (each stages the whole pooled plane: an L2 hit for
    // all but the first) where batch parts x filters alone leave the chip short of workgroups
    const int qpz = (Mq + (int)gridDim.z - 1) / (int)gridDim.z;
    const int qs = (int)blockIdx.z * qpz, qe = min(Mq, qs + qpz);
    const float inv = 1.0f / (float)(1 << lgp);
    const bool pre = Mpo <= EPT * NT;
    const bool is_max = pool_kind == CHEBGCN_POOL_MAX;
    // One pooled vertex of a window: {gradient (scaled for the average), selection byte with the dead flag}.  Requests and
    // their use are separate and free of data-dependent branches: every load of a window is in flight before the first is used
    // (a conditional load behind a loaded value made hipcc wait for each of them in turn)
    const uint8_t* sel8 = sel ? sel : reinterpret_cast<const uint8_t*>(dout);     // (no selection bytes: average without ReLU, ignored)
    auto request = [&](int b, int j, float& g, int& sb, float& o) __attribute__((always_inline)) {
        const size_t po = ((size_t)b * F + f) * Mpo;           // uniform: scalar base + one 32-bit lane offset per load
        const int jj = min(j, Mo - 1);
        g = __builtin_nontemporal_load(dout + po + jj);
        sb = (int)(sel8 + po)[jj];
        o = HAS_OUT ? (out + po)[jj] : 1.f;
    };
    auto finish = [&](int j, float g, int sb, float o) __attribute__((always_inline)) -> float2 {
        if (is_max) {
            if (HAS_OUT && relu && !(o > 0.f)) sb = 0xFF;
        } else {
            g *= inv;
            if (!relu) sb = 0xFF;                    // every member takes its share
        }
        const bool live = j < Mo;
        return make_float2(live ? g : 0.f, __int_as_float(live ? sb : 0xFF));
    };
    for (int q0 = qs; q0 < qe; q0 += QPT * NT) {           // (more than 8192 vertices per split: another pass over the windows)
        int code[QPT][4];
        float4 acc[QPT];
#pragma unroll
        for (int u = 0; u < QPT; ++u) {
            const int q = q0 + u * NT + (int)threadIdx.x;
            acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int v = 4 * q + i;
                code[u][i] = (q < qe && v < M) ? (smap ? smap[v] : v) : -1;
            }
        }
        float ng[EPT], no[EPT];                            // the next window's entries, requested while this one is computed
        int ns[EPT];
        if (pre && b0 < b1) {
#pragma unroll
            for (int k = 0; k < EPT; ++k) request(b0, k * NT + (int)threadIdx.x, ng[k], ns[k], no[k]);     // (unconditional: clamped)
        }
        int buf = 0;
        for (int b = b0; b < b1; ++b) {
            float2* ent = ps_ent + (size_t)buf * Mpo;
            if (pre) {
#pragma unroll
                for (int k = 0; k < EPT; ++k) {
                    const int j = k * NT + (int)threadIdx.x;
                    if (j < Mpo) ent[j] = finish(j, ng[k], ns[k], no[k]);
                }
            } else {
                for (int j = threadIdx.x; j < Mpo; j += NT) {
                    float g, o;
                    int sb;
                    request(b, j, g, sb, o);
                    ent[j] = finish(j, g, sb, o);
                }
            }
            // LDS-only barrier: __syncthreads() also waits (s_waitcnt vmcnt(0)) for the previous window's dy STORES to be
            // acknowledged by memory -- a store round trip per window, 21k cycles of a window's ~2k (measured: 0.21 of the roofline)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (pre && b + 1 < b1) {
#pragma unroll
                for (int k = 0; k < EPT; ++k) request(b + 1, k * NT + (int)threadIdx.x, ng[k], ns[k], no[k]);
            }
#pragma unroll
            for (int u = 0; u < QPT; ++u) {
                const int q = q0 + u * NT + (int)threadIdx.x;
                if (q < qe) {
                    float2 e[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) e[i] = ent[max(code[u][i], 0) >> lgp];
                    float r[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int c = code[u][i];
                        const int sb = __float_as_int(e[i].y), pos = c & pmask;
                        const bool take = c >= 0 && (is_max ? sb == pos : ((sb >> pos) & 1) != 0);
                        r[i] = take ? e[i].x : 0.f;
                    }
                    if (dy) *reinterpret_cast<float4*>(dy + ((size_t)b * F + f) * Mp + 4 * q) = make_float4(r[0], r[1], r[2], r[3]);
                    acc[u].x += r[0]; acc[u].y += r[1]; acc[u].z += r[2]; acc[u].w += r[3];
                }
            }
            buf ^= 1;
        }
        if (BIAS != CHEBGCN_BIAS_NONE) {
#pragma unroll
            for (int u = 0; u < QPT; ++u) {
                const int q = q0 + u * NT + (int)threadIdx.x;
                if (q < qe) *reinterpret_cast<float4*>(part + ((size_t)pb * F + f) * Mp + 4 * q) = acc[u];
            }
        }
        __syncthreads();                                   // (a further pass refills the buffers)
    }
}

// the NPB partial bias gradients of pool_scatter_bwd_kernel, added in part order; b1relu: then over the vertices of a filter
template <int BIAS>
__global__ void __launch_bounds__(256)
pool_bias_reduce_kernel(const float* __restrict__ part, float* __restrict__ dbias, int NPB, int F, int Mp) {
    const int f = blockIdx.y;
    const int Mq = Mp >> 2;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (BIAS == CHEBGCN_BIAS_VERTEX) {
        const int q = blockIdx.x * 256 + threadIdx.x;
        if (q >= Mq) return;
        for (int p = 0; p < NPB; ++p) {
            const float4 o = *reinterpret_cast<const float4*>(part + ((size_t)p * F + f) * Mp + 4 * q);
            t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
        }
        *reinterpret_cast<float4*>(dbias + (size_t)f * Mp + 4 * q) = t;
    } else {
        __shared__ float red[256];
        float s = 0.f;
        for (int q = threadIdx.x; q < Mq; q += 256) {
            t = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int p = 0; p < NPB; ++p) {
                const float4 o = *reinterpret_cast<const float4*>(part + ((size_t)p * F + f) * Mp + 4 * q);
                t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
            }
            s += (t.x + t.y) + (t.z + t.w);
        }
        red[threadIdx.x] = s;
        __syncthreads();
        for (int d = 128; d > 0; d >>= 1) {
            if ((int)threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
            __syncthreads();
        }
        if (threadIdx.x == 0) dbias[f] = red[0];
    }
}

// ---- standalone bias + ReLU + pooling forward (b1relu / b2relu / mpool1 / apool1 called on
// their own, lib_new/models_gcn.py:619-648); the fused form lives in contract.hip ----------
__global__ void __launch_bounds__(256)
brelu_pool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ bias, float* __restrict__ out,
                      uint8_t* __restrict__ argmax, int M, int Mp, int F, int pool, int pool_kind, int relu,
                      int bias_kind, int Mo, int Mpo) {
    const int mo = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y, b = blockIdx.z;
    if (mo >= Mpo) return;
    const size_t oi = ((size_t)b * F + f) * Mpo + mo;
    if (mo >= Mo) { out[oi] = 0.f; return; }
    const float* xp = x + ((size_t)b * F + f) * Mp + (size_t)mo * pool;
    float best = 0.f, sum = 0.f;
    int arg = 0, mask = 0;
    for (int i = 0; i < pool; ++i) {
        float v = xp[i];
        if (bias_kind == CHEBGCN_BIAS_FILTER) v += bias[f];
        else if (bias_kind == CHEBGCN_BIAS_VERTEX) v += bias[(size_t)f * Mp + (size_t)mo * pool + i];
        if (relu) v = fmaxf(v, 0.f);
        if (i == 0 || v > best) { best = v; arg = i; }
        sum += v;
        if (v > 0.f && i < 8) mask |= 1 << i;
    }
    if (pool == 1 || pool_kind == CHEBGCN_POOL_MAX) {
        out[oi] = best;
        if (argmax && pool > 1) argmax[oi] = (uint8_t)arg;
    } else {
        out[oi] = sum / (float)pool;
        if (argmax) argmax[oi] = (uint8_t)mask;
    }
}

// ---- perm_data_3d gather into plane layout (lib_new/coarsening.py:244-265) --------------
// block = 64 output vertices x all F features of one sample, transposed through LDS so
// that reads run along the feature axis of x[S][N][F] and writes along the vertex axis.
__global__ void __launch_bounds__(256)
perm_data_kernel(const float* __restrict__ x, const int32_t* __restrict__ perm,
                 const int32_t* __restrict__ sample, float* __restrict__ out, int N, int M, int Mp, int F) {
    extern __shared__ float tile[];                     // [F][65]
    __shared__ int nodes[64];                           // source vertex of the block's 64 columns (N: none -> 0)
    const int s = blockIdx.y;
    const int i0 = blockIdx.x * 64;
    const size_t src = (size_t)(sample ? sample[s] : s) * N * F;
    if (threadIdx.x < 64) {
        const int i = i0 + threadIdx.x;
        const int node = i < M ? (perm ? perm[i] : i) : N;
        nodes[threadIdx.x] = node < N ? node : N;
    }
    __syncthreads();
    // four gathers in flight per thread (one load, one LDS store per iteration compiles to a full wait per element,
    // behind the index load it depends on)
    for (int e0 = threadIdx.x; e0 < 64 * F; e0 += 4 * 256) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + 256 * u, ec = e < 64 * F ? e : 0;
            const int j = ec / F, f = ec - j * F;
            const int node = nodes[j];
            const float t = x[src + (size_t)(node < N ? node : 0) * F + f];        // unconditional load on a clamped address
            v[u] = node < N ? t : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + 256 * u;
            if (e < 64 * F) {
                const int j = e / F, f = e - j * F;
                tile[f * 65 + j] = v[u];
            }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * F; e += 256) {
        const int f = e >> 6, j = e & 63;
        const int i = i0 + j;
        if (i < Mp) out[((size_t)s * F + f) * Mp + i] = tile[f * 65 + j];
    }
}

__global__ void __launch_bounds__(256)
from_plane_kernel(const float* __restrict__ xp, float* __restrict__ out, int M, int Mp, int F) {
    extern __shared__ float tile[];                     // [F][65]
    const int b = blockIdx.y;
    const int i0 = blockIdx.x * 64;
    for (int e = threadIdx.x; e < 64 * F; e += 256) {
        const int f = e >> 6, j = e & 63;
        const int i = i0 + j;
        tile[f * 65 + j] = (i < M) ? xp[((size_t)b * F + f) * Mp + i] : 0.f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * F; e += 256) {
        const int j = e / F, f = e - j * F;
        const int i = i0 + j;
        if (i < M) out[((size_t)b * M + i) * F + f] = tile[f * 65 + j];
    }
}

// ---- tf.reduce_mean(x, -1) in front of the FC head (models_gcn.py:673) ------------------
__global__ void __launch_bounds__(256)
feature_mean_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int M, int Mp, int F) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (m >= M) return;
    const float* p = x + (size_t)b * F * Mp + m;
    float s = 0.f;
    for (int f0 = 0; f0 < F; f0 += 8) {               // eight planes in flight, added in filter order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(f0 + u < F ? f0 + u : f0) * Mp];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (f0 + u < F) s += v[u];
    }
    y[(size_t)b * M + m] = s / (float)F;
}

__global__ void __launch_bounds__(256)
feature_mean_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int M, int Mp, int F) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (m >= Mp) return;
    const float g = (m < M) ? dy[(size_t)b * M + m] / (float)F : 0.f;
    float* p = dx + (size_t)b * F * Mp + m;
    for (int f = 0; f < F; ++f) p[(size_t)f * Mp] = g;
}

// ---- Adam, TensorFlow form (epsilon outside the bias correction) ------------------------
__global__ void __launch_bounds__(256)
adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
            float* __restrict__ v, int64_t n, float lr_t, float b1, float b2, float eps, float gscale,
            float l2) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float pi = p[i];
        const float gi = fmaf(l2, pi, gscale * g[i]);
        const float mi = m[i] + (1.f - b1) * (gi - m[i]);
        const float vi = v[i] + (1.f - b2) * (gi * gi - v[i]);
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - lr_t * mi / (sqrtf(vi) + eps);
    }
}

// the same with the step size read from device memory: a captured HIP graph of the training step replays with the
// value the host wrote before the launch (lr_t changes every step)
__global__ void __launch_bounds__(256)
adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                float* __restrict__ v, int64_t n, const float* __restrict__ lr_t_dev, float b1, float b2, float eps,
                float gscale, float l2) {
    const float lr_t = *lr_t_dev;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float pi = p[i];
        const float gi = fmaf(l2, pi, gscale * g[i]);
        const float mi = m[i] + (1.f - b1) * (gi - m[i]);
        const float vi = v[i] + (1.f - b2) * (gi * gi - v[i]);
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - lr_t * mi / (sqrtf(vi) + eps);
    }
}

// Adam as above (step size by value, or from device memory when lr_t_dev is set) that also leaves the sum of squares of the
// PRE-update variables it walks -- the L2 term of the loss (models_gcn.py:262-266: regularization * sum tf.nn.l2_loss) -- as one
// partial per workgroup (fixed-order tree inside the workgroup); chebgcn_loss_bookkeeping adds the partials in index order.
__global__ void __launch_bounds__(256)
adam_sq_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
               float lr_t_val, const float* __restrict__ lr_t_dev, float b1, float b2, float eps, float gscale, float l2,
               float* __restrict__ sq_part, int64_t n_reg) {      // elements [n_reg, n): no L2 term, not in the sum of squares
    __shared__ float red[256];
    const float lr_t = lr_t_dev ? *lr_t_dev : lr_t_val;
    float sq = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float pi = p[i];
        const bool rg = i < n_reg;
        sq = rg ? fmaf(pi, pi, sq) : sq;
        const float gi = fmaf(rg ? l2 : 0.f, pi, gscale * g[i]);
        const float mi = m[i] + (1.f - b1) * (gi - m[i]);
        const float vi = v[i] + (1.f - b2) * (gi * gi - v[i]);
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - lr_t * mi / (sqrtf(vi) + eps);
    }
    red[threadIdx.x] = sq;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) sq_part[blockIdx.x] = red[0];
}

// loss = cross_entropy + half_reg * sum(partials);  the ExponentialMovingAverage(0.9) of the loss with its zero-debiasing
// (models_gcn.py:269-275): ema += (1 - decay) * (loss - ema);  loss_average = ema * corr.  One wave, one launch (torch: dot x 2,
// add, lerp, mul).
// (1024 threads: the ~3700 partials of the configs[1] variables were 58 dependent loads per lane of one wave, 16.7 us of the
// step; four per thread and a fixed tree are 4 us.  Deterministic: the order of the additions depends on nparts only.)
__global__ void __launch_bounds__(1024)
loss_bookkeeping_kernel(const float* __restrict__ ce, const float* __restrict__ sq_part, int nparts, float half_reg,
                        float* __restrict__ ema, float decay, float corr_val, const float* __restrict__ corr_dev,
                        float* __restrict__ loss_out, float* __restrict__ loss_average_out) {
    __shared__ float red[16];
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    float s = 0.f;
    for (int i0 = threadIdx.x; i0 < nparts; i0 += 4 * 1024) {
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = i0 + 1024 * u < nparts ? sq_part[i0 + 1024 * u] : 0.f;
        s += (v[0] + v[1]) + (v[2] + v[3]);
    }
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        s = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) s += red[w];
        const float loss = *ce + half_reg * s;
        const float e = *ema + (1.f - decay) * (loss - *ema);
        *ema = e;
        if (loss_out) *loss_out = loss;
        *loss_average_out = e * (corr_dev ? *corr_dev : corr_val);
    }
}

}  // namespace chebgcn

using namespace chebgcn;

extern "C" int chebgcn_brelu_pool_fwd(const float* x, const float* bias, int bias_kind, float* out, uint8_t* argmax,
                                      int B, int M, int F, int pool, int pool_kind, int relu,
                                      chebgcn_stream stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CG_REQUIRE(x && out, "brelu_pool_fwd: NULL argument");
    CG_REQUIRE(B > 0 && M > 0 && F > 0 && F <= 65535 && B <= 65535, "brelu_pool_fwd: bad shape");
    CG_REQUIRE(pool >= 1 && (pool & (pool - 1)) == 0 && pool <= 128 && M % pool == 0, "brelu_pool_fwd: bad pool %d", pool);
    CG_REQUIRE(bias_kind == CHEBGCN_BIAS_NONE || bias, "brelu_pool_fwd: bias is NULL");
    CG_REQUIRE(!(pool_kind == CHEBGCN_POOL_AVG && relu && argmax && pool > 8),
               "brelu_pool_fwd: average pooling keeps a ReLU mask only for pool <= 8");
    const int Mp = plane_stride(M), Mo = M / pool, Mpo = plane_stride(Mo);
    dim3 grid((Mpo + 255) / 256, F, B);
    note_dispatch("brelu_pool_fwd_kernel");
    hipLaunchKernelGGL(brelu_pool_fwd_kernel, grid, dim3(256), 0, stream, x, bias, out, argmax, M, Mp, F, pool,
                       pool_kind, relu, bias_kind, Mo, Mpo);
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

// workgroups along the vertex axis of the gradient kernels below (= per-filter partials of a b1relu layer)
static int brelu_bwd_blocks(int M, int F, int pool, int relu, bool have_mask, int* parts_out, bool bias_only = false) {
    if (pool == 1 && ((relu && have_mask) || bias_only)) return bias_grad_blocks(M, F, parts_out);       // bias_grad_relu_kernel
    // enough workgroups for the chip: small graphs split the batch over 4 or 8 thread groups
    const int parts = ((M + 255) / 256) * F >= 1024 ? 1 : ((M + 63) / 64) * F >= 1024 ? 4 : 8;
    if (parts_out) *parts_out = parts;
    return (M + 256 / parts - 1) / (256 / parts);
}

extern "C" size_t chebgcn_pool_scatter_bwd_workspace(int B, int M, int F, int pool, int bias_kind);
static bool pool_scatter_fits(int M, int pool);
static int pool_scatter_launch(const float* dout, const float* out, const uint8_t* sel, const int32_t* smap, float* dy, float* dbias,
                               int bias_kind, int B, int M, int F, int pool, int pool_kind, int relu, void* workspace,
                               size_t workspace_bytes, hipStream_t stream);

extern "C" size_t chebgcn_brelu_pool_bwd_workspace(int B, int M, int F, int pool, int bias_kind) {
    if (M <= 0 || F <= 0) return 0;
    // pooled layers: the 16-byte-store kernel (pool_scatter_bwd_kernel) leaves per-batch-part partial bias sums
    const size_t pooled = (pool > 1 && plane_stride(M) >= 2048 && pool_scatter_fits(M, pool)) ? chebgcn_pool_scatter_bwd_workspace(B, M, F, pool, bias_kind) : 0;
    const size_t filter = bias_kind == CHEBGCN_BIAS_FILTER ? (size_t)F * ((size_t)(M + 31) / 32 + 1) * sizeof(float) : 0;   // the finest split: 32 vertices per workgroup
    return std::max(pooled, filter);
}

extern "C" int chebgcn_brelu_pool_bwd(const float* dout, const float* out, const uint8_t* argmax, float* dy,
                                      float* dbias, int bias_kind, int B, int M, int F, int pool,
                                      int pool_kind, int relu, void* workspace, size_t workspace_bytes,
                                      chebgcn_stream stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CG_REQUIRE(dout && (dy || (dbias && bias_kind != CHEBGCN_BIAS_NONE)), "brelu_pool_bwd: NULL argument");
    CG_REQUIRE(B > 0 && M > 0 && F > 0 && F <= 65535, "brelu_pool_bwd: bad shape");
    CG_REQUIRE(pool >= 1 && (pool & (pool - 1)) == 0 && pool <= 128 && M % pool == 0, "brelu_pool_bwd: bad pool %d", pool);
    CG_REQUIRE(!relu || out || (pool == 1 && argmax), "brelu_pool_bwd: relu needs the forward output (or its mask at pool 1)");
    CG_REQUIRE(pool == 1 || argmax || (pool_kind == CHEBGCN_POOL_AVG && !relu), "brelu_pool_bwd: pooling needs argmax/mask");
    CG_REQUIRE(bias_kind == CHEBGCN_BIAS_NONE || dbias, "brelu_pool_bwd: dbias is NULL");
    const int Mp = plane_stride(M), Mpo = plane_stride(M / pool);
    // pooled layers: whole 16-byte pieces of dy per store, the pooled plane staged in LDS (pool_scatter_bwd_kernel; the scalar
    // kernel below ran at 0.18-0.22 of the HBM roofline in the six-level pooling network).  It needs the workspace
    // chebgcn_brelu_pool_bwd_workspace() reports; a caller without one gets the scalar kernel.
    // (planes of fewer than 2048 vertices: a workgroup's 512 threads have less than a quad each and a window is a latency chain --
    // the scalar kernel with its batch split over thread groups is faster there: 0.033 against 0.055 ms at M = 792, F = 128)
    if (pool > 1 && dy && Mp >= 2048 && pool_scatter_fits(M, pool) && (pool_kind == CHEBGCN_POOL_MAX || pool <= 8 || !relu) &&
        (bias_kind == CHEBGCN_BIAS_NONE ||
         (workspace && workspace_bytes >= chebgcn_pool_scatter_bwd_workspace(B, M, F, pool, bias_kind))))
        return pool_scatter_launch(dout, out, argmax, nullptr, dy, dbias, bias_kind, B, M, F, pool, pool_kind, relu, workspace,
                                   workspace_bytes, stream);
    int parts = 0;
    // no ReLU, no pooling, no dy: the plain sum of dout over the windows -- the bias gradient of a layer whose gated dy the
    // layer above stored (chebgcn_contract_fwd_gated)
    const bool bias_only = pool == 1 && !relu && !dy;
    const int nblk = brelu_bwd_blocks(M, F, pool, relu, argmax != nullptr, &parts, bias_only);
    float* fpart = nullptr;
    if (bias_kind == CHEBGCN_BIAS_FILTER) {
        CG_REQUIRE(workspace && workspace_bytes >= (size_t)F * nblk * sizeof(float),
                   "brelu_pool_bwd: the per-filter bias gradient needs a workspace of chebgcn_brelu_pool_bwd_workspace() bytes");
        fpart = static_cast<float*>(workspace);
    }
    if (bias_only) {
        const dim3 grid(nblk, F);
#define CG_BGS(BK)                                                                                                         \
    do {                                                                                                                   \
        note_dispatch(parts == 16 ? "bias_grad_sum_kernel<" #BK ",16>" : "bias_grad_sum_kernel<" #BK ",4>");               \
        if (parts == 16)                                                                                                   \
            hipLaunchKernelGGL((bias_grad_relu_kernel<BK, 16, false, false>), grid, dim3(256), 0, stream, dout, nullptr, nullptr, \
                               dbias, fpart, B, M, Mp, F, (size_t)F * Mp, (size_t)Mp);                                     \
        else                                                                                                               \
            hipLaunchKernelGGL((bias_grad_relu_kernel<BK, 4, false, false>), grid, dim3(256), 0, stream, dout, nullptr, nullptr, \
                               dbias, fpart, B, M, Mp, F, (size_t)F * Mp, (size_t)Mp);                                     \
    } while (0)
        if (bias_kind == CHEBGCN_BIAS_FILTER) CG_BGS(CHEBGCN_BIAS_FILTER);
        else CG_BGS(CHEBGCN_BIAS_VERTEX);
#undef CG_BGS
    } else if (pool == 1 && relu && argmax) {                  // ReLU mask of contract_fwd: vertices in fours, dy optional
        const dim3 grid(nblk, F);
#define CG_BGR(BK)                                                                                                         \
    do {                                                                                                                   \
        note_dispatch(parts == 16 ? "bias_grad_relu_kernel<" #BK ",16>" : "bias_grad_relu_kernel<" #BK ",4>");             \
        if (parts == 16)                                                                                                   \
            hipLaunchKernelGGL((bias_grad_relu_kernel<BK, 16>), grid, dim3(256), 0, stream, dout, argmax, dy, dbias, fpart, B, M, \
                               Mp, F, (size_t)F * Mp, (size_t)Mp);                                                         \
        else                                                                                                               \
            hipLaunchKernelGGL((bias_grad_relu_kernel<BK, 4>), grid, dim3(256), 0, stream, dout, argmax, dy, dbias, fpart, B, M,  \
                               Mp, F, (size_t)F * Mp, (size_t)Mp);                                                         \
    } while (0)
        if (bias_kind == CHEBGCN_BIAS_FILTER) CG_BGR(CHEBGCN_BIAS_FILTER);
        else if (bias_kind == CHEBGCN_BIAS_VERTEX) CG_BGR(CHEBGCN_BIAS_VERTEX);
        else CG_BGR(CHEBGCN_BIAS_NONE);
#undef CG_BGR
    } else {
#define CG_BRELU(BK, PARTS)                                                                                           \
    do {                                                                                                              \
        note_dispatch("brelu_pool_bwd_kernel<" #BK "," #PARTS ">");                                                   \
        hipLaunchKernelGGL((brelu_pool_bwd_kernel<BK, PARTS>), dim3(nblk, F), dim3(256), 0, stream, dout, out, argmax, dy, \
                           dbias, fpart, B, M, Mp, F, pool, pool_kind, relu, Mpo);                                    \
    } while (0)
#define CG_BRELU_P(BK)                                                        \
    do {                                                                      \
        if (parts == 1) CG_BRELU(BK, 1);                                      \
        else if (parts == 4) CG_BRELU(BK, 4);                                 \
        else CG_BRELU(BK, 8);                                                 \
    } while (0)
        if (bias_kind == CHEBGCN_BIAS_FILTER) CG_BRELU_P(CHEBGCN_BIAS_FILTER);
        else if (bias_kind == CHEBGCN_BIAS_VERTEX) CG_BRELU_P(CHEBGCN_BIAS_VERTEX);
        else CG_BRELU_P(CHEBGCN_BIAS_NONE);
#undef CG_BRELU_P
#undef CG_BRELU
    }
    CG_HIP(hipGetLastError());
    if (bias_kind == CHEBGCN_BIAS_FILTER) {
        note_dispatch_more("bias_filter_reduce_kernel");
        hipLaunchKernelGGL(bias_filter_reduce_kernel, dim3(F), dim3(64), 0, stream, fpart, dbias, nblk);
        CG_HIP(hipGetLastError());
    }
    return CHEBGCN_OK;
}

// batch parts of pool_scatter_bwd_kernel: enough workgroups for the chip; with a bias gradient each part leaves a partial
// sum per vertex (written and read back once), so at most an eighth of the windows' own traffic goes into them
static int pool_bwd_parts(int B, int F, int bias_kind) {
    int n = (512 + F - 1) / F;
    if (bias_kind != CHEBGCN_BIAS_NONE) n = std::min(n, std::max(1, B / 8));
    return std::max(1, std::min(n, B));
}

extern "C" int chebgcn_pool_gather_fwd(const float* y, const int32_t* pmap, float* out, uint8_t* sel, int B, int M, int F,
                                       int pool, int pool_kind, int relu, chebgcn_stream stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CG_REQUIRE(y && out, "pool_gather_fwd: NULL argument");
    CG_REQUIRE(B > 0 && M > 0 && F > 0 && (int64_t)B * F < (1ll << 31), "pool_gather_fwd: bad shape");
    CG_REQUIRE(pool >= 2 && (pool & (pool - 1)) == 0 && pool <= 128 && M % pool == 0, "pool_gather_fwd: bad pool %d", pool);
    CG_REQUIRE(pool_kind == CHEBGCN_POOL_MAX || pool_kind == CHEBGCN_POOL_AVG, "pool_gather_fwd: bad pool kind");
    CG_REQUIRE(!(pool_kind == CHEBGCN_POOL_AVG && relu && sel && pool > 8), "pool_gather_fwd: average pooling keeps a ReLU mask only for pool <= 8");
    const int Mp = plane_stride(M), Mo = M / pool, Mpo = plane_stride(Mo);
    const size_t lds = (size_t)Mp * sizeof(float);
    CG_REQUIRE(lds <= 160 * 1024, "pool_gather_fwd: a plane of %d vertices does not fit the LDS", M);
    const int nt = Mp >= 8192 ? 512 : 256;
    static size_t lds_set = 0;
    if (lds > lds_set) {
        CG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pool_gather_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set = lds;
    }
    note_dispatch(pmap ? "pool_gather_fwd_kernel<map>" : "pool_gather_fwd_kernel");
    hipLaunchKernelGGL(pool_gather_fwd_kernel, dim3((unsigned)(B * F)), dim3(nt), lds, stream, y, pmap, out, sel, Mp, pool, pool_kind, relu,
                       Mo, Mpo);
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

extern "C" size_t chebgcn_pool_scatter_bwd_workspace(int B, int M, int F, int pool, int bias_kind) {
    (void)pool;
    if (bias_kind == CHEBGCN_BIAS_NONE || B <= 0 || M <= 0 || F <= 0) return 0;
    return (size_t)pool_bwd_parts(B, F, bias_kind) * F * plane_stride(M) * sizeof(float);
}

static bool pool_scatter_fits(int M, int pool) { return (size_t)2 * plane_stride(M / pool) * sizeof(float2) <= 160 * 1024; }

static int pool_scatter_launch(const float* dout, const float* out, const uint8_t* sel, const int32_t* smap, float* dy, float* dbias,
                               int bias_kind, int B, int M, int F, int pool, int pool_kind, int relu, void* workspace,
                               size_t workspace_bytes, hipStream_t stream) {
    const int Mp = plane_stride(M), Mo = M / pool, Mpo = plane_stride(Mo);
    int lgp = 0;
    while ((1 << lgp) < pool) ++lgp;
    const int NPB = pool_bwd_parts(B, F, bias_kind);
    float* part = nullptr;
    if (bias_kind != CHEBGCN_BIAS_NONE) {
        CG_REQUIRE(workspace && workspace_bytes >= (size_t)NPB * F * Mp * sizeof(float),
                   "pool_scatter_bwd: the bias gradient needs a workspace of chebgcn_pool_scatter_bwd_workspace() bytes");
        part = static_cast<float*>(workspace);
    }
    const size_t lds = (size_t)2 * Mpo * sizeof(float2);
    // planes of more than 8192 vertices: 1024 threads, four quads each, cover 16384 vertices -- one workgroup stages the pooled
    // plane of a window once; smaller planes: 512 threads, the source quads split between up to four workgroups where batch
    // parts x filters leave fewer than two workgroups per CU (and beyond 16384 vertices per plane: as many as it takes)
    const bool big = Mp / 4 > 2048;
    const int nth = big ? 1024 : 512;
    const int VS = big ? std::min(8, (Mp / 4 + 4 * nth - 1) / (4 * nth))
                       : std::max(1, std::min(std::min(4, (512 + NPB * F - 1) / (NPB * F)), (Mp / 4 + 511) / 512));
    const dim3 grid(NPB, F, VS);
#define CG_PSB4(BK, HO, EP, NTH)                                                                                                \
    do {                                                                                                                        \
        static size_t lds_set = 0;                                                                                              \
        if (lds > lds_set) {                                                                                                    \
            CG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pool_scatter_bwd_kernel<BK, HO, EP, NTH>),                 \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                                  \
            lds_set = lds;                                                                                                      \
        }                                                                                                                       \
        hipLaunchKernelGGL((pool_scatter_bwd_kernel<BK, HO, EP, NTH>), grid, dim3(NTH), lds, stream, dout, out, sel, smap, dy, part, B, M, \
                           Mp, F, lgp, pool_kind, relu, Mo, Mpo);                                                               \
    } while (0)
#define CG_PSB2(BK, HO)                                                                                                         \
    do {                                                                                                                        \
        if (big) {                                                                                                              \
            if (Mpo <= 4096) CG_PSB4(BK, HO, 4, 1024);                                                                          \
            else CG_PSB4(BK, HO, 8, 1024);                                                                                      \
        } else if (Mpo <= 1024) CG_PSB4(BK, HO, 2, 512);                                                                        \
        else CG_PSB4(BK, HO, 8, 512);                                                                                           \
    } while (0)
#define CG_PSB(BK)                                                                                                              \
    do {                                                                                                                        \
        note_dispatch(smap ? "pool_scatter_bwd_kernel<" #BK "><map>" : "pool_scatter_bwd_kernel<" #BK ">");                     \
        if (out) CG_PSB2(BK, true);                                                                                             \
        else CG_PSB2(BK, false);                                                                                                \
    } while (0)
    if (bias_kind == CHEBGCN_BIAS_VERTEX) CG_PSB(CHEBGCN_BIAS_VERTEX);
    else if (bias_kind == CHEBGCN_BIAS_FILTER) CG_PSB(CHEBGCN_BIAS_FILTER);
    else CG_PSB(CHEBGCN_BIAS_NONE);
#undef CG_PSB2
#undef CG_PSB4
#undef CG_PSB
    CG_HIP(hipGetLastError());
    if (bias_kind == CHEBGCN_BIAS_VERTEX) {
        note_dispatch_more("pool_bias_reduce_kernel<CHEBGCN_BIAS_VERTEX>");
        hipLaunchKernelGGL((pool_bias_reduce_kernel<CHEBGCN_BIAS_VERTEX>), dim3((Mp / 4 + 255) / 256, F), dim3(256), 0, stream, part, dbias,
                           NPB, F, Mp);
    } else if (bias_kind == CHEBGCN_BIAS_FILTER) {
        note_dispatch_more("pool_bias_reduce_kernel<CHEBGCN_BIAS_FILTER>");
        hipLaunchKernelGGL((pool_bias_reduce_kernel<CHEBGCN_BIAS_FILTER>), dim3(1, F), dim3(256), 0, stream, part, dbias, NPB, F, Mp);
    }
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

extern "C" int chebgcn_pool_scatter_bwd(const float* dout, const uint8_t* sel, const int32_t* smap, float* dy, float* dbias,
                                        int bias_kind, int B, int M, int F, int pool, int pool_kind, int relu, void* workspace,
                                        size_t workspace_bytes, chebgcn_stream stream_) {
    CG_REQUIRE(dout && (dy || (dbias && bias_kind != CHEBGCN_BIAS_NONE)), "pool_scatter_bwd: NULL argument");
    CG_REQUIRE(B > 0 && M > 0 && F > 0 && F <= 65535, "pool_scatter_bwd: bad shape");
    CG_REQUIRE(pool >= 2 && (pool & (pool - 1)) == 0 && pool <= 128 && M % pool == 0, "pool_scatter_bwd: bad pool %d", pool);
    CG_REQUIRE(sel || (pool_kind == CHEBGCN_POOL_AVG && !relu), "pool_scatter_bwd: pooling needs the forward's selection bytes");
    CG_REQUIRE(bias_kind == CHEBGCN_BIAS_NONE || dbias, "pool_scatter_bwd: dbias is NULL");
    CG_REQUIRE(pool_scatter_fits(M, pool), "pool_scatter_bwd: a pooled plane of %d vertices does not fit the LDS", M / pool);
    return pool_scatter_launch(dout, nullptr, sel, smap, dy, dbias, bias_kind, B, M, F, pool, pool_kind, relu, workspace,
                               workspace_bytes, (hipStream_t)stream_);
}

extern "C" int chebgcn_relu_grad_bf16(const float* dout, const uint8_t* relu_mask, uint16_t* dy16, float* dbias, int bias_kind,
                                      int B, int M, int F, void* workspace, size_t workspace_bytes, chebgcn_stream stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CG_REQUIRE(dout && relu_mask && dy16, "relu_grad_bf16: NULL argument");
    CG_REQUIRE(B > 0 && M > 0 && F > 0 && F <= 65535, "relu_grad_bf16: bad shape");
    CG_REQUIRE(bias_kind >= 0 && bias_kind <= 2 && (bias_kind == CHEBGCN_BIAS_NONE || dbias), "relu_grad_bf16: dbias is NULL");
    const int Mp = plane_stride(M);
    int parts = 0;
    const int nblk = brelu_bwd_blocks(M, F, 1, 1, true, &parts);
    float* fpart = nullptr;
    if (bias_kind == CHEBGCN_BIAS_FILTER) {
        CG_REQUIRE(workspace && workspace_bytes >= (size_t)F * nblk * sizeof(float),
                   "relu_grad_bf16: the per-filter bias gradient needs a workspace of chebgcn_brelu_pool_bwd_workspace() bytes");
        fpart = static_cast<float*>(workspace);
    }
    const dim3 grid(nblk, F);
#define CG_RG16(BK)                                                                                                         \
    do {                                                                                                                    \
        note_dispatch(parts == 16 ? "bias_grad_relu_kernel<" #BK ",16,bf16>" : "bias_grad_relu_kernel<" #BK ",4,bf16>");    \
        if (parts == 16)                                                                                                    \
            hipLaunchKernelGGL((bias_grad_relu_kernel<BK, 16, true>), grid, dim3(256), 0, stream, dout, relu_mask,          \
                               reinterpret_cast<float*>(dy16), dbias, fpart, B, M, Mp, F, (size_t)F * Mp, (size_t)Mp);      \
        else                                                                                                                \
            hipLaunchKernelGGL((bias_grad_relu_kernel<BK, 4, true>), grid, dim3(256), 0, stream, dout, relu_mask,           \
                               reinterpret_cast<float*>(dy16), dbias, fpart, B, M, Mp, F, (size_t)F * Mp, (size_t)Mp);      \
    } while (0)
    if (bias_kind == CHEBGCN_BIAS_FILTER) CG_RG16(CHEBGCN_BIAS_FILTER);
    else if (bias_kind == CHEBGCN_BIAS_VERTEX) CG_RG16(CHEBGCN_BIAS_VERTEX);
    else CG_RG16(CHEBGCN_BIAS_NONE);
#undef CG_RG16
    CG_HIP(hipGetLastError());
    if (bias_kind == CHEBGCN_BIAS_FILTER) {
        note_dispatch_more("bias_filter_reduce_kernel");
        hipLaunchKernelGGL(bias_filter_reduce_kernel, dim3(F), dim3(64), 0, stream, fpart, dbias, nblk);
        CG_HIP(hipGetLastError());
    }
    return CHEBGCN_OK;
}

// gmean [B][Mp]: the gradient of every filter's output (one plane per window); dy (optional): the gated gradient [B][F][Mp]
static int relu_grad_mean_impl(const float* gmean, const uint8_t* relu_mask, float* dy, float* dbias, int bias_kind, int B, int M,
                               int F, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    CG_REQUIRE(gmean && relu_mask && dbias, "bias_grad_relu_mean: NULL argument");
    CG_REQUIRE(B > 0 && M > 0 && F > 0 && F <= 65535, "bias_grad_relu_mean: bad shape");
    CG_REQUIRE(bias_kind == CHEBGCN_BIAS_FILTER || bias_kind == CHEBGCN_BIAS_VERTEX, "bias_grad_relu_mean: bad bias kind");
    const int Mp = plane_stride(M);
    int parts = 0;
    const int nblk = brelu_bwd_blocks(M, F, 1, 1, true, &parts);
    float* fpart = nullptr;
    if (bias_kind == CHEBGCN_BIAS_FILTER) {
        CG_REQUIRE(workspace && workspace_bytes >= (size_t)F * nblk * sizeof(float),
                   "bias_grad_relu_mean: the per-filter bias gradient needs a workspace of chebgcn_brelu_pool_bwd_workspace() bytes");
        fpart = static_cast<float*>(workspace);
        note_dispatch(parts == 16 ? "bias_grad_relu_kernel<CHEBGCN_BIAS_FILTER,16><mean>" : "bias_grad_relu_kernel<CHEBGCN_BIAS_FILTER,4><mean>");
        note_dispatch_more("bias_filter_reduce_kernel");
        if (parts == 16)
            hipLaunchKernelGGL((bias_grad_relu_kernel<CHEBGCN_BIAS_FILTER, 16>), dim3(nblk, F), dim3(256), 0, stream, gmean, relu_mask,
                               dy, dbias, fpart, B, M, Mp, F, (size_t)Mp, (size_t)0);
        else
            hipLaunchKernelGGL((bias_grad_relu_kernel<CHEBGCN_BIAS_FILTER, 4>), dim3(nblk, F), dim3(256), 0, stream, gmean, relu_mask,
                               dy, dbias, fpart, B, M, Mp, F, (size_t)Mp, (size_t)0);
        hipLaunchKernelGGL(bias_filter_reduce_kernel, dim3(F), dim3(64), 0, stream, fpart, dbias, nblk);
    } else {
        note_dispatch(parts == 16 ? "bias_grad_relu_kernel<CHEBGCN_BIAS_VERTEX,16><mean>" : "bias_grad_relu_kernel<CHEBGCN_BIAS_VERTEX,4><mean>");
        if (parts == 16)
            hipLaunchKernelGGL((bias_grad_relu_kernel<CHEBGCN_BIAS_VERTEX, 16>), dim3(nblk, F), dim3(256), 0, stream, gmean, relu_mask,
                               dy, dbias, fpart, B, M, Mp, F, (size_t)Mp, (size_t)0);
        else
            hipLaunchKernelGGL((bias_grad_relu_kernel<CHEBGCN_BIAS_VERTEX, 4>), dim3(nblk, F), dim3(256), 0, stream, gmean, relu_mask,
                               dy, dbias, fpart, B, M, Mp, F, (size_t)Mp, (size_t)0);
    }
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

extern "C" int chebgcn_bias_grad_relu_mean(const float* gmean, const uint8_t* relu_mask, float* dbias, int bias_kind, int B, int M,
                                           int F, void* workspace, size_t workspace_bytes, chebgcn_stream stream_) {
    return relu_grad_mean_impl(gmean, relu_mask, nullptr, dbias, bias_kind, B, M, F, workspace, workspace_bytes, (hipStream_t)stream_);
}

extern "C" int chebgcn_relu_grad_mean(const float* gmean, const uint8_t* relu_mask, float* dy, float* dbias, int bias_kind, int B,
                                      int M, int F, void* workspace, size_t workspace_bytes, chebgcn_stream stream_) {
    CG_REQUIRE(dy, "relu_grad_mean: dy is NULL");
    return relu_grad_mean_impl(gmean, relu_mask, dy, dbias, bias_kind, B, M, F, workspace, workspace_bytes, (hipStream_t)stream_);
}

extern "C" int chebgcn_perm_data(const float* x, const int32_t* perm, const int32_t* sample, float* out, int S,
                                 int N, int M, int F, chebgcn_stream stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CG_REQUIRE(x && out, "perm_data: NULL argument");
    CG_REQUIRE(S > 0 && N > 0 && M > 0 && F > 0 && S <= 65535, "perm_data: bad shape");
    CG_REQUIRE(perm || M == N, "perm_data: identity permutation needs M == N");
    const int Mp = plane_stride(M);
    const size_t lds = (size_t)F * 65 * sizeof(float);
    CG_REQUIRE(lds <= 64 * 1024, "perm_data: F=%d too large", F);
    dim3 grid((Mp + 63) / 64, S);
    hipLaunchKernelGGL(perm_data_kernel, grid, dim3(256), lds, stream, x, perm, sample, out, N, M, Mp, F);
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

extern "C" int chebgcn_to_plane(const float* x_bmf, float* out_plane, int B, int M, int F, chebgcn_stream stream) {
    return chebgcn_perm_data(x_bmf, nullptr, nullptr, out_plane, B, M, M, F, stream);
}

extern "C" int chebgcn_from_plane(const float* x_plane, float* out_bmf, int B, int M, int F, chebgcn_stream stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CG_REQUIRE(x_plane && out_bmf, "from_plane: NULL argument");
    CG_REQUIRE(B > 0 && M > 0 && F > 0 && B <= 65535, "from_plane: bad shape");
    const int Mp = plane_stride(M);
    const size_t lds = (size_t)F * 65 * sizeof(float);
    CG_REQUIRE(lds <= 64 * 1024, "from_plane: F=%d too large", F);
    dim3 grid((M + 63) / 64, B);
    hipLaunchKernelGGL(from_plane_kernel, grid, dim3(256), lds, stream, x_plane, out_bmf, M, Mp, F);
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

extern "C" int chebgcn_feature_mean_fwd(const float* x, float* y, int B, int M, int F, chebgcn_stream stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CG_REQUIRE(x && y && B > 0 && M > 0 && F > 0 && B <= 65535, "feature_mean_fwd: bad argument");
    dim3 grid((M + 255) / 256, B);
    hipLaunchKernelGGL(feature_mean_fwd_kernel, grid, dim3(256), 0, stream, x, y, M, plane_stride(M), F);
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

extern "C" int chebgcn_feature_mean_bwd(const float* dy, float* dx, int B, int M, int F, chebgcn_stream stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CG_REQUIRE(dy && dx && B > 0 && M > 0 && F > 0 && B <= 65535, "feature_mean_bwd: bad argument");
    const int Mp = plane_stride(M);
    dim3 grid((Mp + 255) / 256, B);
    hipLaunchKernelGGL(feature_mean_bwd_kernel, grid, dim3(256), 0, stream, dy, dx, M, Mp, F);
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

extern "C" int chebgcn_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr_t, float beta1,
                                 float beta2, float eps, float grad_scale, float l2, chebgcn_stream stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CG_REQUIRE(p && g && m && v && n >= 0, "adam_step: bad argument");
    if (n == 0) return CHEBGCN_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p, g, m, v, n, lr_t, beta1, beta2,
                       eps, grad_scale, l2);
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

extern "C" int chebgcn_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* lr_t_dev,
                                     float beta1, float beta2, float eps, float grad_scale, float l2,
                                     chebgcn_stream stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CG_REQUIRE(p && g && m && v && lr_t_dev && n >= 0, "adam_step_dev: bad argument");
    if (n == 0) return CHEBGCN_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adam_dev_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p, g, m, v, n, lr_t_dev, beta1, beta2,
                       eps, grad_scale, l2);
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

extern "C" int chebgcn_adam_partials(int64_t n) {
    int64_t blocks = (n + 255) / 256;
    return (int)(blocks > 4096 ? 4096 : blocks < 0 ? 0 : blocks);
}

extern "C" int chebgcn_adam_step_sq(float* p, const float* g, float* m, float* v, int64_t n, float lr_t, const float* lr_t_dev,
                                    float beta1, float beta2, float eps, float grad_scale, float l2, float* sq_partials,
                                    chebgcn_stream stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CG_REQUIRE(p && g && m && v && sq_partials && n > 0, "adam_step_sq: bad argument");
    note_dispatch("adam_sq_kernel");
    hipLaunchKernelGGL(adam_sq_kernel, dim3((unsigned)chebgcn_adam_partials(n)), dim3(256), 0, stream, p, g, m, v, n, lr_t, lr_t_dev,
                       beta1, beta2, eps, grad_scale, l2, sq_partials, n);
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

extern "C" int chebgcn_adam_step_sq_all(float* p, const float* g, float* m, float* v, int64_t n, int64_t n_reg, float lr_t,
                                        const float* lr_t_dev, float beta1, float beta2, float eps, float grad_scale, float l2,
                                        float* sq_partials, chebgcn_stream stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CG_REQUIRE(p && g && m && v && sq_partials && n > 0 && n_reg >= 0 && n_reg <= n, "adam_step_sq_all: bad argument");
    note_dispatch("adam_sq_kernel<all>");
    hipLaunchKernelGGL(adam_sq_kernel, dim3((unsigned)chebgcn_adam_partials(n)), dim3(256), 0, stream, p, g, m, v, n, lr_t, lr_t_dev,
                       beta1, beta2, eps, grad_scale, l2, sq_partials, n_reg);
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

namespace chebgcn {
__global__ void set_scalars_kernel(float* dst, float v0, float v1) {
    dst[0] = v0;
    dst[1] = v1;
}
}  // namespace chebgcn

extern "C" int chebgcn_set_scalars(float* dst, float v0, float v1, chebgcn_stream stream_) {
    CG_REQUIRE(dst, "set_scalars: NULL argument");
    note_dispatch("set_scalars_kernel");
    hipLaunchKernelGGL(set_scalars_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream_, dst, v0, v1);
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

extern "C" int chebgcn_loss_bookkeeping(const float* cross_entropy, const float* sq_partials, int nparts, float half_reg,
                                        float* ema, float decay, float corr, const float* corr_dev, float* loss_out,
                                        float* loss_average_out, chebgcn_stream stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CG_REQUIRE(cross_entropy && ema && loss_average_out && nparts >= 0 && (nparts == 0 || sq_partials), "loss_bookkeeping: bad argument");
    note_dispatch("loss_bookkeeping_kernel");
    hipLaunchKernelGGL(loss_bookkeeping_kernel, dim3(1), dim3(1024), 0, stream, cross_entropy, sq_partials, nparts, half_reg, ema,
                       decay, corr, corr_dev, loss_out, loss_average_out);
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}
