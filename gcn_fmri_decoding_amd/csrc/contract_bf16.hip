// Dense contraction of the Chebyshev stack on the bf16 matrix cores of gfx950
// (v_mfma_f32_32x32x16_bf16, fp32 accumulate) for wide layers -- BASELINE config 5:
// block_dura = 60, Fout = 256, K = 5, where the fp32-input MFMA of contract.hip is compute
// bound (68 flop/B against a 157 TFLOP/s peak) and the bf16 cores put the op back on the HBM
// roofline.  Same math and epilogue as chebgcn_contract_fwd (lib_new/models_gcn.py:611-648):
//
//       y[b][o][m] = act( sum_{fin,k} W[fin*K+k][o] * stack[k][b][fin][m] + bias ) -> pool over m
//
// passes = 1: both operands rounded to bf16 (RNE)            -> ~2^-9 relative per product;
// passes = 3: each operand split x = hi + lo (two bf16) and  hi*hi + hi*lo + lo*hi accumulated
//             (the lo*lo term, 2^-16 relative, is dropped)   -> ~fp32-grade results at three
//             times the matrix work, which is still below the HBM time of the op.
//
// Mapping.  A workgroup = 128 consecutive vertices of one window x 256 filters; wave w owns
// filters 64w..64w+63 (two 32-row MFMA tiles) for all 128 vertices.  As in contract.hip the
// vertices are the N side: lane c of a half-wave holds vertices 4c..4c+3 in one float4 and
// component r feeds accumulator tile r (vertex <-> MFMA column is a free permutation).  For a
// k-step of 16 reduction rows half-wave g loads rows 8g..8g+7 -- eight 16-byte loads per lane,
// 512 contiguous bytes per half-wave and row -- which is exactly the B fragment layout
// (lane (n, g) holds k = 8g..8g+7 of column n) after an in-register fp32 -> bf16 conversion:
// no LDS staging, no transpose.  The four waves of a workgroup read the same rows at the same
// time (L1 hits); the A operand comes from a bf16 image of W packed once per call in fragment
// order (16 B per lane and tile), resident in L2.
#include "contract_common.h"

namespace chebgcn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// W [FinK][Fout] fp32 -> Wp[part][ks][fo][16] bf16 (part 0 = hi, 1 = lo), zero beyond FinK / Fout
__global__ void __launch_bounds__(256)
pack_w_bf16_kernel(const float* __restrict__ W, __bf16* __restrict__ Wp, int FinK, int Fout, int nks, int FoutP,
                   int parts) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;       // (ks, fo)
    if (idx >= nks * FoutP) return;
    const int ks = idx / FoutP, fo = idx - ks * FoutP;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = ks * 16 + i;
        const float w = (r < FinK && fo < Fout) ? W[(size_t)r * Fout + fo] : 0.f;
        const __bf16 hi = (__bf16)w;
        Wp[(size_t)idx * 16 + i] = hi;
        if (parts > 1) Wp[((size_t)nks * FoutP + idx) * 16 + i] = (__bf16)(w - (float)hi);
    }
}

// k-steps the operands are requested ahead.  The reduction is padded to a multiple of DEPTH
// k-steps (zero weights against the last plane again: an L1/L2 hit), so that the register
// slot of a k-step is static and every iteration issues the same number of loads -- the
// compiler then waits with exact counts (s_waitcnt vmcnt(N)) instead of draining the queue.
constexpr int BF16_DEPTH = 5;

template <int PASSES>
__global__ void __launch_bounds__(256)
contract_fwd_bf16_kernel(FwdArgs a, const bf16x8* __restrict__ Wp, int nks, int FoutP, int ntm, int nitems) {
    constexpr int DEPTH = BF16_DEPTH;
    extern __shared__ size_t rowoff[];                  // [nks*16] plane offset of reduction row r
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, g = lane >> 5;
    for (int r = threadIdx.x; r < nks * 16; r += 256) {
        // rows beyond Fin*K re-read the last plane (finite data) against zero weights
        const int rr = r < a.FinK ? r : a.FinK - 1;
        const int fin = rr / a.K, k = rr - fin * a.K;
        rowoff[r] = (size_t)k * a.slab + (size_t)fin * a.Mp;
    }
    __syncthreads();
    const size_t lo_part = (size_t)nks * FoutP * 2;     // in bf16x8 units

    // work item = (filter group of 256, window, 128 vertices); a workgroup walks its items with
    // the operand pipeline running across item boundaries
    struct Item { int fo0, b, n0; bool valid; const float* base; };
    auto item_of = [&](int it) {
        Item r;
        const int mt = it % ntm, rest = it / ntm;
        r.b = rest % a.B;
        r.fo0 = ((rest / a.B) * 4 + wave) * 64;
        r.n0 = mt * 128 + 4 * c;
        r.valid = r.n0 < a.Mp;
        r.base = a.stack + (size_t)r.b * a.Fin * a.Mp + (r.valid ? r.n0 : 0);     // always a readable address
        return r;
    };

    float4 x[DEPTH][8];
    bf16x8 ah[DEPTH][2], al[DEPTH][2];
    auto request = [&](const float* base, int fo0, int ks, int slot) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[slot][i] = *reinterpret_cast<const float4*>(base + rowoff[ks * 16 + 8 * g + i]);   // shared by the four waves: keep it cacheable
        const int fo = fo0 < a.Fout ? fo0 : 0;          // idle waves (Fout < 256) read tile 0 and store nothing
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const size_t at = ((size_t)ks * FoutP + fo + 32 * t + c) * 2 + g;
            ah[slot][t] = Wp[at];
            if (PASSES == 3) al[slot][t] = Wp[lo_part + at];
        }
    };

    int it = blockIdx.x;
    if (it >= nitems) return;
    Item cur = item_of(it);
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) request(cur.base, cur.fo0, d, d);            // nks >= DEPTH by construction
    for (; it < nitems; it += gridDim.x) {
        const int itn = it + (int)gridDim.x < nitems ? it + (int)gridDim.x : it;   // last item: harmless re-reads
        const Item nxt = item_of(itn);
        f32x16 acc[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[t][r][j] = 0.f;

        for (int ks0 = 0; ks0 < nks; ks0 += DEPTH) {
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                const int ahead = ks0 + u + DEPTH - 1;
                // straight-line (selects, no branch): every iteration issues the same loads
                const bool wrap = ahead >= nks;
                request(wrap ? nxt.base : cur.base, wrap ? nxt.fo0 : cur.fo0, wrap ? ahead - nks : ahead, (u + DEPTH - 1) % DEPTH);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    bf16x8 bh, bl;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float4 xv = x[u][i];
                        const float v = r == 0 ? xv.x : r == 1 ? xv.y : r == 2 ? xv.z : xv.w;
                        bh[i] = (__bf16)v;
                        if (PASSES == 3) bl[i] = (__bf16)(v - (float)bh[i]);
                    }
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        if (PASSES == 3) {
                            acc[t][r] = mfma_bf16(al[u][t], bh, acc[t][r]);        // small terms first
                            acc[t][r] = mfma_bf16(ah[u][t], bl, acc[t][r]);
                        }
                        acc[t][r] = mfma_bf16(ah[u][t], bh, acc[t][r]);
                    }
                }
            }
        }

        if (cur.fo0 < a.Fout) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int fo = cur.fo0 + 32 * t + acc_row(j, g);
                    float v[4] = {acc[t][0][j], acc[t][1][j], acc[t][2][j], acc[t][3][j]};
                    fwd_epilogue_row(a, cur.b, fo, v, cur.n0, cur.valid, c);
                }
        }
        cur = nxt;
    }
}

// k-steps of 16 reduction rows, padded to a multiple of the prefetch depth
static int bf16_ksteps(int FinK) {
    const int nks = (FinK + 15) / 16;
    return (nks + BF16_DEPTH - 1) / BF16_DEPTH * BF16_DEPTH;
}

static bool check_pool_bf16(int pool, int M) {
    return pool >= 1 && pool <= 128 && (pool & (pool - 1)) == 0 && M % pool == 0;
}

}  // namespace chebgcn

using namespace chebgcn;

extern "C" size_t chebgcn_contract_fwd_bf16_workspace(int Fin, int K, int Fout) {
    if (Fin <= 0 || K <= 0 || Fout <= 0) return 0;
    const size_t nks = bf16_ksteps(Fin * K), FoutP = ((size_t)Fout + 255) / 256 * 256;
    return 2 * nks * FoutP * 16 * sizeof(uint16_t);      // hi and lo images
}

extern "C" int chebgcn_contract_fwd_bf16(const float* stack, const float* W, const float* bias, int bias_kind,
                                         float* out, uint8_t* argmax, int B, int M, int Fin, int K, int Fout,
                                         int pool, int pool_kind, int relu, int passes, void* workspace,
                                         size_t workspace_bytes, chebgcn_stream stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CG_REQUIRE(stack && W && out && workspace, "contract_fwd_bf16: NULL argument");
    CG_REQUIRE(B > 0 && M > 0 && Fin > 0 && K > 0 && Fout > 0, "contract_fwd_bf16: bad shape");
    CG_REQUIRE(B <= 65535, "contract_fwd_bf16: B > 65535");
    CG_REQUIRE(passes == 1 || passes == 3, "contract_fwd_bf16: passes must be 1 (bf16) or 3 (split bf16), got %d", passes);
    CG_REQUIRE(check_pool_bf16(pool, M), "contract_fwd_bf16: pool=%d must be a power of two <= 128 dividing M=%d", pool, M);
    CG_REQUIRE(bias_kind == CHEBGCN_BIAS_NONE || bias, "contract_fwd_bf16: bias_kind set but bias is NULL");
    CG_REQUIRE(bias_kind >= 0 && bias_kind <= 2 && (pool_kind == 0 || pool_kind == 1), "contract_fwd_bf16: bad kind");
    CG_REQUIRE(!(pool_kind == CHEBGCN_POOL_AVG && relu && argmax && pool > 8),
               "contract_fwd_bf16: average pooling keeps a ReLU mask only for pool <= 8");
    CG_REQUIRE(workspace_bytes >= chebgcn_contract_fwd_bf16_workspace(Fin, K, Fout),
               "contract_fwd_bf16: workspace of %zu bytes is too small", workspace_bytes);
    FwdArgs a;
    a.stack = stack; a.W = W; a.bias = bias; a.out = out;
    a.argmax = pool > 1 ? argmax : nullptr;
    a.B = B; a.M = M; a.Mp = plane_stride(M); a.Fin = Fin; a.K = K; a.Fout = Fout; a.FinK = Fin * K;
    a.pool = pool; a.pool_kind = pool_kind; a.relu = relu; a.bias_kind = bias_kind;
    a.Mo = M / pool; a.Mpo = plane_stride(a.Mo);
    a.slab = (size_t)B * Fin * a.Mp;
    const int nks = bf16_ksteps(a.FinK), FoutP = (Fout + 255) / 256 * 256;
    hipLaunchKernelGGL(pack_w_bf16_kernel, dim3((nks * FoutP + 255) / 256), dim3(256), 0, stream, W, (__bf16*)workspace,
                       a.FinK, Fout, nks, FoutP, passes == 3 ? 2 : 1);
    CG_HIP(hipGetLastError());
    const int ntm = (M + 127) / 128;
    const int64_t nitems64 = (int64_t)ntm * B * ((Fout + 255) / 256);
    CG_REQUIRE(nitems64 < (1ll << 31), "contract_fwd_bf16: too many tiles");
    const int nitems = (int)nitems64;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const dim3 grid(nitems < cus ? nitems : cus);       // one workgroup per CU (128 accumulator registers per lane)
    const size_t lds = (size_t)nks * 16 * sizeof(size_t);
    CG_REQUIRE(lds <= 64 * 1024, "contract_fwd_bf16: Fin*K = %d too large", a.FinK);
    if (passes == 3)
        hipLaunchKernelGGL(contract_fwd_bf16_kernel<3>, grid, dim3(256), lds, stream, a, (const bf16x8*)workspace, nks, FoutP,
                           ntm, nitems);
    else
        hipLaunchKernelGGL(contract_fwd_bf16_kernel<1>, grid, dim3(256), lds, stream, a, (const bf16x8*)workspace, nks, FoutP,
                           ntm, nitems);
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}
