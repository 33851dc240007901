// Dense contraction of the Chebyshev stack with the learned filter bank on the gfx950
// matrix cores, fused with bias + ReLU + graph pooling, and its two gradients.
//
//   forward (lib_new/models_gcn.py:611-617 + :619-648):
//       y[b][o][m] = act( sum_{fin,k} W[fin*K+k][o] * stack[k][b][fin][m] + bias ) -> pool over m
//   bwd_x:  gstack[k][b][fin][m] = sum_o W[fin*K+k][o] * dy[b][o][m]
//   bwd_w:  dW[fin*K+k][o]       = sum_{b,m} stack[k][b][fin][m] * dy[b][o][m]
//
// All three are f32-in / f32-accumulate MFMA (v_mfma_f32_32x32x2_f32: exact fp32, an
// fmaf chain in k order), so results carry fp32 round-off only.
//
// Why the plane layout makes this cheap: with vertices fastest, the "N" side of the GEMM
// (vertices) is what a wave's lanes index, so the B operand of the MFMA is loaded
// straight from HBM with 16-byte per-lane loads (half-wave = 512 contiguous bytes of one
// plane) -- no LDS staging, no transpose of the [K,M,Fin,N] stack that the reference
// performs (models_gcn.py:612).  A wave owns 128 consecutive vertices: lane c of each
// half-wave holds vertices 4c..4c+3 in the four components of its float4, and component
// r feeds accumulator r (vertex <-> MFMA column is a free permutation).  The A operand
// (a 32 x 2 sliver of W) is 8 B per lane from L1/L2.
#include <algorithm>
#include <cstdlib>
#include "contract_common.h"
#include "bias_grad_body.h"
#ifndef CG_DY_NT
#define CG_DY_NT 0      // dy is read by three kernels of a layer's backward (bias, bwd_w, bwd_x): cached loads, 3.83 -> 3.795 ms per step
#endif
#include <type_traits>

// Waves per SIMD the register allocation is held to (second __launch_bounds__ argument).  Left alone the
// compiler parks the accumulators in AGPRs and spends VGPRs freely: 171 registers for contract_bwd_w_kernel<5>,
// one more than three workgroups per CU allow.  (Holding contract_fwd to four waves or contract_bwd_x to three
// makes them spill: measured slower or equal, left at the compiler's choice.)
#ifndef CG_LB_FWD
#define CG_LB_FWD 1
#endif
#ifndef CG_LB_BWX
#define CG_LB_BWX 1
#endif
#ifndef CG_LB_BWXS
#define CG_LB_BWXS 2      // small launches: two workgroups per CU (three, 168 registers: N=360 34 us against 30)
#endif
// s_setprio around a wave's block of MFMAs: without it the waves of a SIMD that are ready together share the
// matrix pipe round-robin, finish together and then all wait for memory together (a convoy: matrix time and
// memory time add up instead of overlapping)
// Measured at the bench shape: contract_fwd 0.139 -> 0.132 ms, contract_bwd_x 0.147 -> 0.135 ms, step -1.9 %.
#ifndef CG_PRIO
#define CG_PRIO 1
#endif
#if CG_PRIO
#define CG_PRIO_HI() __builtin_amdgcn_s_setprio(3)
#define CG_PRIO_LO() __builtin_amdgcn_s_setprio(0)
#else
#define CG_PRIO_HI() ((void)0)
#define CG_PRIO_LO() ((void)0)
#endif
#ifndef CG_LB_BWW
#define CG_LB_BWW 3          // 48 KB of LDS per workgroup allow three: 0.138 -> 0.120 ms at the bench shape
#endif

namespace chebgcn {

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// --------------------------------------------------------------------------------------
// forward
// --------------------------------------------------------------------------------------
constexpr int FWD_UNROLL = 8;

template <int NT>
__global__ void __launch_bounds__(256, NT == 1 ? CG_LB_FWD : 1)
contract_fwd_kernel(FwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int fo0 = blockIdx.z * (32 * NT);
    const int m0 = (blockIdx.x * 4 + wave) * 128;
    if (m0 >= a.M) return;
    const int n0 = m0 + 4 * c;
    const bool valid = n0 < a.Mp;

    f32x16 acc[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[t][r][j] = 0.f;

    const float* base = a.stack + (size_t)b * a.Fin * a.Mp + (valid ? n0 : 0);
    const int npairs = (a.FinK + 1) >> 1;
    // this lane's reduction index kk = 2*i + h, tracked as (fin, k)
    int fin = h / a.K, k = h % a.K;

    for (int i0 = 0; i0 < npairs; i0 += FWD_UNROLL) {
        float4 bv[FWD_UNROLL];
        float av[FWD_UNROLL][NT];
#pragma unroll
        for (int u = 0; u < FWD_UNROLL; ++u) {
            const int kk = 2 * (i0 + u) + h;
            const bool live = kk < a.FinK;
            // dead iterations re-read the last plane (finite data) against a zero weight
            const int fc = live ? fin : a.Fin - 1, kc = live ? k : a.K - 1;
            const float* p = base + (size_t)kc * a.slab + (size_t)fc * a.Mp;
            bv[u] = valid ? ld_stream(p)
                          : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int fo = fo0 + 32 * t + c;
                av[u][t] = (live && fo < a.Fout) ? a.W[(size_t)kk * a.Fout + fo] : 0.f;
            }
            k += 2;
            if (k >= a.K) { k -= a.K; ++fin; }
            if (k >= a.K) { k -= a.K; ++fin; }
        }
        CG_PRIO_HI();
#pragma unroll
        for (int u = 0; u < FWD_UNROLL; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t][0] = mfma(av[u][t], bv[u].x, acc[t][0]);
                acc[t][1] = mfma(av[u][t], bv[u].y, acc[t][1]);
                acc[t][2] = mfma(av[u][t], bv[u].z, acc[t][2]);
                acc[t][3] = mfma(av[u][t], bv[u].w, acc[t][3]);
            }
        CG_PRIO_LO();
    }

    // ---- epilogue: bias, relu, pool, store ----------------------------------------------
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int fo = fo0 + 32 * t + acc_row(j, h);
            float v[4] = {acc[t][0][j], acc[t][1][j], acc[t][2][j], acc[t][3][j]};
            fwd_epilogue_row(a, b, fo, v, n0, valid, c);
        }
    }
}

// Forward, ring version (Fout <= 32, W small enough for LDS).  The memory system gives the same access pattern with no
// arithmetic 6.6-6.9 TB/s at twelve waves per CU and 4-8 KB in flight each (tools/probes/hbm_stream_probe.hip); the kernel
// above reaches 4.2-4.6: a wave issues eight loads, waits for ALL of them, then runs 32 matrix instructions (2048 cycles)
// with nothing of its own in flight, and every row of its epilogue pays one L2 round trip for its bias.  Here:
//   * a ring of RING operand registers per wave that is refilled as it is consumed (step u: four matrix instructions on
//     slot u, then the load of the pair RING steps ahead into slot u): RING - 1 loads stay in flight THROUGH the matrix work
//     (hipcc turns the in-order uses into s_waitcnt vmcnt(RING - 1));
//   * W and the row offsets (k * slab + fin * Mp, 64 bit) in LDS, filled once per workgroup: the A operand is one
//     conflict-free ds_read_b32, the address of a load one ds_read_b64 + one 64-bit add (the kernel above spent ~25 vector
//     instructions and two branches per load on it); rows beyond Fin*K (padding to whole ring rounds) alias the last
//     row against zero weights, lanes beyond the plane alias vertex 0 and are never stored: no branches in the loop;
//   * the per-vertex bias rows of the epilogue fetched eight at a time into the ring's registers as they fall free.
// Same products in the same order as contract_fwd_kernel<1>: bit-identical results.  Measured (one box, A/B): batch 256
// 0.447 -> 0.385 ms (57.5 -> 67 % of 8 TB/s), batch 64 alone 0.141 -> 0.122 ms, inside the training step 0.114 -> 0.100 ms
// (52 -> 60 %), step 4.07 -> 3.99 ms.  What is left (timing experiments, EXPERIMENTS.md): with a quarter of the matrix
// work the kernel takes 0.372 ms at batch 256 (memory side: 5.5 TB/s for this mix of reads and writes), without the loads
// 0.269 ms; four waves per SIMD (ring of 4, or spilling into 128 registers) and no s_setprio are 2-5 % slower.
#ifndef CG_FWD_RING
#define CG_FWD_RING 8
#endif
constexpr int FILL_U = 8;        // W -> LDS: loads in flight per thread
#if CG_FWD_RING
constexpr int RING = CG_FWD_RING;

#ifndef CG_LB_RING
#define CG_LB_RING 3
#endif
// LEAN: pool == 1, no out_K scatter (the launcher's choice): the row epilogue without the pooling variants
// GATE (LEAN, no bias, no ReLU): the stored result gated by the mask a.gate -- the register slots that prefetch the bias rows
// carry the mask byte of the lane's four vertices instead (FwdArgs::gate)
template <bool LEAN, bool GATE = false>
__global__ void __launch_bounds__(256, CG_LB_RING)
contract_fwd_ring_kernel(FwdArgs a, int nrows_pad) {
    extern __shared__ __align__(16) unsigned char ring_smem[];
    long long* roff = reinterpret_cast<long long*>(ring_smem);                    // [nrows_pad] element offset of row kk
    float* Ws = reinterpret_cast<float*>(ring_smem + (size_t)nrows_pad * 8);       // [nrows_pad][32]
    // (eight loads in flight per thread: written as one load and one LDS store per iteration, hipcc waits for every load
    // before its store -- 20 serial L2 round trips per workgroup at Fin*K = 160)
    for (int i0 = threadIdx.x; i0 < nrows_pad * 32; i0 += 256 * FILL_U) {
        float w[FILL_U];
#pragma unroll
        for (int u = 0; u < FILL_U; ++u) {
            const int idx = i0 + 256 * u, kk = idx >> 5, fo = idx & 31;
            const bool live = kk < a.FinK && fo < a.Fout;
            const float v = a.W[live ? (size_t)kk * a.Fout + fo : 0];       // unconditional load on a clamped address
            w[u] = live ? v : 0.f;
        }
#pragma unroll
        for (int u = 0; u < FILL_U; ++u)
            if (i0 + 256 * u < nrows_pad * 32) Ws[i0 + 256 * u] = w[u];
    }
    for (int kk = threadIdx.x; kk < nrows_pad; kk += 256) {
        const int kc = kk < a.FinK ? kk : a.FinK - 1;
        roff[kk] = (long long)(kc % a.K) * (long long)a.slab + (long long)(kc / a.K) * a.Mp;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int m0 = (blockIdx.x * 4 + wave) * 128;
    if (m0 >= a.M) return;
    const int n0 = m0 + 4 * c;
    const bool valid = n0 < a.Mp;

    f32x16 acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[r][j] = 0.f;

    const float* base = a.stack + (size_t)b * a.Fin * a.Mp + (valid ? n0 : 0);
    const int rounds = nrows_pad / (2 * RING);
    float4 bv[RING];
    float av[RING];
    auto issue = [&](int u, int i) __attribute__((always_inline)) {
        const int kk = 2 * i + h;
        bv[u] = ld_stream(base + roff[kk]);
        av[u] = Ws[kk * 32 + c];
    };
    auto step = [&](int u) __attribute__((always_inline)) {
        CG_PRIO_HI();
        acc[0] = mfma(av[u], bv[u].x, acc[0]);
        acc[1] = mfma(av[u], bv[u].y, acc[1]);
        acc[2] = mfma(av[u], bv[u].z, acc[2]);
        acc[3] = mfma(av[u], bv[u].w, acc[3]);
        CG_PRIO_LO();
    };
#pragma unroll
    for (int u = 0; u < RING; ++u) issue(u, u);
    for (int r = 1; r < rounds; ++r) {
#pragma unroll
        for (int u = 0; u < RING; ++u) {
            step(u);
            issue(u, r * RING + u);
        }
    }
    // last round; with a per-vertex bias, slot u takes the bias of accumulator row j = u as it falls free
    // The bias loads are UNCONDITIONAL: behind a branch hipcc cannot count them and waits for every load in flight
    // (s_waitcnt vmcnt(0)) before each of the last round's matrix blocks and at every row of the epilogue -- sixteen serial
    // round trips at the end of every tile.  One form serves the three kinds: per vertex, the 16 bytes of the lane's four
    // vertices in row fo; per filter, the 16 bytes around bias[fo] (four-byte aligned, inside the array: the launcher sends
    // Fout < 4 to the kernel above) with the component picked afterwards; no bias: the first bytes of the stack, ignored.
    const bool vb = a.bias_kind == CHEBGCN_BIAS_VERTEX, fbk = a.bias_kind == CHEBGCN_BIAS_FILTER;
    const float* bsrc = (vb || fbk) ? a.bias : a.stack;
    const size_t bpitch = vb ? (size_t)a.Mp : fbk ? 1 : 0;
    const int blane = vb ? (valid ? n0 : 0) : 0;
    auto bias_base = [&](int fo) { return fbk ? (fo < a.Fout - 4 ? fo : a.Fout - 4) : (fo < a.Fout ? fo : 0); };
    const size_t mrow = (size_t)(a.Mpo >> 2);
    const uint8_t* gsrc = GATE ? a.gate + (size_t)b * a.Fout * mrow + (valid ? (n0 >> 2) : 0) : nullptr;
    auto bias_row = [&](int j) __attribute__((always_inline)) -> float4 {
        if (GATE) {                                                      // (unconditional, on a clamped row: see above)
            const int fo = acc_row(j, h);
            return make_float4(__int_as_float((int)gsrc[(size_t)(fo < a.Fout ? fo : 0) * mrow]), 0.f, 0.f, 0.f);
        }
        const float* p = bsrc + (size_t)bias_base(acc_row(j, h)) * bpitch + blane;
        typedef f32x4 f32x4_a4 __attribute__((aligned(4)));           // (a cached load: the bias is re-read by every window)
        const f32x4 t = *reinterpret_cast<const f32x4_a4*>(p);
        return make_float4(t.x, t.y, t.z, t.w);
    };
#pragma unroll
    for (int u = 0; u < RING; ++u) {
        step(u);
        bv[u] = bias_row(u);
    }
    FwdArgs ae = a;                                  // the row epilogue below runs without a bias of its own
    ae.bias_kind = CHEBGCN_BIAS_NONE;
    // ---- epilogue: bias, relu, pool, store ----------------------------------------------
    // pool == 1 (every layer of the benchmark network) takes a lean row: fwd_epilogue_row carries every pooling variant
    // behind run-time branches, ~400 instructions per row in the binary
    float ms[4] = {0.f, 0.f, 0.f, 0.f};              // mean_out: sum over this lane's 16 filter rows (after bias + ReLU)
    const float relu_floor = a.relu ? 0.f : -__builtin_inff();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int fo = acc_row(j, h);
        float v[4] = {acc[0][j], acc[1][j], acc[2][j], acc[3][j]};
        if (GATE) {
            const int bits = __float_as_int(bv[j % RING].x);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = ((bits >> r) & 1) ? v[r] : 0.f;
        } else {
            const float4 bb = bv[j % RING];
            const int sel = fo - bias_base(fo);      // per filter: which of the four loaded values is bias[fo]
            const float f = sel == 0 ? bb.x : sel == 1 ? bb.y : sel == 2 ? bb.z : bb.w;
            const bool fo_ok = fo < a.Fout;
            v[0] += vb ? bb.x : (fbk && fo_ok) ? f : 0.f;
            v[1] += vb ? bb.y : (fbk && fo_ok) ? f : 0.f;
            v[2] += vb ? bb.z : (fbk && fo_ok) ? f : 0.f;
            v[3] += vb ? bb.w : (fbk && fo_ok) ? f : 0.f;
        }
        if (LEAN) {
            if (!GATE) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], relu_floor);
            }
            if (fo < a.Fout && valid) {
                const size_t row = (size_t)b * a.Fout + fo;
                if (a.out) *reinterpret_cast<float4*>(a.out + row * a.Mpo + n0) = make_float4(v[0], v[1], v[2], v[3]);
                if (a.relu_mask)
                    a.relu_mask[row * mrow + (n0 >> 2)] =
                        (uint8_t)((v[0] > 0.f ? 1 : 0) | (v[1] > 0.f ? 2 : 0) | (v[2] > 0.f ? 4 : 0) | (v[3] > 0.f ? 8 : 0));
            }
        } else {
            fwd_epilogue_row(ae, b, fo, v, n0, valid, c);
        }
        if (j + RING < 16) bv[j % RING] = bias_row(j + RING);
        if (fo < a.Fout) { ms[0] += v[0]; ms[1] += v[1]; ms[2] += v[2]; ms[3] += v[3]; }
    }
    if (a.mean_out) {                                // + the 16 rows of the other half-wave, / Fout
#pragma unroll
        for (int r = 0; r < 4; ++r) ms[r] = (ms[r] + __shfl_xor(ms[r], 32)) / (float)a.Fout;
        if (h == 0 && valid) *reinterpret_cast<float4*>(a.mean_out + (size_t)b * a.Mp + n0) = make_float4(ms[0], ms[1], ms[2], ms[3]);
    }
}
#endif

// Small launches (atlas-sized graphs: the reference's own 246..1000-node atlases give one to nine 128-vertex tiles per
// window): one wave per tile leaves most of the chip without work -- at M = 380, batch 128 the kernel above runs 384 waves
// on 256 CUs, each through the whole reduction (0.049 ms for 68 MB).  Here the four waves of a workgroup share ONE tile and
// take every fourth pair of reduction rows each; the four partial accumulators are added through LDS in a fixed order (wave
// 0, 1, 2, 3), and every wave finishes a quarter of the filter rows (bias, ReLU, pooling, store).  Four times the waves,
// a quarter of the chain per wave.  (The sum is the same products in four chains instead of one: fp32 round-off differs
// from the big-launch kernel in the last bits, deterministic for a given shape.)
// (round 6: the reduction in two halves of the accumulator rows through 32 KB instead of 64 -- three workgroups per CU instead of
// two; mid-sized launches, N = 1000 at batch 128: 1152 workgroups, ran in 2.25 rounds of 512 resident ones.  Same sums in the
// same order.)
#ifndef CG_LB_SPLITK
#define CG_LB_SPLITK 3
#endif
__global__ void __launch_bounds__(256, CG_LB_SPLITK)
contract_fwd_splitk_kernel(FwdArgs a) {
    __shared__ float red[4][32][64];                   // [wave][accumulator register r*8 + (j & 7)][lane], rows j < 8 then j >= 8
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int m0 = blockIdx.x * 128;
    const int n0 = m0 + 4 * c;
    const bool valid = n0 < a.Mp;

    f32x16 acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[r][j] = 0.f;
    const float* base = a.stack + (size_t)b * a.Fin * a.Mp + (valid ? n0 : 0);
    const int npairs = (a.FinK + 1) >> 1;
    const int fo = c < a.Fout ? c : a.Fout - 1;
    const float wz = c < a.Fout ? 1.f : 0.f;
    // A ring of FWD_UNROLL operand registers refilled as it is consumed (as in contract_fwd_ring_kernel): this wave's pairs
    // are wave, wave + 4, ...; every load is unconditional on a clamped address (a pair beyond Fin*K re-reads the last
    // plane against a zero weight), so hipcc counts the loads in flight instead of waiting for all of them.
    float4 bv[FWD_UNROLL];
    float av[FWD_UNROLL], az[FWD_UNROLL];              // the weight as loaded, and 1 / 0 (applied at the use: no wait at the issue)
    auto issue = [&](int u, int i) __attribute__((always_inline)) {
        const int kk = 2 * i + h;
        const bool live = kk < a.FinK;
        const int kkc = live ? kk : a.FinK - 1;
        const int fc = kkc / a.K, kc = kkc - fc * a.K;
        bv[u] = ld_stream(base + (size_t)kc * a.slab + (size_t)fc * a.Mp);
        av[u] = a.W[(size_t)kkc * a.Fout + fo];
        az[u] = live ? wz : 0.f;
    };
    const int nmine = npairs > wave ? (npairs - wave + 3) >> 2 : 0;            // pairs of this wave
    const int nring = (nmine + FWD_UNROLL - 1) / FWD_UNROLL * FWD_UNROLL;      // whole ring rounds (the padding multiplies by zero)
#pragma unroll
    for (int u = 0; u < FWD_UNROLL; ++u) issue(u, wave + 4 * u);
    for (int r0 = FWD_UNROLL; r0 < nring; r0 += FWD_UNROLL) {      // (the refill is unconditional inside a round: countable)
#pragma unroll
        for (int u = 0; u < FWD_UNROLL; ++u) {
            const float w = av[u] * az[u];
            acc[0] = mfma(w, bv[u].x, acc[0]);
            acc[1] = mfma(w, bv[u].y, acc[1]);
            acc[2] = mfma(w, bv[u].z, acc[2]);
            acc[3] = mfma(w, bv[u].w, acc[3]);
            issue(u, wave + 4 * (r0 + u));
        }
    }
    if (nring > 0) {                                               // last round: nothing to refill
#pragma unroll
        for (int u = 0; u < FWD_UNROLL; ++u) {
            const float w = av[u] * az[u];
            acc[0] = mfma(w, bv[u].x, acc[0]);
            acc[1] = mfma(w, bv[u].y, acc[1]);
            acc[2] = mfma(w, bv[u].z, acc[2]);
            acc[3] = mfma(w, bv[u].w, acc[3]);
        }
    }
    // wave w finishes the accumulator registers j = 4w .. 4w+3 (filter rows acc_row(j, h)): waves 0, 1 in the first half, 2, 3 in the second
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half) __syncthreads();                     // the first half's sums are read
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j8 = 0; j8 < 8; ++j8) red[wave][r * 8 + j8][lane] = acc[r][8 * half + j8];
        __syncthreads();
        if ((wave >> 1) == half) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j8 = 4 * (wave & 1) + jj;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    v[r] = ((red[0][r * 8 + j8][lane] + red[1][r * 8 + j8][lane]) + red[2][r * 8 + j8][lane]) + red[3][r * 8 + j8][lane];
                fwd_epilogue_row(a, b, acc_row(8 * half + j8, h), v, n0, valid, c);
            }
        }
    }
}

// --------------------------------------------------------------------------------------
// bwd_x:  D[kk][m] = sum_o W[kk][o] dy[o][m]
// --------------------------------------------------------------------------------------
struct BwdXArgs {
    const float* dy; const float* W; float* gstack;
    const uint8_t* mask;         // MASK: dy = the gradient of the layer OUTPUT, gated by the ReLU mask of the forward
    int B, M, Mp, Fin, K, Fout, FinK;
    size_t slab;
    // element strides of dy between windows and filters: Fout*Mp and Mp, or Mp and 0 where every filter takes the same plane
    // (the gradient of a filter mean, chebgcn_contract_bwd_x_relu_mean)
    size_t dy_bstride, dy_fstride;
};

// HOLD: dy tile (Fout <= 32 -> 16 float4 per lane) stays in registers across the row tiles.
// SPLIT (small launches, see contract_fwd_splitk_kernel): the four waves of a workgroup share one 128-vertex tile and take
// every fourth tile of 32 output rows each -- no reduction involved, the dy tile is loaded by each of them.
template <bool HOLD, bool MASK, bool SPLIT = false>
__global__ void __launch_bounds__(256, SPLIT ? CG_LB_BWXS : CG_LB_BWX)
contract_bwd_x_kernel(BwdXArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int m0 = SPLIT ? blockIdx.x * 128 : (blockIdx.x * 4 + wave) * 128;
    if (m0 >= a.M) return;
    const int n0 = m0 + 4 * c;
    const bool valid = n0 < a.Mp;
    // every load below is unconditional on a clamped address and masked afterwards: a conditional load is a
    // branch with a full s_waitcnt behind it, i.e. one exposed memory round trip per row
    const float* dyb = a.dy + (size_t)b * a.dy_bstride + (valid ? n0 : 0);
    const uint8_t* mkb = MASK ? a.mask + (size_t)b * a.Fout * (a.Mp >> 2) + (valid ? (n0 >> 2) : 0) : nullptr;
    const int Mq = a.Mp >> 2;
    auto gated = [&](float4 v, int bits) {       // ReluGrad: zero where the forward result was not positive
        return make_float4((bits & 1) ? v.x : 0.f, (bits & 2) ? v.y : 0.f, (bits & 4) ? v.z : 0.f, (bits & 8) ? v.w : 0.f);
    };
    const int nfo2 = (a.Fout + 1) >> 1;
    const int ntiles = (a.FinK + 31) >> 5;

    float4 hold[HOLD ? 16 : 1];
    if (HOLD) {
        int bits[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int fo = 2 * j + h, foc = fo < a.Fout ? fo : 0;
            hold[j] = ld_stream(dyb + (size_t)foc * a.dy_fstride);
            bits[j] = MASK ? (int)mkb[(size_t)foc * Mq] : 15;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) hold[j] = gated(hold[j], (valid && 2 * j + h < a.Fout) ? bits[j] : 0);
    }
    for (int t = SPLIT ? wave : 0; t < ntiles; t += SPLIT ? 4 : 1) {
        f32x16 acc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[r][j] = 0.f;
        const int kkA = t * 32 + c;                       // A row handled by this lane
        const float* wrow = a.W + (size_t)(kkA < a.FinK ? kkA : 0) * a.Fout;
        if (HOLD) {
            float av[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int fo = 2 * j + h;
                av[j] = wrow[fo < a.Fout ? fo : 0];
            }
            CG_PRIO_HI();
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float w = (kkA < a.FinK && 2 * j + h < a.Fout) ? av[j] : 0.f;
                acc[0] = mfma(w, hold[j].x, acc[0]);
                acc[1] = mfma(w, hold[j].y, acc[1]);
                acc[2] = mfma(w, hold[j].z, acc[2]);
                acc[3] = mfma(w, hold[j].w, acc[3]);
            }
            CG_PRIO_LO();
        } else {
            for (int j0 = 0; j0 < nfo2; j0 += 4) {
                float av[4];
                float4 bv[4];
                int bits[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int fo = 2 * (j0 + u) + h, foc = fo < a.Fout ? fo : 0;
                    av[u] = wrow[foc];
                    bv[u] = *reinterpret_cast<const float4*>(dyb + (size_t)foc * a.dy_fstride);
                    bits[u] = MASK ? (int)mkb[(size_t)foc * Mq] : 15;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool live = 2 * (j0 + u) + h < a.Fout;
                    const float w = (kkA < a.FinK && live) ? av[u] : 0.f;
                    const float4 g = gated(bv[u], (valid && live) ? bits[u] : 0);
                    acc[0] = mfma(w, g.x, acc[0]);
                    acc[1] = mfma(w, g.y, acc[1]);
                    acc[2] = mfma(w, g.z, acc[2]);
                    acc[3] = mfma(w, g.w, acc[3]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int kk = t * 32 + acc_row(j, h);
            if (kk < a.FinK && valid) {
                const int fin = kk / a.K, k = kk - fin * a.K;
                float* dst = a.gstack + (size_t)k * a.slab + ((size_t)b * a.Fin + fin) * a.Mp + n0;
                *reinterpret_cast<float4*>(dst) = make_float4(acc[0][j], acc[1][j], acc[2][j], acc[3][j]);
            }
        }
    }
}

// bwd_x, big launches with Fout <= 32: matrix work and stores interleaved inside every wave.
// What the kernel above does per tile of 32 output rows: 16 four-byte loads of W per lane behind the previous tile's 16 KB
// of stores in the CU's in-order memory pipeline, 64 matrix instructions (4096 cycles) with none of the wave's own memory
// traffic in flight, then 16 stores that each wait for ALL FOUR accumulators (the four vertices of a lane's float4 live in
// four accumulators) and compute kk / K, kk % K and a 64-bit address.  Timing experiments (EXPERIMENTS.md): the matrix work
// alone 0.31 ms, the stores alone 0.32 ms, together 0.47 ms at batch 256 -- they add up more than they overlap.  Here:
//   * vertex <-> accumulator mapping: accumulator r holds vertices m0 + 32 r + c (not 4 c + r), so one accumulator register
//     is one 4-byte store of a full 128-byte line per half-wave, straight from the accumulator file, and accumulator r is
//     finished and storable on its own;
//   * two phases per row tile: the 32 matrix instructions of accumulators 0, 1 run while accumulators 2, 3 of the previous
//     tile are stored, then 2, 3 run while 0, 1 of this tile are stored: every wave keeps the matrix pipe and the store
//     stream busy at the same time (the first instruction of a chain takes C = 0: no zeroing);
//   * W^T and the row offsets (k * slab + fin * Mp, 64 bit) in LDS: the A operand is a conflict-free ds_read_b32, the
//     address of a row one ds_read_b64 + one add; the memory pipeline carries nothing but the stores.
// Every output element is the same chain of products in the same order as above: bit-identical results.
#ifndef CG_BWX_LDS
#define CG_BWX_LDS 1
#endif
#ifndef CG_LB_BWXL
#define CG_LB_BWXL 1
#endif
#if CG_BWX_LDS
template <bool MASK>
__global__ void __launch_bounds__(256, CG_LB_BWXL)
contract_bwd_x_lds_kernel(BwdXArgs a, int nrows32) {
    extern __shared__ __align__(16) unsigned char bwx_smem[];
    long long* roff = reinterpret_cast<long long*>(bwx_smem);                     // [nrows32] element offset of stack row kk
    float* Wt = reinterpret_cast<float*>(bwx_smem + (size_t)nrows32 * 8);          // [32][nrows32]: W^T, zero beyond Fin*K / Fout
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int m0 = (blockIdx.x * 4 + wave) * 128;
    const int Mq = a.Mp >> 2;
    const bool wave_live = m0 < a.M;
    bool ok[4];                                                                  // uniform: the 32 vertices of accumulator r lie inside the plane
#pragma unroll
    for (int r = 0; r < 4; ++r) ok[r] = wave_live && m0 + 32 * r < a.Mp;
    // the dy tile first (its latency runs under the LDS fill); unconditional loads on clamped addresses, masked afterwards
    const float* dyb = a.dy + (size_t)b * a.dy_bstride + (wave_live ? m0 : 0) + c;
    // ReLU mask of row fo: one byte per four vertices, 32 bytes = 8 dwords for the 128 vertices of the tile.  Lane c
    // fetches dword c & 7 of its row (one load per row); the bit of vertex 32 r + c sits in byte 8 r + (c >> 2), i.e. in the
    // dword lane 2 r + (c >> 4) of the same half-wave holds: one ds_bpermute per accumulator instead of a byte load.
    // A mask row is Mp / 4 bytes (planes are padded to 32 vertices, not to the 128 of a tile): in the last tile of a window the
    // dwords beyond the row alias dword 0 (their accumulators are dropped by ok[r]); nothing is read past the allocation.
    const int mdw = (wave_live && m0 + 16 * (c & 7) < a.Mp) ? (c & 7) : 0;
    const uint8_t* mkb = MASK ? a.mask + (size_t)b * a.Fout * Mq + ((wave_live ? m0 : 0) >> 2) + 4 * mdw : nullptr;
    float hold[16][4];
    int mword[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int fo = 2 * j + h, foc = fo < a.Fout ? fo : 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) hold[j][r] = CG_DY_NT ? __builtin_nontemporal_load(dyb + (size_t)foc * a.dy_fstride + (ok[r] ? 32 * r : 0))
                                                         : dyb[(size_t)foc * a.dy_fstride + (ok[r] ? 32 * r : 0)];
        mword[j] = MASK ? *reinterpret_cast<const int*>(mkb + (size_t)foc * Mq) : -1;
    }
    for (int i0 = threadIdx.x; i0 < nrows32 * 32; i0 += 256 * FILL_U) {     // (eight loads in flight: see contract_fwd_ring_kernel)
        float w[FILL_U];
#pragma unroll
        for (int u = 0; u < FILL_U; ++u) {
            const int idx = i0 + 256 * u, kk = idx >> 5, fo = idx & 31;
            const bool live = kk < a.FinK && fo < a.Fout;
            const float v = a.W[live ? (size_t)kk * a.Fout + fo : 0];
            w[u] = live ? v : 0.f;
        }
#pragma unroll
        for (int u = 0; u < FILL_U; ++u) {
            const int idx = i0 + 256 * u;
            if (idx < nrows32 * 32) Wt[(idx & 31) * nrows32 + (idx >> 5)] = w[u];
        }
    }
    for (int kk = threadIdx.x; kk < nrows32; kk += 256) {
        const int kc = kk < a.FinK ? kk : a.FinK - 1;
        roff[kk] = (long long)(kc % a.K) * (long long)a.slab + (long long)(kc / a.K) * a.Mp;
    }
    __syncthreads();
    if (!wave_live) return;
    const int bit_sh = 8 * ((c >> 2) & 3) + (c & 3);
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {                                            // ReluGrad: zero where the forward result was not positive
            const int word = MASK ? __builtin_amdgcn_ds_bpermute(4 * (2 * r + (c >> 4) + 32 * h), mword[j]) : -1;
            const bool on = ok[r] && 2 * j + h < a.Fout && ((word >> bit_sh) & 1);
            hold[j][r] = on ? hold[j][r] : 0.f;
        }
    float* gbase = a.gstack + (size_t)b * a.Fin * a.Mp + m0 + c;
    const int ntiles = nrows32 >> 5;                 // whole tiles: the launcher sends Fin*K % 32 != 0 to the kernel above
    f32x16 acc[4];
    f32x16 zero16;
#pragma unroll
    for (int q = 0; q < 16; ++q) zero16[q] = 0.f;
    const float* wt0 = Wt + h * nrows32 + c;
    const long long* ro0 = roff + 4 * h;
    // One phase: (MM) the matrix instructions of accumulators R0, R0 + 1 of row tile t, interleaved with (ST) the stores of
    // accumulators S0, S0 + 1 of row tile ts.  The A value and the row offset of step j + 1 are read from LDS during step j.
    // ALL: the four accumulators lie inside the plane (not so only in the last vertex tile of a window, uniform per wave).
    auto phase = [&](auto R0_, auto S0_, auto MM_, auto ST_, auto ALL_, int t, int ts) __attribute__((always_inline)) {
        constexpr int R0 = decltype(R0_)::value, S0 = decltype(S0_)::value;
        constexpr bool MM = decltype(MM_)::value, ST = decltype(ST_)::value, ALL = decltype(ALL_)::value;
        const float* wt = wt0 + t * 32;
        const long long* ro = ro0 + ts * 32;
        float w = MM ? wt[0] : 0.f;
        long long off = ST ? ro[0] : 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float wn = (MM && j < 15) ? wt[2 * (j + 1) * nrows32] : 0.f;
            const long long offn = (ST && j < 15) ? ro[((j + 1) & 3) + 8 * ((j + 1) >> 2)] : 0;
            if (MM) {
                acc[R0] = mfma(w, hold[j][R0], j == 0 ? zero16 : acc[R0]);
                acc[R0 + 1] = mfma(w, hold[j][R0 + 1], j == 0 ? zero16 : acc[R0 + 1]);
            }
            if (ST) {
                float* dst = gbase + off;
                if (ALL || ok[S0]) dst[32 * S0] = acc[S0][j];
                if (ALL || ok[S0 + 1]) dst[32 * (S0 + 1)] = acc[S0 + 1][j];
            }
            w = wn;
            off = offn;
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I2 = std::integral_constant<int, 2>;
    using Y = std::true_type;
    using N = std::false_type;
    auto run = [&](auto ALL_) __attribute__((always_inline)) {
        phase(I0{}, I2{}, Y{}, N{}, ALL_, 0, 0);                 // accumulators 0, 1 of tile 0
        for (int t = 0; t < ntiles - 1; ++t) {
            phase(I2{}, I0{}, Y{}, Y{}, ALL_, t, t);             // 2, 3 of tile t      | stores of 0, 1 of tile t
            phase(I0{}, I2{}, Y{}, Y{}, ALL_, t + 1, t);         // 0, 1 of tile t + 1  | stores of 2, 3 of tile t
        }
        phase(I2{}, I0{}, Y{}, Y{}, ALL_, ntiles - 1, ntiles - 1);
        phase(I0{}, I2{}, N{}, Y{}, ALL_, 0, ntiles - 1);        // stores of 2, 3 of the last tile
    };
    if (ok[3]) run(Y{});
    else run(N{});
}
#endif

// --------------------------------------------------------------------------------------
// bwd_w:  D[kk][o] = sum_{b,m} stack[kk][b,m] dy[o][b,m]
// Both operands are read along their planes: lane (i, h) owns plane i of its tile and
// pulls 16-byte pieces at vertex m0 + 8q + 4h; the pairing of vertices inside one MFMA
// is irrelevant for a reduction.  A wave accumulates RT row tiles x 1 column tile over a
// strided set of (window, 64-vertex) chunks; the four waves of a workgroup are reduced
// through LDS and every workgroup leaves one partial; reduce_partials sums them in a
// fixed order (deterministic).
// --------------------------------------------------------------------------------------
struct BwdWArgs {
    const float* stack; const float* dy; float* partial;
    const uint8_t* mask;         // MASK: see BwdXArgs
    int B, M, Mp, Fin, K, Fout, FinK;
    int nchunks_m;               // ceil(M / 64)
    int ntiles;                  // ceil(FinK / 32)
    size_t slab;
    size_t dy_bstride, dy_fstride;   // see BwdXArgs
};

// One chunk = 64 consecutive vertices of one window.  Its operand rows -- RT*32 stack planes
// and 32 dy planes, 256 B each -- are brought into LDS by LDS-DMA (global_load_lds, 16 B per
// lane: every wave instruction moves four full rows, fully coalesced, no VGPR round trip).
// LDS image: row pitch 256 B, the 16-byte piece p of row r sits at position p ^ (r & 15); the
// DMA writes linearly, so the swizzle is applied to the SOURCE address (cdna guide rule 21),
// and the MFMA operand reads (lane = row, ds_read_b128) are bank-conflict free.
// A wave owns 16 of the 64 vertices for all RT row tiles; lane (i, h) feeds the MFMA with the
// float4 pieces 2q+h (q = 0, 1) of its row -- the pairing of vertices inside one MFMA is
// irrelevant for a reduction.  The four waves are reduced through LDS in a fixed order and
// every workgroup leaves one partial; reduce_partials sums them deterministically.
constexpr int BW_ROW = 64;       // floats per LDS row (one chunk)

template <int RT, bool MASK>
__global__ void __launch_bounds__(256, CG_LB_BWW)
contract_bwd_w_kernel(BwdWArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [(RT+1)*32][64]
    constexpr int NROWS = (RT + 1) * 32;
    constexpr int NDMA = NROWS / 4;                     // wave instructions per chunk (4 rows each)
    constexpr int PER_WAVE = NDMA / 4;                  // 2 (RT + 1), exact
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int tile0 = blockIdx.y * RT;                  // first row tile of this group
    const int fo0 = blockIdx.z * 32;                    // column tile

    f32x16 acc[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;

    // rows this lane fetches: DMA instruction n = wave + 4u covers rows 4n..4n+3, lane l -> row 4n + l/16,
    // LDS piece l%16, source piece (l%16) ^ (row%16) -- the same for every u because 16 divides the row step.
    // One pointer per instruction (row base + source piece) is all the bookkeeping: the register budget of this
    // kernel decides between two and three workgroups per CU.
    const int rsw = (lane & 15) ^ ((4 * wave + (lane >> 4)) & 15);
    const float* rsrc[PER_WAVE];
#pragma unroll
    for (int u = 0; u < PER_WAVE; ++u) {
        const int n = wave + 4 * u;
        const int row = 4 * n + (lane >> 4);
        if (n < RT * 8) {
            int kk = tile0 * 32 + row;
            if (kk >= a.FinK) kk = 0;                   // masked later (A rows beyond Fin*K)
            const int fin = kk / a.K, k = kk - fin * a.K;
            rsrc[u] = a.stack + (size_t)k * a.slab + (size_t)fin * a.Mp + 4 * rsw;
        } else {
            int fo = fo0 + (row - RT * 32);
            if (fo >= a.Fout) fo = 0;
            rsrc[u] = a.dy + (size_t)fo * a.dy_fstride + 4 * rsw;
        }
    }
    bool a_ok[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) a_ok[t] = (tile0 + t) * 32 + c < a.FinK;
    const bool b_ok = fo0 + c < a.Fout;

    const int total = a.B * a.nchunks_m;
    for (int ch = blockIdx.x; ch < total; ch += gridDim.x) {
        const int b = ch / a.nchunks_m;
        const int m0 = (ch - b * a.nchunks_m) * 64;
        __syncthreads();                                // previous chunk's operand reads are done
        int gate[2] = {15, 15};
        if (MASK) {                                     // the two mask bytes of this lane's pieces, in flight beside the DMA
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int quad = (m0 >> 2) + 4 * wave + 2 * q + h;
                gate[q] = (b_ok && quad < (a.Mp >> 2)) ? a.mask[((size_t)b * a.Fout + fo0 + c) * (a.Mp >> 2) + quad] : 0;
            }
        }
        const ptrdiff_t mo = (m0 + 4 * rsw < a.Mp) ? m0 : -4 * rsw;   // beyond the plane: any valid address, masked below
        const ptrdiff_t so = (ptrdiff_t)b * a.Fin * a.Mp + mo, dof = (ptrdiff_t)b * (ptrdiff_t)a.dy_bstride + mo;
#pragma unroll
        for (int u = 0; u < PER_WAVE; ++u) {
            const int n = wave + 4 * u;
            const float* src = rsrc[u] + (n < RT * 8 ? so : dof);
            if (u >= 2 * RT && !CG_DY_NT) __builtin_amdgcn_global_load_lds(src, lds + (size_t)n * 4 * BW_ROW, 16, 0, 0);   // dy rows: cached
            else __builtin_amdgcn_global_load_lds(src, lds + (size_t)n * 4 * BW_ROW, 16, 0, 2);   // aux 2 = nt: every chunk is read once
        }
        __syncthreads();                                // DMA landed (the barrier's release waits vmcnt(0))

        // operand pieces of this lane: row = its plane, pieces 4*wave + 2q + h
        float4 bv[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int piece = 4 * wave + 2 * q + h;
            const int row = RT * 32 + c;
            float4 v = *reinterpret_cast<const float4*>(lds + row * BW_ROW + 4 * (piece ^ (row & 15)));
            const int n = m0 + 4 * piece;
            v.x = (b_ok && n + 0 < a.M && (gate[q] & 1)) ? v.x : 0.f;
            v.y = (b_ok && n + 1 < a.M && (gate[q] & 2)) ? v.y : 0.f;
            v.z = (b_ok && n + 2 < a.M && (gate[q] & 4)) ? v.z : 0.f;
            v.w = (b_ok && n + 3 < a.M && (gate[q] & 8)) ? v.w : 0.f;
            bv[q] = v;
        }
        CG_PRIO_HI();
#pragma unroll
        for (int t = 0; t < RT; ++t) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int piece = 4 * wave + 2 * q + h;
                const int row = t * 32 + c;
                float4 v = *reinterpret_cast<const float4*>(lds + row * BW_ROW + 4 * (piece ^ (row & 15)));
                const int n = m0 + 4 * piece;
                v.x = (a_ok[t] && n + 0 < a.M) ? v.x : 0.f;
                v.y = (a_ok[t] && n + 1 < a.M) ? v.y : 0.f;
                v.z = (a_ok[t] && n + 2 < a.M) ? v.z : 0.f;
                v.w = (a_ok[t] && n + 3 < a.M) ? v.w : 0.f;
                acc[t] = mfma(v.x, bv[q].x, acc[t]);
                acc[t] = mfma(v.y, bv[q].y, acc[t]);
                acc[t] = mfma(v.z, bv[q].z, acc[t]);
                acc[t] = mfma(v.w, bv[q].w, acc[t]);
            }
        }
        CG_PRIO_LO();
    }

    // ---- workgroup reduction in LDS (fixed order: wave 0, then +1, +2, +3) ----------------
    constexpr int per = RT * 16 * 64;
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    float* p = lds + (t * 16 + j) * 64 + lane;
                    *p = (w == 0) ? acc[t][j] : *p + acc[t][j];
                }
        }
    }
    __syncthreads();
    float* dst = a.partial + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * per;
    for (int o = threadIdx.x; o < per; o += 256) dst[o] = lds[o];
}

// partial: [Z][Y][X][RT*16*64] raw accumulator images, summed in a fixed order by ONE launch:
// Both stages in one launch for few partials (small launches: an atlas-sized layer leaves 192): block (row, y, z), four
// thread groups take every fourth partial (eight in flight), fixed-order LDS sum, scatter to dW[kk][o].  One launch and
// one pass instead of two launches with a round trip through `stage` (6.5 + 4.8 us + a launch gap at N = 360).
__device__ __forceinline__ void reduce_partials_small_body(const float* __restrict__ partial, float* __restrict__ dW, int nx, int ny,
                                                           int rt, int FinK, int Fout, int row, int y, int z) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int per = rt * 16 * 64;
    const float* base = partial + ((size_t)z * ny + y) * nx * per + (size_t)row * 64 + lane;
    float s = 0.f;
    for (int x0 = part; x0 < nx; x0 += 8 * 4) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int x = x0 + 4 * u;
            v[u] = base[(size_t)(x < nx ? x : x0) * per];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (x0 + 4 * u < nx) s += v[u];
    }
    red[part][lane] = s;
    __syncthreads();
    if (part == 0) {
        const float t = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
        const int tt = row >> 4, j = row & 15, h = lane >> 5;
        const int kk = (y * rt + tt) * 32 + acc_row(j, h);
        const int fo = z * 32 + (lane & 31);
        if (kk < FinK && fo < Fout) dW[(size_t)kk * Fout + fo] = t;
    }
}

__global__ void __launch_bounds__(256)
reduce_partials_small(const float* __restrict__ partial, float* __restrict__ dW, int nx, int ny, int rt, int FinK, int Fout) {
    reduce_partials_small_body(partial, dW, nx, ny, rt, FinK, Fout, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

// The same launch also reducing the per-vertex bias gradient of the layer (round 6): an atlas-sized layer's backward is a chain of
// launches of about 5 us each, and the bias reduction (bias_grad_relu_kernel<VERTEX,16>: 192 workgroups at N = 360) was one of them
// -- its workgroups ride behind the 160 of this reduction: blocks [0, nred) add partials, blocks [nred, nred + nbx*F) are the bias
// kernel's (bx, f) jobs.  Same code, same sums, same order as the two launches.
__global__ void __launch_bounds__(256)
reduce_partials_small_bias_kernel(const float* __restrict__ partial, float* __restrict__ dW, int nx, int ny, int nz, int rt, int FinK,
                                  int Fout, const float* __restrict__ dout, const uint8_t* __restrict__ mask, float* __restrict__ dbias,
                                  int B, int M, int Mp, int nbx) {
    const int nrow = rt * 16, nred = nrow * ny * nz;
    const int id = blockIdx.x;                        // (uniform: every thread of a block takes the same arm)
    if (id < nred) {
        const int row = id % nrow, yz = id / nrow;
        reduce_partials_small_body(partial, dW, nx, ny, rt, FinK, Fout, row, yz % ny, yz / ny);
    } else {
        const int j = id - nred;
        bias_grad_relu_body<CHEBGCN_BIAS_VERTEX, 16, false, true>(dout, mask, nullptr, dbias, nullptr, B, M, Mp, Fout,
                                                                   (size_t)Fout * Mp, (size_t)Mp, j % nbx, j / nbx, nbx);
    }
}

// Many partials (big launches: 768 workgroups), ONE launch: block (row, y, z) of 1024 threads = 16 thread groups that take
// every 16th partial each (eight loads in flight), a fixed-order sum over the groups in LDS, scatter to dW[kk][o].  Replaces
// the two-stage pass of rounds 1-3 (two launches with a round trip through a `stage` buffer); measured equal in the step
// (0.107-0.109 ms per layer either way: the partials are not what this op waits for), one launch less.
constexpr int BW_PARTS = 16;
__global__ void __launch_bounds__(64 * BW_PARTS)
reduce_partials_wide(const float* __restrict__ partial, float* __restrict__ dW, int nx, int ny, int rt, int FinK, int Fout) {
    __shared__ float red[BW_PARTS][64];
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int per = rt * 16 * 64;
    const int row = blockIdx.x, y = blockIdx.y, z = blockIdx.z;
    const float* base = partial + ((size_t)z * ny + y) * nx * per + (size_t)row * 64 + lane;
    float s = 0.f;
    for (int x0 = part; x0 < nx; x0 += 8 * BW_PARTS) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int x = x0 + BW_PARTS * u;
            v[u] = base[(size_t)(x < nx ? x : x0) * per];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (x0 + BW_PARTS * u < nx) s += v[u];
    }
    red[part][lane] = s;
    __syncthreads();
    if (part == 0) {
        float t = red[0][lane];
#pragma unroll
        for (int q = 1; q < BW_PARTS; ++q) t += red[q][lane];
        const int tt = row >> 4, j = row & 15, h = lane >> 5;
        const int kk = (y * rt + tt) * 32 + acc_row(j, h);
        const int fo = z * 32 + (lane & 31);
        if (kk < FinK && fo < Fout) dW[(size_t)kk * Fout + fo] = t;
    }
}

static int bw_rt(int ntiles) { return ntiles < 5 ? ntiles : 5; }
static int num_cus() {
    static int cus = 0;                       // cached: hipGetDeviceProperties is slow
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
                  ? prop.multiProcessorCount : 256;
    }
    return cus;
}
// a launch of one wave per 128-vertex tile that gives the chip fewer than two workgroups per CU: share the tiles
// W'[(fo*K + k)][fin] = W[(fin*K + k)][fo]: the weights of the contraction that forms the input gradient from the stack of dy
__global__ void __launch_bounds__(256)
reindex_weights_kernel(const float* __restrict__ W, float* __restrict__ Wt, int Fin, int K, int Fout) {
    const int idx = blockIdx.x * 256 + threadIdx.x;                // over the OUTPUT: consecutive threads write consecutive words
    if (idx >= Fin * K * Fout) return;
    const int fin = idx % Fin, kk = idx / Fin, k = kk % K, fo = kk / K;
    Wt[idx] = W[((size_t)fin * K + k) * Fout + fo];
}

// the same for up to 16 layers in ONE launch (blockIdx.y = layer): within a training step the weights are constant, and five
// launches of 8.6 us each sat on the backward pass's critical path (rocprofv3, round 5)
struct ReindexBatch {
    const float* W[16];
    float* Wt[16];
    int Fin[16], K[16], Fout[16];
};
__global__ void __launch_bounds__(256)
reindex_weights_batch_kernel(ReindexBatch b) {
    const int l = blockIdx.y;
    const int Fin = b.Fin[l], K = b.K[l], Fout = b.Fout[l];
    const float* __restrict__ W = b.W[l];
    float* __restrict__ Wt = b.Wt[l];
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < Fin * K * Fout; idx += gridDim.x * 256) {
        const int fin = idx % Fin, kk = idx / Fin, k = kk % K, fo = kk / K;
        Wt[idx] = W[((size_t)fin * K + k) * Fout + fo];
    }
}

static bool small_launch(int B, int M) { return ((M + 511) / 512) * B < 2 * num_cus(); }
// `groups` = row-tile groups x column tiles of the launch (gridDim.y * gridDim.z): every group walks all chunks with gx workgroups
static int bw_grid_x(int B, int M, int groups) {
    const int cus = num_cus();
    int total = B * ((M + 63) / 64);
    // 48 KB of LDS per workgroup would allow three per CU (rounds 3-5: gx = 3 * cus whatever the number of groups).  Measured in
    // round 6 (EXPERIMENTS 8.6): TWO workgroups per CU over the whole launch -- one group (32*5 -> 32 at the bench launch): 768 /
    // 640 / 512 / 384 / 256 workgroups 0.111 / 0.123 / 0.098 / 0.112 / 0.140 ms; two groups (15*20 -> 32 and 32*10 -> 32 at M = 12672):
    // gx = 768 / 512 / 384 / 256 0.236 / 0.266 / 0.235 / 0.216 ms.  A multiple of 64, at least a quarter of the CUs per group.
    int gx = (2 * cus / (groups < 1 ? 1 : groups) + 63) / 64 * 64;
    if (gx < cus / 4) gx = cus / 4;
    if (const char* e = getenv("CHEBGCN_BWW_GX")) gx = atoi(e) > 0 ? atoi(e) : gx;                  // (experiment knob)
    // small launches: at least three chunks per workgroup -- every workgroup leaves a partial of the whole row-tile group
    // (20 KB at five row tiles) that the reduce kernels read back; one chunk per workgroup made the partials of an
    // atlas-sized layer (N = 360, batch 128) 31 MB and reduce_partials_stage1 18 us beside a 38 us kernel
#ifndef CG_BWW_MINCHUNK
#define CG_BWW_MINCHUNK 3      // (round 6, captured atlas step at N = 360: 192 workgroups 0.754 ms, 256 0.741, 384 0.742, 128 0.779, 768 0.788)
#endif
    if (gx > (total + CG_BWW_MINCHUNK - 1) / CG_BWW_MINCHUNK) gx = (total + CG_BWW_MINCHUNK - 1) / CG_BWW_MINCHUNK;
    if (const char* e = getenv("CHEBGCN_BWW_GX_SMALL")) gx = (atoi(e) > 0 && total < 4 * 2 * cus) ? std::min(atoi(e), total) : gx;   // (experiment knob)
    return gx < 1 ? 1 : gx;
}

}  // namespace chebgcn

using namespace chebgcn;

static int check_pool(int pool, int M) {
    if (pool < 1 || pool > 128 || (pool & (pool - 1)) != 0 || (M % pool) != 0) return 0;
    return 1;
}

extern "C" int chebgcn_reindex_weights(const float* W, float* Wt, int Fin, int K, int Fout, chebgcn_stream stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CG_REQUIRE(W && Wt && W != Wt, "reindex_weights: NULL argument (or in place)");
    CG_REQUIRE(Fin > 0 && K > 0 && Fout > 0 && (int64_t)Fin * K * Fout < (1ll << 30), "reindex_weights: bad shape");
    note_dispatch("reindex_weights_kernel");
    hipLaunchKernelGGL(reindex_weights_kernel, dim3((Fin * K * Fout + 255) / 256), dim3(256), 0, stream, W, Wt, Fin, K, Fout);
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

extern "C" int chebgcn_reindex_weights_batch(int n, const float* const* W, float* const* Wt, const int* Fin, const int* K,
                                             const int* Fout, chebgcn_stream stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CG_REQUIRE(n >= 1 && n <= 16 && W && Wt && Fin && K && Fout, "reindex_weights_batch: bad argument (1..16 layers)");
    ReindexBatch b;
    int most = 0;
    for (int l = 0; l < n; ++l) {
        CG_REQUIRE(W[l] && Wt[l] && W[l] != Wt[l], "reindex_weights_batch: NULL argument (or in place), layer %d", l);
        CG_REQUIRE(Fin[l] > 0 && K[l] > 0 && Fout[l] > 0 && (int64_t)Fin[l] * K[l] * Fout[l] < (1ll << 30),
                   "reindex_weights_batch: bad shape, layer %d", l);
        b.W[l] = W[l]; b.Wt[l] = Wt[l]; b.Fin[l] = Fin[l]; b.K[l] = K[l]; b.Fout[l] = Fout[l];
        most = std::max(most, Fin[l] * K[l] * Fout[l]);
    }
    note_dispatch("reindex_weights_batch_kernel");
    hipLaunchKernelGGL(reindex_weights_batch_kernel, dim3(std::min((most + 255) / 256, 64), n), dim3(256), 0, stream, b);
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

extern "C" int chebgcn_contract_fwd(const float* stack, const float* W, const float* bias, int bias_kind,
                                    float* out, uint8_t* argmax, int B, int M, int Fin, int K, int Fout,
                                    int pool, int pool_kind, int relu, chebgcn_stream stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CG_REQUIRE(stack && W && out, "contract_fwd: NULL argument");
    CG_REQUIRE(B > 0 && M > 0 && Fin > 0 && K > 0 && Fout > 0, "contract_fwd: bad shape");
    CG_REQUIRE(B <= 65535, "contract_fwd: B > 65535");
    CG_REQUIRE(check_pool(pool, M), "contract_fwd: pool=%d must be a power of two <= 128 dividing M=%d", pool, M);
    CG_REQUIRE(bias_kind == CHEBGCN_BIAS_NONE || bias, "contract_fwd: bias_kind set but bias is NULL");
    CG_REQUIRE(bias_kind >= 0 && bias_kind <= 2 && (pool_kind == 0 || pool_kind == 1), "contract_fwd: bad kind");
    FwdArgs a;
    a.stack = stack; a.W = W; a.bias = bias; a.out = out;
    CG_REQUIRE(!(pool_kind == CHEBGCN_POOL_AVG && relu && argmax && pool > 8),
               "contract_fwd: average pooling keeps a ReLU mask only for pool <= 8");
    a.argmax = pool > 1 ? argmax : nullptr;
    a.relu_mask = (pool == 1 && relu) ? argmax : nullptr;
    a.B = B; a.M = M; a.Mp = plane_stride(M); a.Fin = Fin; a.K = K; a.Fout = Fout; a.FinK = Fin * K;
    a.pool = pool; a.pool_kind = pool_kind; a.relu = relu; a.bias_kind = bias_kind;
    a.Mo = M / pool; a.Mpo = plane_stride(a.Mo);
    a.slab = (size_t)B * Fin * a.Mp;
    const int gx = (M + 511) / 512;
    if (Fout > 32) {
        dim3 grid(gx, B, (Fout + 63) / 64);
        note_dispatch("contract_fwd_kernel<2>");
        hipLaunchKernelGGL(contract_fwd_kernel<2>, grid, dim3(256), 0, stream, a);
    } else if (small_launch(B, M)) {
        dim3 grid((M + 127) / 128, B, 1);
        note_dispatch("contract_fwd_splitk_kernel");
        hipLaunchKernelGGL(contract_fwd_splitk_kernel, grid, dim3(256), 0, stream, a);
    } else {
        dim3 grid(gx, B, 1);
#if CG_FWD_RING
        // whole ring rounds of row pairs; W and the row offsets of the padded rows in LDS (136 bytes per row)
        const int nrows_pad = ((a.FinK + 2 * RING - 1) / (2 * RING)) * (2 * RING);
        if ((size_t)nrows_pad * 136 <= 48 * 1024 && RING <= 8 && (bias_kind != CHEBGCN_BIAS_FILTER || Fout >= 4)) {
            if (a.pool == 1 && a.out_K == 0) {
                note_dispatch("contract_fwd_ring_kernel");
                hipLaunchKernelGGL(contract_fwd_ring_kernel<true>, grid, dim3(256), (size_t)nrows_pad * 136, stream, a, nrows_pad);
            } else {
                note_dispatch("contract_fwd_ring_kernel<pool>");
                hipLaunchKernelGGL(contract_fwd_ring_kernel<false>, grid, dim3(256), (size_t)nrows_pad * 136, stream, a, nrows_pad);
            }
            CG_HIP(hipGetLastError());
            return CHEBGCN_OK;
        }
#endif
        note_dispatch("contract_fwd_kernel<1>");
        hipLaunchKernelGGL(contract_fwd_kernel<1>, grid, dim3(256), 0, stream, a);
    }
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

// whole ring rounds of row pairs; W and the row offsets of the padded rows in LDS (136 bytes per row)
static int ring_rows(int FinK) { return ((FinK + 2 * RING - 1) / (2 * RING)) * (2 * RING); }

extern "C" int chebgcn_contract_fwd_mean_supported(int B, int M, int Fin, int K, int Fout) {
    // Fout < 4: the ring kernel's 16-byte load around a per-filter bias would start in front of the array (chebgcn_contract_fwd
    // sends such layers to contract_fwd_kernel<1>); not served here whatever the bias kind
    if (B <= 0 || M <= 0 || Fin <= 0 || K <= 0 || Fout < 4 || Fout > 32 || B > 65535) return 0;
    return !small_launch(B, M) && (size_t)ring_rows(Fin * K) * 136 <= 48 * 1024;
}

extern "C" int chebgcn_contract_fwd_mean(const float* stack, const float* W, const float* bias, int bias_kind, float* mean_out,
                                         uint8_t* relu_mask, int B, int M, int Fin, int K, int Fout, chebgcn_stream stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CG_REQUIRE(stack && W && mean_out, "contract_fwd_mean: NULL argument");
    CG_REQUIRE(bias_kind == CHEBGCN_BIAS_NONE || bias, "contract_fwd_mean: bias_kind set but bias is NULL");
    CG_REQUIRE(bias_kind >= 0 && bias_kind <= 2, "contract_fwd_mean: bad kind");
    if (!chebgcn_contract_fwd_mean_supported(B, M, Fin, K, Fout))
        return fail(CHEBGCN_EUNSUPPORTED, "contract_fwd_mean: shape not served (chebgcn_contract_fwd_mean_supported)");
    FwdArgs a;
    a.stack = stack; a.W = W; a.bias = bias; a.out = nullptr; a.argmax = nullptr;
    a.relu_mask = relu_mask; a.mean_out = mean_out;
    a.B = B; a.M = M; a.Mp = plane_stride(M); a.Fin = Fin; a.K = K; a.Fout = Fout; a.FinK = Fin * K;
    a.pool = 1; a.pool_kind = CHEBGCN_POOL_MAX; a.relu = 1; a.bias_kind = bias_kind;
    a.Mo = M; a.Mpo = a.Mp;
    a.slab = (size_t)B * Fin * a.Mp;
    const int nrows_pad = ring_rows(a.FinK);
    note_dispatch("contract_fwd_ring_kernel<mean>");
    hipLaunchKernelGGL(contract_fwd_ring_kernel<true>, dim3((M + 511) / 512, B, 1), dim3(256), (size_t)nrows_pad * 136, stream, a, nrows_pad);
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

extern "C" int chebgcn_contract_fwd_gated_supported(int B, int M, int Fin, int K, int Fout) {
    if (B <= 0 || M <= 0 || Fin <= 0 || K <= 0 || Fout <= 0 || Fout > 32 || B > 65535) return 0;
    return !small_launch(B, M) && (size_t)ring_rows(Fin * K) * 136 <= 48 * 1024;
}

extern "C" int chebgcn_contract_fwd_gated(const float* stack, const float* W, const uint8_t* gate, float* out, int B, int M,
                                          int Fin, int K, int Fout, chebgcn_stream stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CG_REQUIRE(stack && W && gate && out, "contract_fwd_gated: NULL argument");
    if (!chebgcn_contract_fwd_gated_supported(B, M, Fin, K, Fout))
        return fail(CHEBGCN_EUNSUPPORTED, "contract_fwd_gated: shape not served (chebgcn_contract_fwd_gated_supported)");
    FwdArgs a;
    a.stack = stack; a.W = W; a.bias = nullptr; a.out = out; a.argmax = nullptr; a.gate = gate;
    a.B = B; a.M = M; a.Mp = plane_stride(M); a.Fin = Fin; a.K = K; a.Fout = Fout; a.FinK = Fin * K;
    a.pool = 1; a.pool_kind = CHEBGCN_POOL_MAX; a.relu = 0; a.bias_kind = CHEBGCN_BIAS_NONE;
    a.Mo = M; a.Mpo = a.Mp;
    a.slab = (size_t)B * Fin * a.Mp;
    const int nrows_pad = ring_rows(a.FinK);
    static const int lds_min = [] { const char* e = getenv("CHEBGCN_GATED_LDS"); return e ? atoi(e) : 0; }();
    note_dispatch("contract_fwd_ring_kernel<gated>");
    hipLaunchKernelGGL((contract_fwd_ring_kernel<true, true>), dim3((M + 511) / 512, B, 1), dim3(256),
                       std::max((size_t)nrows_pad * 136, (size_t)lds_min), stream, a, nrows_pad);
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

static int launch_bwd_x(const float* dy, const uint8_t* mask, const float* W, float* gstack, int B, int M, int Fin, int K,
                        int Fout, hipStream_t stream, bool one_plane = false) {
    BwdXArgs a;
    a.dy = dy; a.W = W; a.gstack = gstack; a.mask = mask;
    a.B = B; a.M = M; a.Mp = plane_stride(M); a.Fin = Fin; a.K = K; a.Fout = Fout; a.FinK = Fin * K;
    a.slab = (size_t)B * Fin * a.Mp;
    a.dy_bstride = one_plane ? (size_t)a.Mp : (size_t)Fout * a.Mp;
    a.dy_fstride = one_plane ? 0 : (size_t)a.Mp;
    if (small_launch(B, M)) {
        dim3 sgrid((M + 127) / 128, B, 1);
#define CG_BX(H, MK, SP, G)                                                                     \
    do {                                                                                        \
        note_dispatch("contract_bwd_x_kernel<" #H "," #MK "," #SP ">");                          \
        hipLaunchKernelGGL((contract_bwd_x_kernel<H, MK, SP>), G, dim3(256), 0, stream, a);      \
    } while (0)
        if (mask) {
            if (Fout <= 32) CG_BX(true, true, true, sgrid);
            else CG_BX(false, true, true, sgrid);
        } else {
            if (Fout <= 32) CG_BX(true, false, true, sgrid);
            else CG_BX(false, false, true, sgrid);
        }
        CG_HIP(hipGetLastError());
        return CHEBGCN_OK;
    }
    dim3 grid((M + 511) / 512, B, 1);
#if CG_BWX_LDS
    const int nrows32 = ((a.FinK + 31) / 32) * 32;
    if (Fout <= 32 && a.FinK % 32 == 0 && (size_t)nrows32 * 136 <= 48 * 1024) {
        note_dispatch(mask ? "contract_bwd_x_lds_kernel<true>" : "contract_bwd_x_lds_kernel<false>");
        if (mask) hipLaunchKernelGGL((contract_bwd_x_lds_kernel<true>), grid, dim3(256), (size_t)nrows32 * 136, stream, a, nrows32);
        else hipLaunchKernelGGL((contract_bwd_x_lds_kernel<false>), grid, dim3(256), (size_t)nrows32 * 136, stream, a, nrows32);
        CG_HIP(hipGetLastError());
        return CHEBGCN_OK;
    }
#endif
    if (mask) {
        if (Fout <= 32) CG_BX(true, true, false, grid);
        else CG_BX(false, true, false, grid);
    } else {
        if (Fout <= 32) CG_BX(true, false, false, grid);
        else CG_BX(false, false, false, grid);
    }
#undef CG_BX
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

extern "C" int chebgcn_contract_bwd_x(const float* dy, const float* W, float* gstack, int B, int M,
                                      int Fin, int K, int Fout, chebgcn_stream stream_) {
    CG_REQUIRE(dy && W && gstack, "contract_bwd_x: NULL argument");
    CG_REQUIRE(B > 0 && M > 0 && Fin > 0 && K > 0 && Fout > 0 && B <= 65535, "contract_bwd_x: bad shape");
    return launch_bwd_x(dy, nullptr, W, gstack, B, M, Fin, K, Fout, (hipStream_t)stream_);
}

extern "C" int chebgcn_contract_bwd_x_relu(const float* dout, const uint8_t* relu_mask, const float* W, float* gstack, int B,
                                           int M, int Fin, int K, int Fout, chebgcn_stream stream_) {
    CG_REQUIRE(dout && relu_mask && W && gstack, "contract_bwd_x_relu: NULL argument");
    CG_REQUIRE(B > 0 && M > 0 && Fin > 0 && K > 0 && Fout > 0 && B <= 65535, "contract_bwd_x_relu: bad shape");
    return launch_bwd_x(dout, relu_mask, W, gstack, B, M, Fin, K, Fout, (hipStream_t)stream_);
}

extern "C" int chebgcn_contract_bwd_x_relu_mean(const float* gmean, const uint8_t* relu_mask, const float* W, float* gstack,
                                                int B, int M, int Fin, int K, int Fout, chebgcn_stream stream_) {
    CG_REQUIRE(gmean && relu_mask && W && gstack, "contract_bwd_x_relu_mean: NULL argument");
    CG_REQUIRE(B > 0 && M > 0 && Fin > 0 && K > 0 && Fout > 0 && B <= 65535, "contract_bwd_x_relu_mean: bad shape");
    return launch_bwd_x(gmean, relu_mask, W, gstack, B, M, Fin, K, Fout, (hipStream_t)stream_, true);
}

extern "C" size_t chebgcn_contract_bwd_w_workspace(int B, int M, int Fin, int K, int Fout) {
    if (B <= 0 || M <= 0 || Fin <= 0 || K <= 0 || Fout <= 0) return 0;
    const int ntiles = (Fin * K + 31) / 32, rt = bw_rt(ntiles);
    const int gy = (ntiles + rt - 1) / rt, gz = (Fout + 31) / 32, gx = bw_grid_x(B, M, gy * gz);
    return (size_t)gx * gy * gz * rt * 16 * 64 * sizeof(float);   // one partial per workgroup
}

static bool bwd_w_merges_bias(int B, int M, int Fin, int K, int Fout);
static int launch_bwd_w(const float* stack, const float* dy, const uint8_t* mask, float* dW, void* workspace, int B, int M,
                        int Fin, int K, int Fout, hipStream_t stream, bool one_plane = false, float* dbias_vertex = nullptr) {
    BwdWArgs a;
    a.stack = stack; a.dy = dy; a.partial = (float*)workspace; a.mask = mask;
    a.B = B; a.M = M; a.Mp = plane_stride(M); a.Fin = Fin; a.K = K; a.Fout = Fout; a.FinK = Fin * K;
    a.dy_bstride = one_plane ? (size_t)a.Mp : (size_t)Fout * a.Mp;
    a.dy_fstride = one_plane ? 0 : (size_t)a.Mp;
    a.nchunks_m = (M + 63) / 64;
    a.ntiles = (a.FinK + 31) / 32;
    a.slab = (size_t)B * Fin * a.Mp;
    const int rt = bw_rt(a.ntiles);
    const int gy = (a.ntiles + rt - 1) / rt, gz = (Fout + 31) / 32, gx = bw_grid_x(B, M, gy * gz);
    dim3 grid(gx, gy, gz);
    const size_t lds = (size_t)(rt + 1) * 32 * BW_ROW * sizeof(float);
#define CG_BWK contract_bwd_w_kernel
#define CG_BW(N)                                                                                        \
    case N:                                                                                             \
        note_dispatch(mask ? "contract_bwd_w_kernel<" #N ",true>" : "contract_bwd_w_kernel<" #N ",false>"); \
        if (mask) {                                                                                     \
            CG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(CG_BWK<N, true>),                  \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));          \
            hipLaunchKernelGGL((CG_BWK<N, true>), grid, dim3(256), lds, stream, a);                     \
        } else {                                                                                        \
            CG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(CG_BWK<N, false>),                 \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));          \
            hipLaunchKernelGGL((CG_BWK<N, false>), grid, dim3(256), lds, stream, a);                    \
        }                                                                                               \
        break
    switch (rt) {
        CG_BW(1); CG_BW(2); CG_BW(3); CG_BW(4); CG_BW(5);
        default: return fail(CHEBGCN_EUNSUPPORTED, "contract_bwd_w: rt=%d", rt);
    }
#undef CG_BW
#undef CG_BWK
    CG_HIP(hipGetLastError());
    if (dbias_vertex) {                               // (the caller checked bwd_w_merges_bias: a small launch, the 16-subset bias shape)
        int parts = 0;
        const int nbx = bias_grad_blocks(M, Fout, &parts);
        note_dispatch_more("reduce_partials_small_bias_kernel");
        hipLaunchKernelGGL(reduce_partials_small_bias_kernel, dim3(rt * 16 * gy * gz + nbx * Fout), dim3(256), 0, stream,
                           (const float*)workspace, dW, gx, gy, gz, rt, a.FinK, Fout, dy, mask, dbias_vertex, B, M, a.Mp, nbx);
        CG_HIP(hipGetLastError());
        return CHEBGCN_OK;
    }
    if (gx <= 256) {
        note_dispatch_more("reduce_partials_small");
        hipLaunchKernelGGL(reduce_partials_small, dim3(rt * 16, gy, gz), dim3(256), 0, stream, (const float*)workspace, dW, gx, gy, rt,
                           a.FinK, Fout);
        CG_HIP(hipGetLastError());
        return CHEBGCN_OK;
    }
    note_dispatch_more("reduce_partials_wide");
    hipLaunchKernelGGL(reduce_partials_wide, dim3(rt * 16, gy, gz), dim3(64 * BW_PARTS), 0, stream, (const float*)workspace, dW, gx, gy,
                       rt, a.FinK, Fout);
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

extern "C" int chebgcn_contract_bwd_w_relu_mean(const float* stack, const float* gmean, const uint8_t* relu_mask, float* dW,
                                                void* workspace, size_t workspace_bytes, int B, int M, int Fin, int K,
                                                int Fout, chebgcn_stream stream_) {
    CG_REQUIRE(stack && gmean && relu_mask && dW && workspace, "contract_bwd_w_relu_mean: NULL argument");
    CG_REQUIRE(B > 0 && M > 0 && Fin > 0 && K > 0 && Fout > 0, "contract_bwd_w_relu_mean: bad shape");
    CG_REQUIRE(workspace_bytes >= chebgcn_contract_bwd_w_workspace(B, M, Fin, K, Fout),
               "contract_bwd_w_relu_mean: workspace too small");
    return launch_bwd_w(stack, gmean, relu_mask, dW, workspace, B, M, Fin, K, Fout, (hipStream_t)stream_, true);
}

extern "C" int chebgcn_contract_bwd_w(const float* stack, const float* dy, float* dW, void* workspace,
                                      size_t workspace_bytes, int B, int M, int Fin, int K, int Fout,
                                      chebgcn_stream stream_) {
    CG_REQUIRE(stack && dy && dW && workspace, "contract_bwd_w: NULL argument");
    CG_REQUIRE(B > 0 && M > 0 && Fin > 0 && K > 0 && Fout > 0, "contract_bwd_w: bad shape");
    CG_REQUIRE(workspace_bytes >= chebgcn_contract_bwd_w_workspace(B, M, Fin, K, Fout),
               "contract_bwd_w: workspace too small");
    return launch_bwd_w(stack, dy, nullptr, dW, workspace, B, M, Fin, K, Fout, (hipStream_t)stream_);
}

extern "C" int chebgcn_contract_bwd_w_relu(const float* stack, const float* dout, const uint8_t* relu_mask, float* dW,
                                           void* workspace, size_t workspace_bytes, int B, int M, int Fin, int K, int Fout,
                                           chebgcn_stream stream_) {
    CG_REQUIRE(stack && dout && relu_mask && dW && workspace, "contract_bwd_w_relu: NULL argument");
    CG_REQUIRE(B > 0 && M > 0 && Fin > 0 && K > 0 && Fout > 0, "contract_bwd_w_relu: bad shape");
    CG_REQUIRE(workspace_bytes >= chebgcn_contract_bwd_w_workspace(B, M, Fin, K, Fout),
               "contract_bwd_w_relu: workspace too small");
    return launch_bwd_w(stack, dout, relu_mask, dW, workspace, B, M, Fin, K, Fout, (hipStream_t)stream_);
}

// one launch for the partials' sum and the per-vertex bias gradient: where the weight gradient leaves few partials (a small launch)
// and the bias reduction takes its 16-subset shape (the same condition chebgcn_brelu_pool_bwd applies)
static bool bwd_w_merges_bias(int B, int M, int Fin, int K, int Fout) {
    const int ntiles = (Fin * K + 31) / 32, rt = bw_rt(ntiles);
    const int gy = (ntiles + rt - 1) / rt, gz = (Fout + 31) / 32;
    int parts = 0;
    bias_grad_blocks(M, Fout, &parts);
    return bw_grid_x(B, M, gy * gz) <= 256 && parts == 16;
}

extern "C" int chebgcn_contract_bwd_w_relu_bias_merged(int B, int M, int Fin, int K, int Fout) {
    if (B <= 0 || M <= 0 || Fin <= 0 || K <= 0 || Fout <= 0) return 0;
    return bwd_w_merges_bias(B, M, Fin, K, Fout) ? 1 : 0;
}

extern "C" int chebgcn_contract_bwd_w_relu_bias(const float* stack, const float* dout, const uint8_t* relu_mask, float* dW,
                                                float* dbias, void* workspace, size_t workspace_bytes, int B, int M, int Fin,
                                                int K, int Fout, chebgcn_stream stream_) {
    CG_REQUIRE(stack && dout && relu_mask && dW && dbias && workspace, "contract_bwd_w_relu_bias: NULL argument");
    CG_REQUIRE(B > 0 && M > 0 && Fin > 0 && K > 0 && Fout > 0, "contract_bwd_w_relu_bias: bad shape");
    CG_REQUIRE(workspace_bytes >= chebgcn_contract_bwd_w_workspace(B, M, Fin, K, Fout),
               "contract_bwd_w_relu_bias: workspace too small");
    if (!bwd_w_merges_bias(B, M, Fin, K, Fout))
        return fail(CHEBGCN_EUNSUPPORTED, "contract_bwd_w_relu_bias: not a small launch (chebgcn_contract_bwd_w_relu_bias_merged)");
    return launch_bwd_w(stack, dout, relu_mask, dW, workspace, B, M, Fin, K, Fout, (hipStream_t)stream_, false, dbias);
}
