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
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "contract_common.h"

namespace chebgcn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// W [FinK][Fout] fp32 -> Wp[part][ks][fo][16] bf16 (part 0 = hi, 1 = lo), zero beyond FinK / Fout
__global__ void __launch_bounds__(256)
pack_w_bf16_kernel(const float* __restrict__ W, __bf16* __restrict__ Wp, int FinK, int Fout, int nks, int FoutP,
                   int parts, int ldT) {
    // ldT > 0: the operand is the TRANSPOSE of the row-major matrix W[Fout][ldT] (bwd_x: W^T)
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;       // (ks, fo)
    if (idx >= nks * FoutP) return;
    const int ks = idx / FoutP, fo = idx - ks * FoutP;
    // (all sixteen loads first, then two 16-byte stores per image: element-wise 2-byte stores made this 11 us a call -- eight
    // calls per step of the six-level pooling network)
    float w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = ks * 16 + i;
        const bool in = r < FinK && fo < Fout;
        const size_t at = in ? (ldT > 0 ? (size_t)fo * ldT + r : (size_t)r * Fout + fo) : 0;
        const float v = W[at];
        w[i] = in ? v : 0.f;
    }
    bf16x8 hi[2], lo[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const __bf16 h = (__bf16)w[i];
        hi[i >> 3][i & 7] = h;
        lo[i >> 3][i & 7] = (__bf16)(w[i] - (float)h);
    }
    bf16x8* dst = reinterpret_cast<bf16x8*>(Wp + (size_t)idx * 16);
    dst[0] = hi[0];
    dst[1] = hi[1];
    if (parts > 1) {
        bf16x8* dl = reinterpret_cast<bf16x8*>(Wp + ((size_t)nks * FoutP + idx) * 16);
        dl[0] = lo[0];
        dl[1] = lo[1];
    }
}

// Operand pipeline.  Both operands of a k-step travel by LDS-DMA (global_load_lds, 16 B per lane,
// no register round trip) into a ring of stages in LDS:
//   stage = [16 rows][128 vertices] fp32 of the stack (8 KB) + PARTS x [256 filters][16] bf16 of W (8 KB each).
// What is in flight per CU is DEPTH *distinct* stages -- with register prefetch the waves of a
// workgroup would all request the same rows and only one wave's worth of bytes would be in the
// air.  Every vector-memory operation of a producer wave is such a DMA issued DEPTH steps ahead:
// the queue returns in order, so a short L2 load between them would wait for the HBM ones.
// Hand-placed waits: a producer's own DMAs of the step about to be consumed are DEPTH*NDMA
// operations old (s_waitcnt vmcnt(DEPTH*NDMA)), a barrier WITHOUT a fence (the fence would
// drain the queue) publishes them, and the operands are read with ds_read_b128 in one asm
// block (compiler-visible LDS reads would be ordered behind every DMA in flight).
typedef int i32x4 __attribute__((ext_vector_type(4)));

// NW = compute waves per workgroup = 64-filter slices per work item: 4 (256 filters), or 5 (320) where that pads the
// filter count less -- the gradient wrt the stack at config 5 has 300 "filters" (rows Fin*K): one group of 320
// instead of two of 256.
#ifndef CG_BF16_XCD
#define CG_BF16_XCD 1
#endif
#ifndef CG_X
#define CG_X 0               // 64: in-kernel phase stamps (tools/bbuild.sh, tools/kbench.py --stamps); 0 in production
#endif
__device__ long long g_dbgb[16 * 64];
// third item of workgroup 37: compute waves stamp id = k-step (after its barrier), 40 = k-loop done, 41.. = bias steps, 50 = epilogue
// done; producer waves stamp id = stage of the item (after its vmcnt wait)
#define CG_BSTAMP(cond, id)                                                                                   \
    do {                                                                                                      \
        if ((CG_X & 64) && (cond) && (id) < 64 && lane == 0 && blockIdx.x == 37)                              \
            g_dbgb[wave * 64 + (id)] = (long long)__builtin_readcyclecounter();                               \
    } while (0)
// Who issues the DMAs.  Vector-memory operations of a wave complete IN ORDER and s_waitcnt vmcnt counts loads and stores
// alike: when the waves that multiply also issued the DMAs, the first k-step of an item waited for the 32 result stores of the
// previous item's epilogue to be acknowledged by memory before it could see its operands, with nothing else in flight (timing
// experiment without the epilogue: config 5 forward 0.42 -> 0.21 ms, gradient wrt the stack 0.40 -> 0.20 ms).  So the DMAs
// belong to TWO PRODUCER WAVES (waves NW and NW + 1) that never store: they run the operand ring DEPTH stages ahead, wait for
// their own DMAs with vmcnt and release the NW compute waves through the workgroup barrier; the compute waves never wait on
// vmcnt in the main loop and their stores drain behind the next item's matrix work.  Two producers because vmcnt is a 6-bit
// counter: DEPTH stages x the DMA instructions of a stage must stay below 64 per wave.
// Producer 0 issues what compute waves 0, 1 (and 4: packed W only) used to issue, producer 1 those of waves 2, 3.
//
// X16: the reduction rows arrive as bf16 planes (the gradient wrt the stack reading the dy that chebgcn_relu_grad_bf16 wrote):
// a stage holds [16 rows][128 vertices] bf16 (4 KB, one DMA instruction per four rows), lane (c, g) reads the four vertices
// 4c..4c+3 of its eight rows with ds_read_b64 and transposes 8 x 4 halves into the four operands with v_perm_b32 (as many
// instructions as the fp32 -> bf16 conversions they replace).
//
// NVT: vertex tiles per work item.  With NVT = 1 the NW compute waves are NW filter slices of 64 on ONE tile of 128 vertices
// (layers of 256 filters and more, the gradient wrt the stack).  A layer of 64 or 128 filters would leave three or two of
// four waves multiplying padding (round 4: the config-4 forward, 64 filters, ran at 72 of the 288 TFLOP/s it issued): there
// the waves are NW / NVT filter slices x NVT vertex tiles -- wave w = slice w % NWF of tile w / NWF -- a stage carries the
// 16 reduction rows of all NVT tiles and the NWF slices of the packed W.
template <int PASSES, int NW, bool X16 = false, int NVT = 1>
struct Bf16Cfg {
    static_assert(NW % NVT == 0 && (NVT == 1 || NW == 4), "compute waves = filter slices x vertex tiles");
    static constexpr int NWF = NW / NVT;                          // filter slices of 64
    static constexpr int PARTS = PASSES == 3 ? 2 : 1;
    static constexpr int WPART = NWF * 64 * 32;                   // bytes of one bf16 image of W per stage
    static constexpr int XTILE = X16 ? 4096 : 8192;               // bytes of the reduction rows of one vertex tile per stage
    static constexpr int XPART = NVT * XTILE;
    static constexpr int STAGE = XPART + PARTS * WPART;           // bytes
    static constexpr int NSTAGE = (147456 / STAGE);               // <= 144 KB
    static constexpr int NXDMA = X16 ? 1 : 2;                     // DMA instructions of the reduction rows per four rows, tile and stage
    // filter slice s of the packed W belongs to producer (s / 2) % 2 (one or two slices: s % 2)
    static constexpr int NWS1 = NWF >= 4 ? 2 : NWF / 2;           // slices of producer 1
    static constexpr int NWS0 = NWF - NWS1;                       // ... of producer 0
    static constexpr int NDMA1 = 2 * NXDMA * NVT + 2 * PARTS * NWS1;  // DMA instructions per stage of producer 1
    static constexpr int NDMA0 = 2 * NXDMA * NVT + 2 * PARTS * NWS0;  // ... of producer 0
    static constexpr int DEPTH = (NSTAGE - 2) < 63 / NDMA0 ? (NSTAGE - 2) : 63 / NDMA0;
    static_assert(DEPTH >= 2 && DEPTH * NDMA0 <= 63, "vmcnt is a 6-bit counter");
    // A per-vertex bias (b2relu: [Fout][Mp], as large as the result) travels through the SAME ring: after the k-steps of an
    // item come BIAS_STEPS stages of [wave][8 rows][128 vertices] fp32 -- for compute wave w the rows 32t + acc_row(4i..4i+3, g)
    // of its slice at step 4t + i -- which the compute waves add to their accumulators row by row as they store them.  No compute
    // wave loads anything from memory: the epilogue used to pay a round trip per batch of rows (config 5 forward: 8 per item).
    // A stage must be the same number of DMA instructions whatever it carries (the vmcnt waits are constants): 8 per
    // producer for the bias (2 waves x 8 rows x 512 B), padded with re-reads to NDMA1.  Four compute waves only.
    static constexpr bool RING_BIAS = NW == 4 && !X16 && NDMA1 >= 8;
    static constexpr int BIAS_STEPS = 8;
    // bias stage: 16 KB of rows + the padding DMAs' kilobyte behind them must lie inside the stage
    static_assert(!RING_BIAS || STAGE >= 16384, "bias stage");
    static_assert(!RING_BIAS || (NDMA0 <= 8 && NDMA1 <= 8) || STAGE >= 16384 + 1024, "padding DMAs of a bias stage leave the stage");
};

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int PASSES, int NW, bool X16 = false, int NVT = 1>
__global__ void __launch_bounds__((NW + 2) * 64)
contract_fwd_bf16_kernel(FwdArgs a, const __bf16* __restrict__ Wp, int nks, int FoutP, int ntm, int nitems) {
    static_assert(!X16 || PASSES == 1, "bf16 rows have no low part");
    using C = Bf16Cfg<PASSES, NW, X16, NVT>;
    constexpr int NWF = C::NWF;
    extern __shared__ __attribute__((aligned(16))) char ring[];         // [NSTAGE][STAGE]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, g = lane >> 5;
    const size_t lo_part = (size_t)nks * FoutP * 16;                     // bf16 elements between the hi and lo image

    // work item = (filter group of 64 NWF, window, 128 NVT vertices), walked with the pipeline running (ntm = tiles of 128 NVT).
    // XCD-aware order: workgroup w runs on XCD w % 8 (round-robin dispatch); XCD x takes the vertex tiles x, x + 8, ... for
    // ALL windows, so the per-vertex bias rows of its tiles (a wide layer's bias is [256][Mp] = 10.7 MB, more than one
    // 4 MB L2) stay in ITS L2 across the windows instead of being fetched again by whichever XCD meets the tile next.
    // (config 5, batch 64: forward 0.520 -> 0.482 ms; without a per-vertex bias -- the gradient wrt the stack runs through
    // this kernel too -- the tile order only scatters the stream: 0.426 -> 0.443 ms, so it keeps the linear order)
    const int NX = (CG_BF16_XCD && a.bias_kind == CHEBGCN_BIAS_VERTEX && gridDim.x % 8 == 0 && ntm >= 8) ? 8 : 1;
    const int xw = blockIdx.x % NX, lw = blockIdx.x / NX, Lw = gridDim.x / NX;
    const int ntx = (ntm - xw + NX - 1) / NX;                            // vertex tiles of this XCD
    const int nmine = ntx * (nitems / ntm);                              // its items; this workgroup takes lw, lw + Lw, ...
    auto item_mt = [&](int it) { return xw + NX * (it % ntx); };
    auto item_b = [&](int it) { return (it / ntx) % a.B; };
    auto item_z = [&](int it) { return it / ntx / a.B; };

    if (lw >= nmine) return;
    if (wave >= NW) {
        // ---- producer waves: the operand ring, DEPTH stages ahead of the consumers ---------------------------------
        // Incremental addressing: the item is decomposed once per item, the plane of reduction row r = 16*ks + row advances by
        // 16 rows per step.  Producer pw stands in for the compute waves v = 2 pw, 2 pw + 1 of the original schedule: DMA
        // instruction q of v covers the rows 4v + 2q + (lane >> 5) (fp32) or 4v + (lane >> 4) (X16).
        const int pw = wave - NW;
        int p_it = lw, p_ks = 0;
        const int nbias = (C::RING_BIAS && a.bias_kind == CHEBGCN_BIAS_VERTEX) ? C::BIAS_STEPS : 0;     // bias stages per item
        const int s16f = 16 / a.K, s16k = 16 % a.K;
        int pf[2][2], pk[2][2];
        // (hipcc 7.2: an array of template-dependent size captured by the lambdas below makes the HOST side drop the kernel's stub
        // without a diagnostic -- the library then fails to load with an undefined symbol; a fixed size does not)
        static_assert(NVT <= 4, "p_base");
        const float* p_base[4];                                    // window + vertex part of the source address, per vertex tile
        const __bf16* p_w;                                         // this lane's 16 bytes of the packed W, k-step 0, slice 0
        auto producer_item = [&]() {
            // (X16: addresses in units of float: a bf16 plane is Mp/2 floats long; lane l takes the 16-byte piece l%16 of its row)
#pragma unroll
            for (int t = 0; t < NVT; ++t) {
                int m = (item_mt(p_it) * NVT + t) * 128 + (X16 ? 8 * (lane & 15) : 4 * c);
                if (m >= a.Mp) m = 0;                              // beyond the plane: any readable address, never stored
                p_base[t] = X16 ? a.stack + (((size_t)item_b(p_it) * a.Fin * a.Mp + m) >> 1)
                                : a.stack + (size_t)item_b(p_it) * a.Fin * a.Mp + m;
            }
            p_w = Wp + (size_t)item_z(p_it) * (NWF * 64) * 16 + lane * 8;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < C::NXDMA; ++q) {
                    const int v = 2 * pw + j;
                    const int row = X16 ? 4 * v + (lane >> 4) : 4 * v + 2 * q + g;
                    pf[j][q] = row / a.K;
                    pk[j][q] = row - pf[j][q] * a.K;
                }
        };
        producer_item();
        const size_t w_step = (size_t)FoutP * 16;                  // bf16 elements per k-step of the packed W
        auto w_dma = [&](unsigned stage, int v) {
#pragma unroll
            for (int part = 0; part < C::PARTS; ++part)
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    __builtin_amdgcn_global_load_lds(p_w + (size_t)v * 1024 + part * lo_part + q * 512,
                                                     reinterpret_cast<__attribute__((address_space(3))) void*>(
                                                         stage + C::XPART + part * C::WPART + v * 2048 + q * 1024),
                                                     16, 0, 0);
        };
        auto produce = [&](int slot) {
            const unsigned stage = (unsigned)(size_t)ring + slot * C::STAGE;
            if (C::RING_BIAS && p_ks >= nks) {
                // bias stage i = p_ks - nks: instruction q of compute wave v covers the rows rho = 2q, 2q + 1 (g = rho >> 2,
                // jj = rho & 3) of [v][8 rows][128 vertices]: filter 64 (v % NWF) + 32 (i >> 2) + acc_row(4 (i & 3) + jj, g) at
                // the vertices of tile v / NWF
                const int i = p_ks - nks;
                auto bias_dma = [&](int u) __attribute__((always_inline)) {
                    // (beyond 8: padding -- re-reads of the first ones, all into ONE kilobyte at the start of the W part, which a bias
                    // stage does not use and every shape with padding has: see the static_assert of Bf16Cfg.  Round 5 aimed them at
                    // 16384 + pw*4096 + q*1024, past the end of the stage of <1,4,false,2>: the next slot's first rows)
                    const int uu = u & 7;
                    const int j = uu >> 2, q = uu & 3;
                    const int v = 2 * pw + j;
                    const int rho = 2 * q + g;
                    int m = (item_mt(p_it) * NVT + v / NWF) * 128 + 4 * c;
                    if (m >= a.Mp) m = 0;
                    int fo = item_z(p_it) * (NWF * 64) + 64 * (v % NWF) + 32 * (i >> 2) + acc_row(4 * (i & 3) + (rho & 3), rho >> 2);
                    if (fo >= a.Fout) fo = a.Fout - 1;
                    __builtin_amdgcn_global_load_lds(a.bias + (size_t)fo * a.Mp + m,
                                                     reinterpret_cast<__attribute__((address_space(3))) void*>(
                                                         u < 8 ? stage + v * 4096 + q * 1024 : stage + 16384), 16, 0, 0);
                };
                // (a stage is the same number of DMA instructions whatever it carries: the vmcnt waits are constants per producer)
                if (C::NDMA0 != C::NDMA1 && pw == 0) {
#pragma unroll
                    for (int u = 0; u < C::NDMA0; ++u) bias_dma(u);
                } else {
#pragma unroll
                    for (int u = 0; u < C::NDMA1; ++u) bias_dma(u);
                }
                if (++p_ks == nks + nbias) {
                    p_ks = 0;
                    if (p_it + Lw < nmine) p_it += Lw;
                    producer_item();
                }
                return;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int v = 2 * pw + j;
#pragma unroll
                for (int q = 0; q < C::NXDMA; ++q) {
                    // rows beyond Fin*K re-read the last plane (finite data) against zero weights
                    const bool live = pf[j][q] < a.Fin;
                    const int fin = live ? pf[j][q] : a.Fin - 1, k = live ? pk[j][q] : a.K - 1;
                    const size_t row_off = X16 ? (((size_t)k * a.slab + (size_t)fin * a.Mp) >> 1) : (size_t)k * a.slab + (size_t)fin * a.Mp;
#pragma unroll
                    for (int t = 0; t < NVT; ++t)
                        __builtin_amdgcn_global_load_lds(p_base[t] + row_off, reinterpret_cast<__attribute__((address_space(3))) void*>(
                                                                                  stage + t * C::XTILE + (X16 ? v * 1024 : (4 * v + 2 * q) * 512)), 16, 0, 0);
                    pk[j][q] += s16k;
                    pf[j][q] += s16f;
                    if (pk[j][q] >= a.K) { pk[j][q] -= a.K; ++pf[j][q]; }
                }
                if (NWF >= 4) w_dma(stage, v);
            }
            if (NWF > 4 && pw == 0) w_dma(stage, 4);
            if (NWF == 2) w_dma(stage, pw);
            if (NWF == 1 && pw == 0) w_dma(stage, 0);
            p_w += w_step;
            if (++p_ks == nks + nbias) {
                p_ks = 0;
                if (p_it + Lw < nmine) p_it += Lw;                  // after the last item: harmless re-reads
                producer_item();
            }
        };
        int pslot = 0;
#pragma unroll 1
        for (int d = 0; d < C::DEPTH; ++d) { produce(pslot); pslot = pslot + 1 == C::NSTAGE ? 0 : pslot + 1; }
        const long long steps = (long long)((nmine - lw + Lw - 1) / Lw) * (nks + nbias);
#pragma unroll 1
        for (long long st = 0; st < steps; ++st) {
            produce(pslot);
            pslot = pslot + 1 == C::NSTAGE ? 0 : pslot + 1;
            // this wave's part of the stage about to be consumed has landed ...
            if (pw == 0) wait_vmcnt<C::DEPTH * C::NDMA0>(); else wait_vmcnt<C::DEPTH * C::NDMA1>();
            CG_BSTAMP(st / (nks + nbias) == 2, (int)(st % (nks + nbias)));
            __builtin_amdgcn_s_barrier();                      // ... and the other producer's (no fence: it would drain the queue)
        }
        wait_vmcnt<0>();                                           // the run-ahead DMAs of the last item
        return;
    }

    // ---- compute waves ------------------------------------------------------------------------------------------------
    int cslot = 0;
    for (int it = lw; it < nmine; it += Lw) {
        const int vf = NVT == 1 ? wave : wave % NWF, vt = NVT == 1 ? 0 : wave / NWF;       // filter slice, vertex tile of this wave
        const int fo0 = (item_z(it) * NWF + vf) * 64;
        const int b = item_b(it);
        const int n0 = (item_mt(it) * NVT + vt) * 128 + 4 * c;
        const bool valid = n0 < a.Mp;
        f32x16 acc[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[t][r][j] = 0.f;

#pragma unroll 1
        for (int ks = 0; ks < nks; ++ks) {
            __builtin_amdgcn_s_barrier();                      // the producers have seen stage `cslot` land
            CG_BSTAMP(it == lw + 2 * Lw, ks);
            const unsigned xb = (unsigned)(size_t)(ring + (size_t)cslot * C::STAGE) + vt * C::XTILE + (X16 ? g * 2048 + c * 8 : g * 4096 + c * 16);
            const unsigned ab = (unsigned)(size_t)(ring + (size_t)cslot * C::STAGE) + C::XPART + (64 * vf + c) * 32 + 16 * g;
            const unsigned ab2 = ab + C::WPART;                // the lo image (PASSES == 3)
            f32x4 x[8];
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x2 xh[8];                                       // X16: rows 8g..8g+7, vertices 4c..4c+3 as bf16
            i32x4 ar[2 * C::PARTS];
            if (X16) {
                asm volatile(
                    "ds_read_b64 %0, %10\n ds_read_b64 %1, %10 offset:256\n ds_read_b64 %2, %10 offset:512\n"
                    "ds_read_b64 %3, %10 offset:768\n ds_read_b64 %4, %10 offset:1024\n ds_read_b64 %5, %10 offset:1280\n"
                    "ds_read_b64 %6, %10 offset:1536\n ds_read_b64 %7, %10 offset:1792\n"
                    "ds_read_b128 %8, %11\n ds_read_b128 %9, %11 offset:1024\n s_waitcnt lgkmcnt(0)"
                    : "=&v"(xh[0]), "=&v"(xh[1]), "=&v"(xh[2]), "=&v"(xh[3]), "=&v"(xh[4]), "=&v"(xh[5]), "=&v"(xh[6]), "=&v"(xh[7]),
                      "=&v"(ar[0]), "=&v"(ar[1])
                    : "v"(xb), "v"(ab)
                    : "memory");
            } else if (PASSES == 3) {
                asm volatile(
                    "ds_read_b128 %0, %12\n ds_read_b128 %1, %12 offset:512\n ds_read_b128 %2, %12 offset:1024\n"
                    "ds_read_b128 %3, %12 offset:1536\n ds_read_b128 %4, %12 offset:2048\n ds_read_b128 %5, %12 offset:2560\n"
                    "ds_read_b128 %6, %12 offset:3072\n ds_read_b128 %7, %12 offset:3584\n"
                    "ds_read_b128 %8, %13\n ds_read_b128 %9, %13 offset:1024\n"
                    "ds_read_b128 %10, %14\n ds_read_b128 %11, %14 offset:1024\n s_waitcnt lgkmcnt(0)"
                    : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7]),
                      "=&v"(ar[0]), "=&v"(ar[1]), "=&v"(ar[2 % (2 * C::PARTS)]), "=&v"(ar[3 % (2 * C::PARTS)])
                    : "v"(xb), "v"(ab), "v"(ab2)
                    : "memory");
            } else {
                asm volatile(
                    "ds_read_b128 %0, %10\n ds_read_b128 %1, %10 offset:512\n ds_read_b128 %2, %10 offset:1024\n"
                    "ds_read_b128 %3, %10 offset:1536\n ds_read_b128 %4, %10 offset:2048\n ds_read_b128 %5, %10 offset:2560\n"
                    "ds_read_b128 %6, %10 offset:3072\n ds_read_b128 %7, %10 offset:3584\n"
                    "ds_read_b128 %8, %11\n ds_read_b128 %9, %11 offset:1024\n s_waitcnt lgkmcnt(0)"
                    : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7]),
                      "=&v"(ar[0]), "=&v"(ar[1])
                    : "v"(xb), "v"(ab)
                    : "memory");
            }
            cslot = cslot + 1 == C::NSTAGE ? 0 : cslot + 1;
            bf16x8 ah[2], al[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                ah[t] = __builtin_bit_cast(bf16x8, ar[t]);
                if (PASSES == 3) al[t] = __builtin_bit_cast(bf16x8, ar[(2 + t) % (2 * C::PARTS)]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bf16x8 bh, bl;
                if (X16) {
                    // operand element i = row 8g + i at vertex 4c + r: half (r & 1) of dword (r >> 1) of that row's four
                    u32x4 t;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        t[j] = __builtin_amdgcn_perm(xh[2 * j + 1][r >> 1], xh[2 * j][r >> 1], (r & 1) ? 0x07060302u : 0x05040100u);
                    bh = __builtin_bit_cast(bf16x8, t);
                } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float v = x[i][r];
                    bh[i] = (__bf16)v;
                    if (PASSES == 3) bl[i] = (__bf16)(v - (float)bh[i]);
                }
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if (PASSES == 3) {
                        acc[t][r] = mfma_bf16(al[t], bh, acc[t][r]);        // small terms first
                        acc[t][r] = mfma_bf16(ah[t], bl, acc[t][r]);
                    }
                    acc[t][r] = mfma_bf16(ah[t], bh, acc[t][r]);
                }
            }
        }

        CG_BSTAMP(it == lw + 2 * Lw, 40);
        // ---- epilogue ------------------------------------------------------------------------------------------------
        // rows of this lane: fo0 + 32 t + acc_row(j, g).  The gradient wrt the stack scatters row fo to plane
        // (fo % out_K, b, fo / out_K): quotient and remainder are carried along the rows (steps of 1 and 5: one division per item)
        const int oF = a.out_K > 0 ? a.Fout / a.out_K : 0;
        int pq = 0, pr = 0;
        if (a.out_K > 0) { pq = (fo0 + 4 * g) / a.out_K; pr = (fo0 + 4 * g) - pq * a.out_K; }
        const int q1 = a.out_K > 0 ? 1 / a.out_K : 0, r1 = a.out_K > 0 ? 1 % a.out_K : 0;
        const int q5 = a.out_K > 0 ? 5 / a.out_K : 0, r5 = a.out_K > 0 ? 5 % a.out_K : 0;
        auto plane_of_row = [&]() -> long long {
            return a.out_K > 0 ? ((long long)pr * a.B + b) * oF + pq : -1;
        };
        auto next_row = [&](int d) {                               // d = 1 or 5 rows further
            if (a.out_K > 0) {
                pq += d == 1 ? q1 : q5;
                pr += d == 1 ? r1 : r5;
                if (pr >= a.out_K) { pr -= a.out_K; ++pq; }
            }
        };
        // One row of the tile.  pool == 1 (every wide layer of the reference's networks, and the gradient wrt the stack) takes a
        // lean path: fwd_epilogue_row carries every pooling variant behind run-time branches -- ~430 instructions per row in the
        // binary, and 780 cycles per row measured (phase stamps: 25k of an item's 44k cycles in the gradient wrt the stack).
        const bool lean = a.pool == 1;
        const float relu_floor = a.relu ? 0.f : -__builtin_inff();
        const size_t mrow = (size_t)(a.Mpo >> 2);
        auto emit_rows = [&](auto lean_tag) {
        constexpr bool LEAN = decltype(lean_tag)::value;
        auto put_row = [&](int t, int j, float4 bbv, bool have_bb) {
            const int fo = fo0 + 32 * t + acc_row(j, g);
            float v[4] = {acc[t][0][j], acc[t][1][j], acc[t][2][j], acc[t][3][j]};
            const long long plane = plane_of_row();
            if (!LEAN) {
                fwd_epilogue_row(a, b, fo, v, n0, valid, c, have_bb, bbv, plane);
                return;
            }
            if (a.bias_kind == CHEBGCN_BIAS_FILTER) {
                // two scalar loads (the filter of either half-wave): no entry in the vector-memory queue behind the stores
                const int fb = __builtin_amdgcn_readfirstlane(fo0 + 32 * t + acc_row(j, 0));
                const float b0 = a.bias[fb < a.Fout ? fb : a.Fout - 1], b1 = a.bias[fb + 4 < a.Fout ? fb + 4 : a.Fout - 1];
                const float bb = g ? b1 : b0;
                bbv = make_float4(bb, bb, bb, bb);
            } else if (!have_bb) {
                bbv = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            v[0] = fmaxf(v[0] + bbv.x, relu_floor); v[1] = fmaxf(v[1] + bbv.y, relu_floor);
            v[2] = fmaxf(v[2] + bbv.z, relu_floor); v[3] = fmaxf(v[3] + bbv.w, relu_floor);
            if (fo < a.Fout && valid) {
                const size_t pl = plane >= 0 ? (size_t)plane : (size_t)b * a.Fout + fo;
                if (a.out) *reinterpret_cast<float4*>(a.out + pl * a.Mpo + n0) = make_float4(v[0], v[1], v[2], v[3]);
                if (a.relu_mask)
                    a.relu_mask[((size_t)b * a.Fout + fo) * mrow + (n0 >> 2)] =
                        (uint8_t)((v[0] > 0.f ? 1 : 0) | (v[1] > 0.f ? 2 : 0) | (v[2] > 0.f ? 4 : 0) | (v[3] > 0.f ? 8 : 0));
            }
        };
        const bool ring_bias = C::RING_BIAS && a.bias_kind == CHEBGCN_BIAS_VERTEX;
        if (ring_bias) {
            // the bias rows come through the ring (see Bf16Cfg): stage 4t + i holds the rows j = 4i..4i+3 of tile t
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    __builtin_amdgcn_s_barrier();
                    const unsigned bbase = (unsigned)(size_t)(ring + (size_t)cslot * C::STAGE) + wave * 4096 + g * 2048 + c * 16;
                    f32x4 bq[4];
                    asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:512\n ds_read_b128 %2, %4 offset:1024\n"
                                 "ds_read_b128 %3, %4 offset:1536\n s_waitcnt lgkmcnt(0)"
                                 : "=&v"(bq[0]), "=&v"(bq[1]), "=&v"(bq[2]), "=&v"(bq[3]) : "v"(bbase) : "memory");
                    cslot = cslot + 1 == C::NSTAGE ? 0 : cslot + 1;
                    CG_BSTAMP(it == lw + 2 * Lw, 41 + 4 * t + i);
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        put_row(t, 4 * i + jj, make_float4(bq[jj][0], bq[jj][1], bq[jj][2], bq[jj][3]), true);
                        next_row(jj == 3 ? 5 : 1);
                    }
                }
        } else if (fo0 < a.Fout) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                // a per-vertex bias without the ring (five compute waves): four rows per round trip
                constexpr int JB = 4;
                const bool vb = a.bias_kind == CHEBGCN_BIAS_VERTEX;
#pragma unroll
                for (int j0 = 0; j0 < 16; j0 += JB) {
                    float4 bb[JB];
                    if (vb) {
#pragma unroll
                        for (int j = 0; j < JB; ++j) {
                            const int fo = fo0 + 32 * t + acc_row(j0 + j, g);
                            const int foc = fo < a.Fout ? fo : a.Fout - 1;
                            bb[j] = *reinterpret_cast<const float4*>(a.bias + (size_t)foc * a.Mp + (valid ? n0 : 0));
                        }
                    }
#pragma unroll
                    for (int j = 0; j < JB; ++j) {
                        put_row(t, j0 + j, vb ? bb[j] : make_float4(0.f, 0.f, 0.f, 0.f), vb);
                        next_row(((j0 + j) & 3) == 3 ? 5 : 1);
                    }
                }
            }
        }
        };      // emit_rows
        if (lean) emit_rows(std::true_type{}); else emit_rows(std::false_type{});
        CG_BSTAMP(it == lw + 2 * Lw, 50);
    }
}

// k-steps of 16 reduction rows
static int bf16_ksteps(int FinK) { return (FinK + 15) / 16; }

static bool check_pool_bf16(int pool, int M) {
    return pool >= 1 && pool <= 128 && (pool & (pool - 1)) == 0 && M % pool == 0;
}


// --------------------------------------------------------------------------------------------------
// Weight gradient on the bf16 matrix cores:  dW[kk][o] = sum_{b,m} stack[kk][b,m] * dy[b][o][m]
// (the MatMul gradient of models_gcn.py:616 for wide layers -- config 5 has Fin*K = 300, Fout = 256,
// where the fp32-input MFMA of contract.hip costs 1.7 ms).  Same scheme as contract_bwd_w_kernel: a
// chunk = 64 consecutive vertices of one window; its (RT + CT)*32 operand rows (256 B each) come by
// LDS-DMA with the 16-byte-piece swizzle applied on the source side; a wave owns 16 of the 64
// vertices for all RT x CT tiles: lane (row, h) reads the pieces 2q + h (q = 0, 1) of its row, eight
// consecutive-in-pairs vertices = exactly the K = 16 operand of v_mfma_f32_32x32x16_bf16 after the
// in-register fp32 -> bf16 conversion (A and B use the same pieces, and the order of the vertices
// inside a reduction is free).  CT = 2 column tiles per workgroup halve the re-reads of the stack.
// The four waves are reduced through LDS in a fixed order, every workgroup leaves one partial and two
// small kernels sum the partials in a fixed order: deterministic.
struct BwdWBf16Args {
    const float* stack; const float* dy; float* partial;
    int B, M, Mp, Fin, K, Fout, FinK;
    int nchunks_m;               // ceil(M / 64)
    size_t slab;                 // B*Fin*Mp
};
constexpr int BWB_ROW = 64;      // floats per LDS row (one chunk)
constexpr int BWB_SPLIT = 8;

// MFMA operand of one LDS row: the float4 pieces p0 and p0 + 2 (swizzled) -> eight bf16 (hi, and lo of the split)
template <int PASSES>
__device__ __forceinline__ void bwb_operand(const float* lds, int row, int p0, bool tail, int m0, int M, bf16x8& hi, bf16x8& lo) {
    float v[8];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int piece = p0 + 2 * q;
        const float4 f = *reinterpret_cast<const float4*>(lds + row * BWB_ROW + 4 * (piece ^ (row & 15)));
        v[4 * q + 0] = f.x; v[4 * q + 1] = f.y; v[4 * q + 2] = f.z; v[4 * q + 3] = f.w;
        if (tail) {
            const int n = m0 + 4 * piece;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * q + r] = (n + r < M) ? v[4 * q + r] : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        hi[i] = (__bf16)v[i];
        if (PASSES == 3) lo[i] = (__bf16)(v[i] - (float)hi[i]);
    }
}

template <int RT, int CT, int PASSES>
__global__ void __launch_bounds__(256, 2)      // two waves per SIMD = two workgroups per CU: one fetches while the other multiplies
contract_bwd_w_bf16_kernel(BwdWBf16Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [(RT+CT)*32][64]
    constexpr int NROWS = (RT + CT) * 32;
    constexpr int NDMA = NROWS / 4;                     // wave instructions per chunk (4 rows each)
    constexpr int PER_WAVE = NDMA / 4;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int tile0 = blockIdx.y * RT;                  // first row tile of this group
    const int fo0 = blockIdx.z * 32 * CT;               // first column

    f32x16 acc[RT][CT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int u = 0; u < CT; ++u)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[t][u][j] = 0.f;

    // DMA instruction n = wave + 4u covers rows 4n..4n+3: lane l -> row 4n + l/16, LDS piece l%16,
    // source piece (l%16) ^ (row%16) -- the same for every u because 16 divides the row step
    const int rsw = (lane & 15) ^ ((4 * wave + (lane >> 4)) & 15);
    const float* rsrc[PER_WAVE];
#pragma unroll
    for (int u = 0; u < PER_WAVE; ++u) {
        const int n = wave + 4 * u;
        const int row = 4 * n + (lane >> 4);
        if (n < RT * 8) {
            int kk = tile0 * 32 + row;
            if (kk >= a.FinK) kk = 0;                   // rows beyond Fin*K: dropped by the final scatter
            const int fin = kk / a.K, k = kk - fin * a.K;
            rsrc[u] = a.stack + (size_t)k * a.slab + (size_t)fin * a.Mp + 4 * rsw;
        } else {
            int fo = fo0 + (row - RT * 32);
            if (fo >= a.Fout) fo = 0;
            rsrc[u] = a.dy + (size_t)fo * a.Mp + 4 * rsw;
        }
    }

    const int total = a.B * a.nchunks_m;
    for (int ch = blockIdx.x; ch < total; ch += gridDim.x) {
        const int b = ch / a.nchunks_m;
        const int m0 = (ch - b * a.nchunks_m) * 64;
        const ptrdiff_t mo = (m0 + 4 * rsw < a.Mp) ? m0 : -4 * rsw;   // beyond the plane: any valid address, masked below
        const ptrdiff_t so = (ptrdiff_t)b * a.Fin * a.Mp + mo, dof = (ptrdiff_t)b * a.Fout * a.Mp + mo;
        __syncthreads();                                // previous chunk's operand reads are done
#pragma unroll
        for (int u = 0; u < PER_WAVE; ++u) {
            const int n = wave + 4 * u;
            const float* src = rsrc[u] + (n < RT * 8 ? so : dof);
            __builtin_amdgcn_global_load_lds(src, lds + (size_t)n * 4 * BWB_ROW, 16, 0, 2);   // aux 2 = nt
        }
        __syncthreads();                                // DMA landed (the barrier's release waits vmcnt(0))

        const bool tail = m0 + 64 > a.M;                // only the last chunk of a plane has vertices to mask
        bf16x8 bh[CT], bl[CT];
#pragma unroll
        for (int u = 0; u < CT; ++u) bwb_operand<PASSES>(lds, RT * 32 + u * 32 + c, 4 * wave + h, tail, m0, a.M, bh[u], bl[u]);
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            bf16x8 ah, al;
            bwb_operand<PASSES>(lds, t * 32 + c, 4 * wave + h, tail, m0, a.M, ah, al);
#pragma unroll
            for (int u = 0; u < CT; ++u) {
                acc[t][u] = mfma_bf16(ah, bh[u], acc[t][u]);
                if (PASSES == 3) {
                    acc[t][u] = mfma_bf16(ah, bl[u], acc[t][u]);
                    acc[t][u] = mfma_bf16(al, bh[u], acc[t][u]);
                }
            }
        }
    }

    // ---- workgroup reduction in LDS (fixed order: wave 0, then +1, +2, +3) ----------------
    constexpr int per = RT * CT * 16 * 64;
    static_assert(per <= NROWS * BWB_ROW, "the reduction image must fit the operand buffer");
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int u = 0; u < CT; ++u)
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        float* p = lds + ((t * CT + u) * 16 + j) * 64 + lane;
                        *p = (w == 0) ? acc[t][u][j] : *p + acc[t][u][j];
                    }
        }
    }
    __syncthreads();
    float* dst = a.partial + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * per;
    for (int o = threadIdx.x; o < per; o += 256) dst[o] = lds[o];
}

// --------------------------------------------------------------------------------------------------
// The same weight gradient for WIDE layers (Fin*K > 160 or Fout > 64) in ONE pass over the operands:
// a workgroup of 8 waves owns a 320 x 256 tile of dW -- all of it at config 5 (300 x 256) -- so the
// stack and dy are each read once (the 5x2-tile kernel above reads the stack 4 and dy 2 times there).
// Wave (wy, wz) keeps the 5 x 2 tiles (rows 160 wy.., columns 64 wz..) in 160 accumulator registers;
// no two waves share a tile, so there is no cross-wave reduction.
// Operand ring: 4 buffers x [576 rows][16 vertices] fp32 (36 KB each) filled by LDS-DMA, three chunks
// in flight while one is multiplied: one k-step of v_mfma_f32_32x32x16_bf16 per chunk.  A row is 64 B =
// four 16-byte pieces, piece p of row r at position p ^ ((r >> 2) & 3) (applied on the source side of
// the DMA), which makes the operand reads (16 lanes = 16 rows, same logical piece) conflict free.
// A workgroup walks a CONTIGUOUS range of (window, 16-vertex) chunks, so the second half of every
// 128-byte line it touches is an L2 hit a moment later.  One barrier per chunk; DMA waits are counted
// by hand and the operands are read in asm blocks (compiler-visible LDS reads would wait for every DMA
// in flight).
//
// DY16: dy arrives as bf16 [B][Fout][Mp] (chebgcn_relu_grad_bf16 wrote it: the one-pass kernel rounds dy to bf16 anyway, so the
// results are bit-identical) -- half the bytes of the larger operand.  Its ring rows are 32 B (16 vertices): eight DMA
// instructions per chunk instead of sixteen, piece p of row r at position p ^ ((r >> 3) & 1), and the 16-byte operand read IS
// the matrix operand (no conversion).
// a ring buffer: 320 stack rows + 256 dy rows
constexpr int BWW_DYOFF = 320 * 64;              // byte offset of the dy rows in a ring buffer
template <bool DY16> constexpr int bww_buf() { return BWW_DYOFF + 256 * (DY16 ? 32 : 64); }   // bytes per ring buffer
constexpr int BWW_BUF = bww_buf<false>();
constexpr int BWW_NBUF = 4;

template <int PASSES, bool DY16 = false>
__global__ void __launch_bounds__(512)
contract_bwd_w_bf16_wide_kernel(BwdWBf16Args a, int total_chunks) {
    static_assert(!DY16 || PASSES == 1, "a bf16 dy has no low part");
    constexpr int BUF = bww_buf<DY16>();
    constexpr int NINS = DY16 ? 28 : 36;                          // DMA instructions per chunk (1 KB each)
    constexpr int NU = DY16 ? 4 : 5;                              // ... per wave at most
    extern __shared__ __attribute__((aligned(16))) char ring[];   // [BWW_NBUF][BUF]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wy = wave >> 2, wz = wave & 3;
    const int c = lane & 31, g = lane >> 5;
    const int row0 = blockIdx.y * 320, col0 = blockIdx.z * 256;

    f32x16 acc[5][2];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[t][u][j] = 0.f;

    // DMA instruction n = wave + 8u (n < 36) fills ring rows 16n..16n+15: lane l -> row 16n + l/4,
    // position l%4, source piece (l%4) ^ ((row >> 2) & 3) = (l%4) ^ ((l >> 4) & 3) for every n
    // (DY16: instruction n >= 20 fills the dy rows 32(n-20)..+31: lane l -> row 32(n-20) + l/2, position l%2, source piece
    // (l%2) ^ ((row >> 3) & 1) = (l%2) ^ ((l >> 4) & 1))
    const int spiece = (lane & 3) ^ ((lane >> 4) & 3);
    const float* rsrc[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int n = wave + 8 * u;
        const int R = 16 * n + (lane >> 2);
        if (n < 20) {
            int kk = row0 + R;
            if (kk >= a.FinK) kk = 0;                   // rows beyond Fin*K: dropped by the final scatter
            const int fin = kk / a.K, k = kk - fin * a.K;
            rsrc[u] = a.stack + (size_t)k * a.slab + (size_t)fin * a.Mp + 4 * spiece;
        } else if (DY16) {
            int fo = col0 + 32 * (n - 20) + (lane >> 1);
            if (fo >= a.Fout || n >= NINS) fo = 0;
            // (addresses in units of float: a bf16 plane is Mp/2 floats long, a piece of eight vertices four floats)
            rsrc[u] = a.dy + ((size_t)fo * a.Mp >> 1) + 4 * ((lane & 1) ^ ((lane >> 4) & 1));
        } else {
            int fo = col0 + (R - 320);
            if (fo >= a.Fout || n >= NINS) fo = 0;
            rsrc[u] = a.dy + (size_t)fo * a.Mp + 4 * spiece;
        }
    }
    const int ch0 = (int)((long long)total_chunks * blockIdx.x / gridDim.x);
    const int ch1 = (int)((long long)total_chunks * (blockIdx.x + 1) / gridDim.x);
    auto issue = [&](int ch, int slot) {
        const int b = ch / a.nchunks_m;
        const int m0 = (ch - b * a.nchunks_m) * 16;
        const size_t so = (size_t)b * a.Fin * a.Mp + m0;
        const size_t dof = DY16 ? ((size_t)b * a.Fout * a.Mp + m0) >> 1 : (size_t)b * a.Fout * a.Mp + m0;
        const unsigned base = (unsigned)(size_t)ring + slot * BUF;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int n = wave + 8 * u;
            if (n < NINS) {
                const float* src = rsrc[u] + (n < 20 ? so : dof);
                __builtin_amdgcn_global_load_lds(src, reinterpret_cast<__attribute__((address_space(3))) void*>(base + n * 1024),
                                                 16, 0, 0);
            }
        }
    };
    if (ch0 >= ch1) goto flush;                          // (more workgroups than chunks)
    {
    // prologue: three chunks in flight (past the end of the range: harmless re-reads of the last chunk)
#pragma unroll 1
    for (int d = 0; d < BWW_NBUF - 1; ++d) issue(ch0 + d < ch1 ? ch0 + d : ch1 - 1, d);
    // operand addresses of this lane inside a buffer: row tile * 2048 + c * 64 + 16 * (piece ^ swizzle)
    const int sw = (c >> 2) & 3;
    const unsigned offA0 = (wy * 5) * 2048 + c * 64 + 16 * ((2 * g) ^ sw);
    const unsigned offA1 = (wy * 5) * 2048 + c * 64 + 16 * ((2 * g + 1) ^ sw);
    const unsigned offB0 = DY16 ? BWW_DYOFF + (wz * 2) * 1024 + c * 32 + 16 * (g ^ ((c >> 3) & 1))
                                : BWW_DYOFF + (wz * 2) * 2048 + c * 64 + 16 * ((2 * g) ^ sw);
    const unsigned offB1 = BWW_DYOFF + (wz * 2) * 2048 + c * 64 + 16 * ((2 * g + 1) ^ sw);

    int slot = 0;
#pragma unroll 1
    for (int ch = ch0; ch < ch1; ++ch) {
        // this wave's part of chunk `ch` is (BWW_NBUF-2) chunks of DMAs old
        if (wave < 4) wait_vmcnt<2 * NU>(); else wait_vmcnt<2 * (NU - 1)>();
        __builtin_amdgcn_s_barrier();                    // everybody's part landed; everybody is done with the buffer refilled next
        {
            const int nx = ch + BWW_NBUF - 1;
            issue(nx < ch1 ? nx : ch1 - 1, (slot + BWW_NBUF - 1) & (BWW_NBUF - 1));
        }
        const unsigned buf = (unsigned)(size_t)ring + slot * BUF;
        slot = (slot + 1) & (BWW_NBUF - 1);
        const int m0 = (ch % a.nchunks_m) * 16;
        const bool tail = m0 + 16 > a.M;
        const int nlive = a.M - m0 - 8 * g;             // vertices of this lane's eight that exist (tail chunks)

        f32x4 rb[4];
        if (DY16) {
            const unsigned b0 = buf + offB0;
            asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:1024\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(rb[0]), "=&v"(rb[1]) : "v"(b0) : "memory");
        } else {
            const unsigned b0 = buf + offB0, b1 = buf + offB1;
            asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %5\n ds_read_b128 %2, %4 offset:2048\n"
                         "ds_read_b128 %3, %5 offset:2048\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(rb[0]), "=&v"(rb[1]), "=&v"(rb[2]), "=&v"(rb[3]) : "v"(b0), "v"(b1) : "memory");
        }
        auto split = [&](f32x4 lo4, f32x4 hi4, bf16x8& hi, bf16x8& lo) {
            float v[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
            if (tail) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = i < nlive ? v[i] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                hi[i] = (__bf16)v[i];
                if (PASSES == 3) lo[i] = (__bf16)(v[i] - (float)hi[i]);
            }
        };
        bf16x8 bh[2], bl[2];
        if (DY16) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                bh[u] = __builtin_bit_cast(bf16x8, rb[u]);
                if (tail) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) bh[u][i] = i < nlive ? bh[u][i] : (__bf16)0.f;
                }
            }
        } else {
            split(rb[0], rb[1], bh[0], bl[0]);
            split(rb[2], rb[3], bh[1], bl[1]);
        }

        f32x4 ra[10];
        {
            const unsigned a0 = buf + offA0, a1 = buf + offA1;
            asm volatile("ds_read_b128 %0, %10\n ds_read_b128 %1, %11\n ds_read_b128 %2, %10 offset:2048\n ds_read_b128 %3, %11 offset:2048\n"
                         "ds_read_b128 %4, %10 offset:4096\n ds_read_b128 %5, %11 offset:4096\n ds_read_b128 %6, %10 offset:6144\n"
                         "ds_read_b128 %7, %11 offset:6144\n ds_read_b128 %8, %10 offset:8192\n ds_read_b128 %9, %11 offset:8192\n"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(ra[0]), "=&v"(ra[1]), "=&v"(ra[2]), "=&v"(ra[3]), "=&v"(ra[4]), "=&v"(ra[5]), "=&v"(ra[6]),
                           "=&v"(ra[7]), "=&v"(ra[8]), "=&v"(ra[9])
                         : "v"(a0), "v"(a1) : "memory");
        }
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            bf16x8 ah, al;
            split(ra[2 * t], ra[2 * t + 1], ah, al);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (PASSES == 3) {
                    acc[t][u] = mfma_bf16(al, bh[u], acc[t][u]);        // small terms first
                    acc[t][u] = mfma_bf16(ah, bl[u], acc[t][u]);
                }
                acc[t][u] = mfma_bf16(ah, bh[u], acc[t][u]);
            }
        }
    }
    wait_vmcnt<0>();                                     // the run-ahead DMAs
    }
flush:
    // every wave owns its tiles: the partial goes straight to the workspace in accumulator layout,
    // rows ordered (row tile, column tile, register) as bwb_reduce_stage2 expects for rt = 10, ct = 8
    float* dst = a.partial + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (size_t)(10 * 8 * 16 * 64);
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < 16; ++j)
                dst[(((wy * 5 + t) * 8 + (wz * 2 + u)) * 16 + j) * 64 + lane] = acc[t][u][j];
}

// partial: [Z][Y][X][rows*64] raw accumulator images (rows = RT*CT*16).  Stage 1: block
// (row, y*S + s, z) sums the partials x = s, s+S, ... of one 64-lane accumulator row.
__global__ void __launch_bounds__(256)
bwb_reduce_stage1(const float* __restrict__ partial, float* __restrict__ stage, int nx, int ny, int rows) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const size_t per = (size_t)rows * 64;
    const int row = blockIdx.x;
    const int y = blockIdx.y / BWB_SPLIT, sp = blockIdx.y % BWB_SPLIT, z = blockIdx.z;
    const float* base = partial + ((size_t)z * ny + y) * nx * per + (size_t)row * 64 + lane;
    float s = 0.f;
    for (int x0 = sp + BWB_SPLIT * part; x0 < nx; x0 += 8 * 4 * BWB_SPLIT) {      // eight partials in flight, added in index order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int x = x0 + u * 4 * BWB_SPLIT;
            v[u] = base[(size_t)(x < nx ? x : x0) * per];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (x0 + u * 4 * BWB_SPLIT < nx) s += v[u];
    }
    red[part][lane] = s;
    __syncthreads();
    if (part == 0)
        stage[((((size_t)z * ny + y) * BWB_SPLIT + sp) * rows + row) * 64 + lane] =
            ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
}

// Stage 2: sum the BWB_SPLIT stage rows and scatter from the accumulator layout to dW[kk][o].
__global__ void __launch_bounds__(64)
bwb_reduce_stage2(const float* __restrict__ stage, float* __restrict__ dW, int ny, int rt, int ct, int FinK, int Fout) {
    const int lane = threadIdx.x;
    const int row = blockIdx.x, y = blockIdx.y, z = blockIdx.z;
    const int rows = rt * ct * 16;
    float s = 0.f;
    for (int sp = 0; sp < BWB_SPLIT; ++sp)
        s += stage[((((size_t)z * ny + y) * BWB_SPLIT + sp) * rows + row) * 64 + lane];
    const int tu = row >> 4, j = row & 15, h = lane >> 5;
    const int t = tu / ct, u = tu - t * ct;
    const int kk = (y * rt + t) * 32 + acc_row(j, h);
    const int fo = (z * ct + u) * 32 + (lane & 31);
    if (kk < FinK && fo < Fout) dW[(size_t)kk * Fout + fo] = s;
}

struct BwbPlan { int rt, ct, gx, gy, gz; size_t per; bool wide; };
static BwbPlan bwb_plan(int B, int M, int Fin, int K, int Fout) {
    static int cus = 0;                       // cached: the attribute query is slow
    if (cus == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
    }
    BwbPlan p;
    p.wide = Fin * K > 160 && Fout > 64;      // one pass over the operands: 320 x 256 tile per workgroup
    if (p.wide) {
        p.rt = 10; p.ct = 8;
        p.gy = (Fin * K + 319) / 320;
        p.gz = (Fout + 255) / 256;
        const long long total = (long long)B * ((M + 15) / 16);
        long long gx = cus / (p.gy * p.gz);   // 144 KB of LDS: one workgroup per CU
        if (const char* e = getenv("CHEBGCN_BWB_WIDE_GX")) gx = atoll(e) > 0 ? atoll(e) : gx;       // (experiment knob)
        if (gx > total) gx = total;
        p.gx = gx < 1 ? 1 : (int)gx;
        p.per = (size_t)10 * 8 * 16 * 64;
        return p;
    }
    const int ntiles = (Fin * K + 31) / 32;
    p.rt = ntiles < 5 ? ntiles : 5;
    p.ct = Fout > 32 ? 2 : 1;
    p.gy = (ntiles + p.rt - 1) / p.rt;
    p.gz = (Fout + 32 * p.ct - 1) / (32 * p.ct);
    const long long total = (long long)B * ((M + 63) / 64);
    // Two workgroups per CU over the WHOLE launch (rounds 2-5: 2 * cus per row-tile group): every workgroup leaves a partial of
    // its tile group that two reduce kernels read back -- at 3200 chunks (the 3168-vertex level of the pooling network) the
    // partials of 512 workgroups per group were 38 % of the operands' bytes -- and groups beside each other share the CUs.
    // Measured, same box (EXPERIMENTS 8.6; gx per group 512 / 256 / 128): 32*10 -> 64 (two groups) 0.229 / 0.205 / 0.302 ms,
    // 64*10 -> 64 (four) 0.467 / 0.408 / 0.376, 64*25 -> 64 (ten) 1.116 / 1.058 / 1.062 (64: 1.31) at M = 10466; 0.131 / 0.118 / 0.120
    // at M = 3168.  A multiple of 128 (133 and 266 lose a quarter: the chunk -> workgroup stride meets the channel interleave),
    // between half the CUs and twice the CUs per group.
    long long gx = (2ll * cus / ((long long)p.gy * p.gz) + 127) / 128 * 128;
    gx = std::max<long long>(cus / 2, std::min<long long>(gx, 2ll * cus));
    // Eight groups and more (64*25 -> 64: ten): every group re-reads dy -- 5.81 GB moved against 4.46 algorithmic with 128 workgroups
    // per group (EXPERIMENTS 8.9) -- unless all groups walk the same chunks at the same time: 2 * CUs workgroups over the WHOLE
    // launch, all resident at once, the groups of a chunk range side by side in time (gx per group 128 / 96 / 51 / 48 / 40 / 32:
    // 1.183 / 1.127 / 1.105 / 1.117 / 1.176 / 1.292 ms)
    if (p.gy * p.gz >= 8) gx = std::max<long long>(1, 2ll * cus / ((long long)p.gy * p.gz));
    if (const char* e = getenv("CHEBGCN_BWB_GX")) gx = atoll(e) > 0 ? atoll(e) : gx;                 // (experiment knob)
    if (gx > total) gx = total;
    p.gx = gx < 1 ? 1 : (int)gx;
    p.per = (size_t)p.rt * p.ct * 16 * 64;
    return p;
}

}  // namespace chebgcn

using namespace chebgcn;

static int launch_contract_bf16(const FwdArgs& a, int passes, void* workspace, hipStream_t stream, int nw, bool x16 = false,
                                int nvt = 1);
// vertex tiles per work item of the forward: the four compute waves are 4 / nvt filter slices of 64
static int fwd_bf16_tiles(int Fout) { return Fout <= 64 ? 4 : Fout <= 128 ? 2 : 1; }

extern "C" size_t chebgcn_contract_fwd_bf16_workspace(int Fin, int K, int Fout) {
    if (Fin <= 0 || K <= 0 || Fout <= 0) return 0;
    const size_t nks = bf16_ksteps(Fin * K), FoutP = ((size_t)Fout + 255) / 256 * 256;
    return 2 * nks * FoutP * 16 * sizeof(uint16_t);      // hi and lo images (narrow layers pack 64 or 128 columns: less)
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
    a.relu_mask = (pool == 1 && relu) ? argmax : nullptr;
    a.B = B; a.M = M; a.Mp = plane_stride(M); a.Fin = Fin; a.K = K; a.Fout = Fout; a.FinK = Fin * K;
    a.pool = pool; a.pool_kind = pool_kind; a.relu = relu; a.bias_kind = bias_kind;
    a.Mo = M / pool; a.Mpo = plane_stride(a.Mo);
    a.slab = (size_t)B * Fin * a.Mp;
    const int nvt = fwd_bf16_tiles(Fout), G = 256 / nvt;
    const int nks = bf16_ksteps(a.FinK), FoutP = (Fout + G - 1) / G * G;
    note_dispatch("pack_w_bf16_kernel");
    hipLaunchKernelGGL(pack_w_bf16_kernel, dim3((nks * FoutP + 255) / 256), dim3(256), 0, stream, W, (__bf16*)workspace,
                       a.FinK, Fout, nks, FoutP, passes == 3 ? 2 : 1, 0);
    CG_HIP(hipGetLastError());
    return launch_contract_bf16(a, passes, workspace, stream, 4, false, nvt);
}

// the packed operand is in `workspace`; `a` describes the rows, the planes and the epilogue
static int launch_contract_bf16(const FwdArgs& a, int passes, void* workspace, hipStream_t stream, int nw, bool x16, int nvt) {
    const int B = a.B, M = a.M, Fout = a.Fout;
    const int G = nw / nvt * 64;                         // filters per work item
    const int nks = bf16_ksteps(a.FinK), FoutP = (Fout + G - 1) / G * G;
    const int ntm = (M + 128 * nvt - 1) / (128 * nvt);   // vertex tiles of a work item's size
    const int64_t nitems64 = (int64_t)ntm * B * (FoutP / G);
    CG_REQUIRE(nitems64 < (1ll << 31), "contract_fwd_bf16: too many tiles");
    const int nitems = (int)nitems64;
    static int cus = 0;                                  // cached: the attribute query is slow
    if (cus == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
    }

#define CG_BF16_LAUNCH_XT(P, NW, X, T, TAG)                                                                        \
    do {                                                                                                           \
        constexpr int lds = Bf16Cfg<P, NW, X, T>::NSTAGE * Bf16Cfg<P, NW, X, T>::STAGE;                            \
        const dim3 grid(nitems < cus ? nitems : cus);           /* one persistent workgroup per CU */              \
        note_dispatch_more("contract_fwd_bf16_kernel<" #P "," #NW TAG ">");                                        \
        CG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(contract_fwd_bf16_kernel<P, NW, X, T>),           \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds));                              \
        hipLaunchKernelGGL((contract_fwd_bf16_kernel<P, NW, X, T>), grid, dim3((NW + 2) * 64), lds, stream, a,     \
                           (const __bf16*)workspace, nks, FoutP, ntm, nitems);                                     \
    } while (0)
#define CG_BF16_LAUNCH_X(P, NW, X, TAG) CG_BF16_LAUNCH_XT(P, NW, X, 1, TAG)
#define CG_BF16_LAUNCH(P, NW) CG_BF16_LAUNCH_X(P, NW, false, "")
    if (nvt == 4) {                                      // (the forward of layers of at most 64 filters: four vertex tiles per item)
        if (passes == 3) CG_BF16_LAUNCH_XT(3, 4, false, 4, ",tiles4"); else CG_BF16_LAUNCH_XT(1, 4, false, 4, ",tiles4");
    } else if (nvt == 2) {                               // (at most 128 filters: two slices x two tiles)
        if (passes == 3) CG_BF16_LAUNCH_XT(3, 4, false, 2, ",tiles2"); else CG_BF16_LAUNCH_XT(1, 4, false, 2, ",tiles2");
    } else if (x16) {
        if (nw == 5) CG_BF16_LAUNCH_X(1, 5, true, ",x16"); else CG_BF16_LAUNCH_X(1, 4, true, ",x16");
    } else if (nw == 5) {
        if (passes == 3) CG_BF16_LAUNCH(3, 5); else CG_BF16_LAUNCH(1, 5);
    } else {
        if (passes == 3) CG_BF16_LAUNCH(3, 4); else CG_BF16_LAUNCH(1, 4);
    }
#undef CG_BF16_LAUNCH
#undef CG_BF16_LAUNCH_X
#undef CG_BF16_LAUNCH_XT
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

// ---- gradient of the contraction wrt the stack on the bf16 matrix cores -------------------------
// gstack[k][b][fin][m] = sum_o W[fin*K+k][o] * dy[b][o][m] (the MatMul gradient of
// models_gcn.py:616): the SAME kernel with the roles swapped -- the reduction rows are the Fout planes
// of dy (a "stack" of one slab), the operand is W^T packed in fragment order, and the Fin*K output
// rows are scattered into the slab layout of the gradient stack by the epilogue (FwdArgs::out_K).
// waves per workgroup for `rows` output rows: five (groups of 320) where that pads less than four (256)
static int bwd_x_bf16_waves(int rows) { return (rows + 319) / 320 * 320 < (rows + 255) / 256 * 256 ? 5 : 4; }

extern "C" size_t chebgcn_contract_bwd_x_bf16_workspace(int Fin, int K, int Fout) {
    if (Fin <= 0 || K <= 0 || Fout <= 0) return 0;
    const size_t G = 64 * (size_t)bwd_x_bf16_waves(Fin * K);
    const size_t nks = bf16_ksteps(Fout), RowsP = ((size_t)Fin * K + G - 1) / G * G;
    return 2 * nks * RowsP * 16 * sizeof(uint16_t);
}

static int bwd_x_bf16_impl(const float* dy, bool dy16, const float* W, float* gstack, int B, int M, int Fin, int K, int Fout,
                           int passes, void* workspace, size_t workspace_bytes, hipStream_t stream);

extern "C" int chebgcn_contract_bwd_x_bf16(const float* dy, const float* W, float* gstack, int B, int M, int Fin, int K,
                                           int Fout, int passes, void* workspace, size_t workspace_bytes,
                                           chebgcn_stream stream_) {
    return bwd_x_bf16_impl(dy, false, W, gstack, B, M, Fin, K, Fout, passes, workspace, workspace_bytes, (hipStream_t)stream_);
}

extern "C" int chebgcn_contract_bwd_x_bf16_dy16(const uint16_t* dy16, const float* W, float* gstack, int B, int M, int Fin, int K,
                                                int Fout, void* workspace, size_t workspace_bytes, chebgcn_stream stream_) {
    return bwd_x_bf16_impl(reinterpret_cast<const float*>(dy16), true, W, gstack, B, M, Fin, K, Fout, 1, workspace, workspace_bytes,
                           (hipStream_t)stream_);
}

static int bwd_x_bf16_impl(const float* dy, bool dy16, const float* W, float* gstack, int B, int M, int Fin, int K, int Fout,
                           int passes, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    CG_REQUIRE(dy && W && gstack && workspace, "contract_bwd_x_bf16: NULL argument");
    CG_REQUIRE(B > 0 && M > 0 && Fin > 0 && K > 0 && Fout > 0 && B <= 65535, "contract_bwd_x_bf16: bad shape");
    CG_REQUIRE(passes == 1 || passes == 3, "contract_bwd_x_bf16: passes must be 1 (bf16) or 3 (split bf16), got %d", passes);
    CG_REQUIRE(workspace_bytes >= chebgcn_contract_bwd_x_bf16_workspace(Fin, K, Fout),
               "contract_bwd_x_bf16: workspace of %zu bytes is too small", workspace_bytes);
    FwdArgs a;
    a.stack = dy; a.W = W; a.bias = nullptr; a.out = gstack; a.argmax = nullptr;
    a.B = B; a.M = M; a.Mp = plane_stride(M);
    a.Fin = Fout; a.K = 1; a.FinK = Fout;               // reduction rows: the Fout planes of one window of dy
    a.Fout = Fin * K;                                   // output rows
    a.pool = 1; a.pool_kind = 0; a.relu = 0; a.bias_kind = CHEBGCN_BIAS_NONE;
    a.Mo = M; a.Mpo = a.Mp;
    a.slab = (size_t)B * Fout * a.Mp;
    a.out_K = K;
    const int nw = bwd_x_bf16_waves(a.Fout), G = 64 * nw;
    const int nks = bf16_ksteps(a.FinK), FoutP = (a.Fout + G - 1) / G * G;
    note_dispatch("pack_w_bf16_kernel<transposed>");
    hipLaunchKernelGGL(pack_w_bf16_kernel, dim3((nks * FoutP + 255) / 256), dim3(256), 0, stream, W, (__bf16*)workspace,
                       a.FinK, a.Fout, nks, FoutP, passes == 3 ? 2 : 1, Fout);
    CG_HIP(hipGetLastError());
    return launch_contract_bf16(a, passes, workspace, stream, nw, dy16);
}

extern "C" size_t chebgcn_contract_bwd_w_bf16_workspace(int B, int M, int Fin, int K, int Fout) {
    if (B <= 0 || M <= 0 || Fin <= 0 || K <= 0 || Fout <= 0) return 0;
    const BwbPlan p = bwb_plan(B, M, Fin, K, Fout);
    return ((size_t)p.gx + BWB_SPLIT) * p.gy * p.gz * p.per * sizeof(float);   // partials + stage
}

static int bwd_w_bf16_impl(const float* stack, const float* dy, bool dy16, float* dW, void* workspace, size_t workspace_bytes,
                           int B, int M, int Fin, int K, int Fout, int passes, hipStream_t stream);

extern "C" int chebgcn_contract_bwd_w_bf16(const float* stack, const float* dy, float* dW, void* workspace,
                                           size_t workspace_bytes, int B, int M, int Fin, int K, int Fout, int passes,
                                           chebgcn_stream stream_) {
    return bwd_w_bf16_impl(stack, dy, false, dW, workspace, workspace_bytes, B, M, Fin, K, Fout, passes, (hipStream_t)stream_);
}

extern "C" int chebgcn_bf16_dy16_supported(int B, int M, int Fin, int K, int Fout) {
    if (B <= 0 || M <= 0 || Fin <= 0 || K <= 0 || Fout <= 0) return 0;
    return bwb_plan(B, M, Fin, K, Fout).wide ? 1 : 0;
}

extern "C" int chebgcn_contract_bwd_w_bf16_dy16(const float* stack, const uint16_t* dy16, float* dW, void* workspace,
                                                size_t workspace_bytes, int B, int M, int Fin, int K, int Fout,
                                                chebgcn_stream stream_) {
    CG_REQUIRE(chebgcn_bf16_dy16_supported(B, M, Fin, K, Fout),
               "contract_bwd_w_bf16_dy16: only for wide layers (Fin*K > 160 and Fout > 64), got Fin*K=%d Fout=%d", Fin * K, Fout);
    return bwd_w_bf16_impl(stack, reinterpret_cast<const float*>(dy16), true, dW, workspace, workspace_bytes, B, M, Fin, K, Fout, 1,
                           (hipStream_t)stream_);
}

static int bwd_w_bf16_impl(const float* stack, const float* dy, bool dy16, float* dW, void* workspace, size_t workspace_bytes,
                           int B, int M, int Fin, int K, int Fout, int passes, hipStream_t stream) {
    CG_REQUIRE(stack && dy && dW && workspace, "contract_bwd_w_bf16: NULL argument");
    CG_REQUIRE(B > 0 && M > 0 && Fin > 0 && K > 0 && Fout > 0, "contract_bwd_w_bf16: bad shape");
    CG_REQUIRE(passes == 1 || passes == 3, "contract_bwd_w_bf16: passes must be 1 (bf16) or 3 (split bf16), got %d", passes);
    CG_REQUIRE(workspace_bytes >= chebgcn_contract_bwd_w_bf16_workspace(B, M, Fin, K, Fout),
               "contract_bwd_w_bf16: workspace too small");
    const BwbPlan p = bwb_plan(B, M, Fin, K, Fout);
    BwdWBf16Args a;
    a.stack = stack; a.dy = dy; a.partial = (float*)workspace;
    a.B = B; a.M = M; a.Mp = plane_stride(M); a.Fin = Fin; a.K = K; a.Fout = Fout; a.FinK = Fin * K;
    a.nchunks_m = (M + 63) / 64;
    a.slab = (size_t)B * Fin * a.Mp;
    const dim3 grid(p.gx, p.gy, p.gz);
    if (p.wide) {
        a.nchunks_m = (M + 15) / 16;
        const size_t ldsw = (size_t)BWW_NBUF * BWW_BUF;
        const long long total = (long long)B * a.nchunks_m;
        CG_REQUIRE(total < (1ll << 31), "contract_bwd_w_bf16: too many chunks");
        note_dispatch(dy16 ? "contract_bwd_w_bf16_wide_kernel<1,dy16>"
                           : passes == 3 ? "contract_bwd_w_bf16_wide_kernel<3>" : "contract_bwd_w_bf16_wide_kernel<1>");
        if (dy16) {
            const size_t lds16 = (size_t)BWW_NBUF * bww_buf<true>();
            CG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(contract_bwd_w_bf16_wide_kernel<1, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16));
            hipLaunchKernelGGL((contract_bwd_w_bf16_wide_kernel<1, true>), grid, dim3(512), lds16, stream, a, (int)total);
        } else if (passes == 3) {
            CG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(contract_bwd_w_bf16_wide_kernel<3>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw));
            hipLaunchKernelGGL((contract_bwd_w_bf16_wide_kernel<3>), grid, dim3(512), ldsw, stream, a, (int)total);
        } else {
            CG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(contract_bwd_w_bf16_wide_kernel<1>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw));
            hipLaunchKernelGGL((contract_bwd_w_bf16_wide_kernel<1>), grid, dim3(512), ldsw, stream, a, (int)total);
        }
    }
    const size_t lds = (size_t)(p.rt + p.ct) * 32 * BWB_ROW * sizeof(float);
#define CG_BWB(R, C, P)                                                                                   \
    if (p.rt == R && p.ct == C && passes == P) {                                                          \
        note_dispatch("contract_bwd_w_bf16_kernel<" #R "," #C "," #P ">");                                \
        CG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(contract_bwd_w_bf16_kernel<R, C, P>),    \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                \
        hipLaunchKernelGGL((contract_bwd_w_bf16_kernel<R, C, P>), grid, dim3(256), lds, stream, a);       \
    }
#define CG_BWB4(R) CG_BWB(R, 1, 1) CG_BWB(R, 1, 3) CG_BWB(R, 2, 1) CG_BWB(R, 2, 3)
    CG_BWB4(1) CG_BWB4(2) CG_BWB4(3) CG_BWB4(4) CG_BWB4(5)
#undef CG_BWB4
#undef CG_BWB
    CG_HIP(hipGetLastError());
    const int rows = p.rt * p.ct * 16;
    float* stage = (float*)workspace + (size_t)p.gx * p.gy * p.gz * p.per;
    note_dispatch_more("bwb_reduce_stage1");
    note_dispatch_more("bwb_reduce_stage2");
    hipLaunchKernelGGL(bwb_reduce_stage1, dim3(rows, p.gy * BWB_SPLIT, p.gz), dim3(256), 0, stream,
                       (const float*)workspace, stage, p.gx, p.gy, rows);
    hipLaunchKernelGGL(bwb_reduce_stage2, dim3(rows, p.gy, p.gz), dim3(64), 0, stream, (const float*)stage, dW, p.gy,
                       p.rt, p.ct, a.FinK, Fout);
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

#ifdef CG_EXPERIMENT
extern "C" int chebgcn_debug_stampsb(long long* out) {      // CG_X & 64 builds only (tools/kbench.py --stamps)
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(chebgcn::g_dbgb), sizeof(long long) * 16 * 64) == hipSuccess ? 0 : -1;
}
#endif
