// Atlas-sized graphs (the reference's own: 246..360 brain regions, configure_fmri.py:11; M <= 384 vertices):
// one Chebyshev layer per launch, ON CHIP.
//
//   forward  (lib_new/models_gcn.py:587-629):  T_0 = x, T_1 = L T_0, T_k = 2 L T_{k-1} - T_{k-2},
//                                              y = act( sum_k T_k W_k + bias )          -- recurrence AND contraction
//   backward wrt the input (TF autodiff of the above): G_j = dy W_j^T,
//                                              c_{K-1} = G_{K-1}, c_j = G_j + 2 L^T c_{j+1} - c_{j+2}, dx = G_0 + L^T c_1 - c_2
//
// On the benchmark graph (10466 vertices) a window's T_k is 1.3 MB and the layer is three HBM streams (recurrence,
// contraction, their gradients).  At the sizes the reference actually trains on, a whole window -- M vertices x 32 planes =
// 48 KB at M = 376 -- fits one CU's LDS, and the separate kernels are launch- and latency-bound (round 3: recurrence 30 us
// + contraction 22 us per layer forward for 63 MB of stack, each 2-3x its streaming time; 0.24 of the HBM roofline for the
// step).  Here a workgroup carries (half of) a window through the layer:
//   * lane (c, h) of wave w owns vertex v = 32 w + c and PL planes of it: T_{k-1}, T_{k-2} of its vertex in registers,
//     T_{k-1} of every vertex in an LDS image [vertex][plane] (row pitch 4 x an odd number of floats: sixteen lanes reading
//     sixteen different rows cover every bank once);
//   * its operator row (<= 20 entries: neighbour vertex + value) sits in REGISTERS for the whole launch -- no operator stream
//     at all;
//   * T_k meets W_k on the matrix cores while it is still in registers: y^T[fout][vertex] += W_k^T[fout][fin] T_k[fin][vertex]
//     (v_mfma_f32_32x32x2_f32, exact fp32; the B operand of an instruction is one state plane of each half-wave, the A
//     operand a conflict-free 4-byte LDS read of W);
//   * the stack [K][B][Fin][Mp] is written only when the caller asks for it (training: contract_bwd_w reads it), never read;
//   * the gradient wrt the input needs no stack either: G_j = W_j dy comes off the matrix cores in the accumulator layout,
//     whose rows are made the state planes of the lane that receives them.
// One window per workgroup would fill 128 of the 256 CUs at the reference's batch of 128, and the kernel is bound by the LDS
// reads of its gather: with PL = 8 a window is split between TWO workgroups (16 of the 32 planes each, 8 per lane) -- a plane's
// recurrence needs no other plane; the forward then leaves two partial sums over its 16 input planes each, which
// fused_combine_kernel adds in a fixed order (+ bias, ReLU, mask); the backward computes G_j for its own planes only.
// Sums: a row of L T is an fmaf chain over the row's entries in the caller's (CSR) order; the contraction is an fmaf chain
// over (k, fin) -- fp32 round-off only, deterministic.
#include <string>

#include "contract_common.h"

namespace chebgcn {
#ifndef CG_X
#define CG_X 0               // 64: in-kernel phase stamps (tools/fbuild.sh, tools/fused_check.py --stamps); 0 in production
#endif
__device__ long long g_dbgf[16 * 64];
// workgroup 37: 0 start, 1 operator row in registers, 2 W in LDS, 3 input planes, 4 image of T_0, 5.. after each step, 40 results out
#define CG_FSTAMP(id)                                                                         \
    do {                                                                                      \
        if ((CG_X & 64) && (id) < 64 && lane == 0 && blockIdx.x == 37)                        \
            g_dbgf[wave * 64 + (id)] = (long long)__builtin_readcyclecounter();               \
    } while (0)
namespace {

constexpr int FS_MAXLEN = 20;        // operator entries per row held in registers (template parameter ML: 16 where the graph's rows allow --
                                     // six registers that the twelve-wave kernels otherwise spill)
// floats per vertex row of the LDS image: the planes of the workgroup + 4 (16-byte aligned, 4 x an odd number)
__host__ __device__ constexpr int fs_row(int PL) { return 2 * PL + 4; }

struct FusedArgs {
    const uint32_t* rec;         // per-vertex operator records [Mp][32] (common.h Ell::fs_rec; forward: L~, backward: L~^T)
    const float* in;             // forward: x [B][Fin][Mp];  backward: dout [B][Fout][Mp]
    const float* W;              // [Fin*K][Fout]
    const float* bias;           // forward only
    float* stack;                // forward: [K][B][Fin][Mp] or NULL (slab 0 may be `in` itself)
    float* out;                  // forward: y [B][Fout][Mp] (PL = 8: the partial sums [2][B][32][Mp]);  backward: dx [B][Fin][Mp]
    uint8_t* mask;               // ReLU bit mask [B][Fout][Mp/4]: written by the forward, read by the backward (NULL: none)
    int B, M, Mp, Fin, K, Fout, relu, bias_kind;
    size_t slab;                 // B*Fin*Mp
};

// workgroup barrier for LDS traffic only: __syncthreads() also waits for every global store in flight (s_waitcnt vmcnt(0)) --
// the stack rows of a step would be acknowledged by memory before the next gather could start
__device__ __forceinline__ void fs_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory"); }
__device__ __forceinline__ f32x16 mfma2(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int fs_opq(int x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ int fs_ops(int x) { asm volatile("" : "+s"(x)); return x; }

// row (h = 0) of accumulator register r of the 32x32 layout; acc_row(r, h) = pu(r) + 4 h
__host__ __device__ constexpr int pu(int r) { return (r & 3) + 8 * (r >> 2); }
// State plane i (< PL) of a lane: sp4 + ps<PL>(i) + pls<PL> * h, sp = which half of the window the workgroup carries.
//   PL = 16: the accumulator row pattern itself.
//   PL = 8: workgroup sp owns the planes acc_row(rho, sp), rho = 0..15, and the lane the rho = acc_row(i, h) mod 16 of them.
template <int PL> __host__ __device__ constexpr int ps(int i) { return PL == 16 ? (i & 3) + 8 * (i >> 2) : (i & 3) + 16 * (i >> 2); }
template <int PL> constexpr int pls = PL == 16 ? 4 : 8;

// NW waves = 32*NW vertices.  PL = planes per lane: 16 (one workgroup per window) or 8 (two).  ADJ = false: the layer forward;
// true: its gradient wrt the input.
template <int NW, int PL, bool ADJ, int ML>
__global__ void __launch_bounds__(NW * 64)
fused_layer_kernel(FusedArgs a) {
    constexpr int FS_ROW = fs_row(PL);
    constexpr int NS = 16 / PL;                       // workgroups per window
    constexpr int NI = ADJ ? 16 : PL;                 // planes that come in per lane (backward: all of dy, accumulator pattern)
    extern __shared__ __attribute__((aligned(16))) float fs_smem[];
    float* T = fs_smem;                               // [32*NW][FS_ROW]: row v = the state planes of vertex v, half h at floats PL h ..
    float* Ws = fs_smem + 32 * NW * FS_ROW;           // forward [K][in-plane index 2 PL][fout 32]; backward [K][fout 32][row 32]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int v = wave * 32 + c;
    const bool vok = v < a.M;
    const int Mq = a.Mp >> 2;
    CG_FSTAMP(0);
    const int sp = NS > 1 ? (int)(blockIdx.x % NS) : 0;                      // uniform
    const int sp4 = 4 * sp;
    // A plane access is a UNIFORM base (window, slab, plane index without h: scalar registers) plus ONE per-lane offset (the h
    // part of the plane + the vertex): sixteen 64-bit per-lane addresses per tensor would not fit beside the state.
    const unsigned loff = (unsigned)(4 * h * a.Mp + v);                       // elements: accumulator-pattern planes pu(r) + 4 h
    const unsigned lsoff = (unsigned)((pls<PL> * h + sp4) * a.Mp + v);        // elements: state planes ps(i) + pls h + sp4
    const unsigned lmq = (unsigned)(4 * h * Mq + (v >> 2));                   // bytes: mask rows pu(r) + 4 h
    // Both plane patterns grow with their index: the planes of a lane that exist are a PREFIX (empty beyond the graph).  The
    // counts go through an opaque identity where a phase starts -- sixteen loop-invariant lane masks per tensor, hoisted out of
    // the window loop, were 200 spilled scalar registers.
    int nacc_fout = 0, nst_fin = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) nacc_fout += (pu(i) + 4 * h < a.Fout) ? 1 : 0;
#pragma unroll
    for (int i = 0; i < PL; ++i) nst_fin += (ps<PL>(i) + pls<PL> * h + sp4 < a.Fin) ? 1 : 0;
    const int nv_fout = vok ? nacc_fout : 0, nv_fin = vok ? nst_fin : 0;

    // ---- prologue: this vertex's operator row -> registers, W -> LDS ------------------------------------------------------------
    // Memory round trips, not bytes, are what the prologue costs (phase stamps: 14-20k of a launch's 80-98k cycles with one load
    // per round trip): the row pointers and the first batch of W are requested together, the row's entries as soon as the
    // pointers are back, and W moves in batches of WU loads per thread.
    // forward:  Ws[k][q][fout] = W[plane(q) * K + k][fout], q = i + PL h' the in-plane index of the B operand's lane half;
    // backward: Ws[j][fo][row] = W[fin(row) * K + j][fo], row = accumulator row, fin(row) = the state plane of the lane (row's
    //           half) and register (row's pattern index mod PL) that receives it.  Global reads run along fo in both (one 128-byte
    //           line per 32 lanes); the backward image is written transposed.
    constexpr int WU = 8;
    const int nW = a.K * 1024;
    auto w_src = [&](int idx, int& dst) -> const float* {
        const int k = idx >> 10, r = (idx >> 5) & 31, q = idx & 31;          // q = fo: the fastest index of the global read
        int fin;
        if (!ADJ) {
            fin = r < 2 * PL ? ps<PL>(r % PL) + pls<PL> * (r / PL) + sp4 : 1 << 20;
            dst = idx;
        } else {
            const int rh = (r >> 2) & 1, ri = (r & 3) + 4 * (r >> 3);            // row r = pu(ri) + 4 rh
            fin = ps<PL>(ri % PL) + pls<PL> * rh + sp4;
            dst = (k << 10) + (q << 5) + r;
        }
        return (fin < a.Fin && q < a.Fout) ? a.W + (size_t)(fin * a.K + k) * a.Fout + q : nullptr;
    };
    auto w_batch_load = [&](int base, float (&wv)[WU]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < WU; ++u) {
            int dst;
            const int idx = base + u * NW * 64;
            const float* p = idx < nW ? w_src(idx, dst) : nullptr;
            const float t = *(p ? p : a.W);                                // (unconditional: a branch per load is a round trip per load)
            wv[u] = p ? t : 0.f;
        }
    };
    auto w_batch_store = [&](int base, const float (&wv)[WU]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < WU; ++u) {
            int dst = 0;
            const int idx = base + u * NW * 64;
            if (idx < nW) { w_src(idx, dst); Ws[dst] = wv[u]; }
        }
    };
    // the vertex's record (one 128-byte line: neighbour pairs, length, values) and two batches of W: one round trip for all
    float wv0[WU], wv1[WU];
    w_batch_load(threadIdx.x, wv0);
    w_batch_load(threadIdx.x + WU * NW * 64, wv1);
    const uint32_t* rec = a.rec + (size_t)(vok ? v : 0) * 32;
    unsigned ec[ML / 2];
    float ev[ML];
#pragma unroll
    for (int e = 0; e < ML / 2; ++e) ec[e] = rec[e];
#pragma unroll
    for (int e = 0; e < ML; ++e) ev[e] = __builtin_bit_cast(float, rec[12 + e]);
    const int len = vok ? (int)rec[10] : 0;
    if (!vok) {
#pragma unroll
        for (int e = 0; e < ML / 2; ++e) ec[e] = 0u;
#pragma unroll
        for (int e = 0; e < ML; ++e) ev[e] = 0.f;
    }
    w_batch_store(threadIdx.x, wv0);
    w_batch_store(threadIdx.x + WU * NW * 64, wv1);
    for (int base = threadIdx.x + 2 * WU * NW * 64; base < nW; base += WU * NW * 64) {
        float wv[WU];
        w_batch_load(base, wv);
        w_batch_store(base, wv);
    }
    int lmax = len;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) lmax = max(lmax, __shfl_xor(lmax, d));
    const int lenw = __builtin_amdgcn_readfirstlane(lmax);          // the wave's longest row (uniform)
    CG_FSTAMP(1);
    __syncthreads();
    CG_FSTAMP(2);

    const unsigned hoff = (unsigned)h * (PL * 4u);
    const unsigned own = (unsigned)v * (FS_ROW * 4u) + hoff;                   // this lane's PL floats of the image
    auto put_image = [&](const float (&t)[PL]) __attribute__((always_inline)) {
        float4* p = reinterpret_cast<float4*>(reinterpret_cast<char*>(T) + own);
#pragma unroll
        for (int q = 0; q < PL / 4; ++q) p[q] = make_float4(t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]);
    };
    // (L t)[v] for the planes of this lane: an fmaf chain over the row's entries.  Entries go in chunks of EC: ONE uniform test
    // per chunk, the LDS reads of a chunk issued together (a test per entry made every entry its own basic block: two reads,
    // a full wait, eight multiply-adds, 30 cycles per read instead of 4); entries beyond a row's end read vertex 0 against a
    // zero value.
    constexpr int EC = PL == 16 ? 2 : 4;
    auto gather = [&](float (&g)[PL]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PL; ++i) g[i] = 0.f;
#pragma unroll
        for (int e0 = 0; e0 < ML; e0 += EC) {
            if (e0 < lenw) {                                             // uniform: no divergence, no LDS reads beyond the wave's rows
                float4 t[EC][PL / 4];
#pragma unroll
                for (int u = 0; u < EC; ++u) {
                    const int e = e0 + u;
                    const unsigned ci = (e & 1) ? (ec[e >> 1] >> 16) : (ec[e >> 1] & 0xFFFFu);
                    const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(T) + (ci * (FS_ROW * 4u) + hoff));
#pragma unroll
                    for (int q = 0; q < PL / 4; ++q) t[u][q] = p[q];
                }
#pragma unroll
                for (int u = 0; u < EC; ++u) {
                    const float w = ev[e0 + u];
#pragma unroll
                    for (int q = 0; q < PL / 4; ++q) {
                        g[4 * q] = fmaf(w, t[u][q].x, g[4 * q]);
                        g[4 * q + 1] = fmaf(w, t[u][q].y, g[4 * q + 1]);
                        g[4 * q + 2] = fmaf(w, t[u][q].z, g[4 * q + 2]);
                        g[4 * q + 3] = fmaf(w, t[u][q].w, g[4 * q + 3]);
                    }
                }
            }
        }
    };
    // forward: acc[fout][vertex] += sum over the state planes of Ws[k] x t;  instruction i pairs the planes i of the two halves
    auto product_fwd = [&](int kidx, const float (&t)[PL], f32x16 acc) __attribute__((always_inline)) -> f32x16 {
        const float* w = Ws + kidx * 1024 + c + (PL * 32) * h;
#pragma unroll
        for (int i = 0; i < PL; ++i) acc = mfma2(w[i * 32], t[i], acc);
        return acc;
    };
    // backward: G[row][vertex] = sum_fo Ws[j][fo][row] dy[fo][vertex];  instruction i pairs the dy planes pu(i), pu(i) + 4
    auto product_bwd = [&](int j, const float (&dy)[NI]) __attribute__((always_inline)) -> f32x16 {
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
        const float* w = Ws + j * 1024 + c + 128 * h;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc = mfma2(w[pu(i) * 32], dy[i < NI ? i : 0], acc);
        return acc;
    };

    for (int b = (int)(blockIdx.x / NS); b < a.B; b += (int)(gridDim.x / NS)) {
        // ---- the window's input planes (backward: gated by the ReLU mask) ---------------------------------------------------
        float in[NI];
        {
            const int mp = fs_ops(a.Mp);
            if (!ADJ) {
                const float* src = a.in + (size_t)b * a.Fin * a.Mp;          // uniform
                const int n = fs_opq(nv_fin);
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    in[i] = 0.f;
                    if (i < n) in[i] = (src + (size_t)ps<PL>(i) * mp)[lsoff];
                }
            } else {
                const float* src = a.in + (size_t)b * a.Fout * a.Mp;
                const int n = fs_opq(nv_fout);
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    in[i] = 0.f;
                    if (i < n) in[i] = (src + (size_t)pu(i) * mp)[loff];
                }
                if (a.mask) {
                    const uint8_t* mk = a.mask + (size_t)b * a.Fout * Mq;
                    const int mq = fs_ops(Mq);
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        int bits = 0;
                        if (i < n) bits = (int)(mk + (size_t)pu(i) * mq)[lmq];
                        in[i] = ((bits >> (v & 3)) & 1) ? in[i] : 0.f;
                    }
                }
            }
        }
        CG_FSTAMP(3);
        float cur[PL], prev[PL];
        f32x16 yacc;
#pragma unroll
        for (int q = 0; q < 16; ++q) yacc[q] = 0.f;
        if (!ADJ) {
#pragma unroll
            for (int i = 0; i < PL; ++i) { cur[i] = in[i < NI ? i : 0]; prev[i] = 0.f; }
            if (a.stack && a.stack != a.in) {                            // T_0 = x into slab 0
                float* s0 = a.stack + (size_t)b * a.Fin * a.Mp;
                const int n = fs_opq(nv_fin), mp = fs_ops(a.Mp);
#pragma unroll
                for (int i = 0; i < PL; ++i)
                    if (i < n) (s0 + (size_t)ps<PL>(i) * mp)[lsoff] = cur[i];
            }
        } else {
            const f32x16 g0 = product_bwd(a.K - 1, in);                  // c_{K-1} = G_{K-1} = W_{K-1} dy
#pragma unroll
            for (int i = 0; i < PL; ++i) { cur[i] = g0[i]; prev[i] = 0.f; }
        }
        put_image(cur);
        fs_barrier();                                                     // the image of T_0 / c_{K-1}
        CG_FSTAMP(4);
        for (int step = 1; step < a.K; ++step) {
            const bool last = step == a.K - 1;
            const float f = ADJ ? (last ? 1.f : 2.f) : (step == 1 ? 1.f : 2.f);
            // the matrix instructions of a step are issued IN FRONT of the gather they do not depend on (forward: T_{k-1} x
            // W_{k-1}; backward: G_j = W_j dy, taken into `prev`): the matrix pipe works while the wave gathers
            if (!ADJ) {
                yacc = product_fwd(step - 1, cur, yacc);
            } else {
                const f32x16 gj = product_bwd(a.K - 1 - step, in);       // G_j = W_j dy
#pragma unroll
                for (int i = 0; i < PL; ++i) prev[i] = gj[i] - prev[i];
            }
            float g[PL];
            gather(g);
            CG_FSTAMP(20 + step);
            float nw[PL];
#pragma unroll
            for (int i = 0; i < PL; ++i) nw[i] = ADJ ? fmaf(f, g[i], prev[i]) : fmaf(f, g[i], -prev[i]);
            fs_barrier();                                                 // every gather of this step has read the image
            if (!last) put_image(nw);
#pragma unroll
            for (int i = 0; i < PL; ++i) { prev[i] = cur[i]; cur[i] = nw[i]; }
            if (!ADJ) {
                if (a.stack) {
                    float* sk = a.stack + (size_t)step * a.slab + (size_t)b * a.Fin * a.Mp;
                    const int n = fs_opq(nv_fin), mp = fs_ops(a.Mp);
#pragma unroll
                    for (int i = 0; i < PL; ++i)
                        if (i < n) (sk + (size_t)ps<PL>(i) * mp)[lsoff] = cur[i];
                }
            }
            if (!last) fs_barrier();                                      // the image of T_k / c_j
            CG_FSTAMP(4 + step);
        }
        if (!ADJ) yacc = product_fwd(a.K - 1, cur, yacc);
        // ---- results --------------------------------------------------------------------------------------------------------
        if (!ADJ && NS > 1) {
            // half of the sum over the input planes: raw, for fused_combine_kernel ([half][B][32][Mp]; rows beyond Fout are zero)
            float* ob = a.out + ((size_t)sp * a.B + b) * 32 * a.Mp;
            const int mp = fs_ops(a.Mp);
            if (fs_opq((int)vok)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) (ob + (size_t)pu(r) * mp)[loff] = yacc[r];
            }
        } else if (!ADJ) {
            float* ob = a.out + (size_t)b * a.Fout * a.Mp;
            uint8_t* mb = a.mask ? a.mask + (size_t)b * a.Fout * Mq + wave * 8 : nullptr;
            const int n = fs_opq(nv_fout), nf = fs_opq(nacc_fout), mp = fs_ops(a.Mp);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float y = yacc[r];
                if (r < n) {
                    if (a.bias_kind == CHEBGCN_BIAS_FILTER) y += a.bias[pu(r) + 4 * h];
                    else if (a.bias_kind == CHEBGCN_BIAS_VERTEX) y += (a.bias + (size_t)pu(r) * mp)[loff];
                }
                if (a.relu) y = fmaxf(y, 0.f);
                if (r < n) (ob + (size_t)pu(r) * mp)[loff] = y;
                if (mb) {
                    // bit (v & 3) of byte v / 4 of row fo: the 32 vertices of a half-wave make eight bytes
                    const unsigned long long bal = __ballot(y > 0.f && r < n);
                    const unsigned bits = h ? (unsigned)(bal >> 32) : (unsigned)bal;
                    if (c < 8 && r < nf && wave * 32 + 4 * c < a.Mp)
                        (mb + (size_t)pu(r) * Mq)[4 * h * Mq + c] = (uint8_t)((bits >> (4 * c)) & 15u);
                }
            }
        } else {
            float* dx = a.out + (size_t)b * a.Fin * a.Mp;
            const int n = fs_opq(nv_fin), mp = fs_ops(a.Mp);
#pragma unroll
            for (int i = 0; i < PL; ++i)
                if (i < n) (dx + (size_t)ps<PL>(i) * mp)[lsoff] = cur[i];
        }
        CG_FSTAMP(40);
        // (the next window overwrites the image: every gather of the last step is behind the barrier of that step; with
        // K = 1 nothing ever read the image)
    }
}

// y = act(P_0 + P_1 + bias) (+ ReLU mask) from the two partial sums [2][B][32][Mp] of the split forward: four vertices per
// thread, the two addends in a fixed order.
__global__ void __launch_bounds__(256)
fused_combine_kernel(const float* __restrict__ part, const float* __restrict__ bias, int bias_kind, float* __restrict__ out,
                     uint8_t* __restrict__ mask, int B, int M, int Mp, int Fout, int relu) {
    const int Mq = Mp >> 2;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;               // quad of vertices
    const int fo = blockIdx.y, b = blockIdx.z;
    if (q >= Mq) return;
    const size_t half = (size_t)B * 32 * Mp;
    const float4 p0 = *reinterpret_cast<const float4*>(part + ((size_t)b * 32 + fo) * Mp + 4 * q);
    const float4 p1 = *reinterpret_cast<const float4*>(part + half + ((size_t)b * 32 + fo) * Mp + 4 * q);
    float y[4] = {p0.x + p1.x, p0.y + p1.y, p0.z + p1.z, p0.w + p1.w};
    if (bias_kind == CHEBGCN_BIAS_FILTER) {
        const float bb = bias[fo];
#pragma unroll
        for (int r = 0; r < 4; ++r) y[r] += bb;
    } else if (bias_kind == CHEBGCN_BIAS_VERTEX) {
        const float4 bb = *reinterpret_cast<const float4*>(bias + (size_t)fo * Mp + 4 * q);
        y[0] += bb.x; y[1] += bb.y; y[2] += bb.z; y[3] += bb.w;
    }
    int bits = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (relu) y[r] = fmaxf(y[r], 0.f);
        if (4 * q + r >= M) y[r] = 0.f;                                 // the pad is scratch: keep it finite
        bits |= (y[r] > 0.f) ? (1 << r) : 0;
    }
    *reinterpret_cast<float4*>(out + ((size_t)b * Fout + fo) * Mp + 4 * q) = make_float4(y[0], y[1], y[2], y[3]);
    if (mask) mask[((size_t)b * Fout + fo) * Mq + q] = (uint8_t)bits;
}

// (16 waves -- up to 512 vertices -- would have 128 registers per lane: the state spills)
int fs_waves(int Mp) { return Mp <= 256 ? 8 : Mp <= 384 ? 12 : 0; }
// two workgroups per window while one per window leaves CUs idle
int fs_split(int B, int cus) { return 2 * B <= cus + cus / 2 ? 2 : 1; }
size_t fs_lds(int nw, int PL, int K) { return ((size_t)32 * nw * fs_row(PL) + (size_t)K * 1024) * sizeof(float); }

template <int NW, int PL, bool ADJ, int ML>
int fs_launch(const FusedArgs& a, int cus, hipStream_t stream) {
    const size_t lds = fs_lds(NW, PL, a.K);
    CG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fused_layer_kernel<NW, PL, ADJ, ML>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds));
    static const std::string name = "fused_layer_kernel<" + std::to_string(NW) + "," + std::to_string(PL) + "," + (ADJ ? "true" : "false") + ">";
    note_dispatch(name.c_str());
    constexpr int NS = 16 / PL;
    const int per_cu = (int)((160 * 1024) / lds) < 1 ? 1 : (int)((160 * 1024) / lds);
    int slots = cus * (per_cu > 2 ? 2 : per_cu);                         // (NW = 12: two workgroups are 24 of a CU's 32 waves)
    slots -= slots % NS;
    const int want = a.B * NS;
    const int grid = want < slots ? want : slots;
    hipLaunchKernelGGL((fused_layer_kernel<NW, PL, ADJ, ML>), dim3(grid), dim3(NW * 64), lds, stream, a);
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

template <bool ADJ, int ML>
int fs_dispatch_ml(const FusedArgs& a, int cus, int split, hipStream_t stream) {
    const int nw = fs_waves(a.Mp);
    if (nw == 8) return split == 2 ? fs_launch<8, 8, ADJ, ML>(a, cus, stream) : fs_launch<8, 16, ADJ, ML>(a, cus, stream);
    if (nw == 12) return split == 2 ? fs_launch<12, 8, ADJ, ML>(a, cus, stream) : fs_launch<12, 16, ADJ, ML>(a, cus, stream);
    return fail(CHEBGCN_EUNSUPPORTED, "fused layer: %d vertices", a.M);
}
template <bool ADJ>
int fs_dispatch(const FusedArgs& a, int cus, int split, int max_len, hipStream_t stream) {
    return max_len <= 16 ? fs_dispatch_ml<ADJ, 16>(a, cus, split, stream) : fs_dispatch_ml<ADJ, FS_MAXLEN>(a, cus, split, stream);
}

}  // namespace
}  // namespace chebgcn

using namespace chebgcn;

extern "C" int chebgcn_fused_layer_supported(const chebgcn_graph* g, int B, int Fin, int K, int Fout) {
    if (!g || B <= 0 || Fin <= 0 || K <= 0 || Fout <= 0 || Fin > 32 || Fout > 32) return 0;
    const int nw = fs_waves(g->Mp);
    if (nw == 0 || !g->lds_ok) return 0;
    if (g->fwd.max_len > FS_MAXLEN || g->adj.max_len > FS_MAXLEN || !g->fwd.fs_rec || !g->adj.fs_rec) return 0;
    if (fs_lds(nw, 16, K) > 160 * 1024) return 0;
    return 1;
}

extern "C" size_t chebgcn_fused_layer_workspace(const chebgcn_graph* g, int B, int Fin, int K, int Fout) {
    if (!chebgcn_fused_layer_supported(g, B, Fin, K, Fout)) return 0;
    return fs_split(B, g->num_cus) == 2 ? (size_t)2 * B * 32 * g->Mp * sizeof(float) : 0;
}

extern "C" int chebgcn_fused_layer_fwd(const chebgcn_graph* g, const float* x, const float* W, const float* bias, int bias_kind,
                                       float* stack, float* out, uint8_t* relu_mask, void* workspace, size_t workspace_bytes, int B,
                                       int Fin, int K, int Fout, int relu, chebgcn_stream stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CG_REQUIRE(g && x && W && out, "fused_layer_fwd: NULL argument");
    CG_REQUIRE(bias_kind >= 0 && bias_kind <= 2 && (bias_kind == CHEBGCN_BIAS_NONE || bias), "fused_layer_fwd: bad bias");
    if (!chebgcn_fused_layer_supported(g, B, Fin, K, Fout))
        return fail(CHEBGCN_EUNSUPPORTED, "fused_layer_fwd: shape not served (chebgcn_fused_layer_supported)");
    const int split = fs_split(B, g->num_cus);
    CG_REQUIRE(split == 1 || (workspace && workspace_bytes >= chebgcn_fused_layer_workspace(g, B, Fin, K, Fout)),
               "fused_layer_fwd: workspace of chebgcn_fused_layer_workspace() bytes needed");
    FusedArgs a;
    a.rec = g->fwd.fs_rec;
    a.in = x; a.W = W; a.bias = bias; a.stack = stack; a.out = split == 2 ? (float*)workspace : out;
    a.mask = relu ? relu_mask : nullptr;
    a.B = B; a.M = g->M; a.Mp = g->Mp; a.Fin = Fin; a.K = K; a.Fout = Fout; a.relu = relu; a.bias_kind = bias_kind;
    a.slab = (size_t)B * Fin * g->Mp;
    const int rc = fs_dispatch<false>(a, g->num_cus, split, g->fwd.max_len, stream);
    if (rc != CHEBGCN_OK || split == 1) return rc;
    note_dispatch_more("fused_combine_kernel");
    hipLaunchKernelGGL(fused_combine_kernel, dim3((g->Mp / 4 + 255) / 256, Fout, B), dim3(256), 0, stream, (const float*)workspace, bias,
                       bias_kind, out, relu ? relu_mask : nullptr, B, g->M, g->Mp, Fout, relu);
    CG_HIP(hipGetLastError());
    return CHEBGCN_OK;
}

extern "C" int chebgcn_fused_layer_bwd_x(const chebgcn_graph* g, const float* dout, const uint8_t* relu_mask, const float* W,
                                         float* dx, int B, int Fin, int K, int Fout, chebgcn_stream stream_) {
    CG_REQUIRE(g && dout && W && dx, "fused_layer_bwd_x: NULL argument");
    if (!chebgcn_fused_layer_supported(g, B, Fin, K, Fout))
        return fail(CHEBGCN_EUNSUPPORTED, "fused_layer_bwd_x: shape not served (chebgcn_fused_layer_supported)");
    FusedArgs a;
    a.rec = g->adj.fs_rec;
    a.in = dout; a.W = W; a.bias = nullptr; a.stack = nullptr; a.out = dx; a.mask = const_cast<uint8_t*>(relu_mask);
    a.B = B; a.M = g->M; a.Mp = g->Mp; a.Fin = Fin; a.K = K; a.Fout = Fout; a.relu = 0; a.bias_kind = CHEBGCN_BIAS_NONE;
    a.slab = (size_t)B * Fin * g->Mp;
    return fs_dispatch<true>(a, g->num_cus, fs_split(B, g->num_cus), g->adj.max_len, (hipStream_t)stream_);
}

#ifdef CG_EXPERIMENT
extern "C" int chebgcn_debug_stampsf(long long* out) {      // CG_X & 64 builds only (tools/fused_check.py --stamps)
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(chebgcn::g_dbgf), sizeof(long long) * 16 * 64) == hipSuccess ? 0 : -1;
}
#endif
