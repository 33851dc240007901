/*
 * chebgcn.h -- C ABI of libchebgcn.so: the MI355X (gfx950) implementation of the
 * Chebyshev graph-convolution hot path of zhangyu2ustc/GCN_fmri_decoding.
 *
 * The reference has no FFI of its own: the path is a Python class
 * (lib_new/models_gcn.py, class cgcnn) whose methods emit TensorFlow ops.  Each
 * entry point below replaces the TF ops emitted by the cited reference lines;
 * gcn_fmri_decoding_amd/ (Python, ctypes) binds them behind the reference's
 * cgcnn / chebyshev5 / b1relu / b2relu / mpool1 API.  INTEGRATION.md shows the
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *  - plain C types only; every function returns 0 on success or a negative
 *    CHEBGCN_E* code, never throws; chebgcn_last_error() gives the message of
 *    the last failure on the calling thread.
 *  - all activation pointers are DEVICE pointers owned by the caller (PyTorch's
 *    caching allocator in our host code); kernels are enqueued on `stream`
 *    (a hipStream_t; NULL = the default stream) and the call returns without
 *    synchronising.  Pointers marked "host" are host memory, read before return.
 *  - activation layout is "plane": a logical [B, M, F] tensor of the reference
 *    (B windows, M graph vertices, F features) is stored as [B][F][Mp] floats
 *    with the vertex axis fastest and Mp = chebgcn_plane_stride(M) >= M.  The pad
 *    [M, Mp) of every plane is scratch: kernels may write it and never read it
 *    as data.
 *  - a graph handle is immutable after creation and may be used concurrently.
 */
#ifndef CHEBGCN_H
#define CHEBGCN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CHEBGCN_VERSION 1

enum {
    CHEBGCN_OK = 0,
    CHEBGCN_EINVAL = -1,       /* bad argument / shape */
    CHEBGCN_EHIP = -2,         /* a HIP runtime call failed */
    CHEBGCN_ENOMEM = -3,
    CHEBGCN_EUNSUPPORTED = -4  /* valid but not implemented for this size */
};

enum { CHEBGCN_BIAS_NONE = 0, CHEBGCN_BIAS_FILTER = 1 /* b1relu: [F] */, CHEBGCN_BIAS_VERTEX = 2 /* b2relu: [F][Mp] */ };
enum { CHEBGCN_POOL_MAX = 0, CHEBGCN_POOL_AVG = 1 };

typedef struct chebgcn_graph chebgcn_graph;
typedef void* chebgcn_stream;

int chebgcn_version(void);
const char* chebgcn_last_error(void);
/* Names of the kernel templates the calling thread's last launching entry point enqueued, in launch order, joined by
 * " + " (e.g. "contract_bwd_w_kernel<5,true> + reduce_partials_stage1 + reduce_partials_stage2"); "" before the first
 * launch.  The dispatchers below choose instantiations by shape and by the device's CU count: this is how a test (or a
 * profile) names the one a given call reached.  The pointer stays valid until the thread's next call of this function. */
const char* chebgcn_last_dispatch(void);

/* Padded plane length for M vertices (multiple of 32 floats = 128 B). */
int chebgcn_plane_stride(int M);

/* ---- graph: the constant operand of tf.sparse_tensor_dense_matmul ---------------
 * Replaces models_gcn.py:593-596 (tf.SparseTensor + tf.sparse_reorder of the
 * rescaled Laplacian).  Takes L~ = rescale_L(L, lmax=2) (lib_new/graph.py:146-152)
 * as host CSR (row-major, any column order inside a row), builds device-side
 * length-sorted sliced-ELL images of L~ and of L~^T (the adjoint used by the gradient of
 * SparseTensorDenseMatMul).  The order in which the entries of a row are summed is chosen
 * by the library (LDS bank placement), not the caller's: a row sum is an fp32 fmaf chain
 * over its <= ~17 entries in that fixed order, deterministic from run to run.  The host
 * arrays are copied; the caller keeps ownership. */
int chebgcn_graph_create(int M, int64_t nnz, const int32_t* rowptr /*host [M+1]*/,
                         const int32_t* colidx /*host [nnz]*/, const float* vals /*host [nnz]*/,
                         chebgcn_graph** out);
/* The same, with the number of planes (window x feature columns) a recurrence workgroup carries
 * on chip chosen by the caller: 0 = automatic (what chebgcn_graph_create does: 4 wherever 16 bytes
 * per active vertex fit the 160 KB of LDS, i.e. up to ~10200 active vertices, else 2), 2, or 4
 * (CHEBGCN_EUNSUPPORTED where 4 do not fit).  The
 * choice affects speed and the order of the terms inside a row sum (results agree to fp32
 * round-off).  With 0, a graph that fits four planes keeps a two-plane image as well and a launch
 * of fewer than four plane groups per CU (ceil(B*Fin/4) < 4 * CUs) runs on it: the same window can
 * then differ in its last fp32 bits between batch sizes (training batch, evaluation batch, a
 * data-parallel shard) and between GPUs with different CU counts; planes = 2 or 4 pins one image. */
int chebgcn_graph_create_planes(int M, int64_t nnz, const int32_t* rowptr, const int32_t* colidx,
                                const float* vals, int planes, chebgcn_graph** out);
void chebgcn_graph_destroy(chebgcn_graph* g);
/* what: 0 = M, 1 = nnz, 2 = plane stride Mp, 3 = 1 if the on-chip (LDS) recurrence
 * kernel is used for this graph, 4 = padded ELL slots of L~, 5 = max row length,
 * 6 = planes per workgroup (0, 2, 4), 7 = rows held in the LDS image, 8 = LDS bytes of the
 * image, 9 / 10 / 11 = modelled LDS cycles of one gather pass in the caller's entry order /
 * after the library's bank-aware placement / without any conflict, 12 = 1 if the handle carries
 * the ORDERED operator image: the rows of the caller's matrix are sorted by descending length
 * (isolated vertices last) and the graph was created with planes = 0 -- recurrence launches then
 * run the kernel that moves planes between HBM and registers directly (csrc/recurrence_ord_kernel.h;
 * served: planes of more than 1024 vertices, up to 20476 active ones -- 256 threads per workgroup up to 2048 active vertices
 * (forward recurrences only: the Clenshaw adjoint of such a graph runs the kernel of the caller's order), 512 beyond --, any
 * number of isolated / padding vertices behind them);
 * 13 / 14 / 15 = items 9 / 10 / 11 for that image, 16 = its planes per workgroup (4 up to 10238
 * active vertices, 2 beyond; 0 = no ordered image). */
int chebgcn_graph_query(const chebgcn_graph* g, int what, int64_t* value);

/* ---- Chebyshev recurrence, forward: models_gcn.py:598-610 -----------------------
 * T_0 = x, T_1 = L~ T_0, T_k = 2 L~ T_{k-1} - T_{k-2}.
 * x: [B][Fin][Mp]; stack: [K][B][Fin][Mp].  Slab 0 receives a copy of x unless
 * x == stack (the producer already wrote T_0 in place). */
int chebgcn_recurrence_fwd(const chebgcn_graph* g, const float* x, float* stack,
                           int B, int Fin, int K, chebgcn_stream stream);

/* ---- the same recurrence on the TRANSPOSED operator: T_k(L~^T) x ------------------------------
 * (TF autodiff of models_gcn.py:598-617, associated the other way round.)  The gradient of a layer wrt its input is
 *   dx[b][fin] = sum_{k,fo} W[fin*K+k][fo] * ( T_k(L~^T) dy[b][fo] )
 * -- the forward recurrence on the Fout planes of dy, then chebgcn_contract_fwd(_bf16) of that stack with the re-indexed
 * weights W'[fo*K+k][fin] = W[fin*K+k][fo].  For Fout <= Fin this moves no more bytes than chebgcn_contract_bwd_x +
 * chebgcn_recurrence_bwd (which compute the same sum in Clenshaw form) and runs on the two faster kernels.  Same layouts and
 * in-place rule as chebgcn_recurrence_fwd. */
int chebgcn_recurrence_fwd_t(const chebgcn_graph* g, const float* x, float* stack,
                             int B, int Fin, int K, chebgcn_stream stream);
/* the re-indexed weights of that contraction: Wt[(fo*K + k)*Fin + fin] = W[(fin*K + k)*Fout + fo]  ([Fout*K][Fin] from
 * [Fin*K][Fout], both row-major; one small launch -- the weights change every step) */
int chebgcn_reindex_weights(const float* W, float* Wt, int Fin, int K, int Fout, chebgcn_stream stream);
/* The same for n <= 16 layers in one launch (host arrays of device pointers and shapes): the weights are constant within a
 * training step, cgcnn re-indexes every layer that forms its input gradient this way once, in front of the backward pass. */
int chebgcn_reindex_weights_batch(int n, const float* const* W, float* const* Wt, const int* Fin, const int* K,
                                  const int* Fout, chebgcn_stream stream);
/* The contraction of that stack with the ReluGrad of the layer BELOW in its epilogue (TF autodiff chains the two:
 * models_gcn.py:616 MatMul gradient -> :625/:629 ReluGrad of the previous layer): out[b][fo][m] = gate bit ? sum : 0, `gate` a
 * ReLU mask as chebgcn_contract_fwd leaves it, [B][Fout][Mp/4].  With stack = T_k(L~^T) dy of layer l and W = W' of layer l,
 * `out` IS the gated gradient wrt the output of layer l-1 -- written straight into slab 0 of the stack its own
 * chebgcn_recurrence_fwd_t fills, so the separate ReluGrad pass of layer l-1 (chebgcn_brelu_pool_bwd writing dy) shrinks to the
 * bias reduction.  No bias, no ReLU, no pooling; Fout <= 32 on a big launch (chebgcn_contract_fwd_gated_supported),
 * CHEBGCN_EUNSUPPORTED otherwise.  Same products in the same order as chebgcn_contract_fwd: bit-identical to
 * chebgcn_contract_fwd followed by the gating pass. */
int chebgcn_contract_fwd_gated_supported(int B, int M, int Fin, int K, int Fout);
int chebgcn_contract_fwd_gated(const float* stack, const float* W, const uint8_t* gate, float* out, int B, int M,
                               int Fin, int K, int Fout, chebgcn_stream stream);

/* ---- Chebyshev recurrence, adjoint: gradient of the above wrt x -----------------
 * (TF autodiff of models_gcn.py:598-610, reached from :298-303.)
 * c_{K-1} = G_{K-1}; c_j = G_j + 2 L~^T c_{j+1} - c_{j+2}; dx = G_0 + L~^T c_1 - c_2.
 * gstack: [K][B][Fin][Mp] (read only); dx: [B][Fin][Mp]. */
int chebgcn_recurrence_bwd(const chebgcn_graph* g, const float* gstack, float* dx,
                           int B, int Fin, int K, chebgcn_stream stream);

/* ---- dense contraction + bias + ReLU + pooling, forward -------------------------
 * Replaces models_gcn.py:611-617 (transpose/reshape + tf.matmul with W[Fin*K, Fout],
 * row index fin*K + k), :619-629 (b1relu / b2relu) and :631-648 (mpool1 / apool1).
 * stack: [K][B][Fin][Mp(M)];  W: [Fin*K][Fout] row-major;  bias: [Fout] or
 * [Fout][Mp(M)] or NULL;  out: [B][Fout][Mp(M/pool)];  argmax (may be NULL, only
 * written for max pooling with pool > 1): [B][Fout][Mp(M/pool)] bytes, position of
 * the first maximum inside each window.  relu != 0 applies max(.,0) before pooling.
 * pool must be a power of two <= 128 and divide M.
 * pool == 1 with relu != 0: a non-NULL argmax receives the ReLU MASK, [B][Fout][Mp(M)/4] bytes,
 * bit r of byte i = (out[4i+r] > 0) -- all that the gradient entries *_relu below need of the
 * forward result. */
int chebgcn_contract_fwd(const float* stack, const float* W, const float* bias, int bias_kind,
                         float* out, uint8_t* argmax, int B, int M, int Fin, int K, int Fout,
                         int pool, int pool_kind, int relu, chebgcn_stream stream);

/* ---- the same contraction on the bf16 matrix cores (wide layers: block_dura = 60, Fout = 256,
 * where the fp32-input MFMA is compute bound).  Same operands, epilogue and results layout as
 * chebgcn_contract_fwd; fp32 in HBM, converted in registers, fp32 accumulate.
 * passes = 1: operands rounded to bf16;  passes = 3: operands split into two bf16 each and
 * hi*hi + hi*lo + lo*hi accumulated (fp32-grade results, three times the matrix work).
 * workspace: device scratch of at least chebgcn_contract_fwd_bf16_workspace(Fin, K, Fout) bytes
 * (the packed bf16 image of W, rebuilt on every call). */
size_t chebgcn_contract_fwd_bf16_workspace(int Fin, int K, int Fout);
int chebgcn_contract_fwd_bf16(const float* stack, const float* W, const float* bias, int bias_kind,
                              float* out, uint8_t* argmax, int B, int M, int Fin, int K, int Fout,
                              int pool, int pool_kind, int relu, int passes, void* workspace,
                              size_t workspace_bytes, chebgcn_stream stream);

/* ---- the two gradients of the contraction on the bf16 matrix cores (wide layers; replace the
 * MatMul gradient TensorFlow derives for models_gcn.py:616 when the layer computes in bf16).
 * Same operands, layouts and `passes` as above; fp32 in HBM, fp32 accumulate, deterministic.
 *   bwd_x: gstack[k][b][fin][m] = sum_o  W[fin*K+k][o] * dy[b][o][m]     (overwrites gstack)
 *   bwd_w: dW[fin*K+k][o]       = sum_{b,m} stack[k][b][fin][m] * dy[b][o][m]   (overwrites dW)
 * Workspaces: device scratch of at least the size the *_workspace function reports. */
size_t chebgcn_contract_bwd_x_bf16_workspace(int Fin, int K, int Fout);
int chebgcn_contract_bwd_x_bf16(const float* dy, const float* W, float* gstack, int B, int M, int Fin,
                                int K, int Fout, int passes, void* workspace, size_t workspace_bytes,
                                chebgcn_stream stream);
size_t chebgcn_contract_bwd_w_bf16_workspace(int B, int M, int Fin, int K, int Fout);
int chebgcn_contract_bwd_w_bf16(const float* stack, const float* dy, float* dW, void* workspace,
                                size_t workspace_bytes, int B, int M, int Fin, int K, int Fout,
                                int passes, chebgcn_stream stream);

/* ---- the same two gradients with dy handed over as bf16 (one-pass arithmetic, wide layers).
 * The one-pass kernels round dy to bf16 before it meets the matrix cores; when the ReluGrad pass that produces dy
 * (models_gcn.py:619-629 under TF autodiff) writes it as bf16 in the first place, the largest operand of both gradients is half
 * the bytes and the results are BIT-IDENTICAL to chebgcn_contract_bwd_*_bf16(passes = 1) on the fp32 dy.
 *   chebgcn_relu_grad_bf16: dy16[b][o][m] = bf16( relu_mask bit ? dout[b][o][m] : 0 ), planes [B][F][Mp(M)] of 2-byte
 *       elements, and the bias gradient (fp32, fixed-order sums) in the same pass; relu_mask as chebgcn_contract_fwd(_bf16)
 *       wrote it for a pool == 1 ReLU layer; workspace as for chebgcn_brelu_pool_bwd.
 *   chebgcn_bf16_dy16_supported: 1 where both gradients below take this operand (layers wide enough for the one-pass weight
 *       gradient: Fin*K > 160 and Fout > 64), else 0 -- callers then keep the fp32 dy.
 *   workspaces: chebgcn_contract_bwd_w_bf16_workspace / chebgcn_contract_bwd_x_bf16_workspace. */
int chebgcn_relu_grad_bf16(const float* dout, const uint8_t* relu_mask, uint16_t* dy16, float* dbias, int bias_kind,
                           int B, int M, int F, void* workspace, size_t workspace_bytes, chebgcn_stream stream);
int chebgcn_bf16_dy16_supported(int B, int M, int Fin, int K, int Fout);
int chebgcn_contract_bwd_w_bf16_dy16(const float* stack, const uint16_t* dy16, float* dW, void* workspace,
                                     size_t workspace_bytes, int B, int M, int Fin, int K, int Fout,
                                     chebgcn_stream stream);
int chebgcn_contract_bwd_x_bf16_dy16(const uint16_t* dy16, const float* W, float* gstack, int B, int M, int Fin,
                                     int K, int Fout, void* workspace, size_t workspace_bytes, chebgcn_stream stream);

/* ---- bias + ReLU + pooling on their own (b1relu / b2relu / mpool1 / apool1 called
 * separately, models_gcn.py:619-648); same conventions as the epilogue of contract_fwd.
 * x: [B][F][Mp(M)] -> out: [B][F][Mp(M/pool)]. */
int chebgcn_brelu_pool_fwd(const float* x, const float* bias, int bias_kind, float* out,
                           uint8_t* argmax, int B, int M, int F, int pool, int pool_kind,
                           int relu, chebgcn_stream stream);

/* ---- backward of pooling + ReLU + bias (MaxPoolGrad, ReluGrad, bias reductions) ---
 * dout, out: [B][F][Mp(M/pool)] (out = forward result); argmax as written by the
 * forward; dy: [B][F][Mp(M)] receives d(loss)/d(pre-bias activation); dbias: [F] or
 * [F][Mp(M)] (overwritten) or NULL.  pool == 1 with relu: a non-NULL argmax is the ReLU mask of
 * contract_fwd and replaces `out` (which may then be NULL).  dy == NULL: only dbias is computed; with pool == 1 and relu == 0
 * as well that is the plain sum of dout over the windows (per filter: and vertices) -- the bias gradient of a layer whose
 * gated dy the layer above stored (chebgcn_contract_fwd_gated).
 * workspace: device scratch of at least chebgcn_brelu_pool_bwd_workspace() bytes.  bias_kind CHEBGCN_BIAS_FILTER needs
 * it: the per-filter sum of b1relu (models_gcn.py:619-623) is a two-stage reduction in a fixed order (per-workgroup
 * partials, then one wave per filter), so every gradient this library returns is bit-reproducible from run to run.  A
 * pooled layer (pool > 1) WITH the workspace runs the 16-byte-store kernel of chebgcn_pool_scatter_bwd (its bias sums
 * are per-batch-part partials added in part order); without it (NULL allowed for the other bias kinds) a scalar kernel. */
size_t chebgcn_brelu_pool_bwd_workspace(int B, int M, int F, int pool, int bias_kind);
int chebgcn_brelu_pool_bwd(const float* dout, const float* out, const uint8_t* argmax,
                           float* dy, float* dbias, int bias_kind, int B, int M, int F,
                           int pool, int pool_kind, int relu, void* workspace, size_t workspace_bytes,
                           chebgcn_stream stream);

/* ---- pooling between two vertex orders (mpool1 / apool1, models_gcn.py:631-648, under a relabelling) ----
 * The reference pools `pool` CONSECUTIVE vertices of the tree order coarsening.compute_perm builds (coarsening.py:168-215).
 * A caller that keeps the vertices of a level in another order -- cgcnn relabels each level by descending row length so that
 * the ordered recurrence kernels serve it (HCP_task_fmri_gcn_test8.py:1632-1635: the six-level pooling network) -- pools
 * through index maps instead of through adjacency in memory:
 *   chebgcn_pool_gather_fwd: y [B][F][Mp(M)] (bias and ReLU already applied: chebgcn_contract_fwd with pool = 1) ->
 *       out[b][f][j] = max_i / mean_i  y[b][f][ pmap[j*pool + i] ],  j < M/pool, [B][F][Mp(M/pool)] (padding zeroed).
 *       pmap: int32 [M] on the device, positions in the source order, the members of a cluster listed in the reference's
 *       order (ties resolve alike); NULL = the identity (the tree order itself).  sel [B][F][Mp(M/pool)] (optional, for the
 *       gradient): max -- the winning member, 0xFF where `relu` says y was rectified and the maximum is not positive;
 *       average -- bit i set where member i was positive (pool <= 8).
 *   chebgcn_pool_scatter_bwd: dy[b][f][v] = gradient of source vertex v (MaxPoolGrad / AvgPoolGrad + ReluGrad) from
 *       dout [B][F][Mp(M/pool)] and sel; smap: int32 [M], smap[v] = j*pool + i (v is member i of pooled vertex j), NULL =
 *       the identity; dbias as chebgcn_brelu_pool_bwd (fixed-order sums: per-batch-part partials in `workspace`, at least
 *       chebgcn_pool_scatter_bwd_workspace() bytes, added in part order).  16-byte stores of dy; the pooled plane is staged
 *       in LDS, so planes of more than 10240 pooled vertices are refused (CHEBGCN_EINVAL). */
int chebgcn_pool_gather_fwd(const float* y, const int32_t* pmap, float* out, uint8_t* sel, int B, int M, int F,
                            int pool, int pool_kind, int relu, chebgcn_stream stream);
size_t chebgcn_pool_scatter_bwd_workspace(int B, int M, int F, int pool, int bias_kind);
int chebgcn_pool_scatter_bwd(const float* dout, const uint8_t* sel, const int32_t* smap, float* dy, float* dbias,
                             int bias_kind, int B, int M, int F, int pool, int pool_kind, int relu, void* workspace,
                             size_t workspace_bytes, chebgcn_stream stream);

/* ---- gradients of the contraction (MatMul grads) ---------------------------------
 * dW[fin*K+k][o] = sum_{b,m} stack[k][b][fin][m] * dy[b][o][m]      (overwritten)
 * gstack[k][b][fin][m] = sum_o dy[b][o][m] * W[fin*K+k][o]
 * workspace: device scratch of at least chebgcn_contract_bwd_w_workspace() bytes. */
size_t chebgcn_contract_bwd_w_workspace(int B, int M, int Fin, int K, int Fout);
int chebgcn_contract_bwd_w(const float* stack, const float* dy, float* dW, void* workspace,
                           size_t workspace_bytes, int B, int M, int Fin, int K, int Fout,
                           chebgcn_stream stream);
/* The same two gradients with the ReluGrad of the reference's autodiff folded in (pool == 1
 * layers): dout = d(loss)/d(layer output) [B][Fout][Mp], relu_mask as written by contract_fwd;
 * dy = dout where the mask bit is set, else 0, is formed in registers and never stored.  The bias
 * gradient of such a layer: chebgcn_brelu_pool_bwd(dout, NULL, relu_mask, dy = NULL, dbias, ...). */
int chebgcn_contract_bwd_w_relu(const float* stack, const float* dout, const uint8_t* relu_mask,
                                float* dW, void* workspace, size_t workspace_bytes, int B, int M,
                                int Fin, int K, int Fout, chebgcn_stream stream);
/* chebgcn_contract_bwd_w_relu with the per-vertex bias gradient of the layer (b2relu, models_gcn.py:625-629: dbias[o][m] = sum over
 * the windows of the gated dout, [Fout][Mp]) in its second launch -- the one that adds the per-workgroup partials of dW: an
 * atlas-sized layer's backward is a chain of ~5 us launches, and chebgcn_brelu_pool_bwd(dout, NULL, relu_mask, NULL, dbias, ...) was
 * one of them.  Same sums in the same order as the two calls it replaces.  Served where chebgcn_contract_bwd_w_relu_bias_merged()
 * returns 1 (few partials -- a small launch -- and the bias reduction's small-graph shape); CHEBGCN_EUNSUPPORTED otherwise. */
int chebgcn_contract_bwd_w_relu_bias_merged(int B, int M, int Fin, int K, int Fout);
int chebgcn_contract_bwd_w_relu_bias(const float* stack, const float* dout, const uint8_t* relu_mask,
                                     float* dW, float* dbias, void* workspace, size_t workspace_bytes,
                                     int B, int M, int Fin, int K, int Fout, chebgcn_stream stream);
int chebgcn_contract_bwd_x_relu(const float* dout, const uint8_t* relu_mask, const float* W,
                                float* gstack, int B, int M, int Fin, int K, int Fout,
                                chebgcn_stream stream);
int chebgcn_contract_bwd_x(const float* dy, const float* W, float* gstack, int B, int M,
                           int Fin, int K, int Fout, chebgcn_stream stream);

/* ---- last conv layer + tf.reduce_mean(x, -1) (models_gcn.py:673) in one pass ----------------
 * The mean over the filters of the layer's (bias + ReLU) result is all the head reads (models_gcn.py:671-674), so the
 * last layer need not store its [B][Fout][Mp] output at all:
 *   chebgcn_contract_fwd_mean: mean_out[b][m] = (1/Fout) sum_o relu(y[b][o][m] + bias), [B][Mp]; relu_mask as
 *     chebgcn_contract_fwd writes it (may be NULL for inference).  pool = 1, ReLU.  Served where
 *     chebgcn_contract_fwd_mean_supported() returns 1 (4 <= Fout <= 32, a big launch, Fin*K*136 bytes <= 48 KB);
 *     CHEBGCN_EUNSUPPORTED otherwise (run chebgcn_contract_fwd + chebgcn_feature_mean_fwd).
 *   backward: d(loss)/d(y[b][o][m]) = gmean[b][m] for EVERY filter o, gmean = d(loss)/d(mean) / Fout, [B][Mp] (zero in the
 *     padding): the three gradients of a ReLU-folded layer read that one plane per window instead of a [B][Fout][Mp]
 *     tensor -- the _mean forms of chebgcn_contract_bwd_w_relu / _bwd_x_relu and of the bias reduction
 *     chebgcn_brelu_pool_bwd(dout, NULL, relu_mask, NULL, dbias, ...). */
int chebgcn_contract_fwd_mean_supported(int B, int M, int Fin, int K, int Fout);
int chebgcn_contract_fwd_mean(const float* stack, const float* W, const float* bias, int bias_kind,
                              float* mean_out, uint8_t* relu_mask, int B, int M, int Fin, int K, int Fout,
                              chebgcn_stream stream);
int chebgcn_contract_bwd_w_relu_mean(const float* stack, const float* gmean, const uint8_t* relu_mask,
                                     float* dW, void* workspace, size_t workspace_bytes, int B, int M,
                                     int Fin, int K, int Fout, chebgcn_stream stream);
int chebgcn_contract_bwd_x_relu_mean(const float* gmean, const uint8_t* relu_mask, const float* W,
                                     float* gstack, int B, int M, int Fin, int K, int Fout,
                                     chebgcn_stream stream);
int chebgcn_bias_grad_relu_mean(const float* gmean, const uint8_t* relu_mask, float* dbias, int bias_kind,
                                int B, int M, int F, void* workspace, size_t workspace_bytes,
                                chebgcn_stream stream);
/* The same pass also writing the gated gradient dy[b][o][m] = relu_mask bit ? gmean[b][m] : 0, [B][F][Mp] -- the input of
 * chebgcn_recurrence_fwd_t when the layer's gradient wrt its input is formed by the forward recurrence on dy. */
int chebgcn_relu_grad_mean(const float* gmean, const uint8_t* relu_mask, float* dy, float* dbias, int bias_kind,
                           int B, int M, int F, void* workspace, size_t workspace_bytes,
                           chebgcn_stream stream);

/* ---- atlas-sized graphs: one Chebyshev layer per launch, on chip ---------------------------------
 * models_gcn.py:587-629 (chebyshev5 + b1relu / b2relu, no pooling) for graphs of at most 384 vertices (the
 * reference's own atlases: 246..360 regions, configure_fmri.py:11) and Fin, Fout <= 32: a workgroup carries a whole
 * window through the recurrence in LDS / registers and multiplies every T_k with W_k on the matrix cores while it is
 * still on chip -- the stack is written only if `stack` is non-NULL (training: chebgcn_contract_bwd_w reads it), never
 * read; slab 0 may be x itself.  Same operands, layouts, results (fp32 round-off) and ReLU mask as
 * chebgcn_recurrence_fwd + chebgcn_contract_fwd(pool = 1).  chebgcn_fused_layer_supported() says whether a shape is
 * served (else CHEBGCN_EUNSUPPORTED).
 *   chebgcn_fused_layer_bwd_x: d(loss)/dx of the same layer from d(loss)/d(output) (gated by relu_mask unless NULL):
 *   G_j = dy W_j^T on the matrix cores feeding the adjoint recurrence directly -- replaces chebgcn_contract_bwd_x[_relu]
 *   + chebgcn_recurrence_bwd, no gradient stack in memory. */
int chebgcn_fused_layer_supported(const chebgcn_graph* g, int B, int Fin, int K, int Fout);
/* Device scratch the forward needs (0 for large batches): with fewer than ~3/4 of a window per CU a window is split between
 * two workgroups (half of the input planes each) whose partial sums a second small launch adds in a fixed order. */
size_t chebgcn_fused_layer_workspace(const chebgcn_graph* g, int B, int Fin, int K, int Fout);
int chebgcn_fused_layer_fwd(const chebgcn_graph* g, const float* x, const float* W, const float* bias, int bias_kind,
                            float* stack, float* out, uint8_t* relu_mask, void* workspace, size_t workspace_bytes,
                            int B, int Fin, int K, int Fout, int relu, chebgcn_stream stream);
int chebgcn_fused_layer_bwd_x(const chebgcn_graph* g, const float* dout, const uint8_t* relu_mask, const float* W,
                              float* dx, int B, int Fin, int K, int Fout, chebgcn_stream stream);

/* ---- layout / staging -----------------------------------------------------------
 * perm_data: coarsening.perm_data_3d (lib_new/coarsening.py:244-265) fused with the
 * fp32 cast and batch gather of fit() (models_gcn.py:138-146):
 *   out[s][f][i] = perm[i] < N ? x[sample[s]][perm[i]][f] : 0,   i < M
 * x: [S_total][N][F] row layout (device); perm: int32 [M] (device); sample: int32 [S]
 * (device) or NULL for the identity; out: [S][F][Mp(M)].
 * to_plane / from_plane convert between the reference's [B][M][F] and plane layout. */
int chebgcn_perm_data(const float* x, const int32_t* perm, const int32_t* sample, float* out,
                      int S, int N, int M, int F, chebgcn_stream stream);
int chebgcn_to_plane(const float* x_bmf, float* out_plane, int B, int M, int F, chebgcn_stream stream);
int chebgcn_from_plane(const float* x_plane, float* out_bmf, int B, int M, int F, chebgcn_stream stream);

/* ---- head: tf.reduce_mean(x, -1) (models_gcn.py:673) and its gradient -----------
 * x: [B][F][Mp(M)] -> y: [B][M] (dense);  dy: [B][M] -> dx: [B][F][Mp(M)]. */
int chebgcn_feature_mean_fwd(const float* x, float* y, int B, int M, int F, chebgcn_stream stream);
int chebgcn_feature_mean_bwd(const float* dy, float* dx, int B, int M, int F, chebgcn_stream stream);

/* ---- head: fully connected layer (models_gcn.py:650-656) -------------------------
 *   y[b][o] = act( sum_i x[b][i] * W[i][o] + bias[o] ),  act = ReLU if relu else identity
 * x: [B] rows of I floats, row stride ldx floats (a [B, M] view of a [B, Mp] buffer is fine; ldx % 4 == 0, x 16-byte
 * aligned, else CHEBGCN_EUNSUPPORTED); W: [I][O] dense; bias: [O] or NULL; y: [B][O] dense.  32 x 32 tiles of y, eight waves
 * split the reduction; long reductions are also split across workgroups (partials in `workspace`,
 * chebgcn_fc_fwd_workspace bytes, 0 for short ones) and added in a fixed order by a second launch.
 * chebgcn_fc_fwd_supported: B*O <= 2^20, I <= 2^20. */
int chebgcn_fc_fwd_supported(int B, int I, int O);
size_t chebgcn_fc_fwd_workspace(int B, int I, int O);
int chebgcn_fc_fwd(const float* x, int64_t ldx, const float* W, const float* bias, float* y, void* workspace,
                   size_t workspace_bytes, int B, int I, int O, int relu, chebgcn_stream stream);
/* The layer's three gradients (TF autodiff of :650-656): with gm = g where y > 0, else 0 (ReluGrad; gm = g if y is NULL)
 *   dW[i][o] = sum_b x[b][i] gm[b][o],   db[o] = sum_b gm[b][o],   dx[b][i] = sum_o gm[b][o] W[i][o]
 * g, y: [B][O] dense; dW: [I][O]; db: [O] or NULL; dx: [B] rows of stride lddx, or NULL (first layer of a model whose input
 * needs no gradient); dW NULL skips dW and db.  Same range as chebgcn_fc_fwd; every sum in a fixed order. */
int chebgcn_fc_bwd(const float* x, int64_t ldx, const float* W, const float* g, const float* y, float* dW, float* db,
                   float* dx, int64_t lddx, int B, int I, int O, chebgcn_stream stream);

/* The two scalars a captured training step reads from device memory -- Adam's step size lr_t of this step (models_gcn.py:296:
 * tf.train.AdamOptimizer's lr * sqrt(1 - b2^t) / (1 - b1^t)) and the read factor of the loss average (models_gcn.py:269-275) --,
 * written in one launch in front of the graph's replay: dst[0] = v0, dst[1] = v1. */
int chebgcn_set_scalars(float* dst, float v0, float v1, chebgcn_stream stream);

/* Adam as above (lr_t by value, or read from *lr_t_dev when that is not NULL) which also leaves the sum of squares of the
 * PRE-update variables -- the L2 term of the loss, models_gcn.py:262-266 -- as chebgcn_adam_partials(n) per-workgroup partial
 * sums in sq_partials; and the rest of the loss bookkeeping of a step in one launch:
 *   loss = *cross_entropy + half_reg * sum(sq_partials[0..nparts));   *ema += (1 - decay) * (loss - *ema)   (the
 *   tf.train.ExponentialMovingAverage(0.9) of :269-275);   *loss_average_out = *ema * corr   (corr read from *corr_dev when set:
 *   the zero-debiasing factor 1 / (1 - decay^t));   *loss_out = loss when not NULL.  Fixed-order sums. */
int chebgcn_adam_partials(int64_t n);
int chebgcn_adam_step_sq(float* p, const float* g, float* m, float* v, int64_t n, float lr_t, const float* lr_t_dev,
                         float beta1, float beta2, float eps, float grad_scale, float l2, float* sq_partials,
                         chebgcn_stream stream);
/* ... over ALL variables of a model in one launch: elements [0, n_reg) as chebgcn_adam_step_sq (L2 term l2 * p in the gradient,
 * counted in the sum of squares), elements [n_reg, n) -- the biases, which models_gcn.py:262-266 does not regularise -- plain Adam. */
int chebgcn_adam_step_sq_all(float* p, const float* g, float* m, float* v, int64_t n, int64_t n_reg, float lr_t,
                             const float* lr_t_dev, float beta1, float beta2, float eps, float grad_scale, float l2,
                             float* sq_partials, chebgcn_stream stream);
int chebgcn_loss_bookkeeping(const float* cross_entropy, const float* sq_partials, int nparts, float half_reg, float* ema,
                             float decay, float corr, const float* corr_dev, float* loss_out, float* loss_average_out,
                             chebgcn_stream stream);

/* ---- loss: tf.nn.sparse_softmax_cross_entropy_with_logits + tf.reduce_mean (models_gcn.py:257-259) and its gradient wrt
 * the logits, one launch:  *loss = mean_b( logsumexp(z_b) - z_b[y_b] ),  dlogits[b][c] = (softmax(z_b)[c] - [c == y_b]) / B.
 * logits, dlogits: [B][C] dense; labels: [B] int32 (labels_int64 = 0) or int64 (1), values in [0, C); loss: one float.
 * A label outside [0, C) makes *loss and that row of dlogits NaN (what TensorFlow's GPU kernel returns for it; nothing is
 * read or written outside the row).  Deterministic (fixed-order sums). */
int chebgcn_softmax_xent(const float* logits, const void* labels, int labels_int64, float* loss, float* dlogits, int B,
                         int C, chebgcn_stream stream);

/* ---- optimizer: tf.train.AdamOptimizer step (models_gcn.py:296, TF form) ---------
 * g' = grad_scale * g + l2 * p (per-segment l2 handled by the caller passing
 * segments);  m += (1-b1)(g'-m);  v += (1-b2)(g'^2-v);  p -= lr_t * m / (sqrt(v)+eps)
 * with lr_t = lr*sqrt(1-b2^t)/(1-b1^t) computed by the caller. */
int chebgcn_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr_t,
                      float beta1, float beta2, float eps, float grad_scale, float l2,
                      chebgcn_stream stream);
/* The same step with lr_t read from DEVICE memory when the kernel runs: the form a captured HIP graph of
 * the training step replays (the host writes this step's lr_t into *lr_t_dev ahead of the launch). */
int chebgcn_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* lr_t_dev,
                          float beta1, float beta2, float eps, float grad_scale, float l2,
                          chebgcn_stream stream);

/* ---- host-side index maps (no GPU): lib_new/coarsening.py -----------------------
 * metis_one_level (:120-166): one greedy matching pass, bit-exact incl. the
 * reference's row-length quirk.  rr/cc: int64 [nnz] sorted by rr; vv/weights in the
 * given precision; rid: int64 [N] visiting order; cluster_id: int32 [N] out.
 * The matching score vv*(1.0/weights[tid] + 1.0/weights[nid]) (:153) is evaluated as NumPy evaluates it
 * on scalars of that dtype: _f32 entirely in float32 (NumPy >= 2, NEP 50: a Python float does not widen
 * a float32 scalar), _f64 in float64, and _f32p on float32 inputs PROMOTED to float64 -- what NumPy 1.x
 * (the generation the reference was written for) does with a float32 graph; near-ties of the strict `>`
 * can fall differently between _f32 and _f32p (tests/golden/coarsen_unit_n*.npz holds both outcomes).
 * compute_perm (:168-215) for ONE level: children of `order` (length n_order) among
 * the fine vertices with `parent` (length n_fine); out must hold 2*n_order. */
int chebgcn_metis_one_level_f32(int64_t nnz, const int64_t* rr, const int64_t* cc, const float* vv,
                                const int64_t* rid, const float* weights, int64_t N, int32_t* cluster_id);
int chebgcn_metis_one_level_f32p(int64_t nnz, const int64_t* rr, const int64_t* cc, const float* vv,
                                 const int64_t* rid, const float* weights, int64_t N, int32_t* cluster_id);
int chebgcn_metis_one_level_f64(int64_t nnz, const int64_t* rr, const int64_t* cc, const double* vv,
                                const int64_t* rid, const double* weights, int64_t N, int32_t* cluster_id);
int chebgcn_compute_perm_level(const int32_t* parent, int64_t n_fine, const int64_t* order,
                               int64_t n_order, int64_t* out);


/* ---- vertex order for the ordered recurrence kernels (host only) ----
 * The reference leaves the numbering of a graph's vertices to its caller (the coarsening's tree order, coarsening.py:168-215);
 * the network is invariant under a relabelling as long as everything per-vertex follows (cgcnn.vertex_order).  A graph whose
 * rows come sorted by descending length (graph.length_order) runs the ordered kernels; chebgcn_bank_order refines such an
 * order INSIDE its classes of equal row length so that the LDS reads of the gather spread over the banks (csrc/graph.hip).
 *   rowptr / colidx: CSR structure of the rescaled Laplacian in the length-sorted numbering (host memory);
 *   sweeps: passes of the pairwise-swap descent (0 = identity); perm_out[new label] = old label, M entries;
 *   stats (optional, 3 values): sum over the gather's lane sets of the fullest bank group before / after, swaps made.
 * Returns the identity where no ordered kernel serves the graph (nothing to gain).  Deterministic. */
int chebgcn_bank_order(int M, const int32_t* rowptr, const int32_t* colidx, int sweeps, int32_t* perm_out, int64_t* stats);

#ifdef __cplusplus
}
#endif
#endif /* CHEBGCN_H */
