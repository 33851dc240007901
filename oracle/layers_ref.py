"""Oracle (test infrastructure, see oracle/__init__.py): cgcnn layers, loss, Adam.

NumPy restatement of the layer methods of ``cgcnn`` and the loss/optimizer of
``base_model`` in lib_new/models_gcn.py, each with a hand-derived backward
(the reference relies on TF autodiff, models_gcn.py:297-303):

* ``chebyshev5_fwd/bwd``  -- models_gcn.py:587-617
* ``brelu_fwd/bwd``       -- b1relu :619-623, b2relu :625-629
* ``mpool1_fwd/bwd``      -- :631-639  (``apool1`` :641-648)
* ``fc_fwd/bwd``          -- :650-656
* ``Net``                 -- _inference :658-682, loss :253-276,
                             TF-form Adam (:296)

Shapes follow the reference boundary: activations ``[N, M, F]``, conv weights
``[Fin*K, Fout]`` with row index ``fin*K + k`` (models_gcn.py:611-615).
Everything is computed in the dtype of the inputs (fp32 like the reference,
or fp64 when the tests want a tight comparison against autograd).
"""
import numpy as np
import scipy.sparse as sp

from . import graph_ref


# ----------------------------------------------------------------------------
# chebyshev5  (models_gcn.py:587-617)
# ----------------------------------------------------------------------------

def rescaled_laplacian(L, dtype):
    """csr(L) -> rescale_L(lmax=2) as in models_gcn.py:590-592, cast to dtype."""
    Lr = graph_ref.rescale_L(sp.csr_matrix(L), lmax=2)
    return sp.csr_matrix(Lr.astype(dtype))


def cheb_stack(Lr, x, K):
    """x[N, M, Fin] -> T[K, M, Fin, N]: the recurrence on x0[M, Fin*N] (:598-610)."""
    N, M, Fin = x.shape
    x0 = np.ascontiguousarray(x.transpose(1, 2, 0)).reshape(M, Fin * N)
    T = graph_ref.chebyshev(Lr, x0, K)
    return T.reshape(K, M, Fin, N)


def chebyshev5_fwd(x, L, W, K, return_stack=False):
    """y[N, M, Fout] = Xs[N*M, Fin*K] @ W  (models_gcn.py:611-617)."""
    N, M, Fin = x.shape
    Fout = W.shape[1]
    Lr = rescaled_laplacian(L, x.dtype)
    T = cheb_stack(Lr, x, K)
    Xs = T.transpose(3, 1, 2, 0).reshape(N * M, Fin * K)
    y = (Xs @ W).reshape(N, M, Fout)
    return (y, T) if return_stack else y


def chebyshev5_bwd(dy, L, W, K, T, need_dx=True):
    """Gradients of chebyshev5 given the saved stack T[K, M, Fin, N].

    dW[fin*K+k, o] = sum_{n,m} T[k,m,fin,n] dy[n,m,o]
    G[k,m,fin,n]   = sum_o dy[n,m,o] W[fin*K+k, o]
    adjoint recurrence (transpose of T_k = 2 L T_{k-1} - T_{k-2}):
        for k = K-1..2:  G[k-1] += 2 L^T G[k];  G[k-2] -= G[k]
        if K > 1:        G[0]   += L^T G[1]
    dx = G[0] back in [N, M, Fin].
    """
    Kk, M, Fin, N = T.shape
    Fout = W.shape[1]
    dW = np.einsum('kmfn,nmo->fko', T, dy).reshape(Fin * K, Fout)
    if not need_dx:
        return None, dW
    Lt = sp.csr_matrix(rescaled_laplacian(L, dy.dtype).T)
    W3 = W.reshape(Fin, K, Fout)
    G = np.einsum('nmo,fko->kmfn', dy, W3).reshape(K, M, Fin * N)
    G = np.ascontiguousarray(G)
    for k in range(K - 1, 1, -1):
        G[k - 1] += 2 * Lt.dot(G[k])
        G[k - 2] -= G[k]
    if K > 1:
        G[0] += Lt.dot(G[1])
    dx = G[0].reshape(M, Fin, N).transpose(2, 0, 1)
    return np.ascontiguousarray(dx), dW


# ----------------------------------------------------------------------------
# bias + relu, pooling, fully connected
# ----------------------------------------------------------------------------

def brelu_fwd(x, b):
    """relu(x + b); b is [1,1,F] (b1relu :619-623) or [1,M,F] (b2relu :625-629)."""
    return np.maximum(x + b, 0)


def brelu_bwd(dy, y, b_shape):
    """Given y = relu(x+b): dx = dy*[y>0]; db = dx summed over broadcast axes."""
    dx = dy * (y > 0)
    axes = tuple(i for i, s in enumerate(b_shape) if s == 1)
    db = dx.sum(axis=axes, keepdims=True)
    return dx, db


def mpool1_fwd(x, p):
    """max over p consecutive vertices (:631-639); identity for p == 1.

    SAME padding never contributes to a max; coarsened graphs give M % p == 0.
    """
    if p <= 1:
        return x, None
    N, M, F = x.shape
    Mo = -(-M // p)
    if Mo * p != M:
        pad = np.full((N, Mo * p - M, F), -np.inf, x.dtype)
        x = np.concatenate([x, pad], axis=1)
    xr = x.reshape(N, Mo, p, F)
    arg = xr.argmax(axis=2)                  # first maximum
    return xr.max(axis=2), arg


def mpool1_bwd(dy, arg, p, M):
    if p <= 1:
        return dy
    N, Mo, F = dy.shape
    dx = np.zeros((N, Mo, p, F), dy.dtype)
    n, m, f = np.meshgrid(np.arange(N), np.arange(Mo), np.arange(F), indexing='ij')
    dx[n, m, arg, f] = dy
    return dx.reshape(N, Mo * p, F)[:, :M, :]


def apool1_fwd(x, p):
    """mean over p consecutive vertices (:641-648)."""
    if p <= 1:
        return x
    N, M, F = x.shape
    return x.reshape(N, M // p, p, F).mean(axis=2)


def fc_fwd(x, W, b, relu=True):
    """x @ W + b, optional relu (:650-656)."""
    y = x @ W + b
    return np.maximum(y, 0) if relu else y


def fc_bwd(dy, x, W, y, relu=True):
    if relu:
        dy = dy * (y > 0)
    return dy @ W.T, x.T @ dy, dy.sum(axis=0)


def softmax_xent(logits, labels):
    """mean sparse softmax cross-entropy (:258-259) and its logits gradient."""
    z = logits - logits.max(axis=1, keepdims=True)
    lse = np.log(np.exp(z).sum(axis=1, keepdims=True))
    logp = z - lse
    n = logits.shape[0]
    loss = -logp[np.arange(n), labels].mean()
    g = np.exp(logp)
    g[np.arange(n), labels] -= 1
    return loss, (g / n).astype(logits.dtype)


# ----------------------------------------------------------------------------
# whole network: _inference (:658-682) + loss (:253-276) + Adam
# ----------------------------------------------------------------------------

class Net:
    """Parameters live in a dict keyed like the TF variable scopes:
    ``conv{i}/weights`` [Fin*K, Fout], ``conv{i}/bias`` [1,1,F] or [1,M,F],
    ``fc{i}/weights`` [Min, Mout], ``fc{i}/bias`` [Mout], ``logits/weights``,
    ``logits/bias`` (models_gcn.py:662, :675, :680; names from :343, :351)."""

    def __init__(self, L, F, K, p, M, channel, brelu='b1relu', pool='mpool1',
                 regularization=0.0, dtype=np.float32):
        # keep one Laplacian per conv layer: j advances by log2(p) (:463-469)
        self.L, j = [], 0
        for pp in p:
            self.L.append(L[j])
            j += int(np.log2(pp)) if pp > 1 else 0
        self.F, self.K, self.p, self.M = list(F), list(K), list(p), list(M)
        self.channel, self.brelu, self.pool = channel, brelu, pool
        self.regularization, self.dtype = regularization, dtype

    def param_shapes(self):
        shapes, Fin = {}, self.channel
        Mcur = self.L[0].shape[0]
        for i, (Fo, Kk, pp) in enumerate(zip(self.F, self.K, self.p)):
            Mi = self.L[i].shape[0]
            shapes['conv%d/weights' % (i + 1)] = (Fin * Kk, Fo)
            shapes['conv%d/bias' % (i + 1)] = (1, 1, Fo) if self.brelu == 'b1relu' else (1, Mi, Fo)
            Fin, Mcur = Fo, Mi // pp
        Min = Mcur                      # reduce_mean over features -> [N, M] (:673)
        for i, Mo in enumerate(self.M[:-1]):
            shapes['fc%d/weights' % (i + 1)] = (Min, Mo)
            shapes['fc%d/bias' % (i + 1)] = (Mo,)
            Min = Mo
        shapes['logits/weights'] = (Min, self.M[-1])
        shapes['logits/bias'] = (self.M[-1],)
        return shapes

    def regularized(self, name):
        """L2 terms: conv weights (:615) and every fc weight *and* bias
        (:653-654); conv biases are not regularised (:622, :628)."""
        return not (name.startswith('conv') and name.endswith('bias'))

    def forward(self, params, x, drop_masks=None):
        """Returns logits and a cache for backward.  ``drop_masks[i]`` is the
        already-scaled dropout multiplier (mask/keep_prob) of fc{i+1}."""
        cache = {'conv': [], 'fc': []}
        h = x.astype(self.dtype, copy=False)
        for i in range(len(self.p)):
            W = params['conv%d/weights' % (i + 1)]
            b = params['conv%d/bias' % (i + 1)]
            y, T = chebyshev5_fwd(h, self.L[i], W, self.K[i], return_stack=True)
            a = brelu_fwd(y, b)
            if self.pool == 'mpool1':
                o, arg = mpool1_fwd(a, self.p[i])
            else:
                o, arg = apool1_fwd(a, self.p[i]), None
            cache['conv'].append((T, a, arg, h.shape[1]))
            h = o
        cache['feat_F'] = h.shape[2]
        h = h.mean(axis=-1)
        for i in range(len(self.M) - 1):
            W, b = params['fc%d/weights' % (i + 1)], params['fc%d/bias' % (i + 1)]
            y = fc_fwd(h, W, b, relu=True)
            m = None if drop_masks is None else drop_masks[i]
            cache['fc'].append((h, y, m))
            h = y if m is None else y * m
        cache['logits_in'] = h
        return fc_fwd(h, params['logits/weights'], params['logits/bias'], relu=False), cache

    def loss(self, params, logits, labels):
        ce, dlogits = softmax_xent(logits, labels)
        reg = sum(0.5 * float((params[k].astype(np.float64) ** 2).sum())
                  for k in params if self.regularized(k))
        return ce + self.regularization * reg, dlogits

    def backward(self, params, cache, dlogits):
        grads = {}
        h = cache['logits_in']
        W = params['logits/weights']
        grads['logits/weights'] = h.T @ dlogits
        grads['logits/bias'] = dlogits.sum(axis=0)
        d = dlogits @ W.T
        for i in range(len(self.M) - 2, -1, -1):
            hin, y, m = cache['fc'][i]
            if m is not None:
                d = d * m
            d, gW, gb = fc_bwd(d, hin, params['fc%d/weights' % (i + 1)], y, relu=True)
            grads['fc%d/weights' % (i + 1)] = gW
            grads['fc%d/bias' % (i + 1)] = gb
        Fl = cache['feat_F']
        d = np.repeat(d[:, :, None] / Fl, Fl, axis=2).astype(self.dtype)
        for i in range(len(self.p) - 1, -1, -1):
            T, a, arg, Min = cache['conv'][i]
            if self.pool == 'mpool1':
                d = mpool1_bwd(d, arg, self.p[i], Min)
            elif self.p[i] > 1:
                d = np.repeat(d, self.p[i], axis=1) / self.p[i]
            b = params['conv%d/bias' % (i + 1)]
            d, gb = brelu_bwd(d, a, b.shape)
            grads['conv%d/bias' % (i + 1)] = gb
            d, gW = chebyshev5_bwd(d, self.L[i], params['conv%d/weights' % (i + 1)],
                                   self.K[i], T, need_dx=(i > 0))
            grads['conv%d/weights' % (i + 1)] = gW
        for k in grads:
            if self.regularized(k):
                grads[k] = grads[k] + self.regularization * params[k]
            grads[k] = grads[k].astype(self.dtype)
        return grads


def adam_tf_step(params, grads, state, lr=0.001, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer(0.001) update (models_gcn.py:296):
    lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; p -= lr_t*m/(sqrt(v)+eps)."""
    state['t'] = t = state.get('t', 0) + 1
    lr_t = lr * np.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    for k in params:
        m = state.setdefault('m/' + k, np.zeros_like(params[k]))
        v = state.setdefault('v/' + k, np.zeros_like(params[k]))
        g = grads[k]
        m += (1 - beta1) * (g - m)
        v += (1 - beta2) * (g * g - v)
        params[k] -= (lr_t * m / (np.sqrt(v) + eps)).astype(params[k].dtype)
    return params
