"""Oracle (test infrastructure, see oracle/__init__.py): multi-threaded CPU restatement.

The same network as ``layers_ref.Net`` (lib_new/models_gcn.py:587-682 forward, :253-276
loss, TF-form Adam :296) on torch CPU tensors: ``torch.sparse_csr`` SpMM for the recurrence
(:598-610), ``matmul`` for the contraction (:611-617), autograd for the gradients.  It exists
for ONE purpose -- BASELINE.md section 2, baseline B2: a CPU figure that uses all host cores
(SciPy's SpMM in ``layers_ref`` is single-threaded), timed by ``bench.py``'s ``cpu_baseline``
leg with a warm-up and a median of repeats.  tests/test_oracle_layers.py checks it against
``layers_ref`` so that what is timed is the same arithmetic.  Never imported by the product.
"""
import numpy as np
import scipy.sparse as sp
import torch

from . import graph_ref


def _csr(L):
    Lr = sp.csr_matrix(graph_ref.rescale_L(sp.csr_matrix(L), lmax=2).astype(np.float32))
    Lr.sort_indices()
    return torch.sparse_csr_tensor(torch.as_tensor(Lr.indptr.astype(np.int64)), torch.as_tensor(Lr.indices.astype(np.int64)),
                                   torch.as_tensor(Lr.data), size=Lr.shape)


class TorchNet:
    def __init__(self, L, F, K, p, M, channel, brelu='b1relu', regularization=0.0):
        self.L, j = [], 0
        cache = {}
        for pp in p:
            if id(L[j]) not in cache:
                cache[id(L[j])] = _csr(L[j])
            self.L.append(cache[id(L[j])])
            j += int(np.log2(pp)) if pp > 1 else 0
        self.F, self.K, self.p, self.M = list(F), list(K), list(p), list(M)
        self.channel, self.brelu, self.regularization = channel, brelu, regularization

    def regularized(self, name):
        return not (name.startswith('conv') and name.endswith('bias'))

    def chebyshev5(self, x, L, W, K):
        """models_gcn.py:587-617 on x[N, M, Fin]."""
        N, M, Fin = x.shape
        x0 = x.permute(1, 2, 0).reshape(M, Fin * N)
        xs = [x0]
        if K > 1:
            xs.append(torch.sparse.mm(L, x0))
        for _ in range(2, K):
            xs.append(2 * torch.sparse.mm(L, xs[-1]) - xs[-2])
        T = torch.stack(xs).reshape(K, M, Fin, N).permute(3, 1, 2, 0).reshape(N * M, Fin * K)
        return (T @ W).reshape(N, M, W.shape[1])

    def forward(self, params, x):
        h = x
        for i in range(len(self.p)):
            h = self.chebyshev5(h, self.L[i], params['conv%d/weights' % (i + 1)], self.K[i])
            h = torch.relu(h + params['conv%d/bias' % (i + 1)])
            if self.p[i] > 1:
                N, M, F = h.shape
                h = h.reshape(N, M // self.p[i], self.p[i], F).amax(dim=2)
        h = h.mean(dim=-1)
        for i in range(len(self.M) - 1):
            h = torch.relu(h @ params['fc%d/weights' % (i + 1)] + params['fc%d/bias' % (i + 1)])
        return h @ params['logits/weights'] + params['logits/bias']

    def loss(self, params, logits, labels):
        ce = torch.nn.functional.cross_entropy(logits, labels)
        reg = sum(0.5 * (v * v).sum() for k, v in params.items() if self.regularized(k))
        return ce + self.regularization * reg

    def train_step(self, params, x, labels, state, lr=0.001, beta1=0.9, beta2=0.999, eps=1e-8):
        """forward + loss + backward + TF-form Adam, in place.  Returns the loss."""
        for v in params.values():
            v.requires_grad_(True)
            v.grad = None
        loss = self.loss(params, self.forward(params, x), labels)
        loss.backward()
        state['t'] = t = state.get('t', 0) + 1
        lr_t = lr * np.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
        with torch.no_grad():
            for k, v in params.items():
                m = state.setdefault('m/' + k, torch.zeros_like(v))
                s = state.setdefault('v/' + k, torch.zeros_like(v))
                g = v.grad
                m += (1 - beta1) * (g - m)
                s += (1 - beta2) * (g * g - s)
                v -= lr_t * m / (s.sqrt() + eps)
        return float(loss.detach())
