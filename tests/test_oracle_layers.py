"""The oracle's hand-written backward / Adam against torch.autograd (float64)
and torch.optim-independent closed forms.  CPU only."""
import numpy as np
import pytest
import torch

from conftest import csr_from, load_golden
from oracle import layers_ref as R


def torch_forward(params, Ls, F, K, p, M, x, brelu, reg):
    """Literal torch transcription of models_gcn.py:587-682 + :253-276 with a
    dense rescaled Laplacian -- independent of the oracle's NumPy code."""
    h = x
    for i in range(len(p)):
        Lr = torch.tensor(R.rescaled_laplacian(Ls[i], np.float64).toarray())
        W, b = params['conv%d/weights' % (i + 1)], params['conv%d/bias' % (i + 1)]
        N, Mi, Fin = h.shape
        x0 = h.permute(1, 2, 0).reshape(Mi, Fin * N)
        stack = [x0]
        if K[i] > 1:
            stack.append(Lr @ x0)
        for k in range(2, K[i]):
            stack.append(2 * (Lr @ stack[-1]) - stack[-2])
        xs = torch.stack(stack).reshape(K[i], Mi, Fin, N).permute(3, 1, 2, 0).reshape(N * Mi, Fin * K[i])
        h = torch.relu((xs @ W).reshape(N, Mi, -1) + b)
        if p[i] > 1:
            h = torch.nn.functional.max_pool1d(h.permute(0, 2, 1), p[i]).permute(0, 2, 1)
    h = h.mean(-1)
    for i in range(len(M) - 1):
        h = torch.relu(h @ params['fc%d/weights' % (i + 1)] + params['fc%d/bias' % (i + 1)])
    return h @ params['logits/weights'] + params['logits/bias']


@pytest.mark.parametrize('brelu', ['b1relu', 'b2relu'])
def test_backward_matches_autograd(brelu):
    z = load_golden('layers_n212')
    Ls = [csr_from(z, 'L%d' % i).astype(np.float64) for i in range(4)]
    F, K, p, M, channel, reg = [3, 4, 5], [4, 1, 3], [2, 2, 1], [7, 5], 2, 5e-4
    net = R.Net(Ls, F, K, p, M, channel, brelu=brelu, regularization=reg, dtype=np.float64)
    rs = np.random.RandomState(0)
    params = {k: rs.randn(*s) * 0.3 for k, s in net.param_shapes().items()}
    x = rs.randn(3, Ls[0].shape[0], channel)
    labels = np.array([0, 4, 2])
    logits, cache = net.forward(params, x)
    loss, dlogits = net.loss(params, logits, labels)
    grads = net.backward(params, cache, dlogits)

    tp = {k: torch.tensor(v, requires_grad=True) for k, v in params.items()}
    tl = torch_forward(tp, net.L, F, K, p, M, torch.tensor(x), brelu, reg)
    np.testing.assert_allclose(logits, tl.detach().numpy(), rtol=1e-10, atol=1e-10)
    tloss = torch.nn.functional.cross_entropy(tl, torch.tensor(labels))
    tloss = tloss + reg * sum(0.5 * (v ** 2).sum() for k, v in tp.items() if net.regularized(k))
    assert abs(loss - tloss.item()) < 1e-10
    tloss.backward()
    for k in params:
        np.testing.assert_allclose(grads[k], tp[k].grad.numpy(), rtol=1e-9, atol=1e-11, err_msg=k)


def test_chebyshev5_dx_matches_autograd():
    z = load_golden('layers_n212')
    L = csr_from(z, 'L1').astype(np.float64)
    rs = np.random.RandomState(1)
    N, M, Fin, Fout, K = 2, L.shape[0], 3, 4, 6
    x, W, dy = rs.randn(N, M, Fin), rs.randn(Fin * K, Fout), rs.randn(N, M, Fout)
    y, T = R.chebyshev5_fwd(x, L, W, K, return_stack=True)
    dx, dW = R.chebyshev5_bwd(dy, L, W, K, T)
    tx = torch.tensor(x, requires_grad=True)
    tp = {'conv1/weights': torch.tensor(W, requires_grad=True), 'conv1/bias': torch.full((1, 1, Fout), 1e3, dtype=torch.float64)}
    Lr = torch.tensor(R.rescaled_laplacian(L, np.float64).toarray())
    x0 = tx.permute(1, 2, 0).reshape(M, Fin * N)
    st = [x0, Lr @ x0]
    for k in range(2, K):
        st.append(2 * (Lr @ st[-1]) - st[-2])
    xs = torch.stack(st).reshape(K, M, Fin, N).permute(3, 1, 2, 0).reshape(N * M, Fin * K)
    ty = (xs @ tp['conv1/weights']).reshape(N, M, Fout)
    np.testing.assert_allclose(y, ty.detach().numpy(), rtol=1e-11, atol=1e-11)
    (ty * torch.tensor(dy)).sum().backward()
    np.testing.assert_allclose(dx, tx.grad.numpy(), rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(dW, tp['conv1/weights'].grad.numpy(), rtol=1e-10, atol=1e-10)


def test_adam_tf_form():
    """p -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps): eps outside the bias
    correction (TF), so it differs from torch.optim.Adam; check by hand."""
    rs = np.random.RandomState(2)
    p0 = rs.randn(5)
    params, state = {'w': p0.copy()}, {}
    m = np.zeros(5); v = np.zeros(5); ref = p0.copy()
    for t in range(1, 4):
        g = rs.randn(5)
        R.adam_tf_step(params, {'w': g}, state)
        m = 0.9 * m + 0.1 * g
        v = 0.999 * v + 0.001 * g * g
        ref = ref - 0.001 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) * m / (np.sqrt(v) + 1e-8)
        np.testing.assert_allclose(params['w'], ref, rtol=1e-12)


def test_torch_cpu_restatement_matches_oracle():
    """oracle/torch_cpu_ref.py (the multi-threaded CPU baseline B2 that bench.py times) computes
    what oracle/layers_ref.py computes: logits, loss and one TF-form Adam step, fp32."""
    from oracle import torch_cpu_ref as TR
    z = load_golden('inference_pool_n212')
    Ls = [csr_from(z, 'L%d' % i) for i in range(int(z['nlevels']))]
    cfg = (z['F'].tolist(), z['K'].tolist(), z['p'].tolist(), z['M'].tolist())
    net = R.Net(Ls, *cfg, channel=int(z['channel']), brelu=str(z['brelu']), regularization=5e-4)
    tnet = TR.TorchNet(Ls, *cfg, channel=int(z['channel']), brelu=str(z['brelu']), regularization=5e-4)
    params = {k[len('param:'):]: z[k].copy() for k in z.files if k.startswith('param:')}
    tparams = {k: torch.tensor(v) for k, v in params.items()}
    x, labels = z['x'], np.arange(z['x'].shape[0]) % cfg[3][-1]
    logits, cache = net.forward(params, x)
    with torch.no_grad():
        tl = tnet.forward(tparams, torch.tensor(x)).numpy()
    np.testing.assert_allclose(tl, logits, rtol=2e-5, atol=2e-5 * np.abs(logits).max())
    loss, dlogits = net.loss(params, logits, labels)
    grads = net.backward(params, cache, dlogits)
    R.adam_tf_step(params, grads, {})
    tloss = tnet.train_step(tparams, torch.tensor(x), torch.tensor(labels), {})
    assert abs(tloss - loss) <= 2e-5 * abs(loss)
    for k in params:
        np.testing.assert_allclose(tparams[k].detach().numpy(), params[k], rtol=0, atol=2e-5 * np.abs(params[k]).max() + 1.1e-3 * 0)
