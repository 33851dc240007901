"""The oracle's restatement of the training / evaluation loop (oracle/loop_ref.py <-
lib_new/models_gcn.py:31-184) on CPU: step count, without-replacement sampling driven by the
global NumPy RNG, evaluation cadence, EMA, and the zero-padded last batch of predict."""
import collections

import numpy as np

from conftest import csr_from, load_golden
from oracle import layers_ref as R
from oracle import loop_ref as LR


def _net():
    z = load_golden('inference_pool_n212')
    Ls = [csr_from(z, 'L%d' % i) for i in range(int(z['nlevels']))]
    net = R.Net(Ls, z['F'].tolist(), z['K'].tolist(), z['p'].tolist(), z['M'].tolist(), channel=int(z['channel']),
                brelu=str(z['brelu']), regularization=5e-4)
    params = {k[len('param:'):]: z[k].copy() for k in z.files if k.startswith('param:')}
    return net, params, Ls[0].shape[0], int(z['channel']), int(z['M'][-1])


def test_fit_sampling_and_cadence():
    net, params, M0, C, nclass = _net()
    rs = np.random.RandomState(0)
    S, B = 11, 4
    data, labels = rs.randn(S, M0, C), rs.randint(0, nclass, S)
    np.random.seed(5)
    log = LR.fit(net, params, data, labels, data[:6], labels[:6], num_epochs=2, batch_size=B, eval_frequency=2)
    assert log['num_steps'] == int(2 * S / B) == 5 and log['eval_steps'] == [2, 4, 5]
    # the index stream is the concatenation of permutations drawn whenever < B indices are left (:137-140)
    np.random.seed(5)
    dq, want = collections.deque(), []
    for _ in range(5):
        if len(dq) < B:
            dq.extend(np.random.permutation(S))
        want.append([dq.popleft() for _ in range(B)])
    assert [i.tolist() for i in log['idx']] == want
    first = np.concatenate(log['idx'])[:S]
    assert sorted(first[:8].tolist()) == sorted(set(first[:8].tolist()))       # no repeats inside one permutation
    assert len(log['loss_average']) == 5 and all(np.isfinite(log['loss_average']))
    assert all(0 <= a <= 100 for a in log['accuracies'])


def test_predict_pads_last_batch_with_zero_windows():
    net, params, M0, C, nclass = _net()
    rs = np.random.RandomState(1)
    data, labels = rs.randn(6, M0, C), rs.randint(0, nclass, 6)
    pred, loss = LR.predict(net, params, data, labels, batch_size=4)
    # by hand: batch 1 = windows 0..3; batch 2 = windows 4, 5 and two all-zero windows labelled 0 (:40-54)
    l1 = net.loss(params, net.forward(params, data[:4].astype(np.float32))[0], labels[:4])[0]
    pad = np.zeros((4, M0, C), np.float32)
    pad[:2] = data[4:]
    lg2 = net.forward(params, pad)[0]
    l2 = net.loss(params, lg2, np.array([labels[4], labels[5], 0, 0]))[0]
    assert np.isclose(loss, (l1 + l2) * 4 / 6, rtol=1e-6)
    assert pred.dtype == np.float64 and np.array_equal(pred[4:], lg2.argmax(1)[:2])
    assert np.array_equal(LR.predict(net, params, data, None, 4), pred)
