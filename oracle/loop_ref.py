"""Oracle (test infrastructure, see oracle/__init__.py): the training / evaluation loop.

NumPy restatement of ``base_model.predict`` (lib_new/models_gcn.py:31-71), ``evaluate``
(:73-110) and ``fit`` (:112-184) around ``layers_ref.Net`` -- the callers of the hot path
(SURVEY.md 8(f)1).  What is restated, with the reference lines:

* ``predict``: batches of ``batch_size``; the last batch is ZERO-PADDED, inputs and labels
  (:40-54), so its loss is the mean over ``batch_size`` windows of which the pads are
  all-zero windows with label 0; NaN/Inf batch losses count as 0 (:59-60); the returned loss
  is ``sum(batch losses) * batch_size / size`` (:68).
* ``fit``: ``int(num_epochs * S / batch_size)`` steps (:131); every sample is used before one
  is used a second time -- a deque refilled with ``np.random.permutation(S)`` from the GLOBAL
  NumPy RNG whenever fewer than ``batch_size`` indices are left (:137-140); one TF-form Adam
  step per batch (:146, :296); reported ``loss_average`` = 0.9-EMA of the pre-update loss with a
  zero-initialised shadow (:269-275).  ASSUMPTION (TensorFlow is absent):
  ``tf.train.ExponentialMovingAverage(0.9)`` without ``zero_debias`` on TF >= 1.0 reports the shadow
  as is (first value 0.1 * loss) -- ``zero_debias=False``, the default here; TF 0.12 debiased averages
  of tensors (shadow / (1 - 0.9^t)) -- ``zero_debias=True``; every ``eval_frequency`` steps and at the last step the
  validation set is scored with ``predict`` (:153-158).
* dropout: ``keep_prob`` is fed (:145); the oracle supports keep_prob == 1 only (TF's dropout
  RNG stream cannot be reproduced).

Returns everything a parity test wants to compare: the sampled index sequence, the EMA loss
series, validation accuracies and losses.
"""
import collections

import numpy as np

from . import layers_ref as R


def predict(net, params, data, labels, batch_size):
    """models_gcn.py:31-71.  Returns (predictions float64 [S], loss) or predictions."""
    size = data.shape[0]
    predictions = np.empty(size)
    loss = 0
    for begin in range(0, size, batch_size):
        end = min([begin + batch_size, size])
        batch_data = np.zeros((batch_size,) + data.shape[1:])
        batch_data[:end - begin] = data[begin:end]
        logits, _ = net.forward(params, batch_data.astype(np.float32))
        if labels is not None:
            batch_labels = np.zeros(batch_size)
            batch_labels[:end - begin] = labels[begin:end]
            batch_loss, _ = net.loss(params, logits, batch_labels.astype(np.int64))
            if np.isnan(batch_loss) or np.isinf(batch_loss):
                batch_loss = 0
            loss += batch_loss
        predictions[begin:end] = np.argmax(logits, axis=1)[:end - begin]
    if labels is not None:
        return predictions, loss * batch_size / size
    return predictions


def accuracy(predictions, labels):
    """100 * sklearn.metrics.accuracy_score (:104)."""
    return 100.0 * float(np.mean(np.asarray(predictions) == np.asarray(labels)))


def fit(net, params, train_data, train_labels, val_data, val_labels, num_epochs, batch_size, eval_frequency, zero_debias=False):
    """models_gcn.py:112-184 with dropout keep_prob = 1.  ``params`` is updated in place.
    Consumes ``np.random`` exactly like the reference (one permutation per refill)."""
    n = train_data.shape[0]
    indices = collections.deque()
    num_steps = int(num_epochs * n / batch_size)
    state = {}
    log = {'idx': [], 'loss_average': [], 'eval_steps': [], 'accuracies': [], 'losses': [], 'num_steps': num_steps}
    shadow = 0.0
    for step in range(1, num_steps + 1):
        if len(indices) < batch_size:
            indices.extend(np.random.permutation(n))
        idx = [indices.popleft() for _ in range(batch_size)]
        log['idx'].append(np.array(idx))
        x, y = train_data[idx, :, :].astype(np.float32), np.asarray(train_labels)[idx]
        logits, cache = net.forward(params, x)
        loss, dlogits = net.loss(params, logits, y)
        grads = net.backward(params, cache, dlogits)
        R.adam_tf_step(params, grads, state)
        shadow = 0.9 * shadow + 0.1 * loss                       # ExponentialMovingAverage(0.9), zero-initialised shadow
        log['loss_average'].append(shadow / (1 - 0.9 ** step) if zero_debias else shadow)
        if step % eval_frequency == 0 or step == num_steps:
            pred, vloss = predict(net, params, val_data, val_labels, batch_size)
            log['eval_steps'].append(step)
            log['accuracies'].append(accuracy(pred, val_labels))
            log['losses'].append(vloss)
    return log
