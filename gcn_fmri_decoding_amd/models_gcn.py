"""``cgcnn``: drop-in for ``lib_new/models_gcn.py`` of the reference (class ``cgcnn`` and
the parts of ``base_model`` its callers use), on PyTorch-ROCm with the graph-convolution
arithmetic in libchebgcn.so (gfx950).  Citations are to ``lib_new/models_gcn.py``.

Same constructor keywords (:445-448), same string-selected layer methods
(``filter='chebyshev5'``, ``brelu='b1relu'|'b2relu'``, ``pool='mpool1'|'apool1'``,
:504-506), same layer signatures (``chebyshev5(x, L, Fout, K)`` on ``x[N, M, Fin]`` ->
``[N, M, Fout]``, :587-617), same variable names and shapes (``conv{i}/weights``
``[Fin*K, Fout]`` with row ``fin*K + k``, ``conv{i}/bias``, ``fc{i}/weights`` ...), same
``fit`` / ``evaluate`` / ``predict`` call contract (:31-184).

What is different by design: activations stay on the GPU in plane layout between layers
(no transposes), a standard conv+bias+relu+pool layer runs as two fused kernels, the
gradients are hand-written kernels instead of TF autodiff, parameters live in one flat
buffer (one fused TF-form Adam launch, one RCCL all-reduce per bucket).
"""
import collections
import contextlib
import json
import math
import os
import shutil
import time

import numpy as np
import torch
import torch.nn.functional as Fnn

from . import graph as graph_mod
from . import ops
from ._lib import BIAS_FILTER, BIAS_NONE, BIAS_VERTEX, POOL_AVG, POOL_MAX, plane_stride


class _Spec:
    __slots__ = ('name', 'shape', 'kind', 'regularized', 'fan_in', 'group', 'ref_shape', 'vaxis', 'vkey')

    def __init__(self, name, shape, kind, regularized, fan_in, group, ref_shape, vaxis=None, vkey=None):
        self.name, self.shape, self.kind, self.regularized = name, tuple(shape), kind, regularized
        self.fan_in, self.group, self.ref_shape = fan_in, group, tuple(ref_shape)
        # axis of the STORED tensor that runs over the graph's vertices (per-vertex biases [F, Mp]: 1; the first FC layer's
        # weights [M, O]: 0), None otherwise: under a relabelled vertex order (cgcnn.vertex_order) that axis is stored in
        # the internal order and the accessors (variable / set_variable / state_dict) translate.  ``vkey`` names WHICH order:
        # the conv layer's index (the order of that layer's graph level) or 'head' (the order the last conv layer's output is in)
        self.vaxis, self.vkey = vaxis, vkey


class InternalPlanes(object):
    """A batch in plane storage ``[B, channel, Mp]`` whose vertex axis is ALREADY in a model's internal vertex order
    (``cgcnn.vertex_order``): what ``_gather`` and ``compose_perm`` staging produce and what ``train_step`` /
    ``_inference_storage`` take as is.  A plain tensor handed to those is in the caller's order and is relabelled.

    The order is a property of this wrapper, not of the tensor: nothing a tensor operation returns (a slice, ``.clone()``,
    ``.contiguous()``, ``.to()``, a data-parallel shard) can carry it along by accident or lose it silently.  Batch slices,
    ``clone`` and ``detach`` are offered here and stay wrapped; ``planes`` is the tensor for code that knows what it holds."""
    __slots__ = ('planes', 'owner')

    def __init__(self, planes, owner):
        if isinstance(planes, InternalPlanes):
            planes = planes.planes
        self.planes, self.owner = planes, owner

    shape = property(lambda self: self.planes.shape)
    device = property(lambda self: self.planes.device)

    def __len__(self):
        return self.planes.shape[0]

    def __getitem__(self, windows):
        """A slice / index set over the WINDOW axis only (the vertex axis must stay whole)."""
        if isinstance(windows, tuple):
            raise IndexError('InternalPlanes: index the window axis only')
        picked = self.planes[windows]
        return InternalPlanes(picked if picked.dim() == 3 else picked.unsqueeze(0), self.owner)

    def clone(self):
        return InternalPlanes(self.planes.clone(), self.owner)

    def detach(self):
        return InternalPlanes(self.planes.detach(), self.owner)

    def copy_(self, other):
        """Refill the buffer from another batch in the same internal order (a plain tensor here would be ambiguous)."""
        if not isinstance(other, InternalPlanes) or other.owner is not self.owner:
            raise TypeError('InternalPlanes.copy_: the source must be a batch in the same model\'s internal order')
        self.planes.copy_(other.planes)
        return self


class base_model(object):
    """Counterpart of ``base_model`` (:18-355): run-time interface + variable helpers."""

    def __init__(self, config=None):
        self.regularizers = []          # names of L2-regularised variables (:345, :353)
        self.sess = None                # kept for signature compatibility; unused
        self.config = config
        dev = None
        if isinstance(config, dict):
            dev = config.get('device')
        elif isinstance(config, (str, torch.device)):
            dev = config
        if dev is not None and torch.device(dev).type == 'meta':
            pass        # shape-only model: variables are laid out, nothing can be computed
        elif dev is None:
            if not torch.cuda.is_available():
                raise RuntimeError('cgcnn needs a ROCm GPU (MI355X); no device is visible and '
                                   'there is no CPU execution path')
            dev = torch.device('cuda', torch.cuda.current_device())
        self.device = torch.device(dev)
        self._scope = []
        self._specs = None              # filled during the build pass
        self._params = {}
        self._dp = None                 # optional data-parallel helper (dist.DataParallel)
        self._step_graph_on, self._sg, self._sg_warm = False, None, 0      # enable_step_graph()
        self._step_graph_user = None    # the caller's explicit enable_step_graph(True / False), None = never asked
        self._order = None              # internal vertex order of the INPUT level (cgcnn: graph.length_order), None = the caller's
        self._vtabs = {}                # per-variable vertex orders (_Spec.vkey -> index tables), empty = the caller's everywhere
        self.record_fit = False         # keep the sampled indices and the loss_average series of fit()
        # How the reported ``loss_average`` reads the 0.9-EMA of the loss (:269-275).  The reference calls
        # ``tf.train.ExponentialMovingAverage(0.9)`` on the loss TENSOR and never passes ``zero_debias``: TensorFlow >= 1.0
        # (what ``tf.contrib`` at :337 and the py3.6 bytecode imply) then keeps a zero-initialised shadow and reports it as
        # is -- the first printed value is 0.1 * loss -- while TensorFlow 0.12 debiased every tensor's average
        # (shadow / (1 - 0.9^t): the first value is the loss).  Reporting only; the default follows TF >= 1.0, set
        # ``ema_zero_debias = True`` (or CHEBGCN_EMA_ZERO_DEBIAS=1) for the 0.12 reading.  Unverifiable here (no TensorFlow).
        self.ema_zero_debias = os.environ.get('CHEBGCN_EMA_ZERO_DEBIAS', '0') not in ('0', '', 'false', 'False')

    # ---------------------------------------------------------------- run-time API

    def stage(self, data):
        """Copy a dataset ``[S, M, channel]`` (NumPy, any float dtype, or a torch tensor)
        to the device once, as fp32 in the reference's row layout.  ``fit``/``predict``
        accept the staged tensor directly; batches are then gathered on the GPU."""
        if isinstance(data, torch.Tensor):
            return data.to(self.device, torch.float32).contiguous()
        if not isinstance(data, np.ndarray):
            data = data.toarray()       # sparse matrices, like the reference (:42-46)
        return torch.as_tensor(np.ascontiguousarray(data, np.float32)).to(self.device)

    def _gather(self, data_dev, idx, out=None):
        """``data[idx]`` gathered on the GPU straight into plane storage [B, channel, Mp], in the model's internal vertex
        order (replaces the host gather + feed of :142-146).  ``out``: a buffer of that shape (``step_inputs()``)."""
        S, M, C = data_dev.shape
        if out is None or tuple(out.shape) != (int(idx.numel()), C, ops.plane_stride(M)):
            out = ops.plane_empty(int(idx.numel()), C, M, self.device)
        from . import _lib
        _lib.check(_lib.lib().chebgcn_perm_data(ops._p(data_dev), ops._p(self._order_dev) if self._order is not None else None,
                                                ops._p(idx), ops._p(out), int(idx.numel()), M, M, C, ops._stream()), 'perm_data')
        return self.as_internal(out)

    # ---- internal vertex order (cgcnn.vertex_order) -----------------------------------------------------------------
    def compose_perm(self, perm=None):
        """Index map (device int32) for ``ops.perm_data`` that takes raw data columns straight to the model's internal
        vertex order: ``perm[order]`` for the list ``coarsening.compute_perm`` returned (None: the identity).  Batches staged
        with it are wrapped with ``as_internal``."""
        perm = np.arange(self._M0, dtype=np.int64) if perm is None else np.asarray(perm, np.int64)
        if self._order is not None:
            perm = perm[self._order]
        return torch.as_tensor(perm.astype(np.int32)).to(self.device)

    def as_internal(self, x_storage):
        """Declares plane storage to be in this model's internal vertex order already: returns it wrapped as
        ``InternalPlanes`` (``train_step`` / ``_inference_storage`` relabel a plain tensor themselves, one gather kernel per
        batch).  The declaration lives in the wrapper, never on the tensor."""
        return InternalPlanes(x_storage, self)

    def _to_internal(self, x_storage):
        """The batch as a plain tensor in the internal vertex order."""
        if isinstance(x_storage, InternalPlanes):
            if x_storage.owner is not self and not self._same_order(x_storage.owner):
                raise ValueError('this batch is in the internal vertex order of another model')
            return x_storage.planes
        if self._order is None:
            return x_storage
        return x_storage.index_select(2, self._order_pad)

    def _same_order(self, other):
        a, b = self._order, getattr(other, '_order', None)
        return (a is None and b is None) or (a is not None and b is not None and np.array_equal(a, b))

    def predict(self, data, labels=None, sess=None):
        """Batched prediction (:31-71).  The last batch is zero-padded to ``batch_size``
        (inputs *and* labels) exactly like the reference, so the reported loss matches."""
        data_dev = self.stage(data)
        size = data_dev.shape[0]
        predictions = np.empty(size)
        loss = 0
        was_training = self.training_mode
        self.training_mode = False
        try:
            for begin in range(0, size, self.batch_size):
                end = min(begin + self.batch_size, size)
                idx = np.arange(begin, end)
                x = self._gather(data_dev, torch.as_tensor(idx, dtype=torch.int32).to(self.device))
                if end - begin < self.batch_size:
                    pad = ops.plane_empty(self.batch_size, x.shape[1], data_dev.shape[1], self.device, zero=True)
                    pad[:end - begin] = x.planes
                    x = self.as_internal(pad)
                with torch.no_grad():
                    logits = self._inference_storage(x, 1)
                    batch_pred = self.prediction(logits)
                    if labels is not None:
                        batch_labels = np.zeros(self.batch_size, np.int64)
                        batch_labels[:end - begin] = labels[begin:end]
                        batch_loss = float(self.loss(logits, torch.as_tensor(batch_labels).to(self.device),
                                                     self.regularization)[0])
                        if np.isnan(batch_loss) or np.isinf(batch_loss):
                            batch_loss = 0
                        loss += batch_loss
                predictions[begin:end] = batch_pred[:end - begin].cpu().numpy()
        finally:
            self.training_mode = was_training
        if labels is not None:
            return predictions, loss * self.batch_size / size
        return predictions

    def evaluate(self, data, labels, sess=None, target_name=None, isTrain=False):
        """One evaluation pass (:73-110): (string, accuracy %, weighted F1 %, loss)."""
        import sklearn.metrics
        if not isTrain:
            self._restore_latest()
        labels = np.asarray(labels)
        predictions, loss = self.predict(data, labels, sess)
        if target_name is not None:
            try:
                print(sklearn.metrics.classification_report(labels, predictions, labels=range(len(target_name)),
                                                            target_names=target_name))
                print('Confusion Matrix:')
                print(sklearn.metrics.confusion_matrix(labels, predictions, labels=range(len(target_name))))
            except Exception:
                print('No corresponding assignment between true and predicted labels')
        ncorrects = int(sum(predictions == labels))
        accuracy = 100 * sklearn.metrics.accuracy_score(labels, predictions)
        f1 = 100 * sklearn.metrics.f1_score(labels, predictions, average='weighted')
        string = 'accuracy: {:.2f} ({:d} / {:d}), f1 (weighted): {:.2f}, loss: {:.2e}'.format(
            accuracy, ncorrects, len(labels), f1, loss)
        # (the reference appends a 'time:' line only ``if sess is None``, which never holds there:
        # its local ``sess`` comes out of ``_get_session`` and is always a session, :88, :107-109)
        return string, accuracy, f1, loss

    def fit(self, train_data, train_labels, val_data, val_labels, best_checkpoint_dir=None):
        """Mini-batch training loop (:112-184): ``int(num_epochs*S/batch)`` steps, samples
        drawn without replacement from a shuffled deque (:137-140), evaluation on the
        validation set every ``eval_frequency`` steps, top-3 checkpoints by validation
        accuracy.  Returns (accuracies, losses, t_step).

        Under ``dist.DataParallel`` every rank calls ``fit`` with ITS OWN shard of the training set -- shards of
        equal size (checked: a rank with fewer samples would run fewer steps and the others would wait in the
        all-reduce forever) -- and a NumPy seed of its own (``np.random.seed(base + rank)``; the sampling below
        draws from the global NumPy RNG like the reference, so equal seeds on equal data would make every rank
        train on the same batches).  The validation set is evaluated by every rank (same variables, same
        result); rank 0 alone prints and owns the checkpoint directory."""
        t_process, t_wall = time.process_time(), time.time()
        # data parallel (dist.DataParallel): rank 0 owns the checkpoint directory, and every rank
        # restarts from rank 0's freshly drawn variables (the reference re-runs op_init here, :123)
        rank0 = self._dp is None or self._dp.rank == 0
        say = print if rank0 else (lambda *a, **k: None)
        if self._dp is not None:
            self._dp.check_equal(int(np.shape(train_data)[0]), 'fit(): training samples per rank')
        if rank0:
            shutil.rmtree(self._get_path('checkpoints'), ignore_errors=True)
            os.makedirs(self._get_path('checkpoints'), exist_ok=True)
        self._init_variables()
        if self._dp is not None:
            self._dp.broadcast_parameters()
        auto_graph = self._step_graph_user is None and not self._step_graph_on and self._auto_step_graph()
        if auto_graph:
            # atlas-sized graphs (what the reference trains on): the step is a chain of ~100 kernels of 3-60 us and launch-bound;
            # captured once as a HIP graph it is the same kernels on the same operands, bit for bit (enable_step_graph).
            # An explicit enable_step_graph(True / False) of the caller is left alone; fit() switches its own choice off
            # again when it returns
            self.enable_step_graph(True, _by_fit=True)
        try:
            return self._fit_loop(train_data, train_labels, val_data, val_labels, t_process, t_wall, rank0, say)
        finally:
            self.fit_captured = self._sg is not None      # whether the loop ran the captured step (tests, bench)
            if auto_graph:
                self.enable_step_graph(False, _by_fit=True)

    def _fit_loop(self, train_data, train_labels, val_data, val_labels, t_process, t_wall, rank0, say):
        train_dev, val_dev = self.stage(train_data), self.stage(val_data)
        train_labels = np.asarray(train_labels)
        n_classes = int(self.M[-1])
        if train_labels.size and (train_labels.min() < 0 or train_labels.max() >= n_classes):
            # (tf.nn.sparse_softmax_cross_entropy_with_logits raises on the CPU and returns NaN on the GPU, :257)
            raise ValueError('fit(): labels must lie in [0, %d); got %d ... %d' % (n_classes, train_labels.min(), train_labels.max()))
        labels_dev = torch.as_tensor(train_labels.astype(np.int64)).to(self.device)
        accuracies, losses = [], []
        best = []
        self.fit_log = {'idx': [], 'loss_average': []}     # filled when ``record_fit`` is set (tests)
        indices = collections.deque()
        n_train = train_dev.shape[0]
        num_steps = int(self.num_epochs * n_train / self.batch_size)
        say('training with {} steps in total with batch_size={} and epochs={} for training_set={}:'.format(
            num_steps, self.batch_size, self.num_epochs, n_train))
        pool_dev = pool_labels = None       # the deque's content on the device, uploaded when it is refilled (once per epoch)
        pool_at = 0
        for step in range(1, num_steps + 1):
            if len(indices) < self.batch_size:
                indices.extend(np.random.permutation(n_train))
                pool_dev = torch.as_tensor(np.asarray(indices, np.int32)).to(self.device)
                pool_labels = labels_dev[pool_dev.long()]
                pool_at = 0
            idx = [indices.popleft() for _ in range(self.batch_size)]
            idx_dev = pool_dev[pool_at:pool_at + self.batch_size]
            batch_labels = pool_labels[pool_at:pool_at + self.batch_size]
            pool_at += self.batch_size
            x = self._gather(train_dev, idx_dev, out=self.step_inputs()[0])      # (straight into the captured step's input buffer)
            learning_rate, loss_average = self.train_step(x, batch_labels)
            if self.record_fit:
                self.fit_log['idx'].append(np.asarray(idx))
                self.fit_log['loss_average'].append(loss_average)
            if step % self.eval_frequency == 0 or step == num_steps:
                loss_average = float(loss_average)
                if np.isnan(loss_average) or np.isinf(loss_average):
                    loss_average = 0
                epoch = step * self.batch_size / n_train
                say('step {} / {} (epoch {:.2f} / {}):'.format(step, num_steps, epoch, self.num_epochs))
                say('  learning_rate = {:.2e}, loss_average = {:.2e}'.format(learning_rate, loss_average))
                # a session is passed, as in the reference (:157): evaluate() then leaves its own time line out
                string, accuracy, f1, loss = self.evaluate(val_dev, val_labels, self._session(), isTrain=True)
                accuracies.append(accuracy)
                losses.append(loss)
                say('  validation {}'.format(string))
                say('  time: {:.0f}s (wall {:.0f}s)'.format(time.process_time() - t_process, time.time() - t_wall))
                if rank0:
                    self._save_best(accuracy, step, best)
                if self._dp is not None:
                    self._dp.barrier()          # checkpoint files are complete before any rank restores
        say('validation accuracy: peak = {:.2f}, mean = {:.2f}'.format(max(accuracies), np.mean(accuracies[-10:])))
        torch.cuda.synchronize(self.device)
        if self.record_fit:
            self.fit_log['loss_average'] = [float(v) for v in self.fit_log['loss_average']]
        t_step = (time.time() - t_wall) / num_steps
        return accuracies, losses, t_step

    # fit() captures the training step by itself where that pays: 'auto' (graphs of at most 512 vertices), True, False
    step_graph = 'auto'

    def _auto_step_graph(self):
        mode = os.environ.get('CHEBGCN_STEP_GRAPH', self.step_graph)
        if mode in (False, 0, '0', 'off', 'False'):
            return False
        ok = (self.device.type == 'cuda' and self.momentum != 0 and self._fusable()
              and (self._dp is None or self._dp.capturable))
        if mode in (True, 1, '1', 'on', 'True'):
            return ok
        graphs = getattr(self, 'graphs', None) or []
        # (N = 360: 1.1-1.3 ms eager, 0.82 captured; N = 1000: 1.90 eager, 1.95 captured -- GPU-bound chains gain nothing)
        return ok and bool(graphs) and all(g.Mp <= 512 for g in graphs)

    def _session(self):
        """Stand-in for the ``tf.Session`` the reference hands around (``sess`` arguments): there is
        no session here, callers only test it against None."""
        return self.sess if self.sess is not None else self

    def get_var(self, name):
        """Value of a variable by its TF name, in the reference's shape (:186-191)."""
        return self.variable(name).detach().cpu().numpy()

    # ---------------------------------------------------------------- graph building

    def build_graph(self, M_0, flag_input=True):
        """Creates every variable by tracing ``_inference`` once on shape-only (meta)
        tensors, then lays them out in one flat buffer (:195-222)."""
        self._specs = []
        x = torch.empty((self.batch_size,) + tuple(M_0), device='meta')
        self._inference(x, 1.0)
        specs, self._specs = self._specs, None
        order = {'head': 0, 'convw': 1, 'convb': 2}
        specs.sort(key=lambda s: order[s.group])        # stable: keeps creation order inside a group
        self._spec_list = specs
        sizes = [int(np.prod(s.shape)) for s in specs]
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        n = int(offs[-1])
        self._flat = torch.zeros(n, dtype=torch.float32, device=self.device)
        self._grad = torch.zeros(n, dtype=torch.float32, device=self.device)
        self._adam_m = torch.zeros(n, dtype=torch.float32, device=self.device)
        self._adam_v = torch.zeros(n, dtype=torch.float32, device=self.device)
        self._params, self._slices = {}, {}
        for s, a, b in zip(specs, offs[:-1], offs[1:]):
            p = torch.nn.Parameter(self._flat[a:b].view(s.shape))
            p.grad = self._grad[a:b].view(s.shape)
            self._params[s.name] = p
            self._slices[s.name] = (int(a), int(b))
        n_head = sum(z for s, z in zip(specs, sizes) if s.group == 'head')
        n_reg = sum(z for s, z in zip(specs, sizes) if s.regularized)
        if any(s.regularized for s in specs if s.group == 'convb') or not all(
                s.regularized for s in specs if s.group != 'convb'):
            raise AssertionError('flat layout assumes regularised variables come first')
        self._n_head, self._n_reg, self._n_total = n_head, n_reg, n
        self.regularizers = [s.name for s in specs if s.regularized]
        self.global_step = 0
        self._loss_ema = None
        self.training_mode = False
        self._init_variables()

    def _reset_counters(self):
        self.global_step = 0
        if getattr(self, '_loss_ema', None) is not None and getattr(self, '_sg', None) is not None:
            self._loss_ema.zero_()          # a captured step holds this tensor: reset it in place
        else:
            self._loss_ema = None

    def _init_variables(self):
        """``tf.global_variables_initializer`` (:213, run at :123)."""
        self._reset_counters()
        if self.device.type == 'meta':
            return
        with torch.no_grad():
            for s in self._spec_list:
                p = self._params[s.name]
                if s.kind == 'const':
                    if len(s.shape) == 2 and s.group == 'convb':       # [F, Mp]: keep the pad at zero
                        p.zero_()
                        p[:, :s.ref_shape[1]] = 0.2
                    else:
                        p.fill_(0.2)
                elif s.kind == 'normal':
                    torch.nn.init.trunc_normal_(p, 0.0, 0.2, -0.4, 0.4)
                else:   # 'he': variance_scaling_initializer(factor=2, FAN_IN, truncated normal)
                    std = math.sqrt(1.3 * 2.0 / s.fan_in)
                    torch.nn.init.trunc_normal_(p, 0.0, std, -2 * std, 2 * std)
            self._adam_m.zero_()
            self._adam_v.zero_()
            self._grad.zero_()
        self._reset_counters()

    def _spec(self, name):
        return next(s for s in self._spec_list if s.name == name)

    def _ref_tensor(self, t, spec):
        """A stored tensor (a variable, its gradient, an Adam moment) in the reference's shape and vertex order: a view
        where the layouts agree, a gathered copy under a relabelled vertex order."""
        tab = self._vtabs.get(spec.vkey) if spec.vaxis is not None else None
        if tab is not None:
            t = t.index_select(spec.vaxis, tab['inv'])
        if spec.group == 'convb':
            if len(spec.shape) == 2:                               # storage [F, Mp] -> [1, M, F]
                return t[:, :spec.ref_shape[1]].t().unsqueeze(0)
            return t.view(spec.ref_shape)                           # [F] -> [1, 1, F]
        return t

    def _ref_assign(self, t, spec, value):
        """``t`` (stored layout) <- ``value`` (reference shape and vertex order)."""
        v = value.to(self.device, torch.float32)
        if spec.group == 'convb':
            if len(spec.shape) == 2:
                v = v.reshape(spec.ref_shape)[0].t()               # [1, M, F] -> [F, M]
                tab = self._vtabs.get(spec.vkey)
                if tab is not None:
                    v = v.index_select(1, tab['order'])
                t[:, :spec.ref_shape[1]].copy_(v)
                return
            t.copy_(v.reshape(t.shape))
            return
        v = v.reshape(spec.shape)
        tab = self._vtabs.get(spec.vkey) if spec.vaxis == 0 else None
        if tab is not None:
            v = v.index_select(0, tab['order'])
        t.copy_(v)

    def variable(self, name):
        """The variable called ``name`` in the reference's shape and vertex order (a view of the parameter where the
        stored layout allows, else a copy: write through ``set_variable``)."""
        return self._ref_tensor(self._params[name], self._spec(name))

    def gradient(self, name):
        """d(loss)/d(variable) of the last step, in the reference's shape and vertex order."""
        return self._ref_tensor(self._params[name].grad, self._spec(name))

    def set_variable(self, name, value):
        with torch.no_grad():
            self._ref_assign(self._params[name], self._spec(name), torch.as_tensor(np.asarray(value, np.float32)))

    def variables(self):
        return [s.name for s in self._spec_list]

    def inference(self, data, dropout):
        """logits for ``data[N, M, channel]`` (device tensor, any layout) (:224-239)."""
        return self._inference(data, dropout)

    def probabilities(self, logits):
        return torch.softmax(logits, dim=1)

    def prediction(self, logits):
        return torch.argmax(logits, dim=1)

    def regularization_term(self):
        """sum of tf.nn.l2_loss over the regularised variables (:262, :345, :353)."""
        return 0.5 * self._flat[:self._n_reg].square().sum()

    def loss(self, logits, labels, regularization):
        """(loss, loss_average): mean softmax cross-entropy + regularization * sum l2_loss,
        and its 0.9-EMA as reported by the reference (:253-276)."""
        cross_entropy = Fnn.cross_entropy(logits, labels.long())
        loss = cross_entropy + regularization * self.regularization_term()
        return loss, loss

    def training(self, loss, learning_rate, decay_steps, decay_rate=0.95, momentum=0.9):
        """Reported learning rate of the current step (:278-313).  As in the reference the
        decayed rate is only reported: the optimizer is Adam(0.001) whenever momentum != 0
        (:291-296)."""
        if decay_rate != 1 and decay_steps:
            return learning_rate * decay_rate ** math.floor(self.global_step / decay_steps)
        return learning_rate

    def train_step(self, x_storage, labels):
        """One optimisation step on a batch in plane storage ``[B, channel, Mp]``:
        forward, loss, backward, (gradient all-reduce), TF-form Adam.  Returns
        (reported learning rate, loss_average tensor).  With ``enable_step_graph(True)`` the step is captured
        once as a HIP graph and replayed (small graphs: the step is a chain of short kernels and launch-bound)."""
        if (self._step_graph_on and (self._dp is None or self._dp.capturable) and self.momentum != 0 and self._fusable()
                and ops.timers is None):
            return self._train_step_graphed(x_storage, labels)
        t = self.global_step + 1
        loss_average = self._step_body(x_storage, labels, self._adam_lr_t(t), self._ema_read(t))
        reported_lr = self.training(None, self.learning_rate, self.decay_steps, self.decay_rate, self.momentum)
        self.global_step += 1
        return reported_lr, loss_average

    def _ema_read(self, t):
        """Factor between the EMA shadow after ``t`` updates and the reported ``loss_average`` (``ema_zero_debias``)."""
        return 1.0 / (1 - 0.9 ** t) if self.ema_zero_debias else 1.0

    @staticmethod
    def _adam_lr_t(t, lr=0.001, b1=0.9, b2=0.999):
        """Step size of step ``t`` of tf.train.AdamOptimizer(0.001) (:296): lr * sqrt(1 - b2^t) / (1 - b1^t)."""
        return lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)

    def _step_body(self, x_storage, labels, lr_t, ema_correction):
        """The device work of one step.  ``lr_t`` / ``ema_correction``: Python floats (eager) or one-element device
        tensors that are read when the kernels run (captured step)."""
        self.training_mode = True
        if not self._fusable():
            self._grad.zero_()          # autograd accumulates into .grad there; the fused path writes every gradient
        else:
            # the fused path WRITES each gradient; a variable that takes no gradient (requires_grad off) would keep
            # the slice of an earlier step -- already all-reduced under data parallelism -- and Adam would apply it again
            for name, p in self._params.items():
                if not p.requires_grad:
                    a, b = self._slices[name]
                    self._grad[a:b].zero_()
        if self._dp is not None:
            self._dp.begin_step()
        logits = self._inference_storage(x_storage, self.dropout)
        # loss and d(loss)/d(logits) in one launch; autograd starts from the logits
        cross_entropy, dlogits = ops.softmax_xent(logits, labels)
        logits.backward(dlogits)
        grad_scale = 1.0
        if self._dp is not None:
            grad_scale = self._dp.finish_step()
        if self._loss_ema is None:
            self._loss_ema = torch.zeros((), dtype=torch.float32, device=self.device)
        if self.momentum != 0 and self.device.type == 'cuda':
            # the loss of this step is evaluated on the PRE-update variables: Adam's pass over the regularised variables leaves
            # the partial sums of their squares (reg * sum l2_loss, :262-266), and one launch finishes the bookkeeping --
            # loss, tf.train.ExponentialMovingAverage(0.9) over it (zero-initialised shadow; read as is or zero-debiased: ema_zero_debias, :269-275)
            nparts = self._apply_adam(grad_scale, lr_t, want_sq=True)
            corr = ema_correction if isinstance(ema_correction, torch.Tensor) and ema_correction.is_cuda else float(ema_correction)
            if isinstance(corr, torch.Tensor) and corr.dim() == 0:
                corr = corr.reshape(1)
            loss_average = ops.loss_bookkeeping(cross_entropy, getattr(self, '_sq_part', None), nparts, 0.5 * self.regularization,
                                                self._loss_ema, corr)
        else:
            with torch.no_grad():
                v = self._flat[:self._n_reg]
                loss = torch.add(cross_entropy, torch.dot(v, v), alpha=0.5 * self.regularization)   # + reg * sum l2_loss
            self._apply_adam(grad_scale, lr_t)
            with torch.no_grad():
                self._loss_ema.lerp_(loss, 0.1)
                loss_average = self._loss_ema * ema_correction
        self.training_mode = False
        return loss_average

    # ---------------------------------------------------------------- captured step (HIP graph)

    def enable_step_graph(self, on=True, _by_fit=False):
        """Run ``train_step`` as ONE captured HIP graph (``torch.cuda.CUDAGraph`` on ROCm = hipGraph): the first two
        calls run eagerly (library initialisation), the third captures -- forward, loss, backward, Adam and the loss
        bookkeeping, the second stream of ``contract_bwd_w`` included -- and every later call copies the batch into
        the graph's input buffers, writes this step's two scalars (Adam's lr_t, the EMA's read factor) and
        replays.  Same kernels, same arithmetic, same results as the eager step; the batch shape must stay fixed.
        Under ``dist.DataParallel`` on RCCL the gradient all-reduces are captured with the step (they are enqueued on
        streams like kernels, forked from and joined to the capture stream) and replayed with it; on other backends
        (gloo) the step stays eager.  While per-kernel event timers are set (``ops.timers``) the step runs eagerly too."""
        if not _by_fit:
            self._step_graph_user = bool(on)      # an explicit choice: fit() does not override it
        self._step_graph_on = bool(on)
        self._drop_step_graph()
        self._sg_warm = 0

    def _drop_step_graph(self):
        sg = getattr(self, '_sg', None)
        if sg is not None and sg.get('cache_keys'):
            ops.drop_cache_keys(sg['cache_keys'])       # scratch from the graph's private pool (ops.drop_cache_keys)
        self._sg = None

    def __del__(self):
        try:
            self._drop_step_graph()
        except Exception:
            pass

    def _train_step_graphed(self, x_storage, labels):
        x_storage = self._to_internal(x_storage)
        sg = self._sg
        if sg is None or tuple(sg['x'].shape) != tuple(x_storage.shape):
            if self._sg_warm < 2:
                self._sg_warm += 1
                t = self.global_step + 1
                loss_average = self._step_body(self.as_internal(x_storage), labels, self._adam_lr_t(t), self._ema_read(t))
                self.global_step += 1
                return self.training(None, self.learning_rate, self.decay_steps, self.decay_rate, self.momentum), loss_average
            self._drop_step_graph()
            try:
                sg = self._sg = self._capture_step(x_storage, labels)
            except Exception as e:                       # a failed capture: say so and keep training eagerly
                import warnings
                warnings.warn('cgcnn: capturing the training step as a HIP graph failed (%s: %s); the step stays eager'
                              % (type(e).__name__, e))
                self._step_graph_on = False
                self._sg = None
                return self.train_step(self.as_internal(x_storage), labels)
        if sg['x'].data_ptr() != x_storage.data_ptr():      # (a caller that gathered its batch into step_inputs() saves the copy)
            sg['x'].copy_(x_storage)
        if sg['labels'].data_ptr() != labels.data_ptr():
            sg['labels'].copy_(labels)
        t = self.global_step + 1
        from . import _lib                                  # this step's two scalars, one launch
        _lib.check(_lib.lib().chebgcn_set_scalars(ops._p(sg['scal']), float(self._adam_lr_t(t)), float(self._ema_read(t)),
                                                  ops._stream()), 'set_scalars')
        sg['graph'].replay()
        reported_lr = self.training(None, self.learning_rate, self.decay_steps, self.decay_rate, self.momentum)
        self.global_step += 1
        return reported_lr, sg['loss_average'].clone()

    def step_inputs(self):
        """The input buffers of the captured training step, ``(x [B, channel, Mp] in the internal vertex order, labels [B])``, or
        ``(None, None)`` while there is none (eager steps, the first calls before the capture).  A caller that gathers its batch
        straight into them (``_gather(..., out=x)``, ``ops.perm_data(..., out=x)``; hand ``as_internal(x)`` and the labels buffer
        to ``train_step``) saves the two copies ``train_step`` would make."""
        sg = getattr(self, '_sg', None)
        if sg is None or not self._step_graph_on:
            return None, None
        return sg['x'], sg['labels']

    def _capture_step(self, x_storage, labels):
        if ops.timers is not None:
            raise RuntimeError('per-kernel event timers cannot run inside a captured step')
        dev = self.device
        scal = torch.zeros(2, dtype=torch.float32, device=dev)
        sg = {'x': x_storage.detach().clone(), 'labels': labels.detach().clone(), 'scal': scal, 'lr_t': scal[0:1], 'ema_c': scal[1:2]}
        if self._loss_ema is None:
            self._loss_ema = torch.zeros((), dtype=torch.float32, device=dev)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        before = ops.cache_keys()
        cgcnn._captures = getattr(cgcnn, '_captures', 0) + 1
        ops.capture_tag = cgcnn._captures
        # No cyclic garbage collection while the capture is open: a model that became garbage earlier (models are cyclic: bound
        # layer methods) is collected whenever an allocation triggers the collector -- on THIS thread, possibly inside the capture --
        # and its device graphs' finalizers call hipFree, which is not permitted while a stream captures and invalidates it
        # (round 6: reproduced with a discarded twin model; the rare first-capture failures of round 5 were the same thing).
        # torch.cuda.graph() collects once on entry; the collector comes back on when the capture is closed.
        import gc
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            # 'thread_local': only what THIS thread does between begin and end can fail the capture.  The backward pass runs on
            # autograd's device thread; under the default ('global') any potentially-unsafe runtime call of any other thread
            # while the capture is open -- the caching allocator growing a pool from the autograd thread, a collected
            # object releasing an event -- invalidates it, and a failed capture leaves the process unusable (seen as a
            # rare, order-dependent failure of the first capture after the multi-process tests)
            with torch.cuda.graph(graph, capture_error_mode='thread_local'):
                sg['loss_average'] = self._step_body(self.as_internal(sg['x']), sg['labels'], sg['lr_t'], sg['ema_c'][0])
        finally:
            ops.capture_tag = None
            if gc_was_on:
                gc.enable()
        sg['graph'] = graph
        sg['cache_keys'] = ops.cache_keys() - before       # scratch allocated on the capture streams: it dies with this graph
        return sg

    def _apply_adam(self, grad_scale=1.0, lr_t=None, want_sq=False):
        """TF-form Adam over the flat buffers; ``want_sq``: the pass over the regularised variables also leaves the partial sums
        of their squares in ``self._sq_part`` (returns how many)."""
        if self.momentum == 0:
            # tf.train.GradientDescentOptimizer branch (:288-289)
            lr = self.training(None, self.learning_rate, self.decay_steps, self.decay_rate, self.momentum)
            with torch.no_grad():
                g = self._grad * grad_scale
                g[:self._n_reg] += self.regularization * self._flat[:self._n_reg]
                self._flat -= lr * g
            return
        b1, b2 = 0.9, 0.999
        if lr_t is None:
            lr_t = self._adam_lr_t(self.global_step + 1)
        r, n = self._n_reg, self._n_total
        nparts = 0
        if r > 0 and want_sq:
            if getattr(self, '_sq_part', None) is None:
                self._sq_part = torch.zeros(4096, dtype=torch.float32, device=self.device)
            # every variable in ONE launch: the regularised ones (weights: the first r elements of the flat buffer) with the L2 term
            # and the partial sums of their squares, the biases behind them without
            return ops.adam_step_sq_all(self._flat, self._grad, self._adam_m, self._adam_v, r, lr_t, self._sq_part, b1, b2, 1e-8,
                                        grad_scale, self.regularization)
        elif r > 0:
            ops.adam_step(self._flat[:r], self._grad[:r], self._adam_m[:r], self._adam_v[:r], lr_t, b1, b2, 1e-8,
                          grad_scale, self.regularization)
        if n > r:
            ops.adam_step(self._flat[r:], self._grad[r:], self._adam_m[r:], self._adam_v[r:], lr_t, b1, b2, 1e-8,
                          grad_scale, 0.0)
        return nparts

    # ---------------------------------------------------------------- helpers

    def _get_path(self, folder):
        root = os.environ.get('CHEBGCN_HOME', os.getcwd())
        return os.path.join(root, folder, self.dir_name)

    def _ref_view(self, flat, name):
        """The slice of a flat buffer (variables, Adam moments) that belongs to ``name``, in the reference's variable
        shape and vertex order."""
        a, b = self._slices[name]
        spec = self._spec(name)
        return self._ref_tensor(flat[a:b].view(spec.shape), spec)

    def state_dict(self):
        """Checkpoint contents, keyed by the reference's variable names and in its shapes
        (``conv1/weights`` [Fin*K, Fout], ``conv1/bias`` [1, M, F] ...), the Adam moments under
        ``adam_m/<name>`` / ``adam_v/<name>`` (TF: ``<name>/Adam``, ``<name>/Adam_1``), and the step
        counter.  Independent of the internal flat layout."""
        sd = {'global_step': int(self.global_step), 'names': self.variables()}
        arch = getattr(self, '_architecture', None)
        if arch is not None:
            sd['architecture'] = arch()
        for name in self.variables():
            sd[name] = self._ref_view(self._flat, name).detach().cpu().contiguous()
            sd['adam_m/' + name] = self._ref_view(self._adam_m, name).cpu().contiguous()
            sd['adam_v/' + name] = self._ref_view(self._adam_v, name).cpu().contiguous()
        return sd

    def load_state_dict(self, sd):
        """Restore by variable name; shapes are checked.  Optimizer state is optional (a file that
        holds only variables -- e.g. converted from a TF checkpoint -- restores the weights)."""
        missing = [n for n in self.variables() if n not in sd]
        if missing:
            raise KeyError('checkpoint lacks variables %s' % missing)
        with torch.no_grad():
            for name in self.variables():
                for prefix, flat in (('', self._flat), ('adam_m/', self._adam_m), ('adam_v/', self._adam_v)):
                    if prefix + name not in sd:
                        continue
                    spec = self._spec(name)
                    src = torch.as_tensor(np.asarray(sd[prefix + name], np.float32))
                    if tuple(src.shape) != tuple(spec.ref_shape):
                        raise ValueError('checkpoint variable %s%s has shape %s, the model wants %s'
                                         % (prefix, name, tuple(src.shape), tuple(spec.ref_shape)))
                    a, b = self._slices[name]
                    self._ref_assign(flat[a:b].view(spec.shape), spec, src)
        self.global_step = int(sd.get('global_step', 0))

    def _save_best(self, accuracy, step, best, num_to_keep=3):
        """checkmat.BestCheckpointSaver.handle (checkmat.py:43-84, used at models_gcn.py:127, 175):
        keep the ``num_to_keep`` best checkpoints by validation accuracy.  On disk, in
        ``checkpoints/<dir_name>/model``: the JSON index ``best_checkpoints`` mapping
        ``best.ckpt-<step>`` to its value; the text file ``checkpoint`` in tf.train.Saver's format
        (``model_checkpoint_path`` = last one saved, then one ``all_model_checkpoint_paths`` line
        per kept checkpoint in the saver's order: survivors best first, then the new one,
        checkmat.py:70-84) -- the reference's readers parse both (models_gcn.py:89-90, 968-969);
        the weights as a torch file ``best.ckpt-<step>.pt`` holding ``state_dict()`` (variables
        by their reference names and shapes, Adam moments, step, architecture).
        ``best`` is the saver's ordered list of kept checkpoint names (state between calls)."""
        path = os.path.join(self._get_path('checkpoints'), 'model')
        os.makedirs(path, exist_ok=True)
        index = os.path.join(path, 'best_checkpoints')
        current, value = 'best.ckpt-%d' % step, float(accuracy)
        if not os.path.exists(index):
            table = {current: value}
        else:
            with open(index) as f:
                table = json.load(f)
            if len(table) < num_to_keep:
                table[current] = value
            elif not all(v >= value for v in table.values()):
                ranked = sorted(table, key=table.get, reverse=True)          # stable, like the reference's
                worst = ranked.pop(-1)
                for fname in (os.path.join(path, 'checkpoint'), os.path.join(path, worst + '.pt')):
                    if os.path.exists(fname):
                        os.remove(fname)
                best[:] = ranked
                table = {k: table[k] for k in ranked}
                table[current] = value
            else:
                return
        with open(index, 'w') as f:
            json.dump(table, f, indent=3)
        torch.save(self.state_dict(), os.path.join(path, current + '.pt'))
        if current in best:
            best.remove(current)
        best.append(current)
        with open(os.path.join(path, 'checkpoint'), 'w') as f:
            f.write('model_checkpoint_path: "%s"\n' % current)
            for name in best:
                f.write('all_model_checkpoint_paths: "%s"\n' % name)

    def _restore_latest(self):
        """``saver.restore(sess, tf.train.latest_checkpoint(...))`` (:89-90, :326-327): the last
        checkpoint the saver wrote, named by the first line of the ``checkpoint`` file."""
        path = os.path.join(self._get_path('checkpoints'), 'model')
        state = os.path.join(path, 'checkpoint')
        if not os.path.exists(state):
            return
        with open(state) as f:
            latest = f.readline().rstrip('\n').replace('"', '').split(' ')[-1].split('/')[-1]
        self.load_state_dict(torch.load(os.path.join(path, latest + '.pt'), weights_only=True))

    @contextlib.contextmanager
    def variable_scope(self, name):
        self._scope.append(name)
        try:
            yield
        finally:
            self._scope.pop()

    def _var_name(self, leaf):
        return '/'.join(self._scope + [leaf])

    def _weight_initial(self):
        return 'normal' if self.initial == 'normal' else 'he'

    def _get_variable(self, leaf, shape, kind, regularization, group, ref_shape=None, fan_in=None, vaxis=None):
        name = self._var_name(leaf)
        if self._specs is not None:                     # build pass: record and hand out a meta tensor
            if any(s.name == name for s in self._specs):
                raise ValueError('variable %s already exists' % name)
            vkey = None
            if vaxis is not None:
                scope0 = self._scope[0] if self._scope else ''
                vkey = int(scope0[4:]) - 1 if (group != 'head' and scope0.startswith('conv') and scope0[4:].isdigit()) else 'head'
            self._specs.append(_Spec(name, shape, kind, regularization, fan_in, group, ref_shape or shape, vaxis, vkey))
            return torch.empty(tuple(shape), device='meta')
        p = self._params[name]
        if tuple(p.shape) != tuple(shape):
            raise ValueError('variable %s has shape %s, requested %s' % (name, tuple(p.shape), tuple(shape)))
        return p

    def _weight_variable(self, shape, regularization=True):
        """``tf.get_variable('weights', ...)`` in the current scope (:340-347)."""
        group = 'convw' if (self._scope and self._scope[0].startswith('conv')) else 'head'
        # the first FC layer behind the conv stack reads one input per graph vertex: its rows follow the vertex order
        vaxis = 0 if (group == 'head' and getattr(self, '_fc_on_vertices', False)) else None
        self._fc_on_vertices = False
        return self._get_variable('weights', shape, self._weight_initial(), regularization, group,
                                  fan_in=shape[-2] if len(shape) >= 2 else shape[0], vaxis=vaxis)

    def _bias_variable(self, shape, regularization=True):
        """``tf.get_variable('bias', ...)`` initialised to 0.2 (:349-355).  Conv biases are
        stored filter-major ([F] / [F, Mp]) -- ``variable(name)`` gives the TF shape."""
        shape = tuple(int(s) for s in shape)
        if self._scope and self._scope[0].startswith('conv'):
            if regularization:
                raise ValueError('conv biases are not regularised in this layout')
            if len(shape) == 3 and shape[0] == 1 and shape[1] == 1:
                return self._get_variable('bias', (shape[2],), 'const', False, 'convb', ref_shape=shape)
            if len(shape) == 3 and shape[0] == 1:
                return self._get_variable('bias', (shape[2], plane_stride(shape[1])), 'const', False, 'convb',
                                          ref_shape=shape, vaxis=1)
            raise ValueError('conv bias shape %s' % (shape,))
        return self._get_variable('bias', shape, 'const', regularization, 'head')


class cgcnn(base_model):
    """Graph CNN with Chebyshev filters; see the reference's docstring (:405-444) for the
    meaning of F, K, p, M and the training keywords."""

    def __init__(self, config, L, F, K, p, M, filter='chebyshev5', brelu='b1relu', pool='mpool1', initial='normal',
                 channel=1, num_epochs=20, learning_rate=0.1, decay_rate=0.95, decay_steps=None, momentum=0.9,
                 regularization=0, dropout=0, batch_size=100, eval_frequency=200, dir_name='', verbose=True):
        super().__init__(config)
        # consistency checks of :451-460
        if not (len(L) >= len(F) == len(K) == len(p)):
            print(len(L))
            print(len(F), len(K), len(p))
        assert np.all(np.array(p) >= 1)
        p_log2 = np.where(np.array(p) > 1, np.log2(p), 0)
        assert np.all(np.mod(p_log2, 1) == 0)            # powers of 2
        assert len(L) >= np.sum(p_log2)                   # enough coarsening levels
        # one Laplacian per conv layer: the level advances by log2(p) (:462-469)
        M_0 = L[0].shape[0]
        j, self.L = 0, []
        for pp in p:
            self.L.append(L[j])
            j += int(np.log2(pp)) if pp > 1 else 0
        if verbose:
            self._describe(M_0, F, K, p, M, brelu)
        self.F, self.K, self.p, self.M = list(F), list(K), list(p), list(M)
        self.num_epochs, self.learning_rate = num_epochs, learning_rate
        self.decay_rate, self.decay_steps, self.momentum = decay_rate, decay_steps, momentum
        self.regularization, self.dropout = regularization, dropout
        self.batch_size, self.eval_frequency = batch_size, eval_frequency
        self.dir_name = dir_name
        # arithmetic of the contraction AND of its two gradients (ops.ChebConv), not a reference keyword -- set it on the
        # instance: 'auto' (default; per layer, ops.resolve_precision: fp32 matrix instructions up to 32 filters -- every
        # layer of BASELINE configs[1] and of the reference's training.py -- and split bf16, 'bf16x3', for wider layers,
        # where fp32 matrix work would bound the layer; 3e-6 ... 6e-6 of the fp32 result, inside the 1e-5 the north star
        # asks), 'f32' (exact products everywhere), 'bf16', 'bf16x3'.  It is part of the checkpoint's architecture record:
        # a model rebuilt from a checkpoint computes as it was trained
        self.contraction = os.environ.get('CHEBGCN_CONTRACTION', 'auto')
        # last conv layer + tf.reduce_mean(x, -1) (:673) in one kernel where the shape allows (ops.conv_mean_supported);
        # False keeps the two separate (same values up to the order of the sum over the filters)
        self.fuse_feature_mean = os.environ.get('CHEBGCN_FUSE_MEAN', '1') != '0'
        self.filter = getattr(self, filter)
        self.brelu = getattr(self, brelu)
        self.pool = getattr(self, pool)
        self.initial = initial
        self.channel = channel
        self._M0 = int(M_0)
        # Internal vertex order.  The network does not depend on how the vertices of a level are numbered as long as everything
        # per-vertex follows: the model relabels the vertices of every graph level that is large enough by descending number
        # of neighbours (graph.length_order), for which the library has faster recurrence kernels (csrc/recurrence_ord*.hip,
        # planes of more than 1024 vertices, up to 20476 active ones); activations, per-vertex biases and the rows of the first FC layer live in the order of
        # THEIR level, the accessors (variable / set_variable / gradient / state_dict) and the input staging translate.
        # Pooling (models_gcn.py:631-648: p consecutive vertices of the coarsening's tree order) between two levels of which
        # either is relabelled runs through index maps (ops.pool_maps, chebgcn_pool_gather_fwd / _scatter_bwd) instead of
        # through adjacency in memory.  'reference' (or CHEBGCN_VERTEX_ORDER=reference) keeps the caller's numbering everywhere.
        self.vertex_order = os.environ.get('CHEBGCN_VERTEX_ORDER', 'length')
        self.graphs = []
        self._orders = [None] * len(self.L)             # per conv layer: internal position -> reference vertex of its level
        self._pool_maps = [None] * len(self.L)
        self._relabelled = False
        if self.device.type == 'cuda':
            force = self.vertex_order == 'length!'          # experiments: relabel even where no kernel gains from it
            if force:
                self.vertex_order = 'length'
            levels = {}                                     # id(Laplacian) -> (order or None, device graph)
            for Li in self.L:
                if id(Li) in levels:
                    continue
                order = g = None
                # Who gains from sorted rows: the ordered recurrence kernels (big graphs).  The on-chip layer of atlas-sized
                # graphs (csrc/fused_small.hip) would gain 3 % (its waves then gather rows of equal length; captured step at
                # N = 360 0.943 -> 0.911 ms) -- not taken: in the coarsening's tree order spatial neighbours are adjacent and
                # the weight gradients' long cancelling sums over the vertices come out within 2e-7 of float64; in degree
                # order they carry plain fp32 summation noise (1e-4 of their scale, like NumPy's fp32).
                if self.vertex_order == 'length' and self._fusable() and (Li.shape[0] > 1024 or force):
                    # (CHEBGCN_BANK_ORDER=1: the length order refined inside its classes of equal row length against the gather's
                    # LDS bank conflicts, graph.bank_order -- measured in round 6: the conflict cost of the image drops by 16 %, the
                    # kernel by 0-2 %, the configs[1] step not at all (EXPERIMENTS 8.2): off by default)
                    order = (graph_mod.bank_order(Li) if os.environ.get('CHEBGCN_BANK_ORDER', '0') == '1'
                             else graph_mod.length_order(Li))
                    g = ops.Graph(Li, self.device, order=order)
                    if not (g.ordered or force):
                        order = g = None                    # no ordered kernel for this graph size: nothing to gain
                if g is None:
                    g = ops.graph_for(Li, self.device)
                levels[id(Li)] = (order, g)
            self.graphs = [levels[id(Li)][1] for Li in self.L]
            self._orders = [levels[id(Li)][0] for Li in self.L]
            self._relabelled = any(o is not None for o in self._orders)
            if not self._relabelled:
                self.vertex_order = 'reference'
            dev = self.device

            def table(order):
                inv = np.empty_like(order)
                inv[order] = np.arange(len(order))
                return {'order': torch.as_tensor(order).to(dev), 'inv': torch.as_tensor(inv).to(dev)}
            for i, o in enumerate(self._orders):
                if o is not None:
                    self._vtabs[i] = table(o)
            nl = len(self.p)
            for i in range(nl):
                if self.p[i] > 1:
                    src = self._orders[i]
                    dst = self._orders[i + 1] if i + 1 < nl else None      # the head reads the last pooled level in the reference's order
                    if src is not None or dst is not None:
                        self._pool_maps[i] = ops.pool_maps(int(self.p[i]), src, dst, self.L[i].shape[0], dev)
            # the order the LAST conv layer's output is in: what the first FC layer's rows follow
            if self.p[-1] == 1 and self._orders[-1] is not None:
                self._vtabs['head'] = self._vtabs[nl - 1]
            if self._orders[0] is not None:                 # the input level: staging and batches (InternalPlanes)
                order = self._orders[0]
                Mp = plane_stride(self._M0)
                inv = np.empty_like(order)
                inv[order] = np.arange(len(order))
                pad = np.arange(self._M0, Mp)
                self._order = order
                self._order_dev = torch.as_tensor(order.astype(np.int32)).to(self.device)
                self._inv_order_dev = torch.as_tensor(inv).to(self.device)
                self._order_pad = torch.as_tensor(np.concatenate([order, pad])).to(self.device)
                self._inv_order_pad = torch.as_tensor(np.concatenate([inv, pad])).to(self.device)
        self._ctor = dict(L=list(L), F=list(F), K=list(K), p=list(p), M=list(M), filter=filter, brelu=brelu, pool=pool,
                          initial=initial, channel=channel, num_epochs=num_epochs, learning_rate=learning_rate,
                          decay_rate=decay_rate, decay_steps=decay_steps, momentum=momentum,
                          regularization=regularization, dropout=dropout, batch_size=batch_size,
                          eval_frequency=eval_frequency, dir_name=dir_name)
        self.build_graph((M_0, channel))

    @property
    def contraction(self):
        return self._contraction

    @contraction.setter
    def contraction(self, value):
        # (a typo in CHEBGCN_CONTRACTION used to surface as a KeyError deep inside the first backward pass)
        if value != 'auto' and value not in ops.PRECISIONS:
            raise ValueError("contraction must be 'auto' or one of %s, got %r" % (sorted(ops.PRECISIONS), value))
        self._contraction = value

    def layer_precisions(self):
        """The arithmetic each conv layer's contraction (and its two gradients) resolves to under ``contraction``."""
        fins = [self.channel] + list(self.F[:-1])
        return [ops.resolve_precision(self.contraction, fi, k, fo) for fi, k, fo in zip(fins, self.K, self.F)]

    def _architecture(self):
        """What a checkpoint needs to rebuild this model without the script that made it (the
        reference's checkpoints carry a TF meta-graph for the same purpose, :775-776, :979):
        the constructor arguments, the Laplacians as CSR arrays."""
        import scipy.sparse as sp
        arch = {k: v for k, v in self._ctor.items() if k != 'L'}
        arch = {k: ([int(x) if float(x).is_integer() else float(x) for x in v] if isinstance(v, list) else v)
                for k, v in arch.items()}
        arch['contraction'] = self.contraction
        arch['L'] = []
        for Li in self._ctor['L']:
            Li = sp.csr_matrix(Li)
            arch['L'].append({'indptr': torch.as_tensor(Li.indptr.astype(np.int64)),
                              'indices': torch.as_tensor(Li.indices.astype(np.int64)),
                              'data': torch.as_tensor(np.asarray(Li.data)), 'shape': [int(Li.shape[0]), int(Li.shape[1])]})
        return arch

    @classmethod
    def from_checkpoint(cls, sd, config=None, **overrides):
        """Rebuild the model a checkpoint (``state_dict()``) describes and load its variables."""
        import scipy.sparse as sp
        arch = dict(sd['architecture'])
        Ls = [sp.csr_matrix((np.asarray(l['data']), np.asarray(l['indices']), np.asarray(l['indptr'])), shape=tuple(l['shape']))
              for l in arch.pop('L')]
        arch.update(overrides)
        contraction = arch.pop('contraction', 'f32')
        model = cls(config, Ls, verbose=False, **arch)
        model.contraction = contraction
        model.load_state_dict(sd)
        return model

    def _describe(self, M_0, F, K, p, M, brelu):
        L = self.L
        print('NN architecture')
        print('  input: M_0 = {}'.format(M_0))
        for i in range(len(p)):
            print('  layer {0}: cgconv{0}'.format(i + 1))
            print('    representation: M_{0} * F_{1} / p_{1} = {2} * {3} / {4} = {5}'.format(
                i, i + 1, L[i].shape[0], F[i], p[i], L[i].shape[0] * F[i] // p[i]))
            F_last = F[i - 1] if i > 0 else 1
            print('    weights: F_{0} * F_{1} * K_{1} = {2} * {3} * {4} = {5}'.format(
                i, i + 1, F_last, F[i], K[i], F_last * F[i] * K[i]))
            if brelu == 'b1relu':
                print('    biases: F_{} = {}'.format(i + 1, F[i]))
            elif brelu == 'b2relu':
                print('    biases: M_{0} * F_{0} = {1} * {2} = {3}'.format(i + 1, L[i].shape[0], F[i], L[i].shape[0] * F[i]))
        for i in range(len(M)):
            name = 'logits (softmax)' if i == len(M) - 1 else 'fc{}'.format(i + 1)
            print('  layer {}: {}'.format(len(p) + i + 1, name))
            print('    representation: M_{} = {}'.format(len(p) + i + 1, M[i]))
            M_last = M[i - 1] if i > 0 else M_0
            print('    weights: M_{} * M_{} = {} * {} = {}'.format(len(p) + i, len(p) + i + 1, M_last, M[i], M_last * M[i]))
            print('    biases: M_{} = {}'.format(len(p) + i + 1, M[i]))

    # ------------------------------------------------------------------ layers

    def _graph_of(self, L):
        """Device graph for a layer called on its own (logical tensors in the caller's vertex order)."""
        if not self._relabelled:
            for Li, g in zip(self.L, self.graphs):
                if Li is L:
                    return g
        return ops.graph_for(L, self.device)

    def chebyshev5(self, x, L, Fout, K):
        """Chebyshev filtering of x[N, M, Fin] with a [Fin*K, Fout] filter bank (:587-617)."""
        N, M, Fin = x.shape
        W = self._weight_variable([int(Fin) * K, int(Fout)], regularization=True)
        if x.is_meta:
            return torch.empty((N, M, int(Fout)), device='meta')
        y = ops.cheb_conv(ops.plane_storage(x), W, None, self._graph_of(L), K, precision=self.contraction)
        return ops.plane_view(y, M)

    def _brelu(self, x, per_vertex):
        N, M, F = x.shape
        b = self._bias_variable([1, int(M) if per_vertex else 1, int(F)], regularization=False)
        if x.is_meta:
            return x
        kind = BIAS_VERTEX if per_vertex else BIAS_FILTER
        y = ops.BiasReluPool.apply(ops.plane_storage(x), b, int(M), 1, POOL_MAX, True, kind)
        return ops.plane_view(y, M)

    def b1relu(self, x):
        """Bias and ReLU, one bias per filter (:619-623)."""
        return self._brelu(x, False)

    def b2relu(self, x):
        """Bias and ReLU, one bias per vertex per filter (:625-629)."""
        return self._brelu(x, True)

    def _pool(self, x, p, kind):
        if p <= 1:
            return x
        N, M, F = x.shape
        if x.is_meta:
            return torch.empty((N, M // p, F), device='meta')
        y = ops.BiasReluPool.apply(ops.plane_storage(x), None, int(M), int(p), kind, False, BIAS_NONE)
        return ops.plane_view(y, M // p)

    def mpool1(self, x, p):
        """Max pooling of size p over consecutive (tree-ordered) vertices (:631-639)."""
        return self._pool(x, p, POOL_MAX)

    def apool1(self, x, p):
        """Average pooling of size p (:641-648)."""
        return self._pool(x, p, POOL_AVG)

    def fc(self, x, Mout, relu=True):
        """Fully connected layer (:650-656); weights *and* bias are L2-regularised."""
        N, Min = x.shape
        W = self._weight_variable([int(Min), Mout], regularization=True)
        b = self._bias_variable([Mout], regularization=True)
        if x.is_meta:
            return torch.empty((N, Mout), device='meta')
        if self.training_mode and torch.is_grad_enabled() and W.grad is not None and b.grad is not None:
            # training step: the layer writes its gradients straight into the flat gradient buffer (like the
            # conv layers do) -- no accumulate-into-.grad add per variable, no zeroing of the buffer
            dp, names = self._dp, (self._var_name('weights'), self._var_name('bias'))
            return _LinearInto.apply(x, W, b, W.grad, b.grad,
                                     (lambda: dp.head_grads_done(names)) if dp is not None else None, relu)
        if not (torch.is_grad_enabled() and (x.requires_grad or W.requires_grad or b.requires_grad)):
            y = ops.fc_forward(x, W, b, relu) if x.is_cuda else None
            if y is not None:
                return y
        x = torch.addmm(b, x, W)
        return torch.relu(x) if relu else x

    # ------------------------------------------------------------------ network

    def _fusable(self):
        std = lambda m, names: getattr(m, '__func__', None) in [getattr(cgcnn, n) for n in names]
        return (std(self.filter, ['chebyshev5']) and std(self.brelu, ['b1relu', 'b2relu'])
                and std(self.pool, ['mpool1', 'apool1']))

    def _inference(self, x, dropout):
        """Layer loop + head (:658-682) on a logical [N, M, channel] tensor."""
        if x.is_meta or not self._fusable():
            for i in range(len(self.p)):
                with self.variable_scope('conv{}'.format(i + 1)):
                    x = self.filter(x, self.L[i], self.F[i], self.K[i])
                    x = self.brelu(x)
                    x = self.pool(x, self.p[i])
            N, M, F = x.shape
            h = torch.empty((N, M), device='meta') if x.is_meta else ops.FeatureMean.apply(ops.plane_storage(x), int(M))
            return self._head(h, dropout)
        return self._inference_storage(ops.plane_storage(x), dropout)

    def _inference_storage(self, x, dropout):
        """Fused fast path on plane storage ``[N, channel, Mp]``: every conv layer is
        recurrence + (contraction, bias, ReLU, pooling) and writes straight into slab 0 of
        the next layer's Chebyshev stack."""
        if not self._fusable():
            if self._relabelled:
                raise RuntimeError('this model keeps its per-vertex variables in a relabelled vertex order (vertex_order = '
                                   "'length'), which the layer-by-layer path does not know: construct it with "
                                   "CHEBGCN_VERTEX_ORDER=reference to replace filter / brelu / pool methods")
            return self._inference(ops.plane_view(self._to_internal(x), self.L[0].shape[0]), dropout)
        x = self._to_internal(x)
        nl = len(self.p)
        B = x.shape[0]
        per_vertex = getattr(self.brelu, '__func__', None) is cgcnn.b2relu
        pool_kind = POOL_AVG if getattr(self.pool, '__func__', None) is cgcnn.apool1 else POOL_MAX
        stack = None
        # training: the weights of every layer that forms its input gradient by the forward recurrence on dy (ops.dx_by_forward),
        # re-indexed in ONE launch here -- they are constant within the step -- instead of one launch per layer inside backward
        Wts = {}
        if self.training_mode and torch.is_grad_enabled():
            fins = [x.shape[1]] + list(self.F[:-1])
            idxs = [i for i in range(1, nl) if ops.dx_by_forward_shape(self.graphs[i], fins[i], self.K[i], self.F[i], self.contraction)]
            if idxs:
                outs = ops.reindex_weights_batch([self._params['conv%d/weights' % (i + 1)] for i in idxs],
                                                 [(fins[i], self.K[i], self.F[i]) for i in idxs])
                Wts = dict(zip(idxs, outs))
        # layer i's output feeds layer i + 1 and nothing else: the ReluGrad of layer i can run in the epilogue of layer i + 1's
        # input gradient (ops.GateLink; the layers decide by themselves whether their kernels serve it)
        links = [ops.GateLink() for _ in range(nl - 1)] if (self.training_mode and torch.is_grad_enabled()) else None
        for i in range(nl):
            g = self.graphs[i]
            W = self._params['conv%d/weights' % (i + 1)]
            b = self._params['conv%d/bias' % (i + 1)]
            out = next_stack = None
            if i + 1 < nl and g.M // self.p[i] == self.graphs[i + 1].M:
                next_stack = torch.empty((self.K[i + 1], B, self.F[i], self.graphs[i + 1].Mp), dtype=torch.float32,
                                         device=x.device)
                out = next_stack[0]
            # training: the layer's gradients go straight into the flat (zeroed) gradient buffer
            direct = self.training_mode and W.grad is not None and b.grad is not None and torch.is_grad_enabled()
            done = (lambda layer=i + 1: self._dp.layer_done(layer)) if (direct and self._dp is not None) else None
            # the last layer feeds tf.reduce_mean(x, -1) (:673) only: where the kernel can, it returns that mean and never
            # stores its own output; its gradients read one plane per window
            mean = bool(i + 1 == nl and self.fuse_feature_mean and
                        ops.conv_mean_supported(B, g.M, x.shape[1], self.K[i], self.F[i], self.p[i], True, self.contraction))
            x = ops.cheb_conv(x, W, b, g, self.K[i], self.p[i], pool_kind, True,
                              BIAS_VERTEX if per_vertex else BIAS_FILTER, stack=stack, out=out,
                              dW=W.grad if direct else None, dbias=b.grad if direct else None,
                              precision=self.contraction, done=done, mean=mean, pool_maps=self._pool_maps[i], Wt=Wts.get(i),
                              link_in=links[i - 1] if (links and i > 0) else None,
                              link_out=links[i] if (links and i + 1 < nl) else None)
            stack = next_stack
        M_last = self.graphs[-1].M // self.p[-1]
        if mean:
            return self._head(x, dropout)              # already the logical [B, M] view of the [B, Mp] means
        return self._head(ops.FeatureMean.apply(x, M_last), dropout)

    def _head(self, x, dropout):
        """reduce_mean output -> FC stack with dropout -> logits (:674-682)."""
        self._fc_on_vertices = True                    # (build pass: the next weight variable has one row per vertex)
        for i, Mi in enumerate(self.M[:-1]):
            with self.variable_scope('fc{}'.format(i + 1)):
                x = self.fc(x, Mi)
                if not x.is_meta and dropout != 1:
                    x = Fnn.dropout(x, p=1.0 - float(dropout), training=True)   # tf.nn.dropout(x, keep_prob), :677
        with self.variable_scope('logits'):
            x = self.fc(x, self.M[-1], relu=False)
        return x


class _LinearInto(torch.autograd.Function):
    """``x @ W + b`` whose backward WRITES dW and db into the given buffers (views of the model's flat
    gradient buffer: one use of a variable per step) instead of returning them to autograd, then reports
    through ``done`` (dist.DataParallel.head_grads_done: the head's all-reduce starts from it)."""

    @staticmethod
    def forward(ctx, x, W, b, gW, gb, done, relu):
        y = ops.fc_forward(x, W.detach(), b.detach(), relu) if x.is_cuda else None
        if y is None:
            y = torch.addmm(b.detach(), x, W.detach())
            if relu:
                y = torch.relu_(y)
        ctx.save_for_backward(x, W, y if relu else None)
        ctx.bufs = (gW, gb, done)
        return y

    @staticmethod
    def backward(ctx, g):
        x, W, y = ctx.saved_tensors
        gW, gb, done = ctx.bufs
        g = g.contiguous()
        if g.is_cuda:
            r = ops.fc_backward(x, W, g, y, gW, gb, ctx.needs_input_grad[0])
            if r is not None:
                if done is not None:
                    done()
                return r[0], None, None, None, None, None, None
        if y is not None:
            g = torch.ops.aten.threshold_backward(g, y, 0.0)
        torch.mm(x.t(), g, out=gW)
        torch.sum(g, 0, out=gb)
        dx = g @ W.t() if ctx.needs_input_grad[0] else None
        if done is not None:
            done()
        return dx, None, None, None, None, None, None


def get_best_checkpoint(best_checkpoint_dir, select_maximum_value=True):
    """Path of the best checkpoint according to the ``best_checkpoints`` index
    (checkmat.get_best_checkpoint, checkmat.py:121-138); the weights are in ``<path>.pt``."""
    index = os.path.join(best_checkpoint_dir, 'best_checkpoints')
    assert os.path.exists(index)
    with open(index) as f:
        best = json.load(f)
    names = sorted(best, key=best.get, reverse=select_maximum_value)
    return os.path.join(best_checkpoint_dir, names[0])


class model_perf(object):
    """Experiment harness: ``test`` = fit + evaluate on train and test (:936-958)."""

    def __init__(s):
        s.names, s.params = set(), {}
        s.fit_accuracies, s.fit_losses, s.fit_time = {}, {}, {}
        s.train_accuracy, s.train_f1, s.train_loss = {}, {}, {}
        s.test_accuracy, s.test_f1, s.test_loss = {}, {}, {}

    def test(s, model, name, params, train_data, train_labels, val_data, val_labels, test_data, test_labels,
             target_name=None):
        s.params[name] = params
        s.fit_accuracies[name], s.fit_losses[name], s.fit_time[name] = model.fit(
            train_data, train_labels, val_data, val_labels)
        string, s.train_accuracy[name], s.train_f1[name], s.train_loss[name] = model.evaluate(
            train_data, train_labels, target_name=target_name)
        print('\ntrain {}\n'.format(string))
        string, s.test_accuracy[name], s.test_f1[name], s.test_loss[name] = model.evaluate(
            test_data, test_labels, target_name=target_name)
        print('\ntest  {}\n'.format(string))
        s.names.add(name)
        return s

    def predict(s, ckp_path, test_data, test_labels, target_name=None, batch_size=128, trial_dura=17,
                flag_starttr=False, sub_name=None, model=None, config=None):
        """Restore a trained model from ``<ckp_path>/model/`` and score a dataset (:960-1088; call
        site predict_states.py:102-108).  Like the reference it takes the checkpoint named on line 1
        of the saver's ``checkpoint`` file (:968-969), pads the last batch with zeros, SUMS the batch
        losses, and flattens the stacked logits before truncating them (:1022-1023).  The model is
        rebuilt from the checkpoint's architecture record (the reference imports the TF meta-graph,
        :979), or ``model`` is used.  Returns (pred_logits, pred_labels, pred_loss, test_acc)."""
        import sklearn.metrics
        ckp_path = str(ckp_path) + '/' + 'model/'
        lines = [line.rstrip('\n') for line in open(os.path.join(ckp_path, 'checkpoint'))]
        model_name = lines[1].replace('"', '').split(' ')[-1].split('/')[-1]
        print(ckp_path + model_name + '.pt')
        sd = torch.load(ckp_path + model_name + '.pt', weights_only=True)
        if model is None:
            model = cgcnn.from_checkpoint(sd, config=config, batch_size=batch_size)
        else:
            model.load_state_dict(sd)
        test_labels = np.asarray(test_labels)
        data_dev = model.stage(test_data)
        data_size = data_dev.shape[0]
        pred_logits, pred_labels, pred_loss = [], [], 0
        model.training_mode = False
        for begin in range(0, data_size, batch_size):
            end = min([begin + batch_size, data_size])
            idx = torch.arange(begin, end, dtype=torch.int32, device=model.device)
            x = model._gather(data_dev, idx)
            if end - begin < batch_size:
                pad = ops.plane_empty(batch_size, x.shape[1], data_dev.shape[1], model.device, zero=True)
                pad[:end - begin] = x.planes
                x = model.as_internal(pad)
            batch_labels = np.zeros(batch_size, np.int64)
            batch_labels[:end - begin] = test_labels[begin:end]
            with torch.no_grad():
                logits = model._inference_storage(x, 1)
                loss = model.loss(logits, torch.as_tensor(batch_labels).to(model.device), model.regularization)[0]
            pred_logits.append(logits.cpu().numpy())
            pred_labels.append(model.prediction(logits).cpu().numpy())
            pred_loss += float(loss)
        pred_labels = np.stack(pred_labels, axis=0).flatten()[:len(test_labels)]
        pred_logits = np.stack(pred_logits, axis=0).flatten()[:len(test_labels)]
        if target_name is not None:
            print(sklearn.metrics.classification_report(test_labels, pred_labels, labels=range(len(target_name)),
                                                        target_names=target_name))
            print('Confusion Matrix:')
            print(sklearn.metrics.confusion_matrix(test_labels, pred_labels, labels=range(len(target_name))))
        test_acc = []
        ncorrects = int(sum(pred_labels == test_labels))
        accuracy = 100 * sklearn.metrics.accuracy_score(test_labels, pred_labels)
        f1 = 100 * sklearn.metrics.f1_score(test_labels, pred_labels, average='weighted')
        print('accuracy: {:.2f} ({:d} / {:d}), f1 (weighted): {:.2f}, loss: {:.2e}'.format(
            accuracy, ncorrects, len(test_labels), f1, pred_loss))
        test_acc.append(accuracy)
        if sub_name is not None:
            # per-subject weighted F1 per condition and overall (:1040-1064), written like the reference
            import pandas as pd
            try:
                y_pred = np.array(np.split(pred_labels, len(sub_name)))
                y_label = np.array(np.split(test_labels, len(sub_name)))
            except ValueError:
                sub_used = pred_labels.shape[0] // len(sub_name) * len(sub_name)
                y_pred = np.array(np.split(pred_labels[:sub_used], len(sub_name)))
                y_label = np.array(np.split(test_labels[:sub_used], len(sub_name)))
            test_acc = np.zeros((len(sub_name), len(target_name) + 1))
            for subi in range(len(sub_name)):
                for li in range(len(target_name)):
                    mask = y_label[subi, :] == li
                    test_acc[subi, li] = sklearn.metrics.f1_score(y_label[subi, mask], y_pred[subi, mask], average='weighted')
                test_acc[subi, -1] = sklearn.metrics.f1_score(y_label[subi, :], y_pred[subi, :], average='weighted')
            result_df = pd.DataFrame()
            result_df['subject'] = sub_name
            for li, task in enumerate(target_name):
                result_df[task] = test_acc[:, li]
            result_df['avg'] = test_acc[:, -1]
            os.makedirs('train_logs', exist_ok=True)
            result_df.to_csv('train_logs/' + target_name[0].split('_')[-1] + '_f1score_testacc_' + str(len(sub_name)) +
                             'subjects.csv', sep='\t', encoding='utf-8', index=False)
        if flag_starttr:
            # accuracy as a function of the window's position inside its trial (:1066-1087)
            y_pred = np.reshape(pred_labels, (-1, trial_dura))
            y_label = np.reshape(test_labels, (-1, trial_dura))
            test_acc = np.zeros((len(target_name), trial_dura))
            for li in range(len(target_name)):
                print('\n', target_name[li], ':')
                for ti in range(trial_dura):
                    mask = y_label[:, ti] == li
                    nc = int(sum(y_pred[mask, ti] == y_label[mask, ti]))
                    acc = 100 * sklearn.metrics.accuracy_score(y_label[mask, ti], y_pred[mask, ti])
                    f1 = 100 * sklearn.metrics.f1_score(y_label[mask, ti], y_pred[mask, ti], average='weighted')
                    print('start_tr {:d} accuracy: {:.2f} ({:d} / {:d}), f1 (weighted): {:.2f}'.format(ti, acc, nc, int(np.sum(mask)), f1))
                    test_acc[li, ti] = acc
            print('\ntotal:')
            for ti in range(trial_dura):
                nc = int(sum(y_pred[:, ti] == y_label[:, ti]))
                acc = 100 * sklearn.metrics.accuracy_score(y_label[:, ti], y_pred[:, ti])
                f1 = 100 * sklearn.metrics.f1_score(y_label[:, ti], y_pred[:, ti], average='weighted')
                print('start_tr {:d} accuracy: {:.2f} ({:d} / {:d}), f1 (weighted): {:.2f}'.format(ti, acc, nc, len(y_pred), f1))
        return pred_logits, pred_labels, pred_loss, test_acc
