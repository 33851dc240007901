"""Data parallelism for ``cgcnn``: one process per GPU, batch sharded across ranks, gradients
averaged with RCCL (``torch.distributed`` backend "nccl" on ROCm) over xGMI.

The reference is single-process; this is new.  Every hot-path op is
independent per window, so each rank runs the whole network on its shard and the only
exchange is the gradient all-reduce.  All parameters live in one flat fp32 buffer laid out
``[head | conv weights | conv biases]`` (models_gcn.build_graph):

* bucket 0 = the FC head (71 % of the bytes at the benchmark config; its gradients are
  complete first because backward runs head -> conv6 -> ... -> conv1).  Its all-reduce is
  issued from a post-accumulate hook as soon as the last head gradient lands and overlaps
  the whole convolutional backward on RCCL's own stream.
* bucket 1 = conv weights + biases, reduced after backward.

Two large collectives instead of many small ones: xGMI is point-to-point (7 links per GPU),
so a ring is per-link bound and wants big messages.
"""
import torch
import torch.distributed as dist


class DataParallel:
    def __init__(self, model, process_group=None):
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        self.model = model
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self._head_names = [s.name for s in model._spec_list if s.group == 'head']
        self._pending = 0
        self._work = []
        self._hooks = []
        for name in self._head_names:
            p = model._params[name]
            self._hooks.append(p.register_post_accumulate_grad_hook(self._on_head_grad))
        model._dp = self
        self.broadcast_parameters()

    def broadcast_parameters(self, src=0):
        """Start every rank from rank ``src``'s variables and optimizer state."""
        m = self.model
        for buf in (m._flat, m._adam_m, m._adam_v):
            dist.broadcast(buf, src, group=self.group)

    # called by cgcnn.train_step -----------------------------------------------------
    def begin_step(self):
        self._pending = len(self._head_names)
        self._work = []

    def _on_head_grad(self, _param):
        self._pending -= 1
        if self._pending == 0:
            g = self.model._grad[:self.model._n_head]
            self._work.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish_step(self):
        """Reduce what is left, wait, and return the scale that turns sums into means."""
        m = self.model
        if self._pending != 0:          # head hook did not fire (e.g. frozen head): reduce it now
            self._work.append(dist.all_reduce(m._grad[:m._n_head], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        if m._n_total > m._n_head:
            self._work.append(dist.all_reduce(m._grad[m._n_head:], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in self._work:
            w.wait()
        self._work = []
        return 1.0 / self.world

    def remove(self):
        for h in self._hooks:
            h.remove()
        self.model._dp = None
