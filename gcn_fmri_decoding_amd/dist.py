"""Data parallelism for ``cgcnn``: one process per GPU, batch sharded across ranks, gradients
averaged with RCCL (``torch.distributed`` backend "nccl" on ROCm) over xGMI.

The reference is single-process; this is new.  Every hot-path op is
independent per window, so each rank runs the whole network on its shard and the only
exchange is the gradient all-reduce.  All parameters live in one flat fp32 buffer laid out
``[head | conv weights | conv biases]`` (models_gcn.build_graph), conv variables in layer
order inside their regions.  Backward runs head -> conv_n -> ... -> conv_1, so gradients
are reduced in that order, in few large messages (xGMI is point-to-point, 7 links per GPU:
a ring is per-link bound and wants big messages):

* bucket 0 = the FC head (71 % of the bytes at the benchmark config).  Its all-reduce is
  issued from a post-accumulate hook as soon as the last head gradient lands and overlaps
  the whole convolutional backward on RCCL's own stream.
  (``cgcnn.enable_step_graph`` captures the step WITH its collectives on RCCL: the graph replays them like kernels.)
* conv buckets = groups of consecutive layers, last layers first (default: two groups).  The
  conv layers write their gradients straight into the flat buffer (ops.ChebConv, no autograd
  hook fires), so ``ChebConv.backward`` reports a finished layer through ``layer_done``; when
  the first layer of a group is done, the group's weight and bias slices (one contiguous range
  each) go out while the earlier layers still run.  Only the last group is exposed.
"""
import torch
import torch.distributed as dist


class DataParallel:
    def __init__(self, model, process_group=None, conv_groups=2):
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        self.model = model
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self._head_names = [s.name for s in model._spec_list if s.group == 'head']
        self._pending = set()       # head variables whose gradient is not enqueued yet
        self._head_sent = True
        self._work = []
        self._hooks = []
        self._stream_ordered = dist.get_backend(process_group) == 'nccl'
        # the step may be captured as a HIP graph with its collectives inside (cgcnn.enable_step_graph): RCCL's all-reduce
        # is enqueued on a stream like a kernel; other backends run on the host
        self.capturable = self._stream_ordered
        # bench.py --gpus N: HIP events around the tail of the step that waits for the collectives (what of the all-reduce
        # is NOT hidden behind backward); None = off
        self.exposed_events = None
        for name in self._head_names:
            p = model._params[name]
            self._hooks.append(p.register_post_accumulate_grad_hook(lambda _p, name=name: self.head_grads_done((name,))))
        self._plan_conv_buckets(conv_groups)
        model._dp = self
        self.broadcast_parameters()

    # -------------------------------------------------------------------------- buckets
    def _plan_conv_buckets(self, conv_groups):
        """Split the conv layers into ``conv_groups`` runs of consecutive layers; a run is sent
        when its FIRST layer (the last one backward reaches) reports done.  Each run is one
        contiguous slice of the conv-weight region and one of the conv-bias region."""
        m = self.model
        slices = getattr(m, '_slices', None)
        layers = sorted({int(s.name.split('/')[0][4:]) for s in m._spec_list if s.group in ('convw', 'convb')})
        self._buckets = []          # (trigger layer, [(a, b), ...]) in sending order
        self._sent = set()
        if not layers or slices is None:
            return
        n = max(1, min(int(conv_groups), len(layers)))
        per = -(-len(layers) // n)
        runs = [layers[i:i + per] for i in range(0, len(layers), per)]
        for run in reversed(runs):
            ranges = []
            for leaf in ('weights', 'bias'):
                names = ['conv%d/%s' % (i, leaf) for i in run if 'conv%d/%s' % (i, leaf) in slices]
                if names:
                    a, b = min(slices[k][0] for k in names), max(slices[k][1] for k in names)
                    if b - a != sum(slices[k][1] - slices[k][0] for k in names):
                        raise AssertionError('conv variables of layers %s are not contiguous in the flat buffer' % run)
                    ranges.append((a, b))
            self._buckets.append((run[0], ranges))

    def broadcast_parameters(self, src=0):
        """Start every rank from rank ``src``'s variables and optimizer state."""
        m = self.model
        for buf in (m._flat, m._adam_m, m._adam_v):
            dist.broadcast(buf, src, group=self.group)

    def barrier(self):
        dist.barrier(group=self.group)

    def check_equal(self, value, what):
        """Raise on every rank unless ``value`` (an int) is the same on all of them -- e.g. the shard size that
        fixes the number of steps (and so of collectives) a rank runs in ``fit``."""
        dev = self.model._flat.device
        t = torch.tensor([int(value), -int(value)], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        lo, hi = -int(t[1]), int(t[0])
        if lo != hi:
            raise ValueError('%s differs between ranks (min %d, max %d, rank %d has %d): every rank must run the same '
                             'number of steps' % (what, lo, hi, self.rank, int(value)))

    # called by cgcnn.train_step -----------------------------------------------------
    def begin_step(self):
        self._pending = set(self._head_names)
        self._head_sent = False
        self._work = []
        self._sent = set()

    def _reduce(self, a, b):
        g = self.model._grad[a:b]
        if g.is_cuda and not self._stream_ordered:
            # RCCL orders the collective behind the kernels already enqueued on the current stream.  Other backends
            # (gloo, used by the tests that put two ranks on one GPU) were seen reading the slice before the kernels
            # that write it had run: wait for them.
            torch.cuda.current_stream(g.device).synchronize()
        self._work.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def head_grads_done(self, names):
        """The head variables ``names`` have their gradients enqueued.  Reported by autograd's post-accumulate
        hook and by the FC layers of the training step, which write dW / db into the flat buffer themselves
        (both may report the same variable: the set makes that harmless).  The head goes out once, when the
        last one is in."""
        self._pending.difference_update(names)
        if not self._pending and not self._head_sent:
            self._head_sent = True
            self._reduce(0, self.model._n_head)

    def layer_done(self, layer):
        """Conv layer ``layer`` (1-based) has enqueued its gradient kernels (ops.ChebConv.backward)."""
        for i, (trigger, ranges) in enumerate(self._buckets):
            if trigger == layer and i not in self._sent:
                self._sent.add(i)
                for a, b in ranges:
                    self._reduce(a, b)

    def finish_step(self):
        """Reduce what is left, wait, and return the scale that turns sums into means."""
        m = self.model
        if not self._head_sent:         # some head variable never reported (e.g. frozen head): reduce it now
            self._head_sent = True
            self._reduce(0, m._n_head)
            self._pending = set()
        if self._buckets:
            for i, (_, ranges) in enumerate(self._buckets):
                if i not in self._sent:
                    self._sent.add(i)
                    for a, b in ranges:
                        self._reduce(a, b)
        elif m._n_total > m._n_head:
            self._reduce(m._n_head, m._n_total)
        ev = None
        if self.exposed_events is not None and m._grad.is_cuda and not torch.cuda.is_current_stream_capturing():
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()                          # behind the last backward kernel of this rank
        for w in self._work:
            w.wait()
        if ev is not None:
            ev[1].record()                          # the stream may go on (Adam): every bucket has arrived
            self.exposed_events.append(ev)
        self._work = []
        return 1.0 / self.world

    def remove(self):
        for h in self._hooks:
            h.remove()
        self.model._dp = None
