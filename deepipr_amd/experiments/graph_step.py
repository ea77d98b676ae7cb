"""hipGraph capture of a train step.

At the reference's small per-GPU batches (config P: 32 images per GPU, two forwards per step) the step is
host-bound: several hundred kernel launches of a few microseconds each.  Capturing zero_grad -> forward -> losses ->
backward (-> SGD on one GPU) once and replaying it removes the launch overhead; the passport kernels are capture-safe
by contract (enqueue-only C ABI, caller-provided workspaces, no host reads).  Measured on one MI355X:
config-P shard 10.7 ms eager -> 5.0 ms replayed; config R (128 images) 5.8 -> 5.5 ms.  bench.py replays by default;
the trainers do with several GPUs (--graph on one).

Constraints while a graphed step is in use: static batch shape; passport keys must not change (the pooled
key means are cached outside the graph); in-situ kernel timing (deepipr_profile_enable) must stay off.
"""
import torch


class _NoStep:
    """Optimizer proxy handed to step_fn while capturing in 'fwd_bwd' mode: zero_grad is recorded, step() is not."""

    def __init__(self, optimizer):
        self._opt = optimizer

    def zero_grad(self, *a, **k):
        return self._opt.zero_grad(*a, **k)

    def step(self, *a, **k):
        return None


class GraphedTrainStep:
    """step_fn(model, optimizer, data, target) -> tuple of device scalars, captured once and replayed.

    optimizer_in_graph=True  (single GPU): the whole step, optimiser included, is one hipGraph.
    optimizer_in_graph=False (data parallel): zero_grad -> forward -> losses -> backward are captured; after each
        replay optimizer.step() runs eagerly -- with FlatSGD that is: pack the (static-address) gradients bucket by
        bucket, all-reduce them over RCCL, one fused SGD kernel.  No collective is ever captured, so the graph
        contains only this process's own kernels.

    Learning-rate schedules.  Launch arguments are frozen at capture time, so a captured optimiser would keep the
    learning rate it was captured with while MultiStepLR (lr_configs/*.json: decay at epochs 100 and 150) moves
    param_groups on.  Two mechanisms keep a replayed step on the schedule:
      * FlatSGD reads {lr, momentum, weight decay, 1/world} from device memory; `sync_hyper()` refreshes those four
        floats before a replay whenever param_groups changed -- no re-capture;
      * any other optimiser: the hyper-parameters of every param group are compared with the captured ones before
        each replay and the step is re-captured when they differ (a few times per training run)."""

    def __init__(self, step_fn, model, optimizer, data, target, warmup=3, optimizer_in_graph=True):
        self.static_data = data.clone()
        self.static_target = target.clone()
        self.step_fn, self.model = step_fn, model
        self.optimizer = optimizer
        self.optimizer_in_graph = optimizer_in_graph
        self.recaptures = 0
        # warm-up and capture share ONE side stream: everything the kernels keep per (device, stream) -- scratch
        # arena, in-launch exchange words -- is created by the eager warm-up, never inside the capture (a captured
        # zero-fill of the exchange words would be replayed every step and erase their time-out flag)
        self.stream = torch.cuda.Stream()
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):               # warm-up off the caller's stream: MIOpen find,
            for _ in range(warmup):                        # allocator pools, lazily created optimizer state
                step_fn(model, optimizer, self.static_data, self.static_target)
            if warmup == 0:
                # training must not advance, but MIOpen still has to pick its algorithms and SGD has to create
                # its momentum buffers outside the capture: run one step on throw-away copies of the state
                self._dry_run(step_fn, model, optimizer)
        torch.cuda.current_stream().wait_stream(self.stream)
        self._capture()

    def _hyper_signature(self):
        return tuple(tuple((k, v) for k, v in sorted(g.items()) if k != 'params' and isinstance(v, (int, float, bool)))
                     for g in self.optimizer.param_groups)

    def _capture(self):
        import contextlib

        from deepipr_amd import passport_ops
        optimizer = self.optimizer
        kernels = passport_ops.kernels
        optimizer.zero_grad(set_to_none=True)
        # Data-parallel replay with FlatSGD, un-staged form (experiments/staged.py is the default with several GPUs and
        # overlaps the exchange): every collective of a step is enqueued by optimizer.step() AFTER the replay and is
        # waited for before the SGD kernel, i.e. before the next replay starts -- no RCCL kernel ever shares the device
        # with the captured kernels, so the split-channel single-pass kernels (which need all their workgroups
        # co-resident) may stay on inside the graph.  sync_scope never overrides the USER's switch
        # (DEEPIPR_ALLOW_SYNC=0 / kernels.set_user_sync(False)): after an exchange time-out the step is captured again
        # without them.
        exchange_after_replay = (not self.optimizer_in_graph) and hasattr(optimizer, 'set_mode')
        scope = contextlib.nullcontext()
        if exchange_after_replay:
            optimizer.set_mode('single')               # rank-invariant: ONE all-reduce per step, no hooks
            scope = kernels.sync_scope(True)
        if hasattr(optimizer, 'prepare_in_place_capture'):
            optimizer.prepare_in_place_capture()       # pinned chunk table of THIS capture, allocated outside it
        with scope:
            kernels.prepare_stream(self.static_data.device, self.stream)
            self._device_hyper = hasattr(optimizer, 'sync_hyper')
            if self._device_hyper:
                optimizer.sync_hyper()
            self._captured_hyper = self._hyper_signature()
            self.graph = torch.cuda.CUDAGraph()
            captured_opt = optimizer if self.optimizer_in_graph else _NoStep(optimizer)
            # with a process group alive, RCCL's watchdog thread polls events concurrently: only this thread's calls
            # may be checked against the capture ("thread_local"), otherwise its hipEventQuery aborts the capture
            mode = 'global' if self.optimizer_in_graph else 'thread_local'
            from deepipr_amd.distributed import retire_collectives
            retire_collectives()                   # ... and none of its polls may fall inside the capture at all
            with torch.cuda.graph(self.graph, stream=self.stream, capture_error_mode=mode):
                self.outputs = self.step_fn(self.model, captured_opt, self.static_data, self.static_target)
        # the gradient tensors the replayed backward writes (graph-pool memory): `.grad` must point at them whenever the
        # optimiser runs OUTSIDE the graph, also after an eager step in between (a ragged last batch) re-bound it
        self._captured_grads = [(p, p.grad) for g in optimizer.param_groups for p in g['params']]

    def _dry_run(self, step_fn, model, optimizer):
        from deepipr_amd.experiments.staged import dry_run
        dry_run(lambda: step_fn(model, optimizer, self.static_data, self.static_target), model, optimizer)

    def __call__(self, data, target):
        if self._device_hyper:
            self.optimizer.sync_hyper()                     # four floats, only when param_groups changed
        elif self.optimizer_in_graph and self._hyper_signature() != self._captured_hyper:
            self.recaptures += 1                            # e.g. MultiStepLR crossed a milestone
            self._capture()
        self.static_data.copy_(data, non_blocking=True)
        self.static_target.copy_(target, non_blocking=True)
        self.graph.replay()
        if not self.optimizer_in_graph:
            for p, g in self._captured_grads:
                p.grad = g
            self.optimizer.step()
        return self.outputs


class GraphedEval:
    """One batch of the test loops (trainer.py:186-204, trainer_private.py:220-241) -- eval-mode forward, summed cross
    entropy, top-1 class, number of hits -- captured as one hipGraph per batch shape and replayed.

    Why: the eager forward is host-bound (≈ 60 fused nodes through ctypes + Python autograd bookkeeping: 2-3 ms of host time
    for ≈ 1 ms of kernels at batch 128), and the test set runs through it every epoch.  `forward(data)` is the model call
    (`lambda d: model(d, ind=1)` for the private branch).  A graph is keyed on the batch shape (the ragged last batch gets
    its own; at most `max_graphs`, then eager) and stays valid while the things whose ADDRESSES it froze stay put: the
    parameters / buffers (updated in place by training) and every passport layer's pooled key means (passport_ops.PooledKeys:
    a key that was replaced or stepped on yields a new tensor -- seen here before every replay, the shape is captured again).
    Returns (top-1 class [N, 1], summed loss, hits): the graph's own output tensors, overwritten by the next call."""

    def __init__(self, model, forward=None, max_graphs=4):
        self.model = model
        self.forward = forward if forward is not None else model
        self.max_graphs = max_graphs
        self._graphs = {}
        self.captures = 0
        self.stream = None

    def _signature(self):
        from deepipr_amd.models.layers._passport_base import PassportLayerBase
        sig = []
        for m in self.model.modules():
            if isinstance(m, PassportLayerBase) and m.get_bias_key() is not None and m.get_scale_key() is not None:
                sig.append(id(m._pooled_means()[2]))
        sig.extend(t.data_ptr() for t in self.model.parameters())
        sig.extend(t.data_ptr() for t in self.model.buffers())
        return tuple(sig)

    @staticmethod
    def batch(forward, data, target):
        import torch.nn.functional as F
        pred = forward(data)
        loss = F.cross_entropy(pred, target, reduction='sum')
        top = pred.max(1, keepdim=True)[1]
        return top, loss, top.eq(target.view_as(top)).sum()

    def _capture(self, data, target):
        from deepipr_amd import passport_ops
        from deepipr_amd.distributed import retire_collectives
        if self.stream is None:
            self.stream = torch.cuda.Stream()
        static_data, static_target = data.clone(), target.clone()
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):               # library algorithm search, scratch arenas: outside the capture
            self.batch(self.forward, static_data, static_target)
        torch.cuda.current_stream().wait_stream(self.stream)
        passport_ops.kernels.prepare_stream(static_data.device, self.stream)
        graph = torch.cuda.CUDAGraph()
        distributed = torch.distributed.is_available() and torch.distributed.is_initialized()
        retire_collectives()
        with torch.cuda.graph(graph, stream=self.stream, capture_error_mode='thread_local' if distributed else 'global'):
            out = self.batch(self.forward, static_data, static_target)
        self.captures += 1
        return [graph, static_data, static_target, out, self._signature()]

    def __call__(self, data, target):
        if not data.is_cuda or self.model.training or torch.is_grad_enabled():
            return self.batch(self.forward, data, target)
        key = (tuple(data.shape), tuple(target.shape), data.dtype)
        entry = self._graphs.get(key)
        if entry is not None and entry[4] != self._signature():
            entry = None                                    # a key or a parameter moved: the frozen addresses are stale
        if entry is None:
            if key not in self._graphs and len(self._graphs) >= self.max_graphs:
                return self.batch(self.forward, data, target)
            entry = self._graphs[key] = self._capture(data, target)
        graph, static_data, static_target, out, _sig = entry
        static_data.copy_(data, non_blocking=True)
        static_target.copy_(target, non_blocking=True)
        graph.replay()
        return out
