"""Staged backward: the data-parallel train step with the gradient exchange overlapped, also under hipGraph replay.

The reference's multi-GPU story is one process driving nn.DataParallel (experiments/trainer.py:92-93,
trainer_private.py:110-111): gradients are gathered onto GPU 0 after backward, nothing overlaps.  Here every rank
back-propagates its own shard and the gradients travel as a few large RCCL all-reduces in gradient-ready order
(flat_sgd.py).  Eager dispatch can launch a bucket from a gradient hook while earlier layers still back-propagate,
but an eager step is host-bound at the reference's per-GPU batches; a step replayed from ONE hipGraph is
not, but no hook can fire inside it, so round 2 exchanged everything after the replay, un-hidden.  This module does
both: the backward pass is cut into a few STAGES at activations the model names (deepipr_amd/cuts.py,
model.backward_stages()) and

    stage 0   zero_grad, forward, losses, backward down to the first cut
    stage k   backward from cut k-1 down to cut k (torch.autograd.grad on the detached cut leaves; the objective stays
              a root so that sign losses of layers in later stages keep their gradient)
    between   bucket k (= the parameters whose gradients stage k finished) is packed and all-reduced on a SIDE stream
              while the main stream goes on with stage k + 1
    end       remaining buckets as one message, wait, ONE fused SGD kernel (FlatSGD.step)

Replayed form.  The stages are captured back to back into ONE hipGraph; behind every stage whose bucket is to travel
under the rest of backward the capture packs the bucket into the flat gradient buffer and records an EXTERNAL event
(deepipr_event_record: an event-record node in the captured graph).  A step is then: launch the graph; for every such
bucket wait ON THE HOST until the graph has passed the event, then enqueue the all-reduce on a side stream;
FlatSGD.step (remaining buckets, one fused SGD kernel).  The graph is not split and no collective is ever captured.
What this costs was measured on one MI355X with the exchange forced on in a world of one
(profiles/r03_ddp_rehearsal*.jsonl, r03_probe2*.json, r03_staged_probe_*.json):
  * the cuts, torch.autograd.grad per stage and the event-record nodes cost nothing at replay (5.468 ms against
    5.469 ms per step);
  * splitting the step into several hipGraphs instead cost 30-140 us per extra graph;
  * a STREAM-side wait for the event, enqueued right after the launch, sits in a second hardware queue for half a step
    and slows every dispatch of the replayed graph next to it: +160 us per step for ResNet18 V1 (batch 128), +560 us for
    the V2 shard (batch 32, twice the dispatches) -- whether one bucket or three.  Holding the HOST until the event has
    fired and enqueuing the collective then removes that: +0.5-1 % over the step without any exchange (5.50 / 4.99 ms
    against 5.42-5.47 / 4.95-4.97 ms).  The host has nothing else to do meanwhile: what it still has to enqueue for this
    step and the next takes ~0.3 ms, the graph runs for 2-3 ms more (DEEPIPR_STAGED_HOST_WAIT=0 restores the stream-side
    wait).
The order of collectives is fixed by the stage plan, which depends only on the model and the batch shape, i.e. it is
identical on every rank.

Split-channel kernels.  The single-pass norm kernels of layers with fewer channels than CUs exchange partial sums
inside the launch and need all their workgroups co-resident (csrc: res_exchange).  A collective that still runs when
such a kernel starts holds some CUs, so part of its workgroups start only when the collective has finished: the
overlap degenerates to serialisation for that stretch, nothing worse -- the in-launch wait is bounded at seconds, a
collective of this size lasts well under a millisecond, and a collective kernel never waits for a kernel of this
library, so there is no cycle.  Two policies:
    "shared"     (default) collectives may overlap every stage; only the last stage's bucket (ResNet18: 2.7 MB) is exposed;
    "exclusive"  (DEEPIPR_OVERLAP_SYNC=0) a stage that contains split-channel launches never overlaps a collective:
                 outstanding all-reduces are waited for in front of it (the graph is split there, the one place where
                 the host has to hold the stream back) and the bucket finished just before it travels after it.  For
                 ResNet18 that is the last stage (stem, layer1, layer2): layer4's 75 % of the bytes travel under
                 layer3's backward, 25 % (11 MB) after the last stage.
The user's switch DEEPIPR_ALLOW_SYNC=0 removes the split-channel kernels altogether.  A time-out, should one ever
happen, is loud (NaN statistics + the flag Trainer / bench.py check).
"""
import os
import time

import torch

from deepipr_amd import cuts


class Stage:
    __slots__ = ('cut', 'params', 'has_sync')

    def __init__(self, cut, params):
        self.cut, self.params, self.has_sync = cut, params, False


def unwrap(model):
    """The net behind DistributedDataParallel / DualBranch wrappers."""
    while True:
        if hasattr(model, 'module'):
            model = model.module
        elif type(model).__name__ == 'DualBranch':
            model = model.model
        else:
            return model


def plan_stages(model, optimizer):
    """model.backward_stages() -> [Stage]; one stage (no cut) when the model names none or the plan does not
    partition the optimiser's parameters."""
    params = [p for g in optimizer.param_groups for p in g['params']]
    ids = {id(p) for p in params}
    base = unwrap(model)
    spec = base.backward_stages() if hasattr(base, 'backward_stages') else None
    single = [Stage(None, params)]
    if not spec or spec[-1][0] is not None:
        return single
    stages, seen = [], set()
    for cut, modules in spec:
        ps = []
        for m in modules:
            for p in m.parameters():
                if id(p) in ids and id(p) not in seen:
                    seen.add(id(p))
                    ps.append(p)
        if not ps:
            return single
        stages.append(Stage(cut, ps))
    return stages if seen == ids else single


def dry_run(run_once, model, optimizer):
    """Run one step WITHOUT advancing training: parameters, buffers and momentum are restored afterwards.  For what a
    first step has to do outside a hipGraph capture (MIOpen algorithm selection, lazily created optimiser state,
    per-stream kernel state) when the caller wants the capture to be the first real step."""
    base = unwrap(model)
    saved = {k: v.clone() for k, v in base.state_dict().items()}
    had_state = {id(p): ('momentum_buffer' in optimizer.state.get(p, {})) for g in optimizer.param_groups
                 for p in g['params']}
    mom = {id(p): optimizer.state[p]['momentum_buffer'].clone() for g in optimizer.param_groups
           for p in g['params'] if had_state[id(p)] and optimizer.state[p]['momentum_buffer'] is not None}
    flat_buf = optimizer.flat_buf.clone() if hasattr(optimizer, 'flat_buf') else None     # FlatSGD momentum
    run_once()
    with torch.no_grad():
        if flat_buf is not None:
            optimizer.flat_buf.copy_(flat_buf)
        for k, v in base.state_dict().items():
            v.copy_(saved[k])
        for g in optimizer.param_groups:
            for p in g['params']:
                buf = optimizer.state.get(p, {}).get('momentum_buffer')
                if buf is not None:
                    buf.copy_(mom[id(p)]) if id(p) in mom else buf.zero_()
    for m in base.modules():
        if hasattr(m, 'invalidate_key_cache'):
            m.invalidate_key_cache()                    # the in-place restore bumped the keys' version


class StagedStep:
    """step_fn must carry `forward_loss(model, data, target) -> (objective, outputs)` (trainer.train_step_v1 /
    trainer_private.train_step_v23 do).  graph=True (default on the GPU): every stage is a hipGraph captured once on
    the first batch shape; graph=False: the same stages dispatched eagerly (what the CPU / gloo tests run).

    The optimiser is used through the torch.optim interface plus, when it has them, FlatSGD's staged-exchange calls
    (configure_stages / exchange_stages / wait_exchange); any other optimiser simply steps after the last stage."""

    def __init__(self, step_fn, model, optimizer, data, target, graph=None, warmup=3, overlap_sync=None):
        self.forward_loss = step_fn.forward_loss
        self.model, self.optimizer = model, optimizer
        self.graph = data.is_cuda if graph is None else bool(graph)
        if overlap_sync is None:
            overlap_sync = os.environ.get('DEEPIPR_OVERLAP_SYNC', '1') != '0'
        self.overlap_sync = bool(overlap_sync)
        self.stages = plan_stages(model, optimizer)
        self.flat = hasattr(optimizer, 'configure_stages')
        if self.flat and not optimizer.configure_stages([s.params for s in self.stages]):
            self.stages = [Stage(None, [p for g in optimizer.param_groups for p in g['params']])]
            if not optimizer.configure_stages([self.stages[0].params]):
                raise RuntimeError('StagedStep: the optimiser does not hold the parameters of this model')
        self.names = [s.cut for s in self.stages if s.cut]
        self.kernels = None
        if data.is_cuda:
            from deepipr_amd import passport_ops
            self.kernels = passport_ops.kernels
        self.device = data.device
        self.static_data, self.static_target = data.clone(), target.clone()
        self.recaptures = 0
        self._graphs = None
        self._plan, self._events = [], {}
        self._one = None
        self.host_times = None            # set to {} to accumulate HOST time per phase of __call__ (diagnosis)
        self.host_waits = os.environ.get('DEEPIPR_STAGED_HOST_WAIT', '1') != '0'
        # ---- analysis + warm-up (eager, no collective before the last stage has run): which stages hold split-channel
        # launches, MIOpen algorithm selection, allocator pools, lazily created state -- all outside any capture
        if self.graph:
            self.stream = torch.cuda.Stream(device=self.device)
            self.stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.stream):
                self.kernels.prepare_stream(self.device, self.stream)
                self._warm(warmup)
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
            self._capture()
        else:
            self._warm(warmup)

    # ------------------------------------------------------------------ plan
    def _warm(self, warmup):
        def once():
            self._run_eager(self.static_data, self.static_target, analyse=True)
        if warmup == 0:
            dry_run(once, self.model, self.optimizer)      # training must not advance
        for _ in range(warmup):
            once()

    def _scope(self):
        """Split-channel kernels are allowed inside the stages: the schedule below keeps collectives away from the
        stages that use them (or the user chose to share, or switched them off -- sync_scope never overrides that)."""
        if self.kernels is None:
            import contextlib
            return contextlib.nullcontext()
        return self.kernels.sync_scope(True)

    def _exclusive(self, k):
        return self.stages[k].has_sync and not self.overlap_sync

    # ------------------------------------------------------------------ one stage of backward
    def _stage_backward(self, k, state):
        """state = (objective, rec, roots, root_grads) -> roots / root_grads of the next stage."""
        objective, rec, roots, root_grads = state
        st = self.stages[k]
        leaves = rec.down.get(st.cut, []) if st.cut else []
        inputs = list(leaves) + list(st.params)
        grads = torch.autograd.grad(roots, inputs, grad_outputs=root_grads, retain_graph=k + 1 < len(self.stages),
                                    allow_unused=True)
        for p, g in zip(st.params, grads[len(leaves):]):
            p.grad = g
        if not st.cut:
            return None
        ups = rec.up.get(st.cut, [])
        nxt, nxt_g = [objective], [self._one]
        for u, g in zip(ups, grads[:len(leaves)]):
            if g is not None:
                nxt.append(u)
                nxt_g.append(g)
        return objective, rec, nxt, nxt_g

    def _stage0(self, data, target):
        self.optimizer.zero_grad(set_to_none=True)
        rec = cuts.CutRecorder(self.names)
        with rec:
            objective, outputs = self.forward_loss(self.model, data, target)
        # an explicit d objective / d objective = 1 (a constant kept by the stepper): torch would otherwise launch a
        # ones_like fill kernel for the scalar root in every stage
        if self._one is None or self._one.device != objective.device or self._one.dtype != objective.dtype:
            self._one = torch.ones_like(objective)
        state = self._stage_backward(0, (objective, rec, [objective], [self._one]))
        return outputs, state

    # ------------------------------------------------------------------ exchange schedule (identical on every rank)
    def _before_stage(self, k):
        if self.flat and k > 0 and self._exclusive(k):
            self.optimizer.wait_exchange()             # no collective in flight while split-channel kernels run

    def _after_stage(self, k):
        """Stage k's gradients are complete on the current stream."""
        if not self.flat or k + 1 == len(self.stages):
            return                                      # the last stage's bucket goes out in optimizer.step()
        if self._exclusive(k + 1):
            return                                      # would have to be waited for at once: it travels after k + 1
        after = None
        if self.device.type == 'cuda':
            after = torch.cuda.Event()
            after.record(torch.cuda.current_stream(self.device))
        self.optimizer.exchange_stages(self._pending_lo, k + 1, after=after, overlap=True)
        self._pending_lo = k + 1

    # ------------------------------------------------------------------ eager form
    def _run_eager(self, data, target, analyse=False):
        self._pending_lo = 0
        with self._scope():
            count = self.kernels.sync_launches if self.kernels is not None else 0
            outputs, state = self._stage0(data, target)
            for k in range(len(self.stages)):
                if k > 0:
                    if not analyse:
                        self._before_stage(k)
                    state = self._stage_backward(k, state)
                if self.kernels is not None and analyse:
                    now = self.kernels.sync_launches
                    self.stages[k].has_sync = self.stages[k].has_sync or now > count
                    count = now
                if not analyse:
                    self._after_stage(k)
        self.optimizer.step()
        return outputs

    # ------------------------------------------------------------------ captured form
    def _overlapped(self):
        """Stages whose bucket is handed to the side stream right behind them (the rest travels at the end of the step):
        every stage but the last whose successor may overlap a collective."""
        n = len(self.stages)
        return {j for j in range(n - 1) if self.flat and not self._exclusive(j + 1)}

    def _groups(self):
        """Stages -> hipGraphs: ONE graph, split only in front of a stage the host has to hold the stream back for
        (exclusive policy: a stage with split-channel launches waits for the collectives launched before it)."""
        launched = self._overlapped()
        groups = [[0]]
        for k in range(1, len(self.stages)):
            if self.flat and self._exclusive(k) and any(j < k for j in launched):
                groups.append([k])
            else:
                groups[-1].append(k)
        return groups

    def _capture(self):
        from deepipr_amd import _lib
        opt = self.optimizer
        self._device_hyper = hasattr(opt, 'sync_hyper')
        if self._device_hyper:
            opt.sync_hyper()
        mode = 'thread_local' if torch.distributed.is_available() and torch.distributed.is_initialized() else 'global'
        from deepipr_amd.distributed import retire_collectives
        retire_collectives()                       # no RCCL watchdog poll may fall inside the captures below
        pool = torch.cuda.graph_pool_handle()
        self._plan, self._events, self._ranges, state = [], {}, {}, None
        opt.zero_grad(set_to_none=True)
        lo = 0
        with self._scope():
            groups, launched = self._groups(), self._overlapped()
            for group in groups:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool, stream=self.stream, capture_error_mode=mode):
                    for k in group:
                        if k == 0:
                            self.outputs, state = self._stage0(self.static_data, self.static_target)
                        else:
                            state = self._stage_backward(k, state)
                        if not self.flat:
                            continue
                        last = k + 1 == len(self.stages)
                        if last or k in launched:
                            # stage k's gradients are complete: pack the bucket(s) finished so far into the flat buffer
                            # INSIDE the graph (one cat kernel), then -- unless this is the end of the step -- an external
                            # event-record node the side stream will wait for after the launch
                            opt.pack_stages(lo, k + 1)
                            if not last:
                                self._events[k] = _lib.ExternalEvent()
                                self._events[k].record(self.stream)
                                self._ranges[k] = (lo, k + 1)
                            lo = k + 1
                self._plan.append((g, group))
        self._graphs = [g for g, _ in self._plan]
        # the gradient tensors the replays write (graph-pool memory): `.grad` must point at them whenever the optimiser
        # runs, also after an eager step in between (a ragged last batch) re-bound it
        self._captured_grads = [(p, p.grad) for g in opt.param_groups for p in g['params']]

    def __call__(self, data, target):
        if not self.graph:
            return self._run_eager(data, target)
        opt = self.optimizer
        if self._device_hyper:
            opt.sync_hyper()
        self.static_data.copy_(data, non_blocking=True)
        self.static_target.copy_(target, non_blocking=True)
        for p, g in self._captured_grads:
            p.grad = g
        tr = self.host_times
        t = time.perf_counter() if tr is not None else 0.0
        for i, (g, group) in enumerate(self._plan):
            if i:
                opt.wait_exchange()                     # exclusive policy: nothing in flight while this graph runs
            g.replay()
            if tr is not None:
                t = self._lap(tr, 'replay', t)
            for k in group:
                if k in self._events:                   # the bucket(s) packed behind stage k: all-reduce on the side stream
                    lo, hi = self._ranges[k]
                    if self.host_waits:
                        # Hold the HOST until the graph has passed the event, then enqueue the collective: a stream-side
                        # wait that sits in a second hardware queue for half a step slows every dispatch of the replayed
                        # graph next to it (+160 us per step for ResNet18 V1, +560 us for the V2 shard with twice the
                        # dispatches; profiles/r03_ddp_rehearsal*.jsonl).  The host has nothing else to do meanwhile: what
                        # it still has to enqueue for this step and the next takes ~0.3 ms, the graph runs for 2-3 ms more.
                        # The side stream then waits for NOTHING: a stream that waits for an event whose last record was a
                        # node of a captured graph is where RCCL's watchdog thread died one run in five ("operation not
                        # permitted on an event last recorded in a capturing stream" from its hipEventQuery of the
                        # collective's end event; tools/nccl_flake_probe.sh, profiles/r04_nccl_flake_probe.txt).
                        tw = time.perf_counter() if tr is not None else 0.0
                        self._events[k].synchronize()
                        if tr is not None:                  # every wait's length, per stage (bench.py --stage-host-wait-histogram)
                            tr.setdefault('waits', {}).setdefault(k, []).append(time.perf_counter() - tw)
                        opt.exchange_stages(lo, hi, after=False, overlap=True, packed=True)
                    else:
                        opt.exchange_stages(lo, hi, after=self._events[k], overlap=True, packed=True)
            if tr is not None:
                t = self._lap(tr, 'buckets', t)
        if self.flat:
            opt.rest_is_packed = True                   # the graph packed the remaining buckets too (true for THIS step only:
        try:                                            # an eager step in between -- a ragged batch -- packs for itself)
            opt.step()
        finally:
            if self.flat:
                opt.rest_is_packed = False
        if tr is not None:
            self._lap(tr, 'step', t)
        return self.outputs

    @staticmethod
    def _lap(tr, key, t0):
        t1 = time.perf_counter()
        tr[key] = tr.get(key, 0.0) + (t1 - t0)
        tr['calls_' + key] = tr.get('calls_' + key, 0) + 1
        return t1

    def close(self):
        """Drop the captured graphs, THEN the external events their event-record nodes refer to (an event must outlive
        every graph that records it)."""
        self._plan, self._graphs = [], None
        self._events = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def describe(self):
        sizes = self.optimizer.bucket_bytes() if self.flat else []
        return {'stages': [{'cut': s.cut, 'params': len(s.params), 'split_channel_kernels': s.has_sync,
                            'bucket_MB': round(sizes[i] / 1e6, 2) if i < len(sizes) else None}
                           for i, s in enumerate(self.stages)],
                'policy': 'shared' if self.overlap_sync else 'exclusive', 'graph': self.graph,
                'graphs_per_step': len(self._graphs) if self._graphs else 0}
