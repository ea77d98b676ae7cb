"""FlatSGD: SGD(momentum, weight decay) + data-parallel gradient exchange on flat HBM buffers.

The reference steps `optim.SGD(lr, momentum=0.9, weight_decay=1e-4)` (experiments/classification.py:47-50)
and, on several GPUs, lets `nn.DataParallel` gather gradients onto GPU 0 (experiments/trainer.py:92-93).
The stock MI355X replacement would be DistributedDataParallel + a multi-tensor optimiser; profiled on this
workload that costs ~50 extra per-parameter copy/scale kernels and ~0.5 ms per 6.6 ms step.  This module is
the MI355X-first form instead:

  * every parameter is a view into ONE flat fp32 buffer, likewise gradients and momentum (62 tensors,
    44.7 MB each for ResNet18): the optimiser step is a single streaming HIP kernel (20 B / parameter);
  * gradients are exchanged as FEW, LARGE all-reduces (xGMI is point-to-point and per-link bound: large
    messages, no per-tensor collectives): parameters are split into a few buckets in reverse layer order
    (75 % of ResNet18's bytes sit in layer4, whose gradients are ready first); each bucket is packed with one
    `cat` kernel and all-reduced on RCCL's stream as soon as its last gradient has been accumulated, i.e.
    while the earlier layers are still back-propagating; only the small last bucket (2.7 MB) waits for step();
  * one process per GPU; rank 0's parameters/buffers are broadcast once (distributed.broadcast_state).

It is a torch.optim.Optimizer (param_groups, zero_grad, lr schedulers work unchanged) and goes wherever the
reference passes its optimiser: `Trainer(model, FlatSGD(model.parameters(), lr=...), scheduler, device)`.
"""
import os

import torch
import torch.distributed as dist

from deepipr_amd import passport_ops as P

_ALIGN = 64            # floats: every parameter starts on a 256-byte boundary of the flat buffer


class FlatSGD(torch.optim.Optimizer):
    def __init__(self, params, lr=0.01, momentum=0.9, weight_decay=1e-4, bucket_fractions=(0.45, 0.8, 0.95),
                 process_group=None):
        # like torch.optim.SGD, frozen parameters (requires_grad=False) are never touched: they stay out of the
        # flat buffers altogether (no weight decay, no momentum, no exchange)
        params = [p for p in params if p.requires_grad]
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # DEEPIPR_FORCE_DDP=1: run the bucketed exchange even in a world of one (single-GPU rehearsal of the N>1 path)
        self.comm = self.world > 1 or (dist.is_initialized() and os.environ.get('DEEPIPR_FORCE_DDP') == '1')
        # How the gradient exchange is driven -- chosen by whoever drives the step, identical on every rank (never from
        # which hooks happened to fire locally):
        #   'hooks'  eager backward: buckets are launched from post-accumulate-grad hooks, overlapped with backward;
        #   'single' the backward was a replayed hipGraph without stages: ONE pack + ONE all-reduce in step();
        #   'staged' experiments/staged.py runs the backward stage by stage and calls exchange_stage() in between.
        self._mode = 'hooks'
        if self.comm:
            # collectives launched from hooks share the device with backward kernels: the co-residency-dependent
            # split-channel kernels are withheld while this optimiser drives an overlapped exchange (released again
            # by set_mode('single') / configure_stages(), and when the optimiser is garbage-collected)
            P.kernels.withhold_sync(self)
        plist = self.param_groups[0]['params']
        if len(self.param_groups) != 1:
            raise ValueError('FlatSGD keeps one parameter group (one lr / momentum / weight decay)')
        dev = plist[0].device
        if any(p.device != dev or p.dtype != torch.float32 for p in plist):
            raise ValueError('FlatSGD needs fp32 parameters on one device')
        # reverse registration order ~ the order in which backward finishes gradients (last layers first)
        order = list(reversed(plist))
        offsets, total = [], 0
        for p in order:
            offsets.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.flat_param = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_buf = torch.zeros(total, dtype=torch.float32, device=dev)
        self._slots = []
        self._zero_pool = torch.zeros(_ALIGN, dtype=torch.float32, device=dev)
        # {lr, momentum, weight_decay, 1/world} live in device memory and the update kernel reads them there: a step
        # captured into a hipGraph (launch arguments frozen at capture) then still follows an lr schedule --
        # sync_hyper() rewrites the four floats whenever param_groups changed (GraphedTrainStep calls it before
        # every replay; step() calls it when it runs eagerly).
        self._hyper = torch.zeros(4, dtype=torch.float32, device=dev)
        self._hyper_host = None
        # chunk table of the in-place update (_step_in_place): pinned host rows + their device copy, allocated here
        # because nothing may be allocated on the host side while a stream is being captured
        self._tables = []                 # one (pinned host rows, device copy) pair PER capture, kept alive
        self._table_ready = None          # the pair prepare_in_place_capture() made for the next capture
        self.in_place_captures = 0
        self._tab_rows = 0
        if dev.type == 'cuda' and not self.comm and hasattr(P.kernels, 'sgd_chunk'):
            chunk = P.kernels.sgd_chunk()
            self._tab_rows = sum((p.numel() + chunk - 1) // chunk for p in plist)
            self.prepare_in_place_capture()
        with torch.no_grad():
            for p, off in zip(order, offsets):
                view = self.flat_param[off:off + p.numel()].view_as(p)
                view.copy_(p)
                p.data = view                                     # the Parameter now lives in the flat buffer
                self._slots.append((p, off))
        # Buckets in gradient-ready order, cut where the cumulative size passes each fraction.  For ResNet18
        # (reverse order: linear, layer4.1 42 %, layer4.0 33 %, layer3 19 %, rest 6 %) that is four all-reduces
        # of 19 / 15 / 8.5 / 2.7 MB; every bucket but the last is launched from a gradient hook while earlier
        # layers are still back-propagating, so only the small last one is exposed.
        cuts, k = [], 0
        if self.comm and len(order) > 1:
            fr = sorted(f for f in bucket_fractions if 0.0 < f < 1.0)
            for i, (p, off) in enumerate(self._slots):
                acc = off + (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
                while k < len(fr) and acc >= fr[k] * total:
                    if i + 1 < len(order) and (not cuts or cuts[-1] != i + 1):
                        cuts.append(i + 1)
                    k += 1
        bounds = [0] + cuts + [len(order)]
        self._buckets = [(bounds[i], bounds[i + 1]) for i in range(len(bounds) - 1)]
        self._bucket_of = {}
        for b, (lo, hi) in enumerate(self._buckets):
            for i in range(lo, hi):
                self._bucket_of[id(self._slots[i][0])] = b
        self._pending = [hi - lo for lo, hi in self._buckets]
        self._launched = [False] * len(self._buckets)
        self._works = []
        self._hooks = []
        self._paused = False
        self._side = None                 # side stream of the staged exchange (pack + all-reduce off the main stream)
        self._side_busy = False
        self.rest_is_packed = False       # staged mode: the driver packed the last stages' buckets itself (captured)
        self.exposed_events = None        # staged mode, optional: [(start, stop)] events around the exposed exchange
        if self.comm and len(self._buckets) > 1:
            for p, _ in self._slots[:self._buckets[-1][0]]:          # every bucket but the last
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # ------------------------------------------------------------------ bucket plumbing
    def _range(self, b):
        lo, hi = self._buckets[b]
        start = self._slots[lo][1]
        last_p, last_off = self._slots[hi - 1]
        end = last_off + (last_p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        return start, end

    def _pack(self, b):
        """Gradients of bucket b -> their slots of flat_grad with ONE cat kernel (alignment pads and parameters
        that received no gradient are filled from a zero buffer)."""
        lo, hi = self._buckets[b]
        self._pack_slots(lo, hi)

    def _pack_slots(self, lo, hi):
        start = self._slots[lo][1]
        last_p, last_off = self._slots[hi - 1]
        end = last_off + (last_p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        pieces = []
        for p, off in self._slots[lo:hi]:
            n = p.numel()
            pad = (n + _ALIGN - 1) // _ALIGN * _ALIGN - n
            if p.grad is None:
                pieces.append(self._zeros(n + pad))
                continue
            pieces.append(p.grad.reshape(-1))
            if pad:
                pieces.append(self._zeros(pad))
        torch.cat(pieces, out=self.flat_grad[start:end])

    def _zeros(self, n):
        if self._zero_pool.numel() < n:
            self._zero_pool = torch.zeros(n, dtype=torch.float32, device=self.flat_grad.device)
        return self._zero_pool[:n]

    def _launch(self, b, async_op):
        self._pack(b)
        self._launched[b] = True
        if self.comm:
            start, end = self._range(b)
            w = dist.all_reduce(self.flat_grad[start:end], op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
            if async_op:
                self._works.append(w)

    def pause_hooks(self):
        """Context manager: gradient hooks do nothing inside (used while a hipGraph of forward+backward is being
        captured -- the exchange then happens in step(), after the replay)."""
        opt = self

        class _Pause:
            def __enter__(self):
                opt._paused = True

            def __exit__(self, *exc):
                opt._paused = False
                opt._pending = [hi - lo for lo, hi in opt._buckets]
                return False
        return _Pause()

    def set_mode(self, mode):
        """'hooks' | 'single' | 'staged' (see __init__).  Must be called with the same value on every rank."""
        if mode not in ('hooks', 'single', 'staged'):
            raise ValueError(mode)
        if mode != 'hooks':
            for h in self._hooks:
                h.remove()
            self._hooks = []
            P.kernels.release_sync(self)       # no collective is launched from inside a backward pass any more
        elif self.comm and not self._hooks:
            raise RuntimeError('FlatSGD: the gradient hooks were removed; build a new optimiser for the hooks mode')
        self._mode = mode

    def configure_stages(self, stage_params):
        """Buckets = the parameter lists of the staged backward (experiments/staged.py), in stage order.  Every stage
        must be a contiguous run of the flat layout (reverse registration order) and together they must cover it;
        -> True and mode 'staged', or False (layout does not fit: buckets and mode stay as they were)."""
        index = {id(p): i for i, (p, _) in enumerate(self._slots)}
        bounds, pos = [], 0
        for params in stage_params:
            ids = sorted(index[id(p)] for p in params if id(p) in index)
            if not ids:
                continue
            if ids[0] != pos or ids != list(range(ids[0], ids[-1] + 1)):
                return False
            pos = ids[-1] + 1
            bounds.append((ids[0], pos))
        if pos != len(self._slots):
            return False
        self._buckets = bounds
        self._bucket_of = {id(self._slots[i][0]): b for b, (lo, hi) in enumerate(bounds) for i in range(lo, hi)}
        self._pending = [hi - lo for lo, hi in bounds]
        self._launched = [False] * len(bounds)
        self.set_mode('staged')
        return True

    def bucket_bytes(self):
        return [4 * (self._range(b)[1] - self._range(b)[0]) for b in range(len(self._buckets))]

    def pack_stages(self, lo, hi):
        """Gradients of buckets lo..hi-1 -> their slots of the flat gradient buffer, on the current stream (one `cat`
        kernel).  The staged stepper CAPTURES this call right behind the stage that finished those gradients, so that
        the side stream only has to run the collective."""
        if lo < hi:
            self._pack_slots(self._buckets[lo][0], self._buckets[hi - 1][1])

    def exchange_stages(self, lo, hi, after=None, overlap=True, packed=False):
        """Staged mode: pack the gradients of buckets lo..hi-1 into the flat gradient buffer and all-reduce that
        range as ONE message.  overlap=True: on a side stream that first waits for the event `after` (recorded on the
        main stream behind the stage that finished these gradients; after=False: the caller already held the host
        until that event had passed, the side stream waits for nothing), so the main stream can go on with the next
        stage; the all-reduce is waited for in step() (or by wait_exchange()).  overlap=False: on the current stream.
        packed=True: the range was packed already (pack_stages, captured in the stage's graph): collective only."""
        if lo >= hi:
            return
        cuda = self.flat_grad.is_cuda
        start = self._slots[self._buckets[lo][0]][1]
        end = self._range(hi - 1)[1]

        def go():
            # pack (unless the stage's graph did it) + all-reduce as ORDINARY (async_op=False) operations of whichever
            # stream is current: the bucket travels on the side stream itself.  An async collective runs on the process
            # group's own stream behind two more cross-stream hand-offs: for the LAST bucket, which nothing overlaps
            # with, that measured +150 us per step (one MI355X, exchange forced on in a world of one,
            # profiles/r03_probe2.json); for the overlapped buckets the async form measured slower as well
            # (profiles/r03_ddp_rehearsal_async_buckets.jsonl).
            if not packed:
                self._pack_slots(self._buckets[lo][0], self._buckets[hi - 1][1])
            if self.comm:
                dist.all_reduce(self.flat_grad[start:end], op=dist.ReduceOp.SUM, group=self.group)
        if cuda and overlap:
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.flat_grad.device)
            if after is False:
                pass                                # the caller held the HOST until the gradients were there: nothing to wait for
            elif after is not None:
                after.wait(self._side)              # a torch.cuda.Event, or the external event of a captured stage
            else:
                self._side.wait_stream(torch.cuda.current_stream(self.flat_grad.device))
            with torch.cuda.stream(self._side):
                go()
            self._side_busy = True
        else:
            go()
        for b in range(lo, hi):
            self._launched[b] = True

    def wait_exchange(self):
        """Make the current stream wait for everything exchange_stages() put on the side stream (a stream-side wait:
        the host does not block)."""
        for w in self._works:
            w.wait()
        self._works = []
        if self._side is not None and self._side_busy:
            torch.cuda.current_stream(self.flat_grad.device).wait_stream(self._side)
            self._side_busy = False

    def _on_grad(self, param):
        if self._paused or self._mode != 'hooks':
            return
        b = self._bucket_of[id(param)]
        self._pending[b] -= 1
        if self._pending[b] == 0 and not self._launched[b]:
            self._launch(b, async_op=True)           # overlaps with the rest of backward

    # ------------------------------------------------------------------ optimiser interface
    def sync_hyper(self):
        """param_groups[0] -> the device-resident hyper-parameters, if they changed.  Must run OUTSIDE a hipGraph
        capture / replay (it is a host-to-device copy of four floats on the current stream)."""
        g = self.param_groups[0]
        host = (float(g['lr']), float(g['momentum']), float(g['weight_decay']), 1.0 / self.world)
        if host != self._hyper_host:
            self._hyper.copy_(torch.tensor(host, dtype=torch.float32), non_blocking=False)
            self._hyper_host = host

    def _missing_grads(self):
        return [i for i, (p, _) in enumerate(self._slots) if p.grad is None]

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        capturing = self.flat_param.is_cuda and torch.cuda.is_current_stream_capturing()
        if not capturing:
            self.sync_hyper()
        elif self._hyper_host is None:
            raise RuntimeError('FlatSGD: call sync_hyper() once before capturing step() into a hipGraph')
        missing = self._missing_grads()
        if capturing and not self.comm and not missing and self._step_in_place():
            return loss                                   # gradients read where autograd left them: no packing pass
        # The collective layout follows the MODE, which the driver of the step sets identically on every rank -- never
        # rank-local state such as which hooks happened to fire (ranks issuing different sets of collectives hang).
        if self._mode == 'staged':
            timed = self.exposed_events is not None and self.flat_grad.is_cuda and not capturing
            if timed:                                     # what is NOT hidden behind backward: from here to the update
                ev0 = torch.cuda.Event(enable_timing=True)
                ev0.record()
            rest = [b for b in range(len(self._buckets)) if not self._launched[b]]
            if rest:                                      # stages nobody exchanged yet: one message, current stream
                if rest != list(range(rest[0], len(self._buckets))):
                    raise RuntimeError('FlatSGD: staged exchange must proceed in stage order')
                self.exchange_stages(rest[0], len(self._buckets), overlap=False, packed=self.rest_is_packed)
            self.wait_exchange()
            if timed:
                ev1 = torch.cuda.Event(enable_timing=True)
                ev1.record()
                self.exposed_events.append((ev0, ev1))
        elif self._mode == 'single' or not self.comm or len(self._buckets) == 1:
            # One GPU, or the backward was a replayed hipGraph without stages (nothing to overlap with): ONE pack and
            # ONE all-reduce of the whole buffer.
            self._pack_slots(0, len(self._slots))
            if self.comm:
                dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.group)
        else:                                             # 'hooks': always the bucket layout
            for b in range(len(self._buckets)):
                if not self._launched[b]:
                    self._launch(b, async_op=False)
            for w in self._works:
                w.wait()
        if not missing:
            P.kernels.sgd_momentum_step_dev(self.flat_param, self.flat_grad, self.flat_buf, self._hyper)
        else:
            # torch.optim.SGD skips parameters whose grad is None (no weight decay, no momentum update): update the
            # runs of slots that did receive a gradient, one launch per run (rare path: unused branches)
            skip = set(missing)
            i, n = 0, len(self._slots)
            while i < n:
                if i in skip:
                    i += 1
                    continue
                j = i
                while j + 1 < n and (j + 1) not in skip:
                    j += 1
                lo = self._slots[i][1]
                last_p, last_off = self._slots[j]
                hi = last_off + (last_p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
                P.kernels.sgd_momentum_step_dev(self.flat_param[lo:hi], self.flat_grad[lo:hi], self.flat_buf[lo:hi],
                                                self._hyper)
                i = j + 1
        self._works = []
        self._launched = [False] * len(self._buckets)
        self._pending = [hi - lo for lo, hi in self._buckets]
        return loss

    def prepare_in_place_capture(self):
        """Allocate the chunk table of the NEXT captured in-place update (pinned host rows + device copy).  Nothing may
        be allocated on the host while a stream is being captured, so whoever captures calls this first
        (GraphedTrainStep does); without a prepared table the captured step takes the packed path."""
        if self._tab_rows and self._table_ready is None:
            dev = self.flat_param.device
            self._table_ready = (torch.empty((self._tab_rows, 3), dtype=torch.int64).pin_memory(),
                                 torch.empty((self._tab_rows, 3), dtype=torch.int64, device=dev))

    def _step_in_place(self):
        """One GPU, step being captured into a hipGraph: the gradient tensors' addresses are the ones every replay will
        use, so the update kernel can read them in place through a table of {address, flat offset, count} chunks built
        here once -- no `cat` of 44.7 MB of gradients into flat_grad per step.  -> False if a gradient cannot be
        addressed that way (the packed path is taken then)."""
        chunk = P.kernels.sgd_chunk()
        rows, total = [], 0
        for p, off in self._slots:
            g = p.grad
            if g.dtype != torch.float32 or g.device != self.flat_param.device or not g.is_contiguous():
                return False
            n, base = g.numel(), g.data_ptr()
            for lo in range(0, n, chunk):
                rows.append((base + 4 * lo, off + lo, min(chunk, n - lo)))
            total += n
        if self._table_ready is None or len(rows) > self._table_ready[0].shape[0]:
            return False                                  # (tables are allocated outside captures: prepare_in_place_capture)
        # every capture gets its OWN pinned rows + device copy: the captured memcpy node re-reads the pinned rows on
        # every replay, so a later capture (another batch shape, a re-capture) must not overwrite an earlier one's
        tab_host, tab_dev = self._table_ready
        self._table_ready = None
        self._tables.append((tab_host, tab_dev))
        tab_host[:len(rows)].copy_(torch.tensor(rows, dtype=torch.int64))
        table = tab_dev[:len(rows)]
        table.copy_(tab_host[:len(rows)], non_blocking=True)         # a memcpy node: replays re-read the pinned rows
        P.kernels.sgd_momentum_step_multi(self.flat_param, self.flat_buf, table, total, self._hyper)
        self.in_place_captures += 1
        self._works = []
        self._launched = [False] * len(self._buckets)
        self._pending = [hi - lo for lo, hi in self._buckets]
        return True

    def state_dict(self):
        sd = super().state_dict()
        sd['flat_momentum'] = self.flat_buf.detach().cpu()
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        buf = state_dict.pop('flat_momentum', None)
        super().load_state_dict(state_dict)
        if buf is not None:
            self.flat_buf.copy_(buf.to(self.flat_buf.device))
