"""Passport-layer operators: torch autograd Functions over the HIP C ABI (include/deepipr_hip.h).

PyTorch is plumbing here (device memory, streams, autograd bookkeeping); the arithmetic of the
passport layer runs in csrc/deepipr_hip.hip.  There is no CPU / eager fallback: a tensor that is not a
dense fp32 tensor on an AMD GPU raises, and so does a missing library (deepipr_amd._lib).

Reference lines each operator stands in for (paths relative to kamwoh/DeepIPR):
  gamma_beta          models/layers/passportconv2d.py:142-175  (get_scale / get_bias, passport branch)
  affine_relu         models/layers/passportconv2d.py:220-222
  sign_loss           models/losses/sign_loss.py:18-54
  passport_layer      all of the above in two launches forward, two backward
"""
import ctypes
import functools
import os
import threading
import weakref

import torch
import torch.utils.weak

from deepipr_amd import _lib

MARGIN = 0.1      # models/losses/sign_loss.py:27
L2 = 0.00001      # models/losses/sign_loss.py:53


def _chk(*tensors):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError('deepipr_amd passport ops run on the GPU only (got a %s tensor); there is no '
                               'CPU fallback' % t.device)
        if t.dtype not in (torch.float32, torch.float64, torch.int8):
            raise RuntimeError('deepipr_amd passport ops are fp32-only (got %s)' % t.dtype)
        if not t.is_contiguous():
            raise RuntimeError('deepipr_amd passport ops need dense contiguous tensors')
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError('deepipr_amd passport ops: tensors on different devices (%s, %s)' % (dev, t.device))
    return dev


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream(dev):
    """hipStream_t of torch's current stream on `dev` (raw handle; no Stream object is built on the hot path)."""
    if _raw_stream is not None and dev.index is not None:
        return _raw_stream(dev.index)
    return torch.cuda.current_stream(dev).cuda_stream


class _on:
    """Make `dev` the current HIP device for a launch; a no-op (one integer compare) in the normal
    one-process-per-GPU setting where it already is."""
    __slots__ = ('dev', 'prev')

    def __init__(self, dev):
        self.dev = dev
        self.prev = None

    def __enter__(self):
        idx = self.dev.index
        if idx is not None and torch.cuda.current_device() != idx:
            self.prev = torch.cuda.current_device()
            torch.cuda.set_device(idx)

    def __exit__(self, *exc):
        if self.prev is not None:
            torch.cuda.set_device(self.prev)
        return False


def _p(t):
    return None if t is None else t.data_ptr()


class HipKernels:
    """One method per C-ABI entry point; tensors in, freshly allocated tensors out."""

    def pooled_patch_mean(self, keys, kh, kw, stride, pad):
        # keys [nkeys, B, Ci, H, W] -> m [nkeys, Ci*kh*kw] float64
        dev = _chk(keys)
        nk, b, ci, h, w = keys.shape
        m = torch.empty((nk, ci * kh * kw), dtype=torch.float64, device=dev)
        with _on(dev):
            _lib.check(_lib.lib().deepipr_pooled_patch_mean(_p(keys), nk, b, ci, h, w, kh, kw, stride, pad, _p(m),
                                                           _stream(dev)), 'pooled_patch_mean')
        return m

    def gamma_beta_fwd(self, weight, m):
        dev = _chk(weight, m)
        co = weight.shape[0]
        k = weight.numel() // co
        gb = torch.empty((2, co), dtype=torch.float32, device=dev)
        with _on(dev):
            _lib.check(_lib.lib().deepipr_gamma_beta_fwd(_p(weight), _p(m), co, k, _p(gb[0]), _p(gb[1]),
                                                        _stream(dev)), 'gamma_beta_fwd')
        return gb[0], gb[1]

    def gamma_beta_bwd(self, dgamma, dbeta, m, wshape):
        dev = _chk(dgamma, dbeta, m)
        co = wshape[0]
        dw = torch.empty(wshape, dtype=torch.float32, device=dev)
        with _on(dev):
            _lib.check(_lib.lib().deepipr_gamma_beta_bwd(_p(dgamma), _p(dbeta), _p(m), co, dw.numel() // co, _p(dw),
                                                        _stream(dev)), 'gamma_beta_bwd')
        return dw

    def gamma_beta_bwd_acc(self, dgamma, dbeta, m, dw):
        """dw += dgamma (x) m_scale + dbeta (x) m_bias, in place: `dw` already holds the data conv's wgrad."""
        dev = _chk(dgamma, dbeta, m, dw)
        co = dw.shape[0]
        with _on(dev):
            _lib.check(_lib.lib().deepipr_gamma_beta_bwd_acc(_p(dgamma), _p(dbeta), _p(m), co, dw.numel() // co,
                                                            _p(dw), _stream(dev)), 'gamma_beta_bwd_acc')
        return dw

    def gamma_beta_fwd_multi(self, weights, ms, out=None):
        """gamma / beta of several passport layers in ONE launch (deepipr_gamma_beta_fwd_multi): weights[i] [Co_i, ...],
        ms[i] [2, K_i] float64.  -> [(gamma_i, beta_i)], views of one [2 * sum(Co)] allocation (`out` to reuse one)."""
        dev = _chk(*weights, *ms)
        n = len(weights)
        if not 0 < n <= _lib.GEMV_MAX_LAYERS or len(ms) != n:
            raise RuntimeError('gamma_beta_fwd_multi: 1..%d layers per call' % _lib.GEMV_MAX_LAYERS)
        cos = [w.shape[0] for w in weights]
        total = sum(cos)
        if out is None:
            out = torch.empty(2 * total, dtype=torch.float32, device=dev)
        arr = (_lib.GemvLayer * n)()
        res, off, base = [], 0, out.data_ptr()
        for i, (w, m) in enumerate(zip(weights, ms)):
            co = cos[i]
            arr[i] = _lib.GemvLayer(w.data_ptr(), m.data_ptr(), base + 4 * off, base + 4 * (off + co), co,
                                    w.numel() // co)
            res.append((out[off:off + co], out[off + co:off + 2 * co]))
            off += 2 * co
        with _on(dev):
            _lib.check(_lib.lib().deepipr_gamma_beta_fwd_multi(arr, n, _stream(dev)), 'gamma_beta_fwd_multi')
        return res

    def gamma_beta_bwd_multi(self, dgammas, dbetas, ms, dws, accumulate):
        """dW_i (+)= dgamma_i (x) m_scale_i + dbeta_i (x) m_bias_i for several layers in one launch."""
        dev = _chk(*dgammas, *dbetas, *ms, *dws)
        n = len(dws)
        if not 0 < n <= _lib.GEMV_MAX_LAYERS:
            raise RuntimeError('gamma_beta_bwd_multi: 1..%d layers per call' % _lib.GEMV_MAX_LAYERS)
        arr = (_lib.Rank2Layer * n)()
        for i in range(n):
            co = dws[i].shape[0]
            arr[i] = _lib.Rank2Layer(dgammas[i].data_ptr(), dbetas[i].data_ptr(), ms[i].data_ptr(), dws[i].data_ptr(),
                                     co, dws[i].numel() // co)
        with _on(dev):
            _lib.check(_lib.lib().deepipr_gamma_beta_bwd_multi(arr, n, int(bool(accumulate)), _stream(dev)),
                       'gamma_beta_bwd_multi')
        return dws

    def gamma_beta_dkey(self, dgamma, dbeta, weight, key_shape, stride, pad):
        dev = _chk(dgamma, dbeta, weight)
        co, ci, kh, kw = weight.shape
        b, _, h, w = key_shape
        lib = _lib.lib()
        ws = torch.empty(lib.deepipr_gamma_beta_dkey_workspace_bytes(ci, kh, kw), dtype=torch.uint8, device=dev)
        dkeys = torch.empty((2, b, ci, h, w), dtype=torch.float32, device=dev)
        with _on(dev):
            _lib.check(lib.deepipr_gamma_beta_dkey(_p(dgamma), _p(dbeta), _p(weight), co, b, ci, h, w, kh, kw,
                                                  stride, pad, _p(dkeys), _p(ws), _stream(dev)), 'gamma_beta_dkey')
        return dkeys[0], dkeys[1]

    def affine_relu_fwd(self, xhat, gamma, beta, relu):
        dev = _chk(xhat, gamma, beta)
        n, c = xhat.shape[0], xhat.shape[1]
        y = torch.empty_like(xhat)
        with _on(dev):
            _lib.check(_lib.lib().deepipr_affine_relu_fwd(_p(xhat), _p(gamma), _p(beta), _p(y), n, c,
                                                         xhat.numel() // (n * c), int(relu), _stream(dev)),
                       'affine_relu_fwd')
        return y

    def affine_relu_bwd(self, dy, xhat, gamma, beta, relu):
        dev = _chk(dy, xhat, gamma, beta)
        n, c = xhat.shape[0], xhat.shape[1]
        hw = xhat.numel() // (n * c)
        lib = _lib.lib()
        ws = torch.empty(lib.deepipr_affine_relu_bwd_workspace_bytes(n, c, hw), dtype=torch.uint8, device=dev)
        dx = torch.empty_like(xhat)
        dgb = torch.empty((2, c), dtype=torch.float32, device=dev)
        with _on(dev):
            _lib.check(lib.deepipr_affine_relu_bwd(_p(dy), _p(xhat), _p(gamma), _p(beta), _p(dx), _p(dgb[0]),
                                                  _p(dgb[1]), n, c, hw, int(relu), _p(ws), _stream(dev)),
                       'affine_relu_bwd')
        return dx, dgb[0], dgb[1]

    def sign_loss_fwd(self, gamma, b, alpha, margin=MARGIN, l2=L2):
        dev = _chk(gamma, b)
        c = gamma.numel()
        out = torch.empty(2, dtype=torch.float32, device=dev)
        bits = torch.empty(c, dtype=torch.int8, device=dev)
        with _on(dev):
            _lib.check(_lib.lib().deepipr_sign_loss_fwd(_p(gamma), _p(b), alpha, margin, l2, c, _p(out[0]),
                                                       _p(out[1]), _p(bits), _stream(dev)), 'sign_loss_fwd')
        return out[0], out[1], bits

    def sign_loss_bwd(self, dloss, gamma, b, alpha, margin=MARGIN, l2=L2):
        dev = _chk(dloss, gamma, b)
        dg = torch.empty_like(gamma)
        with _on(dev):
            _lib.check(_lib.lib().deepipr_sign_loss_bwd(_p(dloss), _p(gamma), _p(b), alpha, margin, l2,
                                                       gamma.numel(), _p(dg), _stream(dev)), 'sign_loss_bwd')
        return dg

    def passport_fwd(self, xhat, weight, m, b, alpha, relu, margin=MARGIN, l2=L2):
        """-> y, gamma, beta, loss, acc, bits (the last three None when b is None)."""
        dev = _chk(xhat, weight, m, b)
        n, c = xhat.shape[0], xhat.shape[1]
        hw = xhat.numel() // (n * c)
        k = weight.numel() // c
        y = torch.empty_like(xhat)
        gb = torch.empty((2, c), dtype=torch.float32, device=dev)
        sl = bits = None
        if b is not None:
            sl = torch.empty(2, dtype=torch.float32, device=dev)
            bits = torch.empty(c, dtype=torch.int8, device=dev)
        with _on(dev):
            _lib.check(_lib.lib().deepipr_passport_fwd(
                _p(xhat), _p(weight), _p(m), _p(b), alpha, margin, l2, n, c, hw, k, int(relu), _p(y), _p(gb[0]),
                _p(gb[1]), _p(sl[0]) if sl is not None else None, _p(sl[1]) if sl is not None else None, _p(bits),
                _stream(dev)), 'passport_fwd')
        if sl is None:
            return y, gb[0], gb[1], None, None, None
        return y, gb[0], gb[1], sl[0], sl[1], bits

    def passport_bwd(self, dy, xhat, gamma, beta, m, b, alpha, dloss, dgamma_extra, dbeta_extra, wshape, relu,
                     margin=MARGIN, l2=L2):
        """-> dxhat, dW (None when wshape is None: the caller accumulates it into the conv's wgrad), dgamma, dbeta."""
        dev = _chk(dy, xhat, gamma, beta, m, b, dloss, dgamma_extra, dbeta_extra)
        n, c = xhat.shape[0], xhat.shape[1]
        hw = xhat.numel() // (n * c)
        lib = _lib.lib()
        ws = torch.empty(lib.deepipr_passport_bwd_workspace_bytes(n, c, hw), dtype=torch.uint8, device=dev)
        dx = torch.empty_like(xhat)
        dw = torch.empty(wshape, dtype=torch.float32, device=dev) if wshape is not None else None
        dgb = torch.empty((2, c), dtype=torch.float32, device=dev)
        with _on(dev):
            _lib.check(lib.deepipr_passport_bwd(
                _p(dy), _p(xhat), _p(gamma), _p(beta), _p(m), _p(b), alpha, margin, l2, _p(dloss),
                _p(dgamma_extra), _p(dbeta_extra), n, c, hw, (dw.numel() // c) if dw is not None else 0, int(relu),
                _p(dx), _p(dw), _p(dgb[0]), _p(dgb[1]), _p(ws), _stream(dev)), 'passport_bwd')
        return dx, dw, dgb[0], dgb[1]


    # Scratch that is written and consumed inside one call (partial sums, the backward channel table) is kept in
    # a per-(device, stream) arena instead of being allocated per call: on one stream, calls are ordered, so
    # reuse is safe (also inside a captured hipGraph, which replays the same order).  Saves three allocator
    # round-trips per layer call on the host.
    _arena = {}

    # bench.py sets this while it times kernels: the calls that serve PASSPORT layers open the library's profile scope
    profile_passport = False

    _ws_bytes = {}

    def _bn_ws_bytes(self, n, c, hw):
        key = (n, c, hw)
        v = self._ws_bytes.get(key)
        if v is None:
            v = self._ws_bytes[key] = _lib.lib().deepipr_passport_bn_workspace_bytes(n, c, hw)
        return v

    def _scratch(self, dev, stream, nbytes):
        key = (dev.index, stream)
        buf = self._arena.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
            self._arena[key] = buf
        return buf.data_ptr()

    # In-launch exchange words of the register-resident kernels (include/deepipr_hip.h, DEEPIPR_SYNC_WORDS): one
    # zero-initialised buffer per (device, stream), owned by the library afterwards.
    #
    # Who may take the split-channel form (its workgroups must all be co-resident, which a concurrent kernel of
    # another stream -- an RCCL collective -- can delay):
    #   sync_user      the USER's switch (DEEPIPR_ALLOW_SYNC=0 or set_user_sync(False)): honoured on every path,
    #                  graph capture included; what check_exchange() tells the user to clear after a time-out;
    #   withheld       owners that currently launch collectives from inside a backward pass (FlatSGD in hooks mode,
    #                  DDP): while any is alive, eager calls do not get the exchange words;
    #   sync_scope()   an explicit decision for a region by a driver that KNOWS the region's concurrency (the staged
    #                  stepper: a stage that never overlaps a collective may use the split form although the optimiser
    #                  exchanges gradients) -- it overrides `withheld`, never the user's switch.
    _sync = {}
    sync_user = os.environ.get('DEEPIPR_ALLOW_SYNC', '1') != '0'
    _sync_withheld = None
    _sync_scoped = None
    sync_launches = 0                  # calls that were handed exchange words for a split-channel plan (S > 1)

    @property
    def allow_sync(self):
        if not self.sync_user:
            return False
        if self._sync_scoped is not None:
            return self._sync_scoped
        return not (self._sync_withheld is not None and len(self._sync_withheld))

    def set_user_sync(self, on):
        self.sync_user = bool(on)

    def withhold_sync(self, owner):
        import weakref
        if self._sync_withheld is None:
            self._sync_withheld = weakref.WeakSet()
        self._sync_withheld.add(owner)

    def release_sync(self, owner):
        if self._sync_withheld is not None:
            self._sync_withheld.discard(owner)

    def sync_scope(self, allowed):
        """Context manager: inside, allow_sync == (allowed and the user's switch), whatever owners withhold it."""
        k = self

        class _Scope:
            def __enter__(self):
                self.prev = k._sync_scoped
                k._sync_scoped = bool(allowed)

            def __exit__(self, *exc):
                k._sync_scoped = self.prev
                return False
        return _Scope()

    _slices = {}

    def bn_slices(self, n, c, hw):
        """Workgroups per channel of the single-pass BatchNorm kernels for this shape when exchange words are given
        (1 = no in-launch exchange)."""
        key = (n, c, hw)
        v = self._slices.get(key)
        if v is None:
            v = self._slices[key] = _lib.lib().deepipr_passport_bn_slices(n, c, hw)
        return v

    def _sync_words(self, dev, stream):
        if not self.allow_sync:
            return None
        key = (dev.index, stream)
        buf = self._sync.get(key)
        if buf is None:
            # The words must be created OUTSIDE a hipGraph capture: a captured zero-fill would be replayed with every
            # step and erase the time-out flag.  Whoever captures prepares its stream first (prepare_stream).
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('deepipr_amd: the exchange words of stream %#x must exist before it is captured: '
                                   'call passport_ops.kernels.prepare_stream(device, stream) first' % stream)
            buf = self._sync[key] = torch.zeros(_lib.SYNC_WORDS, dtype=torch.int32, device=dev)
        return buf.data_ptr()

    def prepare_stream(self, dev, stream=None):
        """Create the per-(device, stream) state the kernels need on `stream` (a torch.cuda.Stream, default: the
        current one) ahead of a hipGraph capture on it."""
        dev = torch.device(dev)
        if dev.index is None:
            dev = torch.device('cuda', torch.cuda.current_device())
        raw = (stream.cuda_stream if stream is not None else _stream(dev))
        if self.allow_sync and (dev.index, raw) not in self._sync:
            with torch.cuda.device(dev):
                self._sync[(dev.index, raw)] = torch.zeros(_lib.SYNC_WORDS, dtype=torch.int32, device=dev)
        self._scratch(dev, raw, 1 << 20)

    def sync_timeouts(self):
        """Number of (device, stream) exchange buffers whose bounded in-kernel wait ever expired (0 = healthy).
        One small D2H read per buffer: call it once per epoch, not per step."""
        return sum(int(buf[_lib.SYNC_TIMEOUT_WORD].item() != 0) for buf in self._sync.values())

    def check_exchange(self):
        """Raise if an in-launch partial-sum exchange of the single-pass norm kernels ever timed out (its outputs were
        poisoned with NaN).  The words are re-armed so that training can be restarted, e.g. with set_user_sync(False)."""
        n = self.sync_timeouts()
        if n:
            self.reset_sync_words()
            raise RuntimeError(
                'deepipr_amd: %d exchange buffer(s) report an expired in-kernel wait of the single-pass norm kernels '
                '(their workgroups were not co-resident: is another process or stream using this GPU?). The affected '
                'outputs were poisoned with NaN. Set DEEPIPR_ALLOW_SYNC=0 (or passport_ops.kernels.set_user_sync(False)) '
                'to use the three-launch form for the split-channel layers; a captured step must be captured again '
                'after that.' % n)

    def reset_sync_words(self):
        for buf in self._sync.values():
            buf.zero_()

    _resident = {}

    def bn_resident(self, n, c, hw):
        """Bit 0 / 1: forward / backward of this shape take the single-pass kernels (deepipr_passport_bn_resident)."""
        key = (n, c, hw, self.allow_sync)
        v = self._resident.get(key)
        if v is None:
            v = self._resident[key] = _lib.lib().deepipr_passport_bn_resident(n, c, hw, int(self.allow_sync))
        return v

    def bn_dual_supported(self, n, c, hw):
        """Both directions of this shape can take the dual form (deepipr_bn_dual_tail_*: a projection block's last two
        norm layers + tail in one launch)."""
        key = ('dual', n, c, hw, self.allow_sync)
        v = self._resident.get(key)
        if v is None:
            v = self._resident[key] = bool(_lib.lib().deepipr_bn_dual_tail_supported(n, c, hw, int(self.allow_sync)))
        return v

    def bn_dual_tail_fwd(self, xa, xb, ga, ba, gb, bb, stats_a, stats_b, relu_a=True, relu_b=True):
        """out = relu(act_a(bn_a(xa)) + act_b(bn_b(xb))) with learnable per-channel weight / bias, batch statistics;
        stats_* = (running_mean, running_var, num_batches_tracked, momentum, eps).  -> out, table_a, table_b."""
        dev = _chk(xa, xb)
        n, c = xa.shape[0], xa.shape[1]
        hw = xa.numel() // (n * c)
        st = _stream(dev)
        out = torch.empty_like(xa)
        tables = torch.empty((2, c, 8), dtype=torch.float32, device=dev)
        sync = self._sync_words(dev, st)
        if sync is not None and self.bn_slices(n, c, hw) > 1:
            self.sync_launches += 1
        (rma, rva, nbta, moma, epsa), (rmb, rvb, nbtb, momb, epsb) = stats_a, stats_b
        with _on(dev):
            _lib.check(_lib.lib().deepipr_bn_dual_tail_fwd(
                xa.data_ptr(), xb.data_ptr(), _p(ga), _p(ba), _p(gb), _p(bb), _p(rma), _p(rva), _p(nbta), _p(rmb),
                _p(rvb), _p(nbtb), float(moma), float(momb), float(epsa), float(epsb), int(relu_a), int(relu_b), n, c, hw,
                out.data_ptr(),
                tables[0].data_ptr(), tables[1].data_ptr(), sync, st), 'bn_dual_tail_fwd')
        return out, tables[0], tables[1]

    def bn_dual_tail_bwd(self, dy, dy2, out, xa, xb, table_a, table_b, relu_a=True, relu_b=True):
        """-> dxa, dxb, dgamma_a, dbeta_a, dgamma_b, dbeta_b."""
        dev = _chk(dy, dy2, out, xa, xb)
        n, c = xa.shape[0], xa.shape[1]
        hw = xa.numel() // (n * c)
        st = _stream(dev)
        dxa, dxb = torch.empty_like(xa), torch.empty_like(xb)
        d = torch.empty((4, c), dtype=torch.float32, device=dev)
        pd = d.data_ptr()
        sync = self._sync_words(dev, st)
        if sync is not None and self.bn_slices(n, c, hw) > 1:
            self.sync_launches += 1
        with _on(dev):
            _lib.check(_lib.lib().deepipr_bn_dual_tail_bwd(
                dy.data_ptr(), _p(dy2), out.data_ptr(), xa.data_ptr(), xb.data_ptr(), table_a.data_ptr(),
                table_b.data_ptr(), dxa.data_ptr(), dxb.data_ptr(), pd, pd + 4 * c, pd + 8 * c, pd + 12 * c, int(relu_a),
                int(relu_b), n, c, hw, sync, st), 'bn_dual_tail_bwd')
        return dxa, dxb, d[0], d[1], d[2], d[3]

    def passport_bn_fwd(self, x, weight, m, gamma_in, beta_in, b, alpha, relu, running_mean, running_var, nbt,
                        momentum, eps, training, margin=MARGIN, l2=L2, residual=None, pre=False, out=None):
        """BatchNorm(affine=False) + passport affine + ReLU (+ sign loss) from the conv output x; with `residual`
        (single-pass shapes only) y = relu(that + residual), the tail of a residual block.
        pre=True: gamma_in / beta_in ARE the passport gamma / beta of `weight` (computed by the net's batched GEMV
        launch): the layer's own GEMV is skipped, they are returned as gamma / beta.
        -> y, table[C,8], gamma, beta, loss, acc, bits  (gamma/beta None on the W-less public branch)."""
        dev = _chk(x, weight, residual)                     # the small per-channel vectors come from this module's own code
        if pre:
            pre_gamma, pre_beta, weight = gamma_in, beta_in, None
        n, c = x.shape[0], x.shape[1]
        hw = x.numel() // (n * c)
        lib = _lib.lib()
        st = _stream(dev)
        y = torch.empty_like(x) if out is None else out         # `out`: a dense tensor of x's shape (half of a StackShare buffer)
        # one small allocation: [table C*8 | gamma C | beta C | loss, acc] floats (+ bits as int8 when needed)
        small = torch.empty(c * 10 + 2, dtype=torch.float32, device=dev)
        base = small.data_ptr()
        p_gamma, p_beta, p_loss = base + 32 * c, base + 36 * c, base + 40 * c
        bits = None
        k = 0
        if weight is not None:
            k = weight.numel() // c
        if b is not None:
            bits = torch.empty(c, dtype=torch.int8, device=dev)
        ws = self._scratch(dev, st, self._bn_ws_bytes(n, c, hw)) if training else None
        sync = self._sync_words(dev, st) if training else None
        if sync is not None and self.bn_slices(n, c, hw) > 1:
            self.sync_launches += 1
        scoped = self.profile_passport and (b is not None or pre or weight is not None)
        if scoped:
            _lib.profile_scope(True)
        with _on(dev):
            _lib.check(lib.deepipr_passport_bn_fwd(
                x.data_ptr(), _p(weight), _p(m), _p(gamma_in), _p(beta_in), _p(b), alpha, margin, l2,
                _p(running_mean), _p(running_var), _p(nbt), momentum, eps, int(training), n, c, hw, k, int(relu),
                y.data_ptr(), base, p_gamma if weight is not None else None, p_beta if weight is not None else None,
                p_loss if b is not None else None, (p_loss + 4) if b is not None else None, _p(bits), _p(residual),
                ws, sync, st), 'passport_bn_fwd')
        if scoped:
            _lib.profile_scope(False)
        table = small[:8 * c].view(c, 8)
        gamma = beta = loss = acc = None
        if weight is not None:
            gamma, beta = small[8 * c:9 * c], small[9 * c:10 * c]
        elif pre:
            gamma, beta = pre_gamma, pre_beta
        if b is not None:
            loss, acc = small[10 * c], small[10 * c + 1]
        return y, table, gamma, beta, loss, acc, bits

    def passport_bn_bwd(self, dy, x, table, m, b, alpha, dloss, dgamma_extra, dbeta_extra, wshape, relu, training,
                        margin=MARGIN, l2=L2, dy2=None, tail_out=None, dx_out=None, dres_out=None):
        """-> dx, dW (None when wshape is None), dgamma, dbeta [, dres when tail_out is given: the fused residual
        tail, dres = (dy + dy2) * [tail_out > 0] is the shortcut's gradient]."""
        dev = _chk(dy, x, dy2, tail_out)
        n, c = x.shape[0], x.shape[1]
        hw = x.numel() // (n * c)
        lib = _lib.lib()
        st = _stream(dev)
        nws = (self._bn_ws_bytes(n, c, hw) + 255) // 256 * 256
        scratch = self._scratch(dev, st, nws + 32 * c)          # partial sums | backward channel table
        dx = torch.empty_like(x) if dx_out is None else dx_out
        dw = torch.empty(wshape, dtype=torch.float32, device=dev) if wshape is not None else None
        dgb = torch.empty((2, c), dtype=torch.float32, device=dev)
        pg = dgb.data_ptr()
        dres = (torch.empty_like(x) if dres_out is None else dres_out) if tail_out is not None else None
        sync = self._sync_words(dev, st)
        if sync is not None and self.bn_slices(n, c, hw) > 1:
            self.sync_launches += 1
        scoped = self.profile_passport and (m is not None or b is not None)
        if scoped:
            _lib.profile_scope(True)
        with _on(dev):
            _lib.check(lib.deepipr_passport_bn_bwd(
                dy.data_ptr(), x.data_ptr(), table.data_ptr(), _p(m), _p(b), alpha, margin, l2, _p(dloss),
                _p(dgamma_extra), _p(dbeta_extra), int(training), n, c, hw,
                (dw.numel() // c) if dw is not None else 0, int(relu), dx.data_ptr(), _p(dw), pg, pg + 4 * c,
                scratch + nws, scratch, sync, _p(dy2), _p(tail_out), _p(dres), st),
                'passport_bn_bwd')
        if scoped:
            _lib.profile_scope(False)
        if tail_out is not None:
            return dx, dw, dgb[0], dgb[1], dres
        return dx, dw, dgb[0], dgb[1]

    # ---- GroupNorm / InstanceNorm-fused layer (include/deepipr_hip.h: deepipr_passport_gn_*) ----
    _gn_ok = {}

    def gn_supported(self, n, c, hw, groups):
        key = (n, c, hw, groups)
        v = self._gn_ok.get(key)
        if v is None:
            v = self._gn_ok[key] = bool(_lib.lib().deepipr_passport_gn_supported(n, c, hw, groups))
        return v

    def passport_gn_fwd(self, x, weight, m, gamma_in, beta_in, b, alpha, relu, groups, eps, margin=MARGIN, l2=L2):
        """GroupNorm(groups, affine=False) + affine + ReLU (+ sign loss) from the conv output x.
        -> y, stats[N*groups,2], gamma, beta, loss, acc, bits (gamma/beta None on the W-less branch)."""
        dev = _chk(x, weight)
        n, c = x.shape[0], x.shape[1]
        hw = x.numel() // (n * c)
        st = _stream(dev)
        y = torch.empty_like(x)
        small = torch.empty(2 * n * groups + 2 * c + 2, dtype=torch.float32, device=dev)
        base = small.data_ptr()
        p_gamma = base + 8 * n * groups
        p_beta, p_loss = p_gamma + 4 * c, p_gamma + 8 * c
        bits = torch.empty(c, dtype=torch.int8, device=dev) if b is not None else None
        k = weight.numel() // c if weight is not None else 0
        with _on(dev):
            _lib.check(_lib.lib().deepipr_passport_gn_fwd(
                x.data_ptr(), _p(weight), _p(m), _p(gamma_in), _p(beta_in), _p(b), alpha, margin, l2, groups, eps,
                n, c, hw, k, int(relu), y.data_ptr(), base, p_gamma if weight is not None else None,
                p_beta if weight is not None else None, p_loss if b is not None else None,
                (p_loss + 4) if b is not None else None, _p(bits), st), 'passport_gn_fwd')
        stats = small[:2 * n * groups].view(n * groups, 2)
        gamma = beta = loss = acc = None
        if weight is not None:
            o = 2 * n * groups
            gamma, beta = small[o:o + c], small[o + c:o + 2 * c]
        if b is not None:
            o = 2 * n * groups + 2 * c
            loss, acc = small[o], small[o + 1]
        return y, stats, gamma, beta, loss, acc, bits

    def passport_gn_bwd(self, dy, x, stats, gamma, beta, m, b, alpha, dloss, dgamma_extra, dbeta_extra, wshape, relu,
                        groups, margin=MARGIN, l2=L2):
        """-> dx, dW (None when wshape is None), dgamma, dbeta."""
        dev = _chk(dy, x)
        n, c = x.shape[0], x.shape[1]
        hw = x.numel() // (n * c)
        st = _stream(dev)
        ws = self._scratch(dev, st, n * 2 * c * 8)
        dx = torch.empty_like(x)
        dw = torch.empty(wshape, dtype=torch.float32, device=dev) if wshape is not None else None
        dgb = torch.empty((2, c), dtype=torch.float32, device=dev)
        pg = dgb.data_ptr()
        with _on(dev):
            _lib.check(_lib.lib().deepipr_passport_gn_bwd(
                dy.data_ptr(), x.data_ptr(), stats.data_ptr(), _p(gamma), _p(beta), _p(m), _p(b), alpha, margin, l2,
                _p(dloss), _p(dgamma_extra), _p(dbeta_extra), groups, n, c, hw,
                (dw.numel() // c) if dw is not None else 0, int(relu), dx.data_ptr(), _p(dw), pg, pg + 4 * c, ws, st),
                'passport_gn_bwd')
        return dx, dw, dgb[0], dgb[1]

    def add_relu_fwd(self, a, b):
        dev = _chk(a, b)
        out = torch.empty_like(a)
        with _on(dev):
            _lib.check(_lib.lib().deepipr_add_relu_fwd(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(),
                                                      _stream(dev)), 'add_relu_fwd')
        return out

    def relu_bwd(self, dy, out, dy2=None):
        """(dy [+ dy2]) * [out > 0]."""
        dev = _chk(dy, out, dy2)
        dx = torch.empty_like(out)
        with _on(dev):
            _lib.check(_lib.lib().deepipr_relu_bwd2(dy.data_ptr(), _p(dy2), out.data_ptr(), dx.data_ptr(),
                                                   out.numel(), _stream(dev)), 'relu_bwd')
        return dx

    def subsample2(self, x):
        """x[:, :, ::2, ::2] as a dense tensor (even H, W): the pixel gather in front of a 1x1 stride-2 convolution."""
        dev = _chk(x)
        n, c, h, w = x.shape
        y = torch.empty((n, c, h // 2, w // 2), dtype=torch.float32, device=dev)
        with _on(dev):
            _lib.check(_lib.lib().deepipr_subsample2(_p(x), _p(y), n * c, h, w, _stream(dev)), 'subsample2')
        return y

    def upsample2_zero(self, dy, x_shape):
        """The adjoint: dx[:, :, ::2, ::2] = dy, zero elsewhere (every element written)."""
        dev = _chk(dy)
        n, c, h, w = x_shape
        dx = torch.empty(x_shape, dtype=torch.float32, device=dev)
        with _on(dev):
            _lib.check(_lib.lib().deepipr_upsample2_zero(_p(dy), _p(dx), n * c, h, w, _stream(dev)), 'upsample2_zero')
        return dx

    def maxpool3x3s2_fwd(self, x):
        """nn.MaxPool2d(3, 2, 1)(x) -> (y, slot): slot = the maximum's window position, one byte per output element."""
        dev = _chk(x)
        n, c, h, w = x.shape
        oh, ow = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        y = torch.empty((n, c, oh, ow), dtype=torch.float32, device=dev)
        slot = torch.empty((n, c, oh, ow), dtype=torch.uint8, device=dev)
        with _on(dev):
            _lib.check(_lib.lib().deepipr_maxpool3x3s2_fwd(_p(x), _p(y), slot.data_ptr(), n * c, h, w, _stream(dev)), 'maxpool3x3s2_fwd')
        return y, slot

    def maxpool3x3s2_bwd(self, dy, slot, x_shape):
        dev = _chk(dy)
        n, c, h, w = x_shape
        dx = torch.empty(x_shape, dtype=torch.float32, device=dev)
        with _on(dev):
            _lib.check(_lib.lib().deepipr_maxpool3x3s2_bwd(_p(dy), slot.data_ptr(), _p(dx), n * c, h, w, _stream(dev)), 'maxpool3x3s2_bwd')
        return dx

    def maxpool2x2s2_fwd(self, x):
        """nn.MaxPool2d(2, 2)(x) -> (y, slot) for even H and W a multiple of 4 (deepipr_maxpool2x2s2_fwd)."""
        dev = _chk(x)
        n, c, h, w = x.shape
        y = torch.empty((n, c, h // 2, w // 2), dtype=torch.float32, device=dev)
        slot = torch.empty((n, c, h // 2, w // 2), dtype=torch.uint8, device=dev)
        with _on(dev):
            _lib.check(_lib.lib().deepipr_maxpool2x2s2_fwd(_p(x), _p(y), slot.data_ptr(), n * c, h, w, _stream(dev)), 'maxpool2x2s2_fwd')
        return y, slot

    def maxpool2x2s2_bwd(self, dy, slot, x_shape):
        dev = _chk(dy)
        n, c, h, w = x_shape
        dx = torch.empty(x_shape, dtype=torch.float32, device=dev)
        with _on(dev):
            _lib.check(_lib.lib().deepipr_maxpool2x2s2_bwd(_p(dy), slot.data_ptr(), _p(dx), n * c, h, w, _stream(dev)), 'maxpool2x2s2_bwd')
        return dx

    def scalar_sums(self, terms, n_a):
        """-> 3 floats {sum of the first n_a single-element tensors, sum of the rest, both}: left-to-right fp32 adds in ONE launch
        (deepipr_scalar_sums) instead of len(terms) one-element aten::add launches."""
        dev = _chk(*terms)
        out = torch.empty(3, dtype=torch.float32, device=dev)
        arr = (ctypes.c_void_p * len(terms))(*[t.data_ptr() for t in terms])
        with _on(dev):
            _lib.check(_lib.lib().deepipr_scalar_sums(ctypes.addressof(arr), n_a, len(terms) - n_a, out.data_ptr(), _stream(dev)),
                       'scalar_sums')
        return out

    def sgd_momentum_step(self, flat_param, flat_grad, flat_buf, lr, momentum, weight_decay, grad_scale=1.0):
        """In-place SGD(momentum, weight decay) over flat fp32 buffers of equal length."""
        dev = _chk(flat_param, flat_grad, flat_buf)
        with _on(dev):
            _lib.check(_lib.lib().deepipr_sgd_momentum_step(_p(flat_param), _p(flat_grad), _p(flat_buf),
                                                           flat_param.numel(), lr, momentum, weight_decay,
                                                           grad_scale, _stream(dev)), 'sgd_momentum_step')

    # ---- head of the train step: cross-entropy + top-1 in one launch (include/deepipr_hip.h: deepipr_ce_*) ----
    def ce_usable(self, logits, target):
        return (logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 2 and target.dim() == 1
                and target.dtype == torch.int64 and target.device == logits.device
                and bool(_lib.lib().deepipr_ce_top1_supported(logits.shape[0], logits.shape[1])))

    def ce_top1_fwd(self, logits, target):
        """-> loss (mean CE), top-1 accuracy in percent, lse[N]."""
        dev = _chk(logits)
        n, c = logits.shape
        out = torch.empty(2 + n, dtype=torch.float32, device=dev)
        base = out.data_ptr()
        st = _stream(dev)
        ws = self._scratch(dev, st, 16 * n)                 # per-row loss terms and hits, f64 (written and consumed here)
        with _on(dev):
            _lib.check(_lib.lib().deepipr_ce_top1_fwd(logits.data_ptr(), target.data_ptr(), n, c, base, base + 4,
                                                     base + 8, ws, st), 'ce_top1_fwd')
        return out[0], out[1], out[2:]

    def ce_bwd(self, dloss, logits, target, lse):
        dev = _chk(dloss, logits, lse)
        n, c = logits.shape
        dlogits = torch.empty_like(logits)
        with _on(dev):
            _lib.check(_lib.lib().deepipr_ce_bwd(dloss.data_ptr(), logits.data_ptr(), target.data_ptr(),
                                                lse.data_ptr(), n, c, dlogits.data_ptr(), _stream(dev)), 'ce_bwd')
        return dlogits

    # ---- the CIFAR-geometry classifier: avg-pool + Linear in one launch (include/deepipr_hip.h: deepipr_pooled_linear_*) ----
    def pooled_linear_supported(self, n, c, hw, k):
        return bool(_lib.lib().deepipr_pooled_linear_supported(n, c, hw, k))

    def pooled_linear_fwd(self, x, weight, bias):
        """-> logits [N, K] = Linear(mean over the map of x [N, C, H, W]), pooled [N, C] (for the backward)."""
        dev = _chk(x, weight, bias)
        n, c, h, w = x.shape
        k = weight.shape[0]
        pooled = torch.empty((n, c), dtype=torch.float32, device=dev)
        logits = torch.empty((n, k), dtype=torch.float32, device=dev)
        with _on(dev):
            _lib.check(_lib.lib().deepipr_pooled_linear_fwd(x.data_ptr(), weight.data_ptr(), _p(bias), pooled.data_ptr(),
                                                           logits.data_ptr(), n, c, h * w, k, _stream(dev)), 'pooled_linear_fwd')
        return logits, pooled

    def pooled_linear_bwd(self, dlogits, weight, pooled, x_shape, with_bias):
        """-> dx [N, C, H, W], dW [K, C], db [K] (None without a bias)."""
        dev = _chk(dlogits, weight, pooled)
        n, c, h, w = x_shape
        k = weight.shape[0]
        dx = torch.empty(x_shape, dtype=torch.float32, device=dev)
        dw = torch.empty_like(weight)
        db = torch.empty(k, dtype=torch.float32, device=dev) if with_bias else None
        with _on(dev):
            _lib.check(_lib.lib().deepipr_pooled_linear_bwd(dlogits.data_ptr(), weight.data_ptr(), pooled.data_ptr(), dx.data_ptr(),
                                                           dw.data_ptr(), _p(db), n, c, h * w, k, _stream(dev)), 'pooled_linear_bwd')
        return dx, dw, db

    # ---- data convolution: weight gradient on the fp32 matrix cores (include/deepipr_hip.h: deepipr_conv_wgrad) ----
    _wgrad_ws = {}

    def conv_wgrad_workspace(self, n, ci, co, h, w, kh, kw, stride, pad):
        """Bytes of split-K workspace deepipr_conv_wgrad needs for this problem; 0 = shape outside the kernel (the
        caller keeps the library's weight gradient)."""
        key = (n, ci, co, h, w, kh, kw, stride, pad)
        v = self._wgrad_ws.get(key)
        if v is None:
            v = self._wgrad_ws[key] = int(_lib.lib().deepipr_conv_wgrad_workspace_bytes(*key))
        return v

    def set_conv_arith(self, mode):
        """'fp32' (default: the fp32 MFMA) or 'bf16x3' (opt-in, also DEEPIPR_CONV_ARITH=bf16x3: fp32 operands split exactly
        into three bf16 words, six products on the bf16 matrix cores, fp32 accumulation -- one fp32 rounding per product)
        for the 3x3 stride-1 weight gradients; process-wide (deepipr_conv_set_arith).  -> the previous mode."""
        before = self.conv_arith()
        _lib.check(_lib.lib().deepipr_conv_set_arith({'fp32': 0, 'bf16x3': 1}[mode]), 'conv_set_arith')
        self._wgrad_ws.clear()
        return before

    def conv_arith(self):
        return ('fp32', 'bf16x3')[_lib.lib().deepipr_conv_get_arith()]

    def set_conv_algo(self, algo):
        """'winograd' (default: F(2x2, 3x3) around the fp32 MFMA for the 3x3 stride-1 forward / backward-data) or 'direct'
        (the implicit GEMM); process-wide (deepipr_conv_set_algo, also DEEPIPR_CONV_ALGO at load time).  -> the previous one."""
        before = self.conv_algo()
        _lib.check(_lib.lib().deepipr_conv_set_algo({'direct': 0, 'winograd': 1}[algo]), 'conv_set_algo')
        self._conv_ok.clear()
        self._conv_ws.clear()
        self._conv_wino.clear()
        self._wgrad_ws.clear()
        return before

    def conv_algo(self):
        return ('direct', 'winograd')[_lib.lib().deepipr_conv_get_algo()]

    def conv_algo_is_winograd(self):
        return _lib.lib().deepipr_conv_get_algo() == 1

    _conv_wino = {}

    def conv_is_winograd(self, n, ci, co, h, w, k, stride, pad, direction):
        """This call (deepipr_conv_fwd_ws / _dgrad_ws of this shape) takes the Winograd kernel."""
        key = (n, ci, co, h, w, k, stride, pad, direction)
        v = self._conv_wino.get(key)
        if v is None:
            v = self._conv_wino[key] = bool(_lib.lib().deepipr_conv_algo_of(*key))
        return v

    def conv_wgrad(self, x, dy, wshape, stride, pad, dgamma=None, dbeta=None, m=None):
        """dW of conv(x, W) for upstream gradient dy, or None when the shape is outside the kernel.  With dgamma /
        dbeta / m the passport branch's rank-2 term is added in the same pass (deepipr_gamma_beta_bwd_acc's result)."""
        n, ci, h, w = x.shape
        co, _ci, kh, kw = wshape
        if _ci != ci or tuple(dy.shape) != (n, co, (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1):
            return None
        nbytes = self.conv_wgrad_workspace(n, ci, co, h, w, kh, kw, stride, pad)
        if not nbytes:
            return None
        dev = _chk(x, dy, dgamma, dbeta, m)
        st = _stream(dev)
        ws = self._scratch(dev, ('wgrad', st), nbytes)
        dw = torch.empty(wshape, dtype=torch.float32, device=dev)
        with _on(dev):
            _lib.check(_lib.lib().deepipr_conv_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), n, ci, co, h, w, kh, kw,
                                                    stride, pad, _p(dgamma), _p(dbeta), _p(m), ws, nbytes, st),
                       'conv_wgrad')
        return dw

    # ---- data convolution: forward / backward-data on the fp32 matrix cores (deepipr_conv_fwd / _dgrad) ----
    _conv_ok = {}

    def conv_supported(self, n, ci, co, h, w, k, stride, pad, direction):
        """direction 0 = forward, 1 = backward-data; h, w: the convolution's input map."""
        key = (n, ci, co, h, w, k, stride, pad, direction)
        v = self._conv_ok.get(key)
        if v is None:
            v = self._conv_ok[key] = bool(_lib.lib().deepipr_conv_supported(*key))
        return v

    def conv_fwd(self, x, weight, stride, pad, pre=None):
        """conv2d(x, weight) (no bias, square kernel / stride / padding), or None when the shape is outside the kernel.
        pre: the weight's (Uf, Ud) Winograd images of THIS step (wino_transform) -- taken where the call runs the Winograd
        kernel (deepipr_conv_fwd_pre: bit-identical, the filters are not transformed again by every workgroup)."""
        n, ci, h, w = x.shape
        co, _ci, k, _k = weight.shape
        if _ci != ci or _k != k or not self.conv_supported(n, ci, co, h, w, k, stride, pad, 0):
            return None
        dev = _chk(x, weight)
        y = torch.empty((n, co, h // stride, w // stride), dtype=torch.float32, device=dev)
        st = _stream(dev)
        nbytes = self.conv_workspace(n, ci, co, h, w, k, stride, pad, 0)       # split K over workgroups: deep layers
        ws = self._scratch(dev, ('conv', st), nbytes) if nbytes else None
        with _on(dev):
            if pre is not None and pre[0] is not None and self.conv_is_winograd(n, ci, co, h, w, k, stride, pad, 0):
                _lib.check(_lib.lib().deepipr_conv_fwd_pre(x.data_ptr(), pre[0].data_ptr(), y.data_ptr(), n, ci, co, h, w, ws,
                                                          nbytes, st), 'conv_fwd_pre')
            else:
                _lib.check(_lib.lib().deepipr_conv_fwd_ws(x.data_ptr(), weight.data_ptr(), y.data_ptr(), n, ci, co, h, w, k, stride,
                                                         pad, ws, nbytes, st), 'conv_fwd')
        return y

    # ---- Winograd images of the weights, once per step (include/deepipr_hip.h: deepipr_conv_wino_transform_multi) ----
    _wino_images = torch.utils.weak.WeakIdKeyDictionary()      # weight tensor OBJECT (by identity) -> [its data address, Uf, Ud]

    def wino_image_bytes(self, co, ci):
        return int(_lib.lib().deepipr_conv_wino_image_bytes(co, ci))

    def wino_transform(self, weights, backward=True):
        """Writes the Winograd images of `weights` ([Co][Ci][3][3], Co and Ci multiples of 32) in one launch per
        _lib.WINO_MAX_LAYERS of them -> [(Uf, Ud or None)].  The image buffers belong to the weight tensor OBJECT passed here
        (pass the parameters themselves, not temporaries): they stay put while its storage does -- what a replayed hipGraph
        needs -- are replaced when the object's storage is re-pointed (`p.data = ...`, model.to(), a FlatSGD built after a
        warm-up forward: ADVICE r05), go when the object goes, and are valid until the weights change."""
        out, todo = [], []
        for w in weights:
            co, ci = w.shape[0], w.shape[1]
            dev = _chk(w)
            buf = self._wino_images.get(w)
            nfl = self.wino_image_bytes(co, ci) // 4
            if buf is None or buf[0] != w.data_ptr() or buf[1].numel() != nfl or buf[1].device != w.device:
                buf = self._wino_images[w] = [w.data_ptr(), torch.empty(nfl, dtype=torch.float32, device=dev), None]
            if backward and buf[2] is None:
                buf[2] = torch.empty(nfl, dtype=torch.float32, device=dev)
            out.append((buf[1], buf[2] if backward else None))
            todo.append((w, buf[1], buf[2] if backward else None, co, ci, dev))
        for lo in range(0, len(todo), _lib.WINO_MAX_LAYERS):
            chunk = todo[lo:lo + _lib.WINO_MAX_LAYERS]
            arr = (_lib.WinoLayer * len(chunk))()
            for i, (w, uf, ud, co, ci, _dev) in enumerate(chunk):
                arr[i] = _lib.WinoLayer(w.data_ptr(), uf.data_ptr(), _p(ud), co, ci)
            dev = chunk[0][5]
            with _on(dev):
                _lib.check(_lib.lib().deepipr_conv_wino_transform_multi(ctypes.addressof(arr), len(chunk), _stream(dev)),
                           'conv_wino_transform_multi')
        return out


    _conv_ws = {}

    def conv_workspace(self, n, ci, co, h, w, k, stride, pad, direction):
        """Bytes of split-K workspace deepipr_conv_fwd_ws / _dgrad_ws use for this problem (0: the plain form)."""
        key = (n, ci, co, h, w, k, stride, pad, direction)
        v = self._conv_ws.get(key)
        if v is None:
            v = self._conv_ws[key] = int(_lib.lib().deepipr_conv_workspace_bytes(*key))
        return v

    def conv_dgrad(self, dy, weight, x_shape, stride, pad, pre=None):
        """Gradient of conv2d(x, weight) with respect to x, or None when the shape is outside the kernel.  pre: as conv_fwd
        (the backward-data image Ud written before the forward pass of the same step)."""
        n, ci, h, w = x_shape
        co, _ci, k, _k = weight.shape
        if _ci != ci or _k != k or not self.conv_supported(n, ci, co, h, w, k, stride, pad, 1):
            return None
        dev = _chk(dy, weight)
        dx = torch.empty((n, ci, h, w), dtype=torch.float32, device=dev)
        st = _stream(dev)
        nbytes = self.conv_workspace(n, ci, co, h, w, k, stride, pad, 1)
        ws = self._scratch(dev, ('conv', st), nbytes) if nbytes else None
        with _on(dev):
            if pre is not None and pre[1] is not None and self.conv_is_winograd(n, ci, co, h, w, k, stride, pad, 1):
                _lib.check(_lib.lib().deepipr_conv_dgrad_pre(dy.data_ptr(), pre[1].data_ptr(), dx.data_ptr(), n, ci, co, h, w, ws,
                                                            nbytes, st), 'conv_dgrad_pre')
            else:
                _lib.check(_lib.lib().deepipr_conv_dgrad_ws(dy.data_ptr(), weight.data_ptr(), dx.data_ptr(), n, ci, co, h, w, k,
                                                           stride, pad, ws, nbytes, st), 'conv_dgrad')
        return dx

    def sgd_chunk(self):
        return _lib.lib().deepipr_sgd_momentum_chunk()

    def sgd_momentum_step_multi(self, flat_param, flat_buf, table, total_elements, hyper):
        """SGD with the gradients read in place: `table` is an int64 device tensor [entries, 3] of
        {gradient chunk address, offset into the flat buffers, count} (include/deepipr_hip.h)."""
        dev = _chk(flat_param, flat_buf, hyper)
        if table.dtype != torch.int64 or not table.is_cuda or not table.is_contiguous():
            raise RuntimeError('sgd_momentum_step_multi: table must be a contiguous int64 device tensor')
        with _on(dev):
            _lib.check(_lib.lib().deepipr_sgd_momentum_step_multi(_p(flat_param), _p(flat_buf), table.data_ptr(),
                                                                 table.shape[0], int(total_elements), _p(hyper),
                                                                 _stream(dev)), 'sgd_momentum_step_multi')

    def sgd_momentum_step_dev(self, flat_param, flat_grad, flat_buf, hyper):
        """The same with {lr, momentum, weight_decay, grad_scale} read from the 4-float DEVICE tensor `hyper`, so a
        step captured in a hipGraph follows a learning-rate schedule."""
        dev = _chk(flat_param, flat_grad, flat_buf, hyper)
        with _on(dev):
            _lib.check(_lib.lib().deepipr_sgd_momentum_step_dev(_p(flat_param), _p(flat_grad), _p(flat_buf),
                                                               flat_param.numel(), _p(hyper), _stream(dev)),
                       'sgd_momentum_step_dev')


kernels = HipKernels()


# ---------------------------------------------------------------------------------------------
# pooled passport patches, cached per key version (keys are constant buffers while training)
# ---------------------------------------------------------------------------------------------
class PooledKeys:
    """Caches m = pooled_patch_mean([skey, key]) for one layer.  The cache key is the storage address
    and in-place version counter of both tensors (an optimiser step on trainable keys,
    passport_attack_3.py:232-243, bumps the version); every code path of the layer that REPLACES or refills the
    keys (set_key, lazily drawn random keys, load_state_dict) clears the cache explicitly, because a new tensor
    can land on a freed tensor's address with an equal version, and writes through `.data` or a c10d broadcast do
    not bump the version at all (whoever does that calls invalidate_key_cache(): distributed.broadcast_state, the
    dry run of a graph capture).  Inference-mode tensors carry no version counter: for them the means are
    recomputed on every call; under plain no_grad the cache is used like anywhere else."""

    def __init__(self):
        self._sig = None
        self._m = None

    def get(self, skey, key, kh, kw, stride, pad):
        if skey.shape != key.shape:
            raise RuntimeError('passport scale key %s and bias key %s must have the same shape'
                               % (tuple(skey.shape), tuple(key.shape)))
        if skey.is_inference() or key.is_inference():
            sig = None                                    # inference tensors carry no version counter: never cache
        else:
            sig = (skey.data_ptr(), skey._version, key.data_ptr(), key._version, tuple(key.shape), key.device,
                   kh, kw, stride, pad)
        if sig is None or sig != self._sig:
            with torch.no_grad():
                both = torch.stack([skey.detach(), key.detach()]).to(torch.float32).contiguous()
                self._m = kernels.pooled_patch_mean(both, kh, kw, stride, pad)
            self._sig = sig
        return self._m

    def clear(self):
        self._sig = self._m = None


def _grad_or_none(g):
    return None if g is None else g.contiguous()


# ---------------------------------------------------------------------------------------------
# autograd Functions
# ---------------------------------------------------------------------------------------------
class _AffineReLU(torch.autograd.Function):
    """y = relu?(gamma*xhat + beta) with free-standing gamma/beta (learnable scale/bias of the public
    branch, models/layers/passportconv2d_private.py:140-141,162-163)."""

    @staticmethod
    def forward(ctx, xhat, gamma, beta, relu):
        xhat, gamma, beta = xhat.contiguous(), gamma.contiguous(), beta.contiguous()
        y = kernels.affine_relu_fwd(xhat, gamma, beta, relu)
        ctx.save_for_backward(xhat, gamma, beta)
        ctx.relu = relu
        ctx.set_materialize_grads(False)
        return y

    @staticmethod
    def backward(ctx, dy):
        xhat, gamma, beta = ctx.saved_tensors
        if dy is None:
            return None, None, None, None
        dx, dg, db = kernels.affine_relu_bwd(dy.contiguous(), xhat, gamma, beta, ctx.relu)
        return dx, dg, db, None


class _GammaBeta(torch.autograd.Function):
    """(gamma, beta) = pooled passport conv of (skey, key) with the layer's own weight."""

    @staticmethod
    def forward(ctx, weight, skey, key, m, stride, pad):
        weight = weight.contiguous()
        gamma, beta = kernels.gamma_beta_fwd(weight, m)
        ctx.save_for_backward(weight, m)
        ctx.geom = (tuple(key.shape), stride, pad)
        ctx.set_materialize_grads(False)
        return gamma, beta

    @staticmethod
    def backward(ctx, dgamma, dbeta):
        weight, m = ctx.saved_tensors
        key_shape, stride, pad = ctx.geom
        if dgamma is None and dbeta is None:
            return None, None, None, None, None, None
        zeros = None
        if dgamma is None or dbeta is None:
            zeros = torch.zeros(weight.shape[0], dtype=torch.float32, device=weight.device)
        dgamma = zeros if dgamma is None else dgamma.contiguous()
        dbeta = zeros if dbeta is None else dbeta.contiguous()
        dw = kernels.gamma_beta_bwd(dgamma, dbeta, m, weight.shape) if ctx.needs_input_grad[0] else None
        dsk = dk = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dsk, dk = kernels.gamma_beta_dkey(dgamma, dbeta, weight, key_shape, stride, pad)
        return dw, dsk if ctx.needs_input_grad[1] else None, dk if ctx.needs_input_grad[2] else None, None, None, None


class _SignLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gamma, b, alpha, l2):
        gamma, b = gamma.contiguous().view(-1), b.contiguous().view(-1)
        loss, acc, bits = kernels.sign_loss_fwd(gamma, b, float(alpha), MARGIN, float(l2))
        ctx.save_for_backward(gamma, b)
        ctx.cfg = (float(alpha), float(l2))
        ctx.mark_non_differentiable(acc, bits)
        ctx.set_materialize_grads(False)
        return loss, acc, bits

    @staticmethod
    def backward(ctx, dloss, _dacc, _dbits):
        gamma, b = ctx.saved_tensors
        if dloss is None:
            return None, None, None, None
        alpha, l2 = ctx.cfg
        return kernels.sign_loss_bwd(dloss.contiguous(), gamma, b, alpha, MARGIN, l2), None, None, None


# The data convolution (models/layers/passportconv2d.py:218, conv2d.py:31).  This library's kernels stand in for the vendor
# library where they measured faster on MI355X (tools/conv_bench.py, tools/wgrad_bench.py, tools/conv1x1_bench.py):
#   3x3 stride 1         all three directions on the Winograd kernels (F(2x2, 3x3) forward / backward-data, F(3x3, 2x2) weight
#                        gradient), CIFAR and ImageNet map widths; the direct implicit GEMMs with DEEPIPR_CONV_ALGO=direct
#   3x3 / 1x1 stride 2   CIFAR map widths: deepipr_conv_fwd / _dgrad / _wgrad (the vendor library wraps its NHWC solvers for them in
#                        layout transposes and zero fills)
#   1x1 (any stride)     forward / backward-data as plain GEMMs over NCHW (_gemm_1x1), weight gradient deepipr_conv_1x1.inc;
#                        stride 2 at ImageNet widths behind / in front of the pixel gather (deepipr_subsample2 / _upsample2_zero)
# DEEPIPR_OWN_CONV = auto (default) | all (every shape the kernels support) | 0 (vendor library only);
# DEEPIPR_OWN_WGRAD=0 switches only the weight gradient off.  All of them are bit-reproducible.
OWN_CONV = os.environ.get('DEEPIPR_OWN_CONV', 'auto')
OWN_WGRAD = os.environ.get('DEEPIPR_OWN_WGRAD', '1') != '0' and OWN_CONV != '0'
# the pre-transformed form of the Winograd forward / backward-data kernels (wino_weights): on; and for weights of at most this
# many filters (Co * Ci)
WINO_PRE = os.environ.get('DEEPIPR_WINO_PRE', '1') != '0' and OWN_CONV != '0'
WINO_PRE_MAX_FILTERS = int(os.environ.get('DEEPIPR_WINO_PRE_MAX_FILTERS', 1 << 30))
# ... and from this batch size up: the one transform launch per step (138 MB written for ResNet18) is paid back by the 26
# convolution launches only when they are long enough (measured, bench.py replay: config R 128 images -1.5 %, V3 66 + 66
# stacked -9.6 %, AlexNet 64 -4.6 %; config-P shard 32 images +1.6 %: off)
WINO_PRE_MIN_BATCH = int(os.environ.get('DEEPIPR_WINO_PRE_MIN_BATCH', 48))
# ... below it only the SMALL weights (at most this many filters: layer1 / layer2 of ResNet18, 0.9 MB of images): their
# convolutions run the most workgroups per filter -- every one of them transforming the same 64 x 64 or 128 x 128 filters -- and
# their transform launch is short (config-P shard, 32 images: 1.859 -> 1.837 ms; with layer3 too 1.841, with none 1.859)
WINO_PRE_SMALL_FILTERS = int(os.environ.get('DEEPIPR_WINO_PRE_SMALL_FILTERS', 16384))


_OWN_CONV_FROM_ENV = 'DEEPIPR_OWN_CONV' in os.environ


def prefer_own_kernels():
    """Data-parallel runs: every convolution this library has a kernel for takes it, whatever the speed policy says
    (OWN_CONV 'auto' -> 'all'; an explicit DEEPIPR_OWN_CONV in the environment is left alone).  The ranks of a job must run
    the SAME kernels -- weak-scaling time is the slowest rank's, and their gradients should agree bit for bit -- and the one
    thing that differed between ranks in round 4's rehearsal was the vendor library's solver choice for the shapes `auto`
    leaves to it (rank 1 reads rank 0's find records but takes its own immediate-mode decision where one is missing:
    5.6e-5 of scale between the ranks' gradients).  This library's kernels leave nothing to choose and are
    bit-reproducible; what stays with the vendor library afterwards is the 3-channel stem's forward and the classifier GEMM.
    Costs a few us per step on one GPU (layer4.0's stride-2 forward / backward-data at 2 048 positions).  -> the policy now."""
    global OWN_CONV, OWN_WGRAD
    if not _OWN_CONV_FROM_ENV and OWN_CONV == 'auto':
        OWN_CONV = 'all'
        OWN_WGRAD = os.environ.get('DEEPIPR_OWN_WGRAD', '1') != '0'
    # ... and what this library has no instance for runs on the vendor library's DETERMINISTIC solvers (round 6: the V3 shard's 66
    # images on 4-wide maps are outside the stride-2 kernels -- a multiple of four images there -- and the vendor backward-data
    # that took them accumulates with atomics: a rank did not even agree with ITSELF bit for bit,
    # profiles/r06_nrank_rehearsal.jsonl: same_rank_repeat_agrees false).  DEEPIPR_VENDOR_DETERMINISTIC=0 leaves the choice alone.
    if os.environ.get('DEEPIPR_VENDOR_DETERMINISTIC', '1') != '0':
        torch.backends.cudnn.deterministic = True
    return OWN_CONV


def _own_ok(t, w):
    return t.is_cuda and t.dtype == torch.float32 and t.dim() == 4 and w.dim() == 4 and w.shape[2] == w.shape[3]


# auto: a stride-2 3x3 convolution goes to the own forward / backward-data kernels when it has at least this many output
# positions (N * OH * OW); below that (ResNet18's layer3.0 at batch <= 32, layer4.0 at batch < 128) the 64 x 64-tile kernels
# leave CUs idle and the vendor library, shims included, is faster (tools/conv_bench.py at batch 128 and 32,
# profiles/r04_conv_bench*.json).  At exactly 2 048 positions (layer4.0 of config R, the first passport layer's own data
# convolution) the two are at par -- 168 us against 164 with the vendor library's five layout transposes and zero fill -- and
# the own kernels take it (round 6): no vendor convolution and no layout shim is left in the config-R step behind the stem, and
# the step's last bits no longer depend on the vendor library's solver choice.  The 1x1 stride-2 shortcuts always win (8-21 us
# against 26-46).
OWN_MIN_POSITIONS = int(os.environ.get('DEEPIPR_OWN_CONV_MIN_POSITIONS', 2048))
# ... and a stride-1 3x3 convolution from this many output positions up (ResNet18 at batch 128: layer1 and layer2, where the
# direct kernel runs 78-82 us against Winograd's 88; 0 = never)
OWN_S1_MIN_POSITIONS = int(os.environ.get('DEEPIPR_OWN_CONV_S1_MIN_POSITIONS', 32768))


def _own_policy(n, h, w, k, stride, backward_data=False):
    if OWN_CONV == 'all':
        return True
    # torch.backends.cudnn.deterministic / torch.use_deterministic_algorithms: this library's kernels are bit-reproducible;
    # the vendor library's stride-2 backward-data is not always -- MIOpen's immediate mode can pick a split-K solver that
    # accumulates with atomics for exactly the shapes the speed policy leaves to it (layer4.0 of ResNet18 at small
    # batches: profiles/r03_determinism.md, r04_pytest_gpu_5_red.log) -- so a run that asks for determinism takes the own
    # kernel for those whatever the speed policy says.  (Forward and stride-1 stay with the policy: the vendor kernels
    # there are deterministic, and which kernel runs is part of what the parity tests pin.)
    if backward_data and stride == 2 and (torch.backends.cudnn.deterministic or torch.are_deterministic_algorithms_enabled()):
        return True
    if stride == 1:
        if k == 1:
            return OWN_1X1                          # k_conv1x1_gemm against the BLAS library's batched GEMM (tools/conv1x1_bench.py)
        return k == 3 and OWN_S1_MIN_POSITIONS > 0 and n * h * w >= OWN_S1_MIN_POSITIONS
    if stride != 2:
        return False
    return k == 1 or n * (h // 2) * (w // 2) >= OWN_MIN_POSITIONS


def _own_split(n, ci, co, h, wd, k, stride, pad, direction):
    """The planner splits K at least four ways for this call (few output positions: the deep layers at small batches, where
    the 64 x 64 tiles alone fill a quarter of the chip or less) -- there the own kernel beats the vendor library with its
    layout shims (512 -> 512 on 4x4 maps at batch 32: 34 us against 29 + 15; tools/conv_bench.py --batch 32,
    profiles/r04_conv_bench_bs32.json)."""
    nbytes = kernels.conv_workspace(n, ci, co, h, wd, k, stride, pad, direction)
    return nbytes >= 4 * 4 * n * (co if direction == 0 else ci) * (h // stride if direction == 0 else h) * (wd // stride if direction == 0 else wd)


def _own_fwd(x_in, w, stride, pad):
    if OWN_CONV == '0' or not _own_ok(x_in, w):
        return False
    n, ci, h, wd = x_in.shape
    if not kernels.conv_supported(n, ci, w.shape[0], h, wd, w.shape[2], stride, pad, 0):
        return False
    if kernels.conv_is_winograd(n, ci, w.shape[0], h, wd, w.shape[2], stride, pad, 0):
        return True                                   # 3x3 stride 1: the Winograd kernel beats both the direct one and the library
    return _own_policy(n, h, wd, w.shape[2], stride) or _own_split(n, ci, w.shape[0], h, wd, w.shape[2], stride, pad, 0)


def _own_dgrad(x_shape, w, stride, pad, dy):
    if OWN_CONV == '0' or not _own_ok(dy, w):
        return False
    n, ci, h, wd = x_shape
    if not kernels.conv_supported(n, ci, w.shape[0], h, wd, w.shape[2], stride, pad, 1):
        return False
    if kernels.conv_is_winograd(n, ci, w.shape[0], h, wd, w.shape[2], stride, pad, 1):
        return True
    return (_own_policy(n, h, wd, w.shape[2], stride, backward_data=True)
            or _own_split(n, ci, w.shape[0], h, wd, w.shape[2], stride, pad, 1))


def _own_wgrad(x_in, w, stride, pad):
    if not (OWN_WGRAD and _own_ok(x_in, w)):
        return False
    n, ci, h, wd = x_in.shape
    return bool(kernels.conv_wgrad_workspace(n, ci, w.shape[0], h, wd, w.shape[2], w.shape[3], stride, pad))


# A 1x1 pad-0 convolution is a GEMM over NCHW as it stands -- y[n] = W [Co, Ci] @ x[n] [Ci, HW] -- and its backward-data the
# transposed one.  Forward and backward-data go to the BLAS library as exactly that (a plain library GEMM: no convolution
# solver to choose -- in immediate mode the vendor convolution library answers some of these with an NHWC implicit GEMM
# wrapped in layout transposes, profiles/r06b_steady_state_r50.md); the weight gradient is this library's kernel
# (deepipr_conv_1x1.inc).  Stride 2 (the projection shortcuts at ImageNet map widths, where deepipr_conv_fwd has no
# instance): the same GEMMs behind / in front of the pixel gather (deepipr_subsample2 / deepipr_upsample2_zero).
GEMM_1X1 = os.environ.get('DEEPIPR_GEMM_1X1', '1') != '0'
# ... and where this library's own NCHW GEMM (deepipr_conv_fwd / _dgrad with k = 1, stride 1: k_conv1x1_gemm) has an instance it
# takes the forward / backward-data GEMMs instead of the BLAS library (DEEPIPR_OWN_1X1=0: the BLAS library for all of them)
OWN_1X1 = os.environ.get('DEEPIPR_OWN_1X1', '1') != '0'


def _gemm_1x1(x_shape, w, stride, pad, t):
    return (GEMM_1X1 and OWN_CONV != '0' and _own_ok(t, w) and w.shape[2] == 1 and pad == 0 and w.shape[1] == x_shape[1]
            and (stride == 1 or (stride == 2 and x_shape[2] % 2 == 0 and x_shape[3] % 2 == 0)))


def _gathered(ctx, x_in):
    """x_in[:, :, ::2, ::2] of a 1x1 stride-2 convolution: the copy its forward left with the node, or a fresh one."""
    xs = getattr(ctx, 'gathered', None) if ctx is not None else None
    return xs if xs is not None else kernels.subsample2(x_in)


def _conv_fwd(x_in, w, stride, pad, ctx=None):
    """The data convolution of a node's forward.  ctx: the node -- it keeps the weight's Winograd images of this step
    (_wino_pre; None outside wino_weights()) for its backward-data pass."""
    pre = _wino_pre(w)
    if ctx is not None:
        ctx.wino_pre = pre
    if _own_fwd(x_in, w, stride, pad):
        return kernels.conv_fwd(x_in, w, stride, pad, pre)
    if _gemm_1x1(x_in.shape, w, stride, pad, x_in):
        xs = x_in if stride == 1 else kernels.subsample2(x_in)
        if ctx is not None and stride == 2:
            ctx.gathered = xs                          # the weight gradient's operand
        n, ci, h, wd = xs.shape
        if stride == 2 and OWN_1X1 and kernels.conv_supported(n, ci, w.shape[0], h, wd, 1, 1, 0, 0):
            return kernels.conv_fwd(xs, w, 1, 0)       # the own GEMM on the gathered pixels
        # bmm with the weight as a stride-0 batch (torch.matmul would fold the batch into the GEMM's rows through a transposed
        # COPY of the activations: 40 ms per ResNet50 step, profiles/r06c_steady_state_r50_matmul_copy.md)
        return torch.bmm(w.view(1, w.shape[0], ci).expand(n, -1, -1), xs.view(n, ci, h * wd)).view(n, w.shape[0], h, wd)
    return torch.ops.aten.convolution(x_in, w, None, [stride, stride], [pad, pad], [1, 1], False, [0, 0], 1)


def _conv_dgrad(dconv, x_in, w, stride, pad, pre=None):
    if _own_dgrad(x_in.shape, w, stride, pad, dconv):
        return kernels.conv_dgrad(dconv, w, x_in.shape, stride, pad, pre)
    if _gemm_1x1(x_in.shape, w, stride, pad, dconv):
        n, co, oh, ow = dconv.shape
        ci = w.shape[1]
        if stride == 2 and OWN_1X1 and kernels.conv_supported(n, ci, co, oh, ow, 1, 1, 0, 1):
            return kernels.upsample2_zero(kernels.conv_dgrad(dconv, w, (n, ci, oh, ow), 1, 0), tuple(x_in.shape))
        dxs = torch.bmm(w.view(1, co, ci).transpose(1, 2).expand(n, -1, -1), dconv.reshape(n, co, oh * ow)).view(n, ci, oh, ow)
        return dxs if stride == 1 else kernels.upsample2_zero(dxs, tuple(x_in.shape))
    return torch.ops.aten.convolution_backward(dconv, x_in, w, None, [stride, stride], [pad, pad], [1, 1], False, [0, 0],
                                               1, [True, False, False])[0]


def _wgrad_operand(ctx, x_in, w, stride, pad):
    """-> (input, stride) deepipr_conv_wgrad takes for this convolution's weight gradient -- the input itself, or the gathered
    pixels of a 1x1 stride-2 convolution at stride 1 -- or None (the vendor library's weight gradient)."""
    if _own_wgrad(x_in, w, stride, pad):
        return x_in, stride
    if OWN_WGRAD and stride == 2 and _gemm_1x1(x_in.shape, w, stride, pad, x_in):
        n, ci, h, wd = x_in.shape
        if kernels.conv_wgrad_workspace(n, ci, w.shape[0], h // 2, wd // 2, 1, 1, 1, 0):
            return _gathered(ctx, x_in), 1
    return None


def _conv_bwd_acc(ctx, dconv, x_in, w, stride, pad, dg, db, m, defer=None):
    """Backward of the data convolution that ran inside a fused passport node.  -> dx_in, dW, deferred.
    Own weight-gradient kernel (deepipr_conv_wgrad): MIOpen's dgrad, then the wgrad with the passport branch's rank-2
    term added in its reduction pass -- the shared weight's three-way gradient is complete when it is first written.
    Otherwise MIOpen's dgrad / wgrad, then the rank-2 update added INTO that wgrad (deepipr_gamma_beta_bwd_acc) -- or,
    with `defer` = (share, index), left to the layer group's one launch (_Rank2Group), which is handed the wgrad buffer
    (deferred = True: the caller sends dgamma / dbeta to the group node)."""
    need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
    if not (need_dx or need_dw):                      # frozen first layer: nothing flows further
        return None, None, False
    opnd = _wgrad_operand(ctx, x_in, w, stride, pad) if need_dw else None
    if opnd is not None:
        dconv = dconv.contiguous()
        xw, sw = opnd
        if w.shape[1] % 32 == 0:                       # (every instance but the 3-channel stem's adds the rank-2 term in its reduction)
            dw = kernels.conv_wgrad(xw, dconv, w.shape, sw, pad, dg, db, m)
        else:                                          # the 3-channel stem instance has no fused rank-2 term: the separate accumulate pass
            dw = kernels.gamma_beta_bwd_acc(dg, db, m, kernels.conv_wgrad(xw, dconv, w.shape, sw, pad))
        dx = _conv_dgrad(dconv, x_in, w, stride, pad, getattr(ctx, 'wino_pre', None)) if need_dx else None
        return dx, dw, False
    own_dx = need_dx and (_own_dgrad(x_in.shape, w, stride, pad, dconv) or _gemm_1x1(x_in.shape, w, stride, pad, dconv))
    dx, dw, _ = torch.ops.aten.convolution_backward(
        dconv, x_in, w, None, [stride, stride], [pad, pad], [1, 1], False, [0, 0], 1,
        [need_dx and not own_dx, need_dw, False])
    if own_dx:
        dx = _conv_dgrad(dconv.contiguous(), x_in, w, stride, pad, getattr(ctx, 'wino_pre', None))
    if need_dw:
        dw = dw.contiguous()
        if defer is not None:
            defer[0].wgrads[defer[1]] = dw               # completed and handed to autograd by the group node
            return dx, None, True
        dw = kernels.gamma_beta_bwd_acc(dg, db, m, dw)
    return dx, dw, False


class StackShare:
    """The two branches of a V2 / V3 dual forward (trainer_private.py:159-171: model(x, ind=0) and model(x, ind=1) over the
    same batch) as HALVES OF ONE BUFFER.  Behind the point where the branches part, the private passport layers convolve two
    different inputs with the same weight; run in lockstep, that is ONE convolution of the 2N-image stack -- forward,
    backward-data and weight gradient (which then already is the branches' sum) -- while the norm + affine kernels still
    run per branch (their own batch statistics, the public branch's learnable scale / bias against the private branch's
    passport gamma / beta).  One StackShare per layer call pair hands the branch nodes the halves to write -- forward
    output, backward dx, the tail's shortcut gradient -- so that stacking and un-stacking never copy, and carries the
    private branch's dgamma / dbeta to the stacked convolution's weight gradient, whose reduction pass adds the passport
    branch's rank-2 term (deepipr_conv_wgrad): the shared weight's gradient is complete when it is first written."""
    __slots__ = ('n', 'bufs', 'rank2')

    def __init__(self, n):
        self.n, self.bufs, self.rank2 = n, {}, None

    def half(self, tag, b, like):
        """Branch b's half ([n, ...] view) of the buffer `tag`, shaped and placed like the per-branch tensor `like`."""
        buf = self.bufs.get(tag)
        if buf is None:
            buf = self.bufs[tag] = torch.empty((2 * self.n,) + tuple(like.shape[1:]), dtype=like.dtype, device=like.device)
        return buf[b * self.n:(b + 1) * self.n]


def _adjacent_halves(a, b):
    """a and b are the two dense halves, in this order, of one allocation."""
    return (a is not None and b is not None and a.shape == b.shape and a.dtype == b.dtype and a.device == b.device
            and a.is_contiguous() and b.is_contiguous() and a.numel() > 0
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
            and b.storage_offset() == a.storage_offset() + a.numel())


def _stack_pair(a, b):
    """[a; b] along the batch: without a copy when they are adjacent halves of one buffer, torch.cat otherwise (None = zeros)."""
    if a is None and b is None:
        return None
    if a is None:
        a = torch.zeros_like(b)
    if b is None:
        b = torch.zeros_like(a)
    if _adjacent_halves(a, b):
        shape = (2 * a.shape[0],) + tuple(a.shape[1:])
        return torch.empty(0, dtype=a.dtype, device=a.device).set_(a.untyped_storage(), a.storage_offset(), shape)
    return torch.cat([a.contiguous(), b.contiguous()], dim=0)


class _Unstack(torch.autograd.Function):
    """[2N, ...] -> its two [N, ...] halves (views).  Backward puts the halves' gradients back together: no copy when the
    branch nodes wrote them into the halves of one StackShare buffer."""

    @staticmethod
    def forward(ctx, x):
        n = x.shape[0] // 2
        ctx.set_materialize_grads(False)
        return x[:n], x[n:]

    @staticmethod
    def backward(ctx, g0, g1):
        return _stack_pair(g0, g1)


class _Restack(torch.autograd.Function):
    """(a, b) -> [a; b] along the batch (no copy for the halves of one StackShare buffer); backward hands the halves out."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.set_materialize_grads(False)
        return _stack_pair(a, b)

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None
        n = g.shape[0] // 2
        g = g.contiguous()
        return g[:n], g[n:]


def unstack(x):
    return _Unstack.apply(x.contiguous())


def restack(a, b):
    return _Restack.apply(a, b)


class _Conv2dOwn(torch.autograd.Function):
    """`conv(x)` of a plain bias-free convolution (models/layers/conv2d.py:31; passportconv2d.py:218 when the data
    convolution runs outside the fused node) with this library's kernels wherever the policy above picks them -- forward
    (deepipr_conv_fwd), backward-data (deepipr_conv_dgrad), weight gradient (deepipr_conv_wgrad) -- and the vendor library
    for the rest, per direction."""

    @staticmethod
    def forward(ctx, x, w, stride, pad, share=None):
        x, w = x.contiguous(), w.contiguous()
        ctx.save_for_backward(x, w)
        ctx.geom = (stride, pad)
        ctx.share = share                          # StackShare: may carry the private branch's (dgamma, dbeta, m) at backward time
        ctx.set_materialize_grads(False)
        return _conv_fwd(x, w, stride, pad, ctx)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, pad = ctx.geom
        if dy is None:
            return None, None, None, None, None
        dy = dy.contiguous()
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dx = dw = None
        r2 = ctx.share.rank2 if ctx.share is not None else None     # the passport branch's rank-2 term rides in the wgrad
        if ctx.share is not None:
            ctx.share.rank2 = None
        opnd = _wgrad_operand(ctx, x, w, stride, pad) if need_dw else None
        if opnd is not None:
            xw, sw = opnd
            if r2 is not None and w.shape[1] % 32 == 0:
                dw = kernels.conv_wgrad(xw, dy, w.shape, sw, pad, *r2)
                r2 = None
            else:
                dw = kernels.conv_wgrad(xw, dy, w.shape, sw, pad)
            need_dw = False
        if need_dx and (_own_dgrad(x.shape, w, stride, pad, dy) or _gemm_1x1(x.shape, w, stride, pad, dy)):
            dx = _conv_dgrad(dy, x, w, stride, pad, getattr(ctx, 'wino_pre', None))
            need_dx = False
        if need_dx or need_dw:
            vdx, vdw, _ = torch.ops.aten.convolution_backward(dy, x, w, None, [stride, stride], [pad, pad], [1, 1], False,
                                                              [0, 0], 1, [need_dx, need_dw, False])
            dx = vdx if need_dx else dx
            dw = vdw if need_dw else dw
        if r2 is not None and dw is not None:
            dw = kernels.gamma_beta_bwd_acc(r2[0], r2[1], r2[2], dw.contiguous())
        return dx, dw, None, None, None


_nnmod = torch.nn.modules.module
_GLOBAL_FWD_HOOKS, _GLOBAL_FWD_PRE_HOOKS = _nnmod._global_forward_hooks, _nnmod._global_forward_pre_hooks
_GLOBAL_BWD_HOOKS, _GLOBAL_BWD_PRE_HOOKS = _nnmod._global_backward_hooks, _nnmod._global_backward_pre_hooks


def conv_plain(conv, x):
    """THE eligibility test of the own-kernel route, in one place: conv(x) may bypass Module.__call__ -- a plain
    bias-free square nn.Conv2d on the GPU that nobody hooked (module hooks, the process-wide module hooks that
    Module.__call__ also runs -- FlopCounterMode, register_module_forward_hook --, a wrapped forward) outside autocast.
    conv2d() routes by it, and PassportLayerBase.stackable() asks it before a dual forward commits to lockstep branches
    (conv2d(share=...) has no module-call fallback)."""
    return (OWN_CONV != '0' and x.is_cuda and conv.bias is None and conv.groups == 1 and tuple(conv.dilation) == (1, 1)
            and conv.padding_mode == 'zeros' and conv.stride[0] == conv.stride[1]
            and isinstance(conv.padding, tuple) and conv.padding[0] == conv.padding[1]
            and conv.kernel_size[0] == conv.kernel_size[1]
            and not (conv._forward_hooks or conv._forward_pre_hooks or conv._backward_hooks or conv._backward_pre_hooks)
            and not (_GLOBAL_FWD_HOOKS or _GLOBAL_FWD_PRE_HOOKS or _GLOBAL_BWD_HOOKS or _GLOBAL_BWD_PRE_HOOKS)   # Module.__call__ runs them
            and not torch.is_autocast_enabled()
            and type(conv) is torch.nn.Conv2d and 'forward' not in conv.__dict__)      # nor wrapped its forward


def conv2d(conv, x, share=None):
    """conv(x) for an nn.Conv2d.  A plain convolution nobody hooked (conv_plain) for which this library has a kernel in at
    least one direction goes through _Conv2dOwn; anything else is the module call.  share: a StackShare whose private-branch
    node leaves its dgamma / dbeta for this convolution's weight gradient (the caller has asked conv_plain:
    PassportLayerBase.stackable) -- always through _Conv2dOwn then."""
    if conv_plain(conv, x):
        st, pd, w = conv.stride[0], conv.padding[0], conv.weight
        if share is not None:
            return _Conv2dOwn.apply(x, w, st, pd, share)
        if ((torch.is_grad_enabled() and w.requires_grad and _own_wgrad(x, w, st, pd)) or _own_fwd(x, w, st, pd)
                or _gemm_1x1(x.shape, w, st, pd, x)
                or (torch.is_grad_enabled() and x.requires_grad and _own_dgrad(x.shape, w, st, pd, x))):
            return _Conv2dOwn.apply(x, w, st, pd)
    if share is not None:
        raise RuntimeError('deepipr_amd: conv2d(share=...) needs a plain bias-free square convolution on the GPU')
    return conv(x)


class _Rank2Share:
    """What the layers of one _Rank2Group leave for it during backward: MIOpen's wgrad buffer per layer."""
    __slots__ = ('wgrads',)

    def __init__(self):
        self.wgrads = {}


class _Rank2Group(torch.autograd.Function):
    """The passport branch's weight gradient of SEVERAL layers in one launch (deepipr_gamma_beta_bwd_multi).

    Forward is bookkeeping only: the gamma / beta pairs the batched GEMV launch already produced (gamma_beta_batch) are
    handed out as outputs of this node, so that autograd knows they come from the layers' weights.  Each layer's fused
    node takes its pair as gamma_in / beta_in; in backward it returns dgamma / dbeta (sign-loss gradient included)
    for them instead of launching its own rank-2 update, returns NO weight gradient itself and leaves MIOpen's wgrad
    buffer of its weight with the share.  This node's backward runs when ALL its layers have been through backward
    (autograd's dependency count), adds dgamma_i (x) m_scale_i + dbeta_i (x) m_bias_i INTO those buffers in one launch
    and returns them as the weights' gradients: autograd only ever sees a weight's complete three-way gradient, as
    one tensor (a weight used by two forward passes -- schemes V2 / V3 -- gets one such tensor per pass, summed by
    autograd as before).  ResNet18: 5 launches of 4-7 us -> one of 11 us (back to back) / 18 us (in the step) for
    67 MB (profiles/r03_gemv_bench.json).  A group never spans two backward stages of the staged step
    (gamma_beta_batch)."""

    @staticmethod
    def forward(ctx, share, ms, pairs, *weights):
        ctx.share, ctx.ms, ctx.n = share, ms, len(weights)
        ctx.set_materialize_grads(False)
        return tuple(t.view_as(t) for pair in pairs for t in pair)

    @staticmethod
    def backward(ctx, *grads):
        share, dgs, dbs, ms, dws, out = ctx.share, [], [], [], [], [None] * ctx.n
        for i in range(ctx.n):
            dg, db = grads[2 * i], grads[2 * i + 1]
            if dg is None and db is None:
                continue                                   # this layer took no part (or did its own update)
            dw = share.wgrads.pop(i, None)
            if dw is None or dg is None or db is None:
                raise RuntimeError('deepipr_amd: rank-2 group: layer %d sent dgamma / dbeta without its wgrad buffer' % i)
            dgs.append(dg.contiguous())
            dbs.append(db.contiguous())
            ms.append(ctx.ms[i])
            dws.append(dw)
            out[i] = dw
        if share.wgrads:
            raise RuntimeError('deepipr_amd: rank-2 group: wgrad buffers left without dgamma / dbeta: %s'
                               % sorted(share.wgrads))
        if dws:
            kernels.gamma_beta_bwd_multi(dgs, dbs, ms, dws, accumulate=True)
        return (None, None, None) + tuple(out)


class _PassportLayer(torch.autograd.Function):
    """The fused passport layer after the norm: two launches forward, two backward.

    inputs : xhat, weight, skey, key, b (or None), m (pooled means, no grad)
    outputs: y, gamma, beta, loss, acc, bits  (loss/acc/bits are zero-size dummies when b is None)
    """

    @staticmethod
    def forward(ctx, xhat, weight, skey, key, b, m, alpha, relu, stride, pad, conv_inside=False):
        xhat, weight = xhat.contiguous(), weight.contiguous()
        x_in = None
        if conv_inside:                               # no norm between conv and affine: the data conv runs in this node
            x_in = xhat
            xhat = _conv_fwd(x_in, weight, stride, pad, ctx)
        bb = None if b is None else b.contiguous().view(-1)
        y, gamma, beta, loss, acc, bits = kernels.passport_fwd(xhat, weight, m, bb, float(alpha), relu)
        ctx.save_for_backward(xhat, weight, gamma, beta, m, bb, x_in)
        ctx.cfg = (float(alpha), relu, stride, pad, tuple(key.shape))
        ctx.set_materialize_grads(False)
        if bb is None:
            loss = acc = xhat.new_empty(0)
            bits = torch.empty(0, dtype=torch.int8, device=xhat.device)
        ctx.mark_non_differentiable(acc, bits)
        return y, gamma, beta, loss, acc, bits

    @staticmethod
    def backward(ctx, dy, dgamma_extra, dbeta_extra, dloss, _dacc, _dbits):
        xhat, weight, gamma, beta, m, bb, x_in = ctx.saved_tensors
        alpha, relu, stride, pad, key_shape = ctx.cfg
        if dy is None:
            dy = torch.zeros_like(xhat)
        dl = None if (bb is None or dloss is None) else dloss.contiguous()
        dx, dw, dg, db = kernels.passport_bwd(dy.contiguous(), xhat, gamma, beta, m, bb, alpha, dl,
                                              _grad_or_none(dgamma_extra), _grad_or_none(dbeta_extra),
                                              None if x_in is not None else weight.shape, relu)
        if x_in is not None:
            dx, dw, _ = _conv_bwd_acc(ctx, dx, x_in, weight, stride, pad, dg, db, m)
        dsk = dk = None
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            dsk, dk = kernels.gamma_beta_dkey(dg, db, weight, key_shape, stride, pad)
        return (dx, dw, dsk if ctx.needs_input_grad[2] else None, dk if ctx.needs_input_grad[3] else None,
                None, None, None, None, None, None, None)


class _PassportBNLayer(torch.autograd.Function):
    """BatchNorm2d(affine=False) + passport affine + ReLU + sign loss, fused (deepipr_passport_bn_fwd / _bwd): one
    register-resident launch per direction when the activation fits, three otherwise; the normalised activation
    is never written.  With weight=None it is the public branch of a PassportPrivateBlock (learnable gamma_in /
    beta_in).  With `residual` the block's tail relu(layer + residual) is folded in as well.  The output comes
    out twice (y, y'): two handles of the same tensor for the two consumers of a residual block's output, whose
    gradients are then summed inside the backward kernel (tail form) instead of by an ATen add.

    `conv` = (stride, pad): `x` is the layer's INPUT and the data convolution with the same `weight` runs inside
    this node (aten::convolution -> MIOpen, unchanged).  The point is the shared weight's three-way gradient
    (models/layers/passportconv2d.py:148,169,218 use one W three times): backward calls MIOpen's wgrad and then ADDS
    the passport branch's rank-2 update into that buffer (deepipr_gamma_beta_bwd_acc, 8 B per weight) instead of
    writing a second full-size dW that autograd sums with an extra 12 B-per-weight add kernel."""

    @staticmethod
    def forward(ctx, x, weight, skey, key, gamma_in, beta_in, b, m, running_mean, running_var, nbt, residual, cfg):
        alpha, relu, stride, pad, training, momentum, eps, conv = cfg[:8]
        ctx.defer = cfg[8] if len(cfg) > 8 else None     # (Rank2Group share, index): see _Rank2Group
        ctx.stack = stack = cfg[9] if len(cfg) > 9 else None     # (StackShare, branch, dx into the share?): see StackShare
        x = x.contiguous()
        w = None if weight is None else weight.contiguous()
        x_in = None
        if conv is not None:
            x_in = x
            x = _conv_fwd(x_in, w, stride, pad, ctx)
        gi = None if gamma_in is None else gamma_in.contiguous().view(-1)
        bi = None if beta_in is None else beta_in.contiguous().view(-1)
        bb = None if b is None else b.contiguous().view(-1)
        res = None if residual is None else residual.contiguous()
        # weight AND gamma_in / beta_in: the latter are this weight's passport gamma / beta, already computed by the net's
        # batched GEMV launch (gamma_beta_batch); backward is the passport branch's either way
        y, table, gamma, beta, loss, acc, bits = kernels.passport_bn_fwd(
            x, w, m, gi, bi, bb, float(alpha), relu, running_mean, running_var, nbt, float(momentum), float(eps),
            training, residual=res, pre=(w is not None and gi is not None),
            out=None if stack is None else stack[0].half('y', stack[1], x))
        ctx.tail = res is not None
        ctx.save_for_backward(x, w, table, m, bb, y if ctx.tail else None, x_in)
        ctx.cfg = (float(alpha), relu, stride, pad, training, None if key is None else tuple(key.shape))
        ctx.set_materialize_grads(False)          # unused outputs arrive as None, not as freshly filled zeros
        # running_mean / running_var / num_batches_tracked are plain buffers updated in place by the kernel
        if gamma is None:
            gamma = beta = x.new_empty(0)             # placeholders (no kernel): the public branch has no gamma
        if loss is None:
            loss = acc = x.new_empty(0)
            bits = torch.empty(0, dtype=torch.int8, device=x.device)
        ctx.mark_non_differentiable(acc, bits)
        return y, y.detach().view_as(y), gamma, beta, loss, acc, bits

    @staticmethod
    def backward(ctx, dy, dy2, dgamma_extra, dbeta_extra, dloss, _dacc, _dbits):
        x, w, table, m, bb, tail_out, x_in = ctx.saved_tensors
        alpha, relu, stride, pad, training, key_shape = ctx.cfg
        if dy is None:
            dy, dy2 = dy2, None
        if dy is None:
            dy = torch.zeros_like(x)
        if dy2 is not None and not ctx.tail:
            n_, c_ = x.shape[0], x.shape[1]
            if not (kernels.bn_resident(n_, c_, x.numel() // (n_ * c_)) & 2):
                dy, dy2 = dy + dy2, None                    # 3-launch form: the two gradients are added here
            # else: the single-pass backward sums them itself (deepipr_passport_bn_bwd, dy2 without tail_out)
        dl = None if (bb is None or dloss is None) else dloss.contiguous()
        if w is None:
            dgamma_extra = dbeta_extra = None
        in_node_conv = x_in is not None
        stack = ctx.stack
        # stacked branches: the rank-2 term of the private branch goes to the stacked convolution's weight gradient
        to_conv = stack is not None and w is not None and not in_node_conv and ctx.needs_input_grad[1]
        # with the convolution inside this node the fresh dW is NOT written: the rank-2 update goes into MIOpen's
        # wgrad below (wshape None = "no dW" for the kernel; dgamma / dbeta still carry the sign-loss gradient)
        out = kernels.passport_bn_bwd(dy.contiguous(), x, table, m, bb, alpha, dl,
                                      _grad_or_none(dgamma_extra), _grad_or_none(dbeta_extra),
                                      None if (w is None or in_node_conv or to_conv) else w.shape, relu, training,
                                      dy2=None if dy2 is None else dy2.contiguous(), tail_out=tail_out,
                                      dx_out=stack[0].half('dx', stack[1], x) if (stack is not None and stack[2]) else None,
                                      dres_out=stack[0].half('dres', stack[1], x) if (stack is not None and ctx.tail) else None)
        dx, dw, dg, db = out[:4]
        dres = out[4] if ctx.tail else None
        if to_conv:
            stack[0].rank2 = (dg, db, m)
        deferred = False
        if in_node_conv:
            # the rank-2 update of several layers in ONE launch: this node hands MIOpen's wgrad and dgamma / dbeta to
            # the group node (_Rank2Group), which runs once all its layers have been here
            deferred = (ctx.defer is not None and ctx.needs_input_grad[1] and ctx.needs_input_grad[4]
                        and ctx.needs_input_grad[5])
            dx, dw, deferred = _conv_bwd_acc(ctx, dx, x_in, w, stride, pad, dg, db, m,
                                             defer=ctx.defer if deferred else None)
        dsk = dk = None
        if w is not None and (ctx.needs_input_grad[2] or ctx.needs_input_grad[3]):
            dsk, dk = kernels.gamma_beta_dkey(dg, db, w, key_shape, stride, pad)
        own = w is None or deferred                       # dgamma / dbeta travel on to whoever produced gamma_in / beta_in
        return (dx, dw, dsk if ctx.needs_input_grad[2] else None, dk if ctx.needs_input_grad[3] else None,
                dg if own else None, db if own else None, None, None, None, None, None, dres, None)


class _BNDualTail(torch.autograd.Function):
    """out = relu(act_a(bn_a(xa)) + act_b(bn_b(xb))): a projection block's convbn_2 and shortcut norm layers
    (BatchNorm2d with learnable weight / bias, each with or without its own ReLU -- the reference builds both with one,
    models/resnet_passport.py:26-30) and the block's tail in ONE launch per direction
    (deepipr_bn_dual_tail_fwd / _bwd; models/resnet_passport.py:67-85).  The output comes out twice, like every
    block output (two consumers, whose gradients the backward kernel sums).  Bit-identical to the two separate fused
    layer calls; if the dual form is not available at backward time (the exchange words were withheld meanwhile) the
    separate backward kernels run on the same saved tensors."""

    @staticmethod
    def forward(ctx, xa, xb, ga, ba, gb, bb, stats_a, stats_b, relus):
        xa, xb = xa.contiguous(), xb.contiguous()
        out, ta, tb = kernels.bn_dual_tail_fwd(xa, xb, ga.contiguous(), ba.contiguous(), gb.contiguous(),
                                               bb.contiguous(), stats_a, stats_b, *relus)
        ctx.relus = relus
        ctx.save_for_backward(xa, xb, ta, tb, out)
        ctx.set_materialize_grads(False)
        return out, out.detach().view_as(out)

    @staticmethod
    def backward(ctx, dy, dy2):
        xa, xb, ta, tb, out = ctx.saved_tensors
        if dy is None:
            dy, dy2 = dy2, None
        if dy is None:
            dy = torch.zeros_like(out)
        dy = dy.contiguous()
        dy2 = None if dy2 is None else dy2.contiguous()
        n, c = xa.shape[0], xa.shape[1]
        if kernels.bn_dual_supported(n, c, xa.numel() // (n * c)):
            dxa, dxb, dga, dba, dgb, dbb = kernels.bn_dual_tail_bwd(dy, dy2, out, xa, xb, ta, tb, *ctx.relus)
        else:
            dxa, _dw, dga, dba, dres = kernels.passport_bn_bwd(dy, xa, ta, None, None, 0.0, None, None, None, None,
                                                               ctx.relus[0], True, dy2=dy2, tail_out=out)
            dxb, _dw, dgb, dbb = kernels.passport_bn_bwd(dres, xb, tb, None, None, 0.0, None, None, None, None,
                                                         ctx.relus[1], True)
        return dxa, dxb, dga, dba, dgb, dbb, None, None, None


def bn_dual_tail_usable(bn_a, bn_b, shape):
    """Both norms are BatchNorm2d(affine) on batch statistics with the default running average and the shape
    [N, C, H, W] takes the dual form in both directions.  DEEPIPR_NO_DUAL_TAIL=1 (or DEEPIPR_TAIL_FUSION=0) keeps the
    two separate fused layer calls."""
    if os.environ.get('DEEPIPR_NO_DUAL_TAIL') == '1' or os.environ.get('DEEPIPR_TAIL_FUSION', '1') == '0':
        return False
    for bn in (bn_a, bn_b):
        if not (isinstance(bn, torch.nn.BatchNorm2d) and bn.affine and bn.momentum is not None and bn.training
                and bn.track_running_stats):
            return False
    n, c, h, w = shape
    return kernels.bn_dual_supported(n, c, h * w)


def bn_dual_tail(xa, xb, bn_a, bn_b, relu_a=True, relu_b=True):
    """-> the pair of handles of relu(act_a(bn_a(xa)) + act_b(bn_b(xb))) (see _BNDualTail)."""
    stats = [(bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.momentum, bn.eps) for bn in (bn_a, bn_b)]
    return _BNDualTail.apply(xa, xb, bn_a.weight, bn_a.bias, bn_b.weight, bn_b.bias, stats[0], stats[1],
                             (bool(relu_a), bool(relu_b)))


def _bn_apply(x, weight, skey, key, gamma_in, beta_in, b, m, bn, alpha, relu, stride, pad, residual, conv=None,
              defer=None, stack=None):
    cfg = (alpha, bool(relu), stride, pad, _bn_uses_batch_stats(bn), bn.momentum, bn.eps, conv, defer, stack)
    return _PassportBNLayer.apply(x, weight, skey, key, gamma_in, beta_in, b, m, bn.running_mean, bn.running_var,
                                  bn.num_batches_tracked if bn.training else None, residual, cfg)


def passport_bn_layer(x, weight, skey, key, b, m, bn, alpha, relu, stride, pad, residual=None, conv_inside=False,
                      pre=None, stack=None):
    """Fused passport branch on the conv output `x`; `bn` is the layer's nn.BatchNorm2d(affine=False).
    conv_inside: `x` is the layer's INPUT; the data convolution runs inside the node and the shared weight's
    gradient is accumulated in place (see _PassportBNLayer).
    pre = (gamma, beta[, share, index]) already computed for this weight by the batched GEMV launch (gamma_beta_batch);
    with share / index the rank-2 weight gradient is left to the layer group's launch (_Rank2Group).
    -> y, gamma, beta, loss, acc, bits; with `residual` y is the PAIR of handles of relu(layer + residual)."""
    pg, pb = pre[:2] if pre is not None else (None, None)
    defer = tuple(pre[2:4]) if (pre is not None and len(pre) >= 4 and conv_inside) else None
    y, y2, gamma, beta, loss, acc, bits = _bn_apply(x, weight, skey, key, pg, pb, b, m, bn, alpha, relu, stride,
                                                    pad, residual, (stride, pad) if conv_inside else None, defer, stack)
    return ((y, y2) if residual is not None else y), gamma, beta, loss, acc, bits


_WINO_CONVS = weakref.WeakKeyDictionary()      # model -> (module count, its 3x3 stride-1 convolutions that have a Winograd image)
# {weight address: (weight shape, Uf, Ud)} while a net's forward pass is under wino_weights() -- per THREAD (threaded
# forwards of replicas, nn.DataParallel style, must not see each other's tables)
_WINO_TLS = threading.local()


def _wino_pre(w):
    """(Uf, Ud) of this weight for the current forward pass, or None."""
    table = getattr(_WINO_TLS, 'table', None)
    if table is None:
        return None
    hit = table.get(w.data_ptr())
    return hit[1:] if hit is not None and hit[0] == tuple(w.shape) else None


class wino_weights:
    """Context manager for a net's forward pass: the Winograd images U = G g G^T of ALL its 3x3 stride-1 convolutions' weights
    in ONE launch up front (kernels.wino_transform -> deepipr_conv_wino_transform_multi), so that the forward and -- through
    the autograd nodes, which keep the pair -- the backward-data kernels of this step copy them global -> LDS instead of
    transforming the same filters in every workgroup.  The weights must not change between the forward pass
    and its backward pass (autograd's own rule for saved tensors); the images are rewritten by the next forward pass.  A
    replayed hipGraph contains the transform launch like any other kernel of the step.
    DEEPIPR_WINO_PRE=0 switches it off (every workgroup transforms its filters itself: bit-identical results)."""

    def __init__(self, model, x):
        self.model = model
        self.on = torch.is_tensor(x) and bool(x.is_cuda) and WINO_PRE and kernels.conv_algo_is_winograd()
        self.max_filters = 0
        if self.on:
            self.max_filters = WINO_PRE_MAX_FILTERS if x.shape[0] >= WINO_PRE_MIN_BATCH else min(WINO_PRE_MAX_FILTERS, WINO_PRE_SMALL_FILTERS)
            self.on = self.max_filters > 0

    def _convs(self):
        # the list is kept per model and rebuilt when the model's module set changed (model surgery after a first forward:
        # a replaced convolution must get its image, a removed one must not be transformed every step)
        mods = list(self.model.modules())
        key = (len(mods), sum(id(m) for m in mods))
        hit = _WINO_CONVS.get(self.model)
        if hit is None or hit[0] != key:
            hit = _WINO_CONVS[self.model] = (key, [
                m for m in mods
                if type(m) is torch.nn.Conv2d and m.kernel_size == (3, 3) and m.stride == (1, 1) and m.padding == (1, 1)
                and m.groups == 1 and m.dilation == (1, 1) and m.bias is None and m.padding_mode == 'zeros'
                and m.in_channels % 32 == 0 and m.out_channels % 32 == 0])
        return hit[1]

    def __enter__(self):
        self.before = getattr(_WINO_TLS, 'table', None)
        if not self.on or self.before is not None:          # (a nested forward keeps the outer table)
            return self
        ws = [m.weight for m in self._convs() if m.weight.is_cuda and m.weight.dtype == torch.float32 and m.weight.is_contiguous()
              and self.max_filters >= m.weight.shape[0] * m.weight.shape[1]]
        if ws:
            backward = torch.is_grad_enabled()           # (the backward-data images only when a backward pass can follow)
            with torch.no_grad():
                imgs = kernels.wino_transform(ws, backward=backward)
            _WINO_TLS.table = {w.data_ptr(): (tuple(w.shape),) + pair for w, pair in zip(ws, imgs)}
        return self

    def __exit__(self, *exc):
        _WINO_TLS.table = self.before
        return False


def with_wino_weights(forward):
    """Decorator for a net's forward(self, x, ...) / forward_dual: the whole pass under wino_weights(self, x); x may be
    passed by keyword."""
    @functools.wraps(forward)
    def wrapped(self, *args, **kwargs):
        with wino_weights(self, args[0] if args else kwargs.get('x')):
            return forward(self, *args, **kwargs)
    return wrapped


class gamma_beta_batch:
    """Context manager for a net's forward: gamma / beta of ALL its passport layers that will take the fused BatchNorm
    form in ONE launch (deepipr_gamma_beta_fwd_multi) instead of one GEMV launch per layer -- ResNet18: the five
    layer4 weights, 33.6 MB streamed once by a launch that fills the chip.  They depend only on the weights and the
    cached pooled keys, so they are computed up front; each layer picks its pair up (PassportLayerBase._gb_pre) and its
    autograd node treats it as its own GEMV's result.  Backward: the layers of one GROUP (`group_of(layer)`: the
    backward stage the layer belongs to, models' backward_stages(); one group without it) leave the rank-2 update of
    the shared weights' gradients -- and handing those gradients to autograd -- to one launch per group (_Rank2Group);
    a layer alone in its group, or whose convolution runs outside the fused node, does its own as before.  Pairs nobody
    used are dropped on exit.
    DEEPIPR_NO_GEMV_BATCH=1 switches the batching off, DEEPIPR_NO_RANK2_BATCH=1 only the backward part."""

    def __init__(self, layers, force_passport=False, ind=0, group_of=None):
        self.layers, self.args, self.used, self.group_of = layers, (force_passport, ind), [], group_of

    def __enter__(self):
        if os.environ.get('DEEPIPR_NO_GEMV_BATCH') == '1':
            return self
        reqs = []
        for m in self.layers:
            r = m.batched_gamma_beta_request(*self.args)
            if r is not None:
                reqs.append((m, r))
        grouped = torch.is_grad_enabled() and os.environ.get('DEEPIPR_NO_RANK2_BATCH') != '1'
        for lo in range(0, len(reqs), _lib.GEMV_MAX_LAYERS):
            chunk = reqs[lo:lo + _lib.GEMV_MAX_LAYERS]
            if len(chunk) < 2:
                break                                  # a single layer: its own fused call does the same work
            with torch.no_grad():
                pairs = kernels.gamma_beta_fwd_multi([w.detach().contiguous() for _, (w, _m) in chunk],
                                                     [mm for _, (_w, mm) in chunk])
            groups = {}
            for idx, (m, (w, _mm)) in enumerate(chunk):
                if grouped and w.requires_grad:
                    groups.setdefault(self.group_of(m) if self.group_of is not None else 0, []).append(idx)
            pre = {idx: pairs[idx] for idx in range(len(chunk))}
            for members in groups.values():
                if len(members) < 2:
                    continue
                share = _Rank2Share()
                outs = _Rank2Group.apply(share, [chunk[i][1][1] for i in members], [pairs[i] for i in members],
                                         *[chunk[i][1][0] for i in members])
                for j, i in enumerate(members):
                    pre[i] = (outs[2 * j], outs[2 * j + 1], share, j)
            for idx, (m, _r) in enumerate(chunk):
                m._gb_pre = pre[idx]
                self.used.append(m)
        return self

    def __exit__(self, *exc):
        for m in self.used:
            m._gb_pre = None
        return False


_STAGE_TABLES = weakref.WeakKeyDictionary()      # model -> {submodule: backward stage}; nothing lands in the model's state


def stage_groups(model):
    """layer -> index of the backward stage (model.backward_stages()) it belongs to: the `group_of` of
    gamma_beta_batch.  Cached per model, keyed on the module objects themselves (weakly): a layer swapped in after the
    first forward (fine-tune / attack scripts) misses the table and has it rebuilt rather than landing in stage -1, and
    a recycled id() cannot file a layer under another layer's stage; a deep copy of the model starts
    without a table."""
    table = _STAGE_TABLES.get(model)

    def build():
        fresh = weakref.WeakKeyDictionary()
        for k, (_cut, mods) in enumerate(model.backward_stages()):
            for mod in mods:
                for sub in mod.modules():
                    fresh.setdefault(sub, k)
        _STAGE_TABLES[model] = fresh
        return fresh

    if table is None:
        table = build()
    state = {'table': table, 'rebuilt': False}

    def group_of(layer):
        k = state['table'].get(layer)
        if k is None and not state['rebuilt']:
            state['table'], state['rebuilt'] = build(), True
            k = state['table'].get(layer)
        return -1 if k is None else k
    return group_of


def bn_affine_relu(x, gamma, beta, bn, relu=True, residual=None, fork=False, stack=None):
    """Fused public branch: BatchNorm2d(affine=False) + learnable gamma/beta + ReLU; with `residual` the pair of
    handles of relu(that + residual).  fork=True hands the output out as a pair of handles as well (a layer whose
    output has two consumers without being a block's tail -- the CIFAR stem): the consumers' gradients then reach the
    backward kernel separately and are summed there instead of by an ATen add pass."""
    out = _bn_apply(x, None, None, None, gamma, beta, None, None, bn, 0.0, relu, 1, 0, residual, stack=stack)
    return (out[0], out[1]) if (residual is not None or fork) else out[0]


def conv_out_shape(x, conv):
    """Shape of conv(x) for a plain square-geometry nn.Conv2d (no dilation / groups)."""
    k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
    return (x.shape[0], conv.out_channels, (x.shape[2] + 2 * p - k) // s + 1, (x.shape[3] + 2 * p - conv.kernel_size[1]) // s + 1)


def bn_tail_fusable(bn, x):
    """The residual tail can be folded into this layer's kernels: a BatchNorm2d on batch statistics and a shape
    that takes the single-pass form in both directions.  `x`: the conv output, or its shape."""
    # DEEPIPR_TAIL_FUSION=0 keeps the separate tail kernels (deepipr_add_relu_fwd / deepipr_relu_bwd2); both forms are
    # bit-identical (tests/test_parity_gpu.py::test_tail_fusion_is_bit_identical_at_model_level).
    if os.environ.get('DEEPIPR_TAIL_FUSION', '1') == '0':
        return False
    if not isinstance(bn, torch.nn.BatchNorm2d) or bn.momentum is None or not _bn_uses_batch_stats(bn):
        return False
    if not isinstance(x, torch.Tensor):
        n, c, h, w = x
        return kernels.bn_resident(n, c, h * w) == 3
    if x.dim() != 4 or x.dtype != torch.float32:
        return False
    n, c = x.shape[0], x.shape[1]
    return kernels.bn_resident(n, c, x.numel() // (n * c)) == 3


def _bn_uses_batch_stats(bn):
    return bn.training or bn.running_mean is None


def bn_is_fusable(bn):
    """nn.BatchNorm2d without affine parameters and with the default exponential running average."""
    return (isinstance(bn, torch.nn.BatchNorm2d) and not bn.affine and bn.momentum is not None)


class _PassportGNLayer(torch.autograd.Function):
    """GroupNorm / InstanceNorm (affine=False) + passport affine + ReLU + sign loss, fused (deepipr_passport_gn_*):
    one register-resident kernel per direction (+ the gamma/beta GEMV and the dgamma/dbeta/dW finish).  With
    weight=None it is the W-less branch: learnable gamma_in / beta_in (public branch, or a ConvBlock's affine
    norm), either of which may be None (InstanceNorm2d without affine)."""

    @staticmethod
    def forward(ctx, x, weight, skey, key, gamma_in, beta_in, b, m, cfg):
        alpha, relu, stride, pad, groups, eps, conv_inside = cfg
        x = x.contiguous()
        w = None if weight is None else weight.contiguous()
        x_in = None
        if conv_inside:                               # the data convolution runs inside this node (see _PassportBNLayer)
            x_in = x
            x = _conv_fwd(x_in, w, stride, pad, ctx)
        gi = None if gamma_in is None else gamma_in.contiguous().view(-1)
        bi = None if beta_in is None else beta_in.contiguous().view(-1)
        bb = None if b is None else b.contiguous().view(-1)
        y, stats, gamma, beta, loss, acc, bits = kernels.passport_gn_fwd(x, w, m, gi, bi, bb, float(alpha), relu,
                                                                         int(groups), float(eps))
        used_g, used_b = (gamma, beta) if w is not None else (gi, bi)
        ctx.save_for_backward(x, w, stats, used_g, used_b, m, bb, x_in)
        ctx.cfg = (float(alpha), relu, stride, pad, int(groups), None if key is None else tuple(key.shape))
        ctx.set_materialize_grads(False)
        if gamma is None:
            gamma = beta = x.new_empty(0)
        if loss is None:
            loss = acc = x.new_empty(0)
            bits = torch.empty(0, dtype=torch.int8, device=x.device)
        ctx.mark_non_differentiable(acc, bits)
        return y, gamma, beta, loss, acc, bits

    @staticmethod
    def backward(ctx, dy, dgamma_extra, dbeta_extra, dloss, _dacc, _dbits):
        x, w, stats, g, bt, m, bb, x_in = ctx.saved_tensors
        alpha, relu, stride, pad, groups, key_shape = ctx.cfg
        if dy is None:
            dy = torch.zeros_like(x)
        dl = None if (bb is None or dloss is None) else dloss.contiguous()
        if w is None:
            dgamma_extra = dbeta_extra = None
        dx, dw, dg, db = kernels.passport_gn_bwd(dy.contiguous(), x, stats, g, bt, m, bb, alpha, dl,
                                                 _grad_or_none(dgamma_extra), _grad_or_none(dbeta_extra),
                                                 None if (w is None or x_in is not None) else w.shape, relu, groups)
        if x_in is not None:
            dx, dw, _ = _conv_bwd_acc(ctx, dx, x_in, w, stride, pad, dg, db, m)
        dsk = dk = None
        if w is not None and (ctx.needs_input_grad[2] or ctx.needs_input_grad[3]):
            dsk, dk = kernels.gamma_beta_dkey(dg, db, w, key_shape, stride, pad)
        return (dx, dw, dsk if ctx.needs_input_grad[2] else None, dk if ctx.needs_input_grad[3] else None,
                dg if (w is None and ctx.needs_input_grad[4]) else None,
                db if (w is None and ctx.needs_input_grad[5]) else None, None, None, None)


def norm_groups(norm):
    """Groups of a norm that deepipr_passport_gn_* can fold in: nn.GroupNorm -> num_groups; nn.InstanceNorm2d on
    instance statistics -> one group per channel; anything else -> 0."""
    if isinstance(norm, torch.nn.GroupNorm):
        return norm.num_groups
    if isinstance(norm, torch.nn.InstanceNorm2d) and not norm.track_running_stats:
        return norm.num_features
    return 0


def gn_is_fusable(norm, x):
    """GroupNorm / InstanceNorm2d whose (sample, group) chunk of x fits the register-resident kernels."""
    groups = norm_groups(norm)
    if not groups or x.dim() != 4 or x.dtype != torch.float32:
        return False
    n, c = x.shape[0], x.shape[1]
    return kernels.gn_supported(n, c, x.numel() // (n * c), groups)


def gn_supported_shape(norm, shape):
    """gn_is_fusable for a conv output that does not exist yet: `shape` = (N, C, H, W)."""
    groups = norm_groups(norm)
    n, c, h, w = shape
    return bool(groups) and kernels.gn_supported(n, c, h * w, groups)


def passport_gn_layer(x, weight, skey, key, b, m, norm, alpha, relu, stride, pad, conv_inside=False):
    """Fused passport branch on the conv output `x` (conv_inside: on the layer's input, the data conv runs inside
    the node); `norm` is the layer's GroupNorm / InstanceNorm2d (affine=False)."""
    cfg = (alpha, bool(relu), stride, pad, norm_groups(norm), norm.eps, bool(conv_inside))
    return _PassportGNLayer.apply(x, weight, skey, key, None, None, b, m, cfg)


def gn_affine_relu(x, gamma, beta, norm, relu=True):
    """Fused W-less branch: GroupNorm / InstanceNorm2d + per-channel gamma/beta (None = 1 / 0) + ReLU."""
    cfg = (0.0, bool(relu), 1, 0, norm_groups(norm), norm.eps, False)
    return _PassportGNLayer.apply(x, None, None, None, gamma, beta, None, None, cfg)[0]


class _MaxPool3x3s2(torch.autograd.Function):
    """nn.MaxPool2d(3, 2, 1) of the ImageNet stem (models/resnet_passport.py:94-98): one pass forward, a one-byte argmax, a
    gather backward in ATen's order (bit-identical to F.max_pool2d both ways)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        y, slot = kernels.maxpool3x3s2_fwd(x)
        ctx.save_for_backward(slot)
        ctx.x_shape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        slot, = ctx.saved_tensors
        return kernels.maxpool3x3s2_bwd(dy.contiguous(), slot, ctx.x_shape)


OWN_POOL = os.environ.get('DEEPIPR_OWN_POOL', '1') != '0'      # 0: every max-pool through the library (A/B)


class _MaxPool2x2s2(torch.autograd.Function):
    """nn.MaxPool2d(2, 2) of the CIFAR AlexNet (models/alexnet_passport.py:30-38 of the reference): a one-byte argmax, the
    backward a select (bit-identical to F.max_pool2d both ways)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        y, slot = kernels.maxpool2x2s2_fwd(x)
        ctx.save_for_backward(slot)
        ctx.x_shape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        slot, = ctx.saved_tensors
        return kernels.maxpool2x2s2_bwd(dy.contiguous(), slot, ctx.x_shape)


def max_pool(pool, x):
    """pool(x) for an nn.MaxPool2d: this library's kernels for the ImageNet stem's 3x3 / 2 / pad 1 pool and the CIFAR AlexNet's
    2x2 / 2 pools of a float32 CUDA map nobody hooked, the module call otherwise."""
    def two(v):
        return (v, v) if isinstance(v, int) else tuple(v)
    if (OWN_POOL and type(pool) is torch.nn.MaxPool2d and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and two(pool.stride if pool.stride is not None else pool.kernel_size) == (2, 2)
            and two(pool.dilation) == (1, 1) and not pool.ceil_mode and not pool.return_indices
            and not (pool._forward_hooks or pool._forward_pre_hooks or pool._backward_hooks or pool._backward_pre_hooks)
            and not (_GLOBAL_FWD_HOOKS or _GLOBAL_FWD_PRE_HOOKS or _GLOBAL_BWD_HOOKS or _GLOBAL_BWD_PRE_HOOKS)
            and not torch.is_autocast_enabled() and 'forward' not in pool.__dict__):
        if two(pool.kernel_size) == (3, 3) and two(pool.padding) == (1, 1):
            return _MaxPool3x3s2.apply(x)
        # the CIFAR AlexNet's 2x2 / 2 pools: even maps whose width is a multiple of 4 (32 / 16 / 8 wide)
        if (two(pool.kernel_size) == (2, 2) and two(pool.padding) == (0, 0) and x.shape[2] % 2 == 0 and x.shape[3] % 4 == 0
                and x.numel() > 0):
            return _MaxPool2x2s2.apply(x)
    return pool(x)


class _CrossEntropyTop1(torch.autograd.Function):
    """Mean cross-entropy and top-1 accuracy (percent) of logits [N, C] in one launch, backward in one."""

    @staticmethod
    def forward(ctx, logits, target):
        logits, target = logits.contiguous(), target.contiguous()
        loss, top1, lse = kernels.ce_top1_fwd(logits, target)
        ctx.save_for_backward(logits, target, lse)
        ctx.mark_non_differentiable(top1)
        ctx.set_materialize_grads(False)
        return loss, top1

    @staticmethod
    def backward(ctx, dloss, _dtop1):
        if dloss is None:
            return None, None
        logits, target, lse = ctx.saved_tensors
        return kernels.ce_bwd(dloss.contiguous(), logits, target, lse), None


class _ScalarSums(torch.autograd.Function):
    """(a, b, a + b) with a / b the left-to-right sums of two groups of scalar tensors; backward hands every term the gradient
    of the sums it is part of (no kernel when only a + b is differentiated: the usual case, the step's objective)."""

    @staticmethod
    def forward(ctx, n_a, *terms):
        ctx.n_a, ctx.n, ctx.shapes = n_a, len(terms), [t.shape for t in terms]
        ctx.set_materialize_grads(False)
        out = kernels.scalar_sums([t.detach().reshape(1) for t in terms], n_a)
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, ga, gb, gt):
        def both(x, y):
            return y if x is None else (x if y is None else x + y)
        da, db = both(ga, gt), both(gb, gt)

        def shaped(g, shape):                            # (a loss kept as a one-element vector gets its gradient in that shape)
            return None if g is None else g.reshape(shape)
        return (None,) + tuple(shaped(da if i < ctx.n_a else db, ctx.shapes[i]) for i in range(ctx.n))


def scalar_sums(a_terms, b_terms):
    """(sum(a_terms), sum(b_terms), their sum) of scalar tensors, summed left to right like the reference's loops
    (`loss_public + loss_private`, `sign_loss += m.loss`, `loss + sign_loss`: experiments/trainer.py:140-145,
    trainer_private.py:163-173).  CUDA fp32 scalars: one launch; anything else (host tensors in the CPU tests of the host logic):
    the same chain of torch adds."""
    terms = list(a_terms) + list(b_terms)
    if (len(terms) >= 2 and len(terms) <= 48
            and all(isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.numel() == 1 for t in terms)):
        a, b, tot = _ScalarSums.apply(len(a_terms), *terms)
        return (a if a_terms else None), (b if b_terms else None), tot

    def chain(ts):
        total = None
        for t in ts:
            total = t if total is None else total + t
        return total
    a, b = chain(a_terms), chain(b_terms)
    return a, b, (a if b is None else (b if a is None else a + b))


class _PooledLinear(torch.autograd.Function):
    """logits = Linear(avg_pool_to_1x1(x).flatten(1)) in one launch, backward in one (deepipr_pooled_linear_*)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x, weight = x.contiguous(), weight.contiguous()
        logits, pooled = kernels.pooled_linear_fwd(x, weight, None if bias is None else bias.contiguous())
        ctx.save_for_backward(weight, pooled)
        ctx.x_shape, ctx.with_bias = tuple(x.shape), bias is not None
        ctx.set_materialize_grads(False)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        if dlogits is None:
            return None, None, None
        weight, pooled = ctx.saved_tensors
        dx, dw, db = kernels.pooled_linear_bwd(dlogits.contiguous(), weight, pooled, ctx.x_shape, ctx.with_bias)
        return dx, dw, db


OWN_HEAD = os.environ.get('DEEPIPR_OWN_HEAD', '1') != '0'


def pooled_linear(linear, x):
    """linear(adaptive_avg_pool2d(x, 1).flatten(1)) -- the classifier of the reference's ResNets (models/resnet_passport.py:127-129
    there: F.avg_pool2d(out, 4) on the 4x4 map, view, self.linear).  A plain nn.Linear nobody hooked, outside autocast, on a shape
    the kernel takes (at most 128 classes, HW a multiple of 4: the CIFAR-geometry heads) goes through one launch per direction;
    anything else -- the 1000-class ImageNet heads among them -- through the library ops.  DEEPIPR_OWN_HEAD=0: always the library."""
    if (OWN_HEAD and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and type(linear) is torch.nn.Linear
            and 'forward' not in linear.__dict__ and linear.weight.dtype == torch.float32
            and not (linear._forward_hooks or linear._forward_pre_hooks or linear._backward_hooks or linear._backward_pre_hooks)
            and not (_GLOBAL_FWD_HOOKS or _GLOBAL_FWD_PRE_HOOKS or _GLOBAL_BWD_HOOKS or _GLOBAL_BWD_PRE_HOOKS)
            and not torch.is_autocast_enabled() and linear.in_features == x.shape[1]
            and kernels.pooled_linear_supported(x.shape[0], x.shape[1], x.shape[2] * x.shape[3], linear.out_features)):
        return _PooledLinear.apply(x, linear.weight, linear.bias)
    out = torch.nn.functional.adaptive_avg_pool2d(x, (1, 1))
    return linear(out.view(out.size(0), -1))



def cross_entropy_top1(pred, target):
    """-> (F.cross_entropy(pred, target), top-1 accuracy in percent) -- experiments/trainer.py:136,149.  On the GPU
    both come from one fused launch (and one for the backward); anything the kernel does not take (other dtypes,
    more than 2^20 logits) goes through the library ops."""
    if kernels.ce_usable(pred, target):
        return _CrossEntropyTop1.apply(pred, target)
    loss = torch.nn.functional.cross_entropy(pred, target)
    with torch.no_grad():
        top1 = pred.argmax(dim=1).eq(target).float().sum() * (100.0 / target.size(0))
    return loss, top1


class _AddReLU(torch.autograd.Function):
    """out = relu(a + b), the residual tail of a block, in one pass; the backward mask is read from `out`."""

    @staticmethod
    def forward(ctx, a, b):
        out = kernels.add_relu_fwd(a.contiguous(), b.contiguous())
        ctx.save_for_backward(out)
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return None, None
        (out,) = ctx.saved_tensors
        d = kernels.relu_bwd(dy.contiguous(), out)
        return d, d


def _add_relu_fusable(a, b):
    return (a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.shape == b.shape
            and a.numel() >= ADD_RELU_MIN_ELEMENTS)


def add_relu(a, b):
    """relu(a + b) through the fused kernel for same-shape fp32 GPU tensors, the library ops otherwise."""
    if _add_relu_fusable(a, b):
        return _AddReLU.apply(a, b)
    return torch.relu(a + b)


class _AddReLUFork(torch.autograd.Function):
    """relu(a + b) handed out twice -- once for each consumer of a residual block's output (the next block's first
    conv and its identity / projection shortcut).  Backward therefore receives the two gradients separately and
    forms (g1 + g2) * [out > 0] in one pass (deepipr_relu_bwd2) instead of autograd's add kernel + the mask pass."""

    @staticmethod
    def forward(ctx, a, b):
        out = kernels.add_relu_fwd(a.contiguous(), b.contiguous())
        ctx.save_for_backward(out)
        ctx.set_materialize_grads(False)
        return out, out.detach().view_as(out)

    @staticmethod
    def backward(ctx, g1, g2):
        if g1 is None and g2 is None:
            return None, None
        (out,) = ctx.saved_tensors
        if g1 is None:
            g1, g2 = g2, None
        d = kernels.relu_bwd(g1.contiguous(), out, None if g2 is None else g2.contiguous())
        return d, d


def add_relu_fork(a, b):
    """-> (out, out'): the same values, to be used by the two consumers of a residual block's output."""
    if _add_relu_fusable(a, b):
        return _AddReLUFork.apply(a, b)
    out = torch.relu(a + b)
    # two distinct handles also here: each collects its own consumers' gradients and the two sums meet in ONE add,
    # whatever the number of consumers (the shared trunk of a V2 / V3 dual forward gives each handle two) -- the same
    # association as the fused kernels' dy + dy2 and as the staged backward's per-handle leaves
    return out.view_as(out), out.view_as(out)


ADD_RELU_MIN_ELEMENTS = 1 << 18


def affine_relu(xhat, gamma, beta, relu=True):
    return _AffineReLU.apply(xhat, gamma.reshape(-1), beta.reshape(-1), bool(relu))


def gamma_beta(weight, skey, key, m, stride, pad):
    return _GammaBeta.apply(weight, skey, key, m, stride, pad)


def sign_loss(gamma, b, alpha, l2=L2):
    """-> (loss, acc, bits): loss = sum(alpha*relu(-b*gamma+0.1)) + l2*sum(gamma^2)."""
    return _SignLoss.apply(gamma, b, alpha, l2)


def passport_layer(xhat, weight, skey, key, b, m, alpha, relu, stride, pad, conv_inside=False):
    """conv_inside (only when NO norm sits between conv and affine): `xhat` is the layer's input, the data conv runs
    inside the node and the shared weight's gradient is accumulated in place."""
    return _PassportLayer.apply(xhat, weight, skey, key, b, m, alpha, bool(relu), stride, pad, bool(conv_inside))
