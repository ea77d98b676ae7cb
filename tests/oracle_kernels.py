"""A stand-in for deepipr_amd.passport_ops.kernels built from oracle/np_passport.py.

TEST-ONLY: lets the CPU suite exercise the *host* logic of the product (autograd wiring, module API,
trainers, DDP) without a GPU by monkeypatching `passport_ops.kernels`.  The product itself never
imports this; on a GPU box the real HIP kernels run (tests/test_parity_gpu.py).
"""
import numpy as np
import torch

from oracle import np_passport as npp


def _n(t):
    return t.detach().cpu().numpy().astype(np.float64)


def _t(a, like, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).to(like.device)


class OracleKernels:
    def pooled_patch_mean(self, keys, kh, kw, stride, pad):
        out = []
        for j in range(keys.shape[0]):
            s, n = npp.pooled_patch_sum(_n(keys[j]), kh, kw, stride, pad)
            out.append(s / n)
        return _t(np.stack(out), keys, torch.float64)

    def gamma_beta_fwd(self, weight, m):
        w = _n(weight).reshape(weight.shape[0], -1)
        mm = _n(m)
        return _t(w @ mm[0], weight), _t(w @ mm[1], weight)

    def gamma_beta_bwd(self, dgamma, dbeta, m, wshape):
        mm = _n(m)
        dw = np.outer(_n(dgamma), mm[0]) + np.outer(_n(dbeta), mm[1])
        return _t(dw.reshape(tuple(wshape)), dgamma)

    def gamma_beta_dkey(self, dgamma, dbeta, weight, key_shape, stride, pad):
        z = np.zeros(tuple(key_shape))
        _, dsk, dk = npp.gamma_beta_bwd(_n(dgamma), _n(dbeta), _n(weight), z, z, stride, pad, need_dkey=True)
        return _t(dsk, weight), _t(dk, weight)

    def affine_relu_fwd(self, xhat, gamma, beta, relu):
        y = npp.affine_relu_fwd(xhat.detach().numpy(), gamma.detach().numpy().reshape(-1),
                                beta.detach().numpy().reshape(-1), relu)     # f32, like the kernel
        return _t(y, xhat)

    def affine_relu_bwd(self, dy, xhat, gamma, beta, relu):
        g32, b32 = gamma.detach().numpy().reshape(-1), beta.detach().numpy().reshape(-1)
        mask_src = npp.affine_relu_fwd(xhat.detach().numpy(), g32, b32, False)
        dyn = _n(dy)
        dz = np.where(mask_src > 0, dyn, 0.0) if relu else dyn
        dx = dz * _n(gamma).reshape(1, -1, 1, 1)
        return (_t(dx, xhat), _t((dz * _n(xhat)).sum(axis=(0, 2, 3)), xhat), _t(dz.sum(axis=(0, 2, 3)), xhat))

    def sign_loss_fwd(self, gamma, b, alpha, margin=npp.MARGIN, l2=npp.L2):
        g, bb = _n(gamma).reshape(-1), _n(b).reshape(-1)
        z32 = (-b.detach().numpy().reshape(-1) * gamma.detach().numpy().reshape(-1) + np.float32(margin))
        loss = (alpha * np.maximum(z32.astype(np.float64), 0)).sum() + l2 * (g ** 2).sum()
        acc = (np.sign(bb) == np.sign(g)).mean()
        return (_t(np.float64(loss), gamma).reshape(()), _t(np.float64(acc), gamma).reshape(()),
                _t(np.sign(g), gamma, torch.int8))

    def sign_loss_bwd(self, dloss, gamma, b, alpha, margin=npp.MARGIN, l2=npp.L2):
        z32 = (-b.detach().numpy().reshape(-1) * gamma.detach().numpy().reshape(-1) + np.float32(margin))
        g = np.where(z32 > 0, -alpha * _n(b).reshape(-1), 0.0) + 2 * l2 * _n(gamma).reshape(-1)
        return _t(float(dloss) * g, gamma)

    def passport_fwd(self, xhat, weight, m, b, alpha, relu, margin=npp.MARGIN, l2=npp.L2):
        gamma, beta = self.gamma_beta_fwd(weight, m)
        y = self.affine_relu_fwd(xhat, gamma, beta, relu)
        if b is None:
            return y, gamma, beta, None, None, None
        loss, acc, bits = self.sign_loss_fwd(gamma, b, alpha, margin, l2)
        return y, gamma, beta, loss, acc, bits

    def passport_bwd(self, dy, xhat, gamma, beta, m, b, alpha, dloss, dgamma_extra, dbeta_extra, wshape, relu,
                     margin=npp.MARGIN, l2=npp.L2):
        dx, dg, db = self.affine_relu_bwd(dy, xhat, gamma, beta, relu)
        if dgamma_extra is not None:
            dg = dg + dgamma_extra
        if dbeta_extra is not None:
            db = db + dbeta_extra
        if dloss is not None:
            dg = dg + self.sign_loss_bwd(dloss, gamma, b, alpha, margin, l2)
        return dx, (self.gamma_beta_bwd(dg, db, m, wshape) if wshape is not None else None), dg, db

    # ---- BatchNorm-fused entry points: stock ATen batch_norm on the host as the checker ----
    allow_sync = True
    sync_launches = 0

    def withhold_sync(self, owner):
        self.allow_sync = False

    def release_sync(self, owner):
        self.allow_sync = True

    def set_user_sync(self, on):
        self.allow_sync = bool(on)

    def sync_scope(self, allowed):
        import contextlib
        return contextlib.nullcontext()

    def bn_resident(self, n, c, hw):
        return 3                      # the host logic is exercised as if every shape took the single-pass form

    def passport_bn_fwd(self, x, weight, m, gamma_in, beta_in, b, alpha, relu, running_mean, running_var, nbt,
                        momentum, eps, training, margin=npp.MARGIN, l2=npp.L2, residual=None, pre=False, out=None):
        x64 = x.detach().double()
        if training:
            mean = x64.mean(dim=(0, 2, 3))
            var = x64.var(dim=(0, 2, 3), unbiased=False)
            if running_mean is not None:
                mm = x.numel() // x.shape[1]
                with torch.no_grad():
                    running_mean.mul_(1 - momentum).add_(momentum * mean.float())
                    running_var.mul_(1 - momentum).add_(momentum * (var * mm / max(1, mm - 1)).float())
                    if nbt is not None:
                        nbt.add_(1)
        else:
            mean, var = running_mean.double(), running_var.double()
        invstd = 1.0 / torch.sqrt(var + eps)
        if weight is not None and not pre:
            gamma, beta = self.gamma_beta_fwd(weight, m)
        else:                         # learnable gamma / beta, or the passport pair precomputed by the batched GEMV
            gamma, beta = gamma_in.detach().reshape(-1), beta_in.detach().reshape(-1)
        xh = ((x64 - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)).float()
        y = self.affine_relu_fwd(xh, gamma, beta, relu)
        if residual is not None:      # fused tail of a residual block
            y = torch.relu(y + residual.detach())
        table = torch.zeros(x.shape[1], 8)
        table[:, 0], table[:, 1], table[:, 2], table[:, 3] = mean.float(), invstd.float(), gamma, beta
        if b is None:
            y = y if out is None else out.copy_(y)
            return y, table, (gamma if weight is not None else None), (beta if weight is not None else None), None, None, None
        loss, acc, bits = self.sign_loss_fwd(gamma, b, alpha, margin, l2)
        y = y if out is None else out.copy_(y)
        return y, table, gamma, beta, loss, acc, bits

    def passport_bn_bwd(self, dy, x, table, m, b, alpha, dloss, dgamma_extra, dbeta_extra, wshape, relu, training,
                        margin=npp.MARGIN, l2=npp.L2, dy2=None, tail_out=None, dx_out=None, dres_out=None):
        mean, invstd, gamma, beta = [table[:, i].double() for i in range(4)]
        dres = None
        if tail_out is not None:
            d = dy.detach() if dy2 is None else dy.detach() + dy2.detach()
            dres = torch.where(tail_out > 0, d, torch.zeros_like(d))
            dy = dres
        elif dy2 is not None:                 # two consumers, no tail: the gradients are summed by the kernel
            dy = dy.detach() + dy2.detach()
        x64, dy64 = x.detach().double(), dy.detach().double()
        xh = (x64 - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)
        z32 = npp.affine_relu_fwd(xh.float().numpy(), table[:, 2].numpy(), table[:, 3].numpy(), False)
        dz = torch.where(torch.from_numpy(z32 > 0), dy64, torch.zeros_like(dy64)) if relu else dy64
        s_dzx, s_dz = (dz * xh).sum(dim=(0, 2, 3)), dz.sum(dim=(0, 2, 3))
        mm = x.numel() // x.shape[1]
        c2, c3 = (s_dz / mm, s_dzx / mm) if training else (torch.zeros_like(s_dz), torch.zeros_like(s_dz))
        dx = (gamma * invstd).view(1, -1, 1, 1) * (dz - c2.view(1, -1, 1, 1) - xh * c3.view(1, -1, 1, 1))
        dg, db = s_dzx.float(), s_dz.float()
        if dgamma_extra is not None:
            dg = dg + dgamma_extra
        if dbeta_extra is not None:
            db = db + dbeta_extra
        if dloss is not None:
            dg = dg + self.sign_loss_bwd(dloss, table[:, 2].contiguous(), b, alpha, margin, l2)
        dw = self.gamma_beta_bwd(dg, db, m, wshape) if wshape is not None else None
        if tail_out is not None:
            dx = dx.float() if dx_out is None else dx_out.copy_(dx)
            dres = dres if dres_out is None else dres_out.copy_(dres)
            return dx, dw, dg, db, dres
        return (dx.float() if dx_out is None else dx_out.copy_(dx)), dw, dg, db

    # ---- GroupNorm / InstanceNorm-fused entry points: float64 group statistics on the host as the checker ----
    def gn_supported(self, n, c, hw, groups):
        return groups > 0 and c % groups == 0 and hw % 4 == 0 and (c // groups) * hw <= 24576

    @staticmethod
    def _gn_xhat(x, groups, stats):
        n, c = x.shape[0], x.shape[1]
        x64 = x.detach().double().reshape(n, groups, -1)
        mean, invstd = stats[:, 0].double().view(n, groups, 1), stats[:, 1].double().view(n, groups, 1)
        return ((x64 - mean) * invstd).reshape(x.shape)

    def passport_gn_fwd(self, x, weight, m, gamma_in, beta_in, b, alpha, relu, groups, eps, margin=npp.MARGIN,
                        l2=npp.L2):
        n, c = x.shape[0], x.shape[1]
        x64 = x.detach().double().reshape(n, groups, -1)
        mean = x64.mean(dim=2)
        var = x64.var(dim=2, unbiased=False)
        stats = torch.stack([mean, 1.0 / torch.sqrt(var + eps)], dim=2).reshape(n * groups, 2).float()
        if weight is not None:
            gamma, beta = self.gamma_beta_fwd(weight, m)
        else:
            gamma = torch.ones(c) if gamma_in is None else gamma_in.detach().reshape(-1)
            beta = torch.zeros(c) if beta_in is None else beta_in.detach().reshape(-1)
        xh = self._gn_xhat(x, groups, stats).float()
        y = self.affine_relu_fwd(xh, gamma, beta, relu)
        if b is None:
            return (y, stats, (gamma if weight is not None else None), (beta if weight is not None else None),
                    None, None, None)
        loss, acc, bits = self.sign_loss_fwd(gamma, b, alpha, margin, l2)
        return y, stats, gamma, beta, loss, acc, bits

    def passport_gn_bwd(self, dy, x, stats, gamma, beta, m, b, alpha, dloss, dgamma_extra, dbeta_extra, wshape, relu,
                        groups, margin=npp.MARGIN, l2=npp.L2):
        n, c = x.shape[0], x.shape[1]
        g32 = torch.ones(c) if gamma is None else gamma.detach().float().reshape(-1)
        b32 = torch.zeros(c) if beta is None else beta.detach().float().reshape(-1)
        xh = self._gn_xhat(x, groups, stats)
        dy64 = dy.detach().double()
        z32 = npp.affine_relu_fwd(xh.float().numpy(), g32.numpy(), b32.numpy(), False)
        dz = torch.where(torch.from_numpy(z32 > 0), dy64, torch.zeros_like(dy64)) if relu else dy64
        dg, db = (dz * xh).sum(dim=(0, 2, 3)).float(), dz.sum(dim=(0, 2, 3)).float()
        gz = (dz * g32.double().view(1, -1, 1, 1)).reshape(n, groups, -1)
        xg = xh.reshape(n, groups, -1)
        invstd = stats[:, 1].double().view(n, groups, 1)
        dx = invstd * (gz - gz.mean(dim=2, keepdim=True) - xg * (gz * xg).mean(dim=2, keepdim=True))
        if dgamma_extra is not None:
            dg = dg + dgamma_extra
        if dbeta_extra is not None:
            db = db + dbeta_extra
        if dloss is not None:
            dg = dg + self.sign_loss_bwd(dloss, g32.contiguous(), b, alpha, margin, l2)
        dw = self.gamma_beta_bwd(dg, db, m, wshape) if wshape is not None else None
        return dx.reshape(x.shape).float(), dw, dg, db

    def sgd_momentum_step(self, flat_param, flat_grad, flat_buf, lr, momentum, weight_decay, grad_scale=1.0):
        with torch.no_grad():
            d = flat_grad * grad_scale + weight_decay * flat_param
            flat_buf.mul_(momentum).add_(d)
            flat_param.add_(flat_buf, alpha=-lr)

    def sgd_momentum_step_dev(self, flat_param, flat_grad, flat_buf, hyper):
        lr, momentum, weight_decay, grad_scale = [float(v) for v in hyper.tolist()]
        self.sgd_momentum_step(flat_param, flat_grad, flat_buf, lr, momentum, weight_decay, grad_scale)

    def gamma_beta_bwd_acc(self, dgamma, dbeta, m, dw):
        with torch.no_grad():
            dw.add_(self.gamma_beta_bwd(dgamma, dbeta, m, dw.shape))
        return dw

    def check_exchange(self):
        return None

    def prepare_stream(self, dev, stream=None):
        return None

    # ---- head of the train step (deepipr_ce_*): float64 restatement of F.cross_entropy + top-1
    def ce_usable(self, logits, target):
        return logits.dtype == torch.float32 and logits.dim() == 2 and target.dtype == torch.int64

    def ce_top1_fwd(self, logits, target):
        x = logits.detach().double()
        lse = torch.logsumexp(x, dim=1)
        loss = (lse - x.gather(1, target.view(-1, 1)).view(-1)).mean()
        top1 = x.argmax(dim=1).eq(target).double().mean() * 100.0
        return loss.float(), top1.float(), lse.float()

    def ce_bwd(self, dloss, logits, target, lse):
        x = logits.detach().double()
        p = torch.exp(x - lse.double().view(-1, 1))
        p[torch.arange(x.shape[0]), target] -= 1.0
        return (p * (dloss.double() / x.shape[0])).float()
