"""Cut points of a net's autograd graph, for the staged backward of deepipr_amd/experiments/staged.py.

The data-parallel step exchanges gradients in a few large buckets in gradient-ready order (flat_sgd.py).  To overlap
bucket k's all-reduce with the rest of the backward pass when that backward is replayed from hipGraphs -- where no
Python hook can fire -- the backward is cut into stages at activations the model names itself:

    out, skip = cuts.mark('layer4.0', out, skip)        # in the model's forward: the inputs of layer4's first block

Without an active recorder `mark` returns its arguments untouched (zero cost, the autograd graph stays connected).
Under a recorder that asked for this name, every marked tensor is replaced by a detached leaf for the rest of the
forward; the recorder keeps (upstream handle, leaf) pairs, and the stepper back-propagates stage by stage:
d objective / d leaf from one stage is the grad_output of the upstream handle in the next.  Names may repeat (the
dual forward of schemes V2 / V3 marks every cut twice); pairs are kept in call order.

The reference has no counterpart (nn.DataParallel gathers gradients through one process,
experiments/trainer.py:92-93); this is part of the MI355X-first data-parallel design.
"""
import threading

import torch

_tls = threading.local()


class CutRecorder:
    def __init__(self, names):
        self.names = set(n for n in names if n)
        self.up = {}            # name -> [upstream tensors], in call order
        self.down = {}          # name -> [detached leaves the forward continued from]

    def __enter__(self):
        if getattr(_tls, 'rec', None) is not None:
            raise RuntimeError('cuts: recorders do not nest')
        _tls.rec = self
        return self

    def __exit__(self, *exc):
        _tls.rec = None
        return False


def mark(name, *tensors):
    """-> the tensors to continue the forward with (the arguments themselves unless a recorder cuts here)."""
    rec = getattr(_tls, 'rec', None)
    if rec is None or name not in rec.names or not torch.is_grad_enabled():
        return tensors if len(tensors) != 1 else tensors[0]
    out = []
    for t in tensors:
        if not t.requires_grad:
            out.append(t)
            continue
        leaf = t.detach().requires_grad_(True)
        rec.up.setdefault(name, []).append(t)
        rec.down.setdefault(name, []).append(leaf)
        out.append(leaf)
    return tuple(out) if len(out) != 1 else out[0]
