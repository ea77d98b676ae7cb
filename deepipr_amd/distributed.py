"""One process per GPU: torch.distributed on the `nccl` backend (which IS RCCL on ROCm) over xGMI.

The reference's multi-GPU story is single-process nn.DataParallel (experiments/trainer.py:92-93), which
re-broadcasts every parameter each forward and gathers gradients onto GPU 0 through one Python process
(and silently drops the sign loss).  Here every rank owns a full replica, a shard of the
batch and identical passport keys; the only data-path collective is the bucketed gradient all-reduce
that DistributedDataParallel overlaps with backward.  gamma / beta / sign loss are recomputed on every
rank: their dW is rank-identical, so the all-reduce mean leaves it unchanged and the objective equals
the single-process one (mean CE over the global batch + sign loss).

Gradient volume per step: 44.7 MB (ResNet18-c10).  A ring all-reduce moves 2*(N-1)/N * S per GPU over
one xGMI link per direction (~153 GB/s): ~0.5 ms at N=8 -- small against the step, and hidden behind
backward by three ~16 MB buckets (few, large messages: xGMI is point-to-point, per-link bound).
"""
import os

import torch
import torch.distributed as dist

from deepipr_amd.models._builders import PASSPORT_TYPES

BUCKET_MB = 16


def env_world():
    return int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's environment.  Returns (rank, local_rank, world)."""
    rank, local_rank, world = env_world()
    if (world > 1 or os.environ.get('DEEPIPR_FORCE_DDP') == '1') and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def check_keys_materialised(model):
    """key_type='random' draws keys lazily from numpy's global RNG on the first forward
    (models/layers/passportconv2d.py:206,210-216); they must exist before replicas are synchronised."""
    for name, m in model.named_modules():
        if isinstance(m, PASSPORT_TYPES) and (m.get_bias_key() is None or m.get_scale_key() is None):
            raise RuntimeError('passport keys of %s are not set: run one forward (key_type="random") or '
                               'set_intermediate_keys before replicate()' % name)


def broadcast_state(model, src=0):
    """Rank `src`'s parameters and buffers (weights, keys, signature bits, norm statistics) to every rank."""
    if not dist.is_initialized():
        return
    with torch.no_grad():
        seen = set()
        for t in list(model.parameters()) + list(model.buffers()):
            if t is None or t.data_ptr() in seen:
                continue
            seen.add(t.data_ptr())
            dist.broadcast(t, src)
    # c10d writes into the key tensors without bumping their autograd version counter, which is what the
    # pooled-passport cache keys on: drop the cached pooled means explicitly.
    for m in model.modules():
        if isinstance(m, PASSPORT_TYPES):
            m.invalidate_key_cache()


def replicate(model, device, bucket_mb=BUCKET_MB, static_graph=True):
    """Synchronise `model` with rank 0 and wrap it for data-parallel training.

    broadcast_buffers=False: after the one-time sync above the keys and signature bits never change, and
    norm running statistics stay per-rank exactly as under the reference's DataParallel (rank 0's are the
    ones that get saved)."""
    force = os.environ.get('DEEPIPR_FORCE_DDP') == '1'       # exercise the wrapped path on a single GPU
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return model
    check_keys_materialised(model)
    broadcast_state(model, 0)
    from deepipr_amd import passport_ops
    ids = [device.index] if device.type == 'cuda' else None
    ddp = torch.nn.parallel.DistributedDataParallel(
        model, device_ids=ids, broadcast_buffers=False, gradient_as_bucket_view=True, bucket_cap_mb=bucket_mb,
        static_graph=static_graph)     # same parameters used every step: lets DDP skip per-iteration bookkeeping
    # DDP's bucket all-reduces share the device with backward: the co-residency-dependent split-channel kernels are
    # withheld for as long as the wrapper lives
    passport_ops.kernels.withhold_sync(ddp)
    return ddp


def max_over_ranks(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == 'nccl':
            dist.barrier(device_ids=[torch.cuda.current_device()])      # no guessing which GPU the barrier runs on
        else:
            dist.barrier()


def retire_collectives(seconds=0.0):
    """Call before a hipGraph capture when an RCCL process group is alive: return only when its watchdog thread has RETIRED
    every collective issued so far.  The watchdog polls the end event of each unretired collective about every 100 ms; a
    poll that falls INSIDE a stream capture of the main thread comes back as `hipErrorCapturedEvent` ("operation not
    permitted on an event last recorded in a capturing stream") on ROCm 7.2 / torch 2.10 -- also in capture_error_mode
    "thread_local" -- and the watchdog answers by terminating the process (one fresh process in five that captures within a
    few hundred ms of its last collective: tools/nccl_flake_probe.sh, profiles/r04_nccl_flake_probe.txt).
    A CONDITION, not a pause (round 4 slept half a second and hoped): drain the device, so that every collective's end
    event has fired, then block in ProcessGroup._wait_for_pending_works() of every RCCL group -- c10d's own
    waitForPendingWorks: it returns when the watchdog's work list is EMPTY, and an empty list is nothing to poll.  The
    calling thread issues no collective between this call and the end of its capture (bench.py, StepRunner: the capture
    follows immediately), so the list stays empty for as long as it matters.  `seconds` > 0 adds a pause on top (the probe's
    A/B knob); a torch without the call falls back to half a second of pause, as before."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    import time
    try:                                                        # private c10d state: a torch that moved it gets the pause instead
        groups = [pg for pg in list(dist.distributed_c10d._world.pg_map) if _is_rccl(pg)]
    except Exception:
        if dist.get_backend() == 'nccl':
            torch.cuda.synchronize()
            time.sleep(max(seconds, 0.5))
        return
    if not groups:
        return
    torch.cuda.synchronize()
    waited = True
    for pg in groups:
        wait = getattr(pg, '_wait_for_pending_works', None)
        if wait is None:
            waited = False
        else:
            try:
                wait()
            except Exception:
                waited = False
    if seconds > 0 or not waited:
        time.sleep(seconds if waited else max(seconds, 0.5))


def _is_rccl(pg):
    try:
        return dist.get_backend(pg) == 'nccl'
    except Exception:                                           # a group this rank is not part of
        return False


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()


def ranks_seen(device):
    """How many ranks a collective actually reaches: the SUM all-reduce of a one per rank (1 without a process group).
    bench.py reports it next to WORLD_SIZE, which only says what the launcher promised."""
    if not dist.is_initialized():
        return 1
    t = torch.ones(1, dtype=torch.float32, device='cpu' if dist.get_backend() == 'gloo' else device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(round(float(t.item())))


def rank0_first(fn, device=None):
    """Run `fn` on rank 0, then -- behind a barrier -- on every other rank.

    The first convolution of every shape runs MIOpen's find step (cudnn.benchmark, train_v1.py:8), which measures the
    candidate solvers and records the winner in the per-user find database on disk.  N ranks started together race on
    that file and, worse, may each measure a different winner: the ranks of one job would then run different kernels
    (weak-scaling time is the slowest rank's; run-to-run bits differ between ranks).  With rank 0 first, its records are
    on disk when the others look: they take the same solvers and skip the measurement.  -> (fn's result, seconds the
    find phase took on this rank)."""
    import time
    t0 = time.perf_counter()
    many = dist.is_initialized() and dist.get_world_size() > 1
    res = None
    if not many or dist.get_rank() == 0:
        res = fn()
        if device is not None and torch.device(device).type == 'cuda':
            torch.cuda.synchronize(device)
    if many:
        barrier()
        if dist.get_rank() != 0:
            res = fn()
            if device is not None and torch.device(device).type == 'cuda':
                torch.cuda.synchronize(device)
        barrier()
    return res, time.perf_counter() - t0


LAST_AGREEMENT = {}        # what the last gradients_agree() saw (worst distances, where, non-finite sums)


def gradient_digest(grads, device):
    """-> (int64 bit-level checksum, float64 [n, 3] of (sum, sum |g|, ndim) per gradient tensor), on `device`."""
    bits = torch.zeros((), dtype=torch.int64, device=device)
    rows = []
    for g in grads:
        if g is None:
            continue
        g = g.detach().contiguous()
        bits = bits + g.view(torch.int32).to(torch.int64).sum()
        rows.append(torch.stack([g.double().sum(), g.double().abs().sum(),
                                 torch.tensor(float(g.dim()), dtype=torch.float64, device=g.device)]))
    table = torch.stack(rows) if rows else torch.zeros(0, 3, dtype=torch.float64, device=device)
    return bits, table


def digests_agree(a, b):
    """Two gradient_digest()s of the same forward + backward -> (bit for bit, weights within 1e-5, detail).
    The 1e-5 bar is |sum difference| <= 1e-5 * sum |g| per WEIGHT tensor (dim >= 2).  The 1-D tensors (the affine
    parameters of a norm layer that another norm layer follows) are reported, not judged: the gradient of such a shift is
    a sum that cancels to almost nothing (the next normalisation removes a per-channel constant), so its low bits are
    rounding noise of the summation order -- 1e-3 of its own scale between two runs of the very same kernels as soon as
    one vendor kernel in the backward pass accumulates with atomics."""
    (bits_a, t_a), (bits_b, t_b) = a, b
    d = (t_a[:, 0] - t_b[:, 0]).abs()
    rel = torch.nan_to_num(d / (t_a[:, 1] + 1e-30), nan=float('inf'))
    weights = t_a[:, 2] >= 2
    worst_w = float(rel[weights].max()) if bool(weights.any()) else 0.0
    worst_v = float(rel[~weights].max()) if bool((~weights).any()) else 0.0
    nonfinite = int((~torch.isfinite(t_a[:, :2])).sum() + (~torch.isfinite(t_b[:, :2])).sum())
    detail = {'worst_rel_weights': worst_w, 'worst_rel_1d': worst_v, 'nonfinite_sums': nonfinite}
    return bool(int(bits_a) == int(bits_b)), bool(worst_w <= 1e-5 and nonfinite == 0), detail


def launch_census_agrees(counts, device):
    """counts: this rank's {profile slot: launches} of one forward + backward (deepipr_amd._lib.profile_read()).  Do all
    ranks report the same table, i.e. did they route every layer to the same kernel family of this library (and hence the
    same set of layers to the vendor library)?  -> (equal on every rank, rank 0's non-zero entries); (None, None) without a
    process group.  One small all-gather."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return None, None
    names = sorted(counts)
    mine = torch.tensor([int(counts[k]) for k in names], dtype=torch.int64,
                        device='cpu' if dist.get_backend() == 'gloo' else device)
    every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(every, mine)
    same = all(bool(torch.equal(every[0], t)) for t in every[1:])
    return same, {k: int(v) for k, v in zip(names, every[0].tolist()) if v}


def gradients_agree(grads, device):
    """After every rank ran the SAME forward + backward (same weights, same batch): do all ranks hold the same gradients?
    -> (bit for bit, within 1e-5 of scale -- see digests_agree); (None, None) without a process group.  Bit-identical
    gradients mean the ranks ran the same, deterministic kernels (the vendor library's solver choice included); two
    small all-gathers.  LAST_AGREEMENT keeps the worst distances."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return None, None
    bits, table = gradient_digest(grads, device)
    if dist.get_backend() == 'gloo':                        # gloo gathers host tensors (two small vectors)
        table, bits = table.cpu(), bits.cpu()
    world = dist.get_world_size()
    tables = [torch.zeros_like(table) for _ in range(world)]
    dist.all_gather(tables, table)
    allb = [torch.zeros_like(bits) for _ in range(world)]
    dist.all_gather(allb, bits)
    bitwise, close, detail = True, True, {'worst_rel_weights': 0.0, 'worst_rel_1d': 0.0, 'nonfinite_sums': 0}
    for r in range(1, world):
        b, c, d = digests_agree((allb[0], tables[0]), (allb[r], tables[r]))
        bitwise, close = bitwise and b, close and c
        detail = {k: max(detail[k], d[k]) for k in detail}
    LAST_AGREEMENT.clear()
    LAST_AGREEMENT.update(detail)
    return bool(bitwise), bool(close)
