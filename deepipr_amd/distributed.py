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


def gradients_agree(grads, device):
    """After every rank ran the SAME forward + backward (same weights, same batch): do all ranks hold the same gradients?
    -> (bit for bit, within 1e-5 of scale); (None, None) without a process group.  Bit-identical gradients mean the ranks
    ran the same, deterministic kernels (the vendor library's solver choice included); two small all-gathers."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return None, None
    bits = torch.zeros((), dtype=torch.int64, device=device)
    sums = []
    for g in grads:
        if g is None:
            continue
        g = g.detach().contiguous()
        bits = bits + g.view(torch.int32).to(torch.int64).sum()
        sums.append(torch.stack([g.double().sum(), g.double().abs().sum()]))
    vec = torch.cat([bits.to(torch.float64).reshape(1)] + sums) if sums else bits.to(torch.float64).reshape(1)
    if dist.get_backend() == 'gloo':                        # gloo gathers host tensors (two small vectors)
        vec, bits = vec.cpu(), bits.cpu()
    world = dist.get_world_size()
    gathered = [torch.zeros_like(vec) for _ in range(world)]
    dist.all_gather(gathered, vec)
    allb = [torch.zeros_like(bits) for _ in range(world)]
    dist.all_gather(allb, bits)
    bitwise = all(int(b.item()) == int(allb[0].item()) for b in allb)
    ref = gathered[0][1:].view(-1, 2)
    close = True
    for g in gathered[1:]:
        d = (g[1:].view(-1, 2)[:, 0] - ref[:, 0]).abs()
        close = close and bool((d <= 1e-5 * ref[:, 1] + 1e-30).all())
    return bool(bitwise), bool(close)
