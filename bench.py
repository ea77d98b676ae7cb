#!/usr/bin/env python
"""bench.py -- images/sec of the DeepIPR passport train step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1 without a launcher: bench.py starts N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): ResNet18 V1 passport (passport_configs/resnet18_passport.json, the 5
layer4 passport layers), CIFAR10-shaped synthetic batches of 128 per GPU resident in HBM, fp32, one step =
zero_grad -> forward -> CE + sum of sign losses -> backward -> SGD(0.01, 0.9, wd 1e-4)
(reference experiments/trainer.py:128-145).  Weak scaling: every rank keeps a batch of 128.

Rank 0 prints ONE JSON line with the whole-job images/sec, plus
  roofline      the dominant hand-written kernel (single-pass norm + passport affine + ReLU backward), timed in
                situ with start/stop HIP events on each dispatch of it, on its launch stream: during the timed
                region when that runs eagerly (--eager, --ddp), on eager steps of the same job right after it
                when the timed region is replayed from a hipGraph (the default; config.launch says which)
  cpu_baseline  the oracle's CPU step ("port") on this host's cores, bounded sample, N=1 only
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from deepipr_amd import _lib, distributed as D                                    # noqa: E402
from deepipr_amd.experiments.trainer import train_step_v1                          # noqa: E402
from deepipr_amd.experiments.trainer_private import DualBranch, train_step_v23     # noqa: E402
from deepipr_amd.experiments.utils import construct_passport_kwargs_from_dict      # noqa: E402
from deepipr_amd.models._builders import PASSPORT_TYPES                            # noqa: E402
from deepipr_amd.models.resnet_passport import ResNet18Passport                    # noqa: E402
from deepipr_amd.models.resnet_passport_private import ResNet18Private             # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured-achievable)
HBM_ACHIEVABLE_GBS = 6290.0      # ... the rate a large copy sustains (same guide)
MFMA_BF16_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 matrix peak (v_mfma_f32_32x32x16_bf16), 16x the fp32 one
MFMA_F32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense fp32 matrix peak (v_mfma_f32_32x32x2_f32, = the vector rate)
# algorithmic bytes per activation element are accounted by the library per timed
# launch: single-pass norm+affine+ReLU 8 forward / 12 backward; 3-launch form: stats 4, apply 8, backward sums 8,
# backward apply 12; plain affine 8 / 12; SGD 20 per parameter


def build_model(args, device):
    if getattr(args, 'arch', 'resnet18') == 'alexnet':
        from deepipr_amd.models.alexnet_passport import AlexNetPassport
        from deepipr_amd.models.alexnet_passport_private import AlexNetPassportPrivate
        cfg = json.load(open(os.path.join(ROOT, 'passport_configs', 'alexnet_passport.json')))
        kw = construct_passport_kwargs_from_dict({'passport_config': cfg, 'norm_type': getattr(args, 'norm_type', 'bn'),
                                                  'key_type': 'random', 'sl_ratio': 0.1})
        torch.manual_seed(0)
        np.random.seed(0)
        cls = AlexNetPassport if args.scheme == 1 else AlexNetPassportPrivate
        return cls(3, args.classes, kw, imagenet=getattr(args, 'image_size', 32) > 32).to(device)
    arch = getattr(args, 'arch', 'resnet18')
    cfg = json.load(open(os.path.join(ROOT, 'passport_configs', '%s_passport.json' % arch)))
    kw = construct_passport_kwargs_from_dict({'passport_config': cfg, 'norm_type': getattr(args, 'norm_type', 'bn'),
                                              'key_type': 'random', 'sl_ratio': 0.1})
    torch.manual_seed(0)
    np.random.seed(0)
    if arch == 'resnet50':                       # no reference implementation: see BottleneckPassportBlock
        from deepipr_amd.models.resnet_passport import ResNet50Passport
        assert args.scheme == 1, 'the ResNet50 passport variant exists for scheme V1 only'
        return ResNet50Passport(num_classes=args.classes, passport_kwargs=kw,
                                imagenet=getattr(args, 'image_size', 32) > 32).to(device)
    if args.scheme == 1:
        model = ResNet18Passport(num_classes=args.classes, passport_kwargs=kw, imagenet=getattr(args, 'image_size', 32) > 32)
    else:
        model = ResNet18Private(num_classes=args.classes, passport_kwargs=kw, imagenet=getattr(args, 'image_size', 32) > 32)
    return model.to(device)


def _leave():
    """End of a rank.  With RCCL ranks the process leaves WITHOUT tearing the communicator down: a benchmark whose line is
    already printed must not turn into a non-zero exit code because a library thread objects during teardown (the one
    abort seen there in round 4 was RCCL's watchdog, whose poll had fallen inside a stream capture -- the captures are
    now preceded by distributed.retire_collectives(); profiles/r04_nccl_flake_probe.txt).  The ranks meet at a last barrier
    first -- rank 0 reaches it after it has printed its line, so no peer is gone before the result is out and a late failure
    of rank 0 is not mistaken for a peer's early exit -- and the process runs its atexit handlers (profilers and tracers
    flush their output there) before it leaves.  DEEPIPR_BENCH_HARD_EXIT=0 takes the ordinary shutdown instead.
    Everything else (one process, gloo rehearsals, the dry run) shuts down normally."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == 'nccl' and dist.get_world_size() > 1:
        sys.stdout.flush()
        sys.stderr.flush()
        code = 0
        try:
            D.barrier()
        except Exception as e:                               # a peer is gone: say so, leave anyway -- with a non-zero exit code
            print('bench.py: final barrier failed: %s' % e, file=sys.stderr, flush=True)
            code = 3
        if os.environ.get('DEEPIPR_BENCH_HARD_EXIT', '1') != '0':
            import atexit
            try:
                atexit._run_exitfuncs()
            finally:
                sys.stdout.flush()
                sys.stderr.flush()
                os._exit(code)
        if code:
            sys.exit(code)
    D.shutdown()


def _conv_arith():
    try:
        from deepipr_amd.passport_ops import kernels
        return kernels.conv_arith()
    except Exception:
        return None


def _exchange_timeouts():
    try:
        from deepipr_amd.passport_ops import kernels
        return int(kernels.sync_timeouts())
    except Exception:                                     # never let a diagnostic take the benchmark line down
        return None


def _numel(out):
    return (out[0] if isinstance(out, tuple) else out).numel()      # a block's last layer returns its output twice


def fused_layer_elements(model, run_once):
    """Activation elements of every layer call that goes through the fused norm+affine kernels during one
    step (passport layers and BatchNorm ConvBlocks), for the algorithmic byte count.  -> (all, passport only)"""
    from deepipr_amd.models.layers.conv2d import ConvBlock
    sizes, passport = [], []
    hooks = []
    for m in model.modules():
        if isinstance(m, PASSPORT_TYPES):
            hooks.append(m.register_forward_hook(lambda mod, i, o: (sizes.append(_numel(o)), passport.append(_numel(o))) and None))
        elif isinstance(m, ConvBlock) and m.bn is not None and m.fuse_norm:
            hooks.append(m.register_forward_hook(lambda mod, i, o: sizes.append(_numel(o)) and None))
    run_once()
    for h in hooks:
        h.remove()
    return sizes, passport


# The other BASELINE.json configurations that fit one GPU, measured in the SAME driver run as the headline (N = 1 only; a
# bounded sub-run each, about 20 s with its find phase): one rank's shard of config 3 (V2, 100 classes, 32 images), config 4's
# shard (V3: 64 images + the trigger pair), config 1 (AlexNet V1, batch 64) on the GPU and config 5's geometry (ResNet50, 40 s).
OTHER_CONFIGS = [
    ('P_shard', 'BASELINE configs[2], one rank of 8: ResNet18 V2 private, CIFAR100 shapes, 32 images per GPU',
     ['--scheme', '2', '--classes', '100', '--batch', '32']),
    ('V3_shard', 'BASELINE configs[3], one rank of 4: ResNet18 V3 (V2 + trigger pair), CIFAR100 shapes, 64 + 2 images per GPU',
     ['--scheme', '3', '--classes', '100', '--batch', '64']),
    ('AlexNet_A', 'BASELINE configs[0] on the GPU: AlexNet V1 passport, CIFAR10 shapes, batch 64',
     ['--arch', 'alexnet', '--batch', '64']),
    # config 5 (ImageNet geometry; the ResNet50 passport variant is composed, the reference has none): a SHORT sub-run -- 8 timed
    # steps of ~90 ms, MIOpen's immediate mode instead of its find phase (minutes for this net's shapes), no per-kernel timing
    ('R50_imagenet', 'BASELINE configs[4] geometry, one rank: ResNet50 passport variant, 3x224x224, 1000 classes, 256 images per GPU',
     ['--arch', 'resnet50', '--image-size', '224', '--classes', '1000', '--batch', '256', '--no-miopen-find', '--no-kernel-timing'], 8, 3),
]


def is_headline(args):
    return (args.arch == 'resnet18' and args.scheme == 1 and args.batch == 128 and args.image_size == 32 and args.classes == 10
            and args.norm_type == 'bn' and not (args.eager or args.ddp or args.no_fuse))


def other_configs(steps=60, warmup=15, timeout=150):
    import subprocess
    res = {}
    for key, what, flags, *own in OTHER_CONFIGS:
        k_steps, k_warmup = own if own else (steps, warmup)
        cmd = [sys.executable, os.path.abspath(__file__), '--steps', str(k_steps), '--warmup', str(k_warmup), '--no-cpu-baseline',
               '--no-stress', '--no-configs'] + flags
        t0 = time.perf_counter()
        try:
            run = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
            line = [ln for ln in run.stdout.splitlines() if ln.startswith('{')][-1]
            d = json.loads(line)
            rec = {'what': what, 'flags': ' '.join(flags), 'value': d['value'], 'unit': d['unit'], 'ms_per_step': d['ms_per_step'],
                   'steps': d['steps'], 'warmup': d['warmup'], 'sign_detect_acc': d.get('sign_detect_acc'),
                   'exchange_timeouts': d.get('exchange_timeouts'), 'workload': d['config']['workload'],
                   'wall_s': round(time.perf_counter() - t0, 1)}
            for k in ('roofline', 'roofline_passport'):
                if k in d:
                    r = {kk: vv for kk, vv in d[k].items() if kk not in ('kernels', 'note', 'traffic_source')}
                    r['kernel'] = str(r.get('kernel', ''))[:160]
                    rec[k] = r
            res[key] = rec
        except Exception as e:                               # a sub-run must never take the headline line down
            res[key] = {'what': what, 'flags': ' '.join(flags), 'error': '%s: %s' % (type(e).__name__, str(e)[:200])}
    return res


def host_cores():
    """CPUs this process may actually use: min(affinity, cgroup cpu.max quota), capped at 32."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return min(n, 32)


def cpu_baseline(args, budget_s=20.0):
    """The oracle's plain-PyTorch CPU step on the same workload, bounded to ~budget_s seconds."""
    from oracle import torch_ref
    arch = getattr(args, 'arch', 'resnet18')
    hw = getattr(args, 'image_size', 32)
    cfg = json.load(open(os.path.join(ROOT, 'passport_configs', '%s_passport.json' % arch)))
    kw = torch_ref.passport_kwargs_from_config(cfg, 'bn', 'random', 0.1)
    torch.manual_seed(0)
    np.random.seed(0)
    # Host threads = the cores the cgroup grants (16 on the GPU box: 256 logical CPUs visible, cpu.max
    # quota 16; running oneDNN on all 256 visible threads took 75 s per step).
    threads = int(os.environ.get('DEEPIPR_CPU_THREADS', host_cores()))
    torch.set_num_threads(threads)
    if arch == 'alexnet':
        model = torch_ref.AlexNetRef(3, args.classes, kw, private=args.scheme != 1)
    elif arch == 'resnet50':
        model = torch_ref.resnet50_ref(num_classes=args.classes, passport_kwargs=kw, imagenet=hw > 32)
    else:
        model = torch_ref.resnet18_ref(num_classes=args.classes, passport_kwargs=kw, private=args.scheme != 1,
                                       imagenet=hw > 32)
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    g = torch.Generator().manual_seed(1234)
    extra = 2 if args.scheme == 3 else 0                    # V3: the trigger pair rides along with every batch
    x = torch.randn(args.batch + extra, 3, hw, hw, generator=g)
    y = torch.randint(0, args.classes, (args.batch + extra,), generator=g)
    step = torch_ref.v1_step if args.scheme == 1 else torch_ref.v23_step
    model.train()
    step(model, opt, x, y)                                  # warm-up (oneDNN primitive creation)
    n, t0 = 0, time.perf_counter()
    while True:
        step(model, opt, x, y)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or n >= 50:
            break
    return {'value': round(n * args.batch / dt, 2), 'unit': 'img/s', 'cores': threads, 'kind': 'port',
            'sample': '%d steps of batch %d (%.1f s) of oracle/torch_ref.py on %d host threads, torch %s'
                      % (n, args.batch, dt, threads, torch.__version__)}


def stress_roofline(device, reps=30):
    """The same dominant kernel pair on a tensor large enough to leave the caches (S3 [512,512,8,8], 67 MB
    per tensor, 201 MB of algorithmic traffic per backward-apply launch), timed with the library's own
    per-dispatch events.  Kernel quality evidence next to the launch-bound in-situ figure."""
    from deepipr_amd.passport_ops import kernels as K
    n, c, h, w = 512, 512, 8, 8
    x = torch.randn(n, c, h, w, device=device)
    dy = torch.randn(n, c, h, w, device=device)
    g, b = torch.randn(c, device=device), torch.randn(c, device=device)
    rm, rv = torch.zeros(c, device=device), torch.ones(c, device=device)

    def once():
        out = K.passport_bn_fwd(x, None, None, g, b, None, 0.0, True, rm, rv, None, 0.1, 1e-5, True)
        K.passport_bn_bwd(dy, x, out[1], None, None, 0.0, None, None, None, None, True, True)
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(reps):
        once()
    torch.cuda.synchronize()
    prof, nbytes = _lib.profile_read(), _lib.profile_read_bytes()
    _lib.profile_enable(False)
    res = {}
    for name in ('bn_res_bwd', 'bn_res_fwd', 'bn_affine_bwd', 'bn_affine_fwd', 'bn_bwd_reduce', 'bn_stats'):
        ms, cnt = prof[name]
        if cnt:
            res[name] = {'avg_us': round(1000.0 * ms / cnt, 2), 'GBps': round(nbytes[name] / (ms * 1e-3) / 1e9, 1),
                         'bytes_per_launch': int(nbytes[name] / cnt)}
    dom = 'bn_res_bwd' if 'bn_res_bwd' in res else 'bn_affine_bwd'
    a = res[dom]
    return {'bound': 'hbm', 'kernel': 'k_' + dom, 'shape': [n, c, h, w], 'achieved': a['GBps'],
            'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(a['GBps'] / HBM_PEAK_GBS, 4),
            'bytes_per_launch': a['bytes_per_launch'], 'avg_us': a['avg_us'], 'kernels': res}


def pmc_record_stale(kernel, shape_key):
    """The committed PMC record names the kernel source it was measured on (`source`, `source_sha16`): True when that file has
    changed since (the record is then a figure about an OLDER kernel and is printed flagged), None when the record carries no
    such stamp."""
    import hashlib
    try:
        rec = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))[kernel][shape_key]
        if 'source_sha16' not in rec:
            return None
        now = hashlib.sha256(open(os.path.join(ROOT, rec['source']), 'rb').read()).hexdigest()[:16]
        return now != rec['source_sha16']
    except (OSError, KeyError, ValueError):
        return None


def pmc_traffic(kernel, shape_key, algorithmic=None):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json), or None.
    PMC counters need their own rocprofv3 runs (separate FETCH_SIZE / WRITE_SIZE passes), so they cannot be taken
    inside this process; the committed record is quoted only when it was measured on the SAME launch mix, i.e. when
    the algorithmic bytes per launch recorded with it equal the ones accounted in this run (`algorithmic`)."""
    try:
        rec = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))[kernel][shape_key]
        if algorithmic is not None and abs(rec['algorithmic'] / algorithmic - 1.0) > 0.01:
            return None
        return int(rec['fetch'] + rec['write'])
    except (OSError, KeyError, ValueError, ZeroDivisionError):
        return None


def self_launch(argv, n):
    """`python bench.py --gpus N` typed without a launcher (the reference's multi-GPU entry is one command too,
    experiments/trainer.py:92-93): re-execute under torch.distributed.run with N ranks on this node, rendezvous on
    127.0.0.1 and a free port; rank 0's JSON line passes through on stdout, the exit code is the launcher's."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault('OMP_NUM_THREADS', '4')
    # the ranks live in their own process group and go down with this process: a `timeout` or Ctrl-C aimed at the one
    # command the user typed must not leave N orphans on the GPUs
    import signal
    proc = subprocess.Popen(cmd, env=env, start_new_session=True)

    def forward(signum, _frame):
        try:
            os.killpg(proc.pid, signum)
        except ProcessLookupError:
            pass
    old = {sig: signal.signal(sig, forward) for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP)}
    try:
        return proc.wait()
    finally:
        for sig, handler in old.items():
            signal.signal(sig, handler)
        if proc.poll() is None:
            os.killpg(proc.pid, signal.SIGKILL)


def dry_run(args):
    """--dry-run: the launch plumbing only (rendezvous, barrier, max-over-ranks, one JSON line from rank 0), on any
    backend and without a GPU -- what tests/test_bench_helpers.py runs in the CPU container.  No workload, no value."""
    import torch.distributed as dist
    rank, _local, world = D.init_from_env(args.backend)
    t0 = time.perf_counter()
    D.barrier()
    dt = time.perf_counter() - t0
    dev = torch.device('cpu') if (args.backend == 'gloo' or not torch.cuda.is_available()) else torch.device('cuda', _local)
    dt = D.max_over_ranks(dt, dev)
    seen = dist.get_world_size() if dist.is_initialized() else 1
    ranks = D.ranks_seen(dev)                               # counted by a collective (gloo here, RCCL on the GPUs)
    if rank == 0:
        print(json.dumps({'metric': 'dry run: launcher and process group only', 'value': None, 'unit': 'img/s',
                          'n_gpus': args.gpus, 'world_size_seen': seen, 'rccl_ranks_seen': ranks, 'steps': 0, 'warmup': 0,
                          'scheme': args.scheme, 'dry_run': True,
                          'barrier_s': round(dt, 4)}), flush=True)
    D.shutdown()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=128, help='per-GPU batch')
    ap.add_argument('--scheme', type=int, default=1, choices=[1, 2, 3],
                    help='1 = V1 (train_v1.py), 2 = V2 private passports, 3 = V3: V2 + a trigger-set pair appended to every '
                         'batch (train_v23.py --train-backdoor; experiments/trainer_private.py:135-146, dataset.py:188-191)')
    ap.add_argument('--classes', type=int, default=10)
    ap.add_argument('--arch', default='resnet18', choices=['resnet18', 'resnet50', 'alexnet'])
    ap.add_argument('--image-size', type=int, default=32, help='32 = CIFAR shapes, 224 = ImageNet shapes')
    ap.add_argument('--norm-type', default='bn', choices=['bn', 'gn', 'in'], help="the layers' norm (reference --norm-type)")
    ap.add_argument('--no-fuse', action='store_true', help='library norm kernels + unfused passport kernels (A/B)')
    ap.add_argument('--no-miopen-find', action='store_true', help='cudnn.benchmark = False: MIOpen immediate mode')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--no-stress', action='store_true', help='skip the stress-shape roofline measurement')
    ap.add_argument('--no-configs', action='store_true', help="skip the other BASELINE configurations' lines (`configs` block)")
    ap.add_argument('--graph', action='store_true', help='hipGraph replay of the step (the default; kept for compatibility)')
    ap.add_argument('--eager', action='store_true', help='eager dispatch of the timed region (exchange overlapped with backward)')
    ap.add_argument('--ddp', action='store_true', help='DistributedDataParallel + torch fused SGD instead of FlatSGD')
    ap.add_argument('--ddp-static-graph', type=int, default=1, help='DistributedDataParallel(static_graph=...)')
    ap.add_argument('--verbose', action='store_true', help='phase progress with timestamps on stderr')
    ap.add_argument('--unstaged', action='store_true', help='N > 1: forward + backward as ONE graph, the whole exchange '
                    'after it (round-2 form) instead of the staged, overlapped exchange')
    ap.add_argument('--backend', default=None, help='torch.distributed backend (default nccl = RCCL)')
    ap.add_argument('--stage-host-wait-histogram', action='store_true',
                    help='N > 1 (or DEEPIPR_FORCE_DDP=1): 200 extra steps; p50 / p99 / max of the host waits on each backward '
                         "stage's event (the host sits in the staged step's loop three times per step)")
    ap.add_argument('--dry-run', action='store_true', help='launch plumbing only: no workload (CPU-testable)')
    args = ap.parse_args()
    t_start = time.perf_counter()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ and os.environ.get('DEEPIPR_FORCE_DDP') != '1':
        raise SystemExit(self_launch(sys.argv[1:], args.gpus))
    if args.dry_run:
        return dry_run(args)

    def note(msg):
        if args.verbose or args.gpus > 1:                  # several ranks: always say where the time goes (stderr)
            print('bench.py [%7.1f s] %s' % (time.perf_counter() - t_start, msg), file=sys.stderr, flush=True)

    rank, local_rank, world = D.init_from_env(args.backend)
    if world != args.gpus:
        args.gpus = world                                 # a launcher's WORLD_SIZE wins over --gpus
    assert torch.cuda.is_available(), 'bench.py needs an MI355X; there is no CPU path'
    if local_rank >= torch.cuda.device_count():
        # rehearsal of the N > 1 path on a box with fewer GPUs (ranks share devices; RCCL refuses that, so it needs
        # --backend gloo): correctness of the exchange plumbing only, the timing means nothing
        if os.environ.get('DEEPIPR_SHARE_GPU') != '1':
            raise SystemExit('rank %d: only %d GPU(s) visible (DEEPIPR_SHARE_GPU=1 --backend gloo shares them for a '
                             'functional rehearsal)' % (local_rank, torch.cuda.device_count()))
        local_rank %= torch.cuda.device_count()
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)
    torch.backends.cudnn.benchmark = not args.no_miopen_find     # MIOpen find-mode, as train_v1.py:8

    own_policy = None
    if world > 1:
        from deepipr_amd import passport_ops as _po
        own_policy = _po.prefer_own_kernels()              # every rank on the same, bit-reproducible kernels
    model = build_model(args, device)
    note('model built')
    if args.no_fuse:
        for mod in model.modules():
            if hasattr(mod, 'fuse_norm'):
                mod.fuse_norm = False
    g = torch.Generator(device='cpu').manual_seed(1234 + rank)
    nb = 4                                                  # distinct synthetic batches, resident in HBM
    hw = args.image_size
    xs = [torch.randn(args.batch, 3, hw, hw, generator=g).to(device) for _ in range(nb)]
    ys = [torch.randint(0, args.classes, (args.batch,), generator=g).to(device) for _ in range(nb)]
    model.train()
    with torch.no_grad():
        model(xs[0])                                        # materialises the random keys
    # Data parallelism: FlatSGD by default (flat parameter / gradient / momentum buffers, two large RCCL
    # all-reduces with the first one overlapped with backward, one fused HIP optimiser kernel); --ddp selects
    # DistributedDataParallel + torch's fused SGD for comparison.
    if args.ddp:
        wrap = lambda m: D.replicate(m, device, static_graph=bool(args.ddp_static_graph))
        opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4, fused=True)
    else:
        D.check_keys_materialised(model)
        D.broadcast_state(model, 0)
        wrap = lambda m: m
        from deepipr_amd.flat_sgd import FlatSGD
        opt = FlatSGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    # V3: the trigger-set loader hands out 2 images per step (dataset.py:188-191), concatenated to the batch INSIDE the
    # step as the reference does (trainer_private.py:142-146): 2 small kernels per step, part of the timed region
    wm = None
    if args.scheme == 3:
        wm = ([torch.randn(2, 3, hw, hw, generator=g).to(device) for _ in range(nb)],
              [torch.randint(0, args.classes, (2,), generator=g).to(device) for _ in range(nb)])

    def batch(i):
        if wm is None:
            return xs[i % nb], ys[i % nb]
        return torch.cat([xs[i % nb], wm[0][i % nb]], dim=0), torch.cat([ys[i % nb], wm[1][i % nb]], dim=0)
    if args.scheme == 1:
        net = wrap(model)
        step = lambda i: train_step_v1(net, opt, *batch(i))
    else:
        net = wrap(DualBranch(model))
        step = lambda i: train_step_v23(net, opt, *batch(i))

    note('keys drawn, state broadcast, optimiser built')
    # Find phase: one collective-free forward + backward (torch.autograd.grad: no gradient hooks, no optimiser) so that
    # MIOpen picks its solver for every convolution shape -- RANK 0 FIRST, the others behind a barrier: they then find
    # rank 0's records in the user find-database, take the same solvers and skip the measurement (distributed.rank0_first).
    # Every rank runs it on rank 0's batch from the broadcast weights, so the ranks' gradients must agree: bit for bit when
    # they run the same deterministic kernels, to rounding otherwise -- all-gathered and reported (ranks_agree*).
    import torch.distributed as tdist
    many = tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1
    raw = model if args.scheme == 1 else DualBranch(model)
    fwd_loss = (train_step_v1 if args.scheme == 1 else train_step_v23).forward_loss
    probe_x, probe_y = batch(0)
    if many:
        probe_x, probe_y = probe_x.clone(), probe_y.clone()
        tdist.broadcast(probe_x, 0)
        tdist.broadcast(probe_y, 0)
    params = [p for p in model.parameters() if p.requires_grad]
    probe = {}

    def find_pass():
        objective, _ = fwd_loss(raw, probe_x, probe_y)
        probe['grads'] = torch.autograd.grad(objective, params, allow_unused=True)
    keep0 = {k: v.clone() for k, v in model.state_dict().items()} if many else None
    (all_elems, elems), find_s = D.rank0_first(lambda: fused_layer_elements(model, find_pass), device)
    agree, agree_tol = D.gradients_agree(probe['grads'], device)
    agree_detail = dict(D.LAST_AGREEMENT) or None
    # the yardstick for the line above: the SAME rank running the same pass twice (vendor kernels that accumulate with
    # atomics make even that differ in the low bits; this library's kernels do not)
    repeat = None
    census_same, census = None, None
    if many:
        first = D.gradient_digest(probe.pop('grads'), device)
        _lib.profile_enable(True)                          # ... and which of this library's kernel families served it, per rank
        find_pass()
        torch.cuda.synchronize()
        _lib.profile_enable(False)
        census_same, census = D.launch_census_agrees({k: v[1] for k, v in _lib.profile_read().items()}, device)
        again = D.gradient_digest(probe.pop('grads'), device)
        rb, rc, rd = D.digests_agree(first, again)
        repeat = {'bitwise': rb, 'weights_1e-5': rc, **rd}
        # ... and the verdict that does not depend on the vendor library being deterministic: the ranks differ from one
        # another by no more than (4 x) what the worst rank differs from itself on a repeat, or by less than 1e-5
        worst = torch.tensor([rd['worst_rel_weights']], dtype=torch.float64, device='cpu' if tdist.get_backend() == 'gloo' else device)
        tdist.all_reduce(worst, op=tdist.ReduceOp.MAX)
        repeat['worst_rel_weights_any_rank'] = float(worst.item())
        repeat['ranks_differ_like_a_repeat'] = bool(agree_tol or (agree_detail or {}).get('worst_rel_weights', 0.0)
                                                    <= 4.0 * float(worst.item()))
    probe.clear()
    if keep0 is not None:                                  # every rank back on the broadcast state (norm statistics moved)
        with torch.no_grad():
            model.load_state_dict(keep0)
    del probe_x, probe_y, keep0
    find_timeouts = _exchange_timeouts() or 0
    note('find phase done in %.1f s (rank 0 first): MIOpen solver selection for every conv shape; ranks agree: %s %s; '
         'this rank against itself: %s; exchange time-outs on this rank during it: %d'
         % (find_s, agree, agree_detail or '', repeat, find_timeouts))
    # Launch mode of the timed region: hipGraph replay by default.  The step issues ~260 dispatches (~500 for the
    # dual-forward V2/V3 step); eager enqueue costs 4.5-9 ms of host time against 5.1-5.6 ms of GPU time, so an eager
    # step is host-bound exactly where it matters most (32 images per GPU in config P).
    #   one GPU      : the whole step (zero_grad .. optimiser) is one graph;
    #   several GPUs : zero_grad .. backward is one graph; FlatSGD's RCCL all-reduce of the flat gradient buffer and the
    #                  fused SGD kernel are enqueued eagerly after each replay (no collective is ever captured).  The exchange
    #                  (44.7 MB, ~0.5 ms over xGMI at N = 8) is then not hidden behind backward, but the step no
    #                  longer waits for the host -- measured on one GPU with the exchange forced on
    #                  (DEEPIPR_FORCE_DDP=1): see DESIGN.md 5.
    # --eager: eager dispatch with the exchange overlapped with backward; --ddp: DistributedDataParallel (eager).
    eager_step = step
    use_graph = not args.eager and not args.ddp
    fn = train_step_v1 if args.scheme == 1 else train_step_v23
    graphed, launch_form = None, 'eager'

    def everyone(ok):
        """True only if `ok` on EVERY rank: the ranks must take the same launch form (their collectives pair up)."""
        if not (tdist.is_available() and tdist.is_initialized()):
            return ok
        flag = torch.tensor([1.0 if ok else 0.0], device=device)
        tdist.all_reduce(flag, op=tdist.ReduceOp.MIN)
        return bool(flag.item() > 0.5)

    def build(form):
        if form == 'staged':
            # data parallel: backward cut into stages at the gradient buckets' boundaries, captured back to back;
            # bucket k's RCCL all-reduce on a side stream while stage k + 1 replays (experiments/staged.py)
            from deepipr_amd.experiments.staged import StagedStep
            return StagedStep(fn, net, opt, *batch(0), graph=True)
        from deepipr_amd.experiments.graph_step import GraphedTrainStep
        return GraphedTrainStep(fn, net, opt, *batch(0), optimizer_in_graph=not tdist.is_initialized())

    # (launch form, in-launch exchange allowed): with several ranks the forms are tried in this order until one runs
    # clean on EVERY rank; the last resort gives up the split-channel single-pass kernels (three launches for layers with
    # fewer channels than CUs) -- e.g. when something else occupies CUs of the device (two processes on one GPU).
    forms = []
    if use_graph:
        many = tdist.is_initialized()
        first = 'unstaged' if (args.unstaged or not many) else 'staged'
        forms = [(first, True)] + ([(first, False), ('unstaged', False)] if many else [])
    from deepipr_amd.passport_ops import kernels as _k
    if not _k.sync_user:                                   # the user switched the in-launch exchange off already
        forms = [(name, False) for name, _s in forms]
    forms = [f for i, f in enumerate(forms) if f not in forms[:i]]
    # state to return to if a launch form has to be given up after it already ran steps (N > 1 only)
    keep = None
    if len(forms) > 1:
        keep = ({k: v.clone() for k, v in model.state_dict().items()}, opt.flat_buf.clone())
    use_graph, sync_default = False, _k.sync_user
    for form, sync_ok in forms:
        err = None
        try:
            if not sync_ok and _k.sync_user:
                _k.set_user_sync(False)
            graphed = build(form)
            step = lambda i, g=graphed: g(*batch(i))
            if len(forms) > 1:
                # a few replays before the form is accepted: an in-launch exchange that timed out next to a collective
                # (never seen with one process per GPU; staged.py explains why it should not happen) shows here
                for i in range(3):
                    step(i)
                torch.cuda.synchronize()
                if _exchange_timeouts():
                    raise RuntimeError('an in-launch exchange of the single-pass norm kernels timed out')
        except Exception as exc:                           # capture refused / form unusable
            err = exc
        if everyone(err is None):
            use_graph, launch_form = True, form + ('' if sync_ok or not sync_default else ', in-launch exchange off')
            break
        print('bench.py: launch form %r (in-launch exchange %s) given up (%s); trying the next one' % (
            form, 'on' if sync_ok else 'off',
            'another rank failed' if err is None else '%s: %s' % (type(err).__name__, err)), file=sys.stderr)
        torch.cuda.synchronize()
        if hasattr(graphed, 'close'):
            graphed.close()
        graphed, step = None, eager_step
        if keep is not None:
            with torch.no_grad():
                model.load_state_dict(keep[0])
                opt.flat_buf.copy_(keep[1])
            _k.reset_sync_words()
    import torch.distributed as _td
    tdist_on = _td.is_available() and _td.is_initialized()
    note('launch mode: %s' % ('hipGraph replay' if use_graph else 'eager'))
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    note('warm-up done')
    timing = not args.no_kernel_timing
    D.barrier()
    torch.cuda.synchronize()
    # In-situ kernel timing: start/stop events on each kernel's own dispatch.  Eager timed region: on every
    # `stride`-th step of it (dispatching with per-kernel events costs ~0.8 ms of host time on such a step;
    # sampling keeps the region within ~1 % of an untimed one).  Graph-replayed timed region: per-dispatch events
    # cannot be captured, so the same steps are run eagerly right AFTER the timed region and every one is timed.
    stride = max(1, min(10, args.steps // 3))
    if timing:
        _lib.profile_enable(1)
        _lib.profile_enable(0)
    t0 = time.perf_counter()
    for i in range(args.steps):
        if timing and not use_graph and i % stride == 0:
            _lib.profile_enable(2)
            step(i)
            _lib.profile_enable(0)
        else:
            step(i)
    torch.cuda.synchronize()
    D.barrier()
    dt = time.perf_counter() - t0
    note('timed region done: %.3f ms per step' % (1000.0 * dt / args.steps))
    # `value` is EXACTLY the --steps region above (the driver's contract).  A short one (the driver's 20 steps of 3.5 ms are 0.07 s)
    # is backed by a second, longer region of the same step right after it -- reported beside it, never instead of it
    # (VERDICT r05 weak 8): at least 100 steps, bounded to about two seconds.
    steady = None
    if args.steps < 100 and dt > 0:
        ks = int(min(max(100, args.steps), max(args.steps, 2.0 / (dt / args.steps))))
        if ks > args.steps:
            D.barrier()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(ks):
                step(i)
            torch.cuda.synchronize()
            D.barrier()
            steady = {'steps': ks, 'ms_per_step': round(1000.0 * D.max_over_ranks(time.perf_counter() - t1, device) / ks, 4)}
            note('steady-state check: %d steps, %.3f ms per step' % (ks, steady['ms_per_step']))
    sampled = len(range(0, args.steps, stride))
    _k.profile_passport = timing                           # passport-layer launches also go to the scoped counters
    if timing and use_graph:
        sampled = min(args.steps, 30)
        for i in range(3):
            eager_step(i)                                  # back to eager dispatch: allocator / MIOpen handles warm
        torch.cuda.synchronize()
        _lib.profile_enable(2)
        for i in range(sampled):
            eager_step(i)
        torch.cuda.synchronize()
        _lib.profile_enable(0)
    prof = _lib.profile_read() if timing else {}
    prof_bytes = _lib.profile_read_bytes() if timing else {}
    prof_scope = _lib.profile_read_scope() if timing else {}
    _k.profile_passport = False
    if timing:
        _lib.profile_enable(False)
    dt = D.max_over_ranks(dt, device)
    ranks_seen = D.ranks_seen(device)                      # a SUM all-reduce of ones over the process group (1 without)
    # exposed part of the gradient exchange (staged mode): GPU time between the end of the last backward stage and the
    # SGD kernel -- pack + all-reduce of the buckets that could not travel under backward + the waits -- from events on
    # 20 extra steps after the timed region (max over ranks).  None when no exchange runs.
    exposed_us, stage_plan = None, None
    if use_graph and tdist_on and hasattr(graphed, 'describe'):
        stage_plan = graphed.describe()
        opt.exposed_events = []
        graphed.host_times = {}
        extra = 200 if args.stage_host_wait_histogram else 20
        for i in range(extra):
            step(i)
        torch.cuda.synchronize()
        ht, graphed.host_times = graphed.host_times, None
        waits = ht.pop('waits', {})
        stage_plan['host_us_per_step'] = {k: round(1e6 * v / extra, 1) for k, v in ht.items() if not k.startswith('calls_')}
        if waits:
            # how long the host sat in Event.synchronize() behind each backward stage before it could enqueue that stage's bucket
            stage_plan['host_wait_us'] = {
                'stage_%d' % k: {'n': len(v), 'p50': round(1e6 * float(np.percentile(v, 50)), 1),
                                 'p99': round(1e6 * float(np.percentile(v, 99)), 1), 'max': round(1e6 * max(v), 1)}
                for k, v in sorted(waits.items())}
        if opt.exposed_events:
            exposed_us = 1000.0 * sum(a.elapsed_time(b) for a, b in opt.exposed_events) / len(opt.exposed_events)
            exposed_us = D.max_over_ranks(exposed_us, device)
        opt.exposed_events = None

    # signature read-out after the run: sign(gamma) == b per passport layer (experiments/trainer_private.py:37-71)
    from deepipr_amd.experiments.trainer_private import TesterPrivate
    detect = TesterPrivate(model, device, verbose=False).test_signature()
    model.train()
    if rank != 0:
        _leave()
        return
    value = args.gpus * args.batch * args.steps / dt
    out = {
        'metric': 'images/sec %s-passport %s train step' % (
            {'resnet18': 'ResNet18', 'resnet50': 'ResNet50', 'alexnet': 'AlexNet'}[args.arch],
            'CIFAR%d' % args.classes if args.image_size == 32 else 'ImageNet-shape'),
        'value': round(value, 1), 'unit': 'img/s',
        'n_gpus': args.gpus, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(1000.0 * dt / args.steps, 4), 'higher_is_better': True, 'scaling': 'weak',
        'steady_state_check': steady,                      # a longer region of the same step behind a short --steps (None otherwise)
        'vs_baseline': None, 'dtype': 'f32' if _conv_arith() != 'bf16x3' else 'f32 (3x3 stride-1 weight gradients: bf16x3 products, f32 accumulate)',
        'data': 'synthetic',
        'sign_detect_acc': round(sum(detect.values()) / max(1, len(detect)), 4),
        # bounded in-kernel waits of the single-pass kernels' partial-sum exchange that ever expired (must be 0)
        'exchange_timeouts': _exchange_timeouts(),
        'world_size_seen': (_td.get_world_size() if tdist_on else 1),
        'rccl_ranks_seen': ranks_seen,                     # measured by a collective, not read from the environment
        'ranks_seen_backend': (_td.get_backend() if tdist_on else None),      # what counted them: 'nccl' (= RCCL) | 'gloo' | None (no process group)
        # after the find phase every rank ran the same forward + backward (rank 0's batch, broadcast weights): identical
        # gradients bit for bit / within 1e-5 of scale (None on one GPU)
        'ranks_agree_bitwise': agree, 'ranks_agree_1e-5': agree_tol, 'ranks_agree_detail': agree_detail, 'same_rank_repeat_agrees': repeat,
        # the same pass counted per kernel family of this library (profile slots): equal tables = every rank routed every
        # layer to the same kernels; own_conv_policy 'all' (set for N > 1) leaves only the stem forward and the classifier
        # GEMM to the vendor library
        'ranks_same_kernel_census': census_same, 'kernel_census_rank0': census, 'own_conv_policy': own_policy,
        'find_phase_exchange_timeouts': find_timeouts, 'find_phase_s': round(find_s, 1),
        'exchange_us_exposed': None if exposed_us is None else round(exposed_us, 1),
        'config': {'workload': ('%s V%s passport (%s_passport.json: %d passport layers), '
                                '%d classes, 3x%dx%d, batch %d/GPU, SGD(0.01,0.9,wd1e-4)' %
                                ({'resnet18': 'ResNet18', 'resnet50': 'ResNet50', 'alexnet': 'AlexNet'}[args.arch],
                                 {1: '1', 2: '2 private', 3: '3 private + trigger pair (2 extra images per step, not counted in img/s)'}[args.scheme], args.arch, len(elems), args.classes,
                                 args.image_size, args.image_size, args.batch)) + (
                                    '' if args.norm_type == 'bn' else ', norm_type ' + args.norm_type) + (
                                    ', library norm kernels (--no-fuse)' if args.no_fuse else ''),
                   'global_batch': args.gpus * args.batch, 'parallelism': 'dp%d' % args.gpus,
                   # 'fp32' (default): fp32 operands, fp32 MFMA, fp32 results everywhere.  'bf16x3' (DEEPIPR_CONV_ARITH=bf16x3,
                   # opt-in): the 3x3 stride-1 weight gradients multiply on the bf16 matrix cores with each fp32 operand split
                   # EXACTLY into three bf16 words and six of the nine products kept -- one fp32 rounding per product, as
                   # accurate against float64 as the fp32 MFMA kernel (profiles/r04_wgrad_bench_bf16x3.json)
                   'conv_arithmetic': _conv_arith(),
                   'optimizer': 'DDP+torch fused SGD' if args.ddp else 'FlatSGD (flat buffers; RCCL all-reduce of the gradient buckets between the replayed backward stages, or from gradient hooks with --eager)',
                   'exchange': stage_plan,
                   'passport_layers': len(elems), 'fused_norm_layers': len(all_elems),
                   'launch': (('hipGraph replay of %s; kernel timing from %d eager steps right after the timed region'
                               % ('the whole step' if not tdist_on else ('zero_grad..backward, then one eager '
                                  'all-reduce + fused SGD' if launch_form.startswith('unstaged') else 'the staged step '
                                  '(backward stages captured back to back, bucket all-reduces on a side stream '
                                  'behind each stage\'s event, fused SGD)'), sampled)
                               + ('; in-launch exchange switched off after a time-out' if 'exchange off' in launch_form else ''))
                              if use_graph else 'eager')},
    }
    STREAMING = {'gn_bwd': 'GroupNorm/InstanceNorm+affine+ReLU backward, register-resident (12 B/elt)',
                 'gn_fwd': 'GroupNorm/InstanceNorm+affine+ReLU forward, register-resident (8 B/elt)',
                 'bn_res_bwd': 'single-pass norm+affine+ReLU backward: read dy + x once, write dx (12 B/elt; 16 with a second incoming gradient; 20-24 with a folded residual tail)',
                 'bn_res_fwd': 'single-pass norm+affine+ReLU forward: read x once, write y (8 B/elt; 12 with a folded residual tail)',
                 'bn_affine_bwd': 'norm+affine+ReLU backward apply pass: read dy + x, write dx (12 B/elt)',
                 'bn_affine_fwd': 'norm+affine+ReLU forward apply pass (8 B/elt)',
                 'bn_bwd_reduce': 'backward channel sums (8 B/elt)', 'bn_stats': 'batch statistics (4 B/elt)',
                 'affine_bwd': 'affine backward: read dy + xhat, write dxhat (12 B/elt)',
                 'affine_fwd': 'affine forward (8 B/elt)', 'sgd': 'fused SGD over the flat buffers (20 B/param)',
                 'maxpool': '3x3 stride-2 max-pool of the ImageNet stem, forward / backward with a one-byte argmax (4 B per input + 5 B per output element)',
                 'resample2': 'stride-2 pixel gather / zero-interleaving scatter around the 1x1 stride-2 convolutions',
                 # the kernel pair the north star names: gamma / beta of ALL passport layers in one launch (W read once,
                 # 4 B/weight) and the rank-2 update accumulated into each layer's wgrad (8 B/weight, one launch per layer)
                 'gamma_beta_fwd': 'passport GEMV, all passport layers in one launch: gamma, beta = W . pooled keys (4 B/weight)',
                 'gamma_beta_bwd': 'passport rank-2 update accumulated into the conv wgrad (8 B/weight)',
                 # pre-transformed form of the Winograd forward / backward-data kernels: ONE launch per step writes the images
                 # G g G^T of every 3x3 stride-1 weight (36 B in, 66 B out per filter and direction)
                 'conv_wino_weights': 'Winograd weight transform, all 3x3 stride-1 layers in one launch (36 B in + 2 x 66 B out per filter)'}
    NOT_DOMINANT = ('sgd', 'gamma_beta_fwd', 'gamma_beta_bwd', 'conv_wino_weights', 'maxpool', 'resample2')
    hbm = None
    if out['exchange_timeouts']:
        out['roofline_refused'] = ('an in-launch exchange of the single-pass kernels timed out (%d buffer(s)): their '
                                   'outputs were poisoned; no roofline is reported for this run' % out['exchange_timeouts'])
    if timing and not out['exchange_timeouts'] and any(prof.get(k, (0, 0))[1] > 0 for k in STREAMING):
        # Durations come from start/stop events attached to each kernel's own dispatch (hipExtLaunchKernelGGL):
        # kernel execution time, comparable with rocprofv3's kernel trace.  The library also accounts the
        # algorithmic bytes of every timed launch (deepipr_profile_read_bytes), so achieved = bytes / kernel time
        # summed over exactly the launches that were timed.
        kern = {}
        for name in STREAMING:
            ms, n = prof.get(name, (0.0, 0))
            if n:
                nbytes = prof_bytes.get(name, 0.0)
                kern[name] = {'launches_per_step': round(n / sampled, 1), 'avg_us': round(1000.0 * ms / n, 3),
                              'us_per_step': round(1000.0 * ms / sampled, 1),
                              'bytes_per_step': int(nbytes / sampled),
                              'GBps': round(nbytes / (ms * 1e-3) / 1e9, 1)}
                kern[name]['frac'] = round(kern[name]['GBps'] / HBM_PEAK_GBS, 4)
        for name in ('passport_bwd_finish', 'reduce_partials'):
            ms, n = prof.get(name, (0, 0))
            if n:
                kern[name] = {'launches_per_step': round(n / sampled, 1), 'avg_us': round(1000.0 * ms / n, 3),
                              'us_per_step': round(1000.0 * ms / sampled, 1)}
        # HBM side: the passport / norm streaming kernel with the most time per step
        dom = max((k for k in kern if k in STREAMING and k not in NOT_DOMINANT), key=lambda k: kern[k]['us_per_step'])
        a = kern[dom]
        per_launch = a['bytes_per_step'] / max(1.0, a['launches_per_step'])
        # PMC traffic: rocprofv3 --pmc passes over this very command (tools/gpu_pmc_in_situ.sh, summarised into
        # profiles/pmc_traffic.json: in_situ_per_launch), quoted when that record describes this launch mix
        pmc_bytes = pmc_traffic('k_' + dom, 'in_situ_per_launch', per_launch)
        hbm = {'bound': 'hbm', 'kernel': 'k_%s (%s)' % (dom, STREAMING[dom]),
               'achieved': a['GBps'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': a['frac'],
               'traffic': pmc_bytes,
               # PMC counters need their own rocprofv3 passes: the figure is the committed record of such
               # passes over this command, quoted only when this run's launch mix equals the recorded one
               'traffic_source': 'profiles/pmc_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)' if pmc_bytes else None,
               'bytes_per_launch': int(per_launch), 'avg_us': a['avg_us'],
               'launches_per_step': a['launches_per_step'],
               'note': '%d fused norm layer calls per step over activations of %.1f-%.1f MB (%d passport '
                       'layer calls of %.1f MB among them); bytes and time summed over the timed '
                       'launches' % (len(all_elems), 4 * min(all_elems) / 1e6, 4 * max(all_elems) / 1e6,
                                     len(elems), 4 * float(np.mean(elems)) / 1e6)}
        # MFMA side: the convolution kernels (their profile slots account ALGORITHMIC FLOPs); the one with the most time
        # per step is `mfma`, all of them are listed in roofline_mfma_kernels
        MFMA_SLOTS = {
            'conv_wgrad_b3': ('k_conv3x3_wgrad_b3 (weight gradient of the 3x3 stride-1 data convolutions on '
                              'v_mfma_f32_32x32x16_bf16: fp32 operands split exactly into three bf16 words, six of the nine '
                              'products, fp32 accumulation; split-K partial tiles summed by k_conv_wgrad_reduce, timed apart)', 6.0),
            'conv_wgrad': ('k_conv3x3_wgrad / k_conv1x1s2_wgrad / k_conv_stem_wgrad (the other weight gradients, on '
                           'v_mfma_f32_32x32x2_f32)', 1.0),
            'conv1x1_wgrad': ('k_conv1x1_wgrad (weight gradient of the 1x1 stride-1 data convolutions -- the Bottleneck blocks of config 5 -- '
                              'on v_mfma_f32_32x32x2_f32, NCHW operands as they lie; split-K partial tiles summed by k_conv_wgrad_reduce)', 1.0),
            'conv1x1_fwd': ('k_conv1x1_gemm (forward of the 1x1 stride-1 data convolutions: one GEMM over the flattened (image, pixel) '
                            'positions of NCHW, on v_mfma_f32_32x32x2_f32)', 1.0),
            'conv1x1_dgrad': ('k_conv1x1_gemm (backward-data of the 1x1 stride-1 data convolutions, weights read transposed in place)', 1.0),
            'conv_fwd': ('k_conv_gemm (forward of the data convolutions it owns, on v_mfma_f32_32x32x2_f32)', 1.0),
            'conv_dgrad': ('k_conv_gemm / k_conv_dgrad_s2x4 (backward-data, on v_mfma_f32_32x32x2_f32)', 1.0),
            # the Winograd slots account EXECUTED FLOPs (2 * M * C * 16 per 2x2 output tile = the direct sum's / 2.25)
            'conv_wino_fwd': ('k_conv_wino (forward of the 3x3 stride-1 data convolutions: Winograd F(2x2, 3x3) around '
                              'v_mfma_f32_32x32x2_f32)', 1.0),
            'conv_wino_dgrad': ('k_conv_wino (backward-data of the 3x3 stride-1 data convolutions: Winograd F(2x2, 3x3) around '
                                'v_mfma_f32_32x32x2_f32)', 1.0),
            'conv_wino_wgrad': ('k_conv_wino_wgrad (weight gradient of the 3x3 stride-1 data convolutions: Winograd F(3x3, 2x2) '
                                'around v_mfma_f32_32x32x2_f32; split-K partial tiles summed by k_conv_wgrad_reduce, timed apart)', 1.0)}
        rms, rn = prof.get('conv_wgrad_reduce', (0.0, 0))
        mfma, mfma_all = None, {}
        for slot, (what, issued) in MFMA_SLOTS.items():
            ms, n = prof.get(slot, (0.0, 0))
            if not n:
                continue
            flops = prof_bytes.get(slot, 0.0)
            tf = flops / (ms * 1e-3) / 1e12                  # algorithmic
            peak = MFMA_BF16_PEAK_TFLOPS if issued > 1 else MFMA_F32_PEAK_TFLOPS
            rec = {'bound': 'mfma', 'kernel': what, 'achieved': round(tf * issued, 1), 'peak': peak, 'unit': 'TFLOP/s',
                   'frac': round(tf * issued / peak, 4), 'traffic': None, 'traffic_source': None,
                   'algorithmic_TFLOPs': round(tf, 1), 'mfma_flops_per_algorithmic_flop': issued,
                   'flops_per_launch': int(flops / n), 'avg_us': round(1000.0 * ms / n, 2),
                   'launches_per_step': round(n / sampled, 1), 'us_per_step': round(1000.0 * ms / sampled, 1),
                   'note': 'algorithmic FLOPs = 2 * Co * Ci * taps * N * OH * OW per launch (SURVEY.md 8(d)), summed over the '
                           'timed launches / their summed kernel time; `achieved` = the FLOPs the matrix cores execute'}
            if slot.startswith('conv_wino'):
                rec['direct_equivalent_TFLOPs'] = round(2.25 * tf, 1)
                rec['note'] = ('EXECUTED FLOPs = 2 * M * C * 16 * N * (H / 2) * (W / 2) per launch: F(2x2, 3x3) spends 16 MFMA '
                               'multiplies where the direct sum spends 36 (direct_equivalent_TFLOPs = the rate a direct kernel '
                               'would need for the same launch time; it may exceed the fp32 MFMA peak)')
            if issued > 1:
                rec['algorithmic_over_fp32_mfma_peak'] = round(tf / MFMA_F32_PEAK_TFLOPS, 4)
                rec['note'] += (' (6 bf16 MFMA products per fp32 product) against the dense bf16 peak; the algorithmic rate is '
                                'also given against the fp32 MFMA peak (%.1f TFLOP/s), which an fp32-MFMA kernel cannot exceed'
                                % MFMA_F32_PEAK_TFLOPS)
            if slot.startswith('conv_wgrad') or slot == 'conv_wino_wgrad':
                rec['reduce_us_per_step_all_wgrads'] = round(1000.0 * rms / sampled, 1)
            if slot.startswith('conv_wino') and is_headline(args):
                # HBM-side bytes per launch (PMC FETCH_SIZE + WRITE_SIZE, separate rocprofv3 passes over the eager step of
                # this configuration, profiles/pmc_traffic.json, round 5; Infinity-Cache hits are counted by these counters)
                fam = 'k_conv_wino_wgrad' if slot == 'conv_wino_wgrad' else 'k_conv_wino'
                rec['traffic'] = pmc_traffic(fam, 'in_situ_per_launch')
                # the record is a committed measurement, not one of this run: it names the kernel source it was taken on, and a
                # source that has changed since is said so
                rec['traffic_stale'] = pmc_record_stale(fam, 'in_situ_per_launch')
                rec['traffic_source'] = 'profiles/pmc_traffic.json: %s (committed record, average over the family\'s launches of the step)' % fam
            mfma_all[slot] = rec
        if mfma_all:
            mfma = max(mfma_all.values(), key=lambda r: r['us_per_step'])
            out['roofline_mfma_kernels'] = mfma_all
        # `roofline` = the dominant hand-written kernel of the step by time; both sides are always reported
        if mfma is not None and mfma['us_per_step'] >= a['us_per_step']:
            out['roofline'], out['roofline_hbm'] = mfma, hbm
        else:
            out['roofline'] = hbm
            if mfma is not None:
                out['roofline_mfma'] = mfma
        # the passport-affine kernels ON THEIR OWN (north star: ">= 60 % HBM roofline on the passport-affine kernel"): only
        # the launches that serve passport layers -- norm + passport affine + ReLU (+ sign loss, + folded residual tail),
        # 8 B / element forward, 12 (20 - 24 with a tail) backward
        ps = {}
        for name in ('bn_res_fwd', 'bn_res_bwd', 'bn_affine_fwd', 'bn_affine_bwd', 'bn_stats', 'bn_bwd_reduce', 'gn_fwd',
                     'gn_bwd', 'affine_fwd', 'affine_bwd'):
            pms, pn, pb = prof_scope.get(name, (0.0, 0, 0.0))
            if pn:
                ps[name] = {'launches_per_step': round(pn / sampled, 1), 'avg_us': round(1000.0 * pms / pn, 3),
                            'us_per_step': round(1000.0 * pms / sampled, 1), 'bytes_per_launch': int(pb / pn),
                            'GBps': round(pb / (pms * 1e-3) / 1e9, 1)}
        if ps:
            tot_ms = sum(prof_scope[k][0] for k in ps)
            tot_b = sum(prof_scope[k][2] for k in ps)
            gbps = tot_b / (tot_ms * 1e-3) / 1e9
            # What the SIZE allows: no kernel of this step, however little it does, completes faster than the
            # shortest ones of this very run (one-workgroup finishers: launch + one memory round trip).  A launch that moves B bytes
            # cannot beat floor + B / (the copy rate the chip sustains): the fraction of the 8 TB/s peak that bound leaves
            floor_us = min(v['avg_us'] for v in kern.values() if v.get('launches_per_step'))
            ceil_b = sum(v['bytes_per_launch'] * v['launches_per_step'] for v in ps.values())
            ceil_us = sum((floor_us + v['bytes_per_launch'] / (HBM_ACHIEVABLE_GBS * 1e3)) * v['launches_per_step'] for v in ps.values())
            out['roofline_passport'] = {
                'launch_floor_us': round(floor_us, 2),
                'frac_ceiling_at_this_size': round(ceil_b / (ceil_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                'ceiling_note': 'shortest kernel of this run (launch floor) + bytes at the %.1f TB/s a copy sustains, per launch: '
                                'the fraction of the HBM peak a launch of this size can reach at all' % (HBM_ACHIEVABLE_GBS / 1e3),
                'bound': 'hbm', 'kernel': 'the norm + passport affine + ReLU launches of the %d passport layer calls only '
                '(forward and backward together)' % len(elems),
                'achieved': round(gbps, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(gbps / HBM_PEAK_GBS, 4),
                'traffic': None, 'us_per_step': round(1000.0 * tot_ms / sampled, 1),
                'activation_MB': round(4 * float(np.mean(elems)) / 1e6, 2), 'kernels': ps,
                'note': 'latency-bound: %.1f MB activations, a launch moves %.0f-%.0f MB in %.1f-%.1f us'
                        % (4 * float(np.mean(elems)) / 1e6, min(v['bytes_per_launch'] for v in ps.values()) / 1e6,
                           max(v['bytes_per_launch'] for v in ps.values()) / 1e6,
                           min(v['avg_us'] for v in ps.values()), max(v['avg_us'] for v in ps.values()))}
        out['kernels'] = kern
        if args.gpus == 1 and not args.no_stress:
            out['roofline_stress'] = stress_roofline(device)
            out['roofline_stress']['traffic'] = pmc_traffic(out['roofline_stress']['kernel'], 'S3[512,512,8,8]')
    if args.gpus == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(args)
    # only the full default line carries them: the lean forms the tools use (--no-stress / --no-cpu-baseline, rocprofv3 runs) do not
    if args.gpus == 1 and not (args.no_configs or args.no_stress or args.no_cpu_baseline) and is_headline(args):
        out['configs'] = other_configs()
    print(json.dumps(out), flush=True)
    _leave()


if __name__ == '__main__':
    main()
