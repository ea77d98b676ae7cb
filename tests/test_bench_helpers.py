"""bench.py's host-side helpers (no GPU): the PMC traffic record is quoted only for the launch mix it was measured
on, and the committed record describes the default workload."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location('bench_module', os.path.join(ROOT, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_pmc_traffic_is_quoted_only_for_the_same_launch_mix():
    bench = _bench()
    rec = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))['k_bn_res_bwd']['in_situ_per_launch']
    alg = rec['algorithmic']
    assert bench.pmc_traffic('k_bn_res_bwd', 'in_situ_per_launch', alg) == int(rec['fetch'] + rec['write'])
    assert bench.pmc_traffic('k_bn_res_bwd', 'in_situ_per_launch', alg * 1.005) is not None     # same mix, rounding
    assert bench.pmc_traffic('k_bn_res_bwd', 'in_situ_per_launch', alg * 1.2) is None           # another mix
    assert bench.pmc_traffic('k_no_such_kernel', 'in_situ_per_launch', alg) is None
    assert bench.pmc_traffic('k_bn_res_bwd', 'no_such_shape', alg) is None
    # the single pass moves every byte once: measured traffic within 2 % of the algorithmic bytes
    assert abs((rec['fetch'] + rec['write']) / alg - 1.0) < 0.02
    # config R, batch 128: 20 launches per step; 12 B/elt on 11 plain layers, 16 on the stem (two incoming gradients
    # summed in the kernel), 24 / 20 on the 7 + 1 tail layers = 1 350 565 888 algorithmic bytes per step
    assert abs(alg * 20 - 1350565888) < 1024


def test_host_cores_is_positive_and_bounded():
    bench = _bench()
    assert 1 <= bench.host_cores() <= 32


def test_bench_self_launches_n_ranks_when_typed_without_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment must start two ranks itself (the driver runs
    that command verbatim on a multi-GPU node).  --dry-run --backend gloo exercises exactly that plumbing without a
    GPU: torch.distributed.run re-execution, rendezvous on 127.0.0.1, barrier, one JSON line from rank 0."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run', '--backend', 'gloo'],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['world_size_seen'] == 2 and rec['dry_run'] is True and rec['value'] is None
