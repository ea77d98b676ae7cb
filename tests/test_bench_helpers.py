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
    # config R, batch 128: 20 fused layer calls per step; 12 B/elt on 11 plain layers, 16 on the stem (two incoming
    # gradients summed in the kernel), 24 / 20 on the 7 + 1 tail layers = 1 350 565 888 algorithmic bytes per step; the
    # two plain projection blocks (layer2.0, layer3.0) run tail layer + shortcut layer as one launch at 28 instead of
    # 24 + 12 B/elt: 18 launches, 8 B x 6 291 456 elements less
    assert abs(alg * 18 - (1350565888 - 8 * 6291456)) < 1024


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
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--scheme', '3', '--dry-run', '--backend', 'gloo'],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['world_size_seen'] == 2 and rec['dry_run'] is True and rec['value'] is None
    assert rec['rccl_ranks_seen'] == 2                       # counted by an all-reduce of ones, not read from the environment


def test_self_launcher_takes_its_ranks_down_with_it(tmp_path):
    """`timeout` / Ctrl-C aimed at the one command the user typed must not leave rank processes behind: the launcher
    puts them in their own process group and forwards SIGTERM to it."""
    import signal
    import subprocess
    import sys
    import time
    script = tmp_path / 'sleeper.py'
    script.write_text('import os, sys, time\nopen(sys.argv[1] + "/pid%s" % os.environ["RANK"], "w").write(str(os.getpid()))\ntime.sleep(120)\n')
    code = ('import sys; sys.path.insert(0, %r); import bench; bench.__file__ = %r; '
            'sys.exit(bench.self_launch([%r], 2))' % (ROOT, str(script), str(tmp_path)))
    p = subprocess.Popen([sys.executable, '-c', code], cwd=ROOT)
    deadline = time.time() + 90
    while time.time() < deadline and not all((tmp_path / ('pid%d' % r)).exists() for r in (0, 1)):
        time.sleep(0.2)
    pids = [int((tmp_path / ('pid%d' % r)).read_text()) for r in (0, 1)]
    p.send_signal(signal.SIGTERM)
    p.wait(timeout=60)
    time.sleep(1.0)
    for pid in pids:
        alive = True
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            alive = False
        assert not alive, 'rank process %d survived its launcher' % pid
