"""train_v1.py / train_v23.py keep the reference's command-line surface and drive the trainers end to end
(CPU here, with the oracle-backed kernels patched in; `--device cpu` exists for exactly this)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def cpu_kernels(monkeypatch):
    from deepipr_amd import passport_ops
    from tests.oracle_kernels import OracleKernels
    monkeypatch.setattr(passport_ops, 'kernels', OracleKernels())
    monkeypatch.chdir(ROOT)


def test_reference_flags_are_accepted():
    from deepipr_amd.experiments.cli import make_parser
    ref_flags = ['--arch', 'resnet', '--batch-size', '32', '--epochs', '3', '--lr', '0.1', '--dataset', 'cifar100',
                 '--norm-type', 'gn', '--key-type', 'image', '--sign-loss', '0.5', '--use-trigger-as-passport',
                 '--train-passport', '--train-backdoor', '--train-private', '--pretrained-path', 'x.pth',
                 '--lr-config', 'lr_configs/imagenet.json', '--passport-config', 'passport_configs/resnet18_passport.json',
                 '--save-interval', '2', '--eval', '--exp-id', '7', '--tag', 't', '--transfer-learning',
                 '--tl-dataset', 'caltech-101', '--tl-scheme', 'ftal']
    for private in (False, True):
        a = make_parser(private).parse_args(ref_flags)
        assert a.arch == 'resnet' and a.batch_size == 32 and a.tl_scheme == 'ftal' and a.exp_id == 7
    assert make_parser(False).parse_args([]).train_private is False
    assert make_parser(True).parse_args([]).train_private is True          # train_v23.py:42-43
    assert make_parser(False).parse_args([]).key_type == 'shuffle'         # train_v1.py:31
    with pytest.raises(SystemExit):
        make_parser(True).parse_args(['--arch', 'resnet9'])                # train_v23.py:13 drops resnet9


def test_train_v1_shuffle_keys_end_to_end(cpu_kernels, tmp_path):
    sys.path.insert(0, ROOT)
    import train_v1
    out = train_v1.main(['--arch', 'alexnet', '--train-passport', '--key-type', 'shuffle', '--epochs', '1',
                         '--batch-size', '4', '--synthetic-samples', '8', '--device', 'cpu',
                         '--logdir', str(tmp_path)])
    h = out['history'][0]
    assert h['train_sign_loss'] > 0 and 0.0 <= h['train_sign_acc'] <= 1.0
    assert os.path.exists(os.path.join(out['logdir'], 'models', 'last.pth'))
    assert os.path.exists(os.path.join(out['logdir'], 'history.csv'))
    import torch
    sd = torch.load(os.path.join(out['logdir'], 'models', 'last.pth'))
    assert tuple(sd['features.4.key'].shape) == (1, 192, 8, 8) and tuple(sd['features.6.skey'].shape) == (1, 256, 8, 8)


def test_train_v23_backdoor_end_to_end(cpu_kernels, tmp_path):
    sys.path.insert(0, ROOT)
    import train_v23
    out = train_v23.main(['--arch', 'alexnet', '--train-backdoor', '--key-type', 'random', '--epochs', '1',
                          '--batch-size', '4', '--synthetic-samples', '8', '--dataset', 'cifar100', '--device', 'cpu',
                          '--logdir', str(tmp_path)])
    h = out['history'][0]
    assert 'train_acc_public' in h and 'train_acc_private' in h and 'valid_total_acc' in h
    assert any(k.startswith('valid_s_private_features.') for k in h)
    assert out['logdir'].endswith(os.path.join('alexnet_cifar100_v3', '1'))


def test_experiment_ids_best_checkpoint_and_eval(cpu_kernels, tmp_path, capsys):
    """experiments/base.py:76-83,110-150: a run takes the smallest unused experiment id, writes best.pth next to
    last.pth, and --eval --exp-id N evaluates that run's best.pth (warning when there is none)."""
    sys.path.insert(0, ROOT)
    import train_v1
    common = ['--arch', 'alexnet', '--train-passport', '--key-type', 'random', '--batch-size', '4',
              '--synthetic-samples', '8', '--device', 'cpu', '--logdir', str(tmp_path)]
    first = train_v1.main(common + ['--epochs', '1'])
    second = train_v1.main(common + ['--epochs', '1'])
    assert first['logdir'].endswith(os.path.join('alexnet_cifar10_v1', '1'))
    assert second['logdir'].endswith(os.path.join('alexnet_cifar10_v1', '2'))
    for f in ('best.pth', 'last.pth'):
        assert os.path.exists(os.path.join(second['logdir'], 'models', f))
    res = train_v1.main(common + ['--eval', '--exp-id', '2'])
    assert set(res) >= {'loss', 'acc'}
    assert 'No such Experiment' not in capsys.readouterr().out
    train_v1.main(common + ['--eval', '--exp-id', '9'])
    assert 'No such Experiment' in capsys.readouterr().out


def test_use_trigger_as_passport_switches_the_passport_images(cpu_kernels, tmp_path):
    """--use-trigger-as-passport (experiments/classification.py:37-40): the passport images are drawn from the trigger
    set instead of the validation set, so the keys -- activations of those images in the plain net -- differ, with
    everything else (seeds, weights, python `random` draws) equal."""
    sys.path.insert(0, ROOT)
    import torch
    import train_v1
    common = ['--arch', 'alexnet', '--train-passport', '--key-type', 'image', '--epochs', '1', '--batch-size', '4',
              '--synthetic-samples', '8', '--device', 'cpu', '--logdir', str(tmp_path)]
    import random

    def run(extra):
        random.seed(5)                            # passport_generator.get_key draws the image indices from `random`
        return train_v1.main(common + extra)
    a, b, c = run([]), run(['--use-trigger-as-passport']), run([])
    ka, kb, kc = (torch.load(os.path.join(o['logdir'], 'models', 'last.pth'))['features.4.key'] for o in (a, b, c))
    assert tuple(ka.shape) == (1, 192, 8, 8)
    assert torch.equal(ka, kc)                    # same flags, same keys
    assert not torch.equal(ka, kb)                # other passport images, other keys
