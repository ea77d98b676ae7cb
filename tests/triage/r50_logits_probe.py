import json, os, sys
import numpy as np, torch
sys.path.insert(0, '/root/repo')
os.chdir('/root/repo')
from tests import test_models_gpu as T
from oracle import torch_ref, patterns
from deepipr_amd.experiments.utils import construct_passport_kwargs_from_dict
from deepipr_amd.models.resnet_passport import ResNet50Passport
cfg = json.load(open('passport_configs/resnet50_passport.json'))
kw = construct_passport_kwargs_from_dict({'passport_config': cfg, 'norm_type': 'bn', 'key_type': 'random', 'sl_ratio': T.ALPHA})
for n in (8, 32):
    torch.manual_seed(0); np.random.seed(0)
    prod = ResNet50Passport(num_classes=1000, passport_kwargs=kw).to('cuda:0')
    ref = torch_ref.resnet50_ref(num_classes=1000, passport_kwargs=torch_ref.passport_kwargs_from_config(cfg, 'bn', 'random', T.ALPHA))
    x, y = patterns.batch(n, 3, 224, 224, 1000)
    prod.train(); ref.train()
    with torch.no_grad():
        prod(x[:2].to('cuda:0')); ref(x[:2])
    patterns.fill_state(prod); patterns.fill_state(ref)
    for m in prod.modules():
        if hasattr(m, 'invalidate_key_cache'): m.invalidate_key_cache()
    ref64 = ref.double().to('cuda:0')
    with torch.no_grad():
        op = prod(x.to('cuda:0')); orf = ref64(x.to('cuda:0').double())
    sc = float(orf.abs().max())
    print('n', n, 'logits err/scale vs float64 oracle on GPU: %.2e (scale %.3g)' % (float((op.double() - orf).abs().max()) / sc, sc), flush=True)
