"""How much of the step is host (enqueue) time?  Eager, R config."""
import os, sys, time, argparse
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from deepipr_amd.experiments.trainer import train_step_v1
args = argparse.Namespace(batch=128, classes=10, scheme=1)
dev = torch.device('cuda:0')
torch.backends.cudnn.benchmark = True
xs = [torch.randn(128, 3, 32, 32, device=dev) for _ in range(4)]
ys = [torch.randint(0, 10, (128,), device=dev) for _ in range(4)]
for fused in (False, True):
    model = bench.build_model(args, dev); model.train()
    with torch.no_grad(): model(xs[0])
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4, fused=fused)
    for i in range(15): train_step_v1(model, opt, xs[i % 4], ys[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(40): train_step_v1(model, opt, xs[i % 4], ys[i % 4])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('fused_sgd=%s enqueue %.3f ms/step, wall %.3f ms/step' % (fused, (t1 - t0) / 40 * 1e3, (t2 - t0) / 40 * 1e3), flush=True)
# host cost of the passport op wrappers alone
from deepipr_amd.passport_ops import kernels as K
x = torch.randn(128, 512, 4, 4, device=dev); dy = torch.randn_like(x)
w = torch.randn(512, 4608, device=dev); m = torch.rand(2, 4608, device=dev, dtype=torch.float64)
b = torch.ones(512, device=dev); rm = torch.zeros(512, device=dev); rv = torch.ones(512, device=dev)
nbt = torch.zeros((), dtype=torch.int64, device=dev); dl = torch.ones((), device=dev)
torch.cuda.synchronize()
import cProfile, pstats
def loop():
    for _ in range(200):
        out = K.passport_bn_fwd(x, w, m, None, None, b, 0.1, True, rm, rv, nbt, 0.1, 1e-5, True)
        K.passport_bn_bwd(dy, x, out[1], m, b, 0.1, dl, None, None, (512, 4608), True, True)
t0 = time.perf_counter(); loop(); t1 = time.perf_counter(); torch.cuda.synchronize()
print('passport_bn fwd+bwd wrapper host time: %.1f us per pair' % ((t1 - t0) / 200 * 1e6))
pr = cProfile.Profile(); pr.enable(); loop(); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(12)
