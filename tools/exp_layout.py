"""Experiment: plain ResNet18 (library ops only) train step, NCHW vs channels_last, fp32, batch 128."""
import os, sys, time
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepipr_amd.models.resnet_normal import ResNet18
torch.backends.cudnn.benchmark = True
dev = 'cuda:0'
for fmt in (torch.contiguous_format, torch.channels_last):
    torch.manual_seed(0)
    m = ResNet18(num_classes=10).to(dev).to(memory_format=fmt)
    opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    x = torch.randn(128, 3, 32, 32, device=dev).contiguous(memory_format=fmt)
    y = torch.randint(0, 10, (128,), device=dev)
    def step():
        opt.zero_grad(set_to_none=True)
        F.cross_entropy(m(x), y).backward()
        opt.step()
    for _ in range(15): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40): step()
    torch.cuda.synchronize()
    print(fmt, '%.3f ms/step' % ((time.perf_counter() - t0) / 40 * 1000), flush=True)
