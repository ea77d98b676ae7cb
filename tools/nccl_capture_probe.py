#!/usr/bin/env python
"""Gate for distributed.retire_collectives(): N cycles of {RCCL all-reduce, hipGraph capture right behind it} in ONE process.

    python tools/nccl_capture_probe.py --cycles 300 --mode retire      # the product's guard: must print ok 300/300
    python tools/nccl_capture_probe.py --cycles 300 --mode none        # no guard: the watchdog's poll lands in a capture and
                                                                       # terminates the process (shows the probe is sensitive)

RCCL's watchdog polls the end event of every collective it has not retired yet about every 100 ms; a poll inside a stream
capture of the main thread returns hipErrorCapturedEvent and the watchdog aborts the process (ROCm 7.2 / torch 2.10).  Each
cycle issues three collectives and captures a burst of small kernels immediately afterwards.  A world of one is enough: the
work items and the watchdog are the same.
Measured (profiles/r05_nccl_capture_probe.txt): --mode retire 300 / 300 in 36.7 s -- 122 ms per cycle, i.e. the call really
blocks until the watchdog's next poll has emptied its list -- and --mode none ALSO 300 / 300 (5.1 s): this in-process loop does
not reproduce the abort, which needs the real step's capture (tools/nccl_flake_probe.sh: fresh processes of the one-rank
exchange test, 6 of 78 aborted before any guard).  So this tool shows what the guard costs and that it returns; the
fresh-process probe stays the gate for whether it is sufficient."""
import argparse
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepipr_amd import distributed as D           # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--cycles', type=int, default=300)
ap.add_argument('--mode', default='retire', choices=['retire', 'none', 'sleep'])
ap.add_argument('--ops', type=int, default=1500, help='small kernels per capture')
args = ap.parse_args()
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', str(29700 + os.getpid() % 200))
dist.init_process_group('nccl', rank=0, world_size=1)
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
x = torch.ones(1 << 16, device=dev)
t = torch.ones(1 << 20, device=dev)
side = torch.cuda.Stream()
done = 0
t0 = time.perf_counter()
for i in range(args.cycles):
    for _ in range(3):
        dist.all_reduce(t)
    if args.mode == 'retire':
        D.retire_collectives()
    elif args.mode == 'sleep':
        torch.cuda.synchronize()
        time.sleep(0.5)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode='thread_local'):
        y = x
        for _ in range(args.ops):
            y = y + 1.0
    g.replay()
    torch.cuda.synchronize()
    done += 1
    print('cycle %d ok' % done, file=sys.stderr, flush=True) if done % 50 == 0 else None
print('{"mode": "%s", "ok": %d, "cycles": %d, "seconds": %.1f}' % (args.mode, done, args.cycles, time.perf_counter() - t0), flush=True)
os._exit(0)
