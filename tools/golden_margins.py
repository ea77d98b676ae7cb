#!/usr/bin/env python
"""How far the HIP path sits from the reference goldens, per key family: worst max|got - gold| / max(1, max|gold|) over all
golden model cases (fused norm on / off).  The bars of tests/test_parity_gpu.py::test_model_cases_match_reference_goldens
are set from this (x 10).   python tools/golden_margins.py [--json out.json]"""
import collections
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import runner                      # noqa: E402
from oracle.cases import CASES                 # noqa: E402
from tests.impls import ProductImpl, load_golden   # noqa: E402


def main():
    worst = collections.defaultdict(lambda: (0.0, None))
    for name in CASES:
        gold = load_golden(os.path.join(ROOT, 'tests', 'golden'), name)
        for fuse in (True, False):
            if not fuse and CASES[name]['norm'] == 'none':
                continue
            got = runner.collect(name, ProductImpl('cuda:0', fuse_norm=fuse))
            for k, b in gold.items():
                fam = k.split('/')[0]
                if fam in ('bits', 'ctor_b') or k not in got:
                    continue
                a, b = np.asarray(got[k], dtype=np.float64), np.asarray(b, dtype=np.float64)
                if a.shape != b.shape or not b.size:
                    continue
                err = float(np.abs(a - b).max()) / max(1.0, float(np.abs(b).max()))
                if err > worst[fam][0]:
                    worst[fam] = (err, '%s[%s] %s' % (name, 'fused' if fuse else 'library norm', k))
    out = {fam: {'worst': v[0], 'where': v[1]} for fam, v in sorted(worst.items())}
    for fam, v in out.items():
        print('%-14s %.2e  %s' % (fam, v['worst'], v['where']))
    if '--json' in sys.argv:
        json.dump(out, open(sys.argv[sys.argv.index('--json') + 1], 'w'), indent=1)


if __name__ == '__main__':
    main()
