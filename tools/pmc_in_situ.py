#!/usr/bin/env python
"""In-situ HBM traffic of the hand-written kernels inside the bench.py train step.

Input: the per-dispatch counter CSVs of two SEPARATE rocprofv3 passes over the same command
    rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- python bench.py --eager --no-kernel-timing ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace ... -- python bench.py --eager --no-kernel-timing ...
(MI355X_MICROARCH.md, HBM / rocprofv3 section: the two counters do not fit one pass; both are reported in KB;
FETCH_SIZE under-reports a wide coalesced read stream by half on gfx950 and is doubled here).  Every step of the
job launches the same mix, so the average over ALL dispatches of a kernel family is its in-situ per-launch traffic.

    python tools/pmc_in_situ.py FETCH.csv WRITE.csv [bench.json] > in_situ.json
With the JSON line of an un-profiled bench.py run of the same command, the algorithmic bytes per launch that
bench.py accounted are put next to the measured ones."""
import collections
import csv
import json
import re
import sys


def family(name):
    name = name.replace('void ', '').replace('(anonymous namespace)::', '')
    m = re.match(r'(k_[a-z0-9_]+)', name)
    if m and m.group(1).startswith('k_sgd_momentum'):
        return 'k_sgd'
    if m and m.group(1).startswith('k_bn_dual_'):        # a projection block's two norm layers in one launch: accounted
        return 'k_bn_res_' + m.group(1)[len('k_bn_dual_'):]   # by the library (and bench.py) with the k_bn_res_* family
    return m.group(1) if m else None


def per_family(path):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(path)):
        fam = family(row['Kernel_Name'])
        if fam is None:
            continue
        acc[fam][0] += float(row['Counter_Value'])
        acc[fam][1] += 1
    return acc


def main():
    fetch, write = per_family(sys.argv[1]), per_family(sys.argv[2])
    bench = json.load(open(sys.argv[3])) if len(sys.argv) > 3 else {}
    kern = bench.get('kernels', {})
    out = {}
    for fam in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(fam, [0.0, 0])
        w, nw = write.get(fam, [0.0, 0])
        rec = {'fetch': round(2.0 * 1024.0 * f / max(nf, 1), 1), 'write': round(1024.0 * w / max(nw, 1), 1),
               'dispatches_fetch_pass': nf, 'dispatches_write_pass': nw}
        rec['total'] = round(rec['fetch'] + rec['write'], 1)
        k = kern.get(fam[2:])
        if k and k.get('launches_per_step') and 'bytes_per_step' in k:
            rec['algorithmic'] = round(k['bytes_per_step'] / k['launches_per_step'], 1)
            if rec['algorithmic']:
                rec['traffic_over_algorithmic'] = round(rec['total'] / rec['algorithmic'], 4)
        out[fam] = rec
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
