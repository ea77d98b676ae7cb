#!/usr/bin/env python
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) of tools/res_bench.py / kbench.py:
average counter value per launch for every hand-written kernel, keyed by (kernel, grid size), in bytes.

    python tools/pmc_summary.py FETCH.csv WRITE.csv > traffic.json

Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are reported
in KB; FETCH_SIZE under-reports a wide coalesced read stream by half on gfx950 and is doubled."""
import collections
import csv
import json
import re
import sys


def per_kernel(path):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(path)):
        name = row['Kernel_Name']
        if '(anonymous namespace)::k_' not in name and not name.startswith('k_') and ' k_' not in name:
            continue
        if 'at::native' in name:
            continue
        short = re.sub(r'\(.*$', '', name.replace('void ', '').replace('(anonymous namespace)::', ''))
        key = '%s|grid=%s|wg=%s' % (short, row['Grid_Size'], row['Workgroup_Size'])
        acc[key][0] += float(row['Counter_Value'])
        acc[key][1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}


def main():
    fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
    out = {}
    for k in sorted(set(fetch) | set(write)):
        out[k] = {'fetch': round(2.0 * 1024.0 * fetch.get(k, 0.0), 1), 'write': round(1024.0 * write.get(k, 0.0), 1)}
        out[k]['total'] = round(out[k]['fetch'] + out[k]['write'], 1)
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
