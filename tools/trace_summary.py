#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace CSV over the steady-state steps only.

MIOpen's find mode (cudnn.benchmark) and the warm-up pollute whole-process statistics, so the window is
cut by a once-per-step kernel: it spans the last `--steps` occurrences of the cross-entropy forward
kernel (k_ce_finish; nll_loss_forward* for the library head), i.e. exactly that many full train steps.

    python tools/trace_summary.py <kernel_trace.csv> --steps 20 [--top 25] > profiles/<name>.md
"""
import argparse
import csv
import collections


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('csv')
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--top', type=int, default=25)
    ap.add_argument('--out-csv', default=None, help='also write kernel name, calls/step, us/step to this CSV')
    ap.add_argument('--gaps', type=int, default=0, help='also list the N largest idle gaps between consecutive kernels')
    ap.add_argument('--markers-per-step', type=int, default=1, help='occurrences of the marker kernel per train step '
                    '(2 for the dual-branch V2 / V3 step: two cross-entropy heads)')
    ap.add_argument('--marker', default=None, help='once-per-step kernel (default: k_ce_finish, else nll_loss_forward)')
    args = ap.parse_args()
    rows = []
    with open(args.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    for marker in ([args.marker] if args.marker else ['k_ce_finish', 'nll_loss_forward']):
        marks = [i for i, r in enumerate(rows) if marker in r[2]]
        if len(marks) > args.steps * args.markers_per_step:
            break
    else:
        raise SystemExit('only %d marker kernels in the trace' % len(marks))
    lo, hi = marks[-args.steps * args.markers_per_step - 1], marks[-1]
    win = rows[lo:hi]
    wall = rows[hi][0] - rows[lo][0]
    stats = collections.defaultdict(list)
    for s, e, n in win:
        stats[n].append(e - s)
    busy = sum(sum(v) for v in stats.values())
    print('# steady-state kernel summary: %d steps, %d dispatches (%.1f per step)' %
          (args.steps, len(win), len(win) / args.steps))
    print('wall per step %.3f ms, GPU busy per step %.3f ms (%.1f %% of wall)\n' %
          (wall / args.steps / 1e6, busy / args.steps / 1e6, 100.0 * busy / wall))
    print('| kernel | calls/step | avg us | min us | max us | us/step | % busy |')
    print('|---|---|---|---|---|---|---|')
    order = sorted(stats.items(), key=lambda kv: -sum(kv[1]))
    shown = 0
    for n, v in order:
        mine = '(anonymous namespace)::k_' in n
        if shown >= args.top and not mine:
            continue
        shown += 1
        short = n.replace('void ', '').replace('(anonymous namespace)::', '')
        short = short[:90]
        print('| %s | %.1f | %.2f | %.2f | %.2f | %.1f | %.2f |' %
              (short, len(v) / args.steps, sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3,
               sum(v) / args.steps / 1e3, 100.0 * sum(v) / busy))
    if args.out_csv:
        with open(args.out_csv, 'w') as f:
            w = csv.writer(f)
            w.writerow(['kernel', 'calls_per_step', 'us_per_step'])
            for n, v in order:
                w.writerow([n[:160], round(len(v) / args.steps, 2), round(sum(v) / args.steps / 1e3, 2)])
    if args.gaps:
        gaps(win, args.steps, args.gaps)


def gaps(win, steps, top):
    """Idle time between consecutive kernels of the window (end of the latest-ending kernel so far -> next start)."""
    out, end = [], win[0][1]
    for (s, e, n), prev in zip(win[1:], win[:-1]):
        if s > end:
            out.append((s - end, prev[2], n))
        end = max(end, e)
    total = sum(g for g, _, _ in out)
    print('\nidle between kernels: %.1f us per step in %d gaps per step; the largest:' %
          (total / steps / 1e3, len(out) // steps))
    agg = {}
    for g, a, b in out:
        k = (a.replace('void ', '')[:60], b.replace('void ', '')[:60])
        agg.setdefault(k, []).append(g)
    for (a, b), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:top]:
        print('  %8.1f us/step  (%4.1f x %6.1f us)  after %s  before %s' %
              (sum(v) / steps / 1e3, len(v) / steps, sum(v) / len(v) / 1e3, a, b))


if __name__ == '__main__':
    main()
