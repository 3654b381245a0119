#!/usr/bin/env python3
"""Condense rocprofv3 outputs (kernel stats CSV + PMC counter CSVs) into a
small text summary that is committed under profiles/."""
import csv
import glob
import os
import sys


def find(root, pattern):
    return sorted(glob.glob(os.path.join(root, '**', pattern), recursive=True))


def main():
    out = sys.argv[1]
    for f in find(os.path.join(out, 'prof_trace'), '*kernel_stats.csv'):
        print('== kernel stats:', os.path.relpath(f, out))
        rows = list(csv.DictReader(open(f)))
        for r in rows[:20]:
            print('  %-60s calls=%s total_ns=%s avg_ns=%s pct=%s' % (
                r.get('Name', '')[:60], r.get('Calls'), r.get('TotalDurationNs'), r.get('AverageNs'),
                r.get('Percentage')))
    for name in ('prof_fetch', 'prof_write'):
        for f in find(os.path.join(out, name), '*counter_collection.csv'):
            print('== counters:', os.path.relpath(f, out))
            agg = {}
            for r in csv.DictReader(open(f)):
                k = (r.get('Kernel_Name', '')[:50], r.get('Counter_Name'))
                v = float(r.get('Counter_Value', 0) or 0)
                a = agg.setdefault(k, [0, 0.0])
                a[0] += 1
                a[1] += v
            for (kn, cn), (cnt, tot) in sorted(agg.items(), key=lambda x: -x[1][1])[:16]:
                print('  %-50s %-12s dispatches=%d sum=%.0f avg=%.1f' % (kn, cn, cnt, tot, tot / cnt))


if __name__ == '__main__':
    main()
