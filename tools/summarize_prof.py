#!/usr/bin/env python3
"""Condense rocprofv3 rocpd databases (kernel trace + PMC passes) into a small
text summary that is committed under profiles/.
usage: summarize_prof.py <dir with prof_trace/ prof_fetch/ prof_write/>"""
import glob
import os
import sqlite3
import sys


def dbs(root, sub):
    return sorted(glob.glob(os.path.join(root, sub, '**', '*.db'), recursive=True))


def main():
    out = sys.argv[1]
    for db in dbs(out, 'prof_trace'):
        con = sqlite3.connect(db)
        print('== rocprofv3 --kernel-trace --stats (%s): name, calls, total_us, avg_us, pct' % os.path.relpath(db, out))
        for name, calls, total, avg, pct in con.execute('select name,total_calls,total_duration,average,percentage from top_kernels'):
            print('  %-78s %5d %12.1f %10.1f %6.2f' % (name[:78], calls, total, avg, pct))
    for sub in ('prof_fetch', 'prof_write'):
        for db in dbs(out, sub):
            con = sqlite3.connect(db)
            print('== rocprofv3 --pmc (%s): kernel, counter, dispatches, avg value (KB), avg duration us' % os.path.relpath(db, out))
            q = ('select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection '
                 'group by kernel_name, counter_name order by sum(value) desc')
            for kn, cn, n, v, d in con.execute(q):
                print('  %-70s %-11s %4d %14.1f %10.1f' % (kn[:70], cn, n, v, d / 1e3))


if __name__ == '__main__':
    main()
