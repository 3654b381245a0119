#!/usr/bin/env python3
"""Condense rocprofv3 rocpd databases (kernel trace + PMC passes) into a small
text summary that is committed under profiles/, plus traffic.json (HBM bytes per
launch and kernel) that bench.py reports as roofline.traffic.
usage: summarize_prof.py <dir with prof_trace/ prof_fetch/ prof_write/> [the bench.py arguments of the profiled command]"""
import glob
import json
import re
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
    per_kernel = {}
    for sub in ('prof_fetch', 'prof_write'):
        for db in dbs(out, sub):
            con = sqlite3.connect(db)
            print('== rocprofv3 --pmc (%s): kernel, counter, dispatches, avg value (KB), avg duration us' % os.path.relpath(db, out))
            q = ('select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection '
                 'group by kernel_name, counter_name order by sum(value) desc')
            for kn, cn, n, v, d in con.execute(q):
                print('  %-70s %-11s %4d %14.1f %10.1f' % (kn[:70], cn, n, v, d / 1e3))
                m = re.search(r'(k_[a-z0-9_]+)', kn)
                if m:
                    e = per_kernel.setdefault(m.group(1), {'FETCH_SIZE_KB': 0.0, 'WRITE_SIZE_KB': 0.0})
                    # template variants of one kernel (k_seeds<0/1/2>...) add up: they run once per batch each
                    e[cn + '_KB'] = e.get(cn + '_KB', 0.0) + v
    if per_kernel:
        for e in per_kernel.values():
            # FETCH_SIZE/WRITE_SIZE are in KB; gfx950 tallies 128-B read requests at 64 B (MI355X_MICROARCH.md,
            # HBM section), so the fetch side is doubled; narrow requests are then over-estimated, i.e. this is an
            # upper bound of the bytes that crossed the L2 <-> fabric interface
            e['hbm_bytes_per_launch'] = int(2 * e['FETCH_SIZE_KB'] * 1024 + e['WRITE_SIZE_KB'] * 1024)
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
        import bench
        a = bench.build_parser().parse_known_args(sys.argv[2:])[0]   # the workload the passes ran (bench.py's defaults otherwise)
        tj = {'batch': a.batch, 'sent_len': a.sent_len, 'dict_entries': a.dict_entries, 'weights_exp': a.weights_exp,
              'rnn': bool(a.rnn), 'kernels': per_kernel, 'kernel_source_id': bench.kernel_source_id(),
              'note': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `python bench.py`, '
                      'avg per launch; bytes = 2 x FETCH_SIZE KB (gfx950 tallies a 128-B request at 64 B; calibrated for streaming reads by the guide and for random 4-byte gathers by tools/micro/gather_calib.hip, profiles/r02_f_gather_calib.txt) + WRITE_SIZE KB'}
        with open(os.path.join(out, 'traffic.json'), 'w') as f:
            json.dump(tj, f, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
