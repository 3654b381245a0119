#!/usr/bin/env python3
"""Static instruction histogram of one kernel's gfx950 assembly per source line.

usage: line_hist.py kernel.s [file-substring]   (assembly from hipcc -S -gline-tables-only)
Weights: quarter-rate VALU (32-bit multiplies, 64-bit mad) count 4 issue slots, the rest 1.
Used to find where k_sweep's ~1000 VALU instructions per boundary go."""
import re
import sys
from collections import defaultdict

QUARTER = ('v_mul_lo_u32', 'v_mul_hi_u32', 'v_mul_hi_i32', 'v_mad_u64_u32', 'v_mad_i64_i32', 'v_mul_lo_i32')


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else 'k_sweep.h'
    files = {}
    cur = None
    hist = defaultdict(lambda: defaultdict(int))
    for line in open(path):
        s = line.strip()
        m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2))
            continue
        m = re.match(r'\.loc\s+(\d+)\s+(\d+)', s)
        if m:
            cur = (int(m.group(1)), int(m.group(2)))
            continue
        m = re.match(r'([vs]_[a-z0-9_]+|ds_[a-z0-9_]+|global_[a-z0-9_]+|flat_[a-z0-9_]+|scratch_[a-z0-9_]+|buffer_[a-z0-9_]+)\b', s)
        if m and cur:
            op = m.group(1)
            kind = 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'lds' if op.startswith('ds_') else 'vmem'
            if op.startswith('s_waitcnt'):
                kind = 'wait'
            w = 4 if op.startswith(QUARTER) else 1
            hist[cur][kind] += 1
            if kind == 'valu':
                hist[cur]['valu_w'] += w
    rows = []
    for (f, l), h in hist.items():
        rows.append((files.get(f, '?'), l, h))
    rows.sort(key=lambda r: (r[0], r[1]))
    tot = defaultdict(int)
    print('%-28s %6s %6s %6s %5s %5s %5s' % ('file:line', 'valu', 'valu_w', 'salu', 'lds', 'vmem', 'wait'))
    for f, l, h in rows:
        for k, v in h.items():
            tot[k] += v
        if want in f:
            print('%-28s %6d %6d %6d %5d %5d %5d' % (f.split('/')[-1] + ':' + str(l), h['valu'], h['valu_w'], h['salu'], h['lds'], h['vmem'], h['wait']))
    print('total', dict(tot))
    other = defaultdict(lambda: defaultdict(int))
    for f, l, h in rows:
        if want not in f:
            for k, v in h.items():
                other[f.split('/')[-1]][k] += v
    for f, h in other.items():
        print('  (other) %-24s' % f, dict(h))


if __name__ == '__main__':
    main()
