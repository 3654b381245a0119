#!/usr/bin/env python3
"""(GPU box, developer tool) run-to-run spread of `jumanpp_gpu corpus -o file` on the bench workload, and what disturbs it.

  python tools/gpu_cli_probe.py            # the command on its own: back to back, --devices=0,0, pinned to the GPU's NUMA
                                           # node, after idle pauses                  (profiles/r04_ag_cli_probe.txt)
  python tools/gpu_cli_probe.py parent     # next to a parent process that first has no device state, then a torch
                                           # context, then a live analysis context, then destroys it
                                           #                                          (profiles/r04_ah_cli_probe2.txt)
"""
import gc
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def sh(c):
    try:
        return subprocess.run(c, shell=True, capture_output=True, text=True, timeout=20).stdout.strip()
    except Exception as e:
        return 'ERR %s' % e


def main():
    args = bench.build_parser().parse_args([])
    cache = args.cache
    mdic, model, img = bench.make_workload(args, cache)
    corpus = bench.make_corpus(args, mdic, cache, args.batch * 16, args.seed + 1)
    import __graft_entry__ as ge
    cli = ge.build_host()
    out = os.path.join(cache, 'cli_probe_out.txt')

    def run(tag, prefix=(), flags=()):
        t0 = time.perf_counter()
        p = subprocess.run(list(prefix) + [cli, '--model=' + model, '--batch=%d' % args.batch, '--timing', '-o', out, corpus] + list(flags),
                           capture_output=True, text=True)
        wall = time.perf_counter() - t0
        last = (p.stderr.strip().splitlines() or [''])[-1]
        m = re.search(r'sent_per_s=([0-9.e+]+)', last)
        g = re.search(r'gpu_ms=([0-9.]+)', last)
        print('%-44s %9.0f sentences/s  gpu_ms %s  process wall %.2f s' % (tag, float(m.group(1)) if m else 0, g.group(1) if g else '?', wall), flush=True)
        if os.path.exists(out):
            os.remove(out)

    if 'spread' in sys.argv[1:]:
        # (round 6) a session in which 8 of 11 device-text runs of the bench leg ran at 0.5 M sentences/s with five times
        # the GPU time per batch (profiles/r06_n_bench.json): the command back to back with the whole --timing output,
        # the memory state of the box before each run, and the three suspects switched off one at a time
        def meminfo():
            want = ('MemFree', 'Cached', 'Dirty', 'Writeback', 'Unevictable', 'Mlocked')
            kv = {}
            for line in open('/proc/meminfo'):
                k, v = line.split(':', 1)
                if k in want:
                    kv[k] = int(v.split()[0]) // 1024
            return ' '.join('%s=%dM' % (k, kv.get(k, -1)) for k in want)

        def cg(name):
            try:
                return open('/sys/fs/cgroup/' + name).read().strip().replace('\n', ' ')
            except OSError:
                return '?'

        def cgstat():
            cs = dict(x.split() for x in cg('cpu.stat').split('  ')) if False else {}
            raw = cg('cpu.stat').split()
            cs = dict(zip(raw[0::2], raw[1::2])) if len(raw) % 2 == 0 else {}
            ev = cg('memory.events').split()
            me = dict(zip(ev[0::2], ev[1::2])) if len(ev) % 2 == 0 else {}
            return {'throttled_usec': int(cs.get('throttled_usec', 0)), 'nr_throttled': int(cs.get('nr_throttled', 0)),
                    'mem_high': int(me.get('high', 0)), 'mem_max': int(me.get('max', 0)), 'mem_current_M': int(cg('memory.current') or 0) >> 20 if cg('memory.current').isdigit() else -1}
        print('cgroup: cpu.max=%s memory.max=%s memory.high=%s cpuset=%s' % (cg('cpu.max'), cg('memory.max'), cg('memory.high'), cg('cpuset.cpus.effective')[:60]), flush=True)

        def run2(tag, flags=(), env=None):
            before = cgstat()
            print('-- %s | %s' % (tag, meminfo()), flush=True)
            e = dict(os.environ)
            e.update(env or {})
            t0 = time.perf_counter()
            p = subprocess.run([cli, '--model=' + model, '--batch=%d' % args.batch, '--timing', '-o', out, corpus] + list(flags),
                               capture_output=True, text=True, env=e)
            wall = time.perf_counter() - t0
            keep = [ln for ln in p.stderr.strip().splitlines() if ln.startswith(('startup:', 'reserve:', 'devices=', 'batches:', 'exit:', 'prepin:'))]
            for ln in keep:
                print('   ' + ln[:260])
            after = cgstat()
            print('   process wall %.2f s | cgroup: cpu throttled %+d ms in %+d periods, memory.events high %+d max %+d, memory.current %d M' % (
                wall, (after['throttled_usec'] - before['throttled_usec']) // 1000, after['nr_throttled'] - before['nr_throttled'],
                after['mem_high'] - before['mem_high'], after['mem_max'] - before['mem_max'], after['mem_current_M']), flush=True)
            for f in [out] + [out + '.part%04d' % k for k in range(8)]:
                if os.path.exists(f):
                    os.remove(f)
        # r06_r: no CPU throttling, no memory events, and the ~4 s land anywhere -- inside the pipeline (a slow run) or in
        # the start-up / exit of a run whose pipeline was normal (process wall 5 s).  Suspect: the process BEFORE, which
        # leaves without tearing its device state down (--clean-exit restores the destructors)
        mode = [a for a in sys.argv[1:] if a != 'spread']
        for i in range(10):
            run2('default %d' % i)
        for i in range(10):
            run2('--clean-exit %d' % i, flags=['--clean-exit'])
        for i in range(6):
            time.sleep(3.0)
            run2('default after a 3 s pause %d' % i)
        return
    if 'parent' not in sys.argv[1:]:
        print(sh('lscpu | grep -i "numa\\|socket\\|^CPU(s)"'))
        print('gpu numa:', sh('cat /sys/class/drm/card*/device/numa_node'), '| nproc', sh('nproc'))
        print('cgroup cpus:', sh('cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null; cat /sys/fs/cgroup/cpu.max 2>/dev/null'))
        for i in range(5):
            run('back to back %d' % i)
        for i in range(3):
            run('--devices=0,0 %d' % i, flags=['--devices=0,0'])
        node = sh('cat /sys/class/drm/card*/device/numa_node | head -1')
        cpus = sh('cat /sys/devices/system/node/node%s/cpulist' % (node if node not in ('', '-1') else '0'))
        for i in range(3):
            run('taskset -c %s %d' % (cpus[:20], i), prefix=['taskset', '-c', cpus])
        time.sleep(12)
        run('after 12 s idle')
        run('right after')
        time.sleep(12)
        run('after 12 s idle, --devices=0,0', flags=['--devices=0,0'])
        run('right after, --devices=0,0', flags=['--devices=0,0'])
        return
    import numpy as np
    for i in range(3):
        run('no GPU state in the parent %d' % i)
    import torch
    import jumanpp_amd as J
    dev = torch.device('cuda', 0)
    x = torch.zeros(1 << 20, device=dev)
    torch.cuda.synchronize()
    for i in range(3):
        run('parent: torch context only %d' % i)
    batches = bench.load_batches(corpus, args.batch, np)
    ctx = J.Context(img, use_rnn=True)
    text, offs = batches[0]
    t = torch.frombuffer(bytearray(text), dtype=torch.uint8).to(dev)
    o = torch.from_numpy(offs.astype(np.int32)).to(dev)
    for it in range(6):
        r = ctx.analyze_device(t.data_ptr(), o.data_ptr(), len(offs) - 1, len(text), None)
        r.release()
    torch.cuda.synchronize()
    for i in range(4):
        run('parent: + live analysis context %d' % i)
    for it in range(40):
        r = ctx.analyze_device(t.data_ptr(), o.data_ptr(), len(offs) - 1, len(text), None)
        r.release()
    torch.cuda.synchronize()
    run('right after 40 parent batches')
    run('again')
    del ctx, r, x
    gc.collect()
    torch.cuda.empty_cache()
    for i in range(3):
        run('parent: context destroyed %d' % i)


if __name__ == '__main__':
    main()
