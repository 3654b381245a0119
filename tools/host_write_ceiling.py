#!/usr/bin/env python3
"""(GPU box, developer tool; no GPU work) What the host can take: the file-to-file CLI ends in pwritev() calls that copy
the text of a batch into the page cache at sequenced offsets (host/jumanpp_gpu_main.cc, the writer stage).  This tool
does exactly that and nothing else -- T threads, each writing its own ranges of one output file from a buffer that is
already in memory -- for T = 1 .. cores, so that DESIGN section 6 can say where the ceiling of `jumanpp_gpu --devices=0-7
... -o file` lies however many GPUs feed it, and whether more writer threads or one file per device would lift it.
  python tools/host_write_ceiling.py [GB per run, default 4] [directory, default the bench cache]"""
import os
import sys
import tempfile
import threading
import time

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
where = sys.argv[2] if len(sys.argv) > 2 else os.path.join(tempfile.gettempdir(), 'jppgpu_bench_cache')
os.makedirs(where, exist_ok=True)
chunk = 8 << 20                      # one pwritev call (the CLI's helpers write ranges of a ~150 MB batch)
block = bytearray(os.urandom(1 << 20) * 8)
total = int(gb * (1 << 30)) // chunk * chunk
cores = len(os.sched_getaffinity(0))
print('cores %d, %.1f GB per run in %d MB calls, directory %s' % (cores, total / 2**30, chunk >> 20, where))


def run(threads, files):
    paths = [os.path.join(where, 'write_ceiling_%d.bin' % k) for k in range(files)]
    fds = [os.open(p, os.O_CREAT | os.O_TRUNC | os.O_WRONLY, 0o644) for p in paths]
    n_chunks = total // chunk
    nxt = [0]
    lock = threading.Lock()

    def work():
        while True:
            with lock:
                k = nxt[0]
                nxt[0] += 1
            if k >= n_chunks:
                return
            fd = fds[k % files]
            off = (k // files) * chunk
            done = 0
            while done < chunk:
                done += os.pwritev(fd, [memoryview(block)[done:]], off + done)
    ts = [threading.Thread(target=work) for _ in range(threads)]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    dt = time.perf_counter() - t0
    for fd in fds:
        os.close(fd)
    for p in paths:
        os.remove(p)
    return total / dt / 1e9


print('one output file (what the CLI writes):')
for t in [1, 2, 4, 8, 16, 32]:
    if t > 2 * cores:
        break
    print('  %2d writer threads: %6.2f GB/s' % (t, max(run(t, 1) for _ in range(2))))
print('one file per writer (an output shard per device):')
for t in [2, 4, 8, 16]:
    if t > 2 * cores:
        break
    print('  %2d writer threads, %2d files: %6.2f GB/s' % (t, t, max(run(t, t) for _ in range(2))))


def run_mapped(threads):
    """the same bytes into ONE file through a shared mapping: the file is grown with ftruncate, every thread copies its
    ranges into the mapping (page faults instead of write() calls: no lock on the inode)"""
    import ctypes
    import mmap
    path = os.path.join(where, 'write_ceiling_m.bin')
    fd = os.open(path, os.O_CREAT | os.O_TRUNC | os.O_RDWR, 0o644)
    os.ftruncate(fd, total)
    mm = mmap.mmap(fd, total, mmap.MAP_SHARED, mmap.PROT_READ | mmap.PROT_WRITE)
    base = ctypes.addressof(ctypes.c_char.from_buffer(mm))
    src = ctypes.addressof(ctypes.c_char.from_buffer(block))
    n_chunks = total // chunk
    nxt = [0]
    lock = threading.Lock()

    def work():
        while True:
            with lock:
                k = nxt[0]
                nxt[0] += 1
            if k >= n_chunks:
                return
            ctypes.memmove(base + k * chunk, src, chunk)
    ts = [threading.Thread(target=work) for _ in range(threads)]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    dt = time.perf_counter() - t0
    del base
    try:
        mm.close()
    except BufferError:
        pass
    os.close(fd)
    os.remove(path)
    return total / dt / 1e9


print('one output file through a shared mapping (ftruncate + memcpy):')
for t in [1, 2, 4, 8, 16]:
    if t > 2 * cores:
        break
    print('  %2d copying threads: %6.2f GB/s' % (t, max(run_mapped(t) for _ in range(2))))
print('JUMAN text is 2.36 KB per 40-codepoint sentence: 1 GB/s = 0.42 M sentences/s; lattice text (-s 32, 220 codepoints) 48 KB: 1 GB/s = 21 k sentences/s')
