#!/bin/bash
# (GPU box, round 2 session F) rocprofv3 kernel trace + HBM counter passes of the bench command, configs[4]
# trace, FETCH_SIZE calibration for random 4-byte gathers
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > "$OUT/r02f_pytest.log" 2>&1; tail -3 "$OUT/r02f_pytest.log"
rm -rf "$OUT/prof_trace" "$OUT/prof_fetch" "$OUT/prof_write"
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/prof_trace" -o trace -- python "$REPO/bench.py" --steps 8 --warmup 2 --no-cpu-baseline --no-overlap --no-realism --no-cli > "$OUT/r02f_trace_bench.json" 2> "$OUT/prof_trace.log"
rocprofv3 --pmc FETCH_SIZE -d "$OUT/prof_fetch" -o fetch -- python "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-overlap --no-realism --no-cli > "$OUT/prof_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$OUT/prof_write" -o write -- python "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-overlap --no-realism --no-cli > "$OUT/prof_write.log" 2>&1
python "$REPO/tools/summarize_prof.py" "$OUT" 65536 40 300000 1 > "$OUT/r02f_rocprof_summary.txt" 2>&1
cat "$OUT/r02f_rocprof_summary.txt"; cat "$OUT/r02f_trace_bench.json"
# configs[4] shape: kernel trace + counters of the wide variant
rm -rf "$OUT/c5_trace" "$OUT/c5_fetch" "$OUT/c5_write"
rocprofv3 --kernel-trace --stats -d "$OUT/c5_trace" -o trace -- python "$REPO/tools/gpu_config5.py" --device-only > "$OUT/r02f_config5.txt" 2>&1
rocprofv3 --pmc FETCH_SIZE -d "$OUT/c5_fetch" -o fetch -- python "$REPO/tools/gpu_config5.py" --device-only > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$OUT/c5_write" -o write -- python "$REPO/tools/gpu_config5.py" --device-only > /dev/null 2>&1
python - <<'PY' > "$OUT/r02f_config5_rocprof_summary.txt" 2>&1
import glob, sqlite3, os
out = '/root/repo/gpurun_out'
for db in sorted(glob.glob(out + '/c5_trace/**/*.db', recursive=True)):
    con = sqlite3.connect(db)
    print('== rocprofv3 --kernel-trace --stats of tools/gpu_config5.py --device-only: name, calls, total_us, avg_us, pct')
    for name, calls, total, avg, pct in con.execute('select name,total_calls,total_duration,average,percentage from top_kernels'):
        print('  %-78s %5d %12.1f %10.1f %6.2f' % (name[:78], calls, total, avg, pct))
for sub in ('c5_fetch', 'c5_write'):
    for db in sorted(glob.glob(out + '/' + sub + '/**/*.db', recursive=True)):
        con = sqlite3.connect(db)
        print('== rocprofv3 --pmc (%s): kernel, counter, dispatches, avg value (KB)' % sub)
        for kn, cn, n, v in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                                         "where kernel_name like '%k_sweep%' or kernel_name like '%k_rnn%' group by kernel_name, counter_name"):
            print('  %-70s %-11s %4d %14.1f' % (kn[:70], cn, n, v))
PY
cat "$OUT/r02f_config5.txt" "$OUT/r02f_config5_rocprof_summary.txt"
# FETCH_SIZE calibration
rm -rf "$OUT/calib"
rocprofv3 --pmc FETCH_SIZE -d "$OUT/calib" -o calib -- "$REPO/build/micro/gather_calib" > "$OUT/r02f_gather_calib.txt" 2>&1
python - <<'PY' >> "$OUT/r02f_gather_calib.txt" 2>&1
import glob, sqlite3
for db in sorted(glob.glob('/root/repo/gpurun_out/calib/**/*.db', recursive=True)):
    con = sqlite3.connect(db)
    print('== FETCH_SIZE (KB) per dispatch, in launch order')
    for kn, v, d in con.execute("select kernel_name, value, duration from counters_collection order by dispatch_id"):
        print('  %-50s FETCH_SIZE_KB=%.0f' % (kn[:50], v))
PY
cat "$OUT/r02f_gather_calib.txt"
rm -rf "$OUT/calib" "$OUT/c5_trace" "$OUT/c5_fetch" "$OUT/c5_write"
