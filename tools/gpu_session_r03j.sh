#!/bin/bash
# (GPU box, round 3 session J) GPU tests (incl. the table-driven spec path), the default bench line, the kernel trace and
# the HBM / SQ counter passes of the bench command (traffic.json stamped with the kernel source id), configs[4] shape
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
A="--no-cpu-baseline --no-overlap --no-realism --no-cli --no-config5"
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/r03j_pytest.log" 2>&1; tail -4 "$OUT/r03j_pytest.log"
cd /tmp
rm -rf "$OUT/prof_trace" "$OUT/prof_fetch" "$OUT/prof_write"
timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/prof_trace" -o trace -- python "$REPO/bench.py" --steps 8 --warmup 2 $A > "$OUT/r03j_trace_bench.json" 2> "$OUT/prof_trace.log"
timeout 400 rocprofv3 --pmc FETCH_SIZE -d "$OUT/prof_fetch" -o fetch -- python "$REPO/bench.py" --steps 2 --warmup 1 $A > "$OUT/prof_fetch.log" 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d "$OUT/prof_write" -o write -- python "$REPO/bench.py" --steps 2 --warmup 1 $A > "$OUT/prof_write.log" 2>&1
python "$REPO/tools/summarize_prof.py" "$OUT" 65536 40 300000 1 > "$OUT/r03j_rocprof_summary.txt" 2>&1
head -36 "$OUT/r03j_rocprof_summary.txt"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  rm -rf "$OUT/pmc_$i"
  timeout 300 rocprofv3 --pmc $grp -d "$OUT/pmc_$i" -o pmc -- python "$REPO/bench.py" --steps 2 --warmup 1 $A > "$OUT/pmc_$i.log" 2>&1 || echo "pmc group $i failed"
done
python - <<'PY' > "$OUT/r03j_pmc_summary.txt" 2>&1
import glob, sqlite3
print('== rocprofv3 --pmc passes of `python bench.py --steps 2 --warmup 1` (one pass per counter group): kernel, counter, dispatches, avg per launch')
for db in sorted(glob.glob('/root/repo/gpurun_out/pmc_*/**/*.db', recursive=True)):
    con = sqlite3.connect(db)
    q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
         "where kernel_name like '%k_sweep%' or kernel_name like '%k_rnn%' or kernel_name like '%k_t0%' or kernel_name like '%k_seeds%' or kernel_name like '%k_norm%' or kernel_name like '%k_decode%' or kernel_name like '%k_ends%' "
         "group by kernel_name, counter_name")
    try:
        for kn, cn, n, v in con.execute(q):
            print('%-60s %-30s n=%d avg=%.5g' % (kn[:60], cn, n, v))
    except Exception as e:
        print('db', db, 'error', e)
PY
grep "k_sweep" "$OUT/r03j_pmc_summary.txt" | head -30
for i in 1 2 3; do rm -rf "$OUT/pmc_$i"; done
rm -rf "$OUT/prof_trace" "$OUT/prof_fetch" "$OUT/prof_write"
cd "$REPO"
cp "$OUT/traffic.json" "$REPO/profiles/traffic.json"
( time timeout 1200 python bench.py > "$OUT/r03j_bench.json" 2> "$OUT/r03j_bench.err" ) 2> "$OUT/r03j_bench_time.txt"; tail -3 "$OUT/r03j_bench_time.txt"; cut -c1-3000 "$OUT/r03j_bench.json"
