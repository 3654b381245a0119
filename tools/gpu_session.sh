#!/bin/bash
# Runs ON THE GPU BOX through gpurun: ONE parameterised measurement session (it replaces the forty one-off
# tools/gpu_session_r0*.sh of rounds 2-3; they are in the git history).  Outputs go to gpurun_out/<TAG>_*; the
# summaries worth judging are then copied into profiles/ by hand.
#   gpurun --timeout 1500 -- 'bash tools/gpu_session.sh r04a tests quick trace traffic valu bench'
# steps (in the order given):
#   tests[=EXPR]   python -m pytest tests -m gpu -x -q [-k EXPR]                 -> <TAG>_pytest.log
#   quick          the headline only (every extra leg off, no CPU baseline)      -> <TAG>_quick.json
#   bench          the complete default `python bench.py`                        -> <TAG>_bench.json / .err / _time.txt
#   trace          rocprofv3 --kernel-trace --stats of the lean bench command    -> prof_trace/ (summarised by `traffic`)
#   traffic        separate --pmc FETCH_SIZE / WRITE_SIZE passes + summary       -> <TAG>_rocprof_summary.txt, traffic.json
#   valu           SQ instruction / activity counters of the hot kernels         -> <TAG>_pmc_summary.txt
#   variants       A/B of prebuilt build/libjppgpu_*.so (JPPGPU_LIB)             -> <TAG>_variants.txt
#   cfg5           the configs[4]-shape leg alone                                -> <TAG>_config5.json
#   trace5         rocprofv3 --kernel-trace of tools/gpu_config5_trace.py          -> <TAG>_config5_rocprof_summary.txt
#   traffic5       FETCH_SIZE / WRITE_SIZE passes of the same (configs[4] shape)   -> <TAG>_config5_traffic.txt
#   cli            the CLI end-to-end leg alone                                  -> <TAG>_cli.json
#   run=SCRIPT     any other helper under tools/ (python or bash), stdout        -> <TAG>_<script>.txt
# environment: BENCH_ARGS (extra bench.py arguments for quick/trace/traffic/valu/variants), STEPS (default 8)
set -u
TAG="$1"; shift
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
LEAN="--no-cpu-baseline --no-overlap --no-realism --no-cli --no-config5 --no-trainer"
STEPS="${STEPS:-8}"
BA="${BENCH_ARGS:-}"
summ() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('value %.0f  ms/step %.3f  kernels %s' % (d['value'], d['ms_per_step'], d['kernel_ms_per_step']))
    for k in ('roofline', 'roofline_valu', 'roofline_front', 'roofline_rnn', 'parity_sample'):
        if k in d: print(' ', k, json.dumps(d[k], ensure_ascii=False)[:400])
except Exception as e:
    print('no bench line:', e)
PY
}
for step in "$@"; do
  name="${step%%=*}"; arg=""; [ "$name" != "$step" ] && arg="${step#*=}"
  cd "$REPO"
  case "$name" in
    tests)
      if [ -n "$arg" ]; then timeout 1700 python -m pytest tests -m gpu -x -q -k "$arg" > "$OUT/${TAG}_pytest.log" 2>&1
      else timeout 1700 python -m pytest tests -m gpu -x -q > "$OUT/${TAG}_pytest.log" 2>&1; fi
      tail -4 "$OUT/${TAG}_pytest.log" ;;
    quick)
      timeout 600 python bench.py --steps "$STEPS" --warmup 2 $LEAN $BA > "$OUT/${TAG}_quick.json" 2> "$OUT/${TAG}_quick.err"
      summ "$OUT/${TAG}_quick.json" ;;
    bench)
      ( time timeout 1500 python bench.py $BA > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err" ) 2> "$OUT/${TAG}_bench_time.txt"
      tail -3 "$OUT/${TAG}_bench_time.txt"; summ "$OUT/${TAG}_bench.json" ;;
    trace)
      cd /tmp; rm -rf "$OUT/prof_trace"
      timeout 500 rocprofv3 --kernel-trace --stats -d "$OUT/prof_trace" -o trace -- python "$REPO/bench.py" --steps "$STEPS" --warmup 2 $LEAN --no-parity $BA > "$OUT/${TAG}_trace_bench.json" 2> "$OUT/prof_trace.log"
      summ "$OUT/${TAG}_trace_bench.json" ;;
    trace5)
      cd /tmp; rm -rf "$OUT/prof_trace"
      timeout 500 rocprofv3 --kernel-trace --stats -d "$OUT/prof_trace" -o trace -- python "$REPO/tools/gpu_config5_trace.py" > "$OUT/${TAG}_trace5.log" 2> "$OUT/prof_trace.log"
      python "$REPO/tools/summarize_prof.py" "$OUT" > "$OUT/${TAG}_config5_rocprof_summary.txt" 2>&1
      head -24 "$OUT/${TAG}_config5_rocprof_summary.txt"; tail -1 "$OUT/${TAG}_trace5.log"
      rm -rf "$OUT/prof_trace" ;;
    traffic5)
      # FETCH_SIZE / WRITE_SIZE of the configs[4] shape (tools/gpu_config5_trace.py), separate passes -> <TAG>_config5_traffic.txt
      cd /tmp; rm -rf "$OUT/prof_fetch" "$OUT/prof_write" "$OUT/prof_trace"
      timeout 500 rocprofv3 --pmc FETCH_SIZE -d "$OUT/prof_fetch" -o fetch -- python "$REPO/tools/gpu_config5_trace.py" > "$OUT/prof_fetch.log" 2>&1
      timeout 500 rocprofv3 --pmc WRITE_SIZE -d "$OUT/prof_write" -o write -- python "$REPO/tools/gpu_config5_trace.py" > "$OUT/prof_write.log" 2>&1
      python - "$OUT" > "$OUT/${TAG}_config5_traffic.txt" 2>&1 <<'PY'
import glob, os, sqlite3, sys
out = sys.argv[1]
per = {}
for sub, key in (('prof_fetch', 'FETCH_SIZE'), ('prof_write', 'WRITE_SIZE')):
    for db in glob.glob(os.path.join(out, sub, '**', '*.db'), recursive=True):
        con = sqlite3.connect(db)
        for kn, n, v, d in con.execute("select kernel_name, count(*), avg(value), avg(duration) from counters_collection where counter_name = ? group by kernel_name", (key,)):
            per.setdefault(kn, {})[key] = (n, v, d / 1e3)
print('== configs[4] shape (16 384 x 220 codepoints, beam 32, RNN): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; per launch')
print('   bytes = 2 x FETCH_SIZE KB (gfx950 tallies a 128-byte request at 64 B) + WRITE_SIZE KB')
rows = []
for kn, v in per.items():
    f = v.get('FETCH_SIZE', (0, 0.0, 0.0)); w = v.get('WRITE_SIZE', (0, 0.0, 0.0))
    rows.append((2 * f[1] * 1024 + w[1] * 1024, kn, f, w))
for b, kn, f, w in sorted(rows, reverse=True)[:14]:
    dur = f[2] or w[2]
    print('  %-72s %8.2f GB  (fetch %8.2f GB x2, write %7.2f GB)  %9.1f us  %5.2f TB/s' % (kn[:72], b / 1e9, f[1] * 1024 / 1e9, w[1] * 1024 / 1e9, dur, b / 1e12 / (dur * 1e-6) if dur else 0))
PY
      cat "$OUT/${TAG}_config5_traffic.txt"
      rm -rf "$OUT/prof_fetch" "$OUT/prof_write" ;;
    traffic)
      cd /tmp; rm -rf "$OUT/prof_fetch" "$OUT/prof_write"
      timeout 500 rocprofv3 --pmc FETCH_SIZE -d "$OUT/prof_fetch" -o fetch -- python "$REPO/bench.py" --steps 2 --warmup 1 $LEAN --no-parity $BA > "$OUT/prof_fetch.log" 2>&1
      timeout 500 rocprofv3 --pmc WRITE_SIZE -d "$OUT/prof_write" -o write -- python "$REPO/bench.py" --steps 2 --warmup 1 $LEAN --no-parity $BA > "$OUT/prof_write.log" 2>&1
      python "$REPO/tools/summarize_prof.py" "$OUT" $BA > "$OUT/${TAG}_rocprof_summary.txt" 2>&1
      head -14 "$OUT/${TAG}_rocprof_summary.txt"
      rm -rf "$OUT/prof_trace" "$OUT/prof_fetch" "$OUT/prof_write" ;;
    valu)
      cd /tmp; i=0
      for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES" \
                 "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
                 "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT" \
                 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"; do
        i=$((i+1)); rm -rf "$OUT/pmc_$i"
        timeout 400 rocprofv3 --pmc $grp -d "$OUT/pmc_$i" -o pmc -- python "$REPO/bench.py" --steps 2 --warmup 1 $LEAN --no-parity $BA > "$OUT/pmc_$i.log" 2>&1 || echo "pmc group $i failed: $grp"
      done
      python - "$OUT" "$REPO" $BA > "$OUT/${TAG}_pmc_summary.txt" 2>&1 <<'PY'
import glob, json, os, re, sqlite3, sys
out, repo = sys.argv[1], sys.argv[2]
print('== rocprofv3 --pmc passes of `python bench.py --steps 2 --warmup 1` (one pass per counter group): kernel, counter, dispatches, avg per launch')
per = {}
for db in sorted(glob.glob(os.path.join(out, 'pmc_*', '**', '*.db'), recursive=True)):
    con = sqlite3.connect(db)
    q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
         "where kernel_name like '%k_sweep%' or kernel_name like '%k_rnn%' or kernel_name like '%k_t0%' or kernel_name like '%k_seeds%' "
         "or kernel_name like '%k_norm%' or kernel_name like '%k_format%' group by kernel_name, counter_name")
    try:
        for kn, cn, n, v in con.execute(q):
            print('%-64s %-30s n=%d avg=%.6g' % (kn[:64], cn, n, v))
            m = re.search(r'(k_[a-z0-9_]+(<[^>]*>)?)', kn)
            if m:
                per.setdefault(m.group(1), {})[cn] = v
    except Exception as e:
        print('db', db, 'error', e)
# counters.json: what bench.py's roofline_valu reads (profiles/counters.json), stamped like traffic.json
sys.path.insert(0, repo)
import bench
a = bench.build_parser().parse_known_args(sys.argv[3:])[0]
json.dump({'batch': a.batch, 'sent_len': a.sent_len, 'dict_entries': a.dict_entries, 'weights_exp': a.weights_exp, 'rnn': bool(a.rnn),
           'kernel_source_id': bench.kernel_source_id(), 'kernels': per,
           'note': 'rocprofv3 --pmc, one pass per counter group, of `python bench.py --steps 2 --warmup 1` (lean legs); avg per launch'},
          open(os.path.join(out, 'counters.json'), 'w'), indent=1, sort_keys=True)
PY
      grep "k_sweep" "$OUT/${TAG}_pmc_summary.txt" | head -40
      for j in 1 2 3 4; do rm -rf "$OUT/pmc_$j"; done ;;
    variants)
      for so in build/libjppgpu_*.so; do
        echo "== $so"
        JPPGPU_LIB=$PWD/$so timeout 400 python bench.py --steps "$STEPS" --warmup 2 $LEAN --no-parity $BA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['kernel_ms_per_step'])"
      done 2>&1 | tee "$OUT/${TAG}_variants.txt" ;;
    cfg5)
      timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-parity --no-overlap --no-realism --no-cli --no-trainer $BA > "$OUT/${TAG}_config5.json" 2> "$OUT/${TAG}_config5.err"
      python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(json.dumps(d.get('config5'))[:1500])" "$OUT/${TAG}_config5.json" ;;
    cli)
      timeout 900 python bench.py --no-cpu-baseline --no-parity --no-overlap --no-realism --no-config5 --no-trainer $BA > "$OUT/${TAG}_cli.json" 2> "$OUT/${TAG}_cli.err"
      python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(json.dumps(d.get('cli_end_to_end'), ensure_ascii=False)[:3000])" "$OUT/${TAG}_cli.json" ;;
    run)
      base="$(basename "$arg" | cut -d. -f1)"
      case "$arg" in
        *.py|*.py\ *) timeout 1200 python $arg > "$OUT/${TAG}_$base.txt" 2>&1 ;;   # ("run=tools/x.py ARGS" quoted as one step)
        *) timeout 1200 bash $arg > "$OUT/${TAG}_$base.txt" 2>&1 ;;
      esac
      tail -30 "$OUT/${TAG}_$base.txt" ;;
    *) echo "unknown step $step" ;;
  esac
done
