#!/bin/bash
# (GPU box, round 3 session A) state at the start of the round: GPU tests incl. the new headline-shape parity
# test, the default bench line (with parity_sample), stall-attribution counters of the bench command, the
# occupancy curve of k_sweep (dynamic-LDS padding: 4 / 3 / 2 wavefronts per SIMD), gather rates by cache level.
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
A="--no-cpu-baseline --no-overlap --no-realism --no-cli --no-config5"
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/r03a_pytest.log" 2>&1; tail -4 "$OUT/r03a_pytest.log"
( time timeout 900 python bench.py > "$OUT/r03a_bench.json" 2> "$OUT/r03a_bench.err" ) 2> "$OUT/r03a_bench_time.txt"; tail -3 "$OUT/r03a_bench_time.txt"; cut -c1-1500 "$OUT/r03a_bench.json"
build/micro/gather_levels > "$OUT/r03a_gather_levels.txt" 2>&1; cat "$OUT/r03a_gather_levels.txt"
# occupancy curve: 9840 B static LDS -> 16 / 12 / 8 workgroups (= wavefronts) per CU
for pad in 0 3400 9000; do
  JPPGPU_DEV_SWEEP_LDS_PAD=$pad timeout 300 python bench.py --steps 8 --warmup 2 $A 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('lds_pad $pad: sweep %.3f ms, value %.0f' % (j['kernel_ms_per_step']['sweep'], j['value']))" | tee -a "$OUT/r03a_occupancy.txt"
done
cd /tmp
rocprofv3 -L > "$OUT/r03a_counters_avail.txt" 2>&1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  rm -rf "$OUT/pmc_$i"
  timeout 300 rocprofv3 --pmc $grp -d "$OUT/pmc_$i" -o pmc -- python "$REPO/bench.py" --steps 2 --warmup 1 $A > "$OUT/pmc_$i.log" 2>&1 || echo "pmc group $i failed: $grp"
done
python - <<'PY' > "$OUT/r03a_pmc_summary.txt" 2>&1
import glob, sqlite3
print('== rocprofv3 --pmc passes of `python bench.py --steps 2 --warmup 1` (one pass per counter group): kernel, counter, dispatches, avg per launch')
for db in sorted(glob.glob('/root/repo/gpurun_out/pmc_*/**/*.db', recursive=True)):
    con = sqlite3.connect(db)
    q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
         "where kernel_name like '%k_sweep%' or kernel_name like '%k_rnn%' or kernel_name like '%k_t0%' or kernel_name like '%k_seeds%' or kernel_name like '%k_norm%' "
         "group by kernel_name, counter_name")
    try:
        for kn, cn, n, v in con.execute(q):
            print('%-60s %-30s n=%d avg=%.5g' % (kn[:60], cn, n, v))
    except Exception as e:
        print('db', db, 'error', e)
PY
grep "k_sweep" "$OUT/r03a_pmc_summary.txt"
for i in 1 2 3 4 5 6 7; do rm -rf "$OUT/pmc_$i"; done
