#!/bin/bash
# (GPU box, round 2 session D) 24-bit index arithmetic, shared head-row weights, wide-variant staging: tests, bench, configs[4]
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > "$OUT/r02d_pytest.log" 2>&1; tail -5 "$OUT/r02d_pytest.log"
timeout 600 python bench.py --no-realism --no-cpu-baseline > "$OUT/r02d_bench.json" 2> "$OUT/r02d_bench.err"; tail -2 "$OUT/r02d_bench.err"; cat "$OUT/r02d_bench.json"
timeout 300 python tools/gpu_sweep_phases.py --rnn > "$OUT/r02d_phases_default.txt" 2>&1; tail -9 "$OUT/r02d_phases_default.txt"
timeout 300 python tools/gpu_sweep_phases.py --rnn --config5 > "$OUT/r02d_phases_config5.txt" 2>&1; tail -9 "$OUT/r02d_phases_config5.txt"
timeout 600 python tools/gpu_config5.py > "$OUT/r02d_config5.txt" 2>&1; cat "$OUT/r02d_config5.txt"
cd /tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM -d "$OUT/r02d_pmc" -o pmc -- python "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-realism --no-overlap > "$OUT/r02d_pmc.log" 2>&1
cd "$REPO"
python - <<'PY' > "$OUT/r02d_pmc_summary.txt" 2>&1
import glob, sqlite3
for db in sorted(glob.glob('/root/repo/gpurun_out/r02d_pmc/**/*.db', recursive=True)):
    con = sqlite3.connect(db)
    q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
         "where kernel_name like '%k_sweep%' or kernel_name like '%k_rnn%' or kernel_name like '%k_t0%' "
         "group by kernel_name, counter_name")
    for kn, cn, n, v in con.execute(q):
        print('%-60s %-24s n=%d avg=%.5g' % (kn[:60], cn, n, v))
PY
cat "$OUT/r02d_pmc_summary.txt"; rm -rf "$OUT/r02d_pmc"
