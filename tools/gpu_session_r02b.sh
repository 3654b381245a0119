#!/bin/bash
# (GPU box, round 2 session B) A/B of k_sweep variants, SQ counters, new bench line (realism legs, CPU baseline)
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
lscpu > "$OUT/r02b_lscpu.txt" 2>&1
BENCH_ARGS="--no-realism --no-overlap" bash tools/gpu_variants.sh > "$OUT/r02b_variants.txt" 2>&1; cat "$OUT/r02b_variants.txt"
timeout 300 python tools/gpu_sweep_phases.py --rnn > "$OUT/r02b_phases_default.txt" 2>&1; cat "$OUT/r02b_phases_default.txt"
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -d "$OUT/r02b_pmc_$i" -o pmc -- python "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-realism --no-overlap > "$OUT/r02b_pmc_$i.log" 2>&1
done
cd "$REPO"
python - <<'PY' > "$OUT/r02b_pmc_summary.txt" 2>&1
import glob, sqlite3
for db in sorted(glob.glob('/root/repo/gpurun_out/r02b_pmc_*/**/*.db', recursive=True)):
    con = sqlite3.connect(db)
    q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
         "where kernel_name like '%k_sweep%' or kernel_name like '%k_rnn%' or kernel_name like '%k_t0%' or kernel_name like '%k_seeds%' "
         "group by kernel_name, counter_name")
    for kn, cn, n, v in con.execute(q):
        print('%-60s %-24s n=%d avg=%.5g' % (kn[:60], cn, n, v))
PY
cat "$OUT/r02b_pmc_summary.txt"
rm -rf "$OUT"/r02b_pmc_[0-9]
timeout 900 python bench.py > "$OUT/r02b_bench.json" 2> "$OUT/r02b_bench.err"; tail -3 "$OUT/r02b_bench.err"; cat "$OUT/r02b_bench.json"
