#!/bin/bash
# (GPU box) SQ counters of the hot kernels, one rocprofv3 --pmc pass per counter group
export TMPDIR=/tmp
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d "$OUT/pmc_$i" -o pmc -- python "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > "$OUT/pmc_$i.log" 2>&1
done
python - <<'PY'
import glob, sqlite3, os
out = os.environ.get('OUT', os.path.join(os.getcwd()))
for db in sorted(glob.glob('/root/repo/gpurun_out/pmc_*/**/*.db', recursive=True)):
    con = sqlite3.connect(db)
    q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
         "where kernel_name like '%k_sweep%' or kernel_name like '%k_rnn%' or kernel_name like '%k_t0%' or kernel_name like '%k_seeds<1>%' "
         "group by kernel_name, counter_name")
    for kn, cn, n, v in con.execute(q):
        print('%-40s %-24s n=%d avg=%.4g' % (kn[:40], cn, n, v))
PY
