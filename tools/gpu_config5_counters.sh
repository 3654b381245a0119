#!/bin/bash
# (GPU box, developer tool, round 6) SQ counters of the configs[4] shape (tools/gpu_config5_trace.py): is k_sweep<32,*>,
# which moves 2.85 TB/s of lines (profiles/r06_x_config5_traffic.txt: NOT the fabric ceiling of the headline sweep),
# bound by instruction issue, by LDS or by waiting?           -> gpurun_out/<TAG>_config5_counters.txt
set -u
TAG="${1:-r06}"
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1)); rm -rf "$OUT/pmc5_$i"
  timeout 400 rocprofv3 --pmc $grp -d "$OUT/pmc5_$i" -o pmc -- python "$REPO/tools/gpu_config5_trace.py" > "$OUT/pmc5_$i.log" 2>&1 || echo "group $i failed"
done
python - "$OUT" > "$OUT/${TAG}_config5_counters.txt" 2>&1 <<'PY'
import glob, os, sqlite3, sys
out = sys.argv[1]
per = {}
for db in sorted(glob.glob(os.path.join(out, 'pmc5_*', '**', '*.db'), recursive=True)):
    con = sqlite3.connect(db)
    for kn, cn, n, v, d in con.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                                       "where kernel_name like '%k_sweep<32%' or kernel_name like '%k_t0_memo%' or kernel_name like '%k_rnn_chain%' group by kernel_name, counter_name"):
        per.setdefault(kn[:60], {})[cn] = v
        per[kn[:60]]['_us'] = d / 1e3
for kn, c in per.items():
    print('==', kn, '(%.0f us per launch)' % c.get('_us', 0))
    for k in sorted(c):
        if k != '_us':
            print('   %-26s %.6g' % (k, c[k]))
    v, w, a, wc = c.get('SQ_INSTS_VALU'), c.get('SQ_WAVES'), c.get('SQ_ACTIVE_INST_VALU'), c.get('SQ_WAVE_CYCLES')
    us = c.get('_us', 0)
    if v and us:
        print('   -> VALU issue: %.3g wave-instructions x 4 cycles / (%.0f us x 2.4 GHz x 1024 SIMDs) = %.2f of the issue slots' % (v, us, v * 4 / (us * 1e-6 * 2.4e9 * 1024)))
    if c.get('SQ_THREAD_CYCLES_VALU') and a:
        print('   -> active lanes per VALU instruction: %.1f' % (c['SQ_THREAD_CYCLES_VALU'] / a / 4 if a else 0))
    if c.get('SQ_INSTS_LDS') and us:
        print('   -> LDS instructions: %.3g per launch' % c['SQ_INSTS_LDS'])
PY
cat "$OUT/${TAG}_config5_counters.txt"
for j in 1 2 3 4; do rm -rf "$OUT/pmc5_$j"; done
