#!/bin/bash
# (GPU box, developer tool, round 6) SQ counters of k_lat_count / k_lat_write: the configs[4] command through the CLI, one
# pipeline, stages one after the other (--no-pipeline), 16 384 sentences          -> gpurun_out/<TAG>_lattice_counters.txt
set -u
TAG="${1:-r06}"
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
python - <<'PY' > "$OUT/lat_cmd.txt"
import os, sys, tempfile
sys.path.insert(0, os.getcwd())
import bench, __graft_entry__ as ge
a = bench.build_parser().parse_args([]); a.sent_len = 220
cache = os.path.join(tempfile.gettempdir(), 'jppgpu_bench_cache')
mdic, model, img = bench.make_workload(a, cache)
corpus = bench.make_corpus(a, mdic, cache, 16384, 31)
print(ge.build_host(), '--model=' + model, '--batch=8192', '--no-pipeline', '--pipelines-per-device=1', '--clean-exit', '--beam=32', '--global-beam=32',
      '--right-beam=32', '-s', '32', '-o', os.path.join(cache, 'latc_out.txt'), corpus)
PY
CMD="$(tail -1 "$OUT/lat_cmd.txt")"
cd /tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1)); rm -rf "$OUT/pmcl_$i"
  timeout 300 rocprofv3 --pmc $grp -d "$OUT/pmcl_$i" -o pmc -- $CMD > "$OUT/pmcl_$i.log" 2>&1 || echo "group $i failed"
done
python - "$OUT" > "$OUT/${TAG}_lattice_counters.txt" 2>&1 <<'PY'
import glob, os, sqlite3, sys
out = sys.argv[1]
per = {}
for db in sorted(glob.glob(os.path.join(out, 'pmcl_*', '**', '*.db'), recursive=True)):
    con = sqlite3.connect(db)
    for kn, cn, n, v, d in con.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                                       "where kernel_name like '%k_lat_%' group by kernel_name, counter_name"):
        per.setdefault(kn[:40], {})[cn] = v
        per[kn[:40]]['_us'] = d / 1e3
for kn, c in per.items():
    print('==', kn, '(%.0f us per launch of 8 192 sentences x 220 codepoints, N = 32)' % c.get('_us', 0))
    for k in sorted(c):
        if k != '_us':
            print('   %-26s %.6g' % (k, c[k]))
    v, us, w = c.get('SQ_INSTS_VALU'), c.get('_us', 0), c.get('SQ_WAVES')
    if v and us:
        print('   -> VALU issue %.2f of the slots; %.0f VALU wave-instructions per wavefront' % (v * 4 / (us * 1e-6 * 2.4e9 * 1024), v / max(1.0, w or 1)))
    if c.get('SQ_WAVE_CYCLES') and c.get('SQ_WAIT_ANY'):
        print('   -> wavefronts waiting %.0f %% of their cycles' % (100.0 * c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']))
        # (SQ_WAVE_CYCLES counts quad-cycles: MI355X_MICROARCH.md, PMC units)
        if us:
            print('   -> %.1f wavefronts per SIMD resident on average, each alive %.0f us' % (
                4.0 * c['SQ_WAVE_CYCLES'] / (us * 1e-6 * 2.4e9 * 1024), 4.0 * c['SQ_WAVE_CYCLES'] / max(1.0, w or 1) / 2.4e3))
PY
cat "$OUT/${TAG}_lattice_counters.txt"
for j in 1 2; do rm -rf "$OUT/pmcl_$j"; done
