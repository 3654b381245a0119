#!/bin/bash
# (GPU box) gather rate by cache policy, allocation flavour and table size, then the request sizes the L2 sends to the
# fabric per variant (two counter passes)
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; export TMPDIR=/tmp
build/micro/gather_policy
cd /tmp
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_EA0_RDREQ_DRAM_sum"; do
  rm -rf "$OUT/pmc_gp"
  timeout 300 rocprofv3 --pmc $grp -d "$OUT/pmc_gp" -o gp -- "$REPO/build/micro/gather_policy" > "$OUT/gp_pmc.log" 2>&1 || tail -5 "$OUT/gp_pmc.log"
  python - "$OUT/pmc_gp" <<'PY'
import glob, os, sqlite3, sys
for db in sorted(glob.glob(os.path.join(sys.argv[1], '**', '*.db'), recursive=True)):
    con = sqlite3.connect(db)
    rows = list(con.execute("select kernel_name, counter_name, dispatch_id, value from counters_collection order by dispatch_id"))
    by = {}
    for kn, cn, did, v in rows:
        d = by.setdefault((did, kn), {})
        d[cn] = d.get(cn, 0) + v
    seen = {}
    for (did, kn), c in sorted(by.items()):
        n = seen.get(kn, 0); seen[kn] = n + 1
        if n % 3 == 2:   # third repetition of each (allocation, table, variant)
            combo = n // 3
            print('%-22s alloc %d table 2^%d  ' % (kn[:22], combo // 2, 24 + 2 * (combo % 2)) + '  '.join('%s %.4g' % (k.replace('TCC_', '').replace('_sum', ''), v) for k, v in sorted(c.items())))
PY
done
rm -rf "$OUT/pmc_gp"
