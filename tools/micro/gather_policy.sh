#!/bin/bash
# (GPU box) gather rate by cache policy and table size, then the request sizes the L2 sends to the fabric per variant
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; export TMPDIR=/tmp
build/micro/gather_policy
cd /tmp; rm -rf "$OUT/pmc_gp"
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_DRAM_sum -d "$OUT/pmc_gp" -o gp -- "$REPO/build/micro/gather_policy" > /dev/null 2>&1
python - "$OUT/pmc_gp" <<'PY'
import glob, os, sqlite3, sys
for db in sorted(glob.glob(os.path.join(sys.argv[1], '**', '*.db'), recursive=True)):
    con = sqlite3.connect(db)
    rows = list(con.execute("select kernel_name, counter_name, dispatch_id, value from counters_collection order by dispatch_id"))
    by = {}
    for kn, cn, did, v in rows:
        by.setdefault((did, kn), {})[cn] = by.setdefault((did, kn), {}).get(cn, 0) + v
    seen = {}
    for (did, kn), c in sorted(by.items()):
        n = seen.get(kn, 0); seen[kn] = n + 1
        if n % 3 == 2:   # third repetition of each (table, variant)
            print('%-30s table#%d  RDREQ %.4g  32B %.4g  64B %.4g  128B %.4g  DRAM %.4g  hit %.4g miss %.4g' % (
                kn[:30], n // 3, c.get('TCC_EA0_RDREQ_sum', 0), c.get('TCC_EA0_RDREQ_32B_sum', 0), c.get('TCC_EA0_RDREQ_64B_sum', 0),
                c.get('TCC_EA0_RDREQ_128B_sum', 0), c.get('TCC_EA0_RDREQ_DRAM_sum', 0), c.get('TCC_HIT_sum', 0), c.get('TCC_MISS_sum', 0)))
PY
rm -rf "$OUT/pmc_gp"
