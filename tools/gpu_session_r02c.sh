#!/bin/bash
# (GPU box, round 2 session C) exact RNN arithmetic on the device: parity tests, tie audit, configs[4], bench
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
nproc > "$OUT/r02c_cpu.txt"; cat /sys/fs/cgroup/cpu.max >> "$OUT/r02c_cpu.txt" 2>&1; python -c "import os; print(len(os.sched_getaffinity(0)))" >> "$OUT/r02c_cpu.txt"
timeout 1200 python -m pytest tests -m gpu -x -q > "$OUT/r02c_pytest.log" 2>&1; tail -5 "$OUT/r02c_pytest.log"
timeout 600 python tools/rnn_tie_audit.py --bench-workload 5000 --verbose 40 > "$OUT/r02c_tie_audit.txt" 2>&1; tail -2 "$OUT/r02c_tie_audit.txt"
timeout 600 python tools/gpu_config5.py > "$OUT/r02c_config5.txt" 2>&1; cat "$OUT/r02c_config5.txt"
timeout 300 python tools/gpu_sweep_phases.py --rnn > "$OUT/r02c_phases_default.txt" 2>&1; cat "$OUT/r02c_phases_default.txt"
timeout 900 python bench.py --no-realism > "$OUT/r02c_bench.json" 2> "$OUT/r02c_bench.err"; tail -3 "$OUT/r02c_bench.err"; cat "$OUT/r02c_bench.json"
