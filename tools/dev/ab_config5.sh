#!/bin/bash
# (GPU box, developer tool) A/B of prebuilt libraries on the configs[4] shape: per-kernel table of
# tools/gpu_config5_trace.py under rocprofv3 for every build/libjppgpu_<name>.so given
#   gpurun -- 'bash tools/dev/ab_config5.sh TAG base skew'
set -u
TAG="$1"; shift
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"; export TMPDIR=/tmp
for v in "$@"; do
  cd /tmp; rm -rf "$OUT/prof_trace"
  JPPGPU_LIB="$REPO/build/libjppgpu_$v.so" timeout 500 rocprofv3 --kernel-trace --stats -d "$OUT/prof_trace" -o trace -- python "$REPO/tools/gpu_config5_trace.py" > "$OUT/${TAG}_${v}_trace5.log" 2> "$OUT/prof_trace.log"
  python "$REPO/tools/summarize_prof.py" "$OUT" > "$OUT/${TAG}_${v}_config5_rocprof_summary.txt" 2>&1
  echo "== $v"; head -9 "$OUT/${TAG}_${v}_config5_rocprof_summary.txt" | cut -c1-130; tail -1 "$OUT/${TAG}_${v}_trace5.log"
  rm -rf "$OUT/prof_trace"
done
