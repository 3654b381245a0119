#!/bin/bash
# static instruction mix + register / LDS use of one kernel of the shipped library (developer tool)
#   tools/isa/kernel_stats.sh [substring of the mangled kernel name]
PAT="${1:-k_sweepILi8ELi64ELb1ELb1ELi5ELi2ELb0}"
LIB="$(dirname "$0")/../../jumanpp_amd/libjppgpu.so"
TMP=$(mktemp -d)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$LIB" --output="$TMP/dev.co" --unbundle 2>/dev/null || \
  /opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin="$TMP/fat.bin" "$LIB" 2>/dev/null
if [ ! -s "$TMP/dev.co" ]; then
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$TMP/fat.bin" --output="$TMP/dev.co" --unbundle
fi
/opt/rocm/lib/llvm/bin/llvm-objdump -d "$TMP/dev.co" > "$TMP/dev.s"
python3 - "$TMP/dev.s" "$PAT" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2]
blocks = re.split(r'\n(?=[0-9a-f]{16} <)', txt)
for b in blocks:
    head = b.split('\n', 1)[0]
    if pat in head and '>:' in head:
        ops = re.findall(r'^\s+([vs]_[a-z0-9_]+|ds_[a-z0-9_]+|global_[a-z0-9_]+|flat_[a-z0-9_]+|scratch_[a-z0-9_]+|buffer_[a-z0-9_]+)', b, re.M)
        kinds = {}
        for o in ops:
            k = 'wait' if o.startswith('s_waitcnt') else 'nop' if o.startswith('s_nop') else 'valu' if o.startswith('v_') else 'salu' if o.startswith('s_') else 'lds' if o.startswith('ds_') else 'vmem'
            kinds[k] = kinds.get(k, 0) + 1
        q = sum(1 for o in ops if o.startswith(('v_mul_lo_u32', 'v_mul_hi_u32', 'v_mad_u64_u32', 'v_mul_hi_i32', 'v_mad_i64_i32')))
        ex = sum(1 for o in ops if 'saveexec' in o or o in ('s_andn2_b64', 's_or_b64') )
        print(head[:110]); print('  static:', kinds, 'quarter-rate', q, 'exec-mask ops ~', ex, 'scratch', sum(1 for o in ops if o.startswith('scratch_')))
PY
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$TMP/dev.co" 2>/dev/null | python3 -c "
import sys,re
t=sys.stdin.read()
pat=sys.argv[1]
for m in re.finditer(r'\.name:\s+(\S+)', t):
    pass
# print the metadata entry of the kernel
i=t.find(pat)
while i>=0:
    j=t.rfind('- .agpr_count', 0, i); k=t.find('- .agpr_count', i)
    blk=t[j:k]
    if '.vgpr_count' in blk:
        for key in ('.vgpr_count','.sgpr_count','.group_segment_fixed_size','.private_segment_fixed_size','.vgpr_spill_count','.sgpr_spill_count'):
            mm=re.search(re.escape(key)+r':\s+(\d+)', blk)
            if mm: print('  ',key, mm.group(1))
        break
    i=t.find(pat,i+1)
" "$PAT"
rm -rf "$TMP"
