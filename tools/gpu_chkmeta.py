import ctypes, os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import jumanpp_amd as J, golden_io as G
lib = os.path.abspath('build/libjppgpu_chk.so')
ctx = J.Context('tests/golden/mini.img', lib_path=lib)
lines = [l.rstrip('\n') for l in open('tests/golden/mini.txt', encoding='utf-8')]
meta, gold = G.read_gold('tests/golden/mini.gold')
res = ctx.analyze(lines).fetch(full=True)
bad = sum(1 for s in range(len(lines)) if G.compare_sentence(res, s, gold[s], meta, verbose=False))
print('sentences with mismatches:', bad)
buf = (ctypes.c_ulonglong * 16)()
ctypes.CDLL(lib).jppgpu_debug_sweep_dbg(buf)
print("dbg =", list(buf)[:11])
print('n of that sentence:', len(lines[int(buf[1])]) if buf[0] else None)
