import os, subprocess, sys, tempfile
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_parity as tg, test_host_cli as th
ref_tools = os.path.abspath('oracle/_ref')
tmp = tempfile.mkdtemp()
img, lines, _ = tg._fresh_workload(ref_tools, tmp, 20000, 800, 18, 43)
data = th._make_partial_input(lines, 9)
pex = os.path.join(tmp, 'pex.txt'); open(pex, 'wb').write(data)
ref = th._ref_cli(ref_tools, os.path.join(tmp, 'w.model'), ['--partial-input'], pex)
cli = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.abspath('jumanpp_amd/bin/jumanpp_gpu')
out = subprocess.run([cli, '--model=' + img, '--partial-input', pex], capture_output=True).stdout
a, b = th._sentences(out), th._sentences(ref)
print('sentences', len(a), len(b))
bad = [i for i, (x, y) in enumerate(zip(a, b)) if x != y]
print('differing', len(bad), bad[:10])
exs = data.decode().split('\n\n')
for i in bad[:2]:
    print('--- example', i); print(exs[i])
    print('--- ours'); print(b'\n'.join(a[i]).decode()[:1500])
    print('--- ref'); print(b'\n'.join(b[i]).decode()[:1500])
