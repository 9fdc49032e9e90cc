"""Build a textual variant of csrc/esr_hip.hip -> tools/dbg/libesr_var_<name>.so
usage: make_var.py name 'old text' 'new text' ['old2' 'new2' ...]   (each old text must occur exactly once)"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(R, 'ntire2022_esr_amd/csrc/esr_hip.hip')).read()
name = sys.argv[1]
pairs = sys.argv[2:]
for a, b in zip(pairs[0::2], pairs[1::2]):
    a = a.encode().decode('unicode_escape'); b = b.encode().decode('unicode_escape')
    assert s.count(a) == 1, (s.count(a), a[:60])
    s = s.replace(a, b)
src = f'/tmp/esr_var_{name}.hip'
open(src, 'w').write(s)
csrc = os.path.join(R, 'ntire2022_esr_amd/csrc')
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC',
                       '-I', os.path.join(R, 'include'), '-I', csrc, '-o', os.path.join(R, f'tools/dbg/libesr_var_{name}.so'),
                       src, os.path.join(csrc, 'esr_esa.hip'), os.path.join(csrc, 'esr_bsconv.hip')])
print('built', name)
