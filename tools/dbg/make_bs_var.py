"""Build a textual variant of csrc/esr_bsconv.hip -> tools/dbg/libesr_var_<name>.so
usage: make_bs_var.py name 'old' 'new' [...]"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
csrc = os.path.join(R, 'ntire2022_esr_amd/csrc')
s = open(os.path.join(csrc, 'esr_bsconv.hip')).read()
name = sys.argv[1]
for a, b in zip(sys.argv[2::2], sys.argv[3::2]):
    assert s.count(a) == 1, (s.count(a), a[:60])
    s = s.replace(a, b)
src = f'/tmp/esr_bs_{name}.hip'
open(src, 'w').write(s)
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-I', os.path.join(R, 'include'),
                       '-I', csrc, '-o', os.path.join(R, f'tools/dbg/libesr_var_{name}.so'), src, os.path.join(csrc, 'esr_hip.hip'), os.path.join(csrc, 'esr_esa.hip')])
print('built', name)
