#!/bin/bash
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r06p; mkdir -p $O; cd $R
for v in tr0; do echo "== $v"; ESR_HIP_LIB=$R/tools/r06/libesr_$v.so timeout 200 python tools/r06/trace_c64m.py 2>&1 | grep -v amdgpu.ids; done > $O/trace.txt
cat $O/trace.txt
