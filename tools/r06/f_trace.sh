#!/bin/bash
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r06zc; mkdir -p $O; cd $R
for v in t4 t4k2 t4k3 t4k5; do echo "== $v (t4: product; k2 k3 k5: waves skewed by 32 48 80 cycles x wave index behind every tile barrier)"; ESR_HIP_LIB=$R/tools/r06/libesr_$v.so timeout 200 python tools/r06/trace4_c64m.py 2>&1 | grep -v amdgpu.ids; done > $O/trace4.txt
cat $O/trace4.txt
