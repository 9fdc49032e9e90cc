#!/bin/bash
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r04f; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_wino.py -x -q 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_imdn.py tests/test_gpu_multi.py -x -q -k "imdn or f32" 2>&1 | tail -5
bash tools/gpu_ab_c1.sh 2
