#!/bin/bash
# first Winograd GPU call: kernel parity tests, one-launch timings, whole-model check + bench
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/w1; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_wino.py -m gpu -q -x 2>&1 | tail -15
timeout 300 python tools/wino/time_wino.py 2>&1 | tail -12
timeout 600 python -m pytest tests/test_gpu_imdn.py tests/test_gpu_conv.py -m gpu -q -x 2>&1 | tail -5
timeout 300 python tools/quick_time.py -1 2>&1 | tail -14
