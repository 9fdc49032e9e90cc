#!/bin/bash
# conv48r_kernel: correctness (16-bit kernel tests + networks) and the RLFN bench with / without it (ESR_NO_CONV48R=1)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_gpu_h16.py tests/test_gpu_big.py tests/test_gpu_multi.py -m gpu -x -q 2>&1 | tail -4
for v in 0 1 0 1; do if [ $v = 1 ]; then export ESR_NO_CONV48R=1; else unset ESR_NO_CONV48R; fi
  python bench.py --model team04_rlfn --compute bf16 --no-cpu-baseline --steps 30 > /tmp/b.json 2>/dev/null; python tools/show_bench.py /tmp/b.json | head -3 | cut -c1-150; done
